"""ursonet_amd -- MI355X-native implementation of UrsoNet's training/inference hot path.

    from ursonet_amd import net, config          # drop-in for the reference's net.py / config.py
    model = net.UrsoNet(mode="training", config=cfg, model_dir="logs")

Compute lives in liburso_hip.so (hand-written gfx950 HIP kernels, C ABI in include/ursonet_hip.h),
built in-tree by `python -m ursonet_amd.build`.  Sub-modules that launch kernels import
`ursonet_amd.hip`, which raises if the library is missing: there is no CPU fallback.
"""
__version__ = "0.1.0"
