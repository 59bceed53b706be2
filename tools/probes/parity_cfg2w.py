"""bf16 whole-step parity at the cfg2 image size (2 x 512 x 640, ori_resolution 16) against the rounding-aware oracle: measured errors and the
magnitudes of the pre-activations at which device and oracle ReLU decisions differ (tests/test_model_gpu.py::test_training_step_parity_bf16_at_cfg2_width)."""
import sys, os, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
import test_model_gpu as T
import ursonet_amd.hip as hip
from oracle import graph_ref as G
for seed in (1, 2):
    cfg = make_config(dtype="bfloat16", backbone="resnet50", h=512, w=640, batch=2, regress_ori=False, ori_bins=16)
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=seed)
    eng, w0 = T._run_engine(cfg, img, loc, ori)
    dec = T.ReluDecisions(eng, tol=1.0)
    ref, newW = T._oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=G.StorageRounding(torch.bfloat16))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            T._compare_step(eng, ref, newW, 1.0, 1.0, 1.0)
        except AssertionError:
            pass
    print("cfg2 width, seed %d:" % seed, buf.getvalue().strip())
    print("   ReLU decisions: %d flipped of %d (%.1e); worst |pre-activation| / max at a flip %.2e; by magnitude [<1e-3 <3e-3 <1e-2 <3e-2 <1e-1 >=1e-1] %s" % (
        dec.flips, dec.total, dec.flips / dec.total, dec.worst, dec.histogram()), flush=True)
    del eng
