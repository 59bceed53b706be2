"""HIP-event times of the Dense-head shapes of cfg2 through urso_conv_igemm (dense_kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
dt = hip.BF16
junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
def t(fn, n=30):
    ts = []
    for _ in range(n):
        junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
torch.manual_seed(0)
for name, K, N, o32 in (("loc_dense_0 fwd", 2560, 1024, False), ("ori_final fwd", 1024, 4096, True), ("ori_final dgrad", 4096, 1024, False), ("dense_0 dgrad", 1024, 2560, False)):
    x = torch.randn(32, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / 30).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    y = torch.empty(32, N, dtype=torch.float32 if o32 else torch.bfloat16, device="cuda")
    g = hip.geom(32, 1, 1, K, 1, 1, N, 1, 1)
    fl = hip.EPI_OUT_F32 if o32 else hip.EPI_RELU
    f = lambda: hip.conv_igemm(g, dt, fl, x, w, b, None, None, y)
    f(); torch.cuda.synchronize()
    ref = x.float() @ w.float().T + b
    if not o32: ref = ref.relu()
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    print("%-18s K %4d N %4d: %.1f us   relerr %.1e  checksum %.6f" % (name, K, N, t(f), err, float(y.float().double().sum())))
