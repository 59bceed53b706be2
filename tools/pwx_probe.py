#!/usr/bin/env python3
"""Where conv_pwx.hip's time goes: launch time against K (slope = time per 64-deep K-step, intercept = launch + prologue + epilogue) with
the debug switches of option pwx_dbg (1 no copies after the prologue, 2 no MFMAs, 4 no epilogue).  python tools/pwx_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip
dt = hip.BF16
def run(B, H, W, K, N, bn, dbg, iters=30, rounds=3):
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda"); y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    g = hip.geom(B, H, W, K, H, W, N, 1, 1)
    best = 1e9
    with hip.options(pwx=2, pwx_bn=bn, pair=0, pwx_dbg=dbg):
        for r in range(rounds):
            for _ in range(3): hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, bias, None, None, y, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, bias, None, None, y, None)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
for (B, H, W, N, bn) in ((32, 32, 40, 256, 256), (32, 16, 20, 512, 128), (32, 16, 20, 512, 256)):
    for dbg in (0, 1, 2, 3, 4, 7):
        ts = [run(B, H, W, K, N, bn, dbg) for K in (128, 256, 512, 1024, 2048)]
        slope = (ts[4] - ts[2]) / 24.0
        print("M=%d N=%d bn=%d dbg=%d: K=128..2048: %s us   per K-step %.3f us, intercept %.1f us" % (
            B * H * W, N, bn, dbg, " ".join("%6.1f" % t for t in ts), slope, ts[2] - 8 * slope), flush=True)
