#!/usr/bin/env python3
"""c3v_kernel against c3w_kernel (option c3v = 0) at the cfg2 stage-3 size (32 x 64 x 80 x 128 -> 128, 3x3), hot loop and with the caches
flushed between launches.  The kernel's development switches are compile-time (conv_c3.hip: C3V_DBG 1 no MFMAs, 2 no fragment reads, 4 no
epilogue; C3V_SEQ, C3V_NOSB, C3V_PF): build a variant with URSO_LIB_VARIANT=<name> URSO_VARIANT_FLAGS="-DC3V_DBG=1" python -m ursonet_amd.build
and run this script under URSO_LIB_VARIANT=<name>.    python tools/probes/c3v_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip
dt = hip.BF16
B, H, W, C = 32, 64, 80, 128
x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
wf = (torch.randn(C, 3, 3, C, device="cuda") / (3 * C ** 0.5)).to(torch.bfloat16)
bias = torch.randn(C, device="cuda"); y = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
res = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
g = hip.geom(B, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1)
scratch = torch.empty(400 << 20, dtype=torch.uint8, device="cuda")
def run(c3v, dbg, mask, iters=20, rounds=3, cold=False):
    best = 1e9
    with hip.options(c3=3, c3v=c3v):
        fn = lambda: hip.conv_igemm(g, dt, hip.EPI_RELU if not mask else 0, x, wf, bias if not mask else None, None, res if mask else None, y)
        for r in range(rounds):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if cold:
                t = 0.0
                for _ in range(iters):
                    scratch.fill_(1); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); t += e0.elapsed_time(e1)
                best = min(best, t / iters * 1e3)
            else:
                e0.record()
                for _ in range(iters): fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
print("variant", os.environ.get("URSO_LIB_VARIANT", "(default)"))
print("c3w fwd %.1f us  masked %.1f us   (cold: %.1f)" % (run(0, 0, False), run(0, 0, True), run(0, 0, False, cold=True)))
print("c3v fwd %.1f us  masked %.1f us   (cold: %.1f)" % (run(1, 0, False), run(1, 0, True), run(1, 0, False, cold=True)), flush=True)
