#!/usr/bin/env python3
"""Where does the halo conv kernel's time go?  Launch / per-tile / per-step costs from shapes with exactly 1, 2, 3 tiles per block
(Vh = Vw = 32: 4 tiles per image at N = 128) and 18 / 36 / 72 steps per tile, with the epilogue or the main loop switched off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip
dt, tdt = hip.BF16, torch.bfloat16


def t_us(B, H, W, C, N, dbg, iters=30, rounds=3):
    x = torch.randn(B, H, W, C, device="cuda").to(tdt)
    wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).to(tdt)
    bias = torch.randn(N, device="cuda")
    y = torch.empty(B, H, W, N, device="cuda", dtype=tdt)
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    best = 1e9
    with hip.options(hconv=2, hconv_dbg=dbg):
        fn = lambda: hip.conv_igemm(g, dt, hip.EPI_RELU, x, wf, bias, None, None, y)
        for r in range(rounds):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


print("tiles/block steps/tile   full   no-epilogue   no-mainloop   neither")
for C in (128, 256, 512):
    for B in (64, 128, 192):
        r = [t_us(B, 31, 31, C, 128, d) for d in (0, 1, 2, 3)]
        print("%5d %10d   %6.1f %10.1f %12.1f %10.1f" % (B // 64, 9 * C // 64, r[0], r[1], r[2], r[3]), flush=True)
