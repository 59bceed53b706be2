#!/usr/bin/env python3
"""A/B of the register-resident-filter 3x3 kernel (conv_c3.hip, option c3=1) against the DMA kernel (conv_pw.hip, c3=0) in ONE process:
outputs on identical inputs (and a CPU fp32 reference), ragged image sizes, with and without the output mask; then interleaved timings
at the cfg2 / cfg5 stage-2 sizes.
    python tools/c3_check.py [--iters 30] [--rounds 5] [--dtype bf16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ursonet_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30); ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dt = {"bf16": hip.BF16, "f16": hip.F16}[a.dtype]
tdt = hip.TORCH_DT[dt]


def run(B, H, W, mask=False, relu=True, cap=0, C=64):
    torch.manual_seed(B * 1000 + H + W)
    x = torch.randn(B, H, W, C, device="cuda").to(tdt)
    wf = (torch.randn(C, 3, 3, C, device="cuda") / (3 * C ** 0.5)).to(tdt)
    bias = torch.randn(C, device="cuda")
    res = torch.randn(B, H, W, C, device="cuda").to(tdt)
    g = hip.geom(B, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1)
    outs = {}
    for c3 in (0, 1):
        y = torch.full((B, H, W, C), 7.0, device="cuda").to(tdt)
        with hip.options(c3=c3, hconv=0, grid_cap=cap if c3 else 0):
            hip.conv_igemm(g, dt, hip.EPI_RELU if relu else 0, x, wf, bias, None, res if mask else None, y)
        torch.cuda.synchronize()
        outs[c3] = y.float()
    z = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wf.float().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) + bias.cpu()
    if relu: z = F.relu(z)
    if mask: z = z * (res.float().cpu() > 0)
    e0 = float((outs[0].cpu() - z).abs().max() / z.abs().max()); e1 = float((outs[1].cpu() - z).abs().max() / z.abs().max())
    d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    print("C%d B%d %dx%d mask%d relu%d cap%d: new-vs-old %.2e   vs CPU fp32: old %.2e new %.2e" % (C, B, H, W, mask, relu, cap, d, e0, e1), flush=True)
    return max(d, e1)


def bench(B, H, W, mask=False, C=64):
    x = torch.randn(B, H, W, C, device="cuda").to(tdt)
    wf = (torch.randn(C, 3, 3, C, device="cuda") / (3 * C ** 0.5)).to(tdt)
    bias = torch.randn(C, device="cuda")
    res = torch.randn(B, H, W, C, device="cuda").to(tdt)
    y = torch.empty(B, H, W, C, device="cuda", dtype=tdt)
    g = hip.geom(B, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1)
    flops = 2.0 * B * H * W * C * 9 * C
    best = {0: 1e9, 1: 1e9}
    for r in range(a.rounds):
        for c3 in (0, 1):
            with hip.options(c3=c3, hconv=1):
                fn = lambda: hip.conv_igemm(g, dt, hip.EPI_RELU, x, wf, bias, None, res if mask else None, y)
                for _ in range(3): fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters): fn()
                e1.record(); torch.cuda.synchronize()
                best[c3] = min(best[c3], e0.elapsed_time(e1) / a.iters)
    print("bench C%d B%d %dx%d mask%d: other kernel %.1f us (%.0f TF)   register filter %.1f us (%.0f TF)" % (
        C, B, H, W, mask, best[0] * 1e3, flops / best[0] / 1e9, best[1] * 1e3, flops / best[1] / 1e9), flush=True)


worst = 0.0
for C in (64, 128):
    for (B, H, W, cap) in [(1, 4, 32, 0), (2, 8, 64, 0), (1, 5, 33, 0), (3, 17, 23, 0), (2, 30, 70, 8), (2, 64, 96, 8), (4, 64, 80, 0)]:
        for (mask, relu) in ((False, True), (True, False)):
            worst = max(worst, run(B, H, W, mask=mask, relu=relu, cap=cap, C=C))
print("worst %.3e" % worst)
bench(32, 128, 160); bench(32, 128, 160, mask=True); bench(32, 160, 240)
bench(32, 64, 80, C=128); bench(32, 64, 80, mask=True, C=128); bench(32, 80, 120, C=128)
assert worst < 2e-2
