#!/usr/bin/env python3
"""A/B of the halo-tile conv kernel (conv_halo.hip, option hconv=1) against the DMA kernel (conv_pw.hip, hconv=0) in ONE process:
outputs of the two kernels on identical inputs (and a CPU fp32 reference on small shapes), then interleaved HIP-event timings.
    python tools/hconv_check.py [--iters 30] [--rounds 5] [--dtype bf16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ursonet_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30); ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--skip_ref", action="store_true")
a = ap.parse_args()
dt = {"bf16": hip.BF16, "f16": hip.F16}[a.dtype]
tdt = hip.TORCH_DT[dt]


def run(B, H, W, C, N, mask=False, add=False, relu=True, ref=False):
    torch.manual_seed(B * 1000 + H + C)
    x = torch.randn(B, H, W, C, device="cuda").to(tdt)
    wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).to(tdt)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(B, H, W, N, device="cuda").to(tdt)
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    outs = {}
    for hc in (0, 1):
        y = torch.full((B, H, W, N), 7.0, device="cuda").to(tdt)
        with hip.options(hconv=2 if hc else 0):
            hip.conv_igemm(g, dt, hip.EPI_RELU if relu else 0, x, wf, bias, res if add else None, res if mask else None, y)
        torch.cuda.synchronize()
        outs[hc] = y.float()
    d = float((outs[0] - outs[1]).abs().max() / (outs[0].abs().max() + 1e-30))
    msg = "B%d %dx%d C%d N%d add%d mask%d: new-vs-old %.2e" % (B, H, W, C, N, add, mask, d)
    if ref:
        z = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wf.float().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) + bias.cpu()
        if add:
            z = z + res.float().cpu()
        if relu:
            z = F.relu(z)
        if mask:
            z = z * (res.float().cpu() > 0)
        e0 = float((outs[0].cpu() - z).abs().max() / z.abs().max()); e1 = float((outs[1].cpu() - z).abs().max() / z.abs().max())
        msg += "  vs CPU fp32: old %.2e new %.2e" % (e0, e1)
    print(msg, flush=True)
    return d


def bench(B, H, W, C, N, mask=False):
    x = torch.randn(B, H, W, C, device="cuda").to(tdt)
    wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).to(tdt)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(B, H, W, N, device="cuda").to(tdt)
    y = torch.empty(B, H, W, N, device="cuda", dtype=tdt)
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    flops = 2.0 * B * H * W * N * 9 * C
    best = {0: 1e9, 1: 1e9}
    for r in range(a.rounds):
        for hc in (0, 1):
            with hip.options(hconv=2 if hc else 0):
                fn = lambda: hip.conv_igemm(g, dt, hip.EPI_RELU, x, wf, bias, None, res if mask else None, y)
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best[hc] = min(best[hc], e0.elapsed_time(e1) / a.iters)
    print("bench B%d %dx%d C%d N%d mask%d: dma %.1f us (%.0f TF)  halo %.1f us (%.0f TF)" % (
        B, H, W, C, N, mask, best[0] * 1e3, flops / best[0] / 1e9, best[1] * 1e3, flops / best[1] / 1e9), flush=True)


worst = 0.0
for (B, H, W, C, N) in [(2, 8, 10, 64, 128), (3, 16, 20, 128, 128), (2, 32, 40, 256, 256), (5, 17, 23, 64, 256), (1, 64, 80, 128, 128)]:
    for (add, mask) in ((False, False), (True, True)):
        worst = max(worst, run(B, H, W, C, N, mask=mask, add=add, ref=not a.skip_ref))
for (B, H, W, C, N) in [(32, 64, 80, 128, 128), (32, 32, 40, 256, 256), (32, 16, 20, 512, 512)]:
    worst = max(worst, run(B, H, W, C, N))
print("worst new-vs-old %.3e" % worst)
for (B, H, W, C, N) in [(32, 64, 80, 128, 128), (32, 32, 40, 256, 256), (32, 16, 20, 512, 512)]:
    bench(B, H, W, C, N)
    bench(B, H, W, C, N, mask=True)
assert worst < 2e-2, "halo kernel disagrees with the DMA kernel"
