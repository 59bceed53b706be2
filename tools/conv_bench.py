#!/usr/bin/env python3
"""Single-layer micro-benchmark of urso_conv_igemm / urso_conv_wgrad (HIP-event timed), for kernel work.
    python tools/conv_bench.py --shape B,H,W,C,N,k,s[,pad] --mode fwd|dgrad|wgrad [--res] [--mask] [--iters 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="32,128,160,64,256,1,1")
ap.add_argument("--mode", default="fwd"); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--res", action="store_true"); ap.add_argument("--mask", action="store_true")
ap.add_argument("--iters", type=int, default=20); ap.add_argument("--dbg", type=int, default=0)
a = ap.parse_args()
v = [int(x) for x in a.shape.split(",")]
B, H, W, C, N, k, s = v[:7]
pad = v[7] if len(v) > 7 else (k // 2)
dt = {"bf16": hip.BF16, "f32": hip.F32, "f16": hip.F16}[a.dtype]
tdt = hip.TORCH_DT[dt]
OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = torch.randn(B, H, W, C, device="cuda").to(tdt)
wf = (torch.randn(N, k, k, C, device="cuda") / (k * k * C) ** 0.5).to(tdt)
bias = torch.randn(N, device="cuda")
y = torch.empty(B, OH, OW, N, device="cuda", dtype=tdt)
res = torch.randn(B, OH, OW, N, device="cuda").to(tdt)
es = 2 if dt != hip.F32 else 4
if a.mode == "fwd":
    g = hip.geom(B, H, W, C, OH, OW, N, k, k, s, s, pad, pad)
    fn = lambda: hip.conv_igemm(g, dt, hip.EPI_RELU | a.dbg, x, wf, bias, res if a.res else None, res if a.mask else None, y)
    flops = 2.0 * B * OH * OW * N * k * k * C
    byts = (x.numel() + y.numel() * (1 + a.res + a.mask) + wf.numel()) * es
elif a.mode == "dgrad":
    wd = (torch.randn(C, k, k, N, device="cuda") / (k * k * N) ** 0.5).to(tdt)
    dx = torch.empty(B, H, W, C, device="cuda", dtype=tdt)
    g = hip.geom(B, OH, OW, N, H, W, C, k, k, 1, 1, k - 1 - pad, k - 1 - pad, s, s)
    fn = lambda: hip.conv_igemm(g, dt, 0, res, wd, None, x if a.res else None, x if a.mask else None, dx)
    flops = 2.0 * B * OH * OW * N * k * k * C
    byts = (res.numel() + x.numel() * (1 + a.res + a.mask) + wd.numel()) * es
else:
    g = hip.geom(B, H, W, C, OH, OW, N, k, k, s, s, pad, pad)
    ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 16, device="cuda")
    dw = torch.empty(k * k * C * N, device="cuda"); cs = torch.empty(N, device="cuda")
    fn = lambda: hip.conv_wgrad(g, dt, x, res, ws, dw, cs)
    flops = 2.0 * B * OH * OW * N * k * k * C
    byts = (x.numel() + res.numel()) * es + dw.numel() * 4
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print("%s %s %s: %.1f us  %.1f TFLOP/s  %.0f GB/s (algorithmic %.1f MB)" % (a.mode, a.shape, a.dtype, ms * 1e3, flops / ms / 1e9, byts / ms / 1e6, byts / 1e6))
