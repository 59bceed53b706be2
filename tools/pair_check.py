#!/usr/bin/env python3
"""urso_conv_pair (conv_pair.hip) against the two urso_conv_igemm_ex launches it replaces, forward and backward form: comparison of both outputs (and of the emitted ReLU bit mask), then interleaved HIP-event timings at the cfg2 stage-2 size.
    python tools/pair_check.py [--dtype bf16] [--iters 20] [--rounds 5]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dt = {"bf16": hip.BF16, "f16": hip.F16}[a.dtype]
tdt = hip.TORCH_DT[dt]


def tensors(B, H, W, seed=0, c=64):
    torch.manual_seed(seed)
    M = B * H * W
    def T(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(tdt)
    d = dict(M=M, c=c, src=T(M, c), add=T(M, 4 * c), w1=T(4 * c, c, scale=c ** -0.5), w2=T(c, 4 * c, scale=0.5 * c ** -0.5),
             b1=torch.randn(4 * c, device="cuda"), b2=torch.randn(c, device="cuda"), act=T(M, c))
    d["g1"] = hip.geom(B, H, W, c, H, W, 4 * c, 1, 1, 1, 1, 0, 0)
    d["g2"] = hip.geom(B, H, W, 4 * c, H, W, c, 1, 1, 1, 1, 0, 0)
    return d


def separate(d, mode, mid, dst, bits):
    if mode == 0:
        hip.conv_igemm_ex(d["g1"], dt, hip.EPI_RELU | hip.EPI_EMIT_BITS, d["src"], d["w1"], d["b1"], d["add"], None, mid, bits, None)
        hip.conv_igemm_ex(d["g2"], dt, hip.EPI_RELU, mid, d["w2"], d["b2"], None, None, dst, None, None)
    else:
        hip.conv_igemm_ex(d["g1"], dt, hip.EPI_MASK_BITS, d["src"], d["w1"], None, d["add"], bits, mid, None, None)
        hip.conv_igemm_ex(d["g2"], dt, 0, mid, d["w2"], None, None, d["act"], dst, None, None)


def fused(d, mode, mid, dst, bits):
    if mode == 0:
        hip.conv_pair(d["M"], d["c"], dt, 0, d["src"], d["w1"], d["b1"], d["add"], bits, mid, d["w2"], d["b2"], None, dst)
    else:
        hip.conv_pair(d["M"], d["c"], dt, 1, d["src"], d["w1"], None, d["add"], bits, mid, d["w2"], None, d["act"], dst)


def check(B, H, W, cap=0, c=64):
    d = tensors(B, H, W, seed=B + H, c=c)
    M = d["M"]
    worst = 0
    for mode in (0, 1):
        outs = []
        bits_in = torch.randint(0, 256, (M, c // 2), device="cuda", dtype=torch.uint8)
        for fn in (separate, fused):
            mid = torch.full((M, 4 * c), 3.0, device="cuda").to(tdt); dst = torch.full((M, c), 3.0, device="cuda").to(tdt)
            bits = torch.zeros(M, c // 2, device="cuda", dtype=torch.uint8) if mode == 0 else bits_in
            with hip.options(grid_cap=cap if fn is fused else 0):
                fn(d, mode, mid, dst, bits)
            torch.cuda.synchronize()
            outs.append((mid.float(), dst.float(), bits.clone()))
        dm = float((outs[0][0] - outs[1][0]).abs().max()); dd = float((outs[0][1] - outs[1][1]).abs().max())
        db = int((outs[0][2] != outs[1][2]).sum()) if mode == 0 else 0
        print("c%d B%d %dx%d mode %d cap %d: mid diff %.3g  dst diff %.3g  bit-mask bytes differing %d   (|mid| max %.2f |dst| max %.2f)" % (
            c, B, H, W, mode, cap, dm, dd, db, float(outs[0][0].abs().max()), float(outs[0][1].abs().max())), flush=True)
        worst = max(worst, dm / float(outs[0][0].abs().max()), dd / float(outs[0][1].abs().max()), db / (M * c / 2.0) * 10)
    return worst


def bench(B, H, W, c=64):
    d = tensors(B, H, W, c=c)
    M = d["M"]
    mid = torch.empty(M, 4 * c, device="cuda", dtype=tdt); dst = torch.empty(M, c, device="cuda", dtype=tdt)
    bits = torch.randint(0, 256, (M, c // 2), device="cuda", dtype=torch.uint8)
    for mode in (0, 1):
        best = {}
        for r in range(a.rounds):
            for name, fn in (("separate", separate), ("fused", fused)):
                for _ in range(3): fn(d, mode, mid, dst, bits)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters): fn(d, mode, mid, dst, bits)
                e1.record(); torch.cuda.synchronize()
                best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / a.iters * 1e3)
        print("bench c%d B%d %dx%d mode %d: separate %.1f us   fused %.1f us" % (c, B, H, W, mode, best["separate"], best["fused"]), flush=True)


w = 0
for c in (64, 128):
    for (B, H, W, cap) in [(1, 8, 8, 0), (2, 16, 24, 0), (3, 40, 56, 0), (2, 64, 80, 8), (4, 128, 160, 0)]:
        w = max(w, check(B, H, W, cap, c))
print("worst difference", w)
bench(32, 128, 160, 64)
bench(32, 64, 80, 128)
assert w < 1e-2, "fused pair differs from the two separate launches by more than an output rounding step"
