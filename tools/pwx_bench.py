#!/usr/bin/env python3
"""A/B of conv_pwx.hip against conv_pw.hip / conv_pair.hip on the pointwise layers of cfg2 stages 4-5, interleaved rounds in ONE process
(HIP events around `iters` back-to-back launches; a 300 MB scratch write between rounds evicts L2 / MALL).
    python tools/pwx_bench.py [--rounds 5 --iters 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
dt = hip.BF16
B = a.batch
# name, H, W, K, N, form
LAYERS = [("s4 2a fwd (relu)", 32, 40, 1024, 256, "relu"), ("s4 2c dgrad (mask tensor)", 32, 40, 1024, 256, "mask"),
          ("s5 2a fwd (relu)", 16, 20, 2048, 512, "relu"), ("s5 2c dgrad (mask tensor)", 16, 20, 2048, 512, "mask"),
          ("s4a b1 fwd (plain)", 32, 40, 512, 1024, "plain"), ("s5a b1 fwd (plain)", 16, 20, 1024, 2048, "plain"),
          ("s4a 2a fwd (relu)", 32, 40, 512, 256, "relu"), ("s5a 2a fwd (relu)", 16, 20, 1024, 512, "relu"),
          ("s4 2c fwd (add relu bits)", 32, 40, 256, 1024, "addbits"), ("s4 2a dgrad (add maskbits)", 32, 40, 256, 1024, "addmb"),
          ("s5 2c fwd (add relu bits)", 16, 20, 512, 2048, "addbits"), ("s5 2a dgrad (add maskbits)", 16, 20, 512, 2048, "addmb")]
scratch = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device="cuda")
for name, H, W, K, N, form in LAYERS:
    M = B * H * W
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
    bits = torch.empty(M * N // 8, dtype=torch.uint8, device="cuda")
    mb = torch.randint(0, 256, (M * N // 8,), dtype=torch.uint8, device="cuda")
    g = hip.geom(B, H, W, K, H, W, N, 1, 1)
    if form == "relu": fn = lambda: hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, bias, None, None, y, None)
    elif form == "plain": fn = lambda: hip.conv_igemm_ex(g, dt, 0, x, wf, bias, None, None, y, None)
    elif form == "mask": fn = lambda: hip.conv_igemm_ex(g, dt, 0, x, wf, None, None, res, y, None)
    elif form == "addbits": fn = lambda: hip.conv_igemm_ex(g, dt, hip.EPI_RELU | hip.EPI_EMIT_BITS, x, wf, bias, res, None, y, bits)
    else: fn = lambda: hip.conv_igemm_ex(g, dt, hip.EPI_MASK_BITS, x, wf, None, res, mb, y, None)
    flops = 2.0 * M * N * K
    byts = 2.0 * (M * K + M * N * (2 if form in ("mask", "addbits", "addmb") else 1) + N * K)
    variants = [("old", dict(pwx=0)), ("pwx256", dict(pwx=2, pwx_bn=256, pair=0)), ("pwx128", dict(pwx=2, pwx_bn=128, pair=0)), ("auto", dict(pwx=1))]
    best = {k: 1e9 for k, _ in variants}
    for r in range(a.rounds):
        for k, opt in variants:
            with hip.options(**opt):
                fn(); fn()
                scratch.fill_(r)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best[k] = min(best[k], e0.elapsed_time(e1) / a.iters * 1e3)
    print("%-28s M=%6d K=%4d N=%4d  %s   (roofs: %.1f us MFMA, %.1f us HBM@6.3)" % (
        name, M, K, N, "  ".join("%s %6.1f us %6.0f TF" % (k, best[k], flops / best[k] / 1e6) for k, _ in variants),
        flops / 2.5e9, byts / 6.3e6), flush=True)
