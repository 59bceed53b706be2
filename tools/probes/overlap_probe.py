#!/usr/bin/env python3
"""Do an HBM-bound launch and an MFMA-bound launch finish sooner side by side on DISJOINT sets of CUs than one after the other on the whole chip?
Two branches of one captured hipGraph: branch 1 = NA stage-2 pointwise layers (64 -> 256 channels + residual + ReLU at 32 x 128 x 160: ~0.75 GB per
launch, HBM-bound) with option cus = A, branch 2 = NB stage-4 3x3 layers (256 -> 256 at 32 x 32 x 40, MFMA-bound) with cus = 256 - A; against the
same launches on one chain with the whole chip.    python tools/probes/overlap_probe.py [--na 8 --nb 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--na", type=int, default=8); ap.add_argument("--nb", type=int, default=20); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dt, tdt = hip.BF16, torch.bfloat16
dev = "cuda"
hip.set_option("hconv_streamk", 0)        # no hand-over spins between blocks that may not be co-resident

def layer(B, H, W, C, N, k):
    pad = k // 2
    x = torch.randn(B, H, W, C, device=dev).to(tdt)
    wf = (torch.randn(N, k, k, C, device=dev) / (k * k * C) ** 0.5).to(tdt)
    bias = torch.randn(N, device=dev)
    y = torch.empty(B, H, W, N, device=dev, dtype=tdt)
    res = torch.randn(B, H, W, N, device=dev).to(tdt) if k == 1 else None
    g = hip.geom(B, H, W, C, H, W, N, k, k, 1, 1, pad, pad)
    return lambda: hip.conv_igemm(g, dt, hip.EPI_RELU, x, wf, bias, res, None, y)

hbm = [layer(32, 128, 160, 64, 256, 1) for _ in range(2)]       # two sets of tensors: 1.5 GB, nothing stays in the 256 MB memory-side cache
mfma = [layer(32, 32, 40, 256, 256, 3) for _ in range(2)]

def with_cus(n, fn):
    hip.set_option("cus", n)
    try:
        fn()
    finally:
        hip.set_option("cus", 0)

def chain(cus_a, cus_b, forked):
    """Capture NA hbm launches + NB mfma launches: on one chain (forked False) or as two branches."""
    s = torch.cuda.Stream(); side = torch.cuda.Stream()
    def body():
        if not forked:
            for i in range(a.na): with_cus(cus_a, hbm[i & 1])
            for i in range(a.nb): with_cus(cus_b, mfma[i & 1])
            return
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for i in range(a.nb): with_cus(cus_b, mfma[i & 1])
        for i in range(a.na): with_cus(cus_a, hbm[i & 1])
        main.wait_stream(side)
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return g

def time_graph(g):
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps

def only(fns, n, cus):
    s = torch.cuda.Stream()
    def body():
        for i in range(n): with_cus(cus, fns[i & 1])
    with torch.cuda.stream(s): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    return time_graph(g)

print("# %d HBM-bound launches (1x1 64->256 + residual, 32x128x160) and %d MFMA-bound launches (3x3 256->256, 32x32x40), bf16, ms per graph replay" % (a.na, a.nb))
ta, tb = only(hbm, a.na, 0), only(mfma, a.nb, 0)
print("alone, whole chip:   hbm %.3f ms (%.1f us each)   mfma %.3f ms (%.1f us each)   one chain: %.3f ms" % (
    ta, ta * 1e3 / a.na, tb, tb * 1e3 / a.nb, time_graph(chain(0, 0, False))))
for A in (224, 192, 160, 128, 96):
    Bc = 256 - A
    t_a, t_b = only(hbm, a.na, A), only(mfma, a.nb, Bc)
    t_f = time_graph(chain(A, Bc, True))
    print("cus %3d | %3d:  hbm alone %.3f  mfma alone %.3f  side by side %.3f ms   (one chain on the whole chip %.3f)" % (A, Bc, t_a, t_b, t_f, ta + tb))
t_f = time_graph(chain(0, 0, True))
print("both on the whole chip, two branches: %.3f ms" % t_f)
