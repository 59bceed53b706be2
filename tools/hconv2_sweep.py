#!/usr/bin/env python3
"""Which kernel / tile shape is fastest for the 3x3 halo layers of the BASELINE configs: conv_halo.hip (with its hand-over workspace) against every
conv_halo2.hip tile shape that fits, and what the cost model (option hconv2 = 1) picks.   python tools/hconv2_sweep.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursonet_amd import hip
dt = hip.BF16
LAYERS = [("cfg2 stage 4", 32, 32, 40, 256, 256), ("cfg2 stage 5", 32, 16, 20, 512, 512), ("cfg2 stage 3", 32, 64, 80, 128, 128),
          ("cfg4 stage 4", 16, 32, 40, 256, 256), ("cfg4 stage 5", 16, 16, 20, 512, 512), ("cfg4 stage 3", 16, 64, 80, 128, 128),
          ("cfg5 stage 4", 32, 40, 60, 256, 256), ("cfg5 stage 5", 32, 20, 30, 512, 512), ("B8 stage 4", 8, 32, 40, 256, 256), ("B8 stage 5", 8, 16, 20, 512, 512)]


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (name, B, H, W, C, N) in LAYERS:
    x = torch.randn(B, H, W, C, device="cuda").bfloat16(); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, device="cuda")
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    f = lambda: hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, b, None, None, y, None, ws)
    res = {}
    with hip.options(hconv=2, hconv2=0, c3=0):
        res["halo1"] = timeit(f)
    with hip.options(hconv=0, c3=0):
        res["dma"] = timeit(f)
    for shp in (32, 31, 22, 21, 12, 11):
        with hip.options(hconv=2, hconv2=2, hconv2_shape=shp, c3=0):
            if hip.conv_igemm_halo2_shape(g, dt, hip.EPI_RELU, False, True) == shp:
                res[str(shp)] = timeit(f)
    with hip.options(hconv=2, c3=0):
        pick = hip.conv_igemm_halo2_shape(g, dt, hip.EPI_RELU, False, True)
    best = min(res, key=res.get)
    flops = 2.0 * B * H * W * N * 9 * C
    print("%-13s %s  pick %s  best %s (%.0f TF)" % (name, "  ".join("%s %.1f" % kv for kv in res.items()), pick or "halo1", best, flops / res[best] / 1e6), flush=True)
