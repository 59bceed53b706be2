#!/usr/bin/env python3
"""Where conv_halo.hip's time goes on the cfg2 stage-4/5 3x3 layers (option hconv_dbg: 1 no epilogue, 2 no main loop, 4 whole tiles
instead of the (tile, chunk) stream-K schedule).  python tools/hconv_probe.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursonet_amd import hip
dt = hip.BF16
for (B, H, W, C, N) in ((32, 32, 40, 256, 256), (32, 16, 20, 512, 512)):
    x = torch.randn(B, H, W, C, device="cuda").bfloat16(); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, device="cuda")
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    out = []
    for name, o in (("full", 0), ("noEpi", 1), ("noLoop", 2), ("neither", 3), ("wholeTiles", 4), ("wholeTiles noEpi", 5),
                    ("loaders0-3", 16), ("noCopies", 32), ("noBarrier", 64), ("noCopies noBarrier", 96), ("loaders0-3 noBarrier", 80)):
        with hip.options(hconv_dbg=o):
            f = lambda: hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, b, None, None, y, None, ws)
            for _ in range(3): f()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize(); out.append("%s %.1f" % (name, e0.elapsed_time(e1) / 20 * 1e3))
    print((B, H, W, C, N), "  ".join(out), flush=True)
