#!/usr/bin/env python3
"""Where conv_hwgrad.hip's time goes on the cfg2 3x3 layers: urso_conv_wgrad_partial with option hwgrad = 0 (general kernel), 1, and the
timing switches 3 (no MFMAs), 5 (no copies after the first tile), 7 (neither: launch + offset tables + barriers + partial store).
    python tools/hwgrad_probe.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursonet_amd import hip
dt = hip.BF16
for (B, H, W, C, N) in ((32, 32, 40, 256, 256), (32, 64, 80, 128, 128), (32, 16, 20, 512, 512)):
    x = torch.randn(B, H, W, C, device="cuda").bfloat16(); dz = torch.randn(B, H, W, N, device="cuda").bfloat16()
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    out = []
    for name, o in (("old", 0), ("hwg", 1), ("noMFMA", 3), ("noDMA", 5), ("neither", 7)):
        with hip.options(hwgrad=o):
            ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 64, device="cuda")
            for _ in range(3): hip.conv_wgrad_partial(g, dt, x, dz, ws)
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): hip.conv_wgrad_partial(g, dt, x, dz, ws)
            e1.record(); torch.cuda.synchronize(); out.append("%s %.1f" % (name, e0.elapsed_time(e1) / 20 * 1e3))
    print((B, H, W, C, N), "  ".join(out), flush=True)
