#!/usr/bin/env python3
"""bottleneck_layer data gradient (conv_bneck.hip) at the cfg2 size: option bneck = 1 (16 waves per block), 5 (4 waves), 0 (general dilated kernel);
cold (caches flushed) and hot.  python tools/probes/bneck_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip
dt = hip.BF16
B, H, W, C = 32, 16, 20, 2048
dz = torch.randn(B, 8, 10, 32, device="cuda").to(torch.bfloat16)
wd = (torch.randn(C, 3, 3, 32, device="cuda") / 17.0).to(torch.bfloat16)
bits = torch.randint(0, 256, (B * H * W * C // 8,), dtype=torch.uint8, device="cuda")
dx = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
g = hip.geom(B, 8, 10, 32, H, W, C, 3, 3, 1, 1, 2, 2, 2, 2)
scratch = torch.empty(600 << 20, dtype=torch.uint8, device="cuda")
for opt in (1, 5, 0, 1, 5, 0):
    with hip.options(bneck=opt):
        fn = lambda: hip.conv_igemm_ex(g, dt, hip.EPI_MASK_BITS, dz, wd, None, None, bits, dx)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        hot = e0.elapsed_time(e1) / 20 * 1e3
        cold = 0.0
        for _ in range(10):
            scratch.fill_(1); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); cold += e0.elapsed_time(e1)
        print("bneck=%d: hot %.1f us  cold %.1f us" % (opt, hot, cold / 10 * 1e3), flush=True)
