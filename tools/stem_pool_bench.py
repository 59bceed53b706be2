#!/usr/bin/env python3
"""conv1 + max-pool: two launches against urso_stem_conv_pool_fwd at the cfg2 size.  python tools/stem_pool_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ursonet_amd import hip
dt = hip.BF16
B, H, W, N = 32, 512, 640, 64
molded = torch.randn(B, H, W, 4, device="cuda").bfloat16()
wf = (torch.randn(N * 224, device="cuda") / 12).bfloat16(); bias = torch.randn(N, device="cuda")
OH, OW = H // 2, W // 2
g = hip.geom(B, H, W // 2, 8, OH, OW, N, 7, 4, 2, 1, 3, 2)
y = torch.empty(B, OH, OW, N, device="cuda", dtype=torch.bfloat16)
p = torch.empty(B, OH // 2, OW // 2, N, device="cuda", dtype=torch.bfloat16); am = torch.empty(B, OH // 2, OW // 2, N, device="cuda", dtype=torch.uint8)
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
def two():
    hip.conv_igemm(g, dt, hip.EPI_RELU, molded, wf, bias, None, None, y); hip.maxpool_fwd(B, OH, OW, N, dt, y, p, am)
print("conv1 %.1f us  conv1 + maxpool %.1f us  fused %.1f us" % (
    t(lambda: hip.conv_igemm(g, dt, hip.EPI_RELU, molded, wf, bias, None, None, y)), t(two),
    t(lambda: hip.stem_conv_pool_fwd(g, dt, hip.EPI_RELU, molded, wf, bias, p, am))))
with hip.options(pwx_dbg=8):
    print("fused, pool arithmetic switched off (timing only) %.1f us" % t(lambda: hip.stem_conv_pool_fwd(g, dt, hip.EPI_RELU, molded, wf, bias, p, am)))
