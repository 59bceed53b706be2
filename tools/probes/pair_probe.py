#!/usr/bin/env python3
"""The single c -> 4c layers of stages 4 / 5 on the register-filter kernel (conv_pair.hip pair_kernel, VAR 1: + residual, ReLU, bit mask out)
at the cfg2 sizes, hot loop and with the caches flushed, per compile-time variant (PAIR_DBG: 1 no MFMAs, 2 no stores, 4 no add DMA,
8 no epilogue arithmetic).  Build: URSO_LIB_VARIANT=<name> URSO_VARIANT_FLAGS="-DPAIR_DBG=n" python -m ursonet_amd.build; run under
URSO_LIB_VARIANT=<name>.    python tools/probes/pair_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip
dt = hip.BF16
for kv in os.environ.get("URSO_OPTS", "").split():          # e.g. URSO_OPTS="pair_s4=1"
    hip.set_option(kv.split("=")[0], int(kv.split("=")[1]))
scratch = torch.empty(600 << 20, dtype=torch.uint8, device="cuda")
out = []
for (B, H, W, C, N) in [(32, 32, 40, 256, 1024), (32, 16, 20, 512, 2048)]:
    x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(N, C, device="cuda") / C ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda"); y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
    bits = torch.empty(B * H * W * N // 8, dtype=torch.uint8, device="cuda")
    g = hip.geom(B, H, W, C, H, W, N, 1, 1)
    fn = lambda: hip.conv_igemm_ex(g, dt, hip.EPI_RELU | hip.EPI_EMIT_BITS, x, wf, bias, res, None, y, bits)
    best, cold = 1e9, 0.0
    for r in range(3):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    for _ in range(10):
        scratch.fill_(1); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); cold += e0.elapsed_time(e1)
    out.append("%dx%dx%d %d->%d: hot %.1f us cold %.1f us" % (B, H, W, C, N, best, cold / 10 * 1e3))
print("variant %-8s %-12s %s" % (os.environ.get("URSO_LIB_VARIANT", "(default)"), os.environ.get("URSO_OPTS", ""), "   ".join(out)), flush=True)
