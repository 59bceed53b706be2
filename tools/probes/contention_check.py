#!/usr/bin/env python3
"""Does a data-gradient launch give the same bits when other kernels run beside it?  The stride-2 pointwise data gradient with a scattered
destination and a mask tensor (the launch that differed under URSO_WGRAD_STREAM=2), alone and beside (a) a stream of big copies, (b) 3x3 weight
gradients, on a second stream.    python tools/probes/contention_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ursonet_amd import hip
dt, tdt = hip.BF16, torch.bfloat16
torch.manual_seed(0)
def case(B, H, W, C, N, k, s, mask=True, res=False):
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, H, W, C, device="cuda").to(tdt)                     # the forward input (mask) / gradient destination shape
    dz = torch.randn(B, OH, OW, N, device="cuda").to(tdt)
    wd = (torch.randn(C, k, k, N, device="cuda") / (k * k * N) ** 0.5).to(tdt)
    dx = torch.zeros(B, H, W, C, device="cuda", dtype=tdt)
    g = hip.geom(B, OH, OW, N, H, W, C, k, k, 1, 1, k - 1 - pad, k - 1 - pad, s, s)
    return (lambda: hip.conv_igemm(g, dt, 0, dz, wd, None, x if res else None, x if mask else None, dx)), dx
side = torch.cuda.Stream()
big_a, big_b = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"), torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
gw = hip.geom(32, 16, 20, 512, 16, 20, 512, 3, 3, 1, 1, 1, 1)
xw, dzw = torch.randn(32, 16, 20, 512, device="cuda").to(tdt), torch.randn(32, 16, 20, 512, device="cuda").to(tdt)
wsw = torch.empty(hip.conv_wgrad_ws_bytes(gw, dt) // 4 + 16, device="cuda"); dww = torch.empty(9 * 512 * 512, device="cuda"); csw = torch.empty(512, device="cuda")
def stress(kind, n):
    with torch.cuda.stream(side):
        for _ in range(n):
            if kind == "copy": big_a.copy_(big_b)
            else: hip.conv_wgrad(gw, dt, xw, dzw, wsw, dww, csw)
for name, args in (("s2 1x1 512<-128 scatter+mask 4x32x40", (4, 32, 40, 128, 512, 1, 2)), ("s2 1x1 scatter+mask 32x64x80", (32, 64, 80, 128, 512, 1, 2)),
                   ("s1 1x1 1024<-256 mask 32x32x40", (32, 32, 40, 1024, 256, 1, 1)), ("3x3 256 32x32x40", (32, 32, 40, 256, 256, 3, 1)),
                   ("s1 1x1 256<-64 +res 32x128x160", (32, 128, 160, 256, 64, 1, 1, False, True))):
    fn, out = case(*args)
    fn(); torch.cuda.synchronize(); ref = out.clone()
    for kind in ("quiet", "copy", "wgrad"):
        bad = 0
        for it in range(20):
            out.zero_() if "scatter" not in name else None
            torch.cuda.synchronize()
            if kind != "quiet": stress(kind, 6)
            fn()
            torch.cuda.synchronize()
            bad += int(not torch.equal(out, ref))
        print("%-40s beside %-6s: %d of 20 launches differ from the quiet result" % (name, kind, bad))
