#!/usr/bin/env python3
"""conv_halo2.hip against conv_halo.hip on the cfg2 stage-4 / stage-5 3x3 layers: results (bit-compared with conv_halo.hip's whole-tile schedule,
which sums in the same order) and time per launch for every tile shape that fits, with the kernel's timing switches (option hconv_dbg: 1 no
epilogue, 2 no step loop, 32 no copies in the loop, 64 no barrier in the loop).   python tools/hconv2_probe.py [--shapes 32,31] [--quick]"""
import sys, os, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursonet_amd import hip
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="32,31,22,21"); ap.add_argument("--quick", action="store_true"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dt = hip.BF16


def timeit(f, n=a.iters):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W, C, N) in ((a.batch, 32, 40, 256, 256), (a.batch, 16, 20, 512, 512)):
    torch.manual_seed(5)
    x = torch.randn(B, H, W, C, device="cuda").bfloat16(); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); msk = torch.randn(B, H, W, N, device="cuda").bfloat16()
    ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, device="cuda")
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    flops = 2.0 * B * H * W * N * 9 * C
    y0 = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16); y0m = torch.empty_like(y0)
    with hip.options(hconv=2, hconv2=0, hconv_dbg=4):
        hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, b, None, None, y0, None, ws)
        hip.conv_igemm_ex(g, dt, 0, x, wf, b, None, msk, y0m, None, ws)
    with hip.options(hconv=2, hconv2=0):
        yt = torch.empty_like(y0)
        t_old = timeit(lambda: hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, b, None, None, yt, None, ws))
    print((B, H, W, C, N), "conv_halo.hip %.1f us (%.0f TF)" % (t_old, flops / t_old / 1e6), flush=True)
    for shp in [int(s) for s in a.shapes.split(",")]:
        with hip.options(hconv=2, hconv2=2, hconv2_shape=shp):
            y = torch.full((B, H, W, N), 3.0, device="cuda").bfloat16(); ym = torch.full((B, H, W, N), 3.0, device="cuda").bfloat16()
            hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, b, None, None, y, None, ws)
            hip.conv_igemm_ex(g, dt, 0, x, wf, b, None, msk, ym, None, ws)
            torch.cuda.synchronize()
            ok = torch.equal(y, y0) and torch.equal(ym, y0m)
            err = float((y.float() - y0.float()).abs().max())
            out = []
            for name, o in (("full", 0),) + (() if a.quick else (("noEpi", 1), ("noLoop", 2), ("neither", 3), ("noCopies", 32), ("noBarrier", 64), ("noCopies noBarrier", 96), ("emptyKernel", 1024),  ("noReads", 128), ("noMFMA", 256), ("noReads noCopies noBarrier", 224), ("noMFMA noCopies noBarrier", 352))):
                with hip.options(hconv_dbg=o):
                    t = timeit(lambda: hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, b, None, None, yt, None, ws))
                out.append("%s %.1f" % (name, t) + (" (%.0f TF)" % (flops / t / 1e6) if o == 0 else ""))
        print("   shape %d: %s (max diff %.3g)   " % (shp, "bit-identical" if ok else "DIFFERENT", err) + "  ".join(out), flush=True)
