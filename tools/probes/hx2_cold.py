import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
dt = hip.BF16
big = torch.empty(700 * 1024 * 1024 // 4, device="cuda")
for (B, H, W, C, N, shp) in ((32, 32, 40, 256, 256, 32), (32, 16, 20, 512, 512, 31), (32, 16, 20, 512, 512, 22)):
    x = torch.randn(B, H, W, C, device="cuda").bfloat16(); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    for what in ("warm", "cold: everything", "cold: filter only warm", "cold: input only warm"):
        ts = []
        with hip.options(hconv=2, hconv2=2, hconv2_shape=shp):
            for it in range(12):
                if what != "warm":
                    big.fill_(float(it))
                    if "filter only" in what: wf.add_(0)          # touch -> back in the caches
                    if "input only" in what: x.add_(0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); hip.conv_igemm(g, dt, hip.EPI_RELU, x, wf, b, None, None, y); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print((B, H, W, C, N), shp, "%-24s median %.1f us  min %.1f" % (what, ts[len(ts) // 2], ts[0]), flush=True)
