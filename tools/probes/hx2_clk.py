import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
B, H, W, C, N = (32, 32, 40, 256, 256)
x = torch.randn(B, H, W, C, device="cuda").bfloat16(); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).bfloat16()
b = torch.randn(N, device="cuda"); y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, device="cuda")
g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
for name, o in (("full", 0), ("noReads noCopies noBarrier", 224), ("noMFMA", 256), ("readsFirst", 10 << 12), ("noLoop", 2)):
    with hip.options(hconv=2, hconv2=2, hconv2_shape=32, hconv_dbg=o | 2048):
        for _ in range(20):
            hip.conv_igemm_ex(g, hip.BF16, hip.EPI_RELU, x, wf, b, None, None, y, None, ws)
    torch.cuda.synchronize()
    c = ws.view(torch.int64)[1024:1024 + 2 * 226].view(-1, 2).double().cpu()
    cyc, rt = c[:, 0], c[:, 1]
    print("%-28s cycles mean %.0f max %.0f; 100MHz ticks mean %.1f max %.1f -> %.2f us, clock %.3f GHz" % (name, cyc.mean(), cyc.max(), rt.mean(), rt.max(), rt.mean() / 100, cyc.mean() / (rt.mean() * 10)))
