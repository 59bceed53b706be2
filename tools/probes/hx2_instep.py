import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
dt = hip.BF16
B = 32
# a stage-5 block as the step runs it: branch2a (2048 -> 512 pointwise), branch2b (3x3 512 -> 512), branch2c (512 -> 2048 pointwise + residual)
H, W = 16, 20
xin = torch.randn(B, H, W, 2048, device="cuda").bfloat16()
w2a = (torch.randn(512, 1, 1, 2048, device="cuda") / 2048 ** 0.5).bfloat16(); b2a = torch.randn(512, device="cuda")
w2b = (torch.randn(512, 3, 3, 512, device="cuda") / (9 * 512) ** 0.5).bfloat16(); b2b = torch.randn(512, device="cuda")
w2c = (torch.randn(2048, 1, 1, 512, device="cuda") / 512 ** 0.5).bfloat16(); b2c = torch.randn(2048, device="cuda")
a = torch.empty(B, H, W, 512, device="cuda", dtype=torch.bfloat16); bb = torch.empty_like(a); out = torch.empty_like(xin)
g2a = hip.geom(B, H, W, 2048, H, W, 512, 1, 1, 1, 1, 0, 0); g2b = hip.geom(B, H, W, 512, H, W, 512, 3, 3, 1, 1, 1, 1); g2c = hip.geom(B, H, W, 512, H, W, 2048, 1, 1, 1, 1, 0, 0)
ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, device="cuda")
big = torch.empty(300 * 1024 * 1024 // 2, device="cuda", dtype=torch.bfloat16)
for shp in (31, 22, 0):
    for flush in (False, True):
        with hip.options(hconv=2, hconv2=2 if shp else 0, hconv2_shape=shp):
            def block():
                if flush: big.fill_(1.0)
                hip.conv_igemm_ex(g2a, dt, hip.EPI_RELU, xin, w2a, b2a, None, None, a, None, None)
                hip.conv_igemm_ex(g2b, dt, hip.EPI_RELU, a, w2b, b2b, None, None, bb, None, ws)
                hip.conv_igemm_ex(g2c, dt, hip.EPI_RELU, bb, w2c, b2c, xin, None, out, None, None)
            for _ in range(3): block()
            torch.cuda.synchronize()
            hip.prof_enable(True)
            for _ in range(10): block()
            torch.cuda.synchronize()
            recs = hip.prof_collect_ex(); hip.prof_enable(False)
        t = {}
        for kid, ms, fl, by, nl, sym in recs:
            t.setdefault(sym[:28], []).append(ms * 1e3)
        print("shape", shp, "flush" if flush else "chain", {k: round(sorted(v)[len(v) // 2], 1) for k, v in t.items()}, flush=True)
