import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
stage, shp = int(sys.argv[1]), int(sys.argv[2])
B, H, W, C, N = (32, 32, 40, 256, 256) if stage == 4 else (32, 16, 20, 512, 512)
x = torch.randn(B, H, W, C, device="cuda").bfloat16(); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).bfloat16()
b = torch.randn(N, device="cuda"); y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, device="cuda")
g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
with hip.options(hconv=2, hconv2=2 if shp else 0, hconv2_shape=shp):
    for _ in range(6):
        hip.conv_igemm_ex(g, hip.BF16, hip.EPI_RELU, x, wf, b, None, None, y, None, ws)
torch.cuda.synchronize()
