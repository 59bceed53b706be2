"""Three launches of urso_stem_wgrad_pooled at cfg2 (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
B, H, W, N = 32, 512, 640, 64
dt = hip.BF16
torch.manual_seed(0)
molded = (torch.randn(B, H, W, 4, device="cuda") * 60).to(torch.bfloat16); molded[..., 3] = 0
OH, OW = H // 2, W // 2
g = hip.geom(B, H, W // 2, 8, OH, OW, N, 7, 4, 2, 1, 3, 2)
PH, PW = OH // 2, OW // 2
dpool = torch.randn(B, PH, PW, N, device="cuda").to(torch.bfloat16)
am = torch.randint(0, 9, (B, PH, PW, N), dtype=torch.uint8, device="cuda")
ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 16, dtype=torch.float32, device="cuda")
dw = torch.empty(7 * 4 * 8 * N, dtype=torch.float32, device="cuda"); cs = torch.empty(N, dtype=torch.float32, device="cuda")
for _ in range(3):
    hip.stem_wgrad_pooled(g, dt, molded, dpool, am, ws, dw, cs)
torch.cuda.synchronize()
