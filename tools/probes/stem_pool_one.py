"""Three launches of the fused stem + pool kernel at cfg2 (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
B, H, W = 32, 512, 640
dt = hip.BF16; N = 64
torch.manual_seed(0)
molded = (torch.randn(B, H, W, 4, device="cuda") * 60).to(torch.bfloat16); molded[..., 3] = 0
wf = (torch.randn(N * 224, device="cuda") / 12).to(torch.bfloat16)
bias = torch.randn(N, device="cuda") * 0.1
OH, OW = H // 2, W // 2
g = hip.geom(B, H, W // 2, 8, OH, OW, N, 7, 4, 2, 1, 3, 2)
p2 = torch.empty(B, OH // 2, OW // 2, N, dtype=torch.bfloat16, device="cuda"); a2 = torch.empty(p2.shape, dtype=torch.uint8, device="cuda")
for _ in range(3):
    hip.stem_conv_pool(g, dt, molded, wf, bias, p2, a2)
torch.cuda.synchronize()
