"""A/B: conv1 (stem_kernel) + maxpool_fwd_kernel vs the fused stem_pool_kernel at a bench geometry; HIP-event times, cold-ish (other tensors touched between)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursonet_amd import hip
B, H, W = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 512, 640)))
dt = hip.BF16; N = 64
torch.manual_seed(0)
molded = (torch.randn(B, H, W, 4, device="cuda") * 60).to(torch.bfloat16); molded[..., 3] = 0
wf = (torch.randn(N * 224, device="cuda") / 12).to(torch.bfloat16)
bias = torch.randn(N, device="cuda") * 0.1
OH, OW = H // 2, W // 2
g = hip.geom(B, H, W // 2, 8, OH, OW, N, 7, 4, 2, 1, 3, 2)
y = torch.empty(B, OH, OW, N, dtype=torch.bfloat16, device="cuda")
p1 = torch.empty(B, OH // 2, OW // 2, N, dtype=torch.bfloat16, device="cuda"); a1 = torch.empty(p1.shape, dtype=torch.uint8, device="cuda")
p2 = torch.empty_like(p1); a2 = torch.empty_like(a1)
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def t(fn, n=20):
    ts = []
    for _ in range(n):
        junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
def two():
    hip.conv_igemm(g, dt, hip.EPI_RELU, molded, wf, bias, None, None, y); hip.maxpool_fwd(B, OH, OW, N, dt, y, p1, a1)
def one():
    hip.stem_conv_pool(g, dt, molded, wf, bias, p2, a2)
two(); one(); torch.cuda.synchronize()
print("equal values", torch.equal(p1.float(), p2.float()), "argmax", torch.equal(a1, a2), "ndiff", int((p1.float() != p2.float()).sum()), int((a1 != a2).sum()))
print("stem + maxpool: %.1f us   fused: %.1f us" % (t(two), t(one)))
print("stem alone: %.1f us" % t(lambda: hip.conv_igemm(g, dt, hip.EPI_RELU, molded, wf, bias, None, None, y)))
