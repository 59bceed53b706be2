#!/usr/bin/env python3
"""What this box's HBM delivers to simple streaming kernels (torch elementwise/reduction ops), as a yardstick for the algorithmic
GB/s of the HBM-bound conv layers: pure read, copy (1:1), read-heavy 4:1, write-only.
    python tools/hbm_probe.py"""
import torch, json
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
out = {}
for mb in (84, 335, 1340):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    x4 = x.view(-1, 4)
    y1 = torch.empty(x4.shape[0], device="cuda", dtype=torch.bfloat16)
    xf = x.view(torch.int32)
    r = {}
    r["read_sum_i32"] = mb * 1e6 / t(lambda: torch.sum(xf)) / 1e12
    r["copy_1to1"] = 2 * mb * 1e6 / t(lambda: y.copy_(x)) / 1e12
    r["read4_write1"] = 1.25 * mb * 1e6 / t(lambda: torch.sum(x4, dim=1, out=y1)) / 1e12
    r["fill_write"] = mb * 1e6 / t(lambda: y.zero_()) / 1e12
    r["inplace_rw"] = 2 * mb * 1e6 / t(lambda: x.mul_(1.0001)) / 1e12
    out["%dMB" % mb] = {k: round(v, 2) for k, v in r.items()}
    del x, y, y1
print(json.dumps(out))
