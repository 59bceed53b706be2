import sys, os, torch
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from ursonet_amd import hip
dt = hip.BF16
for (B, H, W, C, N) in ((4, 32, 40, 256, 256), (8, 16, 20, 512, 512), (32, 32, 40, 256, 256)):
    torch.manual_seed(5)
    x = torch.randn(B, H, W, C, device="cuda").bfloat16(); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    z = None
    if B <= 8:
        z = (F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wf.double().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) + b.double().cpu())
    outs = {}
    for name, o in (("old", dict(hconv=2, hconv2=0, hconv_dbg=4)), ("dma", dict(hconv=0)), ("n22", dict(hconv=2, hconv2=2, hconv2_shape=22)), ("n31", dict(hconv=2, hconv2=2, hconv2_shape=31)),
                    ("n32", dict(hconv=2, hconv2=2, hconv2_shape=32)), ("n11", dict(hconv=2, hconv2=2, hconv2_shape=11))):
        with hip.options(**o):
            y = torch.full((B, H, W, N), 3.0, device="cuda").bfloat16()
            hip.conv_igemm(g, dt, 0, x, wf, b, None, None, y)
            torch.cuda.synchronize()
        outs[name] = y.float().cpu()
    for n, y in outs.items():
        d = (y != outs["old"])
        msg = "%s: mismatches vs old %d of %d (max %.4f)" % (n, int(d.sum()), d.numel(), float((y - outs["old"]).abs().max()))
        if z is not None:
            msg += "  err vs fp64 %.3e" % float((y.double() - z).abs().max() / z.abs().max())
        if int(d.sum()):
            idx = d.nonzero()[:3].tolist(); msg += "  first " + str(idx)
        print((B, H, W, C, N), msg, flush=True)
