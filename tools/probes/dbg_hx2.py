import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from ursonet_amd import hip
for dt in (hip.BF16, hip.F16):
    tdt = hip.TORCH_DT[dt]
    for (B, H, W, C, N) in ((2, 8, 12, 256, 256), (2, 16, 24, 128, 128), (2, 4, 6, 512, 512), (2, 32, 40, 256, 256), (4, 16, 20, 512, 512)):
        torch.manual_seed(5)
        x = torch.randn(B, H, W, C, device="cuda").to(tdt); wf = (torch.randn(N, 3, 3, C, device="cuda") / (9 * C) ** 0.5).to(tdt)
        b = torch.randn(N, device="cuda"); msk = torch.randn(B, H, W, N, device="cuda").to(tdt)
        g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
        ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, device="cuda")
        z = (F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wf.double().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) + b.double().cpu())
        outs = {}
        for name, o in (("dma", dict(hconv=0, c3=0)), ("halo1", dict(hconv=2, hconv2=0, c3=0)), ("auto", dict(c3=0)), ("auto+c3", dict())):
            with hip.options(**o):
                y = torch.full((B, H, W, N), 3.0, device="cuda").to(tdt); ym = torch.full((B, H, W, N), 3.0, device="cuda").to(tdt)
                hip.conv_igemm_ex(g, dt, hip.EPI_RELU, x, wf, b, None, None, y, None, ws)
                hip.conv_igemm_ex(g, dt, 0, x, wf, None, None, msk, ym, None, ws)
                torch.cuda.synchronize()
                pick = hip.conv_igemm_halo2_shape(g, dt, hip.EPI_RELU, False, True) if hip.conv_igemm_halo_ok(g, dt, hip.EPI_RELU) else -1
            outs[name] = (y.double().cpu(), ym.double().cpu(), pick)
        zr = torch.relu(z); zm = (z - b.double().cpu()) * (msk.double().cpu() > 0)
        for n, (y, ym, pick) in outs.items():
            print(dt, (B, H, W, C, N), "%-8s pick %3d  fwd err %.2e  masked-dgrad err %.2e" % (n, pick, float((y - zr).abs().max() / zr.abs().max()), float((ym - zm).abs().max() / zm.abs().max())), flush=True)
