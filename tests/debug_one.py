import sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
from ursonet_amd import hip
cfg = make_config(dtype="float32", backbone="resnet18", h=128, w=128, batch=2, regress_ori=True)
img, loc, ori, _ = synthetic_batch(cfg, 2, seed=1)
eng = Engine(cfg, "training", seed=3, randomize_bn=True)
eng.load_batch(img, loc, ori); eng.step_eager(); torch.cuda.synchronize()
for name in ["stage1_unit2_conv2", "stage1_unit1_conv2", "stage2_unit2_conv2"]:
    c = eng.convs[name]; n = c.node
    B=2
    G = c.dst.grad.float().cpu().view(B, n.dst.h, n.dst.w, n.cout)
    X = c.src.data.float().cpu().view(B, n.src.h, n.src.w, n.cin)
    got = c.src.grad.float().cpu().view(B, n.src.h, n.src.w, n.cin)
    W = eng.wview(name, "kernel").cpu()
    s = c.scale.cpu()[:n.cout]
    Wf = (W * s)
    x = X.permute(0,3,1,2).clone().requires_grad_(True)
    y = F.conv2d(F.pad(x, (1,1,1,1)), Wf.permute(3,2,0,1))
    (y * G.permute(0,3,1,2)).sum().backward()
    ref = x.grad.permute(0,2,3,1) * (X > 0)
    # residual pending contribution?
    err = (got - ref).abs()
    print(name, "max err", float(err.max()), "ref max", float(ref.abs().max()), "nonzero err frac", float((err > 1e-5).float().mean()))
    idx = (err > 1e-5).nonzero()
    if len(idx):
        print("  b:", idx[:,0].unique().tolist(), " y range:", idx[:,1].min().item(), idx[:,1].max().item(), " x range:", idx[:,2].min().item(), idx[:,2].max().item(), " c:", idx[:,3].unique().tolist()[:40])
        # rerun the kernel standalone
        dstg = torch.empty_like(c.src.grad)
        hip.conv_igemm(c.gd, eng.dt, 0, c.dst.grad, c.wd, None, None, c.src.data, dstg); torch.cuda.synchronize()
        e2 = (dstg.float().cpu().view_as(ref) - ref).abs()
        print("  standalone rerun max err", float(e2.max()))
