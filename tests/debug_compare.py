"""Debug tool (not a test): run one training step on the GPU engine and on a torch-CPU executor
of the SAME graph spec, then print per-tensor errors of every activation, activation gradient and
parameter gradient in forward order.   python tests/debug_compare.py [case]"""
import sys
import os

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import make_config, synthetic_batch  # noqa


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "r18"
    kws = {"r18": dict(backbone="resnet18", h=128, w=128, batch=2, regress_ori=True),
           "r50": dict(backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8),
           "r34": dict(backbone="resnet34", h=64, w=128, batch=3, regress_ori=True, ori_param="euler_angles")}
    dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
    cfg = make_config(dtype=dtype, **kws[case])
    from ursonet_amd.engine import Engine
    from oracle import graph_ref as G
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=1)
    eng = Engine(cfg, "training", seed=3, randomize_bn=True)
    W = eng.get_weights()
    eng.load_batch(img, loc, ori)
    eng.step_eager()
    torch.cuda.synchronize()
    P = G.to_torch(W)
    g = eng.graph
    B = cfg.BATCH_SIZE
    vals = {0: torch.tensor(img).permute(0, 3, 1, 2)}
    for node in g.nodes:
        if node.op == "pool":
            v = G.maxpool_3x3_s2_same(vals[node.src.id])
        else:
            x = vals[node.src.id]
            p = P[node.name]
            if node.dense:
                if x.dim() == 4:
                    x = x.permute(0, 2, 3, 1).reshape(B, -1)
                v = x @ p["kernel"] + p["bias"]
            else:
                k = p["kernel"].permute(3, 2, 0, 1)
                pt, pl = node.pad
                pb = max((node.dst.h - 1) * node.stride + node.kh - x.shape[2] - pt, 0)
                pr = max((node.dst.w - 1) * node.stride + node.kw - x.shape[3] - pl, 0)
                v = F.conv2d(F.pad(x, (pl, pr, pt, pb)), k, p.get("bias"), stride=node.stride)
            if node.bn:
                v = G.batchnorm(v, P[node.bn], False)
            if node.residual is not None:
                v = v + vals[node.residual.id]
            if node.relu:
                v = F.relu(v)
        v.retain_grad()
        vals[node.dst.id] = v
    locp, orip = vals[g.outputs["loc"].id], vals[g.outputs["ori"].id]
    if eng.quat_head:
        orip = orip * torch.rsqrt(torch.clamp((orip * orip).sum(-1, keepdim=True), min=1e-12))
    tl, tor = torch.tensor(loc), torch.tensor(ori)
    ll = G.rel_loss(tl, locp) if cfg.REGRESS_LOC else G.softmax_loss(tl, locp)
    ol = G.one_minus_dot_prod(tor, orip) if cfg.REGRESS_ORI else G.softmax_loss(tor, orip)
    (ll + ol + G.regularizer(P, cfg)).backward()

    def rel(a, b):
        a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)

    print("%-28s %10s %10s %10s" % ("tensor", "act", "act_grad", ""))
    for node in g.nodes:
        t = node.dst
        a = eng.acts[t.id]
        ref = vals[t.id]
        if ref.dim() == 4:
            r = ref.permute(0, 2, 3, 1).reshape(B, -1)
        else:
            r = ref
        got = a.data.float().cpu().view(B, -1)[:, :r.shape[1]] if r.shape[1] != a.data.numel() // B else a.data.float().cpu().view(B, -1)
        ea = rel(got.numpy(), r.detach().numpy())
        eg = float("nan")
        if a.grad is not None and ref.grad is not None:
            rg = ref.grad
            if t.relu:
                rg = rg * (ref > 0)                      # engine stores gradient w.r.t. the pre-ReLU value
            rg = rg.permute(0, 2, 3, 1).reshape(B, -1) if rg.dim() == 4 else rg
            gg = a.grad.float().cpu().view(B, -1)[:, :rg.shape[1]]
            eg = rel(gg.numpy(), rg.numpy())
        name = node.name if node.op == "conv" else "pool"
        flips = int(((got > 0) != (r > 0)).sum()) if t.relu else 0
        print("%-28s %10.2e %10.2e  flips=%d %s" % (name, ea, eg, flips, t))
    grads = eng.get_grads()
    print("\nparameter gradients")
    for ln, ws in P.items():
        for wn, w in ws.items():
            if w.grad is not None:
                print("%-36s %10.2e  |ref|max %.3e" % (ln + "/" + wn, rel(grads[ln][wn], w.grad.numpy()), float(w.grad.abs().max())))


if __name__ == "__main__":
    main()
