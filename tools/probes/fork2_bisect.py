#!/usr/bin/env python3
"""URSO_WGRAD_STREAM=2 against the single chain after N steps: the inputs and the output of the first launch that differs."""
import os, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
SHAPE = dict(h=256, w=320, batch=4)
STEPS = int(os.environ.get("CHK_STEPS", "2"))
def run(mode, sel=None):
    os.environ["URSO_WGRAD_STREAM"] = mode
    if sel is None: os.environ.pop("URSO_WGRAD_DEFER_RE", None)
    else: os.environ["URSO_WGRAD_DEFER_RE"] = sel
    cfg = make_config(backbone="resnet50", regress_ori=False, ori_bins=16, dtype="bfloat16", **SHAPE)
    eng = Engine(cfg, "training", seed=7, randomize_bn=True)
    img, loc, ori, _ = synthetic_batch(cfg, SHAPE["batch"], seed=3)
    eng.load_batch(img, loc, ori)
    for _ in range(STEPS): (eng.step_eager() if os.environ.get("CHK_EAGER") == "1" else eng.step())
    torch.cuda.synchronize()
    out = {}
    c = eng.convs["res3d_branch2c"]
    out["wd"] = c.wd.clone().float(); out["wf"] = c.wf.clone().float()
    out["src.data (mask)"] = c.src.data.clone().float()
    if c.src.bits is not None: out["src.bits"] = c.src.bits.clone().float()
    out["dst.grad (dz)"] = c.dst.grad.clone().float()
    out["src.grad (dx)"] = c.src.grad.clone().float()
    out["flat_v"] = eng.flat_v.clone(); out["flat_w"] = eng.flat_w.clone(); out["flat_g"] = eng.flat_g.clone()
    for nm in ("res3d_branch2b", "res4a_branch2a", "res5c_branch2c"):
        out[nm + ".wf"] = eng.convs[nm].wf.clone().float()
    return out
a = run("0"); b = run("2", os.environ.get("CHK_RE", "res5"))
for k in a:
    print("after %d step(s): %-22s differing %8d of %8d" % (STEPS, k, int(((a[k] - b[k]).abs() > 0).sum()), a[k].numel()))
d = ((a["src.grad (dx)"] - b["src.grad (dx)"]).abs() > 0).view(4, 32, 40, 128)
ch = d.sum(dim=(0, 1, 2))
print("pixels where channel c differs (c = 0..127):", ch.tolist())
px = d.any(dim=3)
print("pixels with any difference: %d of %d (even-even pixels: %d)" % (int(px.sum()), px.numel(), 4 * 16 * 20))
