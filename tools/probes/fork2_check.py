#!/usr/bin/env python3
"""The complementary fork (URSO_WGRAD_STREAM=2) against the single chain: same bits after three replayed steps; which launches left the chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
for kw in (dict(backbone="resnet50", h=256, w=320, batch=4), dict(backbone="resnet18", h=256, w=320, batch=4, regress_ori=True),
           dict(backbone="resnet50", h=512, w=640, batch=32)):
    res = {}
    for mode in ("0", "2"):
        os.environ["URSO_WGRAD_STREAM"] = mode
        k = dict(regress_ori=False, ori_bins=16, dtype="bfloat16"); k.update(kw)
        cfg = make_config(**k)
        eng = Engine(cfg, "training", seed=7, randomize_bn=True)
        img, loc, ori, _ = synthetic_batch(cfg, kw["batch"], seed=3)
        eng.load_batch(img, loc, ori)
        for _ in range(3): eng.step()
        torch.cuda.synchronize()
        res[mode] = (eng.flat_w.clone(), eng.flat_g.clone())
        if mode == "2":
            labs = [l for l in eng.labels["bwd"] if l is not None]
            print(kw, "side stream:", eng.wgrad_stream is not None)
    print("   weights equal:", torch.equal(res["0"][0], res["2"][0]), " gradients equal:", torch.equal(res["0"][1], res["2"][1]))
