#!/usr/bin/env python3
"""Full-size A/B of the gathered-input path of the stride-2 stage-entry layers: same weights and batch, URSO_COMPACT_INPUT=0 / 1 (and the
producer-written copy against the gather pass, option pair), first-step losses and the entry layers' outputs compared.
    python tools/compact_input_check.py [--backbone resnet50 --batch 32 --height 512 --width 640]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd import hip
from ursonet_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="resnet50"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--height", type=int, default=512); ap.add_argument("--width", type=int, default=640)
ap.add_argument("--dtype", default="bfloat16"); ap.add_argument("--ori-bins", type=int, default=16)
a = ap.parse_args()
cfg = make_config(backbone=a.backbone, h=a.height, w=a.width, batch=a.batch, regress_ori=False, ori_bins=a.ori_bins, dtype=a.dtype)
img, loc, ori, _ = synthetic_batch(cfg, a.batch, seed=1)
res = {}
for mode, env, pair in (("strided", "0", 1), ("gather", "1", 0), ("producer", "1", 1)):
    os.environ["URSO_COMPACT_INPUT"] = env
    with hip.options(pair=pair):
        eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
    eng.load_batch(img, loc, ori)
    eng.step_eager(); torch.cuda.synchronize()
    acts = {n: c.dst.data.float().clone() for n, c in eng.convs.items() if n.endswith(("a_branch2a", "a_branch1")) and n[3] in "345"}
    res[mode] = (eng.losses(), acts, [l for l in eng.labels["fwd"] if l and ("sampled" in l or "subsample" in l)])
    print(mode, res[mode][0], res[mode][2])
    del eng
ref = res["strided"]
for mode in ("gather", "producer"):
    for n, t in res[mode][1].items():
        d = float((t - ref[1][n]).abs().max() / (ref[1][n].abs().max() + 1e-30))
        print("  %-9s %-18s max rel diff vs strided %.3e" % (mode, n, d))
