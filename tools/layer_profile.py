#!/usr/bin/env python3
"""Per-launch timing of one training step (HIP events via the library profiler).
    python tools/layer_profile.py [--batch 32 --height 512 --width 640 --dtype bfloat16] > profiles/xxx.txt"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--height", type=int, default=512)
ap.add_argument("--width", type=int, default=640); ap.add_argument("--dtype", default="bfloat16")
ap.add_argument("--backbone", default="resnet50"); ap.add_argument("--ori-bins", type=int, default=16)
ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE")
a = ap.parse_args()
from ursonet_amd import hip
for kv in a.opt:
    k, _, v = kv.partition("=")
    hip.set_option(k, int(v))
cfg = make_config(backbone=a.backbone, h=a.height, w=a.width, batch=a.batch, regress_ori=False, ori_bins=a.ori_bins, dtype=a.dtype)
eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
img, loc, ori, _ = synthetic_batch(cfg, a.batch, seed=1)
eng.load_batch(img, loc, ori)
eng.step_eager(); eng.step_eager()
recs = eng.profile_step()
tot = sum(r[2] for r in recs)
print("# %s B=%d %dx%d %s : %.3f ms per eager step (sum of launches)" % (a.backbone, a.batch, a.height, a.width, a.dtype, tot))
print("%-34s %9s %9s %9s" % ("launch", "us", "TFLOP/s", "GB/s(alg)"))
groups = {}
for label, kid, ms, fl, by in recs:
    print("%-34s %9.1f %9.1f %9.1f" % (label, ms * 1e3, fl / (ms * 1e9) if ms > 0 else 0, by / (ms * 1e6) if ms > 0 else 0))
    g = label.split(":")[0]
    groups.setdefault(g, [0.0, 0.0, 0]); groups[g][0] += ms; groups[g][1] += fl; groups[g][2] += 1
print("\n# totals")
for g, (ms, fl, n) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print("%-12s n=%3d %8.3f ms  %7.1f TFLOP/s" % (g, n, ms, fl / (ms * 1e9) if ms > 0 else 0))
