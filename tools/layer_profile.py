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
ap.add_argument("--train-bn", action="store_true", help="batch-statistics BN (TRAIN_BN=None, config.py:143-147) instead of the frozen default")
a = ap.parse_args()
from ursonet_amd import hip
for kv in a.opt:
    k, _, v = kv.partition("=")
    hip.set_option(k, int(v))
cfg = make_config(backbone=a.backbone, h=a.height, w=a.width, batch=a.batch, regress_ori=False, ori_bins=a.ori_bins, dtype=a.dtype)
if a.train_bn:
    cfg.TRAIN_BN = None
eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
img, loc, ori, _ = synthetic_batch(cfg, a.batch, seed=1)
eng.load_batch(img, loc, ori)
eng.step_eager(); eng.step_eager()
recs = eng.profile_step()
tot = sum(r[2] for r in recs)
print("# %s B=%d %dx%d %s : %.3f ms per eager step (sum of launches)" % (a.backbone, a.batch, a.height, a.width, a.dtype, tot))
def short(sym):
    """_Z15wgrad_tr_kernelIDF16bLi0ELb1EEv9WgradArgs -> wgrad_tr_kernel<...>: the name up to its template arguments."""
    import re
    m = re.match(r"_Z(\d+)", sym)
    if m:
        n = int(m.group(1)); st = m.end()
        return sym[st:st + n] + ("<" + sym[st + n + 1:st + n + 25] + ">" if len(sym) > st + n and sym[st + n] == "I" else "")
    return sym.split("(")[0][:40]
print("%-34s %9s %9s %9s  %s" % ("launch", "us", "TFLOP/s", "GB/s(alg)", "kernel (device symbol, first of the call)"))
groups = {}
for label, kid, ms, fl, by, nl, sym in recs:
    print("%-34s %9.1f %9.1f %9.1f  %s%s" % (label, ms * 1e3, fl / (ms * 1e9) if ms > 0 else 0, by / (ms * 1e6) if ms > 0 else 0, short(sym),
                                          (" +%d" % (nl - 1)) if nl > 1 else ""))
    g = label.split(":")[0]
    groups.setdefault(g, [0.0, 0.0, 0]); groups[g][0] += ms; groups[g][1] += fl; groups[g][2] += 1
print("\n# totals")
for g, (ms, fl, n) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print("%-12s n=%3d %8.3f ms  %7.1f TFLOP/s" % (g, n, ms, fl / (ms * 1e9) if ms > 0 else 0))
