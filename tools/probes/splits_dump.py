#!/usr/bin/env python3
"""Split counts of every layer's weight gradient in the cfg2 plan, and who reduces them (REDUCE pass or the finalisation itself)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config
from ursonet_amd.engine import Engine
cfg = make_config(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16")
eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
tot = {}
for name, c in eng.convs.items():
    s = getattr(c, "splits", None)
    if s is None or not hasattr(c, "K_raw"): continue
    n = c.K_raw * c.npad
    cls = "single" if s == 1 else ("fused<=16" if s <= 16 else ("17..48" if s <= 48 else ">48"))
    t = tot.setdefault(cls, [0, 0, 0]); t[0] += 1; t[1] += n; t[2] += n * s
    print("%-22s K %6d N %5d splits %4d  partial MB %7.1f  %s" % (name, c.K_raw, c.npad, s, n * s * 4 / 1e6, cls))
for k, (cnt, n, ns) in tot.items():
    print("%-10s layers %3d  params %6.2f M  partial bytes %7.1f MB" % (k, cnt, n / 1e6, ns * 4 / 1e6))
