#!/usr/bin/env python3
"""Split-reduction launches of one training step, each timed in isolation (hot loop, HIP events) with the bytes it actually reads:
per bucket and per single layer (a one-layer block map), to separate the access pattern from the layer mix.
    python tools/reduce_time.py [--batch 32 --height 512 --width 640]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd import hip
from ursonet_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--height", type=int, default=512)
ap.add_argument("--width", type=int, default=640); ap.add_argument("--dtype", default="bfloat16")
ap.add_argument("--layers", action="store_true", help="also time every layer of the last bucket on its own")
a = ap.parse_args()
cfg = make_config(backbone="resnet50", h=a.height, w=a.width, batch=a.batch, regress_ori=False, ori_bins=16, dtype=a.dtype)
eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
img, loc, ori, _ = synthetic_batch(cfg, a.batch, seed=1)
eng.load_batch(img, loc, ori)
eng.step_eager(); eng.step_eager()
torch.cuda.synchronize()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def layer_bytes(c):
    return c.splits * (c.K_raw * c.npad + hip.WGRAD_PART_PAD + c.npad) * 4 if c.splits > 1 else 0


buckets = {}
for (tag, op), lab in zip(eng.bwd_ops, eng.labels["bwd"]):
    if lab and lab.startswith("reduce:"):
        names = list(tag)
        by = sum(layer_bytes(eng.convs[n]) for n in names)
        us = timeit(op)
        print("%-16s %3d layers  %7.1f MB  %7.1f us  %5.2f TB/s" % (lab, len(names), by / 1e6, us, by / us / 1e6))
        buckets[lab] = names
if a.layers:
    last = buckets[sorted(buckets)[-1]]
    for i, n in enumerate(last):
        c = eng.convs[n]
        if c.splits <= 1:
            continue
        nb = eng.pbatch.plan(hip.PB_REDUCE, "probe%d" % i, [c.desc_id])
        us = timeit(lambda: eng.pbatch.run(hip.PB_REDUCE, "probe%d" % i, eng.dt))
        by = layer_bytes(c)
        print("  %-18s K %5d N %4d splits %4d blocks %5d  %6.1f MB  %6.1f us  %5.2f TB/s" % (n, c.K_raw, c.npad, c.splits, nb, by / 1e6, us, by / us / 1e6))
