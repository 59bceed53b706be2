#!/usr/bin/env python3
"""Which launches pay for the forked backward pass?  The library's launch profiler (HIP events around every launch, on the launch's own stream) over
one eager step on the single chain and one with the side stream active: per launch, the duration alone and beside the other branch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
from ursonet_amd import hip
cfg = make_config(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16")
eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
img, loc, ori, _ = synthetic_batch(cfg, 32, seed=1)
eng.load_batch(img, loc, ori)
for _ in range(3): eng.step_eager()
def prof(single):
    labels = (eng.labels["prep"] + eng.labels["fwd"] + ["loss"] * (len(eng.loss_pre_ops) + len(eng.loss_ops)) + [l for l in eng.labels["bwd"] if l is not None] + eng.labels["opt"])
    acc = [0.0] * len(labels)
    for _ in range(3):
        torch.cuda.synchronize(); hip.prof_collect(); hip.prof_enable(True)
        eng._single_chain = single
        try:
            eng.step_eager(); torch.cuda.synchronize(); recs = hip.prof_collect_ex()
        finally:
            eng._single_chain = False; hip.prof_enable(False)
        assert len(recs) == len(labels), (len(recs), len(labels))
        for i, r in enumerate(recs): acc[i] += r[1] / 3
    return labels, acc
labs, a = prof(True)
_, b = prof(False)
side = set()
print("%-70s %9s %9s" % ("launch (backward pass only; * = on the side stream)", "chain us", "forked us"))
inb = False
tot = [0, 0, 0, 0]
for l, x, y in zip(labs, a, b):
    if l.startswith(("dgrad", "wgrad", "expand", "reduce", "finalize", "unpack", "bits")):
        tot[0] += x; tot[1] += y
        if abs(y - x) > 0.15 * x and abs(y - x) > 0.004:
            print("%-70s %9.1f %9.1f" % (l[:70], x * 1e3, y * 1e3))
print("backward launches, summed: chain %.3f ms  forked %.3f ms (sum of durations, not wall time)" % (tot[0], tot[1]))
