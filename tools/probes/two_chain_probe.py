#!/usr/bin/env python3
"""Phase-unlocking without new kernels (cheapest variant): the batch as TWO independent half-batch chains replayed
concurrently on two streams, so that one chain's launch ramps (prologue, first copies, store tail: ~8-10 us of every launch whatever its
size) run beside the other chain's streaming / MFMA phases.  Two Engines of B/2 images (own activations, own captured graph), replayed
alternately on two streams, against ONE Engine of B images; grids planned for `cus` CUs each (0 = the whole chip: two full-chip persistent
grids; 128: half the chip per chain).  Prints ms per B images.  Timing experiment only: the two engines do not share weights here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("URSO_WGRAD_STREAM", "0")
import numpy as np
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
from ursonet_amd import hip

B = int(os.environ.get("PROBE_B", "32"))
STEPS = int(os.environ.get("PROBE_STEPS", "40"))


def build(batch, seed):
    cfg = make_config(backbone="resnet50", h=512, w=640, batch=batch, regress_ori=False, ori_bins=16, dtype="bfloat16")
    img, loc, ori, _ = synthetic_batch(cfg, batch, seed=seed)
    u8 = np.clip(np.rint(img + np.asarray(cfg.MEAN_PIXEL, dtype=np.float32)), 0, 255).astype(np.uint8)
    eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
    eng.load_batch_u8(u8, loc, ori)
    return eng


def timed(fn, steps):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = build(B, 1)
one.step()
print("one engine, B = %d, whole chip:                 %.3f ms per %d images" % (B, timed(one.step, STEPS), B), flush=True)
del one
torch.cuda.empty_cache()
for cus in (0, 128, 160, 192):
    with hip.options(cus=cus):
        ea, eb = build(B // 2, 1), build(B // 2, 2)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(sa):
            ea.step()
        with torch.cuda.stream(sb):
            eb.step()
        torch.cuda.synchronize()

        def both():
            with torch.cuda.stream(sa):
                ea._graphs.replay()
            with torch.cuda.stream(sb):
                eb._graphs.replay()

        def serial():
            ea._graphs.replay()
            eb._graphs.replay()
        t_ser = timed(serial, STEPS)
        t_con = timed(both, STEPS)
        print("two engines, B = %d each, grids for %3s CUs: one after the other %.3f ms, on two streams %.3f ms per %d images"
              % (B // 2, cus or "all", t_ser, t_con, B), flush=True)
        del ea, eb
        torch.cuda.empty_cache()
