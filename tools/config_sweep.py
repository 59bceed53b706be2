#!/usr/bin/env python3
"""Run one hipGraph-replayed training step timing for the other BASELINE.json configs (sanity + numbers)."""
import os, sys, time
os.environ.setdefault("URSO_WGRAD_STREAM", "2")      # as bench.py: the forked backward pass is opt-in in the library
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
CASES = {
    "cfg2_r50_bf16": dict(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16"),
    "cfg4_r101_n24_bf16": dict(backbone="resnet101", h=512, w=640, batch=16, regress_ori=False, ori_bins=24, dtype="bfloat16"),
    "cfg5_r50_f16_classify_loc": dict(backbone="resnet50", h=640, w=960, batch=32, regress_ori=False, regress_loc=False, ori_bins=16, loc_bins=16, f16=True),
    "cfg1_r18_quat_f32": dict(backbone="resnet18", h=128, w=128, batch=2, regress_ori=True, dtype="float32"),
    "cfg2_r50_f32": dict(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="float32"),
}
CASES["cfg2_r50_bf16_train_bn"] = dict(CASES["cfg2_r50_bf16"])        # secondary mode: batch-statistics BN (TRAIN_BN=None)
for name in (sys.argv[1:] or CASES):
    cfg = make_config(**CASES[name])
    if name.endswith("train_bn"):
        cfg.TRAIN_BN = None
    eng = Engine(cfg, "training", seed=1, randomize_bn=True)
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=1)
    eng.load_batch(img, loc, ori)
    for _ in range(3): eng.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): eng.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ls = eng.losses(); fl = eng.flops()[1]
    print("%-28s %8.2f ms/step  %8.1f img/s  %6.1f TFLOP/s  loc_loss %.4f ori_loss %.4f  mem %.1f GB" % (
        name, dt * 1e3, cfg.BATCH_SIZE / dt, fl / dt / 1e12, ls["loc_loss"], ls["ori_loss"], torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del eng; torch.cuda.empty_cache()
