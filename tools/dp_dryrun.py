#!/usr/bin/env python3
"""Data-parallel path on ONE GPU (no multi-GPU node is reachable from the build container): what can be measured is the cost of
the DP machinery itself -- the step split into per-bucket hipGraph segments with an RCCL call between them (world size 1, forced
collectives) against the single-graph step -- i.e. the overhead a rank pays even when communication is free.  Also times the
bf16-compressed bucket path.  Prints one JSON line (profiles/r02_dp_dryrun.json)."""
import json, os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["URSO_DP_FORCE_COLLECTIVES"] = "1"
import torch
import torch.distributed as dist
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
from ursonet_amd.dp import DataParallelEngine

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = make_config(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16")
img, loc, ori, _ = synthetic_batch(cfg, 32, seed=1)


def timed(step, n=20, warm=8, reps=3):
    """Median of `reps` timings of n steps (the first collective of a process also pays RCCL's lazy set-up: warm-up covers it)."""
    for _ in range(warm):
        step()
    ms = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ms)[len(ms) // 2]


out = {}
eng = Engine(cfg, "training", seed=1, randomize_bn=True); eng.load_batch(img, loc, ori)
out["single_graph_ms"] = timed(eng.step)
for name, comp in (("dp_segments_fp32_ms", None), ("dp_segments_bf16_ms", "bf16")):
    e2 = Engine(cfg, "training", seed=1, randomize_bn=True); e2.load_batch(img, loc, ori)
    dp = DataParallelEngine(e2, compress=comp)
    out[name] = timed(dp.step)
    out["buckets"] = [(e - s_) * 4 for s_, e, _ in dp.buckets]
    del dp, e2
    torch.cuda.empty_cache()
from ursonet_amd import hip
e3 = Engine(cfg, "training", seed=1, randomize_bn=True)
dp = DataParallelEngine(e3, comm_cus=16)               # re-plans for 240 CUs (option cus): what the reservation costs while nothing is beside the step
e3.load_batch(img, loc, ori)
out["dp_segments_fp32_comm_cus16_ms"] = timed(dp.step)
hip.set_option("cus", 0)
del dp, e3
e4 = Engine(cfg, "training", seed=1, randomize_bn=True)
dp = DataParallelEngine(e4, tail_bytes=1 << 20)        # what a world size > 1 gets by default: the stem-side bucket capped at 1 MiB
e4.load_batch(img, loc, ori)
out["dp_segments_fp32_tail_1MiB_ms"] = timed(dp.step)
out["buckets_tail_1MiB"] = [(e - s_) * 4 for s_, e, _ in dp.buckets]
del dp, e4
out["note"] = ("world size 1, RCCL calls forced: overhead of the per-bucket graph segmentation + stream events + (bf16) the rounding / "
               "error-feedback passes; no bytes move.  No multi-GPU scaling curve exists for this build yet.")
print(json.dumps(out))
dist.destroy_process_group()
