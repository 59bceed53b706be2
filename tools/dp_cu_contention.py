#!/usr/bin/env python3
"""What do CUs held by another kernel cost the training step, and what does planning the step for fewer CUs cost?  One GPU only
(no multi-GPU box is reachable from the build container), so the collective's resident workgroups are played by a stand-in:
tools/probes/occupy.hip -- N workgroups of 256 threads with 64 KiB of LDS each that sleep-spin for the length of the step on a
side stream.  For N in 8, 16, 32 the captured step (cfg2) is timed
    clean           full grids, nothing beside it
    contended       full grids, N workgroups resident beside it (what data parallelism with comm_cus = 0 risks)
    planned         option cus = CUs - N (DataParallelEngine(comm_cus=N)), N workgroups resident beside it
    planned_clean   option cus = CUs - N, nothing beside it (what the reservation costs while no collective runs)
    *_light         the same with 16 KiB of LDS per stand-in workgroup: it can share a CU with the kernels that leave that much free
Prints one JSON line (profiles/r02_dp_cu_contention.json)."""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd import hip
from ursonet_amd.engine import Engine

LIB = os.path.join(ROOT, "tools", "probes", "bin", "liboccupy.so")
if not os.path.exists(LIB):                    # tools/probes/bin/ is not tracked: build the stand-in where hipcc is at hand
    import subprocess
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", LIB, os.path.join(ROOT, "tools", "probes", "occupy.hip")])
occ = ctypes.CDLL(LIB)
occ.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
occ.occupy_launch.restype = ctypes.c_int
dev = torch.device("cuda", 0)
cus = torch.cuda.get_device_properties(dev).multi_processor_count
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(device=dev)
cfg = make_config(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16")
img, loc, ori, _ = synthetic_batch(cfg, 32, seed=1)
TICKS_PER_MS = 100000           # wall_clock64: constant 100 MHz


def timed(eng, n_occ, reps=12, lds=65536, hold_ms=14):
    """Median ms of one graph replay; with n_occ > 0 the stand-in is launched first and outlives the step."""
    for _ in range(5):
        eng.step()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if n_occ:
            rc = occ.occupy_launch(n_occ, int(hold_ms * TICKS_PER_MS), lds, sink.data_ptr(), side.cuda_stream)
            assert rc == 0, rc
            torch.cuda._sleep(200000)          # let the stand-in's workgroups become resident before the step starts (~0.1 ms)
        e0.record(); eng.step(); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return round(statistics.median(ms), 3)


def engine(usable):
    hip.set_option("cus", usable)
    e = Engine(cfg, "training", seed=1, randomize_bn=True)
    e.load_batch(img, loc, ori)
    return e


out = {"cus": cus, "occupier": "256 threads + 64 KiB LDS per workgroup (light: 16 KiB), resident for the whole step", "rows": []}
full = engine(0)
out["clean_ms"] = timed(full, 0)
for n in (8, 16, 32):
    row = {"held_cus": n, "contended_ms": timed(full, n), "contended_light_ms": timed(full, n, lds=16384)}
    e = engine(cus - n)
    row["planned_ms"] = timed(e, n)
    row["planned_light_ms"] = timed(e, n, lds=16384)
    row["planned_clean_ms"] = timed(e, 0)
    out["rows"].append(row)
    del e
    torch.cuda.empty_cache()
hip.set_option("cus", 0)
# stand-in resident for part of the step only (from its start): is the loss proportional to the time the CUs are held?
out["partial_hold_full_grids"] = [{"held_cus": 16, "hold_ms": h, "step_ms": timed(full, 16, hold_ms=h)} for h in (1, 2, 4)]
out["clean_again_ms"] = timed(full, 0)
print(json.dumps(out))
