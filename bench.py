#!/usr/bin/env python3
"""Benchmark of the UrsoNet hot path on MI355X: images/sec of one full training step
(weight prep + forward + losses + backward + global-norm clip + momentum SGD [+ RCCL gradient
all-reduce for N > 1]) of ResNet-50 / bottleneck 32 / ori_resolution 16 soft-classification
head at batch 32 x 512 x 640 x 3 per GPU, bf16 storage with fp32 accumulation (BASELINE.json
configs[1]; the 640x512 of the metric is W x H).  Synthetic SPEED/URSO-shaped inputs already
resident in HBM, random-init (glorot) weights with randomised BN statistics.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) incl. `roofline` (dominant kernel:
the MFMA implicit-GEMM conv, timed live with HIP events through the library's launch profiler)
and `cpu_baseline` (the torch-CPU oracle restatement of the same step, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_MFMA = {"bfloat16": 2500.0, "float16": 2500.0, "float32": 157.3}      # dense TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def bench_config(dtype, batch, h, w, backbone="resnet50", ori_bins=16):
    from util import make_config
    return make_config(backbone=backbone, h=h, w=w, batch=batch, regress_ori=False, regress_loc=True, ori_bins=ori_bins,
                       bottleneck=32, branch=1024, dtype=dtype)


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def host_cpu_info():
    """lscpu-style description of the host: model string, sockets, physical cores, hardware threads."""
    model, phys, sockets = "unknown", set(), set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v; sockets.add(v)
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid)); pid = cid = None
    except Exception:
        pass
    return {"model": model, "sockets": max(len(sockets), 1), "physical_cores": len(phys) or physical_cores(), "threads": os.cpu_count()}


def cpu_baseline(cfg_kw, sample_batch, steps):
    """The oracle's training step (fwd+loss+bwd+SGD, fp32, torch-CPU/oneDNN) on the host cores,
    on a bounded sample: `sample_batch` images per step, `steps` timed steps after one warm-up; plus a forward-only figure."""
    from oracle import graph_ref as G
    from util import synthetic_batch
    cores = min(physical_cores(), 128)
    torch.set_num_threads(cores)
    cfg = bench_config("float32", sample_batch, cfg_kw["h"], cfg_kw["w"], cfg_kw["backbone"], cfg_kw["ori_bins"])
    P = G.to_torch(G.init_params(cfg, 1234, randomize_bn=True))
    img, loc, ori, _ = synthetic_batch(cfg, sample_batch, seed=99)
    img, loc, ori = torch.tensor(img), torch.tensor(loc), torch.tensor(ori)
    vel = {}
    G.train_step(P, vel, img, loc, ori, cfg, 1e-3)
    t0 = time.time()
    for _ in range(steps):
        G.train_step(P, vel, img, loc, ori, cfg, 1e-3)
    dt = time.time() - t0
    with torch.no_grad():
        Pn = G.to_torch({ln: {wn: w.detach().numpy() for wn, w in ws.items()} for ln, ws in P.items()}, requires_grad=False)
        G.forward(Pn, img, cfg)
        t1 = time.time()
        for _ in range(steps):
            G.forward(Pn, img, cfg)
        dtf = time.time() - t1
    return {"value": round(sample_batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "forward_only_images_per_sec": round(sample_batch * steps / dtf, 3), "host": host_cpu_info(),
            "sample": "oracle/graph_ref.train_step fp32, %s %dx%d, batch %d, %d timed steps after 1 warm-up (%.1f s; forward-only %.1f s)"
                      % (cfg_kw["backbone"], cfg_kw["h"], cfg_kw["w"], sample_batch, steps, dt, dtf)}


def pcie_inclusive(eng, cfg, batch, steps):
    """The same training step fed the way net.UrsoNet.train feeds it (ursonet_amd.feeder.DeviceFeeder): every step takes a NEW
    uint8 batch + targets from pinned host memory, uploaded on a side stream into a staging buffer while the previous step runs,
    then copied into the engine's input (mean subtraction + cast happen in the first kernel).  Reported beside `value`, never as
    it: `value` is quoted with the inputs resident in HBM."""
    from util import synthetic_batch
    host = []
    for k in range(2):
        img, loc, ori, _ = synthetic_batch(cfg, batch, seed=77 + k)
        u8 = np.clip(np.rint(img + np.asarray(cfg.MEAN_PIXEL, dtype=np.float32)), 0, 255).astype(np.uint8)
        host.append([torch.from_numpy(np.ascontiguousarray(a)).pin_memory() for a in (u8, loc, ori)])
    side = torch.cuda.Stream()
    stage = [[torch.empty(h.shape, dtype=h.dtype, device="cuda") for h in host[0]] for _ in range(2)]
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    consumed = [torch.cuda.Event(), torch.cuda.Event()]

    def upload(k):
        slot = k & 1
        with torch.cuda.stream(side):
            side.wait_event(consumed[slot])                 # the step that read this staging slot has copied it out
            for d, h in zip(stage[slot], host[slot]):
                d.copy_(h, non_blocking=True)
            ev[slot].record(side)

    def run(n, k0):
        for k in range(k0, k0 + n):
            slot = k & 1
            torch.cuda.current_stream().wait_event(ev[slot])
            eng.load_batch_u8(*stage[slot])
            consumed[slot].record(torch.cuda.current_stream())
            upload(k + 1) if k + 1 < k0 + n else None
            eng.step()
    for c in consumed:
        c.record(torch.cuda.current_stream())
    upload(0); run(4, 0)                                    # warm-up (re-captures the graph for the uint8 input)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    upload(4); run(steps, 4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = sum(h.numel() * h.element_size() for h in host[0])
    return {"value": round(batch * steps / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "host_bytes_per_step": nbytes,
            "what": "pinned uint8 batch + targets -> side-stream upload (double buffer) -> step, a new batch every step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--ori-bins", type=int, default=16)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--pcie-steps", type=int, default=20, help="steps of the host-buffer-inclusive leg (0 = skip)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel-policy option for A/B runs (urso_set_option; repeatable); recorded in config.options")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    force_dp = os.environ.get("URSO_DP_FORCE_COLLECTIVES", "0") == "1" and "RANK" in os.environ
    comm_cus = 0
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from ursonet_amd.dp import reserve_comm_cus
        comm_cus = reserve_comm_cus()               # RCCL channels <= the CUs the step's grids leave free (URSO_DP_COMM_CUS; default 0 = none)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ursonet_amd.engine import Engine
    from ursonet_amd import hip
    from util import synthetic_batch
    for kv in args.opt:
        k, _, v = kv.partition("=")
        hip.set_option(k, int(v))

    cfg = bench_config(args.dtype, args.batch, args.height, args.width, args.backbone, args.ori_bins)
    cfg.DP_EXACT_REL_LOSS = os.environ.get("URSO_DP_EXACT_REL_LOSS", "0") == "1"      # default: per-rank loss
    eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
    if world > 1 or force_dp:
        from ursonet_amd.dp import DataParallelEngine
        runner = DataParallelEngine(eng, comm_cus=comm_cus)       # re-plans the step for (CUs - comm_cus): before the batch is loaded
    else:
        runner = eng
    img, loc, ori, _ = synthetic_batch(cfg, args.batch, seed=1234 + rank)
    eng.load_batch(img, loc, ori)                  # inputs resident in HBM before the timed region
    torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):
        runner.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dp_info = None
    if world > 1 or force_dp:                      # outside the timed region: what the backward pass did not hide of the exchange step
        ex = torch.tensor([runner.exposed_comm_ms(5)], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        dp_info = {"exposed_comm_ms": round(float(ex.item()), 4), "bucket_bytes": [(e - s_) * 4 for s_, e, _ in runner.buckets],
                   "comm_cus": comm_cus, "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"), "compress": runner.compress}
    losses = eng.losses()
    if rank != 0:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    fwd_flops, step_flops = eng.flops()

    # ---- per-kernel timing with HIP events (library launch profiler), eager launches on the same stream
    hip.prof_enable(True)
    for _ in range(args.profile_steps):
        eng.step_eager()
    torch.cuda.synchronize()
    recs = hip.prof_collect()
    hip.prof_enable(False)
    agg = {}
    for kid, ms, fl, by in recs:
        a = agg.setdefault(hip.KERNEL_NAMES.get(kid, str(kid)), [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl; a[3] += by
    kernels = {k: {"launches_per_step": v[0] // args.profile_steps, "ms_per_step": round(v[1] / args.profile_steps, 4),
                   "tflops": round(v[2] / (v[1] * 1e9), 1) if v[1] > 0 and v[2] > 0 else None,
                   "gbytes_per_s_algorithmic": round(v[3] / (v[1] * 1e6), 1) if v[1] > 0 and v[3] > 0 else None}
               for k, v in agg.items()}
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    dname, (dn, dms, dfl, dby) = dom
    peak = PEAK_MFMA[args.dtype]
    achieved = dfl / (dms * 1e9) if dms > 0 else 0.0
    # HBM traffic per launch of the dominant kernel: measured offline with rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in
    # separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and committed under profiles/
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")) as f:
            pm = json.load(f)
        if pm.get("workload") == [args.backbone, args.batch, args.height, args.width, args.dtype]:
            traffic = pm["igemm_hbm_bytes_per_launch"]
    except Exception:
        pass
    roofline = {"bound": "mfma", "kernel": dname, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "avg_launch_ms": round(dms / dn, 4), "launches": dn // args.profile_steps,
                "algorithmic_flops_per_launch": dfl / dn,
                "whole_step_frac_of_mfma_peak": round(step_flops / (ms_per_step * 1e-3) / 1e12 / peak, 4)}
    pcie = pcie_inclusive(eng, cfg, args.batch, args.pcie_steps) if (args.pcie_steps > 0 and world == 1) else None
    out = {
        "metric": "images/sec fwd+bwd ResNet50 640x512 bs32/GPU", "value": round(value, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bfloat16" else args.dtype,
        "data": "synthetic",
        "config": {"workload": "%s bottleneck=32 ori_resolution=%d soft-class head + loc regression, batch %d/GPU x %dx%dx3, "
                               "full training step (prep+fwd+loss+bwd+clip+SGD%s)" % (args.backbone, args.ori_bins, args.batch, args.height,
                                                                                     args.width, "+RCCL all-reduce" if world > 1 else ""),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world, "comm_cus": comm_cus, "hipgraph": True,
                   "step_tflop": round(step_flops / 1e12, 3), "options": dict(kv.split("=", 1) for kv in args.opt), "loc_loss": losses["loc_loss"], "ori_loss": losses["ori_loss"]},
        "roofline": roofline, "kernels": kernels, "pcie_inclusive": pcie, "dp": dp_info,
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline({"h": args.height, "w": args.width, "backbone": args.backbone, "ori_bins": args.ori_bins},
                                           args.cpu_sample_batch, args.cpu_steps)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
