"""Data-parallel path on CPU: 2 processes, gloo backend, 127.0.0.1 rendezvous.

Covers (a) bucket planning over the flat gradient buffer, (b) the GradReducer's asynchronous
bucket averaging, (c) the equivalence "2 ranks x batch 2 == 1 rank x batch 4" for the
shard-decomposable losses (soft-label cross-entropy on both heads: classify_loc + classify_ori),
using the oracle's gradients as the per-rank gradients, and (d) the documented NON-equivalence
of the batch-Frobenius rel_loss (net.py:750-762) under per-rank loss semantics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import make_config, synthetic_batch


def test_plan_buckets_backward_order():
    from ursonet_amd.dp import plan_buckets
    layers = [("a", 0, 1000), ("b", 1000, 3000), ("c", 3000, 3500), ("d", 3500, 9000), ("e", 9000, 9100)]
    b = plan_buckets(layers, bucket_bytes=2000 * 4)
    assert b[0] == (3500, 9100, ["e", "d"]) and b[1] == (1000, 3500, ["c", "b"]) and b[2] == (0, 1000, ["a"])
    covered = sorted((s, e) for s, e, _ in b)
    assert covered[0][0] == 0 and covered[-1][1] == 9100 and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    one = plan_buckets(layers, bucket_bytes=1 << 30)
    assert one == [(0, 9100, ["e", "d", "c", "b", "a"])]
    # capped tail: the layers at the start of the buffer that fit form the last bucket on their own
    t = plan_buckets(layers, bucket_bytes=1 << 30, tail_bytes=3000 * 4)
    assert t == [(3000, 9100, ["e", "d", "c"]), (0, 3000, ["b", "a"])]
    assert plan_buckets(layers, bucket_bytes=1 << 30, tail_bytes=100) == one           # nothing fits: unchanged
    assert plan_buckets(layers, bucket_bytes=1 << 30, tail_bytes=1 << 30)[-1] == (0, 9000, ["d", "c", "b", "a"])   # never the whole buffer


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _flat(grads, order):
    return torch.cat([grads[ln][wn].reshape(-1) for ln, wn in order])


def _worker(rank, world, port, regress_loc, out, exact=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import graph_ref as G
    from ursonet_amd.dp import GradReducer, plan_buckets
    cfg = make_config("resnet18", 64, 64, batch=2, regress_ori=False, regress_loc=regress_loc, ori_bins=2, loc_bins=2,
                      bottleneck=8, branch=32)
    W = G.init_params(cfg, 7, randomize_bn=True)
    img, loc, ori, _ = synthetic_batch(cfg, 4, seed=11)              # the global batch; rank r takes samples [2r, 2r+2)
    sl = slice(2 * rank, 2 * rank + 2)
    P = G.to_torch(W)
    rel_global = None
    if exact:
        # DP_EXACT_REL_LOSS: this rank's two squared norms, summed over the ranks by the product's helper
        from ursonet_amd.dp import allreduce_rel_norms
        with torch.no_grad():
            pred_loc, _, _, _ = G.losses(P, torch.tensor(img[sl]), torch.tensor(loc[sl]), torch.tensor(ori[sl]), cfg)
        norms = allreduce_rel_norms(G.rel_norms(torch.tensor(loc[sl]), pred_loc))
        rel_global = (norms, world)
    grads, _, _ = G.gradients(P, torch.tensor(img[sl]), torch.tensor(loc[sl]), torch.tensor(ori[sl]), cfg, rel_global=rel_global)
    order = [(ln, wn) for ln in grads for wn in grads[ln]]
    flat = _flat(grads, order).clone()
    sizes, off = [], 0
    for ln in grads:
        n = sum(grads[ln][wn].numel() for wn in grads[ln])
        sizes.append((ln, off, off + n)); off += n
    buckets = plan_buckets(sizes, bucket_bytes=64 << 10, tail_bytes=48 << 10)     # with the capped stem-side bucket a world size > 1 gets
    assert (buckets[-1][1] - buckets[-1][0]) * 4 <= 48 << 10 and buckets[-1][0] == 0 and len(buckets) >= 2
    red = GradReducer(flat, buckets)
    for k in range(len(buckets)):
        red.launch(k)
    red.wait_all()
    if rank == 0:
        Pf = G.to_torch(W)
        gfull, _, _ = G.gradients(Pf, torch.tensor(img), torch.tensor(loc), torch.tensor(ori), cfg)
        full = _flat(gfull, order)
        out["nb"] = len(buckets)
        out["err"] = float((flat - full).abs().max() / full.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def _run(regress_loc, exact=False):
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, regress_loc, out, exact), nprocs=2, join=True)
    return dict(out)


def test_dp_gradient_equals_big_batch_for_decomposable_losses():
    out = _run(regress_loc=False)
    assert out["nb"] >= 2
    assert out["err"] < 2e-5, out                                   # fp32 round-off only


def test_dp_rel_loss_is_per_rank_not_global():
    """rel_loss is a ratio of batch-wide norms: the mean of shard gradients differs from the
    big-batch gradient; the build documents and keeps per-rank semantics."""
    out = _run(regress_loc=True)
    assert out["err"] > 1e-3, out


def test_dp_exact_rel_loss_equals_big_batch():
    """DP_EXACT_REL_LOSS: with the two norms summed over the ranks (ursonet_amd.dp.allreduce_rel_norms) and the gradient
    pre-scaled by the world size, the AVERAGED shard gradients equal the gradient of the one global-batch loss."""
    out = _run(regress_loc=True, exact=True)
    assert out["err"] < 2e-5, out


def _worker_bf16(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ursonet_amd.dp import GradReducer
    n, steps = 5000, 40
    buckets = [(3000, 5000, ["b"]), (0, 3000, ["a"])]
    gen = torch.Generator().manual_seed(100 + rank)
    flat = torch.zeros(n)
    red = GradReducer(flat, buckets, compress="bf16")
    acc, exact = torch.zeros(n), torch.zeros(n)
    worst_single = 0.0
    for _ in range(steps):
        g = torch.randn(n, generator=gen) * 0.3 + 1.0
        gs = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(gs, g)
        mean = sum(gs) / world
        flat.copy_(g)
        for k in range(len(buckets)):
            red.launch(k)
        red.wait_all()
        worst_single = max(worst_single, float((flat - mean).abs().max() / mean.abs().max()))
        acc += flat; exact += mean
    if rank == 0:
        out["single"] = worst_single
        out["accumulated"] = float((acc - exact).abs().max() / exact.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_buckets_with_error_feedback():
    """GradReducer(compress='bf16'): one step's averaged gradient carries bf16 rounding (~4e-3 of max); over many steps -- what
    momentum SGD integrates -- each rank's own rounding cancels (error feedback), leaving only the rounding of the bf16 additions
    inside the collective: the 40-step sum is 5-10x closer to the exact one than a single step is."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_bf16, args=(2, _free_port(), out), nprocs=2, join=True)
    assert 5e-4 < out["single"] < 8e-3, dict(out)
    assert out["accumulated"] < 2e-3 and out["accumulated"] < 0.3 * out["single"], dict(out)


def test_reserved_comm_cus_bound_rccl_channels(monkeypatch):
    """reserve_comm_cus: the CUs DataParallelEngine plans around are also the bound handed to RCCL (one resident workgroup per
    channel); an explicit NCCL_MAX_NCHANNELS in the environment is left alone, 0 switches the reservation off."""
    from ursonet_amd import dp
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("URSO_DP_COMM_CUS", raising=False)
    assert dp.reserve_comm_cus() == dp.DEFAULT_COMM_CUS == 0 and "NCCL_MAX_NCHANNELS" not in os.environ        # default: no reservation
    assert dp.reserve_comm_cus(16) == 16 and os.environ["NCCL_MAX_NCHANNELS"] == "16"
    assert dp.reserve_comm_cus(24) == 24 and os.environ["NCCL_MAX_NCHANNELS"] == "16"      # a setting already in the environment wins
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    monkeypatch.setenv("URSO_DP_COMM_CUS", "8")
    assert dp.reserve_comm_cus() == 8 and os.environ["NCCL_MAX_NCHANNELS"] == "8"
