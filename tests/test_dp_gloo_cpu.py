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


class _FakeConv(object):
    def __init__(self, bn):
        self.bn = bn


def _schedule_worker(rank, world, port, out):
    """DataParallelEngine over a FAKE engine (an op list + buckets, CPU tensors): the schedule logic of _derive_cuts / _segments /
    step_eager without a GPU."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ursonet_amd.dp import DataParallelEngine, plan_buckets
    # six layers in forward order a..f, each with a BN sibling; gradients are produced in backward order (f first); layer c's
    # gradient is completed late (its "finalize" op comes after b's weight gradient): cuts must follow the LAST producing op
    sizes = [("a", 0, 100), ("bn_a", 100, 110), ("b", 110, 400), ("bn_b", 400, 420), ("c", 420, 900), ("bn_c", 900, 910),
             ("d", 910, 1500), ("bn_d", 1500, 1520), ("e", 1520, 1800), ("bn_e", 1800, 1830), ("f", 1830, 2000), ("bn_f", 2000, 2010)]
    n = 2010
    log = []

    class Eng(object):
        pass
    eng = Eng()
    eng.device = torch.device("cpu")
    eng.flat_w, eng.flat_stats, eng.flat_g = torch.full((n,), float(rank)), torch.full((8,), float(rank)), torch.zeros(n)
    eng.rel_exact, eng.loss_pre_ops, eng.plan_version = False, [], 1
    eng.grad_bucket_bytes, eng.grad_tail_bytes = 600 * 4, 0
    eng.buckets = plan_buckets(sizes, bucket_bytes=600 * 4)
    eng.convs = {k: _FakeConv("bn_" + k) for k in "abcdef"}
    span = {nm: (s, e) for nm, s, e in sizes}

    def producer(names, tag):
        def op():
            log.append(("op", tag))
            for nm in names:
                for x in (nm, "bn_" + nm):
                    s, e = span[x]
                    eng.flat_g[s:e] = (rank + 1) * (1.0 + s)          # rank-dependent: the average is 1.5 * (1 + s)
        return op

    def noop(tag):
        return lambda: log.append(("op", tag))
    eng.prep_ops, eng.fwd_ops, eng.loss_ops = [noop("prep")], [noop("fwd")], [noop("loss")]
    eng.bwd_ops = [("f", noop("wgrad:f")), (None, noop("dgrad:f")), (("f", "e"), producer("fe", "finalize:f,e")), ("d", producer("d", "wgrad:d")),
                   (None, noop("dgrad:d")), ("b", noop("wgrad:b")), (("c", "b"), producer("cb", "finalize:c,b")), (None, noop("dgrad:b")),
                   ("a", producer("a", "wgrad:a")), (None, noop("tail"))]
    seen = {}

    def opt():
        log.append(("op", "optimizer"))
        seen["g"] = eng.flat_g.clone()
    eng.opt_ops = [opt]
    eng._graphs = None
    eng._build_plan = lambda: None
    dp = DataParallelEngine(eng, bucket_bytes=600 * 4, tail_bytes=0)
    launch0 = dp.reducer.launch

    def launch(k):
        log.append(("allreduce", k))
        launch0(k)
    dp.reducer.launch = launch
    dp.step_eager()
    out.put((rank, list(dp.buckets), list(dp.cuts), log, seen["g"].numpy(), eng.flat_w.numpy().copy()))
    dist.destroy_process_group()


def test_dp_schedule_launches_every_bucket_after_its_last_producer_and_before_the_optimizer():
    """2 ranks (gloo): the backward op list is cut where each gradient bucket becomes complete -- after the LAST op that produces any
    layer (or BN sibling) of the bucket -- the bucket's all-reduce is launched right there, all of them are joined before the optimizer,
    which sees the rank-averaged gradient; initial weights are rank 0's."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_schedule_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([out.get(timeout=120) for _ in ps], key=lambda r: r[0])
    for p in ps:
        p.join(60)
    for rank, buckets, cuts, log, g, w in res:
        assert np.all(w == 0.0)                                        # broadcast from rank 0
        names = [b[2] for b in buckets]
        assert sorted(sum(names, [])) == sorted(["a", "b", "c", "d", "e", "f"] + ["bn_" + k for k in "abcdef"])
        pos = {entry: i for i, entry in enumerate(log)}
        last_producer = {"a": "wgrad:a", "b": "finalize:c,b", "c": "finalize:c,b", "d": "wgrad:d", "e": "finalize:f,e", "f": "finalize:f,e"}
        for k, nm in enumerate(names):
            after = max(pos[("op", last_producer[x.replace("bn_", "")])] for x in nm)
            assert pos[("allreduce", k)] > after, (k, nm, log)
            # ... and immediately behind the segment that ends there: no later producer of ANOTHER bucket was launched first unless that
            # bucket's own cut lies later
            assert pos[("allreduce", k)] < pos[("op", "optimizer")]
        assert [e for e in log if e[0] == "allreduce"] == [("allreduce", k) for k in range(len(buckets))]
        assert log[-1] == ("op", "optimizer") and log[-2] == ("op", "tail")
        assert cuts == sorted(cuts)
        s_of = {nm: s for nm, s, e in [("a", 0, 100), ("bn_a", 100, 110), ("b", 110, 400), ("bn_b", 400, 420), ("c", 420, 900), ("bn_c", 900, 910),
                                      ("d", 910, 1500), ("bn_d", 1500, 1520), ("e", 1520, 1800), ("bn_e", 1800, 1830), ("f", 1830, 2000), ("bn_f", 2000, 2010)]}
        for nm, s in s_of.items():
            assert abs(g[s] - 1.5 * (1.0 + s)) < 1e-4 * (1.0 + s), (nm, g[s])


def _replan_worker(port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from ursonet_amd.dp import DataParallelEngine, plan_buckets
    sizes = [("a", 0, 100), ("b", 100, 400), ("c", 400, 1000)]

    class Eng(object):
        pass
    eng = Eng()
    eng.device = torch.device("cpu")
    eng.flat_w, eng.flat_stats, eng.flat_g = torch.zeros(1000), torch.zeros(8), torch.zeros(1000)
    eng.rel_exact, eng.loss_pre_ops, eng.plan_version = False, [], 1
    eng.grad_bucket_bytes, eng.grad_tail_bytes = 600 * 4, 0
    eng.buckets = plan_buckets(sizes, bucket_bytes=600 * 4)
    eng.convs = {k: _FakeConv(None) for k in "abc"}
    eng.prep_ops, eng.fwd_ops, eng.loss_ops, eng.opt_ops = [], [], [], []
    eng.bwd_ops = [("c", lambda: None), ("b", lambda: None), ("a", lambda: None)]
    eng._graphs = None
    eng._build_plan = lambda: None
    live = [[(0, 1000)]]
    eng.trainable_ranges = lambda: live[0]
    dp = DataParallelEngine(eng, bucket_bytes=600 * 4, tail_bytes=0, compress="bf16")
    dp.reducer.resid.fill_(0.25)                         # rounding remainders of a previous step, every parameter
    live[0] = [(100, 400)]                               # set_trainable: only layer b still trains
    eng.plan_version += 1
    dp._derive_cuts()
    out.put(dp.reducer.resid.numpy().copy())
    dist.destroy_process_group()


def test_error_feedback_remainder_of_frozen_layers_is_dropped_on_replan():
    """bf16 gradient buckets keep what the rounding dropped for the next step (error feedback).  After a re-plan (set_trainable) the
    remainder of layers that no longer train must not survive: their gradient slice stays zero, so the old remainder would be added to
    it, all-reduced and applied to the frozen weights by the optimizer (which runs over the whole flat buffer)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_replan_worker, args=(_free_port(), out))
    p.start()
    r = out.get(timeout=120)
    p.join(60)
    assert np.all(r[100:400] == 0.25) and np.all(r[:100] == 0.0) and np.all(r[400:] == 0.0)


def _worker_helpers(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ursonet_amd import dp
    assert dp.launcher_world() == (rank, rank, world)                 # from the launcher's environment, before a group exists
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert dp.launcher_world() == (rank, rank, world)                 # ... and from the group once it does
    t = torch.full((3, 4), float(rank + 1))
    dp.average_over_ranks(t)
    w = torch.arange(5, dtype=torch.float32) * (rank + 1)
    dp.broadcast_(w, 0)
    s = torch.tensor([1.0 + rank, 10.0])
    dp.allreduce_sum_(s)
    out[rank] = (t.tolist(), w.tolist(), s.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_launcher_helpers_of_the_drop_in_boundary(monkeypatch):
    """What net.UrsoNet / UrsoNet.train use under a launcher (ursonet_amd/dp.py): rank / world from the environment, the rank-averaged loss
    history, the weight broadcast and the two-scalar sum of exact rel_loss -- two gloo ranks on CPU tensors.  A plain process is (0, 0, 1)."""
    from ursonet_amd import dp
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert dp.launcher_world() == (0, 0, 1)
    t = torch.ones(2)
    assert dp.average_over_ranks(t) is t and t.tolist() == [1.0, 1.0]          # no process group: identity
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_helpers, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        t, w, s = out[rank]
        assert t == [[1.5] * 4] * 3 and w == [0.0, 1.0, 2.0, 3.0, 4.0] and s == [3.0, 20.0]
