"""Single-node data parallelism: one process per GPU, replicated weights, minibatch sharded by
rank, ONE exchange step per training step -- a sum-all-reduce (averaged) of the flat fp32
gradient buffer over RCCL/xGMI, issued per bucket while the backward pass is still running.

The reference has no multi-GPU code (config.py:20 GPU_COUNT = 1; a commented-out ParallelModel at
net.py:694-697), so the semantics are the ones a tower-parallel Keras model would have: every
rank computes the loss of ITS shard (incl. the batch-Frobenius rel_loss, net.py:750-762), the
gradients are averaged, and every rank applies the identical global-norm-clipped SGD update.

Buckets are contiguous slices of the flat gradient buffer taken from its END (heads, stage 5 ...)
towards its start, because that is the order in which the backward pass finalises gradients.
xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): buckets are large (default 32 MiB)
so each collective is bandwidth- not latency-bound, and there are few of them.
"""
import os

import torch
import torch.distributed as dist

from . import hip


def plan_buckets(layer_sizes, bucket_bytes=32 << 20, tail_bytes=0):
    """layer_sizes: [(layer_name, start, end)] in FLAT (forward) order, contiguous.
    Returns buckets in the order they become ready (backward order):
    [(start, end, [layer names whose gradients live in the slice])].

    tail_bytes > 0: the layers at the START of the buffer (stem, first stage: the last gradients of a backward pass) that fit in
    tail_bytes form a bucket of their own.  The all-reduce of the LAST bucket is the one nothing is left to hide behind -- capped like
    this it is a latency-sized collective, and the large bucket before it overlaps with the backward pass of the tail's layers
    (ResNet-50: stem + stage 2 hold 0.9 MB of gradients but a quarter of the backward pass's time)."""
    n_tail = 0
    if tail_bytes > 0:
        base = layer_sizes[0][1]
        while n_tail < len(layer_sizes) - 1 and (layer_sizes[n_tail][2] - base) * 4 <= tail_bytes:
            n_tail += 1
    body, tail = layer_sizes[n_tail:], layer_sizes[:n_tail]
    buckets, cur, cur_end = [], [], None
    for name, s, e in reversed(body):
        if cur_end is None:
            cur_end = e
        cur.append(name)
        if (cur_end - s) * 4 >= bucket_bytes:
            buckets.append((s, cur_end, cur))
            cur, cur_end = [], None
    if cur:
        buckets.append((body[0][1], cur_end, cur))
    if tail:
        buckets.append((tail[0][1], tail[-1][2], [name for name, _, _ in reversed(tail)]))
    return buckets


def allreduce_rel_norms(norms, group=None):
    """Sum {sum (gt-pred)^2, sum gt^2} of rel_loss_graph over the ranks (DP_EXACT_REL_LOSS); synchronous w.r.t. the caller's stream."""
    if dist.is_initialized() and dist.get_world_size(group) > 1 or _force_collectives():
        allreduce_sum_(norms, group)
    return norms


DEFAULT_COMM_CUS = 0           # off until an N > 1 A/B settles it: by the one-GPU model (DESIGN.md section 7 vi) reserving 16 CUs breaks even at
                               # about 2 ms of resident collectives per step; URSO_DP_COMM_CUS / comm_cus= switch it on
DEFAULT_TAIL_BYTES = 1 << 20


def reserve_comm_cus(n=None):
    """Call BEFORE dist.init_process_group: bounds RCCL's resident workgroups (one per channel, NCCL_MAX_NCHANNELS; an explicit
    setting in the environment wins) to the CUs DataParallelEngine leaves free, and returns that number (0 = no reservation, RCCL's own
    channel count).  Measured on one GPU with a stand-in kernel (profiles/r02_dp_cu_contention.json): CUs held beside full-size grids cost
    the step +0.24 ms per ms they are held; planning for 16 fewer CUs costs 3 % of every step and still 6 % of the time they are held."""
    if n is None:
        n = int(os.environ.get("URSO_DP_COMM_CUS", str(DEFAULT_COMM_CUS)))
    n = max(0, int(n))
    if n > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))
    return n


def launcher_world():
    """(rank, local_rank, world_size) as a launcher (`python -m torch.distributed.run`, torchrun) put them in the environment;
    (0, 0, 1) in a plain process.  An already initialised process group wins over the environment."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), int(os.environ.get("LOCAL_RANK", "0")), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if world <= 1:
        return 0, 0, 1
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))), world


def init_from_launcher():
    """One process per GPU: pick this rank's device, bound RCCL's channels (reserve_comm_cus) and join the process group the launcher
    describes -- what `net.UrsoNet(mode='training')` does when WORLD_SIZE > 1, so that `pose_estimator.py train` under torchrun is the
    data-parallel run with no change to the caller.  Backend: RCCL ("nccl"); URSO_DP_BACKEND=gloo moves the exchange to gloo on the
    SAME device tensors (ranks that share one GPU, which RCCL refuses: tests/test_dp_two_ranks_gpu.py).  Returns (rank, local_rank, world)."""
    rank, local, world = launcher_world()
    if world <= 1:
        return rank, local, world
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("ursonet_amd: data-parallel training needs an AMD GPU per rank")
    dev = int(os.environ.get("URSO_DP_DEVICE", local % ndev))
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("URSO_DP_BACKEND", "nccl")
        reserve_comm_cus()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def _host_staged(group=None):
    """gloo between ranks whose tensors live on a GPU (two ranks sharing one device, which RCCL refuses; URSO_DP_BACKEND=gloo): collectives
    go through a host copy, synchronously -- a test transport, never the measured one."""
    return dist.is_initialized() and dist.get_backend(group) != "nccl"


def broadcast_(t, src=0, group=None):
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def allreduce_sum_(t, group=None):
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def average_over_ranks(t, group=None):
    """In place: t <- mean over the ranks of t (a small tensor of logged scalars); identity without a process group."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
        else:
            allreduce_sum_(t, group)
            t.div_(dist.get_world_size(group))
    return t


def _force_collectives():
    import os
    return dist.is_initialized() and os.environ.get("URSO_DP_FORCE_COLLECTIVES", "0") == "1"


class GradReducer(object):
    """Averages slices of a flat gradient tensor across the process group, asynchronously.

    compress='bf16' halves the bytes on the wire (xGMI is the DP bottleneck candidate: 134 MB of fp32 gradients per step at
    cfg2): each bucket is rounded to bf16 before the all-reduce and expanded afterwards, with ERROR FEEDBACK in fp32 -- what the
    rounding dropped on this rank is kept (`resid`) and added to the next step's gradient, so the fp32 master weights see every
    bit of gradient eventually (the rounding error does not accumulate as a bias).  Default None: exact fp32 averaging."""

    def __init__(self, flat_g, buckets, group=None, compress=None, comm=None):
        assert compress in (None, "bf16")
        self.flat_g, self.buckets, self.group = flat_g, buckets, group
        self.compress = compress
        self.comm = comm                    # ursonet_amd.hip.Comm: the C-ABI transport (urso_comm_*) instead of torch.distributed
        if compress:
            self.cbuf = torch.zeros_like(flat_g, dtype=torch.bfloat16)
            self.resid = torch.zeros_like(flat_g)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.works = []
        import os
        self.force = os.environ.get("URSO_DP_FORCE_COLLECTIVES", "0") == "1"     # exercise the RCCL calls with one rank

    def launch(self, k):
        """Start the all-reduce of bucket k (call after the kernels producing it were enqueued)."""
        if self.world == 1 and not self.force and self.comm is None:
            return
        s, e, _ = self.buckets[k]
        t = self.flat_g[s:e]
        back = None
        if self.compress:
            r, c = self.resid[s:e], self.cbuf[s:e]
            if t.is_cuda:                   # one library pass (urso_bucket_round_ef): c = bf16(g + r), r = (g + r) - float(c); g is only read
                hip.bucket_round_ef(t, r, c)
            else:                           # CPU tensors (the gloo tests of the schedule): the same arithmetic in torch
                t.add_(r)                   # error feedback: last step's rounding remainder
                c.copy_(t)                  # round to bf16
                torch.sub(t, c.float(), out=r)  # what this rounding dropped, kept for the next step
            back, t = t, c
        if self.comm is not None:           # urso_comm_allreduce_bucket: RCCL average on the communicator's own stream
            self.comm.allreduce_bucket(t)
            self.works.append((None, None, back, t))
        elif self.backend == "nccl":        # RCCL: averaging happens inside the collective
            self.works.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None, back, t))
        elif t.is_cuda:                     # gloo under device tensors (ranks sharing one GPU): host-staged and synchronous, see _host_staged
            allreduce_sum_(t, self.group)
            self.works.append((None, t, back, t))
        else:                               # gloo (CPU tests): sum, then scale on wait
            self.works.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t, back, t))

    def wait_all(self):
        if self.comm is not None and self.works:
            self.comm.wait()                # the compute stream waits for every bucket launched so far
        for w, scale, back, t in self.works:
            if w is not None:
                w.wait()
            if scale is not None:
                scale.div_(self.world)
            if back is not None:            # averaged bf16 gradient -> the fp32 buffer the optimizer reads
                if back.is_cuda:
                    hip.bucket_expand_bf16(t, back)
                else:
                    back.copy_(t)
        self.works = []


class _OptionOverride(object):
    """Process-wide kernel-policy options that DataParallelEngine overrides while collectives run beside the step (`cus`, `hconv_streamk`),
    reference-counted: wrappers nest, the value from before the FIRST override comes back when the LAST wrapper that asked for it closes,
    and a second wrapper asking for another value of a held option is an error instead of a silent split-count mismatch."""
    held = {}                               # option -> [value before the first override, override value, count]

    @classmethod
    def acquire(cls, name, value):
        h = cls.held.get(name)
        if h is None:
            cls.held[name] = [hip.get_option(name), int(value), 1]
            hip.set_option(name, int(value))
        elif h[1] != int(value):
            raise RuntimeError("option %r is held at %d by another DataParallelEngine; this one needs %d -- close() the other first" % (name, h[1], value))
        else:
            h[2] += 1

    @classmethod
    def release(cls, name):
        """True when this release restored the option (the last holder is gone)."""
        h = cls.held.get(name)
        if h is None:
            return False
        h[2] -= 1
        if h[2] > 0:
            return False
        hip.set_option(name, h[0])
        del cls.held[name]
        return True


def _release_holds(holds):
    """weakref.finalize callback of DataParallelEngine: the process-wide options a collected wrapper still held."""
    for name in list(holds):
        _OptionOverride.release(name)
    del holds[:]


class DataParallelEngine(object):
    """Wraps an Engine: broadcasts the initial weights, splits the captured step into
    [prep+forward+loss+backward-part-0], [backward-part-1], ..., [optimizer] hipGraphs and
    interleaves the bucket all-reduces between their replays.  Construct it BEFORE the first load_batch: a bucket size or comm_cus
    that differs from the engine's plan rebuilds the plan, which reallocates the activation (and input) buffers."""

    def __init__(self, engine, bucket_bytes=32 << 20, group=None, compress=None, comm=None, comm_cus=None, tail_bytes=None):
        self.eng, self.group, self.compress, self.comm = engine, group, compress, comm
        self._exposed_events = None
        # CUs left to the collective's resident workgroups.  Every persistent conv grid is a static partition of the tile stream over
        # "all CUs"; a CU held by an RCCL workgroup makes the blocks that wanted it wait for a whole stream to finish -- a second wave.
        # With comm_cus > 0 the library plans grids, weight-gradient splits and workspaces for (CUs - comm_cus) instead (option `cus`,
        # include/ursonet_hip.h), process-wide: one process drives one GPU.  tools/dp_cu_contention.py measures both sides of the
        # trade on one GPU (profiles/r02_dp_cu_contention.json).  Default: URSO_DP_COMM_CUS, else DEFAULT_COMM_CUS = 0 = full grids
        # (reserve_comm_cus above keeps RCCL inside a non-zero reservation).  Only applied when collectives actually run (world > 1).
        explicit = comm_cus is not None              # a forced-collective run with one rank reserves only when asked to
        if comm_cus is None:
            comm_cus = int(os.environ.get("URSO_DP_COMM_CUS", str(DEFAULT_COMM_CUS)))
        self.comm_cus = max(0, int(comm_cus))
        self.world = dist.get_world_size(group)
        # DP_EXACT_REL_LOSS: the location loss is ONE ratio of norms over the global batch (net.py:750-762); its two squared norms are
        # summed over the ranks between forward and backward and the gradient is pre-scaled by the world size (undone by the averaging)
        self.rel_exact = bool(getattr(engine, "rel_exact", False)) and bool(engine.loss_pre_ops)
        if self.rel_exact:
            engine.rel_scale.fill_(float(self.world))
        eng = engine
        broadcast_(eng.flat_w, 0, group)
        broadcast_(eng.flat_stats, 0, group)
        # gradient buckets are planned by the engine (it batches the gradient finalisation per bucket)
        replan = False
        self._holds = []                                # options this wrapper holds (_OptionOverride), given back by close()
        try:
            self._init_policy(eng, explicit, comm, bucket_bytes, tail_bytes, replan)
        except BaseException:
            self.close()                                # an exception here must not leave the process on the reduced chip / without stream-K
            raise
        # a wrapper that is dropped without close() must not keep `cus` / `hconv_streamk` held for the rest of the process
        import weakref
        self._finalizer = weakref.finalize(self, _release_holds, self._holds)
        self._finalizer.atexit = False                  # (nothing to give back at interpreter exit)

    def _init_policy(self, eng, explicit, comm, bucket_bytes, tail_bytes, replan):
        if self.comm_cus and eng.device.type == "cuda" and (self.world > 1 or (explicit and _force_collectives())):
            total = torch.cuda.get_device_properties(eng.device).multi_processor_count
            usable = max(8, total - self.comm_cus)
            if hip.get_option("cus") != usable or "cus" in _OptionOverride.held:
                before = hip.get_option("cus")
                _OptionOverride.acquire("cus", usable)  # split counts follow it: the plan below must be rebuilt under the new value (process-wide until close())
                self._holds.append("cus")
                replan = before != usable
        # conv_halo.hip's accumulator hand-over between blocks ("stream-K") needs every block of the launch resident: a finishing block spins on
        # the pieces of the runs behind it.  RCCL's workgroups hold CUs while the backward pass runs, so a producer can sit in the queue behind
        # them for a collective's duration with a consumer CU spinning on it.  No hand-over while collectives run beside the step (the
        # whole-tile kernel of conv_halo2.hip takes those layers; option hconv_streamk, restored by close()).
        if eng.device.type == "cuda" and (self.world > 1 or _force_collectives() or comm is not None) and (
                hip.get_option("hconv_streamk") or "hconv_streamk" in _OptionOverride.held):
            _OptionOverride.acquire("hconv_streamk", 0)
            self._holds.append("hconv_streamk")
            eng._graphs = None                          # the kernel choice is made at launch time: re-capture under the new policy
        # the last bucket's all-reduce has nothing left to hide behind: cap it (plan_buckets) when ranks really exchange gradients.  On one
        # GPU the extra finalisation group costs 0.06 ms (0.7 %) and buys nothing, so the plain engine keeps its plan.
        if tail_bytes is None:
            tail_bytes = DEFAULT_TAIL_BYTES if self.world > 1 else eng.grad_tail_bytes
        self._prev_flags = (bool(getattr(eng, "no_wgrad_fork", False)), bool(getattr(eng, "no_fused_sqnorm", False)))
        if getattr(eng, "wgrad_stream", None) is not None or not getattr(eng, "no_wgrad_fork", False):
            # the step runs as graph segments between collectives here, not through Engine.capture (which checks a forked graph against the chain):
            # the backward pass stays one chain
            replan = replan or getattr(eng, "wgrad_stream", None) is not None
            eng.no_wgrad_fork = True
        if getattr(eng, "fused_sqnorm", False) or not getattr(eng, "no_fused_sqnorm", False):
            eng.no_fused_sqnorm = True                  # the clip norm is that of the AVERAGED gradient: taken after the all-reduces, by urso_sqnorm
            replan = replan or bool(getattr(eng, "fused_sqnorm", False))
        if bucket_bytes != eng.grad_bucket_bytes or int(tail_bytes) != eng.grad_tail_bytes or replan:
            eng.grad_bucket_bytes = int(bucket_bytes)
            eng.grad_tail_bytes = int(tail_bytes)
            eng._graphs = None
            eng._build_plan()
        self._derive_cuts()

    def close(self):
        """Give the CUs reserved for the collectives back: option `cus` is process-wide, and an Engine planned after this wrapper is gone
        would otherwise size its grids for the smaller chip."""
        holds, self._holds = list(getattr(self, "_holds", [])), []
        fin = getattr(self, "_finalizer", None)
        if fin is not None:
            fin.detach()                                # close() ran: nothing left for the collector to release
            self._finalizer = None
        prev = getattr(self, "_prev_flags", None)
        if prev is not None:
            # the engine goes back to what it was before it was wrapped: used on its own again it gets its forked backward pass and the
            # gradient norm from the finalisation blocks back (close() used to leave both switched off)
            self._prev_flags = None
            eng = self.eng
            if (bool(getattr(eng, "no_wgrad_fork", False)), bool(getattr(eng, "no_fused_sqnorm", False))) != prev:
                eng.no_wgrad_fork, eng.no_fused_sqnorm = prev
                eng._graphs = None
                if "cus" not in holds and hasattr(eng, "_build_plan"):
                    eng._build_plan()                   # (with `cus` held the re-plan below does it, under the restored option)
        if "hconv_streamk" in holds:
            _OptionOverride.release("hconv_streamk")
            self.eng._graphs = None
        if "cus" in holds:
            restored = _OptionOverride.release("cus")
            # the wrapped engine planned its split counts, partial workspaces and grouped weight-gradient launches for the reduced CU count:
            # once the option is back (the last wrapper holding it is gone) re-plan it for the whole chip (its captured graph goes with the
            # plan; plan_version moves); while another wrapper still holds the reduced chip this engine's plan already matches the option
            self.eng._graphs = None
            if restored or prev is not None:
                self.eng._build_plan()
        self._graphs = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _derive_cuts(self):
        """Bucket list, reducer and the backward cut points of the engine's CURRENT plan (set_trainable / a bucket-size change
        rebuild the plan: stale cuts would start an all-reduce before its bucket's finalisation ran)."""
        eng, group = self.eng, self.group
        self.plan_version = eng.plan_version
        self.buckets = eng.buckets
        prev = getattr(self, "reducer", None)
        self.reducer = GradReducer(eng.flat_g, self.buckets, group, compress=self.compress, comm=self.comm)
        if prev is not None and self.compress and prev.flat_g is eng.flat_g:
            self.reducer.resid, self.reducer.cbuf = prev.resid, prev.cbuf      # a re-plan (set_trainable) keeps the error-feedback remainder
            # ... of the parameters that still train: a frozen layer's gradient slice stays zero, and its old rounding remainder would be
            # added to it, all-reduced and applied to the frozen weights (the optimizer runs over the whole flat buffer)
            ranges = getattr(eng, "trainable_ranges", None)
            if ranges is not None:
                live = torch.zeros_like(self.reducer.resid, dtype=torch.bool)
                for (s_, e_) in ranges():
                    live[s_:e_] = True
                self.reducer.resid.mul_(live)
        # split the backward op list where each bucket becomes complete
        last_op_of_layer = {}
        for i, (tag, _) in enumerate(eng.bwd_ops):
            for name in ((tag,) if isinstance(tag, str) else (tag or ())):       # an op may complete several layers (batched)
                last_op_of_layer[name] = i
                bn = eng.convs[name].bn
                if bn:
                    last_op_of_layer[bn] = i
        cuts = []
        for (_, _, names) in self.buckets:
            idx = [last_op_of_layer[n] for n in names if n in last_op_of_layer]
            cuts.append(max(idx) + 1 if idx else 0)
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1])
        self.cuts = cuts
        self._graphs = None

    def _segments(self):
        eng = self.eng
        segs, prev = [], 0
        for k, c in enumerate(self.cuts):
            ops = [op for _, op in eng.bwd_ops[prev:c]]
            if k == 0:
                ops = ([] if self.rel_exact else eng.prep_ops + eng.fwd_ops + eng.loss_pre_ops) + eng.loss_ops + ops
            segs.append(ops)
            prev = c
        tail = [op for _, op in eng.bwd_ops[prev:]]
        return segs, tail + eng.opt_ops

    def capture(self):
        eng = self.eng
        segs, last = self._segments()
        torch.cuda.synchronize(eng.device)
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(side):
            saved = eng.save_train_state()
            eng.step_eager()
            eng.restore_train_state(saved)
        torch.cuda.current_stream(eng.device).wait_stream(side)
        torch.cuda.synchronize(eng.device)
        from .engine import _no_gc
        with hip.capture_lock, _no_gc():     # no engine / graph may be garbage-collected, no feeder thread may enter HIP, while a stream is capturing
            self._capture_graphs(eng, segs, last)

    def _capture_graphs(self, eng, segs, last):
        graphs = []
        pool = None
        self._pre_graph = None
        if self.rel_exact:                   # [prep + forward + norms]  -> all-reduce of 2 floats ->  [loss + backward ...]
            self._pre_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._pre_graph):
                for op in eng.prep_ops + eng.fwd_ops + eng.loss_pre_ops:
                    op()
            pool = self._pre_graph.pool()
        for ops in segs + [last]:
            if not ops:                      # e.g. a bucket made only of frozen layers
                graphs.append(None)
                continue
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, pool=pool):
                for op in ops:
                    op()
            pool = gr.pool()
            graphs.append(gr)
        self._graphs = graphs

    def step(self):
        if self.plan_version != self.eng.plan_version:
            self._derive_cuts()
        if self._graphs is None:
            self.capture()
        if self._pre_graph is not None:
            self._pre_graph.replay()
            allreduce_rel_norms(self.eng.rel_norms, self.group)       # the compute stream waits for it before the next replay
        for k in range(len(self.buckets)):
            if self._graphs[k] is not None:
                self._graphs[k].replay()
            self.reducer.launch(k)           # RCCL runs on its own stream, ordered after the replay
        ev = self._exposed_events
        if ev is not None:
            ev[0].record()                   # the backward pass is enqueued up to here ...
        self.reducer.wait_all()              # compute stream waits for every bucket
        if ev is not None:
            ev[1].record()                   # ... and this one completes only once the last collective has
        self._graphs[-1].replay()            # global-norm clip + momentum SGD on the averaged gradient

    def step_eager(self):
        """The same schedule as step() launched op by op instead of by graph replay (debugging; the gloo CPU test of the schedule):
        backward segment k, then the all-reduce of bucket k, ..., every collective joined, then the tail of the backward pass and
        the optimizer."""
        if self.plan_version != self.eng.plan_version:
            self._derive_cuts()
        if hasattr(self.eng, "_check_plan_options"):
            self.eng._check_plan_options()
        segs, last = self._segments()
        if self.rel_exact:
            for op in self.eng.prep_ops + self.eng.fwd_ops + self.eng.loss_pre_ops:
                op()
            allreduce_rel_norms(self.eng.rel_norms, self.group)
        for k, ops in enumerate(segs):
            for op in ops:
                op()
            self.reducer.launch(k)
        self.reducer.wait_all()
        for op in last:
            op()

    def exposed_comm_ms(self, steps=5):
        """Mean time per step the compute stream spends waiting for collectives after its last backward kernel: the part of the exchange
        step that the backward pass did not hide (two events around the join, outside the captured graphs)."""
        dev = self.eng.device
        total = 0.0
        for _ in range(steps):
            self._exposed_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.step()
            torch.cuda.synchronize(dev)
            total += self._exposed_events[0].elapsed_time(self._exposed_events[1])
        self._exposed_events = None
        return total / max(steps, 1)
