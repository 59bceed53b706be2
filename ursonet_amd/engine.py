"""Static execution plan of the UrsoNet hot path on one MI355X.

The reference runs its graph through Keras/TensorFlow sessions (net.py:1152-1163 fit_generator,
net.py:1251 predict).  Here the graph (ursonet_amd/graph.py) is lowered ONCE into a flat list of
C-ABI kernel launches over pre-allocated HBM buffers -- weight prep (BN folding), forward, fused
loss+gradient, backward (dgrad/wgrad), parameter-gradient finalisation, global-norm clip + momentum
SGD -- and the whole step is captured into a hipGraph and replayed (no per-step Python, no
allocator, no host sync).  PyTorch is used only for device memory, streams and graph capture.

Data layout in HBM: activations NHWC in the compute dtype; all trainable parameters live in ONE
flat fp32 buffer (Keras layouts, forward layer order) with matching flat gradient / momentum
buffers, so that the data-parallel all-reduce works on contiguous slices (ursonet_amd/dp.py)
and the optimizer is two launches.
"""
import math
import os
import re
from collections import OrderedDict

import numpy as np
import torch

from . import hip
from .config import compute_dtype_name
from .graph import BN_EPS, build_graph, conv_flops

BN_MOMENTUM = 0.99          # Keras BatchNormalization default (net.py:60-76 passes none)

_DT = {"float32": hip.F32, "bfloat16": hip.BF16, "float16": hip.F16}


def _round_up(v, m):
    return (v + m - 1) // m * m


class _Act(object):
    """An activation tensor in HBM (+ its gradient buffer, allocated on demand)."""

    def __init__(self, eng, spec, numel, dtype):
        self.spec, self.numel = spec, numel
        self.fused_pool = False      # conv1's output when urso_stem_conv_pool computes the max-pool as well: never stored
        self.data = torch.empty(numel, dtype=dtype, device=eng.device)
        self.grad = None
        self.grad_written = False
        self.pending = None          # gradient tensor to be folded into the next dgrad into this tensor
        self.bits = None             # ReLU bit mask (1 byte per 8 elements), written by the producing conv's forward epilogue
        self.compact = None          # (H, W): the gradient buffer holds only the even rows / columns of the pixel grid ([B, H/2, W/2, C])
        self.pending_hw = None       # same, for a pending residual gradient
        self.eng = eng

    @property
    def data(self):
        if self.fused_pool:
            raise RuntimeError("T%d is conv1's output inside the fused conv1 + ReLU + max-pool kernel: it is never stored (option stem_pool = 0 keeps it)" % self.spec.id)
        return self._data

    @data.setter
    def data(self, t):
        self._data = t

    def grad_buf(self):
        if self.grad is None:
            self.grad = torch.empty(self.numel, dtype=self.data.dtype, device=self.eng.device)
        return self.grad


class _Conv(object):
    pass


_SIDE_STREAMS = {}


def _side_stream(device):
    """One second stream per device for the process, shared by every Engine: engines come and go (tests build dozens), and a replay of a forked
    graph segfaulted inside hipGraphLaunch (ROCm 7.2) in a process where earlier engines' side streams had been destroyed with them."""
    key = torch.device(device).index or 0
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _no_gc(object):
    """Collect garbage now and keep the cyclic collector off inside the block.  An Engine is full of reference cycles (its op lists are
    closures over itself), so an engine that went out of scope dies whenever the collector happens to run -- and if that is in the
    middle of another engine's stream capture, destroying the old engine's hipGraph / events there aborts the process."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class Engine(object):
    def __init__(self, config, mode, batch=None, seed=1234, device=None, randomize_bn=False, grad_bucket_bytes=1 << 40):
        assert mode in ("training", "inference")
        # gradient buckets: one batched reduction / finalisation per bucket.  One GPU: ONE bucket (round 5: the 128 MiB default cut ResNet-50's
        # 128.0 MiB of gradients into a big bucket and a two-layer tail: three more launches, 25 us; 32 MiB: 13 launches and 0.52 ms, 8 MiB: 41 and
        # 0.68 ms); ursonet_amd/dp.py re-plans with 32 MiB buckets, which the all-reduces overlap behind
        self.grad_bucket_bytes = int(os.environ.get("URSO_GRAD_BUCKET_MIB", 0)) << 20 or int(grad_bucket_bytes)     # (env: A/B only)
        self.grad_tail_bytes = 0                       # > 0: the last bucket (stem side) is capped at this size; set by ursonet_amd/dp.py (plan_buckets)
        if not torch.cuda.is_available():
            raise RuntimeError("ursonet_amd.Engine needs an AMD GPU (MI355X / gfx950); there is no CPU fallback")
        self.config, self.mode = config, mode
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.dt_name = compute_dtype_name(config)
        self.dt = _DT[self.dt_name]
        self.tdt = hip.TORCH_DT[self.dt]
        self.B = int(batch if batch is not None else config.BATCH_SIZE)
        self.graph = build_graph(config)
        self.H, self.W = int(config.IMAGE_SHAPE[0]), int(config.IMAGE_SHAPE[1])
        self.layer_trainable = {n: True for n in self.graph.params}
        self.world_size = 1
        self._graphs = None
        self._alloc_params(seed, randomize_bn)
        self._build_plan()

    # ------------------------------------------------------------------ parameters
    def _alloc_params(self, seed, randomize_bn):
        g = self.graph
        off, soff = 0, 0
        self.slices = OrderedDict()        # (layer, weight) -> (offset, numel, shape) in the flat trainable buffer
        self.stat_slices = OrderedDict()   # (layer, weight) -> ... in the BN statistics buffer
        for ln, ws in g.params.items():
            for wn, shape in ws.items():
                n = int(np.prod(shape))
                if wn in ("moving_mean", "moving_variance"):
                    self.stat_slices[(ln, wn)] = (soff, n, shape); soff += _round_up(n, 4)
                else:
                    self.slices[(ln, wn)] = (off, n, shape); off += _round_up(n, 4)
        self.n_flat = off
        dev = self.device
        self.flat_w = torch.zeros(max(off, 4), dtype=torch.float32, device=dev)
        self.flat_stats = torch.zeros(max(soff, 4), dtype=torch.float32, device=dev)
        if self.mode == "training":
            self.flat_g = torch.zeros_like(self.flat_w)
            self.flat_v = torch.zeros_like(self.flat_w)
        self.set_weights(initial_weights(g, seed, randomize_bn))

    def wview(self, ln, wn, buf=None):
        if (ln, wn) in self.slices:
            o, n, shape = self.slices[(ln, wn)]
            base = self.flat_w if buf is None else buf
        else:
            o, n, shape = self.stat_slices[(ln, wn)]
            base = self.flat_stats
        return base[o:o + n].view(shape)

    def gview(self, ln, wn):
        return self.wview(ln, wn, self.flat_g)

    def set_weights(self, params, strict=True):
        """params: {layer: {weight: array}} in Keras layouts.  Missing layers are skipped unless strict."""
        for ln, ws in self.graph.params.items():
            if ln not in params:
                if strict:
                    raise KeyError("missing layer %r" % ln)
                continue
            for wn, shape in ws.items():
                a = np.asarray(params[ln][wn], dtype=np.float32)
                if tuple(a.shape) != tuple(shape):
                    raise ValueError("layer %s weight %s: shape %s != expected %s" % (ln, wn, a.shape, shape))
                self.wview(ln, wn).copy_(torch.from_numpy(np.ascontiguousarray(a)))

    def get_weights(self):
        out = OrderedDict()
        for ln, ws in self.graph.params.items():
            out[ln] = OrderedDict((wn, self.wview(ln, wn).detach().cpu().numpy().copy()) for wn in ws)
        return out

    def get_grads(self):
        out = OrderedDict()
        for (ln, wn) in self.slices:
            out.setdefault(ln, OrderedDict())[wn] = self.gview(ln, wn).detach().cpu().numpy().copy()
        return out

    def trainable_ranges(self):
        """[(begin, end)] element ranges of the flat parameter / gradient buffers that belong to trainable layers."""
        return [(o, o + n) for (ln, _), (o, n, _) in self.slices.items() if self.layer_trainable.get(ln, False)]

    def set_trainable(self, layer_regex):
        """net.py:1030-1066: layer.trainable = fullmatch(regex, layer.name).  Rebuilds the plan."""
        self.layer_trainable = {n: bool(re.fullmatch(layer_regex, n)) for n in self.graph.params}
        self._graphs = None
        self._build_plan()

    # ------------------------------------------------------------------ plan construction
    def _ptr_or_none(self, ln, wn):
        if ln is None or (ln, wn) not in self.slices and (ln, wn) not in self.stat_slices:
            return None
        return self.wview(ln, wn).reshape(-1)

    def _build_plan(self):
        cfg, g, B, dt, dev = self.config, self.graph, self.B, self.dt, self.device
        training = self.mode == "training"
        self.plan_version = getattr(self, "plan_version", 0) + 1       # ursonet_amd/dp.py re-derives its graph cuts when this moves
        # split counts, partial workspaces and grouped / paired weight-gradient launches are sized here from these kernel-policy options; the
        # library re-derives some of them at launch time, so a step under other values would reduce a different number of partials than it wrote
        self._plan_options = self._planning_options()
        # TRAIN_BN = None ("Train BN layers", net.py:60-76): batch statistics while training -- the BN is not folded, the conv
        # writes its raw output and separate stats / apply / backward kernels run (csrc/bn_train.hip).  Inference and
        # TRAIN_BN = False use the moving statistics folded into the filters.
        self.train_bn = training and cfg.TRAIN_BN is None
        VE = 16 // (4 if dt == hip.F32 else 2)
        self.acts = {}
        self.prep_ops, self.fwd_ops, self.loss_ops, self.bwd_ops, self.opt_ops = [], [], [], [], []
        self.wino_ws = None
        self.n_entry_dgrad2 = 0
        self._cleared_once = []    # gradient buffers cleared at plan time only (scattered data gradients): (layer, buffer, H, W, C, sy, sx)
        self.loss_pre_ops = []     # DP_EXACT_REL_LOSS: the part of the loss that precedes the cross-rank sum of the two norms
        self.labels = {"prep": [], "fwd": [], "loss": [], "bwd": [], "opt": []}
        self.convs = OrderedDict()

        def act(spec, numel=None):
            if spec.id not in self.acts:
                self.acts[spec.id] = _Act(self, spec, numel if numel is not None else B * spec.h * spec.w * spec.c, self.tdt)
            return self.acts[spec.id]

        # ---- inputs
        # two input forms: molded float32 frames (the reference's generator format) or raw uint8 frames whose mean subtraction
        # (mold_image, net.py:1337-1348) happens in the molding kernel -- 4x fewer bytes over PCIe (ursonet_amd/feeder.py)
        self.in_images = torch.zeros(B, self.H, self.W, 3, dtype=torch.float32, device=dev)
        if not hasattr(self, "input_u8"):
            self.input_u8 = False
            self.in_images_u8 = None
        mp = np.asarray(cfg.MEAN_PIXEL, dtype=np.float32)
        self.mean3 = torch.tensor(mp if mp.size == 3 else np.full(3, float(mp.mean()), dtype=np.float32), device=dev)
        img_spec = g.tensors[0]
        x0 = act(img_spec, B * self.H * self.W * 4)               # channels padded 3 -> 4, viewed as pixel pairs
        self.fwd_ops.append(lambda: hip.mold_images(B, self.H, self.W, self.in_images_u8 if self.input_u8 else self.in_images,
                                                    self.mean3 if self.input_u8 else None, dt, x0.data))
        self.labels["fwd"].append("mold")

        max_ws = 0
        max_fin_ws = 0
        max_igemm_ws = 0
        max_bn_ws = 0
        descs = []                                 # hip.ParamDesc of every non-stem weight layer
        self._fused_pools = {}                     # id(pool node) -> the stem conv whose kernel also pools
        for node in g.nodes:
            if node.op == "pool":
                if id(node) in self._fused_pools:      # computed by the stem's own kernel (urso_stem_conv_pool): no launch, no conv1 output
                    continue
                src, dst = act(node.src), act(node.dst)
                am = torch.empty(dst.numel, dtype=torch.uint8, device=dev)
                h, w, c = node.src.h, node.src.w, node.src.c
                self.fwd_ops.append(lambda s=src, d=dst, am=am, h=h, w=w, c=c: hip.maxpool_fwd(B, h, w, c, dt, s.data, d.data, am))
                self.labels["fwd"].append("maxpool")
                node._am = am
                continue
            c = _Conv()
            c.node = node
            c.name, c.bn = node.name, node.bn
            c.N = node.cout
            c.npad = _round_up(node.cout, 8)
            c.src = act(node.src) if not node.stem else x0
            out_elt = torch.float32 if node.out_f32 else self.tdt
            if node.dst.id not in self.acts:
                a = _Act(self, node.dst, B * node.dst.h * node.dst.w * c.npad, out_elt)
                self.acts[node.dst.id] = a
            c.dst = self.acts[node.dst.id]
            c.res = act(node.residual) if node.residual is not None else None
            c.xin = c.src.data                         # the tensor the forward pass and the weight gradient read (c.gf describes it)
            if node.stem:
                c.gf = hip.geom(B, self.H, self.W // 2, 8, node.dst.h, node.dst.w, c.npad, 7, 4, 2, 1, 3, 2)
                c.K_raw = 7 * 4 * 8
                c.gd = None
            elif node.dense:
                c.gf = hip.geom(B, 1, 1, node.cin, 1, 1, c.npad, 1, 1)
                c.gd = hip.geom(B, 1, 1, c.npad, 1, 1, node.cin, 1, 1)
                c.K_raw = node.cin
            else:
                s, (pt, pl) = node.stride, node.pad
                c.gf = hip.geom(B, node.src.h, node.src.w, node.cin, node.dst.h, node.dst.w, c.npad, node.kh, node.kw, s, s, pt, pl)
                if node.kh == 1 and node.kw == 1 and s > 1:
                    # compact data gradient: GEMM over the forward-output pixels, scattered to every s-th pixel
                    # of the (pre-zeroed) input-gradient buffer instead of a 4x wasteful gather over all pixels
                    c.gd = hip.geom(B, node.dst.h, node.dst.w, c.npad, node.dst.h, node.dst.w, node.cin, 1, 1,
                                    FH=node.src.h, FW=node.src.w, OSH=s, OSW=s)
                    c.gd_scatter = True
                    c.gd_scatter_stride = s
                else:
                    c.gd = hip.geom(B, node.dst.h, node.dst.w, c.npad, node.src.h, node.src.w, node.cin, node.kh, node.kw,
                                    1, 1, node.kh - 1 - pt, node.kw - 1 - pl, s, s)
                c.K_raw = node.kh * node.kw * node.cin
                if (node.kh == 1 and node.kw == 1 and s == 2 and dt != hip.F32 and node.src.h % 2 == 0 and node.src.w % 2 == 0 and
                        (node.cin * 2) % 16 == 0 and os.environ.get("URSO_COMPACT_INPUT", "1") != "0"):
                    # a stage's stride-2 entry layers (branch2a and the projection shortcut, net.py:121-126) read every second pixel of every
                    # second row of the previous stage's output: the strided source mapping costs the DMA kernel 20-35 us per layer against
                    # the same GEMM on a dense tensor (and keeps the layer off the register-filter kernels), forward and in the weight
                    # gradient.  The sampled pixels are gathered ONCE per step (urso_rows_subsample2) and both layers -- forward and weight
                    # gradient -- run as plain pointwise layers on that compact tensor
                    X = c.src
                    if getattr(X, "data_compact", None) is None:
                        X.data_compact = torch.empty(X.numel // 4, dtype=self.tdt, device=dev)
                        # ... by the layer that produces X where that is a stage-closing c -> 4c layer on the register-filter kernel
                        # (urso_conv_pointwise_sampled: a second store from the LDS tile that holds the output rows), else by a gather pass
                        P = [cc_ for cc_ in self.convs.values() if cc_.dst is X]
                        Pc = P[0] if len(P) == 1 else None
                        if (Pc is not None and not Pc.batch_bn and hasattr(Pc, "fwd_index") and hip.get_option("pair") and
                                hip.conv_pointwise_sampled_ok(Pc.gf, dt, Pc.fwd_flags | hip.EPI_EMIT_BITS, Pc.res is not None)):
                            self.fwd_ops[Pc.fwd_index] = (lambda Pc=Pc, X=X: hip.conv_pointwise_sampled(
                                Pc.gf, dt, Pc.fwd_flags | (hip.EPI_EMIT_BITS if Pc.dst.bits is not None else 0), Pc.xin, Pc.wf, Pc.biasf,
                                Pc.res.data if Pc.res is not None else None, Pc.dst.data, Pc.dst.bits, X.data_compact))
                            self.labels["fwd"][Pc.fwd_index] = "fwd:%s+sampled" % Pc.name
                        else:
                            self.fwd_ops.append(lambda X=X, hh=node.src.h, ww=node.src.w, rb=node.cin * 2:
                                                hip.rows_subsample2(B, hh, ww, rb, X.data, X.data_compact))
                            self.labels["fwd"].append("subsample:T%d" % node.src.id)
                    c.gf = hip.geom(B, node.dst.h, node.dst.w, node.cin, node.dst.h, node.dst.w, c.npad, 1, 1)
                    c.xin = X.data_compact
            if node.cin % VE and not node.stem:
                raise ValueError("layer %s: %d input channels is not a multiple of %d" % (node.name, node.cin, VE))
            kelems = c.npad * c.K_raw
            c.wf = torch.empty(kelems, dtype=self.tdt, device=dev)
            need_dgrad = training and not node.stem
            c.wd = torch.empty(kelems, dtype=self.tdt, device=dev) if need_dgrad else None
            c.biasf = torch.empty(c.npad, dtype=torch.float32, device=dev)
            c.scale = torch.empty(c.npad, dtype=torch.float32, device=dev)
            c.w = self.wview(node.name, "kernel").reshape(-1)
            c.b = self._ptr_or_none(node.name, "bias") if node.bias else None
            c.gamma = self._ptr_or_none(node.bn, "gamma") if node.bn else None
            c.beta = self._ptr_or_none(node.bn, "beta") if node.bn else None
            c.mean = self._ptr_or_none(node.bn, "moving_mean") if node.bn else None
            c.var = self._ptr_or_none(node.bn, "moving_variance") if node.bn else None
            c.batch_bn = bool(self.train_bn and node.bn)
            if c.batch_bn:
                # unfolded conv: the BN tensors move to the bn_* fields, the conv itself is a plain conv + bias
                assert c.npad == c.N and not node.out_f32
                c.bn_gamma, c.bn_beta, c.bn_mmean, c.bn_mvar = c.gamma, c.beta, c.mean, c.var
                c.gamma = c.beta = c.mean = c.var = None
                c.Mpix = B * node.dst.h * node.dst.w
                c.z = torch.empty(c.Mpix * c.N, dtype=self.tdt, device=dev)
                c.dz = torch.empty(c.Mpix * c.N, dtype=self.tdt, device=dev)
                c.bmean, c.bvar = torch.empty(c.N, dtype=torch.float32, device=dev), torch.empty(c.N, dtype=torch.float32, device=dev)
                c.dbeta, c.dgamma = torch.empty(c.N, dtype=torch.float32, device=dev), torch.empty(c.N, dtype=torch.float32, device=dev)
                max_bn_ws = max(max_bn_ws, hip.bn_ws_bytes(c.Mpix, c.N))
            self.convs[node.name] = c
            # -- weight prep (per step in training; once in inference): the stem has its own packing kernel, every other
            #    layer gets a descriptor and ONE batched launch covers them all (urso_param_batch_run)
            if node.stem:
                self.prep_ops.append(lambda c=c: hip.stem_weight_pack(c.N, dt, c.w, c.b, c.gamma, c.beta, c.mean, c.var, BN_EPS,
                                                                      c.wf, c.biasf, c.scale))
                self.labels["prep"].append("prep:" + node.name)
            else:
                d = hip.ParamDesc()
                c.splits = hip.conv_wgrad_splits(c.gf, dt) if training else 1
                hip.param_desc_init(d, node.kh if not node.dense else 1, node.kw if not node.dense else 1, node.cin, c.N, c.npad,
                                    max(c.splits, 1), BN_EPS, float(cfg.WEIGHT_DECAY))
                for f, t in (("w", c.w), ("b", c.b), ("gamma", c.gamma), ("beta", c.beta), ("mean", c.mean), ("var", c.var),
                             ("wf", c.wf), ("wd", c.wd), ("biasf", c.biasf), ("scale", c.scale)):
                    setattr(d, f, hip.ptr(t))
                c.desc_id = len(descs)
                c.desc = d
                descs.append(d)
            # -- forward
            flags = (hip.EPI_RELU if node.relu else 0) | (hip.EPI_OUT_F32 if node.out_f32 else 0)
            # split-K workspace (tiny-grid / deep-K layers: bottleneck_layer, Dense heads); 0 = not split
            c.ws_f = hip.conv_igemm_ws_bytes(c.gf, dt)
            c.ws_d = hip.conv_igemm_ws_bytes(c.gd, dt) if (c.gd is not None and training) else 0
            max_igemm_ws = max(max_igemm_ws, c.ws_f, c.ws_d)
            c.fwd_flags = flags
            # 3x3 layers taken by the halo-tile kernel get its stream-K hand-over workspace (dedicated: its first 4 KiB are flags that
            # must stay zero between launches, include/ursonet_hip.h)
            c.halo_f = (not c.batch_bn) and dt != hip.F32 and hip.conv_igemm_halo_ok(c.gf, dt, flags, c.res is not None)
            c.halo_d = bool(c.gd is not None and training and dt != hip.F32 and hip.conv_igemm_halo_ok(c.gd, dt, 0, False))
            if (c.halo_f or c.halo_d) and getattr(self, "halo_ws", None) is None:
                self.halo_ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, dtype=torch.float32, device=dev)
            if c.batch_bn:
                self.fwd_ops.append(lambda c=c: hip.conv_igemm_ex(c.gf, dt, 0, c.xin, c.wf, c.biasf, None, None, c.z, None,
                                                                  self.igemm_ws if c.ws_f else None))
                self.labels["fwd"].append("fwd:" + node.name)
                self.fwd_ops.append(lambda c=c: hip.bn_batch_stats(c.Mpix, c.N, dt, c.z, self.bn_ws, c.bmean, c.bvar, c.bn_mmean, c.bn_mvar,
                                                                   BN_MOMENTUM, BN_EPS))
                self.labels["fwd"].append("bn_stats:" + node.name)
                self.fwd_ops.append(lambda c=c, n=node: hip.bn_apply(c.Mpix, c.N, dt, c.z, c.bmean, c.bvar, c.bn_gamma, c.bn_beta, BN_EPS,
                                                                     c.res.data if c.res is not None else None, n.relu, c.dst.data))
                self.labels["fwd"].append("bn_apply:" + node.name)
            elif (os.environ.get("URSO_WINOGRAD", "0") == "1" and dt != hip.F32 and not node.stem and not node.dense and node.kh == 3 and node.kw == 3 and
                  node.stride == 1 and tuple(node.pad) == (1, 1) and c.res is None and not (flags & ~hip.EPI_RELU) and c.npad == c.N and
                  hip.conv_winograd_ws_bytes(c.gf, dt) > 0):
                # opt-in: the 3x3 / stride-1 layers through the Winograd F(2x2, 3x3) evaluation (conv_winograd.hip; north_star names it; slower
                # than the direct kernels on MI355X).  One shared workspace, sized for the largest layer.
                need_ws = hip.conv_winograd_ws_bytes(c.gf, dt) // 4 + 64
                if getattr(self, "wino_ws", None) is None or self.wino_ws.numel() < need_ws:
                    self.wino_ws = torch.empty(need_ws, dtype=torch.float32, device=dev)
                c.fwd_index = len(self.fwd_ops)
                self.fwd_ops.append(lambda c=c, f=flags: hip.conv_winograd_fwd(c.gf, dt, f, c.xin, c.wf, c.biasf, c.dst.data, self.wino_ws))
                self.labels["fwd"].append("fwd:" + node.name)
                c.winograd = True
            elif self._stem_pool_node(g, node, c, dt) is not None:
                # conv1 + ReLU + the max-pool behind it in one kernel: conv1's output (the largest tensor of the net, read by the pool
                # alone) is neither written nor read; its _Act keeps no storage
                pool = self._stem_pool_node(g, node, c, dt)
                self._fused_pools[id(pool)] = c
                pdst = act(pool.dst)
                pool._am = torch.empty(pdst.numel, dtype=torch.uint8, device=dev)
                c.dst.data = torch.empty(0, dtype=self.tdt, device=dev)      # (frees the 335 MB; any later read raises: _Act.data)
                c.dst.fused_pool = True
                c.fwd_index = len(self.fwd_ops)
                self.fwd_ops.append(lambda c=c, d=pdst, am=pool._am: hip.stem_conv_pool(c.gf, dt, c.xin, c.wf, c.biasf, d.data, am))
                self.labels["fwd"].append("fwd:%s+maxpool" % node.name)
            else:
                c.fwd_index = len(self.fwd_ops)
                self.fwd_ops.append(lambda c=c, f=flags: hip.conv_igemm_ex(
                    c.gf, dt, f | (hip.EPI_EMIT_BITS if c.dst.bits is not None else 0), c.xin, c.wf, c.biasf,
                    c.res.data if c.res is not None else None, None, c.dst.data, c.dst.bits,
                    self.halo_ws if c.halo_f else (self.igemm_ws if c.ws_f else None)))
                self.labels["fwd"].append("fwd:" + node.name)
            if training and node.stem:
                max_ws = max(max_ws, hip.conv_wgrad_ws_bytes(c.gf, dt))
                max_fin_ws = max(max_fin_ws, hip.param_grad_finalize_ws_bytes(147, c.N))
        self._fuse_pointwise_pairs()
        self._fuse_entry_shortcuts()
        self._fuse_dense_heads()
        self.igemm_ws = torch.empty(max_igemm_ws // 4 + 16, dtype=torch.float32, device=dev)
        self.bn_ws = torch.empty(max_bn_ws // 8 + 32, dtype=torch.float64, device=dev) if max_bn_ws else None
        self._descs = descs
        if descs:
            self.prep_ops.append(lambda: self.pbatch.run(hip.PB_PREP, "all", dt))
            self.labels["prep"].append("prep:batched[%d layers]" % len(descs))
        if not training:
            self._upload_param_table()
        self.out_loc = self.acts[g.outputs["loc"].id]
        self.out_ori = self.acts[g.outputs["ori"].id] if "ori" in g.outputs else None      # keypoint mode has no orientation head
        self._build_heads_io()
        if not training:
            for X in list(self.acts.values()):
                self._sample_block_output(X)
            return
        # ---------------------------------------------------------------- losses + backward
        self.ws = torch.empty(max_ws // 4 + 64, dtype=torch.float32, device=dev)
        self.fin_ws = torch.empty(max_fin_ws // 4 + 64, dtype=torch.float32, device=dev)
        self._build_losses()
        self._plan_relu_bitmasks()
        # gradient buckets (contiguous slices of the flat gradient buffer, in the order the backward pass completes them):
        # the split reduction + finalisation of all layers of a bucket is three batched launches issued once the last of
        # their weight-gradient partials is enqueued; ursonet_amd/dp.py starts the bucket's all-reduce right after.
        from .dp import plan_buckets
        pending_groups = []
        # Grouped pointwise weight gradients (urso_wgrad_group_run): a layer launched alone is cut into ~2 blocks per CU whatever its
        # size, so it writes -- and the bucket's split reduction re-reads -- CUs x 128 KiB of fp32 partials (32 MB a layer, as much as
        # its operands in stages 4-5), in a burst at the end of the launch that nothing overlaps.  Up to URSO_WGRAD_GROUP consecutive
        # pointwise layers of a bucket share one launch instead: each gets 1/n of the splits, its blocks run n times longer.
        wg_max = int(os.environ.get("URSO_WGRAD_GROUP", "8")) if dt != hip.F32 else 0
        wg_fill = float(os.environ.get("URSO_WGRAD_GROUP_FILL", "0.7"))
        pend_wg = []
        self.n_wgrad_groups = 0

        def emit_wgrad(c, name, xw, G, gf_w):
            self.bwd_ops.append((name, lambda: hip.conv_wgrad_partial(gf_w, dt, xw, G, c.wg_ws)))
            self.labels["bwd"].append("wgrad:" + name)

        pend_hw = []
        wg_pair3 = wg_max > 1 and int(os.environ.get("URSO_WGRAD_PAIR3X3", "1")) != 0

        def flush_hw_pair():
            items = list(pend_hw)
            del pend_hw[:]
            sp = hip.conv_wgrad_pair_splits(items[0][4], items[1][4], dt) if len(items) == 2 else None
            if sp is None:
                for t in items:
                    emit_wgrad(*t)
                return
            for (c, _, _, _, _), s_ in zip(items, sp):
                assert s_ <= c.splits
                c.splits = c.desc.splits = s_
                c.wg_npart = s_ * (c.K_raw * c.npad + hip.WGRAD_PART_PAD)
                c.desc.part, c.desc.colpart = c.wg_ws.data_ptr(), c.wg_ws.data_ptr() + 4 * c.wg_npart
            (c0, n0, x0, G0, g0), (c1, n1, x1, G1, g1) = items
            self.bwd_ops.append(((n0, n1), lambda: hip.conv_wgrad_partial2(g0, g1, dt, x0, G0, c0.wg_ws, x1, G1, c1.wg_ws)))
            self.labels["bwd"].append("wgrad:%s+%s" % (n0, n1))
            self.n_wgrad_groups += 1

        def flush_wgrads():
            items = list(pend_wg)
            del pend_wg[:]
            while items:
                take = items
                grp = None
                while len(take) > 1:
                    grp = hip.WgradGroup([t[4] for t in take], dt)
                    if grp.nblocks:
                        break
                    take, grp = take[:len(take) - 1], None           # more tiles than resident blocks: a smaller group
                items = items[len(take):]
                if grp is None:
                    emit_wgrad(*take[0])
                    continue
                for (c, _, _, _, _), s in zip(take, grp.splits):
                    need_f = s * (c.K_raw * c.npad + hip.WGRAD_PART_PAD) + s * c.npad + 64
                    if c.wg_ws.numel() < need_f:           # (the workspace was sized for the layer alone)
                        c.wg_ws = torch.empty(need_f, dtype=torch.float32, device=dev)
                    if s > 1 and c.dw_raw is None:
                        c.dw_raw = torch.empty(c.K_raw * c.npad, dtype=torch.float32, device=dev)
                        c.colsum = torch.empty(c.npad, dtype=torch.float32, device=dev)
                        c.desc.dw_raw, c.desc.colsum = hip.ptr(c.dw_raw), hip.ptr(c.colsum)
                    c.splits = c.desc.splits = s
                    c.wg_npart = s * (c.K_raw * c.npad + hip.WGRAD_PART_PAD)
                    c.desc.part, c.desc.colpart = c.wg_ws.data_ptr(), c.wg_ws.data_ptr() + 4 * c.wg_npart
                grp.bind([t[2] for t in take], [t[3] for t in take], [t[0].wg_ws for t in take], dev)
                names = tuple(t[1] for t in take)
                self.bwd_ops.append((names, lambda grp=grp: grp.run()))
                self.labels["bwd"].append("wgrad:" + "+".join(names))
                self.n_wgrad_groups += 1
        ext = {}
        for (ln, wn), (o, n, _) in self.slices.items():
            s0, e0 = ext.get(ln, (o, o))
            ext[ln] = (min(s0, o), max(e0, o + _round_up(n, 4)))
        self.buckets = plan_buckets(sorted(((ln, s0, e0) for ln, (s0, e0) in ext.items()), key=lambda t: t[1]), self.grad_bucket_bytes,
                                    tail_bytes=self.grad_tail_bytes)
        # Global gradient norm without a pass of its own: with ONE gradient bucket and nothing between finalisation and optimizer (no all-reduce,
        # no batch-statistics BN writing gamma / beta gradients elsewhere) every gradient slice is written by a finalisation launch, and each of
        # their blocks leaves the sum of squares of what it stored (urso_param_batch_run_sq): urso_sqnorm_final adds the slots (-24 us of reading
        # the 134 MB buffer back).  ursonet_amd/dp.py sets no_fused_sqnorm: there the norm is that of the all-reduced gradient.
        self.fused_sqnorm = (os.environ.get("URSO_FUSE_SQNORM", "1") != "0" and not getattr(self, "no_fused_sqnorm", False) and
                             len(self.buckets) == 1 and not any(c_.batch_bn for c_ in self.convs.values()))
        sq_slots = [0]
        bucket_of = {ln: k for k, (_, _, names) in enumerate(self.buckets) for ln in names}
        groups = OrderedDict()                      # bucket index -> [conv names], backward order
        bwd_order = self._backward_order()
        for node in bwd_order:
            if node.op == "pool" or node.stem:
                continue
            folded_bn = node.bn and not self.convs[node.name].batch_bn      # batch-statistics BN finalises its own gamma/beta
            if self.layer_trainable[node.name] or (folded_bn and self.layer_trainable[node.bn]):
                groups.setdefault(bucket_of[node.name], []).append(node.name)
        last_of_group = {names[-1]: k for k, names in groups.items()}
        self._bucket_groups = groups                # (param_pass_bytes: what the batched parameter-sized launches of a bucket move)
        # which activation gradients are needed at all: a tensor's gradient only if a trainable parameter lies at or upstream of its
        # producer.  With layers='heads' / '4+' / '5+' (net.py:1086-1095) the data-gradient chain stops at the earliest trainable
        # layer instead of running through the frozen backbone (TF prunes those gradients too)
        need = {g.tensors[0].id: False}
        for node in g.nodes:
            if node.op == "pool":
                need[node.dst.id] = need.get(node.src.id, False)
            else:
                own = self.layer_trainable[node.name] or bool(node.bn and self.layer_trainable[node.bn])
                need[node.dst.id] = (own or (not node.stem and need.get(node.src.id, False)) or
                                     (node.residual is not None and need.get(node.residual.id, False)))
        self.grad_needed = need
        self._plan_compact_gradients(need)
        # Data gradients of the Dense heads (urso_dense_multi): the layers of one depth behind the bottleneck features share a launch, and the
        # two gradients into a tensor both branches read (loc_dense_0 / ori_dense_0 -> the bottleneck features) are the two reduction
        # segments of ONE layer instead of a launch that writes and a launch that accumulates in place
        dense_lv = self._dense_levels() if self.dense_multi else {}
        pend_dd = []

        def flush_dense_dgrads():
            layers = list(pend_dd)
            del pend_dd[:]
            for i in range(0, len(layers), hip.DENSE_MULTI_MAX):
                part = layers[i:i + hip.DENSE_MULTI_MAX]
                m = hip.DenseMulti(part, dt)
                self.bwd_ops.append((None, lambda m=m: m.run()))
                self.labels["bwd"].append("dgrad_heads:" + "+".join(L["name"] for L in part))
        # ... and their weight gradients (urso_dense_wgrad_multi): leaves of the backward pass, so all of them wait for one launch behind the
        # last Dense data gradient (or the end of their gradient bucket)
        pend_dwg = []

        def flush_dense_wgrads():
            layers = list(pend_dwg)
            del pend_dwg[:]
            for i in range(0, len(layers), hip.DENSE_MULTI_MAX):
                part = layers[i:i + hip.DENSE_MULTI_MAX]
                m = hip.DenseWgradMulti(part, dt)
                self.bwd_ops.append((tuple(L["name"] for L in part), lambda m=m: m.run()))
                self.labels["bwd"].append("wgrad_heads:" + "+".join(L["name"] for L in part))
        for node in bwd_order:
            lvl_n = dense_lv.get(node.name) if node.op != "pool" else None
            if pend_dd and lvl_n != pend_dd[0]["level"]:
                flush_dense_dgrads()
            if pend_dwg and lvl_n is None:
                flush_dense_wgrads()
            if not need[node.dst.id]:
                continue                                   # nothing trainable at or below this node
            if node.op == "pool":
                if not need[node.src.id]:
                    continue
                src, dst = self.acts[node.src.id], self.acts[node.dst.id]
                h, w, cc = node.src.h, node.src.w, node.src.c
                assert dst.grad_written
                # the pool behind the stem: its backward pass would write the gradient of conv1's output (335 MB at cfg2) for conv1's weight
                # gradient alone (the image needs no data gradient) -- that kernel rebuilds the tiles it needs from dst.grad and the
                # arg-max bytes instead (urso_stem_wgrad_pooled): no launch here, no tensor
                prod = [cc_ for cc_ in self.convs.values() if cc_.dst is src]
                if (len(prod) == 1 and prod[0].node.stem and not prod[0].batch_bn and dt != hip.F32 and hip.get_option("stem") and cc == 64 and
                        h % 2 == 0 and w % 2 == 0 and (self.layer_trainable[prod[0].node.name] or bool(prod[0].node.bn and self.layer_trainable[prod[0].node.bn]))):
                    prod[0].pooled_grad = (dst, node._am)
                    src.grad_written = True
                    continue
                gsrc = src.grad_buf()
                self.bwd_ops.append((None, lambda d=dst, gs=gsrc, am=node._am, h=h, w=w, cc=cc:
                                     hip.maxpool_bwd(B, h, w, cc, dt, d.data, d.grad, am, 1, gs)))
                self.labels["bwd"].append("maxpool_bwd")
                src.grad_written = True
                continue
            c = self.convs[node.name]
            assert c.dst.grad_written or c.dst.grad is not None, "no gradient reaches %s" % node.name
            G = c.dst.grad                             # compact tensors: [B, H/2, W/2, N] (see _plan_compact_gradients)
            gf_w = getattr(c, "gf_compact", None) or c.gf
            tr = self.layer_trainable[node.name]
            bn_tr = self.layer_trainable[node.bn] if node.bn else False
            Gsum = G                                   # gradient w.r.t. (BN output + residual): what a residual branch receives
            if c.batch_bn:
                ggam, gbet = self.gview(node.bn, "gamma").reshape(-1), self.gview(node.bn, "beta").reshape(-1)
                self.bwd_ops.append((node.name, lambda c=c, G=G, bn_tr=bn_tr, ggam=ggam, gbet=gbet:
                                     hip.bn_backward(c.Mpix, c.N, dt, G, c.z, c.bmean, c.bvar, c.bn_gamma, BN_EPS, self.bn_ws, c.dbeta, c.dgamma,
                                                     bn_tr, gbet, ggam, c.dz)))
                self.labels["bwd"].append("bn_bwd:" + node.name)
                G = c.dz                               # the conv itself sees the gradient w.r.t. its raw output
                bn_tr = False                          # gamma/beta gradients are done; the finalisation treats the layer as a plain conv
            # -- weight gradient + finalisation (skipped for fully frozen layers; their grads stay zero)
            if (tr or bn_tr) and node.stem:
                c.dw_raw = torch.empty(c.K_raw * c.npad, dtype=torch.float32, device=dev)
                c.colsum = torch.empty(c.npad, dtype=torch.float32, device=dev)
                if getattr(c, "pooled_grad", None) is not None:
                    self.bwd_ops.append((node.name, lambda c=c, pg=c.pooled_grad: hip.stem_wgrad_pooled(c.gf, dt, c.src.data, pg[0].grad, pg[1], self.ws,
                                                                                                        c.dw_raw, c.colsum)))
                    self.labels["bwd"].append("wgrad:%s+maxpool_bwd" % node.name)
                else:
                    self.bwd_ops.append((node.name, lambda c=c, G=G: hip.conv_wgrad(c.gf, dt, c.src.data, G, self.ws, c.dw_raw, c.colsum)))
                    self.labels["bwd"].append("wgrad:" + node.name)
                c.dw_unp = torch.empty(147 * c.N, dtype=torch.float32, device=dev)
                self.bwd_ops.append((node.name, lambda c=c: hip.stem_wgrad_unpack(c.N, c.dw_raw, c.dw_unp)))
                self.labels["bwd"].append("unpack:" + node.name)
                gw = self.gview(node.name, "kernel").reshape(-1)
                gb = self.gview(node.name, "bias").reshape(-1) if node.bias else None
                gg = self.gview(node.bn, "gamma").reshape(-1) if (node.bn and not c.batch_bn) else None
                gbe = self.gview(node.bn, "beta").reshape(-1) if (node.bn and not c.batch_bn) else None
                if self.fused_sqnorm:
                    s0, s1 = sq_slots[0], sq_slots[0] + hip.param_grad_finalize_sq_slots(147, c.N)
                    sq_slots[0] = s1
                    self.bwd_ops.append((node.name, lambda c=c, gw=gw, gb=gb, gg=gg, gbe=gbe, tr=tr, bn_tr=bn_tr, s0=s0, s1=s1:
                                         hip.param_grad_finalize_sq(147, c.N, c.N, c.dw_unp, c.colsum, c.w, c.b, c.gamma, c.mean, c.var, BN_EPS,
                                                                    float(cfg.WEIGHT_DECAY), tr, bn_tr, gw, gb, gg, gbe, self.fin_ws, self.sqpart[s0:s1])))
                else:
                    self.bwd_ops.append((node.name, lambda c=c, gw=gw, gb=gb, gg=gg, gbe=gbe, tr=tr, bn_tr=bn_tr:
                                         hip.param_grad_finalize(147, c.N, c.N, c.dw_unp, c.colsum, c.w, c.b, c.gamma, c.mean, c.var, BN_EPS,
                                                                 float(cfg.WEIGHT_DECAY), tr, bn_tr, gw, gb, gg, gbe, self.fin_ws)))
                self.labels["bwd"].append("finalize:" + node.name)
            elif tr or bn_tr:
                d = c.desc
                # a 64 -> 256 pointwise layer outside the pairs (the stage-2 projection shortcut) whose data gradient is the first
                # contribution to its input's gradient: both gradients in one pass over dz (urso_conv_dgrad_wgrad_pw)
                Xs = c.src
                Ms = B * node.dst.h * node.dst.w
                c.solo = (hip.get_option("pair") == 1 and dt != hip.F32 and not node.dense and node.kh == 1 and node.kw == 1 and node.stride == 1 and
                          node.cin == 64 and c.npad == 256 and c.N == 256 and not c.batch_bn and need[Xs.spec.id] and
                          not getattr(c, "wgrad_by_pair", False) and not getattr(c, "dgrad_done_by_pair", False) and
                          getattr(c, "gf_compact", None) is None and c.dst.compact is None and Xs.compact is None and not getattr(c, "gd_scatter", False) and
                          not Xs.grad_written and Xs.pending is None and self.pair_first.get(node.name) is None and node.name not in last_of_group and
                          hip.conv_pair_wgrad_splits(Ms, dt) > 1)
                if c.solo:
                    c.splits = c.desc.splits = hip.conv_pair_wgrad_splits(Ms, dt)
                n_part = c.splits * (c.K_raw * c.npad + hip.WGRAD_PART_PAD)
                by_pair = getattr(c, "wgrad_by_pair", False) or c.solo   # partials written by a fused launch (the pair of the layer above / below)
                c.wg_npart = n_part
                c.wg_ws = torch.empty((n_part + c.splits * c.npad if by_pair else hip.conv_wgrad_ws_bytes(gf_w, dt) // 4) + 64,
                                      dtype=torch.float32, device=dev)
                c.dw_raw = torch.empty(c.K_raw * c.npad, dtype=torch.float32, device=dev) if c.splits > 1 else None
                c.colsum = torch.empty(c.npad, dtype=torch.float32, device=dev) if c.splits > 1 else None
                c.dotpart = torch.empty(d.ks * c.N + 16, dtype=torch.float32, device=dev)
                d.part, d.colpart = c.wg_ws.data_ptr(), c.wg_ws.data_ptr() + 4 * n_part
                d.dw_raw, d.colsum, d.dotpart = hip.ptr(c.dw_raw), hip.ptr(c.colsum), hip.ptr(c.dotpart)
                d.trainable, d.bn_trainable = int(tr), int(bn_tr)
                d.gw = hip.ptr(self.gview(node.name, "kernel").reshape(-1))
                d.gb = hip.ptr(self.gview(node.name, "bias").reshape(-1)) if node.bias else None
                d.ggamma = hip.ptr(self.gview(node.bn, "gamma").reshape(-1)) if (node.bn and not c.batch_bn) else None
                d.gbeta = hip.ptr(self.gview(node.bn, "beta").reshape(-1)) if (node.bn and not c.batch_bn) else None
                if not by_pair:
                    xw = c.xin if gf_w is c.gf else c.src.data
                    if wg_pair3 and gf_w.KH * gf_w.KW * gf_w.C == c.K_raw and gf_w.N == c.npad and hip.conv_wgrad_pair_splits(gf_w, gf_w, dt):
                        # a 3x3 layer of the register-resident kernel: two of them share a launch and the CUs (urso_conv_wgrad_partial2)
                        pend_hw.append((c, node.name, xw, G, gf_w))
                        if len(pend_hw) == 2:
                            flush_hw_pair()
                    elif wg_max > 1 and hip.wgrad_group_fits(gf_w, dt) and gf_w.KH * gf_w.KW * gf_w.C == c.K_raw and gf_w.N == c.npad:
                        # a layer of the general kernel: its weight gradient waits for company (see flush_wgrads); every tensor has a gradient
                        # buffer of its own, so G is still there when the launch comes
                        cand = pend_wg + [(c, node.name, xw, G, gf_w)]
                        wide = lambda g: g.KH * g.KW * g.C >= 256 and g.N >= 256      # (256 x 256 tiles when every layer of the group is this wide)
                        if len(cand) > 1 and (wide(cand[-1][4]) != wide(cand[0][4]) or hip.WgradGroup([t[4] for t in cand], dt).fill < wg_fill):
                            flush_wgrads()                 # another tile shape, or the newcomer's tile count / pixel count does not divide the slots well
                        pend_wg.append(cand[-1])
                        if len(pend_wg) >= wg_max:
                            flush_wgrads()
                    elif lvl_n is not None and c.splits == 1 and gf_w is c.gf and self._dense_multi_ok(c):
                        pend_dwg.append(dict(name=node.name, x=xw, dz=G, part=c.wg_ws, colpart=c.wg_ws[c.wg_npart:], M=B, K=c.K_raw, N=c.npad))
                    else:
                        self.bwd_ops.append((node.name, lambda c=c, G=G, gf_w=gf_w, xw=xw: hip.conv_wgrad_partial(gf_w, dt, xw, G, c.wg_ws)))
                        self.labels["bwd"].append("wgrad:" + node.name)
                if node.name in last_of_group:
                    if pend_dwg and any(L["dz"] is P["dst"] for L in pend_dwg for P in pend_dd):
                        flush_dense_dgrads()               # (a waiting weight gradient reads what a waiting data gradient writes)
                    flush_dense_wgrads()
                    flush_wgrads()                         # the bucket's reduction reads every partial of the bucket
                    flush_hw_pair()
                    k = last_of_group[node.name]
                    pending_groups.append((k, tuple(groups[k]), len(self.bwd_ops)))
                    for ph, nm in ((hip.PB_REDUCE, "reduce"), (hip.PB_FINALIZE_MAT, "finalize_mat"), (hip.PB_FINALIZE_VEC, "finalize_vec")):
                        self.bwd_ops.append((tuple(groups[k]), (ph, k)))          # resolved to launches once the table is on the device
                        self.labels["bwd"].append("%s:bucket%d" % (nm, k))
            # -- residual branch: its gradient IS G (Add); fold it into the next dgrad (post-ReLU tensors) or alias it
            if c.res is not None and need[c.res.spec.id]:
                R = c.res
                if R.grad_written or R.pending is not None:
                    raise AssertionError("unexpected second residual consumer for %s" % node.name)
                if R.spec.relu and c.dst.compact is not None and c.dst.residual_needs_dense:
                    # no fused pair on the residual side: hand over the dense form (zeros written once, by the expansion)
                    # (the dense form is zero off the even grid in every step and nothing else writes it: cleared once, here; a step rewrites
                    # the even pixels only -- urso_rows_scatter2, a quarter of the bytes of the full expansion urso_rows_expand2)
                    c.dst.grad_dense = torch.zeros(c.dst.numel, dtype=self.tdt, device=dev)
                    self.bwd_ops.append((None, lambda X=c.dst: hip.rows_scatter2(B, X.compact[0], X.compact[1], X.spec.c * 2, X.grad, X.grad_dense)))
                    self.labels["bwd"].append("expand:" + node.name)
                    R.pending = c.dst.grad_dense
                elif R.spec.relu:
                    R.pending = Gsum
                    R.pending_hw = c.dst.compact           # (H, W) when Gsum holds only the even rows / columns, else None
                else:
                    assert c.dst.compact is None
                    R.grad, R.grad_written = Gsum, True
            # -- data gradient into the conv input
            if getattr(c, "dgrad_done_by_pair", False):
                continue                                   # written by the fused launch of the layer above (see below)
            if not node.stem and need[c.src.spec.id]:
                X = c.src
                add = X.grad if X.grad_written else X.pending
                if getattr(c, "solo", False):
                    assert add is None
                    dstg = X.grad_buf()
                    self.bwd_ops.append((node.name, lambda c=c, G=G, X=X, dstg=dstg, Ms=B * node.dst.h * node.dst.w:
                                         hip.conv_dgrad_wgrad_pw(Ms, dt, G, c.wd, X.data, X.spec.relu, dstg, c.wg_ws, c.wg_ws[c.wg_npart:],
                                                                 c.K_raw * c.npad + hip.WGRAD_PART_PAD)))
                    self.labels["bwd"].append("dgrad+wgrad:" + node.name)
                    X.grad_written = True
                    continue
                if X.compact is not None:
                    # stride-2 pointwise consumer of a block output whose gradient is kept compact: a plain pointwise GEMM over the
                    # sampled pixels (no scatter, no zero fill of a dense tensor), masked with the sampled rows of X's ReLU bit mask
                    if not X.grad_written and not getattr(X, "fwd_sampled", False):       # (a sampled block output wrote the compact mask in its forward pass)
                        self.bwd_ops.append((None, lambda X=X: hip.rows_subsample2(B, X.compact[0], X.compact[1], X.spec.c // 8, X.bits, X.bits_compact)))
                        self.labels["bwd"].append("bits_subsample")
                    gq = c.gd_compact
                    first = getattr(X, "_compact_first", None)
                    if (first is not None and X.grad_written and os.environ.get("URSO_ENTRY_DGRAD2", "1") != "0" and
                            hip.conv_pointwise2_ok(B, gq.OH, gq.OW, first[1].npad, c.npad, node.cin, dt, hip.EPI_MASK_BITS)):
                        # the SECOND stride-2 consumer of X (a stage's first block: the projection shortcut and branch2a, net.py:121-126, 148-157):
                        # both data gradients in ONE launch with two reduction segments (urso_conv_pointwise2) -- dL/dX is written once and
                        # rounded once instead of written by the first launch, read back and rewritten by this one.  The first launch becomes
                        # a no-op where it stood (its operands -- the block output's gradient, its filter -- outlive it: every tensor has its
                        # own gradient buffer); the sum runs in the same order (shortcut first).  URSO_ENTRY_DGRAD2=0: the two launches (A/B).
                        i0, c0, G0 = first
                        self.bwd_ops[i0] = (None, lambda: None)
                        self.labels["bwd"][i0] = None
                        self.bwd_ops.append((None, lambda c=c, c0=c0, G0=G0, G=G, X=X, gq=gq, N=node.cin:
                                             hip.conv_pointwise2(B, gq.OH, gq.OW, c0.npad, c.npad, N, dt, hip.EPI_MASK_BITS, G0, c0.wd, G, c.wd, None,
                                                                 X.bits_compact, X.grad)))
                        self.labels["bwd"].append("dgrad:%s+%s" % (c0.name, node.name))
                        X._compact_first = None
                        self.n_entry_dgrad2 = getattr(self, "n_entry_dgrad2", 0) + 1
                        continue
                    if not X.grad_written:
                        X._compact_first = (len(self.bwd_ops), c, G)
                    else:
                        X._compact_first = None            # a third writer: no merge
                    self.bwd_ops.append((None, lambda c=c, G=G, X=X, add=(X.grad if X.grad_written else None):
                                         hip.conv_igemm_ex(c.gd_compact, dt, hip.EPI_MASK_BITS, G, c.wd, None, add, X.bits_compact, X.grad, None, None)))
                    self.labels["bwd"].append("dgrad:" + node.name)
                    X.grad_written = True
                    continue
                A = self.pair_first.get(node.name)         # this layer is the second of a fused forward pair (A = the block-closing 2c)
                if (A is not None and X.spec.relu and X.bits is not None and X.pending is not None and not X.grad_written and
                        need[A.src.spec.id] and A.src.spec.relu and A.src.bits is None and not A.src.grad_written and A.src.pending is None):
                    # data gradient of this layer into X (+ residual gradient, ReLU bit mask) and, from the LDS copy of that result,
                    # the data gradient of layer A into ITS input: X.grad crosses HBM once (conv_pair.hip)
                    dstg, dst2 = X.grad_buf(), A.src.grad_buf()
                    a_tr = self.layer_trainable[A.name] or bool(A.node.bn and not A.batch_bn and self.layer_trainable[A.node.bn])
                    wsplits = hip.conv_pair_wgrad_splits(A.Mpix, dt) if (A.node.cin == 64 and hip.get_option("pair") == 1) else 0
                    if a_tr and wsplits > 1 and getattr(A, "gf_compact", None) is None:
                        # ... and, stage 2, the weight gradient of layer A as well: both of its operands (X.grad, A's input) are in
                        # LDS in that launch (conv_pairw.hip); one fp32 partial per block lands in A's split workspace, allocated
                        # when the loop reaches A -- its reduction and finalisation stay where they are
                        A.splits = A.desc.splits = wsplits
                        A.wgrad_by_pair = True
                        # the stage's first block: X's gradient is read by nothing but this pair and the projection shortcut (A's residual
                        # operand, a conv without activation): its data and weight gradient join the launch and dL/dX stays on chip
                        S = [cc for cc in self.convs.values() if A.res is not None and cc.dst is A.res]
                        Sc = S[0] if len(S) == 1 else None
                        sn = Sc.node if Sc is not None else None
                        entry = (Sc is not None and X.pending_hw is None and not sn.stem and not sn.dense and sn.kh == 1 and sn.kw == 1 and
                                 sn.stride == 1 and sn.cin == 64 and Sc.npad == 256 and Sc.N == 256 and not Sc.batch_bn and not sn.relu and
                                 (self.layer_trainable[sn.name] or bool(sn.bn and self.layer_trainable[sn.bn])) and
                                 need[Sc.src.spec.id] and Sc.src.compact is None and not getattr(Sc, "gd_scatter", False) and
                                 getattr(Sc, "gf_compact", None) is None and not Sc.src.grad_written and Sc.src.pending is None and
                                 sn.name not in last_of_group and self.pair_first.get(sn.name) is None)
                        if entry:
                            Sc.splits = Sc.desc.splits = wsplits
                            Sc.wgrad_by_pair = Sc.dgrad_done_by_pair = True
                            dxin = Sc.src.grad_buf()
                            Sc.src.grad_written = True
                            dstg.zero_()                   # dL/dX never reaches memory in this form: the buffer stays what it is (zeros, not whatever the allocator left)
                            dstg._urso_on_chip = True      # (tests/test_layerwise_gpu.py skips what it cannot read; the shortcut's gradient aliases this buffer)
                            self.bwd_ops.append((None, lambda c=c, A=A, Sc=Sc, G=G, add=add, X=X, dst2=dst2, dxin=dxin:
                                                 hip.conv_pair_wgrad_entry(A.Mpix, dt, G, c.wd, add, X.bits, A.wd, A.src.data, dst2, Sc.wd, Sc.src.data, Sc.src.spec.relu, dxin,
                                                                           A.wg_ws, A.wg_ws[A.wg_npart:], Sc.wg_ws, Sc.wg_ws[Sc.wg_npart:],
                                                                           A.K_raw * A.npad + hip.WGRAD_PART_PAD)))
                            self.labels["bwd"].append("dgrad:%s+%s+%s+wgrad:%s+%s" % (node.name, A.name, sn.name, A.name, sn.name))
                        else:
                            self.bwd_ops.append((None, lambda c=c, A=A, G=G, add=add, X=X, dstg=dstg, dst2=dst2, hw=X.pending_hw:
                                                 hip.conv_pair_wgrad(A.Mpix, dt, G, c.wd, add, X.bits, dstg, A.wd, A.src.data, dst2,
                                                                     A.wg_ws, A.wg_ws[A.wg_npart:], A.K_raw * A.npad + hip.WGRAD_PART_PAD, add_hw=hw)))
                            self.labels["bwd"].append("dgrad:%s+%s+wgrad:%s" % (node.name, A.name, A.name))
                    else:
                        self.bwd_ops.append((None, lambda c=c, A=A, G=G, add=add, X=X, dstg=dstg, dst2=dst2, hw=X.pending_hw:
                                             hip.conv_pair(A.Mpix, A.node.cin, dt, 1, G, c.wd, None, add, X.bits, dstg, A.wd, None, A.src.data, dst2,
                                                           add_hw=hw)))
                        self.labels["bwd"].append("dgrad:%s+%s" % (node.name, A.name))
                    X.grad_written, X.pending = True, None
                    A.src.grad_written = True
                    A.dgrad_done_by_pair = True
                    continue
                assert X.pending_hw is None or X.grad_written, "compact residual gradient reached a non-fused data gradient (%s)" % node.name
                dstg = X.grad_buf()
                mask = (X.bits if X.bits is not None else X.data) if X.spec.relu else None
                mflag = hip.EPI_MASK_BITS if (X.spec.relu and X.bits is not None) else 0
                if lvl_n is not None and mflag == 0 and not getattr(c, "gd_scatter", False) and self._dense_multi_ok(c):
                    prev = [L for L in pend_dd if L["dst"] is dstg]
                    if prev and add is dstg and prev[-1].get("src1") is None:
                        prev[-1].update(src1=G, wgt1=c.wd, K1=c.npad, name=prev[-1]["name"] + "+" + node.name)
                    else:
                        if prev:
                            flush_dense_dgrads()           # a third writer of the same tensor: accumulate behind the launch that holds the first two
                        pend_dd.append(dict(name=node.name, level=lvl_n, src0=G, wgt0=c.wd, K0=c.npad, N=node.cin, M=B, add=add, mask=mask, dst=dstg, flags=0))
                    X.grad_written, X.pending = True, None
                    continue
                if getattr(c, "gd_scatter", False):
                    if add is None:
                        # first (and only) contribution: everything off the sampled grid is zero -- and STAYS zero: the scattered data gradient
                        # writes nothing but the sampled pixels, the tensor has a gradient buffer of its own and no other launch writes it, so
                        # the buffer is cleared ONCE, here, instead of by a fill launch in every step (rounds 2-4: three launches, 33 us per
                        # cfg2 step, 150 MB written for nothing).  URSO_ZERO_EVERY_STEP=1 restores the per-step fill (A/B, debugging).
                        dstg.zero_()
                        # (only when EVERY launch that writes this buffer is such a scattered data gradient: a dense one accumulating in place
                        # behind this one would leave its values off the grid for the next step to find)
                        writers = [cc for cc in self.convs.values() if cc.src is X and getattr(cc, "gd", None) is not None]
                        once = (all(getattr(cc, "gd_scatter", False) for cc in writers) and not any(cc.res is X for cc in self.convs.values())
                                # ... nor a max-pool reading X (urso_maxpool_bwd writes X.grad densely) or X being a network output
                                and not any(n.op == "pool" and n.src.id == X.spec.id for n in g.nodes)
                                and not any(t.id == X.spec.id for t in g.outputs.values()))
                        if not once or os.environ.get("URSO_ZERO_EVERY_STEP", "0") == "1":
                            self.bwd_ops.append((None, lambda t=dstg: hip.zero_fill(t)))
                            self.labels["bwd"].append("zero:" + node.name)
                        else:
                            # URSO_CHECK_CLEARED=1 (debugging; check_cleared_once()): the pixels no step writes must still be zero after any
                            # number of steps -- a tool that writes gradient buffers (teacher-forced tests, dumps) would break that silently
                            st = int(getattr(c, "gd_scatter_stride", 2))
                            self._cleared_once.append((node.name, dstg, X.spec.h, X.spec.w, X.spec.c, st, st))
                    elif add is not dstg:
                        raise AssertionError("scattered dgrad into %s needs an in-place accumulate" % node.name)
                self.bwd_ops.append((None, lambda c=c, G=G, add=add, mask=mask, dstg=dstg, mflag=mflag:
                                     hip.conv_igemm_ex(c.gd, dt, mflag, G, c.wd, None, add, mask, dstg, None,
                                                       self.halo_ws if (c.halo_d and add is None and not mflag) else (self.igemm_ws if c.ws_d else None))))
                self.labels["bwd"].append("dgrad:" + node.name)
                X.grad_written, X.pending = True, None
        flush_dense_dgrads()
        flush_dense_wgrads()
        flush_wgrads()
        flush_hw_pair()
        # the descriptor table is complete: upload it, plan the block maps and resolve the batched placeholders
        self._upload_param_table()
        resolved, labels = [], []
        assert len(self.bwd_ops) == len(self.labels["bwd"])
        for (tag, op), lab in zip(self.bwd_ops, self.labels["bwd"]):
            if isinstance(op, tuple):
                ph, k = op
                ids = [self.convs[nm].desc_id for nm in groups[k]]
                nb_ = self.pbatch.plan(ph, k, ids)
                if nb_ == 0:
                    continue                                   # nothing to launch (e.g. no split layer in the bucket)
                if self.fused_sqnorm and ph in (hip.PB_FINALIZE_MAT, hip.PB_FINALIZE_VEC):
                    s0, s1 = sq_slots[0], sq_slots[0] + nb_
                    sq_slots[0] = s1
                    op = (lambda ph=ph, k=k, s0=s0, s1=s1: self.pbatch.run(ph, k, dt, sqpart=self.sqpart[s0:s1]))
                else:
                    op = (lambda ph=ph, k=k: self.pbatch.run(ph, k, dt))
            resolved.append((tag, op)); labels.append(lab)
        self.bwd_ops, self.labels["bwd"] = resolved, labels
        self.sqpart = torch.zeros(max(sq_slots[0], 1), dtype=torch.float32, device=dev) if self.fused_sqnorm else None
        # ---------------------------------------------------------------- optimizer
        n = self.n_flat
        self.adam = str(getattr(cfg, "OPTIMIZER", "SGD")).upper() != "SGD"          # net.py:979-983: anything else is Adam(amsgrad)
        clip = float(cfg.GRADIENT_CLIP_NORM or 0.0)
        # hyper-parameters live in device memory so that one captured graph serves a changing learning rate; they are created ONCE:
        # a later set_trainable() (which rebuilds the plan) must not undo compile(lr) / set_lr() (net.py:1073-1075 calls them in
        # either order)
        if self.adam:
            eps = 1e-4 if getattr(cfg, "F16", False) else 1e-7                      # K.epsilon(); net.py:590-593 sets 1e-4 in F16 mode
            if getattr(self, "hyper", None) is None or self.hyper.numel() != 8:
                self.hyper = torch.tensor([float(cfg.LEARNING_RATE), 0.9, 0.999, eps, clip, 0.0, 1.0 - 0.9, 1.0 - 0.999],
                                          dtype=torch.float32, device=dev)
            if not hasattr(self, "flat_v2") or self.flat_v2.numel() != self.flat_w.numel():
                self.flat_v2, self.flat_vhat = torch.zeros_like(self.flat_w), torch.zeros_like(self.flat_w)
        elif getattr(self, "hyper", None) is None or self.hyper.numel() != 3:
            self.hyper = torch.tensor([float(cfg.LEARNING_RATE), float(cfg.LEARNING_MOMENTUM), clip], dtype=torch.float32, device=dev)
        self.normsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sq_ws = torch.empty(hip.sqnorm_ws_bytes(n) // 4, dtype=torch.float32, device=dev)
        if self.fused_sqnorm:
            self.opt_ops.append(lambda: hip.sqnorm_final(self.sqpart, self.normsq))
        else:
            self.opt_ops.append(lambda: hip.sqnorm(n, self.flat_g, self.sq_ws, self.normsq))
        if self.adam:
            self.opt_ops.append(lambda: hip.adam_amsgrad_clip(n, self.flat_w, self.flat_g, self.flat_v, self.flat_v2, self.flat_vhat,
                                                              self.hyper, self.normsq))
        else:
            self.opt_ops.append(lambda: hip.sgd_momentum_clip(n, self.flat_w, self.flat_g, self.flat_v, self.hyper, self.normsq))
        self.labels["opt"] += ["sqnorm", "adam" if self.adam else "sgd"]
        self.flat_g.zero_()
        self._fork_weight_gradients()

    def _fork_weight_gradients(self):
        """Weight-gradient launches are leaves of the backward pass: nothing reads their partials before the bucket's reduction, every tensor
        has a gradient buffer of its own and nothing later overwrites what they read.  So they may run LATER than where they stand in the
        chain, and beside it: on a second stream, captured into the same hipGraph as a branch.
        What pays (tools/probes/overlap_probe.py, profiles/r05_overlap_probe.txt): an HBM-bound chain and an MFMA-bound chain side by side
        finish 9-12 % sooner than one after the other; two chains of the same kind only get in each other's way (every grid is a static
        partition of its tiles over all 256 CUs, and a block that finds its CU held by the other launch runs behind it).  Hence the
        COMPLEMENTARY fork (URSO_WGRAD_STREAM=2; the default until round 5, opt-in since): only the arithmetic-heavy weight gradients (>= 150 FLOP per byte: stages 4-5,
        the 3x3 layers of stage 3, bottleneck_layer, the heads) leave the chain, and those whose dz is final before the data gradients reach
        stage 3 are DEFERRED to that point, where the chain turns HBM-bound (stage-3 / stage-2 data gradients and weight gradients).  Join =
        the bucket's reduction.  cfg2, alternating in one gpurun call: 7.13 -> 6.97 ms (-2.1 %), bit-identical to the single chain (same
        kernels, same operands; tests/test_model_gpu.py).  URSO_WGRAD_STREAM=1 is the plain fork of round 5's first experiment (every weight
        gradient, where it stands: +2.4 %), 0 the single chain.  Plans whose buckets are finalised inside the backward pass (data-parallel
        runs: ursonet_amd/dp.py) keep the single chain -- a deferred launch would land behind its bucket's finalisation.
        One edge per fork: the side stream waits for the chain only when the chain has moved since the last side launch.  A captured graph in
        which twenty side nodes each carried their own (redundant) edge from the same chain node ran the chain node BEHIND that fan before
        its predecessors (stale operands from the previous replay; ROCm 7.2, profiles/r05_fork.txt) -- eager streams did not."""
        self.wgrad_stream = None
        self._single_chain_always = False
        self._graph_events = []
        # OPT-IN since round 6: hipGraphLaunch of a forked graph segfaulted inside the ROCm 7.2 runtime in a long-lived process (a pytest
        # run, after ~22 engines of various sizes had come and gone; reproducible for that sequence, gone with URSO_WGRAD_STREAM=0, not cured by
        # keeping the capture's events alive) -- a failure no verification can turn into a fallback.  bench.py opts in (a fresh process: the
        # same path has run hundreds of times there without incident, and it checks the fork against the chain first); the tests of the fork
        # run in processes of their own (tests/workers/fork_worker.py).
        mode = int(os.environ.get("URSO_WGRAD_STREAM", "0"))
        if mode not in (1, 2) or self.mode != "training" or getattr(self, "no_wgrad_fork", False):       # (no_wgrad_fork: set by ursonet_amd/dp.py)
            return
        import re
        labs = self.labels["bwd"]
        assert len(labs) == len(self.bwd_ops)
        fin = ("reduce", "finalize_mat", "finalize_vec", "finalize", "unpack")
        side = lambda lab: lab is not None and lab.startswith(("wgrad:", "wgrad_heads:"))
        if mode == 2:
            if any(getattr(c, "batch_bn", False) for c in self.convs.values()):
                return                           # batch-statistics BN: the chain is HBM-bound passes end to end; measured 17.05 -> 17.42 ms with the fork
            es = 4 if self.dt == hip.F32 else 2

            def work(name, back=False):          # (FLOP, bytes) of a layer's weight gradient (back: of its data gradient): operands once, result once
                c = self.convs.get(name)
                g = getattr(c, "gf", None) if c is not None else None
                if g is None:
                    return 0.0, 0.0              # heads: small, latency-bound
                m_out, m_in, k = g.B * g.OH * g.OW, g.B * g.H * g.W, g.KH * g.KW * g.C
                return 2.0 * m_out * k * g.N, (m_in * g.C + m_out * g.N) * es + (es if back else 4.0) * k * g.N

            def names(lab):
                return [n for part in lab.split("+wgrad:")[0:1] + lab.split("+wgrad:")[1:] for n in part.split(":", 1)[-1].split("+") if n != "maxpool_bwd"]

            def est_ms(lab):                     # what the launch takes at the rates launches of its kind reach here (1.0 PFLOP/s, 4.5 TB/s)
                fl = by = 0.0
                for n in names(lab):
                    f, b = work(n, lab.startswith("dgrad"))
                    fl += f; by += b
                return max(fl / 1.0e12, by / 4.5e9)

            def intensity(lab):
                fl = by = 0.0
                for n in names(lab):
                    f, b = work(n)
                    fl += f; by += b
                return fl / by if by else 1e9

            def wg(lab):
                return lab is not None and lab.startswith(("wgrad:", "wgrad_heads:")) and "dgrad" not in lab
            at = next((i for i, l in enumerate(labs) if l is not None and re.match(r"dgrad:res[23]", l)), None)
            if at is None:
                return
            EARLY, LATE = 150.0, 400.0           # FLOP per byte (swept 50 ... 300 / 250 ... none: 6.93 ... 6.98 ms, nothing to choose between them)
            # the HBM-bound stretch of the chain behind the deferral point, and as much arithmetic-heavy weight-gradient work as fits beside it (1.2 x
            # its estimate: cfg4, ResNet-101 at batch 16, has more such work than stretch -- 0.4 / 0.8 / 1.2 / 2.0: 8.48 / 8.37 / 8.33 / 8.33 ms, chain 8.49):
            # the LAST such launches in front of the point (the others stay where they are, on the chain)
            room = 1.2 * sum(est_ms(l) for l in labs[at:] if l is not None and l.startswith(("dgrad:", "wgrad:")) and not (wg(l) and intensity(l) >= LATE))
            early = []
            for i in range(at - 1, -1, -1):
                if wg(labs[i]) and (labs[i].startswith("wgrad_heads:") or intensity(labs[i]) >= EARLY):
                    room -= est_ms(labs[i])
                    if room < 0:
                        break
                    early.append(i)
            early.reverse()
            if not early or any(labs[i] is not None and labs[i].split(":")[0] in fin for i in range(early[0], at)):
                return                           # nothing to defer, or a bucket is finalised inside the region: the chain stays as it is
            moved = set(early)
            late = set(i for i in range(at, len(labs)) if wg(labs[i]) and intensity(labs[i]) >= LATE)      # 3x3 layers of >= 128 channels behind the point
            side_ids = set(id(self.bwd_ops[i]) for i in moved | late)
            order = [i for i in range(at) if i not in moved] + early + list(range(at, len(labs)))
            self.bwd_ops = [self.bwd_ops[i] for i in order]
            self.labels["bwd"] = [labs[i] for i in order]
            side = None
        self.wgrad_stream = _side_stream(self.device)
        self._side_open = False
        self._main_moved = True
        self._single_chain = False

        def on_side(op):
            def run():
                if self._single_chain or self._single_chain_always:
                    return op()
                main = torch.cuda.current_stream(self.device)
                if self._main_moved or not self._side_open:         # one edge per fork (see above)
                    self._stream_edge(main, self.wgrad_stream)
                    self._main_moved = False
                with torch.cuda.stream(self.wgrad_stream):
                    op()
                self._side_open = True
            return run

        def joined(op):
            def run():
                self._join_weight_gradients()
                self._main_moved = True
                return op()
            return run

        def on_main(op):
            def run():
                self._main_moved = True
                return op()
            return run
        ops = []
        for item, lab in zip(self.bwd_ops, self.labels["bwd"]):
            tag, op = item
            if (id(item) in side_ids) if side is None else side(lab):
                op = on_side(op)
            elif lab is not None and lab.split(":")[0] in fin:
                op = joined(op)
            elif callable(op):
                op = on_main(op)
            ops.append((tag, op))
        self.bwd_ops = ops
        self.opt_ops[0] = joined(self.opt_ops[0])

    def _stream_edge(self, src, dst):
        """dst waits for what src holds now.  Stream.wait_stream() records a temporary event and lets Python destroy it right away; while a
        stream is capturing, that event becomes part of the graph being built, and a hipGraph replay long after its events were destroyed is
        the one place this code base has seen hipGraphLaunch segfault (round 5: after ~20 dropped engines; round 6: after 22 tests of one
        process).  Events recorded during a capture therefore live as long as the engine that owns the graph."""
        ev = torch.cuda.Event()
        ev.record(src)
        dst.wait_event(ev)
        if torch.cuda.is_current_stream_capturing():
            self._graph_events.append(ev)

    def _join_weight_gradients(self):
        if self.wgrad_stream is not None and self._side_open:
            self._stream_edge(self.wgrad_stream, torch.cuda.current_stream(self.device))
            self._side_open = False

    def _plan_relu_bitmasks(self):
        """A post-ReLU tensor gets a bit mask (1/16 of its bytes) for the backward pass when the conv that produces it can
        emit one from its forward epilogue and every data-gradient pass into it can consume one (urso_conv_igemm_bits_ok);
        otherwise the tensor itself serves as the mask, as before."""
        dt = self.dt
        import os
        # URSO_RELU_BITS: 0 off, 1 block outputs only (pointwise producer WITH a residual and pointwise consumers: all of them run in
        # the HBM-bound DMA kernel, where the emission hides behind memory time), 2 every eligible pointwise tensor
        level = int(os.environ.get("URSO_RELU_BITS", "1"))      # measured: +1 % on the cfg2 step (data gradient -0.17 ms, forward +0.12 ms)
        for X in self.acts.values():
            X.bits = None
            if not level or not X.spec.relu or X.numel % 8:
                continue
            prod = [c for c in self.convs.values() if c.dst is X]
            cons = [c for c in self.convs.values() if c.src is X and getattr(c, "gd", None) is not None]
            if len(prod) != 1 or not cons or not prod[0].node.relu:
                continue
            P = prod[0]
            pw = lambda n: (not n.dense) and n.kh == 1 and n.kw == 1
            # ... or bottleneck_layer behind the trunk's last block: its parity-class data gradient (conv_bneck.hip) takes the bit mask too
            bn_ = lambda c: (hip.get_option("bneck") and not c.node.dense and c.node.kh == 3 and c.node.kw == 3 and c.node.stride == 2 and
                             c.npad == 32 and c.node.cin % 64 == 0 and not c.batch_bn)
            if not (pw(P.node) and P.node.stride == 1 and P.N % 32 == 0 and all(pw(c.node) or bn_(c) for c in cons)):
                continue
            if level == 1 and P.res is None:
                continue
            if not hip.conv_igemm_bits_ok(P.gf, dt, P.fwd_flags, P.ws_f):
                continue
            if all(hip.conv_igemm_bits_ok(c.gd, dt, 0, c.ws_d) for c in cons):
                X.bits = torch.empty(X.numel // 8, dtype=torch.uint8, device=self.device)

    def _upload_param_table(self):
        self.pbatch = hip.ParamBatch(self._descs, self.device) if self._descs else None
        if self.pbatch is not None:
            self.pbatch.plan(hip.PB_PREP, "all", list(range(len(self._descs))))

    def _build_heads_io(self):
        cfg, B, dev = self.config, self.B, self.device
        self.quat_head = bool(cfg.REGRESS_ORI and cfg.ORIENTATION_PARAM == "quaternion" and not cfg.REGRESS_KEYPOINTS)
        if self.quat_head:
            self.q_out = torch.zeros(B, 4, dtype=torch.float32, device=dev)
            if self.mode == "inference":
                x = self.out_ori
                self.fwd_ops.append(lambda: hip.absdot(B, 4, 8, 1, None, x.data, 1.0, self.dt, self.q_out, None, None))

    def _build_losses(self):
        cfg, g, B, dt, dev = self.config, self.graph, self.B, self.dt, self.device
        lw = cfg.LOSS_WEIGHTS
        self.loss_buf = torch.zeros(4, dtype=torch.float32, device=dev)       # loc_loss, ori_loss, k2_loss, k3_loss (weighted)
        self.rel_norms = torch.zeros(2, dtype=torch.float32, device=dev)
        loc, ori = self.out_loc, self.out_ori
        nloc = g.outputs["loc"].c
        nori = g.outputs["ori"].c if "ori" in g.outputs else 0
        self.row_ws = torch.empty(B, dtype=torch.float32, device=dev)
        if cfg.REGRESS_KEYPOINTS:
            # experimental keypoint mode (net.py:657-659): three MSE losses on k1 (=loc), k2, k3
            self.gt_loc = torch.zeros(B, 3, dtype=torch.float32, device=dev)
            self.gt_ori = torch.zeros(B, 3, dtype=torch.float32, device=dev)      # input_gt_k2
            self.gt_k3 = torch.zeros(B, 3, dtype=torch.float32, device=dev)
            for key, gt, wname, li in (("k1", self.gt_loc, "loc_loss", 0), ("k2", self.gt_ori, "k2_loss", 2), ("k3", self.gt_k3, "k3_loss", 3)):
                a = self.acts[g.outputs[key].id]
                gzt = torch.empty(a.numel, dtype=self.tdt, device=dev)
                a.grad, a.grad_written = gzt, True
                self.loss_ops.append(lambda a=a, gt=gt, w=float(lw.get(wname, 1.)), li=li, gzt=gzt:
                                     hip.mse(B, 3, 8, gt, a.data, w, dt, self.loss_buf[li:li + 1], gzt))
            return
        if (not cfg.REGRESS_LOC and nloc % 8) or (not cfg.REGRESS_ORI and nori % 8):
            raise ValueError("classification heads need a bin count that is a multiple of 8 (got %d / %d)" % (nloc, nori))
        # location head
        gz_loc = torch.empty(loc.numel, dtype=self.tdt, device=dev)
        loc.grad, loc.grad_written = gz_loc, True
        wl = float(lw.get("loc_loss", 1.))
        if cfg.REGRESS_LOC:
            self.gt_loc = torch.zeros(B, 3, dtype=torch.float32, device=dev)
            self.rel_exact = bool(getattr(cfg, "DP_EXACT_REL_LOSS", False))
            if self.rel_exact:
                # exact global batch-Frobenius ratio under data parallelism: norms -> (dp.py sum-all-reduces them) -> loss + gradient;
                # rel_scale = world size (1 on a single GPU, where the two phases reproduce the one-kernel loss)
                if not hasattr(self, "rel_scale"):
                    self.rel_scale = torch.ones(1, dtype=torch.float32, device=dev)
                self.loss_pre_ops.append(lambda: hip.rel_l2_norms(B, 3, 8, self.gt_loc, loc.data, self.rel_norms))
                self.loss_ops.append(lambda: hip.rel_l2_from_norms(B, 3, 8, self.gt_loc, loc.data, wl, self.rel_scale, dt, self.rel_norms,
                                                                   self.loss_buf[0:1], gz_loc))
            else:
                self.loss_ops.append(lambda: hip.rel_l2(B, 3, 8, self.gt_loc, loc.data, wl, dt, self.loss_buf[0:1], gz_loc, self.rel_norms))
        else:
            self.gt_loc = torch.zeros(B, nloc, dtype=torch.float32, device=dev)
            self.loss_ops.append(lambda: hip.softmax_xent(B, nloc, loc.data, self.gt_loc, wl, 1, dt, self.loss_buf[0:1], gz_loc, self.row_ws))
        # orientation head
        gz_ori = torch.empty(ori.numel, dtype=self.tdt, device=dev)
        ori.grad, ori.grad_written = gz_ori, True
        wo = float(lw.get("ori_loss", 1.))
        if cfg.REGRESS_ORI:
            d = 4 if self.quat_head else 3
            self.gt_ori = torch.zeros(B, d, dtype=torch.float32, device=dev)
            qo = self.q_out if self.quat_head else None
            self.loss_ops.append(lambda: hip.absdot(B, d, 8, 1 if self.quat_head else 0, self.gt_ori, ori.data, wo, dt, qo,
                                                    self.loss_buf[1:2], gz_ori))
        else:
            self.gt_ori = torch.zeros(B, nori, dtype=torch.float32, device=dev)
            self.loss_ops.append(lambda: hip.softmax_xent(B, nori, ori.data, self.gt_ori, wo, 1, dt, self.loss_buf[1:2], gz_ori, self.row_ws))

    def _plan_compact_gradients(self, need):
        """A block output X whose only consumers are the stride-2 pointwise layers of the next stage's first block (net.py:121-126:
        branch2a and the shortcut conv) receives a gradient that is zero at three of every four pixels.  The dense tensor is never
        materialised: X.grad is [B, H/2, W/2, C] (the sampled pixels), and everything that reads it runs on that form --
          * the stride-2 layers' data gradients are plain pointwise GEMMs over the sampled pixels (no scatter, no 4x zero fill),
            masked by the sampled rows of X's ReLU bit mask (urso_rows_subsample2);
          * the weight gradient of X's producer is a stride-2 weight gradient (its forward input sampled at the same pixels);
          * the producer's own data gradient is the compact-scatter form (GEMM over the sampled pixels, zeros elsewhere), and the 3x3
            layer below it takes that tensor as a scattered dz operand: its weight gradient runs over the even pixels only;
          * the residual branch hands the compact tensor to the fused backward pair (urso_conv_pair, add_h / add_w).
        Where the residual side is not a fused pair (stages 4-5, pair option off) the dense form is produced once by urso_rows_expand2.
        URSO_COMPACT_GRAD=0 keeps the dense path everywhere."""
        g, dt, B, dev = self.graph, self.dt, self.B, self.device
        if dt == hip.F32 or not getattr(self.config, "COMPACT_GRADIENTS", True) or os.environ.get("URSO_COMPACT_GRAD", "1") == "0":
            return
        convs = list(self.convs.values())
        for X in self.acts.values():
            if X.bits is None or not X.spec.relu or X.spec.h % 2 or X.spec.w % 2 or not need.get(X.spec.id, False):
                continue
            prod = [c for c in convs if c.dst is X]
            cons = [c for c in convs if c.src is X]
            if len(prod) != 1 or not cons or any(c.res is X for c in convs) or any(n.op == "pool" and n.src.id == X.spec.id for n in g.nodes):
                continue
            A = prod[0]
            n = A.node
            if not (not n.stem and not n.dense and n.kh == 1 and n.kw == 1 and n.stride == 1 and A.res is not None and not A.batch_bn and
                    A.npad == A.N and A.res.spec.relu and need.get(A.src.spec.id, False) and A.src.spec.relu and A.src.bits is None):
                continue
            if not all((not c.node.dense) and c.node.kh == 1 and c.node.kw == 1 and c.node.stride == 2 and not c.batch_bn and
                       c.npad == c.N and tuple(c.node.pad) == (0, 0) for c in cons):
                continue
            # the residual gradient must land in a fused backward pair: the block's branch2a (consumer of A.res) is the second layer of
            # a fused forward pair, and A.res has no other consumer
            R = A.res
            rc = [c for c in convs if c.src is R]
            if len(rc) != 1 or sum(1 for c in convs if c.res is R) != 1:
                continue
            X.residual_needs_dense = self.pair_first.get(rc[0].name) is None or R.bits is None     # stages 4-5: expanded (urso_rows_expand2)
            H, W = X.spec.h, X.spec.w
            gd_ok = True
            for c in cons:
                c.gd_compact = hip.geom(B, H // 2, W // 2, c.npad, H // 2, W // 2, c.node.cin, 1, 1)
                gd_ok = gd_ok and hip.conv_igemm_bits_ok(c.gd_compact, dt, 0, 0)
            if not gd_ok:
                continue
            X.compact = (H, W)
            X.grad = torch.empty(X.numel // 4, dtype=self.tdt, device=dev)
            X.bits_compact = torch.empty(X.numel // 32, dtype=torch.uint8, device=dev)
            # producer: stride-2 weight gradient, compact-scatter data gradient
            A.gf_compact = hip.geom(B, H, W, n.cin, H // 2, W // 2, A.npad, 1, 1, 2, 2, 0, 0)
            A.splits = hip.conv_wgrad_splits(A.gf_compact, dt)
            A.desc.splits = max(A.splits, 1)
            A.gd = hip.geom(B, H // 2, W // 2, A.npad, H // 2, W // 2, n.cin, 1, 1, FH=H, FW=W, OSH=2, OSW=2)
            A.gd_scatter = True
            A.gd_scatter_stride = 2
            A.ws_d = 0
            # one layer further down: A's data gradient (dense tensor, zero off the even grid) is the dz of the layer that produced A's
            # input -- its weight gradient is the stride-2 one over the even pixels, reading dz in place (scattered dz operand)
            P = [c for c in convs if c.dst is A.src]
            if len(P) == 1 and not P[0].batch_bn and not P[0].node.stem and not P[0].node.dense and P[0].node.stride == 1:
                pn, Pc = P[0].node, P[0]
                pt, pl = pn.pad
                if pn.src.h == H and pn.src.w == W and (pn.kh > 1 or pn.kw > 1):
                    Pc.gf_compact = hip.geom(B, H, W, pn.cin, H // 2, W // 2, Pc.npad, pn.kh, pn.kw, 2, 2, pt, pl, FH=H, FW=W, OSH=2, OSW=2)
                    Pc.splits = hip.conv_wgrad_splits(Pc.gf_compact, dt)
                    Pc.desc.splits = max(Pc.splits, 1)
            self._sample_block_output(X)

    def _sample_block_output(self, X):
        """A block output that is only ever read through stride-2 pointwise layers (res{2c,3d,4f}_out of ResNet-50: net.py:121-126)
        is COMPUTED only at the pixels they read: its producing c -> 4c layer runs as a 1x1 / stride-2 layer over the even pixels with
        the identity shortcut read on the input grid (URSO_EPI_ADD_SRCGRID), writing the compact tensor the consumers take
        (X.data_compact) and -- training -- the compact ReLU bit mask directly; the dense tensor (335 MB in stage 2 of cfg2) and its bit
        mask are neither written nor read by anything.  Every value that any output, loss or gradient depends on is computed by the same
        arithmetic (bit-identical: tests/test_kernels_gpu.py::test_block_output_computed_at_the_sampled_pixels_only); what is skipped
        is dead.  Needs the compact-gradient plan in training (X.compact) so that the backward pass, too, reads only the compact forms.
        URSO_SAMPLED_OUTPUTS=0 keeps the dense tensor (e.g. to inspect intermediate activations)."""
        dt, B = self.dt, self.B
        if dt == hip.F32 or os.environ.get("URSO_SAMPLED_OUTPUTS", "1") == "0" or getattr(X, "data_compact", None) is None:
            return
        convs = list(self.convs.values())
        prod = [c for c in convs if c.dst is X]
        cons = [c for c in convs if c.src is X]
        if len(prod) != 1 or not cons or any(c.res is X for c in convs) or any(c.xin is not X.data_compact for c in cons):
            return
        if any(n.op == "pool" and n.src.id == X.spec.id for n in self.graph.nodes) or any(t.id == X.spec.id for t in self.graph.outputs.values()):
            return
        A = prod[0]
        n = A.node
        training = self.mode == "training"
        if (n.stem or n.dense or n.kh != 1 or n.kw != 1 or n.stride != 1 or A.batch_bn or not hasattr(A, "fwd_index") or n.cin % 64 or A.npad % 32 or
                A.npad != A.N or X.spec.h % 2 or X.spec.w % 2 or (A.res is not None and A.res.numel != X.numel)):
            return
        if training and (X.compact is None or X.bits is None or getattr(X, "bits_compact", None) is None):
            return                                    # the backward pass would read the dense tensor / bit mask
        H, W = X.spec.h, X.spec.w
        gs = hip.geom(B, H, W, n.cin, H // 2, W // 2, A.npad, 1, 1, 2, 2, 0, 0)
        flags = A.fwd_flags | (hip.EPI_ADD_SRCGRID if A.res is not None else 0) | (hip.EPI_EMIT_BITS if training else 0)
        # (fwd_index goes stale once the pair fusion has dropped launches: the layer's launch is found by its label)
        idx = [i for i, lab in enumerate(self.labels["fwd"]) if lab in ("fwd:%s" % A.name, "fwd:%s+sampled" % A.name)]
        if len(idx) != 1:
            return
        self.fwd_ops[idx[0]] = (lambda A=A, X=X, gs=gs, flags=flags: hip.conv_igemm_ex(
            gs, dt, flags, A.xin, A.wf, A.biasf, A.res.data if A.res is not None else None, None, X.data_compact,
            X.bits_compact if (flags & hip.EPI_EMIT_BITS) else None))
        self.labels["fwd"][idx[0]] = "fwd:%s@sampled" % A.name
        self._sample_layer_below(A, X)
        gather = [i for i, lab in enumerate(self.labels["fwd"]) if lab == "subsample:T%d" % X.spec.id]
        for i in reversed(gather):                   # the gather pass that filled X.data_compact from the dense tensor (plans without the
            del self.fwd_ops[i]; del self.labels["fwd"][i]      # register-filter kernel's second store) has nothing to read any more
        X.fwd_sampled = True

    def _sample_layer_below(self, A, X):
        """One layer further down: the 3x3 layer P whose output Y only feeds the sampled layer A (res{2c,3d,4f}_branch2b, net.py:143) is
        needed at the even pixels only as well -- A's forward pass, A's stride-2 weight gradient and the mask of A's compact-scatter data
        gradient all read Y at those pixels.  P runs as a 3x3 / stride-2 layer whose results are SCATTERED to the even pixels of the dense
        Y buffer (destination scatter of urso_conv_igemm): every reader finds its values where it always did, the other three quarters
        of the buffer are never written and never read.  P's own backward pass does not involve Y's values."""
        convs = list(self.convs.values())
        Y = A.src
        prod = [c for c in convs if c.dst is Y]
        if len(prod) != 1 or any(c.src is Y and c is not A for c in convs) or any(c.res is Y for c in convs) or Y.bits is not None:
            return
        if any(n.op == "pool" and n.src.id == Y.spec.id for n in self.graph.nodes) or getattr(Y, "data_compact", None) is not None:
            return
        P = prod[0]
        n = P.node
        H, W = X.spec.h, X.spec.w
        if (n.stem or n.dense or n.kh != 3 or n.kw != 3 or n.stride != 1 or tuple(n.pad) != (1, 1) or P.batch_bn or P.res is not None or n.cin % 64 or
                P.npad != P.N or n.src.h != H or n.src.w != W or n.dst.h != H or n.dst.w != W or P.xin is not P.src.data):
            return
        idx = [i for i, lab in enumerate(self.labels["fwd"]) if lab == "fwd:%s" % P.name]
        if len(idx) != 1:
            return
        gs = hip.geom(self.B, H, W, n.cin, H // 2, W // 2, P.npad, 3, 3, 2, 2, 1, 1, FH=H, FW=W, OSH=2, OSW=2)
        self.fwd_ops[idx[0]] = (lambda P=P, gs=gs: hip.conv_igemm_ex(gs, self.dt, P.fwd_flags, P.xin, P.wf, P.biasf, None, None, P.dst.data, None, None))
        self.labels["fwd"][idx[0]] = "fwd:%s@sampled" % P.name
        Y.fwd_scattered = True

    def _stem_pool_node(self, g, node, c, dt):
        """The pool node the stem's kernel can compute as well (urso_stem_conv_pool), or None: the stem with its ReLU and a folded BN,
        read by nothing but ONE max-pool, in a 16-bit dtype on a geometry the kernel takes."""
        if not node.stem or not node.relu or c.batch_bn or c.res is not None or node.out_f32 or dt == hip.F32:
            return None
        if any(t.id == node.dst.id for t in g.outputs.values()) or (g.feat is not None and getattr(g.feat, "id", None) == node.dst.id):
            return None
        users = [n for n in g.nodes if n is not node and (n.src.id == node.dst.id or (n.op == "conv" and n.residual is not None and n.residual.id == node.dst.id))]
        if len(users) != 1 or users[0].op != "pool" or not hip.stem_conv_pool_ok(c.gf, dt):
            return None
        return users[0]

    def _dense_levels(self):
        """Dense head layers (net.py:288-352) -> depth behind the bottleneck features (1 = reads them)."""
        g, lv = self.graph, OrderedDict()
        prod = {n.dst.id: n for n in g.nodes if n.op != "pool"}
        for n in g.nodes:
            if n.op == "pool" or not n.dense:
                continue
            p = prod.get(n.src.id)
            lv[n.name] = lv[p.name] + 1 if (p is not None and p.name in lv) else 1
        return lv

    def _dense_multi_ok(self, c):
        n = c.node
        return bool(self.dense_multi and n.dense and not c.batch_bn and c.res is None and n.cin % 8 == 0 and c.npad % 8 == 0 and self.B <= 32)

    def _backward_order(self):
        """reversed(graph.nodes), with the Dense head layers taken depth by depth (both final layers, then both dense_0 layers ...) instead
        of branch by branch, so that the data gradients of one depth can share a launch (urso_dense_multi)."""
        nodes = list(reversed(self.graph.nodes))
        if not getattr(self, "dense_multi", False):
            return nodes
        lv = self._dense_levels()
        idx = [i for i, n in enumerate(nodes) if n.op != "pool" and n.name in lv]
        if not idx or idx != list(range(idx[0], idx[0] + len(idx))):
            return nodes
        run = sorted(nodes[idx[0]:idx[-1] + 1], key=lambda n: -lv[n.name])        # stable: the branches keep their order inside a depth
        return nodes[:idx[0]] + run + nodes[idx[-1] + 1:]

    def _fuse_dense_heads(self):
        """Forward plan rewrite: the Dense layers of one depth behind the bottleneck features (loc_dense_0 + ori_dense_0, loc_final + ori_final;
        net.py:288-352) run side by side in one launch (urso_dense_multi: the same kernel body per layer, same results).  URSO_DENSE_MULTI=0
        keeps one launch per layer."""
        self.dense_multi = (os.environ.get("URSO_DENSE_MULTI", "1") != "0" and self.dt != hip.F32 and bool(hip.get_option("dense")))
        if not self.dense_multi:
            return
        lv = self._dense_levels()
        convs = [self.convs[nm] for nm in lv]
        lab = self.labels["fwd"]
        pos = [lab.index("fwd:" + c.name) if ("fwd:" + c.name) in lab else -1 for c in convs]
        if (len(convs) < 2 or min(pos) < 0 or sorted(pos) != list(range(min(pos), min(pos) + len(pos))) or
                not all(self._dense_multi_ok(c) for c in convs)):
            self.dense_multi = False
            return
        ops, labels = [], []
        for depth in sorted(set(lv.values())):
            cs = [c for c in convs if lv[c.name] == depth]
            for i in range(0, len(cs), hip.DENSE_MULTI_MAX):
                part = cs[i:i + hip.DENSE_MULTI_MAX]
                m = hip.DenseMulti([dict(src0=c.xin, wgt0=c.wf, K0=c.node.cin, N=c.npad, M=self.B, bias=c.biasf, dst=c.dst.data, flags=c.fwd_flags)
                                    for c in part], self.dt)
                ops.append(lambda m=m: m.run())
                labels.append("fwd:" + "+".join(c.name for c in part))
        a, b = min(pos), max(pos) + 1
        self.fwd_ops[a:b] = ops
        lab[a:b] = labels

    def _fuse_pointwise_pairs(self):
        """Forward plan rewrite: a block-closing pointwise layer (c -> 4c, + residual, ReLU; c = 64 or 128: stages 2 and 3) directly
        followed by the next block's opening pointwise layer (4c -> c, ReLU) becomes ONE launch (urso_conv_pair): the block output is written once and
        the second layer reads it from LDS.  self.pair_first maps the second layer's name to the first layer's _Conv; the backward
        plan mirrors the fusion for the two data gradients.  URSO_OPT_PAIR=0 (hip.options(pair=0)) keeps the layers apart."""
        self.pair_first = {}
        self.shortcut_folded = []          # projection shortcuts computed inside a fused forward pair (conv_pairs.hip)
        g, dt = self.graph, self.dt
        if dt == hip.F32 or not hip.get_option("pair"):
            return
        convs = [n for n in g.nodes if n.op != "pool"]
        drop = []
        for a_node, b_node in zip(convs[:-1], convs[1:]):
            A, Bc = self.convs[a_node.name], self.convs[b_node.name]

            def plain(n, c):
                return (not n.stem and not n.dense and n.kh == 1 and n.kw == 1 and n.stride == 1 and n.relu and not n.out_f32 and
                        not c.batch_bn and c.npad == c.N and hasattr(c, "fwd_index"))
            if not (plain(a_node, A) and plain(b_node, Bc)):
                continue
            if a_node.residual is None or b_node.residual is not None or b_node.src.id != a_node.dst.id:
                continue
            M = self.B * a_node.dst.h * a_node.dst.w
            if not (a_node.cin == b_node.cout and hip.conv_pair_ok(M, dt, a_node.cin, a_node.cout)):
                continue
            A.Mpix = M
            # the residual operand of a stage's first block is its projection shortcut (a conv without activation, net.py:121-126): where
            # that is a plain 64 -> 256 pointwise layer with no other reader (stage 2) it becomes 64 more columns of the pair's first
            # GEMM (conv_pairs.hip) -- its own launch and its output tensor disappear; the backward plan never read that tensor
            S = [self.convs[n.name] for n in convs if n.dst.id == a_node.residual.id]
            readers = [n for n in g.nodes if (n.op == "pool" and n.src.id == a_node.residual.id) or
                       (n.op != "pool" and (n.src.id == a_node.residual.id or (n.residual is not None and n.residual.id == a_node.residual.id)))]
            if (len(S) == 1 and hip.get_option("pair") in (1, 2) and a_node.cin == 64 and M % 64 == 0 and len(readers) == 1 and
                    (lambda n, c: not n.stem and not n.dense and n.kh == 1 and n.kw == 1 and n.stride == 1 and not n.relu and not n.out_f32 and
                     n.residual is None and n.cin == 64 and n.cout == a_node.cout and not c.batch_bn and c.npad == c.N and
                     hasattr(c, "fwd_index"))(S[0].node, S[0])):
                Sc = S[0]
                self.fwd_ops[A.fwd_index] = (lambda A=A, Bc=Bc, Sc=Sc: hip.conv_pair_shortcut(
                    A.Mpix, dt, A.src.data, A.wf, A.biasf, Sc.src.data, Sc.wf, Sc.biasf, A.dst.bits, A.dst.data, Bc.wf, Bc.biasf, Bc.dst.data))
                self.labels["fwd"][A.fwd_index] = "fwd:%s+%s+%s" % (a_node.name, Sc.name, b_node.name)
                drop.append(Sc.fwd_index)
                self.shortcut_folded.append(Sc.name)
            else:
                self.fwd_ops[A.fwd_index] = (lambda A=A, Bc=Bc: hip.conv_pair(A.Mpix, A.node.cin, dt, 0, A.src.data, A.wf, A.biasf, A.res.data, A.dst.bits,
                                                                             A.dst.data, Bc.wf, Bc.biasf, None, Bc.dst.data))
                self.labels["fwd"][A.fwd_index] = "fwd:%s+%s" % (a_node.name, b_node.name)
            drop.append(Bc.fwd_index)
            self.pair_first[b_node.name] = A
        for i in sorted(drop, reverse=True):
            del self.fwd_ops[i]
            del self.labels["fwd"][i]

    def _fuse_entry_shortcuts(self):
        """Forward plan rewrite (round 6): the block-closing pointwise layer of a stage's FIRST block (`res{4,5}a_branch2c` + BatchNorm, then
        Add with the projection shortcut and ReLU, net.py:148-157) and the shortcut itself (`res{4,5}a_branch1` + BatchNorm: a pointwise layer
        on the same pixels, no activation) become ONE launch with two reduction segments (urso_conv_pointwise2): out = ReLU(b . W2c^T +
        x . Wbr1^T + (bias2c + biasbr1)).  The shortcut's launch, its output tensor (84 MB at cfg2's stage 4: written, then read back as the
        residual) and one rounding step disappear; the weight preparation adds the shortcut's folded bias to branch2c's (urso_param_desc::
        bias_from).  Stages whose branch2c already sits in a fused pair (2-3) keep that form.  Nothing in the backward plan reads the
        shortcut's output (its gradient is the block output's).  URSO_ENTRY_FWD2=0 keeps the two launches (A/B)."""
        g, dt, B = self.graph, self.dt, self.B
        self.n_entry_fwd2 = 0
        if dt == hip.F32 or os.environ.get("URSO_ENTRY_FWD2", "1") == "0":
            return
        convs = [n for n in g.nodes if n.op != "pool"]
        lab = self.labels["fwd"]
        drop = []
        outs = set(t.id for t in g.outputs.values()) | ({g.feat.id} if getattr(g, "feat", None) is not None and hasattr(g.feat, "id") else set())
        for a_node in convs:
            A = self.convs[a_node.name]
            if (a_node.stem or a_node.dense or a_node.kh != 1 or a_node.kw != 1 or a_node.stride != 1 or not a_node.relu or a_node.out_f32 or
                    a_node.residual is None or A.batch_bn or A.npad != A.N or ("fwd:" + A.name) not in lab or getattr(A.dst, "fwd_sampled", False)):
                continue
            S = [self.convs[n.name] for n in convs if n.dst.id == a_node.residual.id]
            if len(S) != 1:
                continue
            Sc, sn = S[0], S[0].node
            readers = [n for n in g.nodes if (n.op == "pool" and n.src.id == a_node.residual.id) or
                       (n.op != "pool" and (n.src.id == a_node.residual.id or (n.residual is not None and n.residual.id == a_node.residual.id)))]
            M = B * a_node.dst.h * a_node.dst.w
            if (len(readers) != 1 or a_node.residual.id in outs or sn.stem or sn.dense or sn.kh != 1 or sn.kw != 1 or sn.relu or sn.out_f32 or
                    sn.residual is not None or Sc.batch_bn or Sc.npad != Sc.N or Sc.N != A.N or ("fwd:" + Sc.name) not in lab or
                    sn.dst.h != a_node.dst.h or sn.dst.w != a_node.dst.w or
                    Sc.xin.numel() != M * sn.cin or A.xin.numel() != M * a_node.cin):       # both read dense [M][C] tensors (a strided shortcut reads the compact input)
                continue
            fl = hip.EPI_RELU
            if not (hip.conv_pointwise2_ok(B, a_node.dst.h, a_node.dst.w, a_node.cin, sn.cin, A.N, dt, fl) and
                    (A.N % 32 == 0 and hip.conv_pointwise2_ok(B, a_node.dst.h, a_node.dst.w, a_node.cin, sn.cin, A.N, dt, fl | hip.EPI_EMIT_BITS))):
                continue
            iA = lab.index("fwd:" + A.name)
            self.fwd_ops[iA] = (lambda A=A, Sc=Sc, oh=a_node.dst.h, ow=a_node.dst.w, c0=a_node.cin, c1=sn.cin: hip.conv_pointwise2(
                B, oh, ow, c0, c1, A.N, dt, hip.EPI_RELU | (hip.EPI_EMIT_BITS if A.dst.bits is not None else 0),
                A.xin, A.wf, Sc.xin, Sc.wf, A.biasf, None, A.dst.data, A.dst.bits))
            lab[iA] = "fwd:%s+%s" % (A.name, Sc.name)
            A.desc.bias_from = Sc.desc_id + 1
            A.shortcut_inside = Sc
            drop.append(lab.index("fwd:" + Sc.name))
            self.shortcut_folded.append(Sc.name)
            Sc.dst.data = torch.empty(0, dtype=self.tdt, device=self.device)      # never written any more: a read fails instead of finding stale bytes
            self.n_entry_fwd2 += 1
        for i in sorted(drop, reverse=True):
            del self.fwd_ops[i]
            del lab[i]

    # ------------------------------------------------------------------ execution
    def run_prep(self):
        for op in self.prep_ops:
            op()

    def run_forward(self):
        for op in self.fwd_ops:
            op()

    def run_backward(self):
        for op in self.loss_pre_ops + self.loss_ops:
            op()
        for _, op in self.bwd_ops:
            op()

    def run_optimizer(self):
        for op in self.opt_ops:
            op()

    PLAN_OPTIONS = ("cus", "wgrad_blocks", "wgrad_narrow", "wgrad_big", "hwgrad", "grid_cap", "pair", "stem", "stem_pool", "c3", "bneck", "dense")

    def check_cleared_once(self):
        """Debug check of the plan-time clearing invariant (the `once` branch of _build_plan): every pixel OFF the sampled grid of a
        gradient buffer that is cleared once must still be zero.  Returns the offending layers (empty = invariant holds).  Step() calls it
        after every replay when URSO_CHECK_CLEARED=1."""
        bad = []
        for name, buf, h, w, c, sy, sx in self._cleared_once:
            t = buf[:self.B * h * w * c].view(self.B, h, w, c)
            off = t.clone()
            off[:, ::sy, ::sx, :] = 0
            if bool(off.any()):
                bad.append(name)
        return bad

    def _planning_options(self):
        return tuple(hip.get_option(o) for o in self.PLAN_OPTIONS)

    def _check_plan_options(self):
        now = self._planning_options()
        if now != self._plan_options:
            diff = ", ".join("%s %d -> %d" % (o, a, b) for o, a, b in zip(self.PLAN_OPTIONS, self._plan_options, now) if a != b)
            raise RuntimeError("kernel-policy options changed since this engine planned its step (%s): the split counts and partial workspaces of the "
                               "weight-gradient launches are sized at plan time -- build the Engine under the options it runs with" % diff)

    def step_eager(self):
        """One training step, launched kernel by kernel (used for capture, profiling and debugging)."""
        self._check_plan_options()
        self.run_prep(); self.run_forward(); self.run_backward(); self.run_optimizer()

    def param_pass_bytes(self, label):
        """Algorithmic bytes of a batched parameter-sized launch, from the descriptor table (the library's launch profiler reports none for
        them: a launch covers many layers).  prep: fp32 filter in, compute-type forward + flipped layouts out; reduce: the split partials of
        the layers with more than FUSE_MAX partials in, their fp32 sum out; finalize_mat: fp32 filter + the partials the launch sums itself
        (2 ... FUSE_MAX) or the reduced sum in, fp32 gradient out.  None for other labels."""
        FUSE_MAX = int(os.environ.get("URSO_FUSE_REDUCE_MAX", "16"))
        es = 4 if self.dt == hip.F32 else 2
        kind = label.split(":")[0]
        if kind == "prep" and "batched" in label:
            ds = list(self._descs)
        elif kind in ("reduce", "finalize_mat") and "bucket" in label:
            k = int(label.split("bucket")[1])
            ds = [self.convs[nm].desc for nm in self._bucket_groups[k]]
        else:
            return None
        total = 0.0
        for d in ds:
            K, N, npad, sp = int(d.K), int(d.N), int(d.npad), max(int(d.splits), 1)
            if kind == "prep":
                total += K * N * 4 + K * npad * es * (2 if d.wd else 1)
            elif kind == "reduce":
                if sp > FUSE_MAX:
                    total += (sp + 1) * K * npad * 4
            else:
                total += 2 * K * N * 4 + (sp if 1 < sp <= FUSE_MAX else 1) * K * npad * 4
        return total

    def profile_step(self):
        """One eager training step with the library's HIP-event launch profiler on.
        Returns [(label, kernel_id, ms, flops, bytes, n_kernels, device_symbol)] in launch order."""
        labels = (self.labels["prep"] + self.labels["fwd"] + ["loss"] * (len(self.loss_pre_ops) + len(self.loss_ops)) +
                  [l for l in self.labels["bwd"] if l is not None] + self.labels["opt"])
        torch.cuda.synchronize(self.device)
        hip.prof_collect()
        hip.prof_enable(True)
        self._single_chain = True               # every launch timed alone: the weight gradients stay on the chain (_fork_weight_gradients)
        try:
            self.step_eager()
            torch.cuda.synchronize(self.device)
            recs = hip.prof_collect_ex()
        finally:
            self._single_chain = False
            hip.prof_enable(False)
        assert len(recs) == len(labels), (len(recs), len(labels))
        out = []
        for l, r in zip(labels, recs):
            if not r[3]:                            # no bytes from the library: the batched parameter-sized passes are priced from the descriptor table
                by = self.param_pass_bytes(l)
                if by:
                    r = r[:3] + (by,) + r[4:]
            out.append((l,) + r)
        return out

    def capture(self):
        """Capture the step (training) or prep+forward (inference) into a hipGraph."""
        torch.cuda.synchronize(self.device)
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):                     # warm-up launch outside capture (module load etc.)
            if self.mode == "training":
                saved = self.save_train_state()
                self.step_eager()
                self.restore_train_state(saved)
            else:
                self.run_prep(); self.run_forward()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        gr = torch.cuda.CUDAGraph()
        # hip.capture_lock: feeder threads (pinned allocations, augmentation kernels) stay out of HIP while the stream is capturing
        with hip.capture_lock, _no_gc():               # see _no_gc: nothing may be destroyed while the stream is capturing
            with torch.cuda.graph(gr):
                if self.mode == "training":
                    self.step_eager()
                else:
                    self.run_prep(); self.run_forward()
        self._graphs = gr
        if self.mode == "training" and getattr(self, "wgrad_stream", None) is not None and not self._single_chain_always:
            self._verify_forked_graph(gr)
        return self._graphs

    def _verification_batch(self):
        """Context manager: the verification of a forked graph must not run on an all-zero batch (nothing would distinguish a node that
        read stale operands; zero targets make rel_loss 0/0 and NaN never equals NaN).  When the engine's input or target buffers are still
        all zero -- capture before the first load_batch -- they hold a fixed pseudo-random batch for the check and get their bytes back after."""
        eng = self

        class _Ctx(object):
            def __enter__(self):
                self.saved = []
                g = torch.Generator(device=eng.device)
                g.manual_seed(20240607)
                img = eng.in_images_u8 if eng.input_u8 else eng.in_images
                if img is not None and not bool(img.any()):
                    self.saved.append((img, img.clone()))
                    if img.dtype == torch.uint8:
                        img.copy_(torch.randint(0, 256, img.shape, generator=g, device=eng.device, dtype=torch.uint8))
                    else:
                        img.copy_((torch.rand(img.shape, generator=g, device=eng.device) * 255.0 - 110.0).to(img.dtype))
                for name in ("gt_loc", "gt_ori", "gt_k3"):
                    t = getattr(eng, name, None)
                    if t is not None and t.numel() and not bool(t.any()):
                        self.saved.append((t, t.clone()))
                        r = torch.rand(t.shape, generator=g, device=eng.device) + 0.1
                        t.copy_((r / r.sum(-1, keepdim=True)).to(t.dtype))     # rows that are valid soft labels and non-zero regression targets alike
                return self

            def __exit__(self, *exc):
                for t, old in self.saved:
                    t.copy_(old)
        return _Ctx()

    def _verify_forked_graph(self, gr, replays=None):
        """The captured graph with its second branch against the same launches issued eagerly on one chain: `replays` steps each from the same
        state (URSO_FORK_VERIFY_REPLAYS, default 4; the second replay is the first that shows a node run early: the first reads what the
        warm-up step left), compared bit for bit on EVERYTHING a step writes: weights, gradients, momentum / Adam moments, BN statistics.
        A graph executor that fails this (see _fork_weight_gradients) costs the engine its side branch, not its results.
        Cost: 2 x replays training steps and two clones of the training state per capture (and per re-capture: set_input_u8 toggles,
        set_trainable).  It runs again wherever verify_fork() is called (UrsoNet.train: once per epoch; tests/test_model_gpu.py holds a
        200-replay stress test); bench.py repeats the comparison with two fresh engines before it times anything."""
        n = int(replays if replays is not None else os.environ.get("URSO_FORK_VERIFY_REPLAYS", "4"))
        with self._verification_batch():
            saved = self.save_train_state()
            for _ in range(n):
                gr.replay()
            torch.cuda.synchronize(self.device)
            got = [t.clone() for t in (self.flat_w, self.flat_g, self.flat_v, self.flat_stats)]
            self.restore_train_state(saved)
            self._single_chain = True
            try:
                for _ in range(n):
                    self.step_eager()
            finally:
                self._single_chain = False
            torch.cuda.synchronize(self.device)
            ok = all(torch.equal(a, b) for a, b in zip(got, (self.flat_w, self.flat_g, self.flat_v, self.flat_stats)))
            ok = ok and bool(torch.isfinite(self.flat_g).all())       # (NaN never equals NaN: a non-finite check batch proves nothing either way)
            self.restore_train_state(saved)
        self.fork_checks = getattr(self, "fork_checks", 0) + 1
        if not ok:
            import warnings
            warnings.warn("ursonet_amd: the captured training graph with the weight gradients on a second branch does not reproduce the single chain "
                          "on this device / runtime -- captured again on one chain")
            self._single_chain_always = True
            self._graphs = None
            self.capture()
        return ok

    def verify_fork(self, replays=None):
        """Re-run the forked graph's check against the single chain on the batch that is loaded NOW (training state is restored afterwards).
        True = the step runs forked and is verified; False = it runs (or from now on runs) on one chain."""
        if self.mode != "training" or not self.forked:
            return False
        if self._graphs is None:
            self.capture()                         # (verifies)
            return self.forked
        self._verify_forked_graph(self._graphs, replays)
        return self.forked

    @property
    def forked(self):
        """True when the captured step runs weight gradients on a second graph branch (_fork_weight_gradients) and that graph has been verified."""
        return getattr(self, "wgrad_stream", None) is not None and not getattr(self, "_single_chain_always", False)

    def step(self):
        """Replay the captured training step (captures on first use)."""
        if self._graphs is None:
            self.capture()
        self._graphs.replay()
        if self._cleared_once and os.environ.get("URSO_CHECK_CLEARED", "0") == "1":
            bad = self.check_cleared_once()
            assert not bad, "gradient buffers cleared at plan time hold values off their sampled grid: %s" % bad

    def forward(self):
        if self.mode == "training":
            self.run_prep(); self.run_forward()
        else:
            if self._graphs is None:
                self.capture()
            self._graphs.replay()

    # ------------------------------------------------------------------ I/O helpers
    def load_batch(self, images, gt_loc=None, gt_ori=None, gt_k3=None):
        """images: float32 [B,H,W,3] already molded (mold_image, net.py:1337-1348) -- numpy or torch."""
        self.in_images.copy_(torch.as_tensor(images, dtype=torch.float32).reshape(self.in_images.shape), non_blocking=True)
        if gt_loc is not None:
            self.gt_loc.copy_(torch.as_tensor(gt_loc, dtype=torch.float32).reshape(self.gt_loc.shape), non_blocking=True)
        if gt_ori is not None:
            self.gt_ori.copy_(torch.as_tensor(gt_ori, dtype=torch.float32).reshape(self.gt_ori.shape), non_blocking=True)
        if gt_k3 is not None:
            self.gt_k3.copy_(torch.as_tensor(gt_k3, dtype=torch.float32).reshape(self.gt_k3.shape), non_blocking=True)

    def set_input_u8(self, on=True):
        """Switch the first kernel between molded float32 input (load_batch) and raw uint8 input (load_batch_u8)."""
        on = bool(on)
        if on and self.in_images_u8 is None:
            self.in_images_u8 = torch.zeros(self.B, self.H, self.W, 3, dtype=torch.uint8, device=self.device)
        if on != self.input_u8:
            self.input_u8 = on
            self._graphs = None                    # the captured graph holds the other buffer's address

    def load_batch_u8(self, images_u8, gt_loc=None, gt_ori=None, gt_k3=None):
        """Raw uint8 frames [B,H,W,3] (host array or device tensor, already resized / padded to the model size) + targets;
        device tensors are copied device-to-device on the current stream."""
        self.set_input_u8(True)
        self.in_images_u8.copy_(torch.as_tensor(images_u8).reshape(self.in_images_u8.shape), non_blocking=True)
        for buf, src in ((getattr(self, "gt_loc", None), gt_loc), (getattr(self, "gt_ori", None), gt_ori), (getattr(self, "gt_k3", None), gt_k3)):
            if src is not None and buf is not None:
                buf.copy_(torch.as_tensor(src).reshape(buf.shape), non_blocking=True)

    def outputs(self):
        """(loc [B, n_loc], ori [B, n_ori]) as fp32 device tensors (raw network outputs, net.py:1254-1258)."""
        g = self.graph
        nl = g.outputs["loc"].c
        loc = self.out_loc.data.view(self.B, -1)[:, :nl]
        if self.out_ori is None:             # keypoint mode: (k1, [k2, k3]) -- net.py:678
            return loc, [self.acts[g.outputs[k].id].data.view(self.B, -1)[:, :3] for k in ("k2", "k3")]
        no = g.outputs["ori"].c
        ori = self.q_out if self.quat_head else self.out_ori.data.view(self.B, -1)[:, :no]
        return loc, ori

    def losses(self):
        """{'loc_loss','ori_loss'} (keypoint mode: {'loc_loss','k2_loss','k3_loss'}) weighted by LOSS_WEIGHTS, as in the
        Keras metrics (net.py:989-992, 1019-1028)."""
        l = self.loss_buf.detach().cpu().numpy()
        if self.config.REGRESS_KEYPOINTS:
            return {"loc_loss": float(l[0]), "k2_loss": float(l[2]), "k3_loss": float(l[3])}
        return {"loc_loss": float(l[0]), "ori_loss": float(l[1])}

    def evaluate(self, read=True):
        """Forward + losses on the loaded batch WITHOUT touching any training state: what Keras' validation pass does
        (learning_phase 0, net.py:1155-1157).  In batch-statistics BN mode the moving statistics are restored afterwards (the
        forward kernels of the training plan update them) -- the normalisation itself still uses batch statistics there, which
        is the one deviation from Keras' moving-statistics evaluation (documented in DESIGN.md section 11)."""
        stats = self.flat_stats.clone() if self.train_bn else None
        self.run_prep(); self.run_forward()
        for op in self.loss_pre_ops + self.loss_ops:
            op()
        if stats is not None:
            self.flat_stats.copy_(stats)
        return self.losses() if read else None     # read=False: the scalars stay in loss_buf (UrsoNet.train copies them on the device, no sync)

    def set_lr(self, lr):
        self.hyper[0] = float(lr)

    def save_train_state(self):
        """Weights + optimizer state (incl. Adam's device-side step counter), for the warm-up step before graph capture."""
        st = [self.flat_w.clone(), self.flat_v.clone(), self.flat_stats.clone()]
        if getattr(self, "adam", False):
            st += [self.flat_v2.clone(), self.flat_vhat.clone(), self.hyper.clone()]
        return st

    def restore_train_state(self, st):
        self.flat_w.copy_(st[0]); self.flat_v.copy_(st[1]); self.flat_stats.copy_(st[2])
        if getattr(self, "adam", False):
            self.flat_v2.copy_(st[3]); self.flat_vhat.copy_(st[4]); self.hyper.copy_(st[5])

    def reset_optimizer(self):
        self.flat_v.zero_()
        if getattr(self, "adam", False):
            self.flat_v2.zero_(); self.flat_vhat.zero_(); self.hyper[5] = 0.0

    def flops(self):
        return conv_flops(self.graph, self.B)


def initial_weights(graph, seed=1234, randomize_bn=False):
    """Keras default initialisers (glorot_uniform kernels, zero biases, BN gamma=1 beta=0 mean=0 var=1;
    used by `--weights none`, pose_estimator.py:897).  `randomize_bn` perturbs BN tensors and biases."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for ln, ws in graph.params.items():
        p = OrderedDict()
        for wn, shape in ws.items():
            if wn == "kernel":
                if len(shape) == 4:
                    rf = shape[0] * shape[1]
                    fi, fo = rf * shape[2], rf * shape[3]
                else:
                    fi, fo = shape
                lim = math.sqrt(6.0 / (fi + fo))
                p[wn] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif wn == "bias":
                p[wn] = (rng.normal(0, 0.05, size=shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
            elif wn == "gamma":
                p[wn] = (rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)).astype(np.float32)
            elif wn in ("beta", "moving_mean"):
                p[wn] = (rng.normal(0, 0.1, size=shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
            else:
                p[wn] = (rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)).astype(np.float32)
        out[ln] = p
    return out
