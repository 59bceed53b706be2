"""Whole-path parity on the GPU: ursonet_amd.Engine (HIP kernels through the C ABI, hipGraph
replay) vs the CPU oracle (oracle/graph_ref.py) on identical weights and inputs.

Tolerances (north_star: outputs within 1e-3 relative fp32):
  fp32 compute : outputs, losses, every parameter gradient and the post-step weights <= 1e-3
                 (relative to the tensor's max |.|);  in practice ~1e-5.
  bf16 compute : activations/weights are rounded to bf16 (8-bit mantissa) at every layer, so
                 outputs are compared at 5e-2 and gradients by cosine similarity (>= 0.95 per tensor,
                 >= 0.99 over all parameters); the bf16 KERNELS are checked tightly in test_kernels_gpu.py.
"""
import os

import numpy as np
import pytest
import torch

from util import make_config, synthetic_batch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.isfinite(a).all()
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _cos(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


class ReluDecisions(object):
    """ReLU is discontinuous: a pre-activation within float rounding of 0 may be kept on one
    implementation and dropped on the other, which changes that element's whole gradient path.
    This hook (oracle.graph_ref.relu) forces the oracle to take the DEVICE's decision, but only
    after asserting that the two decisions differ exclusively where the oracle's own
    pre-activation is within `tol` (relative to the tensor's max) of zero."""

    def __init__(self, eng, tol):
        self.eng, self.tol = eng, tol
        self.flips, self.total = 0, 0
        self.worst, self.mags = 0.0, []          # largest |pre-activation| / max at a flipped decision; all of them (for the histogram kept in profiles/)

    def __call__(self, site, x):
        c = self.eng.convs[site]
        n = c.node
        B = x.shape[0]
        if getattr(c.dst, "fwd_sampled", False):
            # a block output the engine computes only at the pixels its stride-2 consumers read (Engine._sample_block_output): the device's
            # decisions exist on the even grid; elsewhere nothing depends on the tensor and the oracle keeps its own
            m = (x.detach() > 0).clone()
            sub = c.dst.data_compact.float().cpu().view(B, n.dst.h // 2, n.dst.w // 2, c.npad)[..., :n.cout] > 0
            m[:, :, ::2, ::2] = sub.permute(0, 3, 1, 2)
            diff = m != (x.detach() > 0)
            nd = int(diff.sum())
            if nd:
                rel = (x.detach().abs()[diff] / (x.detach().abs().max() + 1e-30)).double()
                worst = float(rel.max()); self.worst = max(self.worst, worst); self.mags.append(rel.cpu())
                assert worst < self.tol, "ReLU decision differs at |pre-activation| = %.2e of max in %s" % (worst, site)
            self.flips += nd
            self.total += diff.numel() // 4
            return m
        if getattr(c.dst, "fused_pool", False):
            # conv1 inside urso_stem_conv_pool: its output exists only pooled, but every decision anything depends on is in the pool's arg-max
            # bytes -- byte = 3 ky + kx of the window's first maximum, + 16 when that maximum is <= 0 (conv_stem.hip) -- i.e. per pooled
            # pixel WHICH conv output the device forwarded and what its ReLU decided.  Those decisions are compared and forced like any
            # other layer's; a conv output that wins no window feeds nothing, forward or backward, and keeps the oracle's own decision.
            pool = [nn for nn in self.eng.graph.nodes if nn.op == "pool" and nn.src.id == n.dst.id][0]
            am = pool._am.cpu().view(B, pool.dst.h, pool.dst.w, pool.dst.c).permute(0, 3, 1, 2).long()
            pos = am < 16
            k = am & 15
            assert int(k.max()) <= 8
            ii = torch.arange(pool.dst.h).view(1, 1, -1, 1) * 2 + k // 3          # 'same' on an even grid pads below / right only: window rows 2i .. 2i+2
            jj = torch.arange(pool.dst.w).view(1, 1, 1, -1) * 2 + k % 3
            assert int(ii.max()) < n.dst.h and int(jj.max()) < n.dst.w, "an arg-max points into the padding"
            xd = x.detach()
            m = (xd > 0).clone()
            bb = torch.arange(B).view(-1, 1, 1, 1).expand_as(am)
            cc = torch.arange(pool.dst.c).view(1, -1, 1, 1).expand_as(am)
            m[bb, cc, ii, jj] = pos
            diff = m != (xd > 0)
            nd = int(diff.sum())
            if nd:
                rel = (xd.abs()[diff] / (xd.abs().max() + 1e-30)).double()
                worst = float(rel.max()); self.worst = max(self.worst, worst); self.mags.append(rel.cpu())
                assert worst < self.tol, "ReLU decision differs at |pre-activation| = %.2e of max in %s (from the pool's arg-max bytes)" % (worst, site)
            self.flips += nd
            self.total += am.numel()
            return m
        dev_act = c.dst.data.float().cpu().view(B, -1)[:, :n.dst.h * n.dst.w * c.npad]
        if x.dim() == 4:
            m = (dev_act.view(B, n.dst.h, n.dst.w, c.npad)[..., :n.cout] > 0).permute(0, 3, 1, 2)
            if getattr(c.dst, "fwd_scattered", False):        # only the even pixels of this tensor are computed (Engine._sample_layer_below)
                keep = torch.zeros_like(m); keep[:, :, ::2, ::2] = True
                m = torch.where(keep, m, x.detach() > 0)
        else:
            m = dev_act.view(B, c.npad)[:, :n.cout] > 0
        diff = m != (x.detach() > 0)
        nd = int(diff.sum())
        if nd:
            rel = (x.detach().abs()[diff] / (x.detach().abs().max() + 1e-30)).double()
            worst = float(rel.max()); self.worst = max(self.worst, worst); self.mags.append(rel.cpu())
            assert worst < self.tol, "ReLU decision differs at |pre-activation| = %.2e of max in %s" % (worst, site)
        self.flips += nd
        self.total += diff.numel()
        return m

    def histogram(self):
        """Counts of flipped decisions by |pre-activation| / tensor max: [< 1e-3, < 3e-3, < 1e-2, < 3e-2, < 1e-1, >= 1e-1]."""
        if not self.mags:
            return [0] * 6
        v = torch.cat(self.mags)
        edges = [0.0, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, float("inf")]
        return [int(((v >= lo) & (v < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]


def _oracle_step(cfg, weights, img, loc, ori, lr, relu_hook=None, q=None, layer_regex=".*"):
    from oracle import graph_ref as G
    P = G.to_torch(weights)
    vel = {}
    t_ori = tuple(torch.tensor(o) for o in ori) if isinstance(ori, (tuple, list)) else torch.tensor(ori)     # keypoint mode: (k2, k3)
    out = G.train_step(P, vel, torch.tensor(img), torch.tensor(loc), t_ori, cfg, lr, layer_regex=layer_regex, relu_hook=relu_hook, q=q)
    newW = {ln: {wn: w.detach().numpy() for wn, w in ws.items()} for ln, ws in P.items()}
    return out, newW


def _run_engine(cfg, img, loc, ori, seed=3, use_graph=True):
    from ursonet_amd.engine import Engine
    eng = Engine(cfg, "training", seed=seed, randomize_bn=True)
    w0 = eng.get_weights()
    eng.load_batch(img, loc, ori)
    if use_graph:
        eng.step()
    else:
        eng.step_eager()
    torch.cuda.synchronize()
    return eng, w0


CASES = [
    ("cfg1_r18_quat", dict(backbone="resnet18", h=128, w=128, batch=2, regress_ori=True)),
    ("r50_softclass", dict(backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8)),
    ("r34_euler", dict(backbone="resnet34", h=64, w=128, batch=3, regress_ori=True, ori_param="euler_angles")),
    ("r50_classify_loc", dict(backbone="resnet50", h=64, w=128, batch=2, regress_ori=False, regress_loc=False, ori_bins=4, loc_bins=4)),
    ("r101_softclass", dict(backbone="resnet101", h=64, w=128, batch=2, regress_ori=False, ori_bins=8)),      # cfg4's trunk, one TRAINING step
    ("r50_no_dense_layer", dict(backbone="resnet50", h=64, w=128, batch=2, regress_ori=False, ori_bins=4, nr_dense=0)),     # NR_DENSE_LAYERS = 0 (net.py:295): the heads' final Dense reads the flattened bottleneck
    ("r18_two_dense_layers", dict(backbone="resnet18", h=128, w=128, batch=2, regress_ori=True, nr_dense=2)),              # NR_DENSE_LAYERS = 2: two Dense + ReLU per branch
    # one step at the real cfg2 WIDTH (512 x 640, ori_resolution 16): the row / tile geometry of every cfg2 layer (160-row tiles of the
    # big-tile pointwise kernel, 641-wide virtual rows of the halo kernel, ...) sits inside an oracle comparison, not only a property check
    ("cfg2_width_r50_n16", dict(backbone="resnet50", h=512, w=640, batch=2, regress_ori=False, ori_bins=16)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_training_step_parity_fp32(name, kw):
    cfg = make_config(dtype="float32", **kw)
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=1)
    eng, w0 = _run_engine(cfg, img, loc, ori)
    # (1) forward parity against the untouched oracle
    ref0, _ = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE)
    gl, go = eng.outputs()
    assert _rel(gl.cpu().numpy(), ref0["loc"].numpy()) < 1e-3
    assert _rel(go.cpu().numpy(), ref0["ori"].numpy()) < 1e-3
    # (2) gradient parity under identical ReLU decisions (decisions may only differ at ~0)
    dec = ReluDecisions(eng, tol=1e-5)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec)
    assert dec.flips <= max(4, 2e-6 * dec.total), "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    ls = eng.losses()
    assert abs(ls["loc_loss"] - ref["loc_loss"]) < 1e-3 * abs(ref["loc_loss"]) + 1e-6
    assert abs(ls["ori_loss"] - ref["ori_loss"]) < 1e-3 * abs(ref["ori_loss"]) + 1e-6
    grads = eng.get_grads()
    worst = ("", 0.0)
    for ln, ws in ref["grads"].items():
        for wn, gref in ws.items():
            e = _rel(grads[ln][wn], gref.numpy())
            if e > worst[1]:
                worst = (ln + "/" + wn, e)
    assert worst[1] < 1e-3, "worst gradient mismatch %s: %.3e" % worst
    assert abs(float(eng.normsq.cpu()) ** 0.5 - ref["grad_norm"]) < 1e-3 * ref["grad_norm"]
    w1 = eng.get_weights()
    for ln, ws in newW.items():
        for wn, wref in ws.items():
            assert _rel(w1[ln][wn], wref) < 1e-4, (ln, wn)


def test_training_step_parity_fp32_multitile_stream():
    """The same whole-step comparison with every persistent conv kernel capped to 8 blocks (urso_set_option grid_cap): each
    block walks 3-24 tiles, so the cross-tile prefetch path of the fp32 igemm_kernel runs inside an oracle-compared step."""
    import ursonet_amd.hip as hip
    with hip.options(grid_cap=8):
        test_training_step_parity_fp32(*CASES[1])


def _compare_step(eng, ref, newW, tol_out, tol_g, tol_w, tol_l2=None, tol_norm=None, check=True):
    """Outputs, losses, every gradient tensor (relative to its max; the big ones also in the Euclidean norm), global norm, post-step
    weights; one assertion that reports all the measured errors."""
    tol_l2 = tol_g if tol_l2 is None else tol_l2
    tol_norm = tol_g if tol_norm is None else tol_norm
    gl, go = eng.outputs()
    ls = eng.losses()
    grads = eng.get_grads()
    worst, worst_big, worst_l2 = ("", 0.0), ("", 0.0), ("", 0.0)
    for ln, ws in ref["grads"].items():
        for wn, gref in ws.items():
            e = _rel(grads[ln][wn], gref.numpy())
            if e > worst[1]:
                worst = (ln + "/" + wn, e)
            if gref.numel() >= 4096 and e > worst_big[1]:
                worst_big = (ln + "/" + wn, e)
            if gref.numel() >= 4096:          # the same tensors in the Euclidean norm: what a mis-scaled or mis-indexed layer moves, whatever single elements do
                e2 = _rel_l2(grads[ln][wn], gref.numpy())
                if e2 > worst_l2[1]:
                    worst_l2 = (ln + "/" + wn, e2)
    w1 = eng.get_weights()
    worst_w = max((_rel(w1[ln][wn], wref), ln + "/" + wn) for ln, ws in newW.items() for wn, wref in ws.items())
    m = {"loc": _rel(gl.cpu().numpy(), ref["loc"].numpy()), "ori": _rel(go.cpu().numpy(), ref["ori"].numpy()),
         "loc_loss": abs(ls["loc_loss"] - ref["loc_loss"]) / (abs(ref["loc_loss"]) + 1e-6),
         "ori_loss": abs(ls["ori_loss"] - ref["ori_loss"]) / (abs(ref["ori_loss"]) + 1e-6),
         "grad": worst[1], "grad_big": worst_big[1], "grad_l2": worst_l2[1], "grad_norm": abs(float(eng.normsq.cpu()) ** 0.5 - ref["grad_norm"]) / ref["grad_norm"], "weights": worst_w[0]}
    print("parity:", {k: "%.2e" % v for k, v in m.items()}, "worst grad", worst[0], "worst big grad", worst_big[0], "worst weight", worst_w[1])
    # tol_g applies to every tensor with >= 4096 elements (filters, dense kernels); the small per-channel tensors (BN gamma/beta, biases:
    # sums with cancellation over 64-2048 channels) get 3 tol_g
    if not check:
        return m
    ok = (m["loc"] < tol_out and m["ori"] < tol_out and m["loc_loss"] < tol_out and m["ori_loss"] < tol_out and
          m["grad_big"] < tol_g and m["grad"] < 3 * tol_g and m["grad_l2"] < tol_l2 and m["grad_norm"] < tol_norm and m["weights"] < tol_w)
    assert ok, "tolerances out %.0e grad %.0e weights %.0e exceeded: %s (worst gradient %s, worst weight %s)" % (
        tol_out, tol_g, tol_w, {k: "%.2e" % v for k, v in m.items()}, worst[0], worst_w[1])
    return worst


@pytest.mark.parametrize("pair", [1, 0], ids=["fused", "apart"])
@pytest.mark.parametrize("cap", [0, 8], ids=["grid", "capped"])
@pytest.mark.parametrize("dtype,tol_out,tol_g", [("bfloat16", 2.5e-2, 1.2e-1), ("float16", 4e-3, 1.6e-2)])
def test_training_step_parity_16bit_same_rounding_points(dtype, tol_out, tol_g, cap, pair):
    """The benchmarked dtype against an oracle that rounds where the device rounds (oracle.graph_ref.StorageRounding: folded
    filters, every stored activation, every activation gradient): outputs, losses, EVERY parameter gradient (relative to the
    tensor's max, the filters and dense kernels also in the Euclidean norm), the global norm and the post-step weights -- not a cosine.
    'capped' additionally forces the multi-tile stream of the DMA conv kernels (conv_pw.hip) and of the fused pair kernel inside this
    oracle-compared step; 'fused' / 'apart' run the plan with and without the fused stage-2/3 pointwise pairs (conv_pair.hip).

    WHERE THE GATES COME FROM (round 4: tools/probes/parity_hist.py, profiles/r04_parity.txt).  What remains between device and oracle is
    the order of the fp32 accumulations (a value within ~1e-6 of a 16-bit rounding boundary rounds the other way, ~3e-4 of all elements) and
    the device's second rounding where two gradient contributions meet in a 16-bit buffer.  Through 50 layers at 2 x 128 x 192 that residue
    is CHAOTIC: the worst of the ~110 filter / dense-kernel gradient tensors, over six data seeds x four kernel plans (fused / apart pairs,
    conv_halo2.hip on / off), measured in bf16 between 1.6e-2 and 7.8e-2 of the tensor's max (4.8e-2 ... 7.2e-2 on at least one seed of
    EVERY plan, the round-3 kernels included: the round-3 gate of 4e-2 passed on the one seed it was run on), 1.2e-2 ... 6.7e-2 in the
    Euclidean norm; outputs <= 1.64e-2, global norm <= 1.44e-2.  The same 24 runs in fp16 -- 8x finer, the same kernels: a mis-scaled or
    mis-indexed layer shows at the same absolute size in both -- measure 1.9e-3 ... 1.1e-2 (max), <= 8.3e-3 (Euclidean), outputs
    <= 1.95e-3, norm <= 1.76e-3: the residue scales with the rounding unit, i.e. it IS rounding.  Gates = ~1.5x the largest value of the
    histogram; fp16's are the ones a systematic error has to get past (a wrong scale of 2 % in one filter tensor fails them)."""
    import ursonet_amd.hip as hip
    from oracle import graph_ref as G
    kw = dict(backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8)
    cfg = make_config(dtype=dtype, **kw)
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=1)
    with hip.options(grid_cap=cap, pair=pair):
        eng, w0 = _run_engine(cfg, img, loc, ori)
    assert len(eng.pair_first) == (5 if pair else 0)          # res2{b,c}, res3{b,c,d}: branch2a fused behind the previous block's branch2c
    q = G.StorageRounding(torch.bfloat16 if dtype == "bfloat16" else torch.float16, unstored=getattr(eng, "shortcut_folded", ()))
    # decisions may differ only where the oracle's own pre-activation is this close to zero (relative to the tensor's max): measured over the 24
    # runs above, bf16 1.4e4 flips < 1e-3, 1.5e3 < 1e-2, one < 3e-2, none beyond; fp16 all < 1.1e-3 (profiles/r04_parity.txt)
    dec = ReluDecisions(eng, tol=4e-2 if dtype == "bfloat16" else 4e-3)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
    assert dec.flips <= 2e-3 * dec.total, "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    _compare_step(eng, ref, newW, tol_out, tol_g, 1e-3, tol_l2=0.85 * tol_g, tol_norm=tol_out)


# Single-seed maximum of the worst filter-gradient tensor at 2 x 128 x 192 (stage-4 maps of 8 x 12 pixels: one ReLU decision that falls the other
# way moves a filter gradient by percent).  Round 5 measured 1.6e-2 ... 7.8e-2 over the five seeds with an oracle that rounded the stage-2 projection
# shortcut to bf16 although the device never stores it; with the faithful rounding model (StorageRounding(unstored=eng.shortcut_folded), round 6)
# the same device results measure 1.6e-2, 1.40e-1 (seed 2: res4e_branch2a/kernel), 2.5e-2, 1.7e-2, 3.3e-2 -- the residue moved, the device did not
# (gpurun call 15: identical numbers with the round-6 plan rewrites on and off, which do not apply at this size).  The gate sits above that
# maximum; what catches a systematic error is the MEDIAN gate below and the teacher-forced per-layer tests (tests/test_layerwise_gpu.py).
SINGLE_SEED_GRAD_GATE = float(__import__("os").environ.get("URSO_SINGLE_SEED_GATE", "1.7e-1"))


def test_training_step_parity_bf16_five_seeds_two_sided_gate():
    """The bf16 whole-step comparison on FIVE data seeds with a two-sided gate.  The single-seed gate above has to
    sit above the chaotic residue's maximum (1.2e-1 of the worst tensor's max), which a uniform 2 % scale error would pass.  Over seeds the
    residue's MEDIAN is small -- profiles/r04_parity.txt: worst filter-gradient tensor 1.6e-2 ... 7.8e-2, median ~3e-2 -- while a systematic
    error moves every seed: median of the worst-tensor error <= 4e-2 (max norm) and <= 3e-2 (Euclidean), median output error <= 1.5e-2,
    and the single-seed maxima as before.  Measured (profiles/r05_parity.txt): worst-tensor 1.6e-2, 2.2e-2, 7.8e-2, 1.6e-2, 2.6e-2 -> median
    2.2e-2; Euclidean median 1.7e-2; outputs median 1.0e-2."""
    from oracle import graph_ref as G
    kw = dict(backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8)
    cfg = make_config(dtype="bfloat16", **kw)
    ms = []
    for seed in (1, 2, 3, 4, 5):
        img, loc, ori, _ = synthetic_batch(cfg, 2, seed=seed)
        eng, w0 = _run_engine(cfg, img, loc, ori)
        q = G.StorageRounding(torch.bfloat16, unstored=getattr(eng, "shortcut_folded", ()))
        dec = ReluDecisions(eng, tol=4e-2)
        ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
        assert dec.flips <= 2e-3 * dec.total
        ms.append(_compare_step(eng, ref, newW, 0, 0, 0, check=False))
        if __import__("os").environ.get("URSO_PARITY_LOG"):
            with open(__import__("os").environ["URSO_PARITY_LOG"], "a") as f:
                f.write("bf16 2x128x192 seed %d: %s\n" % (seed, {k: "%.2e" % v for k, v in ms[-1].items()}))
        _compare_step(eng, ref, newW, 2.5e-2, SINGLE_SEED_GRAD_GATE, 1e-3, tol_l2=0.85 * SINGLE_SEED_GRAD_GATE, tol_norm=2.5e-2)          # every seed: the single-seed maxima
        del eng
    med = {k: float(np.median([m[k] for m in ms])) for k in ms[0]}
    print("median over seeds:", {k: "%.2e" % v for k, v in med.items()})
    log = __import__("os").environ.get("URSO_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write("bf16 2x128x192 five seeds: median %s  max %s\n" % ({k: "%.2e" % v for k, v in med.items()},
                                                                         {k: "%.2e" % max(m[k] for m in ms) for k in ms[0]}))
    assert med["grad_big"] <= 4e-2 and med["grad_l2"] <= 3e-2 and med["loc"] <= 1.5e-2 and med["ori"] <= 1.5e-2 and med["grad_norm"] <= 1e-2, med


def test_training_step_parity_bf16_full_benchmark_batch():
    """ONE oracle-compared training step of the benchmark workload itself: cfg2 at batch 32 x 512 x 640, bf16 (the
    other oracle comparisons at this width run batch 2).  About a minute of oracle time on the GPU box's host cores."""
    from oracle import graph_ref as G
    import os
    torch.set_num_threads(min(os.cpu_count() or 8, 128))
    cfg = make_config(dtype="bfloat16", backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16)
    img, loc, ori, _ = synthetic_batch(cfg, 32, seed=1)
    eng, w0 = _run_engine(cfg, img, loc, ori)
    q = G.StorageRounding(torch.bfloat16, unstored=getattr(eng, "shortcut_folded", ()))
    dec = ReluDecisions(eng, tol=2e-2)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
    assert dec.flips <= 2e-3 * dec.total, "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    print("relu flips:", dec.flips, "of", dec.total, "worst %.2e" % dec.worst, dec.histogram())
    m = _compare_step(eng, ref, newW, 2e-2, 4e-2, 1e-3, tol_l2=2.5e-2, tol_norm=1e-2, check=False)
    log = os.environ.get("URSO_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write("bf16 cfg2 FULL batch 32x512x640: %s; relu flips %d of %d, worst %.2e %s\n"
                    % ({k: "%.2e" % v for k, v in m.items()}, dec.flips, dec.total, dec.worst, dec.histogram()))
    _compare_step(eng, ref, newW, 2e-2, 4e-2, 1e-3, tol_l2=2.5e-2, tol_norm=1e-2)


@pytest.mark.parametrize("case", ["cfg4_r101_n24_b16_bf16", "cfg5_r50_f16_classify_loc_b32"])
def test_training_step_parity_full_size_cfg4_cfg5(case):
    """ONE oracle-compared training step of BASELINE.json configs[3] and configs[4] at their real per-GPU sizes (ResNet-101 / 13,824 bins / batch
    16 x 512 x 640 / bf16; ResNet-50 / fp16 / batch 32 x 640 x 960 / classification location head): outputs, losses, every gradient, the norm,
    the post-step weights against the rounding-aware oracle.  2-4 minutes of oracle time each on the GPU box's host cores, so they run on request
    only (URSO_FULL_SIZE_ORACLE=1; the numbers of this round's run are in profiles/r05_parity.txt); the default suite holds the same comparison
    for cfg2 (test_training_step_parity_bf16_full_benchmark_batch) and these configurations' property tests."""
    import os
    if os.environ.get("URSO_FULL_SIZE_ORACLE", "0") != "1":
        pytest.skip("set URSO_FULL_SIZE_ORACLE=1 (minutes of host time per case)")
    from oracle import graph_ref as G
    torch.set_num_threads(min(os.cpu_count() or 8, 128))
    if case.startswith("cfg4"):
        cfg = make_config(backbone="resnet101", h=512, w=640, batch=16, regress_ori=False, ori_bins=24, dtype="bfloat16")
        qt, tol_out, tol_g, tol_l2, tol_norm, dtol = torch.bfloat16, 2.5e-2, 6e-2, 4e-2, 1.5e-2, 2e-2
    else:
        cfg = make_config(backbone="resnet50", h=640, w=960, batch=32, regress_ori=False, regress_loc=False, ori_bins=16, loc_bins=16,
                          dtype="float16", f16=True)
        qt, tol_out, tol_g, tol_l2, tol_norm, dtol = torch.float16, 4e-3, 1.6e-2, 1e-2, 4e-3, 4e-3
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=15)
    eng, w0 = _run_engine(cfg, img, loc, ori)
    q = G.StorageRounding(qt, unstored=getattr(eng, "shortcut_folded", ()))
    dec = ReluDecisions(eng, tol=dtol)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
    m = _compare_step(eng, ref, newW, tol_out, tol_g, 1e-3, tol_l2=tol_l2, tol_norm=tol_norm, check=False)
    log = os.environ.get("URSO_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write("%s FULL size: %s; relu flips %d of %d, worst %.2e %s\n" % (case, {k: "%.2e" % v for k, v in m.items()}, dec.flips, dec.total, dec.worst, dec.histogram()))
    assert dec.flips <= 2e-3 * dec.total, "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    _compare_step(eng, ref, newW, tol_out, tol_g, 1e-3, tol_l2=tol_l2, tol_norm=tol_norm)


@pytest.mark.parametrize("case", ["cfg4_r101_n24_512x640_bf16", "cfg5_r50_f16_classify_loc_640x960"])
def test_training_step_parity_at_cfg4_cfg5_geometry(case):
    """BASELINE.json configs[3] and configs[4] at their REAL per-layer geometry in the default suite (VERDICT r05 missing 4): ResNet-101 /
    13,824 orientation bins / 512 x 640 / bf16, and ResNet-50 / fp16 / 640 x 960 (SPEED 1200 x 1920 at image_scale 0.5) / classification
    location head with 16^3 bins -- every kernel those plans select, its tile walk over 640- / 960-pixel rows, the 13,824-bin soft-label loss
    held in registers -- inside ONE oracle-compared training step: outputs, losses, every gradient, the norm, the post-step weights.  Batch 2
    instead of 16 / 32 shortens the tile streams only (the full-batch steps stay opt-in: test_training_step_parity_full_size_cfg4_cfg5)."""
    import os
    from oracle import graph_ref as G
    torch.set_num_threads(min(os.cpu_count() or 8, 128))
    if case.startswith("cfg4"):
        cfg = make_config(backbone="resnet101", h=512, w=640, batch=2, regress_ori=False, ori_bins=24, dtype="bfloat16")
        qt, tol_out, tol_g, tol_l2, tol_norm, dtol = torch.bfloat16, 2.5e-2, 6e-2, 4e-2, 1.5e-2, 2e-2
    else:
        cfg = make_config(backbone="resnet50", h=640, w=960, batch=2, regress_ori=False, regress_loc=False, ori_bins=16, loc_bins=16,
                          dtype="float16", f16=True)
        qt, tol_out, tol_g, tol_l2, tol_norm, dtol = torch.float16, 4e-3, 1.6e-2, 1e-2, 4e-3, 4e-3
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=15)
    eng, w0 = _run_engine(cfg, img, loc, ori)
    assert tuple(eng.outputs()[1].shape) == (2, 13824 if case.startswith("cfg4") else 4096)
    q = G.StorageRounding(qt, unstored=getattr(eng, "shortcut_folded", ()))
    dec = ReluDecisions(eng, tol=dtol)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
    m = _compare_step(eng, ref, newW, tol_out, tol_g, 1e-3, tol_l2=tol_l2, tol_norm=tol_norm, check=False)
    txt = "%s batch 2: %s; relu flips %d of %d, worst %.2e %s" % (case, {k: "%.2e" % v for k, v in m.items()}, dec.flips, dec.total, dec.worst, dec.histogram())
    print(txt)
    log = os.environ.get("URSO_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write(txt + "\n")
    assert dec.flips <= 2e-3 * dec.total, "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    _compare_step(eng, ref, newW, tol_out, tol_g, 1e-3, tol_l2=tol_l2, tol_norm=tol_norm)


@pytest.mark.parametrize("pwx", [1, 2], ids=["policy", "pwx_everywhere"])
def test_training_step_parity_bf16_at_cfg2_width(pwx):
    """The benchmarked dtype at the real cfg2 image size (2 x 512 x 640, ori_resolution 16) against the rounding-aware oracle: the
    per-layer geometry of the benchmark (every kernel the cfg2 plan selects, its tile walk over 640-pixel rows) inside an oracle-compared
    step; batch 2 instead of 32 only shortens the tile streams.  pwx = 2 additionally sends every pointwise layer the big-tile kernel
    (conv_pwx.hip) supports through it (fused pairs off so that the stage-2/3 layers reach it too)."""
    import ursonet_amd.hip as hip
    from oracle import graph_ref as G
    cfg = make_config(dtype="bfloat16", backbone="resnet50", h=512, w=640, batch=2, regress_ori=False, ori_bins=16)
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=1)
    with hip.options(pwx=pwx, pair=(1 if pwx == 1 else 0)):
        eng, w0 = _run_engine(cfg, img, loc, ori)
    q = G.StorageRounding(torch.bfloat16, unstored=getattr(eng, "shortcut_folded", ()))
    # ReLU decisions of device and oracle may differ only where the oracle's pre-activation is within 2e-2 of the tensor's max of zero.  Measured
    # at this size (tools/probes/parity_cfg2w.py, profiles/r04_parity.txt; two seeds): 1.2e-3 of the 1.05e8 decisions flip -- 1.0e5 of them below
    # 1e-3 of the max, 2.0e4 below 3e-3, 1.3e3 below 1e-2, NONE above 9.9e-3 -- so the gate sits at twice the largest flip seen (round 3 allowed
    # 8e-2).  Errors measured there: outputs 1.3e-2, worst filter gradient 1.4e-2 (max norm) / 9.6e-3 (Euclidean), global norm 3e-4: at the
    # benchmark's width the gradients average over 25x the pixels of the small case and the 16-bit residue is far from the gates.
    dec = ReluDecisions(eng, tol=2e-2)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
    assert dec.flips <= 2e-3 * dec.total, "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    print("relu flips:", dec.flips, "of", dec.total, "worst %.2e" % dec.worst, dec.histogram())
    _compare_step(eng, ref, newW, 2e-2, 4e-2, 1e-3, tol_l2=2.5e-2, tol_norm=1e-2)


@pytest.mark.parametrize("dtype,tol_out,tol_g", [("bfloat16", 2e-2, 4e-2), ("float16", 4e-3, 7e-3)])
def test_training_step_parity_16bit_register_filter_3x3_everywhere(dtype, tol_out, tol_g):
    """The same oracle comparison with the 128-channel register-filter 3x3 kernel forced onto every layer it can run (option c3 = 3;
    the default policy only takes images its tiles cover to 88 %, which this small test image does not reach), so that its forward and
    masked data-gradient forms, both tile geometries' border handling and the cross-wave reduction sit inside an oracle-compared step."""
    import ursonet_amd.hip as hip
    from oracle import graph_ref as G
    cfg = make_config(dtype=dtype, backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8)
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=1)
    with hip.options(c3=3):
        eng, w0 = _run_engine(cfg, img, loc, ori)
    q = G.StorageRounding(torch.bfloat16 if dtype == "bfloat16" else torch.float16, unstored=getattr(eng, "shortcut_folded", ()))
    dec = ReluDecisions(eng, tol=4 * tol_out)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
    assert dec.flips <= 2e-3 * dec.total, "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    _compare_step(eng, ref, newW, tol_out, tol_g, 1e-3)


def test_keypoint_mode_training_step_parity_fp32():
    """REGRESS_KEYPOINTS (net.py:312-316, 657-659): three Dense(3) heads on the loc trunk, three MSE losses (a13), no
    orientation branch.  One hipGraph step vs the oracle: outputs, the three losses, every gradient, post-step weights."""
    from ursonet_amd.engine import Engine
    cfg = make_config(dtype="float32", backbone="resnet18", h=64, w=128, batch=3, keypoints=True)
    img, loc, _, _ = synthetic_batch(cfg, 3, seed=8)
    rng = np.random.default_rng(4)
    k2, k3 = (loc + rng.normal(0, 0.5, loc.shape)).astype(np.float32), (loc + rng.normal(0, 0.5, loc.shape)).astype(np.float32)
    eng = Engine(cfg, "training", seed=3, randomize_bn=True)
    assert not any(n.startswith("ori_") for n in eng.graph.params)
    w0 = eng.get_weights()
    eng.load_batch(img, loc, k2, k3)
    eng.step(); torch.cuda.synchronize()
    dec = ReluDecisions(eng, tol=1e-5)
    ref, newW = _oracle_step(cfg, w0, img, loc, (k2, k3), cfg.LEARNING_RATE, relu_hook=dec)
    k1d, (k2d, k3d) = eng.outputs()
    assert _rel(k1d.cpu().numpy(), ref["loc"].numpy()) < 1e-3
    assert _rel(k2d.cpu().numpy(), ref["k2"].numpy()) < 1e-3 and _rel(k3d.cpu().numpy(), ref["k3"].numpy()) < 1e-3
    ls = eng.losses()
    for k in ("loc_loss", "k2_loss", "k3_loss"):
        assert abs(ls[k] - ref[k]) < 1e-3 * abs(ref[k]) + 1e-6, k
    grads = eng.get_grads()
    for ln, ws in ref["grads"].items():
        for wn, gref in ws.items():
            assert _rel(grads[ln][wn], gref.numpy()) < 1e-3, (ln, wn)
    w1 = eng.get_weights()
    for ln, ws in newW.items():
        for wn, wref in ws.items():
            assert _rel(w1[ln][wn], wref) < 1e-4, (ln, wn)


def test_set_trainable_preset_gradients_match_oracle():
    """set_trainable('4+') (net.py:1030-1066, 1086-1095): the trainable layers' gradients (incl. the L2 term over trainable
    weights only) equal the oracle's with the same layer_regex; frozen layers have zero gradient and do not move."""
    from ursonet_amd.engine import Engine
    from ursonet_amd.graph import layer_regex
    cfg = make_config(backbone="resnet50", h=64, w=128, batch=2, regress_ori=False, ori_bins=4, dtype="float32")
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=2)
    rx = layer_regex("4+")
    eng = Engine(cfg, "training", seed=1, randomize_bn=True)
    eng.set_trainable(rx)
    w0 = eng.get_weights()
    eng.load_batch(img, loc, ori)
    eng.step(); torch.cuda.synchronize()
    dec = ReluDecisions(eng, tol=1e-5)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, layer_regex=rx)
    grads = eng.get_grads()
    assert "res4a_branch2a" in ref["grads"] and "res3a_branch2a" not in ref["grads"] and "conv1" not in ref["grads"]
    for ln, ws in ref["grads"].items():
        for wn, gref in ws.items():
            assert _rel(grads[ln][wn], gref.numpy()) < 1e-3, (ln, wn)
    for ln in grads:
        if ln not in ref["grads"]:
            assert all(float(np.abs(g).max()) == 0.0 for g in grads[ln].values()), ln
    assert abs(float(eng.normsq.cpu()) ** 0.5 - ref["grad_norm"]) < 1e-3 * ref["grad_norm"]
    w1 = eng.get_weights()
    for ln, ws in newW.items():
        for wn, wref in ws.items():
            assert _rel(w1[ln][wn], wref) < 1e-4, (ln, wn)


def test_graph_replay_equals_eager_and_is_deterministic():
    cfg = make_config(backbone="resnet18", h=64, w=64, batch=2, regress_ori=True, dtype="float32")
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=4)
    e1, _ = _run_engine(cfg, img, loc, ori, use_graph=True)
    e2, _ = _run_engine(cfg, img, loc, ori, use_graph=False)
    assert torch.equal(e1.flat_w, e2.flat_w) and torch.equal(e1.flat_g, e2.flat_g)
    # three more replays == three more eager steps, bit for bit (no atomics anywhere)
    for _ in range(3):
        e1.step(); e2.step_eager()
    torch.cuda.synchronize()
    assert torch.equal(e1.flat_w, e2.flat_w)


def _fork_worker(*args):
    """tests/workers/fork_worker.py in a process of its own (a forked hipGraph replay has segfaulted inside the ROCm runtime in long-lived
    processes: the fork is opt-in, and its tests cannot take the suite down)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.pop("URSO_WGRAD_STREAM", None)
    p = subprocess.run([sys.executable, os.path.join(here, "workers", "fork_worker.py")] + list(args), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, "fork worker failed (rc %d):\n%s" % (p.returncode, out[-3000:])
    line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("name,kw", [("r50", dict(backbone="resnet50", batch=4, h=256, w=320, regress_ori=False, ori_bins=16, dtype="bfloat16")),
                                     ("r101_f16", dict(backbone="resnet101", batch=2, h=192, w=256, regress_ori=False, ori_bins=8, f16=True))])
def test_weight_gradients_beside_the_chain_change_no_bit(name, kw):
    """The complementary fork (Engine._fork_weight_gradients, URSO_WGRAD_STREAM=2; opt-in since round 6, bench.py opts in): the arithmetic-heavy
    weight gradients of stages 4-5 deferred to the point where the data gradients reach stage 3 and run on a second branch of the captured
    graph.  Same kernels, same operands: weights, gradients and momentum after three replayed steps equal the single chain's bit for bit (a
    captured graph that ran a chain node early showed only from the second replay on: the first one reads what the eager warm-up step left).
    Runs in a process of its own (_fork_worker)."""
    import json
    r = _fork_worker("identity", json.dumps(kw))
    assert not r["0"]["side_stream"] and r["2"]["side_stream"] and r["2"]["forked"] and r["2"]["fork_checks"] >= 1
    assert r["2 rejected"]["side_stream"] and not r["2 rejected"]["forked"]
    for mode in ("2", "2 rejected"):
        assert r["equal"][mode], "forked plan (%s) differs from the single chain" % mode
        assert r[mode]["wgrads_right_in_front_of_point"] >= 3 and r[mode]["wgrads_before_point"] == r["0"]["wgrads_before_point"], r


def test_the_fork_is_opt_in(monkeypatch):
    """Library default since round 6: one chain (no second graph branch) unless URSO_WGRAD_STREAM says otherwise."""
    from ursonet_amd.engine import Engine
    monkeypatch.delenv("URSO_WGRAD_STREAM", raising=False)
    cfg = make_config(backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8, dtype="bfloat16")
    eng = Engine(cfg, "training", seed=1)
    assert eng.wgrad_stream is None and not eng.forked and eng.verify_fork() is False


def test_forked_graph_stress_240_replays_equal_the_chain_and_check_survives_an_empty_batch():
    """Long-run check of the forked backward pass (VERDICT r05 item 6 / ADVICE r05): the cfg2-width plan (ResNet-50, bottleneck 32,
    ori_resolution 16, 512 x 640, bf16; batch 2) replayed 240 times with fresh data every 20 replays; at every 20th replay the step is taken
    twice from the same state -- the forked graph, then the same launches eagerly on one chain -- and weights, gradients and momentum must
    agree bit for bit.  A scheduling-dependent reorder that a capture-time check misses has 240 chances here.  Also: a capture BEFORE the
    first load_batch (all-zero inputs and targets: rel_loss is 0/0 there) verifies on a stand-in batch and keeps the fork, leaving the buffers
    as it found them; verify_fork() re-runs the check on demand and leaves the training state untouched.  Runs in a process of its own."""
    r = _fork_worker("stress")
    assert r["forked_on_empty_batch"] and r["buffers_left_zero"] and r["fork_checks_after_capture"] == 1, r
    assert r["compared"] == 12 and not r["replays_that_differ"] and r["finite"] and r["forked_at_end"], r
    assert r["verify_fork"] and r["fork_checks_at_end"] == 2 and r["state_untouched_by_verify"], r


@pytest.mark.parametrize("dtype,tol_out,cos_min", [("bfloat16", 5e-2, 0.95), ("float16", 1e-2, 0.99)])
def test_training_step_parity_16bit(dtype, tol_out, cos_min):
    """bf16 (cfg2) and fp16 (cfg5, F16) storage with fp32 accumulation against the fp32 oracle: outputs within the format's
    rounding, gradient direction per tensor and globally."""
    kw = dict(backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8)
    cfg = make_config(dtype=dtype, **kw)
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=1)
    eng, w0 = _run_engine(cfg, img, loc, ori)
    ref, _ = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE)
    gl, go = eng.outputs()
    assert _rel(gl.cpu().numpy(), ref["loc"].numpy()) < tol_out
    assert _rel(go.cpu().numpy(), ref["ori"].numpy()) < tol_out
    ls = eng.losses()
    assert abs(ls["ori_loss"] - ref["ori_loss"]) < 2e-2 * abs(ref["ori_loss"])
    grads = eng.get_grads()
    allg, allr = [], []
    for ln, ws in ref["grads"].items():
        for wn, gref in ws.items():
            allg.append(grads[ln][wn].ravel()); allr.append(gref.numpy().ravel())
            if gref.numel() >= 4096:
                c = _cos(grads[ln][wn], gref.numpy())
                assert c > cos_min, "%s gradient direction %s/%s cos=%.4f" % (dtype, ln, wn, c)
    assert _cos(np.concatenate(allg), np.concatenate(allr)) > 0.99


@pytest.mark.parametrize("case", ["r50_bf16", "r18_f32_quat", "r50_bf16_heads_only", "r50_bf16_adam"])
def test_gradient_norm_from_the_finalisation_blocks_equals_the_norm_pass(case, monkeypatch):
    """Engine.fused_sqnorm (one gradient bucket, no all-reduce): the finalisation launches leave per-block sums of squares of every gradient value
    they store and urso_sqnorm_final adds them -- the 134 MB gradient buffer is not read back for the clip norm.  Against urso_sqnorm over the
    flat buffer of the same step (another fp32 summation order: 2e-6), against the plan with the separate pass (URSO_FUSE_SQNORM=0: same
    gradients bit for bit, post-step weights within the norm's rounding), and with frozen layers (their slices are zero and belong to no block)."""
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    from ursonet_amd.graph import layer_regex
    if case == "r18_f32_quat":
        cfg = make_config(backbone="resnet18", h=128, w=128, batch=2, regress_ori=True, dtype="float32")
    else:
        cfg = make_config("resnet50", 128, 192, batch=4, regress_ori=False, ori_bins=4, dtype="bfloat16", lr=1e-2)
        if case.endswith("adam"):
            cfg.OPTIMIZER = "ADAM"
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=9)
    res = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("URSO_FUSE_SQNORM", fuse)
        eng = Engine(cfg, "training", seed=3, randomize_bn=True)
        if "heads_only" in case:
            eng.set_trainable(layer_regex("heads"))
        assert eng.fused_sqnorm == (fuse == "1") and len(eng.buckets) == 1
        eng.load_batch(img, loc, ori)
        eng.step(); torch.cuda.synchronize()
        ref = torch.zeros(1, device="cuda")
        hip.sqnorm(eng.n_flat, eng.flat_g, eng.sq_ws, ref)
        torch.cuda.synchronize()
        assert float(ref) > 0 and abs(float(eng.normsq) - float(ref)) <= 2e-6 * float(ref), (case, fuse, float(eng.normsq), float(ref))
        res.append((eng.flat_g.clone(), eng.flat_w.clone(), float(eng.normsq)))
    assert torch.equal(res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-6 * float(res[1][1].abs().max())


def test_frozen_layers_get_no_update():
    """set_trainable('heads') (net.py:1086-1095): backbone weights must not move."""
    from ursonet_amd.engine import Engine
    from ursonet_amd.graph import layer_regex
    cfg = make_config(backbone="resnet50", h=64, w=64, batch=2, dtype="float32")
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=2)
    eng = Engine(cfg, "training", seed=1)
    eng.set_lr(0.0123)                                   # compile(lr) before set_trainable must survive the plan rebuild
    eng.set_trainable(layer_regex("heads"))
    assert abs(float(eng.hyper[0]) - 0.0123) < 1e-9
    # the data-gradient chain stops at the earliest trainable layer: nothing is back-propagated through the frozen backbone
    labs = [l for l in eng.labels["bwd"] if l]
    assert not any(l.startswith(("dgrad:res", "wgrad:res", "maxpool_bwd", "dgrad:bottleneck")) for l in labs), labs
    assert any(l.startswith("wgrad:bottleneck") for l in labs)
    w0 = eng.get_weights()
    eng.load_batch(img, loc, ori)
    eng.step(); torch.cuda.synchronize()
    w1 = eng.get_weights()
    moved = {ln for ln in w0 for wn in w0[ln] if not np.array_equal(w0[ln][wn], w1[ln][wn])}
    assert moved and all(ln.startswith(("loc_", "ori_", "bottleneck")) for ln in moved), moved


def test_inference_forward_matches_oracle_r101():
    from oracle import graph_ref as G
    from ursonet_amd.engine import Engine
    cfg = make_config(backbone="resnet101", h=64, w=128, batch=1, ori_bins=8, dtype="float32")
    img, *_ = synthetic_batch(cfg, 1, seed=5)
    eng = Engine(cfg, "inference", seed=7, randomize_bn=True)
    eng.load_batch(img)
    eng.forward(); torch.cuda.synchronize()
    P = G.to_torch(eng.get_weights(), requires_grad=False)
    loc, ori = G.forward(P, torch.tensor(img), cfg)
    gl, go = eng.outputs()
    assert _rel(gl.cpu().numpy(), loc.numpy()) < 1e-3 and _rel(go.cpu().numpy(), ori.numpy()) < 1e-3


def test_adam_training_steps_match_oracle():
    """OPTIMIZER != 'SGD' -> Adam(amsgrad, clipnorm) (net.py:982-983): two hipGraph replays vs two oracle steps.
    Adam's first updates are ~lr*sign(g), so weights whose gradient is ~0 are excluded from the comparison."""
    from ursonet_amd.engine import Engine
    cfg = make_config(dtype="float32", backbone="resnet18", h=64, w=64, batch=2, regress_ori=True)
    cfg.OPTIMIZER = "ADAM"
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=4)
    eng = Engine(cfg, "training", seed=5, randomize_bn=True)
    assert eng.adam
    w0 = eng.get_weights()
    eng.load_batch(img, loc, ori)
    eng.step(); eng.step()
    torch.cuda.synchronize()
    assert float(eng.hyper[5]) == 2.0
    from oracle import graph_ref as G
    P = G.to_torch(w0); st = {}
    r1 = G.train_step(P, st, torch.tensor(img), torch.tensor(loc), torch.tensor(ori), cfg, cfg.LEARNING_RATE)
    G.train_step(P, st, torch.tensor(img), torch.tensor(loc), torch.tensor(ori), cfg, cfg.LEARNING_RATE)
    w2 = eng.get_weights()
    worst = 0.0
    for ln, ws in P.items():
        for wn, wref in ws.items():
            if wn in ("moving_mean", "moving_variance"):
                continue
            g1 = r1["grads"][ln][wn].numpy()
            sel = np.abs(g1) > 1e-4 * np.abs(g1).max()
            d_ref = (wref.detach().numpy() - w0[ln][wn])[sel]
            d_gpu = (w2[ln][wn] - w0[ln][wn])[sel]
            if sel.any():
                worst = max(worst, float(np.abs(d_gpu - d_ref).max() / (np.abs(d_ref).max() + 1e-30)))
    assert worst < 2e-2, worst


@pytest.mark.parametrize("name,kw", [("r50", dict(backbone="resnet50", h=64, w=128, batch=4, regress_ori=False, ori_bins=4)),
                                     ("r18", dict(backbone="resnet18", h=128, w=128, batch=3, regress_ori=True))])
def test_batch_statistics_bn_mode_parity_fp32(name, kw):
    """TRAIN_BN = None ("Train BN layers", net.py:60-76): batch statistics in the training graph -- forward outputs,
    losses, every gradient and the moving-statistics update against the oracle (its batchnorm(training=None))."""
    from ursonet_amd.engine import Engine
    cfg = make_config(dtype="float32", **kw)
    cfg.TRAIN_BN = None
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=2)
    eng = Engine(cfg, "training", seed=7, randomize_bn=True)
    assert eng.train_bn
    w0 = eng.get_weights()
    eng.load_batch(img, loc, ori)
    eng.step()
    torch.cuda.synchronize()
    ref0, _ = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE)
    gl, go = eng.outputs()
    assert _rel(gl.cpu().numpy(), ref0["loc"].numpy()) < 1e-3
    assert _rel(go.cpu().numpy(), ref0["ori"].numpy()) < 1e-3
    dec = ReluDecisions(eng, tol=1e-4)      # normalised activations: rounding differences are relative to sigma
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec)
    assert dec.flips <= max(4, 1e-5 * dec.total)
    ls = eng.losses()
    assert abs(ls["loc_loss"] - ref["loc_loss"]) < 1e-3 * abs(ref["loc_loss"]) + 1e-6
    assert abs(ls["ori_loss"] - ref["ori_loss"]) < 1e-3 * abs(ref["ori_loss"]) + 1e-6
    grads = eng.get_grads()
    worst = ("", 0.0)
    bn_convs = {c.name for c in eng.convs.values() if c.batch_bn}
    for ln, ws in ref["grads"].items():
        for wn, gref in ws.items():
            if wn == "bias" and ln in bn_convs:
                # a bias in front of a batch-statistics BN has the exact gradient sum(dz) = 0 (+ the tiny L2 term): both sides
                # hold rounding noise only, so the comparison is absolute
                assert np.abs(grads[ln][wn] - gref.numpy()).max() < 2e-5, ln
                continue
            e = _rel(grads[ln][wn], gref.numpy())
            if e > worst[1]:
                worst = (ln + "/" + wn, e)
    assert worst[1] < 2e-3, "worst gradient mismatch %s: %.3e" % worst
    # moving statistics moved towards the batch statistics (momentum 0.99) and are no longer the initial ones
    w1 = eng.get_weights()
    bn = "bn_conv1" if name == "r50" else "bn_conv0"
    assert not np.allclose(w1[bn]["moving_mean"], w0[bn]["moving_mean"])
    assert np.abs(w1[bn]["moving_mean"] - w0[bn]["moving_mean"]).max() < 0.011 * (np.abs(w0[bn]["moving_mean"]).max() + 10 * np.abs(img).max())


def test_exact_rel_loss_mode_single_gpu_equals_default():
    """DP_EXACT_REL_LOSS splits rel_loss into norms -> (all-reduce) -> loss+gradient; on one GPU (world size 1) the
    two-phase kernels must reproduce the one-kernel loss and every gradient."""
    from ursonet_amd.engine import Engine
    res = []
    for exact in (False, True):
        cfg = make_config(dtype="float32", backbone="resnet18", h=64, w=64, batch=3, regress_ori=True)
        cfg.DP_EXACT_REL_LOSS = exact
        img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=6)
        eng = Engine(cfg, "training", seed=9, randomize_bn=True)
        assert bool(getattr(eng, "rel_exact", False)) == exact and (len(eng.loss_pre_ops) == 1) == exact
        eng.load_batch(img, loc, ori)
        eng.step()
        torch.cuda.synchronize()
        res.append((eng.losses(), eng.flat_g.clone(), eng.flat_w.clone()))
    assert abs(res[0][0]["loc_loss"] - res[1][0]["loc_loss"]) < 1e-6 * abs(res[0][0]["loc_loss"])
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-6 * float(res[0][1].abs().max())
    assert float((res[0][2] - res[1][2]).abs().max()) <= 1e-6 * float(res[0][2].abs().max())


def _full_size_properties(cfg, img, loc, ori, runs=2, steps=5):
    """Oracle-free properties at BASELINE.json's full sizes: (a) engines run from the same state produce bit-identical weights
    (fixed-order reductions, no atomics), (b) the loss on a fixed batch goes down, (c) every gradient is finite,
    (d) frozen-BN statistics are untouched."""
    from ursonet_amd.engine import Engine
    finals, first_losses, last_losses = [], [], []
    for run in range(runs):
        eng = Engine(cfg, "training", seed=21, randomize_bn=True)
        stats0 = eng.flat_stats.clone()
        eng.load_batch(img, loc, ori)
        eng.step(); torch.cuda.synchronize()
        l0 = eng.losses()
        assert bool(torch.isfinite(eng.flat_g).all())
        assert float(eng.flat_g.abs().max()) > 0
        for _ in range(steps):
            eng.step()
        torch.cuda.synchronize()
        l1 = eng.losses()
        first_losses.append(l0); last_losses.append(l1)
        finals.append(eng.flat_w.clone())
        assert torch.equal(eng.flat_stats, stats0)
        del eng
        torch.cuda.empty_cache()
    assert all(torch.equal(finals[0], f) for f in finals[1:]), "training is not deterministic"
    assert all(first_losses[0] == l for l in first_losses[1:]) and all(last_losses[0] == l for l in last_losses[1:])
    tot0 = first_losses[0]["loc_loss"] + first_losses[0]["ori_loss"]
    tot1 = last_losses[0]["loc_loss"] + last_losses[0]["ori_loss"]
    assert np.isfinite(tot1) and tot1 < tot0, (tot0, tot1)
    return first_losses[0], last_losses[0]


def test_full_size_cfg2_step_is_deterministic_and_descends():
    """BASELINE.json configs[1] at its real size (ResNet-50, 32 x 512 x 640, bf16)."""
    cfg = make_config(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16", lr=1e-3)
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=12)
    _full_size_properties(cfg, img, loc, ori)


def test_full_size_cfg4_r101_n24_rot_aug_targets():
    """BASELINE.json configs[3] at its real per-GPU size: ResNet-101, ori_resolution 24 (13,824 bins), batch 16 x 512 x 640,
    bf16, with the orientation targets produced the way rot_aug produces them (net.py:415-438): the pose is perturbed on the
    host (augment.rotate_pose) and re-encoded ON THE GPU (urso_encode_ori via augment.encode_orientations)."""
    from ursonet_amd import augment, pose
    cfg = make_config(backbone="resnet101", h=512, w=640, batch=16, regress_ori=False, ori_bins=24, dtype="bfloat16", lr=1e-3)
    img, loc, _, q = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=13)
    codec = pose.OrientationCodec(24, cfg.BETA)
    rng = np.random.default_rng(5)
    q2, loc2 = [], []
    for b in range(cfg.BATCH_SIZE):
        R = augment.euler2SO3_left(*rng.uniform(-10, 10, 3))
        t_new, q_new = augment.rotate_pose(loc[b].astype(np.float64), np.roll(q[b], 1).astype(np.float64), R)
        loc2.append(t_new); q2.append(q_new)
    ori = augment.encode_orientations(np.asarray(q2), codec.H_quat, codec.redundant, cfg.BETA)
    ori = ori.cpu().numpy() if torch.is_tensor(ori) else np.asarray(ori)
    assert ori.shape == (16, 13824) and np.allclose(ori.sum(1), 1, atol=1e-5)
    l0, l1 = _full_size_properties(cfg, img, np.asarray(loc2, dtype=np.float32), ori.astype(np.float32), runs=2, steps=4)
    assert l0["ori_loss"] > 5.0              # ~ln(13824) = 9.5 at initialisation with soft targets


def test_full_size_cfg5_f16_classify_loc_encode_loc_targets():
    """BASELINE.json configs[4] at its real size: ResNet-50, fp16, 32 x 640 x 960 (SPEED 1200x1920 at image_scale 0.5, padded
    to a multiple of 64), classification location head (LOC_BINS_PER_DIM 16 -> 4096) with targets from encode_loc
    (utils.py:349-396 restated in ursonet_amd.pose; SPEED itself builds no location map)."""
    from ursonet_amd import pose
    cfg = make_config(backbone="resnet50", h=640, w=960, batch=32, regress_ori=False, regress_loc=False, ori_bins=16, loc_bins=16,
                      dtype="float16", f16=True, lr=1e-3)
    img, _, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=14)
    rng = np.random.default_rng(6)
    xyz = np.stack([rng.uniform(-0.2, 0.2, 32), rng.uniform(-0.15, 0.15, 32), rng.uniform(5, 35, 32)], 1)
    loc, H = pose.encode_locations(xyz, 16, cfg.BETA, max_lim=[0.3, 0.25, 40.0], min_lim=[-0.3, -0.25, 3.0])      # urso_encode_loc
    loc = loc.cpu().numpy()
    assert loc.shape == (32, 4096) and np.allclose(loc.sum(1), 1, atol=1e-5)
    _full_size_properties(cfg, img, loc, ori, runs=2, steps=4)


def test_data_parallel_engine_single_rank_rccl_matches_plain_engine(monkeypatch):
    """The DP path (bucketed hipGraph segments + RCCL all-reduce between them) with ONE rank and the collectives forced on
    must reproduce the single-graph engine bit for bit (AVG over one rank is the identity); also with DP_EXACT_REL_LOSS."""
    import socket
    import torch.distributed as dist
    from ursonet_amd.engine import Engine
    from ursonet_amd.dp import DataParallelEngine
    from ursonet_amd import hip
    monkeypatch.setenv("URSO_DP_FORCE_COLLECTIVES", "1")
    # the DP path takes the clip norm of the all-reduced gradient with urso_sqnorm; the plain engines it is compared with bit for bit do the same
    # here (their default adds per-block sums of the finalisation launches: another fp32 summation order, test_gradient_norm_from_the_...)
    monkeypatch.setenv("URSO_FUSE_SQNORM", "0")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        hip.set_option("hconv_streamk", 0)      # what the DP wrapper sets while collectives run beside the step (below): the plain engines it is compared with sum in the same order
        for exact in (False, True):
            cfg = make_config(backbone="resnet50", h=64, w=128, batch=4, regress_ori=False, ori_bins=4, dtype="bfloat16")
            cfg.DP_EXACT_REL_LOSS = exact
            img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=3)
            plain = Engine(cfg, "training", seed=8, randomize_bn=True)
            plain.load_batch(img, loc, ori)
            for _ in range(3):
                plain.step()
            eng = Engine(cfg, "training", seed=8, randomize_bn=True, grad_bucket_bytes=8 << 20)
            dp = DataParallelEngine(eng, bucket_bytes=8 << 20)
            assert len(dp.buckets) >= 3
            eng.load_batch(img, loc, ori)
            for _ in range(3):
                dp.step()
            torch.cuda.synchronize()
            assert torch.equal(eng.flat_w, plain.flat_w), "DP (1 rank) differs from the plain engine (exact=%s)" % exact
            assert eng.losses() == plain.losses()
        # CUs reserved for the collective (comm_cus): the DP engine re-plans grids and weight-gradient splits for (CUs - comm_cus), option
        # `cus`, process-wide; a plain engine planned under the same option value is again matched bit for bit
        total = torch.cuda.get_device_properties(0).multi_processor_count
        eng = Engine(cfg, "training", seed=8, randomize_bn=True, grad_bucket_bytes=8 << 20)
        dp = DataParallelEngine(eng, bucket_bytes=8 << 20, comm_cus=total - 64, tail_bytes=256 << 10)
        assert hip.get_option("cus") == 64
        # ... and the stem-side bucket capped (what world size > 1 gets by default): one more, small, last bucket
        assert (dp.buckets[-1][1] - dp.buckets[-1][0]) * 4 <= 256 << 10 and dp.buckets[-1][0] == 0 and "conv1" in dp.buckets[-1][2]
        plain = Engine(cfg, "training", seed=8, randomize_bn=True)
        plain.grad_bucket_bytes, plain.grad_tail_bytes = 8 << 20, 256 << 10      # same buckets: a layer pair whose weight gradients share a launch
        plain._build_plan()                                                      # only when they share a bucket sums in another fp32 order otherwise
        for e, step in ((plain, plain.step), (eng, dp.step)):
            e.load_batch(img, loc, ori)
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        assert torch.equal(eng.flat_w, plain.flat_w) and eng.losses() == plain.losses()
        ex = dp.exposed_comm_ms(2)             # events around the join with the collectives' stream; one rank moves no bytes
        assert 0.0 <= ex < 5.0, ex
        # no accumulator hand-over between blocks (conv_halo.hip's stream-K needs every block resident) while collectives run beside the step:
        # the wrapper switches it off for the process and close() gives both options back and re-plans the engine for the whole chip
        dp.close()                              # gives the CUs back (the engine it wrapped is re-planned for the whole chip)
        assert hip.get_option("cus") == 0
        hip.set_option("hconv_streamk", 1)
        eng2 = Engine(cfg, "training", seed=8, randomize_bn=True, grad_bucket_bytes=8 << 20)
        with DataParallelEngine(eng2, bucket_bytes=8 << 20, comm_cus=total - 64) as dp2:
            assert hip.get_option("hconv_streamk") == 0 and hip.get_option("cus") == 64
            eng2.load_batch(img, loc, ori); dp2.step(); torch.cuda.synchronize()
            v_dp = eng2.plan_version
        assert hip.get_option("hconv_streamk") == 1 and hip.get_option("cus") == 0
        assert eng2.plan_version > v_dp and eng2._graphs is None
        eng2.load_batch(img, loc, ori); eng2.step_eager(); torch.cuda.synchronize()      # planned for the whole chip again: split counts agree with the launches
        assert bool(torch.isfinite(eng2.flat_g).all())
    finally:
        hip.set_option("cus", 0)
        hip.set_option("hconv_streamk", 1)
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_fused_pointwise_pairs_change_nothing_but_the_launch_count(dtype):
    """Engine plan with the stage-2 and stage-3 pointwise pairs fused (urso_conv_pair, default) against the plan with every layer
    launched on its own (option pair=0): same outputs and losses up to the rounding flips of the fused layers' outputs, eleven launches
    fewer (five forward pairs -- the first with res2a_branch1 inside, urso_conv_pair_shortcut -- and five backward pairs in ResNet-50)."""
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    cfg = make_config("resnet50", 64, 128, batch=4, regress_ori=False, ori_bins=4, dtype=dtype, lr=1e-3)
    img, loc, ori, _ = synthetic_batch(cfg, 4, seed=21)
    res = []
    for pair in (1, 0):
        with hip.options(pair=pair):               # (planned AND run under the option: the engine refuses to step under another pair policy)
            eng = Engine(cfg, "training", seed=5, randomize_bn=True)
            eng.load_batch(img, loc, ori); eng.step(); torch.cuda.synchronize()
        res.append((len(eng.fwd_ops), len(eng.bwd_ops), [t.float().clone() for t in eng.outputs()], eng.losses(), None,
                    sorted(eng.pair_first), sum(1 for l in eng.labels["bwd"] if l and l.startswith("dgrad:") and "+" in l), list(eng.shortcut_folded),
                    sum(1 for l in eng.labels["fwd"] if l and l.endswith("@sampled")), sum(1 for l in eng.labels["fwd"] if l and l.startswith("subsample:"))))
    assert res[0][5] == ["res2b_branch2a", "res2c_branch2a", "res3b_branch2a", "res3c_branch2a", "res3d_branch2a"] and res[1][5] == []
    # five fused launches forward (plus the stage-2 projection shortcut, computed inside the first of them), five backward
    # ... and in BOTH plans the three stage-closing layers are computed at the sampled pixels only (Engine._sample_block_output): they write
    # the compact tensor the next stage's entry layers read, no gather pass is left
    assert res[0][8] == 6 and res[1][8] == 6 and res[0][9] == 0 and res[1][9] == 0
    assert res[1][0] - res[0][0] == 6 and res[0][6] == 5 and res[1][6] == 0 and res[0][7] == ["res2a_branch1"] and res[1][7] == []
    tol_out = 2e-2 if dtype == "bfloat16" else 4e-3                               # the output gate of the oracle comparison above
    eo = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[0][2], res[1][2]))
    el = max(abs(res[0][3][k] - res[1][3][k]) / (abs(res[1][3][k]) + 1e-4) for k in res[0][3])
    print("fused vs apart (%s): outputs %.2e losses %.2e" % (dtype, eo, el))
    # the two plans differ by rounding flips in the forward pass only.  Their GRADIENTS are not compared with each other: one ReLU unit
    # of a Dense head that flips for one of the 4 samples moves a whole filter row by O(1), so two valid 16-bit runs differ by >10 %
    # in the tensor-max metric; each plan is compared with the oracle under pinned ReLU decisions instead
    # (test_training_step_parity_16bit_same_rounding_points[... fused / apart])
    assert eo <= tol_out and el <= tol_out


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_weight_gradient_folded_into_the_stage2_backward_pair(dtype):
    """Option pair = 1 (default) lets the stage-2 backward pairs also accumulate the weight gradient of res2{a,b}_branch2c
    (urso_conv_pair_wgrad) into those layers' split workspaces, and the pair behind the stage's first block take the projection shortcut
    res2a_branch1's data and weight gradient along (urso_conv_pair_wgrad_entry: the block-output gradient stays on chip); pair = 2 keeps
    the three weight-gradient launches.  Same forward plan,
    bit-identical data gradients: every gradient is equal bit for bit except those two layers' (kernel, folded BatchNorm), which
    differ by the fp32 summation order of a different pixel split only."""
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    cfg = make_config("resnet50", 64, 128, batch=4, regress_ori=False, ori_bins=4, dtype=dtype, lr=1e-3)
    img, loc, ori, _ = synthetic_batch(cfg, 4, seed=22)
    res = []
    for pair in (1, 2):
        with hip.options(pair=pair):
            eng = Engine(cfg, "training", seed=6, randomize_bn=True)
            eng.load_batch(img, loc, ori); eng.step(); torch.cuda.synchronize()
        res.append((eng.flat_g.clone(), dict(eng.slices), eng.losses(), [l for l in eng.labels["bwd"] if l and l.startswith("dgrad:") and "+wgrad:" in l],
                    sum(1 for l in eng.labels["bwd"] if l and l.startswith("wgrad:")), [l for l in eng.labels["bwd"] if l and l.startswith("dgrad+wgrad:")]))
    assert res[0][3] == ["dgrad:res2c_branch2a+res2b_branch2c+wgrad:res2b_branch2c",
                         "dgrad:res2b_branch2a+res2a_branch2c+res2a_branch1+wgrad:res2a_branch2c+res2a_branch1"]
    assert res[1][3] == [] and res[1][4] - res[0][4] == 3 and res[0][5] == [] and res[1][5] == []
    assert res[0][2] == res[1][2]
    seen = set()
    for (ln, wn), (o, n, _) in res[0][1].items():
        a, b = res[0][0][o:o + n], res[1][0][o:o + n]
        if ln.endswith(("2a_branch2c", "2b_branch2c", "2a_branch1")):        # res2a_branch2c / bn2a_branch2c, res2b_..., res2a_branch1 / bn2a_branch1
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9, (ln, wn)
            assert float(b.abs().max()) > 0
            seen.add(ln)
        else:
            assert torch.equal(a, b), (ln, wn)
    assert len(seen) == 6, seen


def test_step_planned_for_fewer_cus():
    """Option cus (what DataParallelEngine(comm_cus=N) sets before it plans): grids, weight-gradient splits and workspaces are planned
    for 64 of the device's CUs, so every persistent kernel walks a multi-tile stream and every split reduction sums another number of
    partials.  Same step up to summation order (fp32 partial sums of the weight gradients; where a stream-K 3x3 tile is cut).
    The option is held while the step runs: split counts are queried at plan AND launch time and must agree."""
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    cfg = make_config("resnet50", 128, 192, batch=4, regress_ori=False, ori_bins=4, dtype="bfloat16", lr=1e-3)
    img, loc, ori, _ = synthetic_batch(cfg, 4, seed=23)
    res = []
    for cus in (0, 64):
        with hip.options(cus=cus):
            eng = Engine(cfg, "training", seed=7, randomize_bn=True)
            eng.load_batch(img, loc, ori); eng.step(); torch.cuda.synchronize()
            res.append((eng.flat_g.clone(), eng.losses(), eng.flat_w.clone(), sum(getattr(c, "splits", 0) for c in eng.convs.values())))
    assert res[1][3] < res[0][3]                   # fewer partials were planned
    # a stream-K 3x3 tile or a split-K GEMM is cut at other boundaries (fp32 order, then one 16-bit rounding): two valid 16-bit runs, whose
    # rounding flips make tensor-wise gradient comparisons meaningless (see test_fused_pointwise_pairs_...); what a plan/launch
    # disagreement about a split count would produce is garbage or NaN, which these bounds exclude
    for k, v in res[0][1].items():
        assert abs(v - res[1][1][k]) <= 2e-2 * abs(v) + 1e-4, (k, v, res[1][1][k])
    g0, g1 = res[0][0].double(), res[1][0].double()
    assert bool(torch.isfinite(g1).all()) and float(g0.norm()) > 0
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    assert cos >= 0.98 and 0.9 <= float(g1.norm() / g0.norm()) <= 1.1, cos


def test_engine_refuses_to_step_under_other_planning_options():
    """Split counts, partial workspaces and the grouped / paired weight-gradient launches are sized at plan time from kernel-policy options
    (cus, wgrad_blocks, ...), and the library re-derives some split counts at launch time: a step under other values would sum a different
    number of partials than it wrote.  The engine says so instead."""
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    cfg = make_config("resnet50", 64, 128, batch=2, regress_ori=False, ori_bins=4, dtype="bfloat16")
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=2)
    eng = Engine(cfg, "training", seed=7, randomize_bn=True)
    eng.load_batch(img, loc, ori)
    eng.step_eager()
    with hip.options(wgrad_blocks=256):
        with pytest.raises(RuntimeError, match="wgrad_blocks 512 -> 256"):
            eng.step_eager()
    eng.step_eager(); torch.cuda.synchronize()


def test_urso_comm_bucket_averaging_one_rank(monkeypatch):
    """The C-ABI exchange step (urso_comm_*: RCCL bound at run time) with one rank: the average over one rank is the identity, the
    collective runs on the communicator's own stream ordered after the producer kernel, and urso_comm_wait orders the consumer
    after it; then the same transport under DataParallelEngine reproduces the plain engine's step bit for bit."""
    import socket
    import torch.distributed as dist
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    from ursonet_amd.dp import DataParallelEngine
    monkeypatch.setenv("URSO_FUSE_SQNORM", "0")          # the plain engine takes its clip norm the way the DP path must (urso_sqnorm): bit-for-bit comparison below
    comm = hip.Comm(1, 0, hip.comm_unique_id())
    try:
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            x = torch.randn(3_000_001, device="cuda").to(dtype)
            y = x.clone()
            y.mul_(2)                                   # producer on the compute stream
            comm.allreduce_bucket(y)
            comm.wait()
            z = y * 0.5                                 # consumer, ordered after the collective
            torch.cuda.synchronize()
            assert torch.equal(z, x)
        with pytest.raises((hip.UrsoHipError, KeyError)):                # float64 has no URSO dtype
            comm.allreduce_bucket(torch.zeros(4, device="cuda", dtype=torch.float64))
    except BaseException:
        comm.close()
        raise
    cfg = make_config("resnet18", 64, 128, batch=4, regress_ori=True, dtype="bfloat16", lr=1e-2)
    img, loc, ori, _ = synthetic_batch(cfg, 4, seed=2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        ref = Engine(cfg, "training", seed=9, randomize_bn=True)
        ref.load_batch(img, loc, ori); ref.step(); ref.step(); torch.cuda.synchronize()
        eng = Engine(cfg, "training", seed=9, randomize_bn=True)
        dp = DataParallelEngine(eng, bucket_bytes=4 << 20, comm=comm)
        assert len(dp.buckets) >= 2
        eng.load_batch(img, loc, ori); dp.step(); dp.step(); torch.cuda.synchronize()
        assert torch.equal(eng.flat_w, ref.flat_w) and torch.equal(eng.flat_g, ref.flat_g)
    finally:
        dist.destroy_process_group()
        comm.close()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_compact_stage_boundary_gradients_equal_the_dense_path(dtype, monkeypatch):
    """The gradient of a stage's last block output is non-zero only at the pixels the next stage's stride-2 layers sample.  The engine
    keeps it compact ([B, H/2, W/2, C]: Engine._plan_compact_gradients) and runs the strided data gradients as plain GEMMs, the
    producer's weight gradient as a stride-2 weight gradient, its data gradient as a compact scatter and the residual hand-over through
    urso_conv_pair's compact add operand.  Same step with the dense path (URSO_COMPACT_GRAD=0): same losses, activation and weight
    gradients equal up to the fp32 summation order of the re-shaped kernels (a fraction of one storage rounding step)."""
    from ursonet_amd.engine import Engine
    cfg = make_config("resnet50", 128, 192, batch=2, regress_ori=False, ori_bins=4, dtype=dtype, lr=1e-3)
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=33)
    res = []
    monkeypatch.setenv("URSO_SAMPLED_OUTPUTS", "0")        # same forward pass in both plans: this test isolates the gradient side (the next one the forward side)
    for mode in ("1", "0"):
        monkeypatch.setenv("URSO_COMPACT_GRAD", mode)
        eng = Engine(cfg, "training", seed=5, randomize_bn=True)
        eng.load_batch(img, loc, ori); eng.step(); torch.cuda.synchronize()
        compact = sorted(c.name for c in eng.convs.values() if c.dst.compact is not None)
        res.append((compact, eng.get_grads(), eng.losses(), {n: c.src.grad.float().clone() for n, c in eng.convs.items()
                                                             if c.src.grad is not None and c.src.compact is None}))
    assert res[0][0] == ["res2c_branch2c", "res3d_branch2c", "res4f_branch2c"] and res[1][0] == []      # the last block of stages 2-4
    assert res[0][2] == res[1][2]
    assert len(res[0][3]) >= 40
    wa = max((float((gc - res[1][3][n]).norm() / (res[1][3][n].norm() + 1e-30)), n) for n, gc in res[0][3].items())
    ww = max((_rel(res[0][1][ln][wn], res[1][1][ln][wn]), ln + "/" + wn) for ln in res[1][1] for wn in res[1][1][ln])
    print("compact vs dense (%s): worst activation gradient (L2) %.2e at %s, worst weight gradient %.2e at %s" % (dtype, wa[0], wa[1], ww[0], ww[1]))
    # not bit-identical: the compact-scatter data gradient and the stride-2 weight gradient sum in another order, and one fp32 ulp decides
    # a 16-bit rounding here and there; the plans agree to a small fraction of a storage rounding step
    tol = 1e-2 if dtype == "bfloat16" else 2e-3          # measured 1e-3 ... 5e-3 / 1e-3 at this tiny size (3e-7 at cfg2 size: only the summation order of six weight gradients differs)
    assert wa[0] < tol and ww[0] < tol, (wa, ww)


def test_bench_under_torchrun_with_forced_collectives_one_rank():
    """What the driver's N > 1 run does, with the one GPU a test box has: `python -m torch.distributed.run --nproc-per-node 1 bench.py
    --gpus 1` with URSO_DP_FORCE_COLLECTIVES=1 runs the REAL data-parallel path (rendezvous on 127.0.0.1, RCCL communicator, weight
    broadcast, segmented hipGraphs, one RCCL all-reduce per gradient bucket, join before the optimizer).  The JSON line must carry the
    dp record (exposed_comm_ms, bucket_bytes summing to 4 bytes x parameters) and a throughput within 5 % of the plain single-process
    run (a one-rank all-reduce is a copy: the segmentation itself must cost next to nothing)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["bench.py", "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--pcie-steps", "0", "--profile-steps", "1"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("URSO_DP_FORCE_COLLECTIVES", None)
    env["URSO_WGRAD_STREAM"] = "0"        # like with like: data-parallel plans keep the single chain (Engine._fork_weight_gradients)
    plain = subprocess.run([sys.executable] + common, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert plain.returncode == 0, plain.stderr.decode()[-2000:]
    p = json.loads(plain.stdout.decode().strip().splitlines()[-1])
    env["URSO_DP_FORCE_COLLECTIVES"] = "1"
    dp = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", "29541"] + common, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert dp.returncode == 0, dp.stderr.decode()[-2000:]
    d = json.loads([l for l in dp.stdout.decode().strip().splitlines() if l.startswith("{")][-1])
    assert p["dp"] is None and d["dp"] is not None
    assert d["n_gpus"] == 1 and d["metric"] == p["metric"]
    assert d["dp"]["exposed_comm_ms"] >= 0.0 and len(d["dp"]["bucket_bytes"]) >= 2
    # the self-diagnosis of a scaling run: per-rank exposed exchange time and step time, the bucket schedule, RCCL's own account of itself
    assert len(d["dp"]["exposed_comm_ms_per_rank"]) == 1 and len(d["dp"]["ms_per_step_per_rank"]) == 1
    assert abs(d["dp"]["ms_per_step_per_rank"][0] - d["ms_per_step"]) < 1e-2 and d["dp"]["n_buckets"] == len(d["dp"]["bucket_bytes"])
    assert isinstance(d["dp"]["rccl"], list) and len(d["dp"]["rccl"]) >= 1, d["dp"]["rccl"]
    cfg = make_config("resnet50", 512, 640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16")
    from ursonet_amd.graph import build_graph
    # trainable parameters as the flat gradient buffer lays them out (each tensor padded to 4 floats; moving statistics are not gradients)
    n_params = sum((int(np.prod(shp)) + 3) // 4 * 4 for ws in build_graph(cfg).params.values() for wn, shp in ws.items()
                   if wn not in ("moving_mean", "moving_variance"))
    assert sum(d["dp"]["bucket_bytes"]) == 4 * n_params
    assert abs(d["value"] / p["value"] - 1.0) < 0.05, (d["value"], p["value"])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_block_outputs_computed_at_sampled_pixels_change_no_result_bit(dtype, monkeypatch):
    """Engine._sample_block_output: res{2c,3d,4f}_out are only read through stride-2 layers, so their producing layers run over the even
    pixels only and the dense tensors / bit masks are never written.  Against the plan that computes them densely (URSO_SAMPLED_OUTPUTS=0),
    and so does the 3x3 layer below each of them (scattered to the even pixels of its dense buffer, Engine._sample_layer_below).
    With the layers on the same kernel in both plans (pair = 0, c3 = 0, hconv = 0: conv_pw.hip), NOTHING observable may change by a single bit: outputs, losses,
    every gradient, the global norm, the post-step weights and momentum -- over two steps.  Training and inference plans."""
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    # (8 x 512 x 640: every layer whose launch differs between the two plans has more than 128 tiles of 128 x 128 in its dense form, so
    # the dense plan does not send it through the split-K kernel that small grids get -- a different, equally valid summation order)
    cfg = make_config("resnet50", 512, 640, batch=8, regress_ori=False, ori_bins=4, dtype=dtype, lr=1e-2)
    img, loc, ori, _ = synthetic_batch(cfg, 8, seed=41)
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("URSO_SAMPLED_OUTPUTS", mode)
        with hip.options(pair=0, c3=0, hconv=0):
            eng = Engine(cfg, "training", seed=5, randomize_bn=True)
            inf = Engine(cfg, "inference", seed=5, randomize_bn=True)
            eng.load_batch(img, loc, ori); eng.step(); eng.step(); torch.cuda.synchronize()
            inf.load_batch(img); inf.forward(); torch.cuda.synchronize()
        n_s = sum(1 for l in eng.labels["fwd"] if l and l.endswith("@sampled")), sum(1 for l in inf.labels["fwd"] if l and l.endswith("@sampled"))
        res.append((n_s, [t.clone() for t in eng.outputs()], eng.losses(), eng.flat_g.clone(), eng.flat_w.clone(), eng.flat_v.clone(), float(eng.normsq),
                    [t.clone() for t in inf.outputs()], sum(1 for l in eng.labels["bwd"] if l == "bits_subsample")))
    assert res[0][0] == (6, 6) and res[1][0] == (0, 0) and res[0][8] == 0 and res[1][8] == 3       # three block outputs + the three 3x3 layers below them
    a, b = res
    assert all(torch.equal(x, y) for x, y in zip(a[1], b[1])) and a[2] == b[2] and a[6] == b[6]
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])
    assert all(torch.equal(x, y) for x, y in zip(a[7], b[7]))
    assert float(a[3].abs().max()) > 0 and all(np.isfinite(v) for v in a[2].values())


def test_training_step_with_winograd_forward_layers(monkeypatch):
    """URSO_WINOGRAD=1: every 3x3 / stride-1 layer of the forward pass runs the Winograd F(2x2, 3x3) evaluation (conv_winograd.hip);
    backward pass and everything else unchanged.  One bf16 training step against the rounding-aware oracle at the 16-bit gates
    (the sampled block-output plan is switched off so that all sixteen 3x3 layers of ResNet-50 take the Winograd path)."""
    import ursonet_amd.hip as hip
    from oracle import graph_ref as G
    monkeypatch.setenv("URSO_WINOGRAD", "1")
    monkeypatch.setenv("URSO_SAMPLED_OUTPUTS", "0")
    cfg = make_config(dtype="bfloat16", backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, ori_bins=8)
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=1)
    eng, w0 = _run_engine(cfg, img, loc, ori)
    assert sum(1 for c in eng.convs.values() if getattr(c, "winograd", False)) == 16
    q = G.StorageRounding(torch.bfloat16, unstored=getattr(eng, "shortcut_folded", ()))
    dec = ReluDecisions(eng, tol=8e-2)
    ref, newW = _oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
    assert dec.flips <= 4e-3 * dec.total, "too many ReLU decision flips: %d of %d" % (dec.flips, dec.total)
    _compare_step(eng, ref, newW, 3e-2, 6e-2, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_grouped_weight_gradients_in_the_plan(dtype, monkeypatch):
    """URSO_WGRAD_GROUP (default 8): consecutive layers of the general weight-gradient kernel in a gradient bucket share one weight-gradient launch with 1/n of the splits
    each (urso_wgrad_group_run).  Against the per-layer plan (URSO_WGRAD_GROUP=0) on the same step: forward outputs, losses and every
    data gradient are the same launches (bit-identical losses), the weight gradients are the same fp32 sums cut into fewer, longer pieces --
    equal to a few fp32 ulps of the largest entry of their layer; the grouped layers' split counts shrink; a two-bucket plan never lets a
    group straddle the bucket's reduction."""
    from ursonet_amd import hip
    from ursonet_amd.engine import Engine
    cfg = make_config("resnet50", 512, 640, batch=8, regress_ori=False, ori_bins=4, dtype=dtype, lr=1e-2)
    img, loc, ori, _ = synthetic_batch(cfg, 8, seed=43)
    res = []
    for mode in ("8", "0"):
        monkeypatch.setenv("URSO_WGRAD_GROUP", mode)
        eng = Engine(cfg, "training", seed=5, randomize_bn=True, grad_bucket_bytes=24 << 20)
        eng.load_batch(img, loc, ori); eng.step(); torch.cuda.synchronize()
        grouped = [l for l in eng.labels["bwd"] if l and l.startswith("wgrad:") and "+" in l and "maxpool" not in l]
        res.append((eng, grouped, eng.flat_g.clone(), eng.losses(), {n: getattr(c, "splits", 0) for n, c in eng.convs.items()}))
    (e1, g1, fg1, l1, s1), (e0, g0, fg0, l0, s0) = res
    assert len(g1) >= 4 and not g0 and e1.n_wgrad_groups == len(g1) and e0.n_wgrad_groups == 0
    assert l1 == l0
    names = [n for l in g1 for n in l[len("wgrad:"):].split("+")]
    assert len(names) == len(set(names)) >= 16 and sum(s1[n] for n in names) * 2 <= sum(s0[n] for n in names)     # (a single layer may get a few more: 256 x 256 tiles, same >= 8 steps per block)
    # no group straddles a bucket: all members of a launch finalise in the same bucket
    bucket_of = {n: k for k, (_, _, ns) in enumerate(e1.buckets) for n in ns}
    assert len(e1.buckets) >= 3 and all(len({bucket_of[n] for n in l[len("wgrad:"):].split("+")}) == 1 for l in g1)
    for name, (off, n, _) in e1.slices.items():
        a, b = fg1[off:off + n], fg0[off:off + n]
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-12, name
    assert float(fg1.abs().max()) > 0
