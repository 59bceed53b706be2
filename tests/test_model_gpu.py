"""Whole-path parity on the GPU: ursonet_amd.Engine (HIP kernels through the C ABI, hipGraph
replay) vs the CPU oracle (oracle/graph_ref.py) on identical weights and inputs.

Tolerances (north_star: outputs within 1e-3 relative fp32):
  fp32 compute : outputs, losses, every parameter gradient and the post-step weights <= 1e-3
                 (relative to the tensor's max |.|);  in practice ~1e-5.
  bf16 compute : activations/weights are rounded to bf16 (8-bit mantissa) at every layer, so
                 outputs are compared at 5e-2 and gradients by cosine similarity (>= 0.95 per tensor,
                 >= 0.99 over all parameters); the bf16 KERNELS are checked tightly in test_kernels_gpu.py.
"""
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

    def __call__(self, site, x):
        c = self.eng.convs[site]
        n = c.node
        B = x.shape[0]
        dev_act = c.dst.data.float().cpu().view(B, -1)[:, :n.dst.h * n.dst.w * c.npad]
        if x.dim() == 4:
            m = (dev_act.view(B, n.dst.h, n.dst.w, c.npad)[..., :n.cout] > 0).permute(0, 3, 1, 2)
        else:
            m = dev_act.view(B, c.npad)[:, :n.cout] > 0
        diff = m != (x.detach() > 0)
        nd = int(diff.sum())
        if nd:
            worst = float(x.detach().abs()[diff].max() / (x.detach().abs().max() + 1e-30))
            assert worst < self.tol, "ReLU decision differs at |pre-activation| = %.2e of max in %s" % (worst, site)
        self.flips += nd
        self.total += diff.numel()
        return m


def _oracle_step(cfg, weights, img, loc, ori, lr, relu_hook=None):
    from oracle import graph_ref as G
    P = G.to_torch(weights)
    vel = {}
    out = G.train_step(P, vel, torch.tensor(img), torch.tensor(loc), torch.tensor(ori), cfg, lr, relu_hook=relu_hook)
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


def test_frozen_layers_get_no_update():
    """set_trainable('heads') (net.py:1086-1095): backbone weights must not move."""
    from ursonet_amd.engine import Engine
    from ursonet_amd.graph import layer_regex
    cfg = make_config(backbone="resnet50", h=64, w=64, batch=2, dtype="float32")
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=2)
    eng = Engine(cfg, "training", seed=1)
    eng.set_trainable(layer_regex("heads"))
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


def test_full_size_cfg2_step_is_deterministic_and_descends():
    """BASELINE.json configs[1] at its real size (ResNet-50, 32 x 512 x 640, bf16) through properties that do not need the
    oracle: (a) two engines run from the same state produce bit-identical weights after 3 steps (fixed-order reductions,
    no atomics), (b) the loss on a fixed batch goes down, (c) every gradient is finite, (d) frozen-BN statistics unchanged."""
    from ursonet_amd.engine import Engine
    cfg = make_config(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16", lr=1e-3)
    img, loc, ori, _ = synthetic_batch(cfg, cfg.BATCH_SIZE, seed=12)
    finals, first_losses, last_losses = [], [], []
    for run in range(2):
        eng = Engine(cfg, "training", seed=21, randomize_bn=True)
        stats0 = eng.flat_stats.clone()
        eng.load_batch(img, loc, ori)
        eng.step(); torch.cuda.synchronize()
        l0 = eng.losses()
        assert bool(torch.isfinite(eng.flat_g).all())
        for _ in range(5):
            eng.step()
        torch.cuda.synchronize()
        l1 = eng.losses()
        first_losses.append(l0); last_losses.append(l1)
        finals.append(eng.flat_w.clone())
        assert torch.equal(eng.flat_stats, stats0)
        del eng
        torch.cuda.empty_cache()
    assert torch.equal(finals[0], finals[1]), "training is not deterministic"
    assert first_losses[0] == first_losses[1] and last_losses[0] == last_losses[1]
    tot0 = first_losses[0]["loc_loss"] + first_losses[0]["ori_loss"]
    tot1 = last_losses[0]["loc_loss"] + last_losses[0]["ori_loss"]
    assert np.isfinite(tot1) and tot1 < tot0, (tot0, tot1)


def test_data_parallel_engine_single_rank_rccl_matches_plain_engine(monkeypatch):
    """The DP path (bucketed hipGraph segments + RCCL all-reduce between them) with ONE rank and the collectives forced on
    must reproduce the single-graph engine bit for bit (AVG over one rank is the identity); also with DP_EXACT_REL_LOSS."""
    import socket
    import torch.distributed as dist
    from ursonet_amd.engine import Engine
    from ursonet_amd.dp import DataParallelEngine
    monkeypatch.setenv("URSO_DP_FORCE_COLLECTIVES", "1")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
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
    finally:
        dist.destroy_process_group()
