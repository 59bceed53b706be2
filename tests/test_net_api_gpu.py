"""Drop-in API on the GPU: net.UrsoNet train / detect / weights round trip / batched decode, written
the way pose_estimator.py drives the reference (pose_estimator.py:747-758, 321-445, 875-913)."""
import os

import numpy as np
import pytest
import torch

from util import make_config

pytestmark = pytest.mark.gpu


def test_train_checkpoint_resume_and_detect(tmp_path):
    from ursonet_amd import net, utils
    from ursonet_amd.dataset import SyntheticPoses
    cfg = make_config("resnet18", 64, 128, batch=4, regress_ori=False, ori_bins=4, dtype="float32", lr=0.01)
    cfg.NAME = "syn"
    cfg.STEPS_PER_EPOCH, cfg.VALIDATION_STEPS = 6, 2
    ds_train, ds_val = SyntheticPoses(16, 64, 128, cfg, seed=1), SyntheticPoses(8, 64, 128, cfg, seed=2)
    model = net.UrsoNet(mode="training", config=cfg, model_dir=str(tmp_path))
    hist = model.train(ds_train, ds_val, learning_rate=cfg.LEARNING_RATE, epochs=2, layers="all")
    assert len(hist.ori_loss_acc) == 12 and len(hist.loc_loss_acc) == 12 and np.isfinite(hist.ori_loss_acc).all()
    assert np.mean(hist.ori_loss_acc[-3:]) < np.mean(hist.ori_loss_acc[:3])          # it learns on 16 images
    assert model.epoch == 2
    ddir, ck = model.find_last()
    assert ck.endswith("weights_syn_0002.npz") and os.path.exists(ck)
    # resume bookkeeping (pose_estimator.py:889-913)
    m2 = net.UrsoNet(mode="inference", config=_inference_cfg(cfg), model_dir=str(tmp_path))
    m2.load_weights(ck, ck, by_name=True)
    assert m2.epoch == 2 and m2.log_dir == ddir
    w_tr, w_inf = model._engine.get_weights(), m2._engine.get_weights()
    assert all(np.array_equal(w_tr[l][w], w_inf[l][w]) for l in w_tr for w in w_tr[l])
    # detect: list of dicts with raw outputs; matches the training-mode forward on the same image
    img = ds_val.load_image(0)
    res = m2.detect([img], verbose=0)
    assert set(res[0]) == {"loc", "ori"} and res[0]["loc"].shape == (3,) and res[0]["ori"].shape == (64,)
    with pytest.raises(AssertionError):
        m2.detect([img, img])
    # probabilistic soft-argmax on the GPU == the reference-style host decode of the same logits
    from oracle import pose_math as P
    q_gpu = utils.decode_orientations(res[0]["ori"][None], ds_val.ori_histogram_map)[0]
    q_ref = P.decode_orientation(res[0]["ori"], ds_val.ori_histogram_map)
    assert abs(abs(float(np.dot(q_gpu, q_ref))) - 1) < 1e-5
    ang, le, esa = utils.pose_errors(res[0]["loc"], q_gpu, ds_val.load_location(0), ds_val.load_quaternion(0))
    assert np.isfinite(ang).all() and np.isfinite(esa).all()


def _inference_cfg(cfg):
    import copy
    c = copy.copy(cfg)
    c.IMAGES_PER_GPU = 1
    c.update()
    return c


def test_detect_matches_oracle_forward_on_uint8_image(tmp_path):
    from oracle import graph_ref as G
    from ursonet_amd import net
    cfg = make_config("resnet50", 64, 128, batch=1, regress_ori=True, dtype="float32")
    cfg.NAME = "x"
    model = net.UrsoNet(mode="inference", config=cfg, model_dir=str(tmp_path))
    img = np.random.default_rng(0).integers(0, 256, size=(64, 128, 3), dtype=np.uint8)
    out = model.detect([img])[0]
    P = G.to_torch(model._engine.get_weights(), requires_grad=False)
    loc, ori = G.forward(P, torch.tensor(G.mold_image(img, cfg)[None].astype(np.float32)), cfg)
    assert np.abs(out["loc"] - loc.numpy()[0]).max() < 1e-3 * np.abs(loc.numpy()).max()
    assert np.abs(out["ori"] - ori.numpy()[0]).max() < 1e-3


def test_f16_config_runs_fp16_path():
    """cfg5-style: F16=True selects the fp16 MFMA kernels (net.py:590-593 switches Keras to float16)."""
    from ursonet_amd.engine import Engine
    from oracle import graph_ref as G
    from util import synthetic_batch
    cfg = make_config("resnet50", 64, 128, batch=2, regress_ori=False, regress_loc=False, ori_bins=4, loc_bins=4, f16=True)
    eng = Engine(cfg, "training", seed=2, randomize_bn=True)
    assert eng.dt_name == "float16"
    img, loc, ori, _ = synthetic_batch(cfg, 2, seed=3)
    w0 = eng.get_weights()
    eng.load_batch(img, loc, ori); eng.step(); torch.cuda.synchronize()
    ref_l, ref_o = G.forward(G.to_torch(w0, requires_grad=False), torch.tensor(img), cfg)
    gl, go = eng.outputs()
    assert torch.isfinite(gl).all() and torch.isfinite(go).all()
    assert float((go.cpu() - ref_o).abs().max() / ref_o.abs().max()) < 5e-2
