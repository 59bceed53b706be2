"""CPU tests of the product's host logic (no GPU): Config, graph inventory, pose codec, image
resizing, cyclic LR, UrsoNet's checkpoint/log-dir plumbing, weight files, the data generator, the
C-ABI library (loads, exports every symbol of include/ursonet_hip.h, rejects bad arguments), and
the repository layout rules (the product never touches oracle/)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from util import make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "meta.json")))


# ------------------------------------------------------------------ Config
def test_config_defaults_match_reference():
    from ursonet_amd.config import Config
    c = Config()
    for k, v in META["config_defaults"].items():
        got = getattr(c, k)
        got = got.tolist() if isinstance(got, np.ndarray) else got
        assert got == v, (k, got, v)


def test_cli_configs_match_reference_update():
    cases = {"cfg1": ("resnet18", 128, 128, 2), "cfg2": ("resnet50", 512, 640, 32), "cfg4": ("resnet101", 512, 640, 16),
             "cfg5": ("resnet50", 640, 960, 32)}
    for name, (bb, h, w, b) in cases.items():
        c = make_config(bb, h, w, b)
        ref = META["cli_configs"][name]
        assert [int(v) for v in c.IMAGE_SHAPE] == ref["IMAGE_SHAPE"] and c.IMAGE_META_SIZE == ref["IMAGE_META_SIZE"]
        assert c.BATCH_SIZE == ref["BATCH_SIZE"]


def test_config_write_to_file(tmp_path):
    c = make_config()
    p = str(tmp_path / "sub" / "config_0.json")
    c.write_to_file(p)
    d = json.load(open(p))
    assert d["BACKBONE"] == "resnet50" and "MEAN_PIXEL" not in d and d["LOSS_WEIGHTS"]["ori_loss"] == 1.0


# ------------------------------------------------------------------ graph inventory
@pytest.mark.parametrize("kw", [dict(backbone="resnet18", regress_ori=True), dict(backbone="resnet34"), dict(backbone="resnet50"),
                                dict(backbone="resnet101", ori_bins=24), dict(backbone="resnet50", regress_loc=False),
                                dict(backbone="resnet50", regress_ori=True, ori_param="angle_axis")])
def test_graph_layer_names_and_shapes_equal_oracle_inventory(kw):
    from oracle import graph_ref as G
    from ursonet_amd.graph import build_graph, conv_flops
    cfg = make_config(h=128, w=192, **kw)
    g = build_graph(cfg)
    ref = {n: dict(ws) for n, _, ws in G.layer_specs(cfg)}
    assert {n: dict(ws) for n, ws in g.params.items()} == ref
    assert conv_flops(g, 3) == G.algorithmic_flops(cfg, 3)


def test_graph_rejects_bad_image_size_like_reference():
    from ursonet_amd.graph import build_graph
    cfg = make_config(h=100, w=128)
    with pytest.raises(Exception, match="dividable by 2 at least 6 times"):
        build_graph(cfg)


def test_trainable_presets():
    from ursonet_amd.graph import build_graph, layer_regex
    g = build_graph(make_config("resnet50", 64, 64))
    heads = [n for n in g.params if re.fullmatch(layer_regex("heads"), n)]
    assert set(heads) == {"bottleneck_layer", "loc_dense_0", "loc_final", "ori_dense_0", "ori_final"}
    s4 = [n for n in g.params if re.fullmatch(layer_regex("4+"), n)]
    assert "res4a_branch2a" in s4 and "bn5c_branch2c" in s4 and "res3d_branch2c" not in s4 and "conv1" not in s4
    assert all(re.fullmatch(layer_regex("all"), n) for n in g.params)


# ------------------------------------------------------------------ pose codec (product side) vs golden
@pytest.mark.parametrize("n", [4, 8, 16])
def test_product_orientation_codec_matches_reference(n):
    from ursonet_amd import utils
    g = np.load(os.path.join(GOLD, "ori_codec.npz"))
    enc, Hq, red = utils.encode_ori(g["oris"], n, META["beta"], np.array(META["min_lim"]), np.array(META["max_lim"]))
    assert np.array_equal(Hq, g["Hquat_%d" % n]) and np.array_equal(red, g["red_%d" % n])
    assert np.allclose(enc, g["enc_%d" % n], rtol=1e-6, atol=1e-9)
    f = utils.encode_ori_fast(g["oris"][2], META["beta"], Hq, red)
    assert np.allclose(f, g["fast_%d" % n][2], rtol=1e-12, atol=1e-15)


def test_product_encode_loc_and_softmax_match_reference():
    from ursonet_amd import utils
    g = np.load(os.path.join(GOLD, "loc_codec.npz"))
    enc, H = utils.encode_loc(g["locs"], 8, META["beta"], g["max_lim"], g["min_lim"])
    assert np.allclose(H, g["map_8"], atol=1e-12) and np.allclose(enc, g["enc_8"], rtol=1e-5, atol=1e-9)
    s = np.load(os.path.join(GOLD, "softmax.npz"))
    assert np.array_equal(utils.stable_softmax(s["x"][2]), s["y"][2])


def test_pose_errors_vectorised():
    from ursonet_amd.pose import pose_errors
    from oracle import pose_math as P
    rng = np.random.default_rng(0)
    q1 = rng.normal(size=(5, 4)); q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    q2 = rng.normal(size=(5, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    t1, t2 = rng.normal(size=(5, 3)), rng.normal(size=(5, 3)) + 10
    ang, le, esa = pose_errors(t1, q1, t2, q2)
    for i in range(5):
        a, l, e = P.pose_errors(t1[i], q1[i], t2[i], q2[i])
        assert abs(ang[i] - a) < 1e-9 and abs(le[i] - l) < 1e-12 and abs(esa[i] - e) < 1e-9


# ------------------------------------------------------------------ resize / clr
def test_resize_image_geometry_matches_reference():
    from ursonet_amd import utils
    for c in META["resize_geometry"]:
        img = np.zeros((c["h"], c["w"], 3), dtype=np.uint8); img[c["h"] // 2, c["w"] // 2] = 255
        out, window, scale, padding, crop = utils.resize_image(img, min_dim=c["min_dim"], max_dim=c["max_dim"], min_scale=0, mode=c["mode"])
        assert list(out.shape) == c["out_shape"] and list(window) == c["window"] and float(scale) == c["scale"]
        assert [list(p) for p in padding] == c["padding"]
        assert out[window[0] + c["h"] // 2, window[1] + c["w"] // 2, 0] == 255 and out.dtype == np.uint8


def test_clr_triangular_matches_reference_callback():
    """utils.clr_triangular (what UrsoNet.train feeds Engine.set_lr per batch) against the learning rates the reference's
    CyclicLR callback put in force batch by batch (tests/golden/clr.json, generated by importing clr_callback.py)."""
    from ursonet_amd.utils import clr_triangular
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "clr.json")))
    for g in gold.values():
        for i, lr in enumerate(g["lr"]):
            assert clr_triangular(i, g["base_lr"], g["max_lr"], g["step_size"]) == pytest.approx(lr, rel=1e-12, abs=1e-18), i


def test_clr_triangular_shape():
    from ursonet_amd.utils import clr_triangular
    assert clr_triangular(0, 1e-4, 5e-4, 100) == pytest.approx(1e-4)
    assert clr_triangular(100, 1e-4, 5e-4, 100) == pytest.approx(5e-4)
    assert clr_triangular(150, 1e-4, 5e-4, 100) == pytest.approx(3e-4)
    assert clr_triangular(200, 1e-4, 5e-4, 100) == pytest.approx(1e-4)


# ------------------------------------------------------------------ UrsoNet host plumbing (no engine)
def test_ursonet_logdir_checkpoint_and_find_last(tmp_path):
    from ursonet_amd import net
    cfg = make_config("resnet18", 64, 64, regress_ori=True)
    cfg.NAME = "Soyuz"
    m = net.UrsoNet("training", cfg, str(tmp_path), build_engine=False)
    assert m.epoch == 0 and os.path.basename(m.log_dir).startswith("soyuz2")
    assert m.checkpoint_path.endswith("weights_soyuz_{epoch:04d}.h5")
    assert m.find_last() == (None, None)
    d = tmp_path / "soyuz20250101T0000"; d.mkdir()
    (d / "weights_soyuz_0003.npz").write_bytes(b""); (d / "weights_soyuz_0012.npz").write_bytes(b""); (d / "events.x").write_bytes(b"")
    ddir, ck = m.find_last()
    assert ck.endswith("weights_soyuz_0012.npz") and ddir == str(d)
    assert m.get_last_checkpoint("soyuz20250101T0000")[1] == ck
    m.set_log_dir(ck)
    assert m.epoch == 12 and m.log_dir == str(d)
    m.set_log_dir(str(d / "weights_soyuz_0007.h5"))
    assert m.epoch == 7
    with pytest.raises(AssertionError):
        net.UrsoNet("evaluate", cfg, str(tmp_path), build_engine=False)
    with pytest.raises(IOError):
        m.get_imagenet_weights("resnet50")
    layer = m.keras_model.get_layer("stage1_unit1_bn2")
    assert layer.weights == ["stage1_unit1_bn2/%s:0" % w for w in ("gamma", "beta", "moving_mean", "moving_variance")]


def test_weights_file_roundtrip_npz(tmp_path):
    from ursonet_amd import net
    from ursonet_amd.engine import initial_weights
    from ursonet_amd.graph import build_graph
    W = initial_weights(build_graph(make_config("resnet18", 64, 64, regress_ori=True)), 1, True)
    written = net.write_weights_file(str(tmp_path / "weights_x_0001.h5"), W)
    R = net.read_weights_file(written[0])
    assert list(R) == list(W) and all(np.array_equal(R[l][w], W[l][w]) for l in W for w in W[l])
    assert R["conv0"]["kernel"].shape == (7, 7, 3, 64) and "bias" not in R["conv0"]


def test_h5_to_npz_converter_against_a_stand_in_h5py(tmp_path, monkeypatch):
    """tools/h5_to_npz.py (the offline route for the released Keras .h5 files: h5py is absent from this image, so its h5py calls run
    against a minimal stand-in): flat and nested ('model_weights') files, bytes attributes, 'layer/weight:0' dataset names -> the .npz
    twin that read_weights_file / UrsoNet.load_weights take, values bit-equal, layer order kept."""
    import importlib.util
    import sys
    import types
    from ursonet_amd import net
    from ursonet_amd.engine import initial_weights
    from ursonet_amd.graph import build_graph

    class Node(dict):
        def __init__(self):
            super(Node, self).__init__()
            self.attrs = {}
    store = {}

    class File(object):
        def __init__(self, path, mode="r"):
            self.root = store[path]

        def __enter__(self):
            return self.root

        def __exit__(self, *exc):
            return False
    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    W = initial_weights(build_graph(make_config("resnet18", 64, 64, regress_ori=True)), 2, True)
    root = Node()
    root.attrs["layer_names"] = [ln.encode("utf8") for ln in W]
    for ln, ws in W.items():
        g = Node(); g.attrs["weight_names"] = [("%s/%s:0" % (ln, wn)).encode("utf8") for wn in ws]
        for wn, a in ws.items():
            g["%s/%s:0" % (ln, wn)] = a
        root[ln] = g
    nested = Node(); nested["model_weights"] = root
    store["flat.h5"], store["full_model.h5"] = root, nested
    spec = importlib.util.spec_from_file_location("h5_to_npz", os.path.join(ROOT, "tools", "h5_to_npz.py"))
    conv = importlib.util.module_from_spec(spec); spec.loader.exec_module(conv)
    for src in ("flat.h5", "full_model.h5"):
        dst = str(tmp_path / (src[:-3] + ".npz"))
        assert conv.main([src, dst]) == 0
        R = net.read_weights_file(dst)
        assert list(R) == list(W) and all(np.array_equal(R[l][w], W[l][w]) and R[l][w].dtype == W[l][w].dtype for l in W for w in W[l])


def test_build_rebuilds_on_source_hash_not_mtime(tmp_path, monkeypatch):
    """ursonet_amd/build.py: the library is current iff the SHA-256 of its sources, header, flags and compiler version recorded next
    to it (lib/liburso_hip.so.srchash) matches -- file times (a fresh checkout, a pushed snapshot) play no part."""
    from ursonet_amd import build
    build.build(verbose=False)                                     # a no-op when the library is current
    assert os.path.exists(build.LIB) and not build.needs_build()
    h = build.source_hash()
    assert open(build.LIB + ".srchash").read().strip() == h and len(h) == 64
    os.utime(os.path.join(build.CSRC, "common.h"), None)          # a newer mtime alone changes nothing
    assert not build.needs_build()
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DURSO_TEST_FLAG=1"])
    assert build.source_hash() != h and build.needs_build()        # different flags (or sources): stale


def test_data_generator_and_mold_image():
    from ursonet_amd import net
    from ursonet_amd.dataset import SyntheticPoses
    cfg = make_config("resnet50", 64, 128, batch=3, regress_ori=False, ori_bins=4)
    cfg.ROT_AUG = False
    ds = SyntheticPoses(7, 64, 128, cfg, seed=1)
    gen = net.data_generator(ds, cfg, shuffle=False, batch_size=3)
    (imgs, meta, locs, oris), outs = next(gen)
    assert imgs.shape == (3, 64, 128, 3) and imgs.dtype == np.float32 and meta.shape == (3, cfg.IMAGE_META_SIZE) and outs == []
    assert locs.shape == (3, 3) and oris.shape == (3, 64) and np.allclose(oris.sum(1), 1, atol=1e-5)
    raw = ds.load_image(0)
    assert np.allclose(imgs[0], raw.astype(np.float32) - cfg.MEAN_PIXEL)              # mold_image net.py:1346
    # unmold_image truncates (astype(uint8), net.py:1355): off by at most one grey level in float32
    assert np.abs(net.unmold_image(imgs[0], cfg).astype(int) - raw.astype(int)).max() <= 1
    img, m, loc, ori = net.load_image_gt(ds, cfg, 2)
    assert m[0] == 2 and tuple(m[1:4]) == (64, 128, 3) and tuple(m[7:11]) == (0, 0, 64, 128)
    # keypoint mode yields five inputs (net.py:540-545)
    cfgk = make_config("resnet18", 64, 128, batch=2, keypoints=True)
    dsk = SyntheticPoses(5, 64, 128, cfgk, seed=2)
    (imgs, meta, locs, k1, k2), _ = next(net.data_generator(dsk, cfgk, shuffle=False, batch_size=2))
    assert imgs.shape == (2, 64, 128, 3) and k1.shape == (2, 3) and k2.shape == (2, 3)


def test_feeder_shards_every_global_batch_over_the_ranks():
    """Data-parallel feeding (ursonet_amd.feeder.batches, rank= / world=): the ranks walk ONE shuffled order, their shards of a global
    batch are disjoint and rank-ordered, and together they are what a single process drawing world x batch samples with the same private
    shuffle would have drawn -- over several epochs of a dataset whose size is no multiple of the global batch."""
    from ursonet_amd import feeder
    from ursonet_amd.dataset import SyntheticPoses
    cfg = make_config("resnet18", 64, 128, batch=2, regress_ori=False, ori_bins=4)
    cfg.ROT_AUG = False
    ds = SyntheticPoses(11, 64, 128, cfg, seed=1)
    world, bs, nb = 3, 2, 9
    per_rank = []
    for r in range(world):
        g = feeder.batches(ds, cfg, True, bs, molded=False, rank=r, world=world)
        per_rank.append([[int(m[0]) for m in next(g).meta] for _ in range(nb)])       # image_meta[0] = image_id (net.py:1314)
    # the single-process order under the same private generator
    rng, ids, cursor, want = np.random.RandomState(feeder.DP_SHUFFLE_SEED), np.copy(ds.image_ids), -1, []
    for _ in range(nb * world * bs):
        cursor = (cursor + 1) % len(ids)
        if cursor == 0:
            rng.shuffle(ids)
        want.append(int(ids[cursor]))
    for k in range(nb):
        glob = sum((per_rank[r][k] for r in range(world)), [])
        assert glob == want[k * world * bs:(k + 1) * world * bs], (k, glob)
    # world == 1 keeps the reference's generator: NumPy's global RNG shuffles (net.py:489-490)
    np.random.seed(5)
    g = feeder.batches(ds, cfg, True, bs, molded=False)
    got = [int(m[0]) for _ in range(3) for m in next(g).meta]
    np.random.seed(5)
    ids = np.copy(ds.image_ids); np.random.shuffle(ids)
    assert got == [int(i) for i in ids[:6]]


def test_data_generator_skips_up_to_five_bad_samples():
    """net.py:553-559: a failing sample is logged and skipped; the sixth failure re-raises."""
    from ursonet_amd import net
    from ursonet_amd.dataset import SyntheticPoses
    cfg = make_config("resnet50", 64, 128, batch=2, regress_ori=True)

    class Flaky(SyntheticPoses):
        bad = {1, 2}

        def load_image(self, image_id):
            if int(image_id) in self.bad:
                raise IOError("corrupt frame %d" % image_id)
            return super(Flaky, self).load_image(image_id)
    ds = Flaky(6, 64, 128, cfg, seed=3)
    (imgs, meta, locs, oris), _ = next(net.data_generator(ds, cfg, shuffle=False, batch_size=2))
    assert [int(m[0]) for m in meta] == [0, 3]                        # samples 1 and 2 were skipped
    ds.bad = set(range(6))
    with pytest.raises(RuntimeError):
        next(net.data_generator(ds, cfg, shuffle=False, batch_size=2))


def test_sim2real_draws_follow_the_reference_distributions():
    """Host half of the sim2real branch (net.py:390-406): p = 0.5 dice, a permutation of the five stages, parameter ranges of the
    imgaug operators -- CoarseDropout's p is a CHOICE between 0.0 and 0.03 (a list), its mask 2-10 % of the frame size."""
    from ursonet_amd import augment
    rng = np.random.RandomState(0)
    d = augment.sim2real_draw(800, 480, 640, rng)
    assert 0.4 < d["apply"].mean() < 0.6
    assert all(sorted(o) == [0, 1, 2, 3, 4] for o in d["order"])
    assert np.all(d["par"][:, 0, 0] == np.float32(2.55))
    ap = d["apply"]                                                   # stage parameters are drawn only for the samples the dice select (imgaug: at to_deterministic())
    par = d["par"][ap]
    assert np.all(d["par"][~ap][:, 1:4, 0] == 0) and all(m.shape == (1, 1) and not m.any() for m, a_ in zip(d["masks"], ap) if not a_)
    assert par[:, 1, 0].min() >= 0 and par[:, 1, 0].max() <= 1.5 and par[:, 1, 0].std() > 0.3
    assert set(np.unique(par[:, 2, 0])) <= set(range(-20, 21)) and par[:, 2, 0].min() < -15 and par[:, 2, 0].max() > 15
    assert par[:, 3, 0].min() >= 0.5 and par[:, 3, 0].max() <= 2.0
    dh, dw = par[:, 4, 0], par[:, 4, 1]
    assert dh.min() >= 9 and dh.max() <= 48 and dw.min() >= 12 and dw.max() <= 64
    frac = np.array([m.mean() for m, a_ in zip(d["masks"], ap) if a_])
    assert (frac == 0).mean() > 0.3 and 0.01 < frac[frac > 0].mean() < 0.06


def test_resize_image_matches_the_reference_run_with_the_real_scikit_image():
    """utils.resize_image (utils.py:398-511) on uint8 frames against what THE REFERENCE'S function returned with the real scikit-image 0.18.3
    (tests/golden/resize_skimage.npz, generated by tests/golden/make_resize_golden.py under the interpreter of this image that has skimage):
    window, scale and padding exactly; pure-padding cases bit for bit; rescaled images within ONE grey level on at most 2 % of the pixels
    (skimage's anti-aliasing filter runs in uint8 and truncates after each axis -- reproduced; its warp estimates the affine map by least
    squares, so a bilinear value that is an exact integer here can sit 1e-13 below it there and truncate one lower)."""
    import os
    from ursonet_amd import utils
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize_skimage.npz"))
    assert str(z["skimage_version"]) == "0.18.3"
    for case in z["cases"]:
        name, mode = str(case).split()
        a = z[name + "/args"]
        out, window, scale, padding, crop = utils.resize_image(z[name + "/in"], min_dim=int(a[0]), max_dim=int(a[1]) or None, min_scale=float(a[2]) or None, mode=mode)
        ref = z[name + "/out"]
        assert out.shape == ref.shape and out.dtype == ref.dtype == np.uint8 and crop is None
        assert tuple(window) == tuple(z[name + "/window"]) and float(scale) == float(z[name + "/scale"])
        assert np.array_equal(np.asarray(padding), z[name + "/padding"])
        d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
        if float(scale) == 1.0:
            assert d.max() == 0, name
        else:
            assert d.max() <= 1 and (d > 0).mean() <= 0.02, (name, int(d.max()), float((d > 0).mean()))


def test_resize_compat_switch_selects_the_skimage_generation(monkeypatch):
    """URSO_RESIZE_COMPAT: 0.18 (default) truncates integer frames after each anti-aliasing pass, as scikit-image <= 0.18 does
    (golden above); 0.19 smooths in float like scikit-image >= 0.19 -- a shrunk uint8 frame then equals the float frame's result truncated
    once at the end, is never darker than the 0.18 result and differs from it by at most two grey levels (one truncation per smoothed axis)."""
    from ursonet_amd import utils
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
    monkeypatch.delenv("URSO_RESIZE_COMPAT", raising=False)
    old = utils.resize_image(img, min_dim=96, max_dim=128, mode="square")[0]
    monkeypatch.setenv("URSO_RESIZE_COMPAT", "0.19")
    new = utils.resize_image(img, min_dim=96, max_dim=128, mode="square")[0]
    flt = utils.resize_image(img.astype(np.float64), min_dim=96, max_dim=128, mode="square")[0]
    assert new.dtype == np.uint8 and np.array_equal(new, flt.astype(np.uint8))
    d = new.astype(int) - old.astype(int)
    assert d.min() >= 0 and 1 <= d.max() <= 2
    monkeypatch.setenv("URSO_RESIZE_COMPAT", "1.0")
    with pytest.raises(ValueError):
        utils.resize_image(img, min_dim=96, max_dim=128, mode="square")


def test_resize_antialiasing_and_identity():
    """utils._bilinear_resize: identity at equal size; when shrinking, the Gaussian pre-filter (sigma = (s - 1)/2, skimage's
    anti_aliasing default) keeps a one-pixel checkerboard from aliasing into a constant pattern of the wrong mean."""
    from ursonet_amd import utils
    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, size=(12, 16, 3)).astype(np.float64)
    assert np.allclose(utils._bilinear_resize(img, 12, 16), img)
    chk = (np.indices((64, 64)).sum(0) % 2 * 255.0)[:, :, None].repeat(3, 2)
    aa = utils._bilinear_resize(chk, 32, 32)[4:-4, 4:-4]
    noaa = utils._bilinear_resize(chk, 32, 32, anti_aliasing=False)[4:-4, 4:-4]
    assert abs(aa.mean() - 127.5) < 2 and aa.std() < 12
    assert noaa.std() < 1e-9 or noaa.std() > aa.std()                # point sampling: a constant (aliased) or high-contrast pattern
    out, window, scale, padding, crop = utils.resize_image(chk.astype(np.uint8), min_dim=64, max_dim=64, min_scale=0, mode="square")
    assert out.shape == (64, 64, 3) and scale == 1 and window == (0, 0, 64, 64)


def test_augment_host_math_matches_reference_goldens():
    """ursonet_amd.augment's se3 helpers against the reference's own outputs (tests/golden/se3lib_basic.npz)."""
    from ursonet_amd import augment as A
    from ursonet_amd.dataset import Camera
    from oracle import pose_math as P
    g = np.load(os.path.join(ROOT, "tests", "golden", "se3lib_basic.npz"))
    for i in range(len(g["eul"])):
        R = A.euler2SO3_left(*g["eul"][i])
        assert np.array_equal(R, g["e2R"][i])
        assert np.allclose(A.SO32quat(g["e2R"][i]), g["R2q"][i], rtol=0, atol=1e-15)
        assert np.allclose(A.quat_mult(g["qs"][i], g["qs2"][i]), g["qmul"][i], rtol=0, atol=1e-15)
        t = np.array([0.3, -0.2, 7.0])
        tn, qn = A.rotate_pose(t, g["qs2"][i], R)
        assert np.allclose(tn, t @ g["e2R"][i].T) and abs(np.linalg.norm(tn) - np.linalg.norm(t)) < 1e-12
        assert np.allclose(qn, P.quat_mult(P.SO32quat(R), g["qs2"][i]), atol=1e-15)
        assert np.array_equal(A.quat2SO3(g["qs"][i]), g["q2R"][i])            # the reference's own quat2SO3 outputs
        k1, k2 = A.encode_as_keypoints(g["qs"][i], t)
        r1, r2 = P.encode_as_keypoints(g["qs"][i], t)
        assert k1.shape == (3, 1) and np.allclose(k1, r1, atol=1e-15) and np.allclose(k2, r2, atol=1e-15)
        assert np.allclose(k1[:, 0] - t, g["q2R"][i][:, 2]) and np.allclose(k2[:, 0] - t, g["q2R"][i][:, 1])
    K1, K2 = A.encode_as_keypoints(g["qs"][:5], np.tile(np.array([0.3, -0.2, 7.0]), (5, 1)))
    assert K1.shape == (5, 3) and K1.dtype == np.float32 and np.allclose(K2[2] - [0.3, -0.2, 7.0], g["q2R"][2][:, 1], atol=1e-6)
    cam = Camera()                                                           # urso.py:12-22
    assert abs(cam.fx - 640.0) < 1e-9 and cam.fy < 0 and np.allclose(cam.K[:2, 2], [640, 480])
    M = A.rotation_homography(cam.K, np.eye(3))
    assert np.allclose(M, np.eye(3), atol=1e-12)


# ------------------------------------------------------------------ C ABI
def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "ursonet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(urso_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ursonet_amd.hip as hip
    syms = _header_symbols()
    assert len(syms) >= 24
    lib = ctypes.CDLL(hip.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "symbol %s declared in include/ursonet_hip.h is not exported" % s
    assert set(syms) == set(hip.EXPORTED_SYMBOLS), set(syms) ^ set(hip.EXPORTED_SYMBOLS)
    assert hip._lib.urso_abi_version() == 9


def test_policy_options_are_explicit_and_never_read_the_environment():
    """urso_set_option / urso_get_option (include/ursonet_hip.h): compiled-in defaults, unknown names refused, and no getenv
    anywhere in the C sources."""
    import ursonet_amd.hip as hip
    assert hip.get_option("pw_small") == 5 and hip.get_option("grid_cap") == 0 and hip.get_option("wgrad_blocks") == 512
    with hip.options(grid_cap=24, pw_small=1):
        assert hip.get_option("grid_cap") == 24 and hip.get_option("pw_small") == 1
    assert hip.get_option("grid_cap") == 0 and hip.get_option("pw_small") == 5
    with pytest.raises(hip.UrsoHipError):
        hip.set_option("no_such_option", 1)
    with pytest.raises(hip.UrsoHipError):
        hip.set_option("wgrad_blocks", 0)
    csrc = os.path.join(ROOT, "ursonet_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f), errors="replace").read(), f


def test_cabi_argument_validation_without_gpu():
    import ursonet_amd.hip as hip
    g = hip.geom(2, 8, 8, 12, 8, 8, 16, 3, 3, 1, 1, 1, 1)                 # C=12 is not a 16-byte multiple in bf16
    rc = hip._lib.urso_conv_igemm(ctypes.byref(g), hip.BF16, 0, 1, 1, None, None, None, 1, None)
    assert rc == -1 and "multiple of 8" in hip.last_error()
    rc = hip._lib.urso_conv_igemm(ctypes.byref(g), hip.BF16, 0, None, None, None, None, None, None, None)
    assert rc == -1 and "null" in hip.last_error()
    g2 = hip.geom(32, 128, 160, 64, 128, 160, 256, 1, 1)
    ws = hip.conv_wgrad_ws_bytes(g2, hip.BF16)
    assert ws >= 64 * 256 * 4 and ws < (1 << 31)
    assert hip.param_grad_finalize_ws_bytes(4608, 512) > 0 and hip.sqnorm_ws_bytes(10 ** 6) >= 4
    assert hip._lib.urso_sgd_momentum_clip(16, None, None, None, None, None, None) == -1


def test_grouped_weight_gradient_plans_are_consistent_without_gpu():
    """urso_wgrad_group_plan / urso_conv_wgrad_pair_splits are host code: for groups like the engine's (ResNet-50 / 101 stages 3-5 at
    cfg2 and at a small batch, mixed geometries, both tile shapes) the plan never asks for more blocks than stay resident, covers every
    pixel of every layer, hands out every (layer, work id) exactly once in the block map, never splits a layer into blocks of fewer
    than 8 steps, and reports its fill; a pair of 3x3 layers shares the CUs without exceeding them."""
    import ursonet_amd.hip as hip
    dt = hip.BF16
    pw = lambda M, C, N: hip.geom(1, 1, M, C, 1, M, N, 1, 1)
    groups = {
        "stage4 x8": [pw(40960, 256, 1024), pw(40960, 1024, 256)] * 4,
        "stage5 + entry": [pw(10240, 512, 2048), pw(10240, 2048, 512), pw(10240, 1024, 2048), hip.geom(32, 32, 40, 1024, 16, 20, 512, 1, 1, 2, 2, 0, 0)],
        "stage3": [pw(163840, 128, 512), pw(163840, 512, 128)] * 3 + [hip.geom(32, 64, 80, 128, 32, 40, 128, 3, 3, 2, 2, 1, 1)],
        "small batch": [pw(10240, 256, 1024), pw(10240, 512, 1024), pw(10240, 512, 256)],
        "ragged": [pw(8200, 1024, 264), pw(4160, 264, 520), hip.geom(4, 65, 81, 256, 33, 41, 328, 3, 3, 2, 2, 1, 1)],
    }
    for name, gs in groups.items():
        for big in (0, 1):
            with hip.options(wgrad_big=big):
                grp = hip.WgradGroup(gs, dt)
                assert grp.nblocks > 0, name
                is_big = bool(grp.host[0].mode & 4)
                assert all(bool(it.mode & 4) == is_big for it in grp.host) and (is_big <= bool(big))
                ts = 256 if is_big else 128
                assert grp.nblocks <= (256 if is_big else 512) and 0.0 < grp.fill <= 1.0, (name, grp.nblocks, grp.fill)
                buf = (ctypes.c_int32 * (2 * grp.nblocks))()
                assert hip._lib.urso_wgrad_group_plan(grp.n, grp.host, dt, buf, grp.nblocks) == grp.nblocks
            seen = set()
            for b in range(grp.nblocks):
                seen.add((buf[2 * b], buf[2 * b + 1]))
            want = set()
            for i, (it, g) in enumerate(zip(grp.host, gs)):
                K, M = g.KH * g.KW * g.C, g.B * g.OH * g.OW
                assert it.M == M and it.ktiles == -(-K // ts) and it.ntiles == -(-g.N // ts)
                assert it.splits >= 1 and it.m_per_split % 64 == 0 and it.splits * it.m_per_split >= M > (it.splits - 1) * it.m_per_split
                assert it.m_per_split >= 8 * 64 or it.splits == 1
                want |= {(i, w) for w in range(it.ktiles * it.ntiles * it.splits)}
            assert seen == want and len(want) == grp.nblocks, name
    assert hip.WgradGroup([pw(4096, 2048, 2048)] * 5, dt).nblocks == 0                  # does not fit one residency in either tile shape
    # two 3x3 layers in one launch: stage 4 + stage 4, stage 3 + stage 4, stage 5 + stage 5
    c3 = lambda B, H, W, C, N: hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    for a, b in ((c3(32, 32, 40, 256, 256),) * 2, (c3(32, 64, 80, 128, 128), c3(32, 32, 40, 256, 256)), (c3(32, 16, 20, 512, 512),) * 2):
        sp = hip.conv_wgrad_pair_splits(a, b, dt)
        assert sp is not None
        blocks = sum(s * (g.C // 64) * (g.N // 64) for s, g in zip(sp, (a, b)))
        assert 200 <= blocks <= 256 and all(1 <= s <= hip.conv_wgrad_splits(g, dt) for s, g in zip(sp, (a, b)))
    assert hip.conv_wgrad_pair_splits(c3(32, 32, 40, 256, 256), c3(32, 128, 160, 64, 64), dt) is None      # the 64-channel layer has a kernel of its own


# ------------------------------------------------------------------ layout rules
def test_product_never_imports_the_oracle_or_reference():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ursonet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, "product files reference oracle/ or /root/reference: %s" % bad


def test_bench_uses_oracle_only_in_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle", src)]
    i0 = src.index("def cpu_baseline"); i1 = src.index("def main")
    assert uses and all(i0 < u < i1 for u in uses)


def test_c_abi_header_is_plain_c99(tmp_path):
    """include/ursonet_hip.h is the boundary for non-Python hosts: it must compile as strict C99 (no C++ or HIP types), and a C
    translation unit that takes the address of every declared entry point must compile against it."""
    import re
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = open(os.path.join(ROOT, "include", "ursonet_hip.h")).read()
    names = sorted(set(re.findall(r"\b(urso_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 58
    src = tmp_path / "abi.c"
    src.write_text('#include "ursonet_hip.h"\n#include <stddef.h>\ntypedef void (*fn)(void);\nfn table[] = {\n' +
                   "".join("    (fn)%s,\n" % n for n in names) + "};\nsize_t count(void) { return sizeof(table) / sizeof(table[0]); }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-Wno-pedantic", "-c", str(src), "-I", os.path.join(ROOT, "include"),
                        "-o", str(tmp_path / "abi.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_augmentation_consumes_the_global_generator_in_the_reference_order():
    """load_image_gt (net.py:390-438) draws from NumPy's global generator per sample: the sim2real dice (net.py:395), then the rotation
    dice (net.py:415) and, for a camera rotation, three angles (utils.py:33).  augment_samples batches the pixel work but must consume
    that stream identically (the imgaug stage parameters come from imgaug's own generator in the reference: a separate one here), so
    the same seed gives the same poses.  Checked without a GPU by replaying the stream sample by sample (no sample here ends up with
    pixel work: ROT_AUG dice forced <= 0.5 by the choice of seed would be fragile, so the pixel calls are stubbed)."""
    from ursonet_amd import augment, feeder

    class Cfg(object):
        SIM2REAL_AUG, ROT_AUG, ROT_IMAGE_AUG, REGRESS_LOC, REGRESS_ORI, REGRESS_KEYPOINTS = True, True, False, True, True, False
        ORIENTATION_PARAM, BETA = "quaternion", 3
    n = 9
    samples = [feeder.Sample(i, np.zeros((8, 8, 3), np.uint8), np.array([0.0, 0.0, 10.0]), np.array([0.0, 0.0, 0.0, 1.0]), None, None) for i in range(n)]
    seen = {}

    def fake_sim2real_batch(images, draw=None, rng=None):
        seen["apply"] = draw["apply"].copy()

        class T(object):
            def cpu(self):
                return self

            def numpy(self):
                return np.asarray(images)
        return T()

    def fake_rotate(images, t, q, K, pyr):
        seen["pyr"] = np.array(pyr)

        class T(object):
            def cpu(self):
                return self

            def numpy(self):
                return np.asarray(images)
        return T(), np.asarray(t), np.asarray(q)
    orig = augment.sim2real_batch, augment.rotate_cam_batch
    augment.sim2real_batch, augment.rotate_cam_batch = fake_sim2real_batch, fake_rotate
    try:
        class DS(object):
            class camera(object):
                K = np.eye(3)
        np.random.seed(123)
        feeder.augment_samples(samples, DS(), Cfg())
        after = np.random.rand(1)[0]
    finally:
        augment.sim2real_batch, augment.rotate_cam_batch = orig
    np.random.seed(123)
    apply, pyr = [], []
    for i in range(n):                                   # the reference's order, one sample at a time
        apply.append(np.random.rand(1)[0] > 0.5)
        if np.random.rand(1)[0] > 0.5:
            pyr.append((np.random.rand(3) - 0.5) * 20)
    assert np.array_equal(seen["apply"], np.array(apply)) and 0 < sum(apply) < n
    assert np.allclose(seen["pyr"], np.array(pyr)) and 0 < len(pyr) < n
    assert after == np.random.rand(1)[0]                 # and nothing else was drawn from the global stream


def test_resize_equals_scipy_ndimage_the_backend_of_skimage_resize():
    """utils.resize_image's scale != 1 branch (utils.py:457-459: skimage.transform.resize(order=1, mode='constant', preserve_range=True)).
    skimage is not installed here, but its resize (>= 0.19) is two scipy.ndimage calls -- gaussian_filter with sigma = (in/out - 1)/2 per
    shrinking axis (anti_aliasing, the default), then zoom(order=1, grid_mode=True, mode='grid-constant', cval=0) -- and scipy IS
    installed: the product's own bilinear code must reproduce exactly that, on shrinking, growing and mixed factors (real URSO / SPEED
    frames are reduced 2-3x on this path)."""
    import scipy.ndimage as ndi
    from ursonet_amd.utils import _bilinear_resize, resize_image
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, size=(96, 128, 3))
    for oh, ow in ((48, 64), (40, 50), (60, 100), (120, 160), (37, 200), (96, 128)):
        fac = np.array([96 / oh, 128 / ow, 1.0])
        sig = np.maximum(0, (fac - 1) / 2)
        f = ndi.gaussian_filter(img, sig, cval=0, mode="grid-constant") if (sig > 0).any() else img
        ref = ndi.zoom(f, [1 / x for x in fac], order=1, mode="grid-constant", cval=0, grid_mode=True)
        assert ref.shape == (oh, ow, 3)
        assert np.abs(_bilinear_resize(img, oh, ow) - ref).max() < 1e-9
    # through the public function: a 1200 x 1920-like frame at image_scale 0.5 (cfg5), uint8 in / uint8 out, pad64
    u8 = rng.integers(0, 256, size=(150, 240, 3), dtype=np.uint8)
    out, window, scale, padding, crop = resize_image(u8, min_dim=64, max_dim=128, mode="pad64")
    assert scale == 64 / 150 and out.dtype == np.uint8 and out.shape[0] % 64 == 0 and out.shape[1] % 64 == 0
    nh, nw = round(150 * scale), round(240 * scale)
    fac = np.array([150 / nh, 240 / nw, 1.0])
    # (scikit-image <= 0.18 -- the reference's era -- hands the frame to gaussian_filter in its own dtype: uint8 in, uint8 out, truncated after each
    # axis; tests/golden/resize_skimage.npz holds the real thing's outputs)
    ref = ndi.zoom(ndi.gaussian_filter(u8, np.maximum(0, (fac - 1) / 2), cval=0, mode="grid-constant").astype(np.float64), [1 / x for x in fac], order=1,
                   mode="grid-constant", cval=0, grid_mode=True).astype(np.uint8)
    d = np.abs(out[window[0]:window[2], window[1]:window[3]].astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3      # (the filtered frame is integer-valued: a bilinear value that is an exact integer in one arithmetic can sit 1e-13 below it in the other)


def test_algorithmic_bytes_charge_compact_and_sampled_tensors_at_their_real_size():
    """The figures bench.py's roofline prices a launch with (urso_conv_igemm_algorithmic; host arithmetic).  A scattered destination -- the
    compact stage-boundary data gradient of res{2c,3d,4f}_branch2c -- is written, and its mask read, at the B x OH x OW computed pixels only
    (that launch was charged the dense 4x tensor and showed 1.26 of the HBM roof); a strided pointwise layer reads only the
    sampled pixels of its input.  No launch class of the cfg2 step may exceed its roof at ANY duration its bytes allow: the charged bytes
    never exceed input + filter + output + operands of the tensors as they are stored."""
    import ursonet_amd.hip as hip
    B, es = 32, 2
    # dgrad of res2c_branch2c on the compact gradient: dz [B, 64, 80, 256] -> dX at the even pixels of [B, 128, 160, 64], masked
    g = hip.geom(B, 64, 80, 256, 64, 80, 64, 1, 1, 1, 1, 0, 0, 1, 1, FH=128, FW=160, OSH=2, OSW=2)
    fl, by = hip.conv_igemm_algorithmic(g, hip.BF16, 0, has_add=False, has_mask=True)
    M = B * 64 * 80
    assert fl == 2.0 * M * 256 * 64
    assert by == M * 256 * es + 256 * 64 * es + 2 * M * 64 * es                   # dz + filter + (written dX + mask at the written pixels): 126 MB, not 252
    dense = hip.geom(B, 128, 160, 256, 128, 160, 64, 1, 1)
    assert hip.conv_igemm_algorithmic(dense, hip.BF16, 0, False, True)[1] == 4 * M * 256 * es + 256 * 64 * es + 2 * 4 * M * 64 * es
    # stride-2 pointwise entry layer: reads a quarter of its input
    s2 = hip.geom(B, 128, 160, 256, 64, 80, 128, 1, 1, 2, 2)
    assert hip.conv_igemm_algorithmic(s2, hip.BF16, 0)[1] == M * 256 * es + 128 * 256 * es + M * 128 * es
    # bit masks: one byte per 8 outputs, read (MASK_BITS) or written (EMIT_BITS)
    pw = hip.geom(B, 64, 80, 128, 64, 80, 512, 1, 1)
    base = hip.conv_igemm_algorithmic(pw, hip.BF16, 0)[1]
    assert hip.conv_igemm_algorithmic(pw, hip.BF16, hip.EPI_EMIT_BITS, has_add=True)[1] == base + M * 512 * es + M * 512 / 8
    assert hip.conv_igemm_algorithmic(pw, hip.BF16, hip.EPI_MASK_BITS, has_mask=True)[1] == base + M * 512 / 8
    # every conv geometry of the cfg2 plan: charged bytes <= the stored tensors it touches (so measured time >= bytes / 8 TB/s keeps frac <= 1)
    for (h, w, c, n, k, s) in [(256, 320, 8, 64, 7, 2), (128, 160, 64, 64, 3, 1), (128, 160, 64, 256, 1, 1), (64, 80, 128, 128, 3, 1), (64, 80, 512, 128, 1, 1),
                               (32, 40, 256, 256, 3, 1), (32, 40, 1024, 256, 1, 1), (16, 20, 512, 512, 3, 1), (16, 20, 512, 2048, 1, 1), (16, 20, 2048, 32, 3, 2)]:
        oh, ow = -(-h // s), -(-w // s)
        gg = hip.geom(B, h, w, c, oh, ow, n, k, (4 if k == 7 else k), s, s, k // 2, k // 2)
        stored = B * h * w * c * es + n * k * (4 if k == 7 else k) * c * es + 3 * B * oh * ow * n * es
        assert hip.conv_igemm_algorithmic(gg, hip.BF16, 0, True, True)[1] <= stored
