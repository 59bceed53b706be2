#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's own NumPy modules.

Runs only in the build container (needs /root/reference); the outputs
(tests/golden/*.npz, *.json) are committed and are the only thing that travels.
Modules `utils`, `speed` hard-import tensorflow/skimage/cv2 at module scope but the
functions called here never touch them, so empty stub modules are injected.

    python tests/golden/make_golden.py
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def import_reference():
    for n in ["tensorflow", "skimage", "skimage.color", "skimage.io", "skimage.transform", "cv2"]:
        if n not in sys.modules:
            _stub(n)
    sys.modules["skimage"].color = sys.modules["skimage.color"]
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import se3lib, utils, config  # noqa
    return se3lib, utils, config


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def random_unit_quats(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[q[:, 3] < 0] *= -1
    return q


def main():
    se3lib, utils, config = import_reference()
    rng = np.random.default_rng(20250928)

    # ---- 1. se3lib elementary conversions ---------------------------------------------
    eul = np.concatenate([rng.uniform([-180, -90, -180], [180, 90, 180], size=(60, 3)),
                          np.array([[0, 0, 0], [180, 90, 180], [-180, -90, -180], [45, 90, 10.]])])
    e2q = np.stack([np.asarray(se3lib.euler2quat(*e)).ravel() for e in eul])
    e2R = np.stack([np.asarray(se3lib.euler2SO3_left(*e)) for e in eul])
    R2q = np.stack([np.asarray(se3lib.SO32quat(np.asarray(R)), dtype=np.float64) for R in e2R])
    qs = random_unit_quats(rng, 64)
    q2R = np.stack([np.asarray(se3lib.quat2SO3(q)) for q in qs])
    qs2 = random_unit_quats(rng, 64)
    qmul = np.stack([np.asarray(se3lib.quat_mult(a, b)).ravel() for a, b in zip(qs, qs2)])
    qang = np.array([np.asarray(se3lib.angle_between_quats(a, b)).item() for a, b in zip(qs, qs2)])
    np.savez_compressed(os.path.join(OUT, "se3lib_basic.npz"), eul=eul, e2q=e2q, e2R=e2R, R2q=R2q,
                        qs=qs, qs2=qs2, q2R=q2R, qmul=qmul, qang=qang)

    # ---- 2/3. encode_ori, encode_ori_fast, quat_weighted_avg --------------------------
    min_lim, max_lim = np.array([-180, -90, -180]), np.array([180, 90, 180])
    beta = config.Config.BETA
    store = {}
    meta = {"beta": beta, "min_lim": min_lim.tolist(), "max_lim": max_lim.tolist()}
    qs8 = random_unit_quats(rng, 8)
    store["oris"] = qs8
    for n in (4, 8, 16):
        enc, Hq, red = quiet(utils.encode_ori, qs8, n, beta, min_lim, max_lim)
        store[f"enc_{n}"] = enc
        store[f"Hquat_{n}"] = Hq
        store[f"red_{n}"] = red
        fast = np.stack([quiet(utils.encode_ori_fast, q, beta, Hq, red) for q in qs8])
        store[f"fast_{n}"] = fast
        if n in (8, 16):
            # weighted average of the encoded PMFs and of softmax(random logits)
            logits = rng.normal(size=(4, n ** 3)).astype(np.float32) * 3
            store[f"logits_{n}"] = logits
            pm = np.stack([utils.stable_softmax(l) for l in logits])
            store[f"softmax_{n}"] = pm
            allw = np.concatenate([enc[:4].astype(np.float64), pm.astype(np.float64)])
            qa, AA = [], []
            for w in allw:
                q, Hinv = se3lib.quat_weighted_avg(Hq, w)
                q = np.asarray(q, dtype=np.float64).ravel()
                if q[np.argmax(np.abs(q))] < 0:
                    q = -q
                qa.append(q)
                # re-derive A exactly like the reference (float32 accumulation) for the fixture
                A = np.zeros((4, 4), dtype=np.float32)
                for i in range(len(Hq)):
                    a = np.matrix([Hq[i, 0], Hq[i, 1], Hq[i, 2], Hq[i, 3]])
                    A += a.transpose() * a * w[i]
                AA.append(A)
            store[f"wavg_w_{n}"] = allw
            store[f"wavg_q_{n}"] = np.stack(qa)
            store[f"wavg_A_{n}"] = np.stack(AA)
    # n = 24: hashes only (map is 13,824 x 4)
    enc24, Hq24, red24 = quiet(utils.encode_ori, qs8[:3], 24, beta, min_lim, max_lim)
    meta["n24_map_sha256"] = hashlib.sha256(np.ascontiguousarray(Hq24).tobytes()).hexdigest()
    meta["n24_red_count"] = int(red24.sum())
    meta["n24_enc_sha256"] = hashlib.sha256(np.ascontiguousarray(enc24).tobytes()).hexdigest()
    store["enc24_rowsum"] = enc24.sum(axis=1)
    store["enc24_argmax"] = enc24.argmax(axis=1)
    store["enc24_max"] = enc24.max(axis=1)
    meta["red_counts"] = {str(n): int(store[f"red_{n}"].sum()) for n in (4, 8, 16)}
    np.savez_compressed(os.path.join(OUT, "ori_codec.npz"), **store)

    # ---- 4. encode_loc (URSO-style limits, urso.py:84-93 incl. its deg/rad quirk) -----
    fov_x, fov_y = 90.0 * np.pi / 180, 73.7 * np.pi / 180
    theta_x, theta_y = fov_x * np.pi / 360, fov_y * np.pi / 360
    x_max, y_max = np.tan(theta_x), np.tan(theta_y)
    z_min, z_max = 10.0, 40.0
    lmax, lmin = np.array([x_max, y_max, z_max]), np.array([-x_max, -y_max, z_min])
    locs = np.stack([rng.uniform(-0.01, 0.01, 6), rng.uniform(-0.01, 0.01, 6), rng.uniform(12, 38, 6)], axis=1)
    ls = {"locs": locs, "max_lim": lmax, "min_lim": lmin}
    for m in (4, 8):
        enc, H = utils.encode_loc(locs, m, beta, lmax, lmin)
        ls[f"enc_{m}"] = enc
        ls[f"map_{m}"] = H
    np.savez_compressed(os.path.join(OUT, "loc_codec.npz"), **ls)

    # ---- 5. stable_softmax + resize_image geometry ------------------------------------
    sm_in = np.stack([rng.normal(size=64) * s for s in (1.0, 30.0, 300.0)]).astype(np.float64)
    sm_in[2, 5] = 1000.0
    sm_out = np.stack([utils.stable_softmax(x) for x in sm_in])
    np.savez_compressed(os.path.join(OUT, "softmax.npz"), x=sm_in, y=sm_out)
    geo = []
    cases = [(960, 1280, "pad64", 960, 1280), (640, 1000, "pad64", 640, 1000), (512, 640, "pad64", 512, 640),
             (704, 1000, "pad64", 704, None),
             (480, 640, "square", None, 640), (960, 1280, "square", None, 1280), (600, 960, "none", None, None)]
    for (h, w, mode, min_dim, max_dim) in cases:
        img = np.zeros((h, w, 3), dtype=np.uint8)
        out, window, scale, padding, crop = utils.resize_image(img, min_dim=min_dim, max_dim=max_dim, min_scale=0, mode=mode)
        geo.append({"h": h, "w": w, "mode": mode, "min_dim": min_dim, "max_dim": max_dim,
                    "out_shape": list(out.shape), "window": [int(v) for v in window], "scale": float(scale),
                    "padding": [[int(a), int(b)] for a, b in padding]})

    # ---- 6. Config defaults + update() for the five BASELINE CLI combinations ---------
    def cli_config(backbone, width, height, scale, batch, f16=False, regress_ori=False, regress_loc=True,
                   ori_res=16, bottleneck=32, square=False):
        c = config.Config()
        c.ORI_BINS_PER_DIM = ori_res
        c.NR_DENSE_LAYERS = 1
        c.BOTTLENECK_WIDTH = bottleneck
        c.BACKBONE = backbone
        c.F16 = f16
        c.OPTIMIZER = "SGD"
        c.REGRESS_ORI = regress_ori
        c.REGRESS_LOC = regress_loc
        c.IMAGE_RESIZE_MODE = "square" if square else "pad64"
        c.IMAGE_MAX_DIM = round(width * scale)                      # pose_estimator.py:850
        hs = round(height * scale)                                  # pose_estimator.py:856-860
        c.IMAGE_MIN_DIM = hs - hs % 64 + 64 if hs % 64 > 0 else hs
        c.IMAGES_PER_GPU = batch
        c.BATCH_SIZE = c.IMAGES_PER_GPU * c.GPU_COUNT
        c.update()
        return c
    cfgs = {
        "cfg1": cli_config("resnet18", 128, 128, 1.0, 2, regress_ori=True),
        "cfg2": cli_config("resnet50", 1280, 960, 0.5, 32),
        "cfg4": cli_config("resnet101", 1280, 960, 0.5, 16, ori_res=24),
        "cfg5": cli_config("resnet50", 1920, 1200, 0.5, 32, f16=True, regress_loc=False),
    }
    cfg_dump = {}
    for k, c in cfgs.items():
        cfg_dump[k] = {"IMAGE_SHAPE": [int(v) for v in c.IMAGE_SHAPE], "IMAGE_META_SIZE": int(c.IMAGE_META_SIZE),
                       "BATCH_SIZE": int(c.BATCH_SIZE), "IMAGE_MIN_DIM": int(c.IMAGE_MIN_DIM),
                       "IMAGE_MAX_DIM": int(c.IMAGE_MAX_DIM)}
    d = config.Config()
    defaults = {}
    for a in dir(d):
        if a.startswith("__") or callable(getattr(d, a)):
            continue
        v = getattr(d, a)
        defaults[a] = v.tolist() if isinstance(v, np.ndarray) else v
    meta["resize_geometry"] = geo
    meta["cli_configs"] = cfg_dump
    meta["config_defaults"] = defaults
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
