#!/usr/bin/env python3
"""tests/golden/resize_skimage.npz: outputs of THE REFERENCE'S utils.resize_image (utils.py:398-511) with the real scikit-image for the cases that
rescale (scale != 1 goes through skimage.transform.resize; the default interpreter of the build image has no skimage, /opt/conda/bin/python3.9
has 0.18.3).  Run:  /opt/conda/bin/python3.9 tests/golden/make_resize_golden.py   (reads /root/reference, writes next to itself).
tensorflow and cv2 -- imported by utils.py, not touched by resize_image -- are empty stand-ins."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for name in ("tensorflow", "cv2"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, os.environ.get("URSO_REFERENCE", "/root/reference"))
import skimage                                                           # noqa: E402 (the real one)
import utils                                                             # noqa: E402 (the reference's)

rng = np.random.default_rng(42)


def frame(h, w, c=3):
    """A SPEED-like frame: dark background, a bright textured blob, sensor noise."""
    yy, xx = np.mgrid[0:h, 0:w]
    img = rng.normal(8, 3, size=(h, w))
    img += (((yy - 0.45 * h) ** 2 + (xx - 0.55 * w) ** 2) < (0.22 * min(h, w)) ** 2) * (150 + 60 * np.sin(xx / 3.0) * np.cos(yy / 2.0))
    img = np.clip(img, 0, 255).round().astype(np.uint8)
    return np.repeat(img[..., None], c, axis=-1) if c else img


CASES = [  # name, (h, w), min_dim, max_dim, min_scale, mode
    ("speed_half_pad64", (150, 240), 0, 0, 0.5, "pad64"),          # SPEED 1200 x 1920 at image_scale 0.5, in miniature
    ("urso_half_pad64", (120, 160), 0, 0, 0.5, "pad64"),
    ("odd_0p4_pad64", (97, 131), 0, 0, 0.4, "pad64"),
    ("upscale_square", (40, 56), 64, 96, 0, "square"),
    ("downscale_square", (200, 120), 0, 128, 0, "square"),
]
out, meta = {}, []
for name, (h, w), min_dim, max_dim, min_scale, mode in CASES:
    img = frame(h, w)
    res, window, scale, padding, crop = utils.resize_image(img, min_dim=min_dim, max_dim=max_dim or None, min_scale=min_scale or None, mode=mode)       # (pad64 asserts min_dim % 64 == 0: 0, not None)
    out[name + "/in"] = img
    out[name + "/out"] = np.asarray(res)
    out[name + "/window"] = np.asarray(window, dtype=np.int64)
    out[name + "/scale"] = np.asarray(scale, dtype=np.float64)
    out[name + "/padding"] = np.asarray(padding, dtype=np.int64)
    out[name + "/args"] = np.asarray([min_dim, max_dim, min_scale], dtype=np.float64)
    meta.append("%s %s" % (name, mode))
    print(name, img.shape, "->", res.shape, res.dtype, "window", window, "scale", scale)
out["cases"] = np.asarray(meta)
out["skimage_version"] = np.asarray(skimage.__version__)
np.savez_compressed(os.path.join(HERE, "resize_skimage.npz"), **out)
