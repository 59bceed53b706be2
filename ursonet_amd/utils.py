"""Host-side helpers with the reference's `utils` names (utils.py) used on either side of the hot
path: target codec re-exports, image resizing/padding, the triangular cyclic learning rate."""
import random

import numpy as np

from .pose import (encode_loc, encode_ori, encode_ori_fast, euler2quat, stable_softmax,  # noqa: F401
                   OrientationCodec, decode_orientations, pose_errors)


def encode_as_keypoints(oris, centroids, scale=1.0):
    """Drop-in for utils.encode_as_keypoints (utils.py:220-244)."""
    from .augment import encode_as_keypoints as f
    return f(oris, centroids, scale)


def clr_triangular(iteration, base_lr, max_lr, step_size):
    """CyclicLR(mode='triangular').clr() (clr_callback.py:104-111) at `iteration` batches since start;
    iteration 0 -> base_lr (on_train_begin, clr_callback.py:116-117)."""
    cycle = np.floor(1 + iteration / (2 * step_size))
    x = np.abs(iteration / step_size - 2 * cycle + 1)
    return float(base_lr + (max_lr - base_lr) * np.maximum(0, (1 - x)))


def _bilinear_resize(image, out_h, out_w):
    """Plain bilinear resampling (pixel centres aligned, zero outside).  The reference calls
    skimage.transform.resize(order=1, mode='constant', preserve_range=True) (utils.py:457-459); skimage
    is not available here, so rescaled images are NOT bit-identical to the reference's (scale == 1,
    the benchmarked path, never reaches this function)."""
    h, w = image.shape[:2]
    ys = (np.arange(out_h) + 0.5) * h / out_h - 0.5
    xs = (np.arange(out_w) + 0.5) * w / out_w - 0.5
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    img = image.astype(np.float64)
    if img.ndim == 2:
        img = img[:, :, None]

    def tap(yy, xx):
        valid = ((yy >= 0) & (yy < h))[:, None] & ((xx >= 0) & (xx < w))[None, :]
        return img[np.clip(yy, 0, h - 1)][:, np.clip(xx, 0, w - 1)] * valid[:, :, None]
    out = (tap(y0, x0) * ((1 - fy) * (1 - fx))[:, :, None] + tap(y0, x0 + 1) * ((1 - fy) * fx)[:, :, None] +
           tap(y0 + 1, x0) * (fy * (1 - fx))[:, :, None] + tap(y0 + 1, x0 + 1) * (fy * fx)[:, :, None])
    return out if image.ndim == 3 else out[:, :, 0]


def resize_image(image, min_dim=None, max_dim=None, min_scale=None, mode="square"):
    """utils.py:398-511: returns (image, window, scale, padding, crop) for modes none/square/pad64/crop."""
    image_dtype = image.dtype
    h, w = image.shape[:2]
    window = (0, 0, h, w)
    scale = 1
    padding = [(0, 0), (0, 0), (0, 0)]
    crop = None
    if mode == "none":
        return image, window, scale, padding, crop
    if min_dim:
        scale = min_dim / min(h, w)
    if min_scale and scale < min_scale:
        scale = min_scale
    if max_dim and mode != "crop":
        image_max = max(h, w)
        if round(image_max * scale) > max_dim:
            scale = max_dim / image_max
    if scale != 1:
        image = _bilinear_resize(image, round(h * scale), round(w * scale))
    if mode == "square":
        h, w = image.shape[:2]
        top_pad = (max_dim - h) // 2
        bottom_pad = max_dim - h - top_pad
        left_pad = (max_dim - w) // 2
        right_pad = max_dim - w - left_pad
        padding = [(top_pad, bottom_pad), (left_pad, right_pad), (0, 0)] if image.ndim > 2 else \
            [(top_pad, bottom_pad), (left_pad, right_pad)]
        image = np.pad(image, padding, mode='constant', constant_values=0)
        window = (top_pad, left_pad, h + top_pad, w + left_pad)
    elif mode == "pad64":
        h, w = image.shape[:2]
        assert min_dim % 64 == 0, "Minimum dimension must be a multiple of 64"
        if h % 64 > 0:
            max_h = h - (h % 64) + 64
            top_pad = (max_h - h) // 2
            bottom_pad = max_h - h - top_pad
        else:
            top_pad = bottom_pad = 0
        if w % 64 > 0:
            max_w = w - (w % 64) + 64
            left_pad = (max_w - w) // 2
            right_pad = max_w - w - left_pad
        else:
            left_pad = right_pad = 0
        padding = [(top_pad, bottom_pad), (left_pad, right_pad), (0, 0)] if image.ndim > 2 else \
            [(top_pad, bottom_pad), (left_pad, right_pad)]
        image = np.pad(image, padding, mode='constant', constant_values=0)
        window = (top_pad, left_pad, h + top_pad, w + left_pad)
    elif mode == "crop":
        h, w = image.shape[:2]
        y = random.randint(0, (h - min_dim))
        x = random.randint(0, (w - min_dim))
        crop = (y, x, min_dim, min_dim)
        image = image[y:y + min_dim, x:x + min_dim]
        window = (0, 0, min_dim, min_dim)
    else:
        raise Exception("Mode {} not supported".format(mode))
    return image.astype(image_dtype), window, scale, padding, crop
