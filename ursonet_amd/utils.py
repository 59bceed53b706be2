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


def _gaussian_kernel1d(sigma):
    r = max(1, int(4.0 * sigma + 0.5))                      # scipy.ndimage truncate = 4
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    return k / k.sum()


def _smooth_axis(img, sigma, axis):
    """1-D Gaussian along `axis` with zero ('constant') boundary, as skimage's anti-aliasing pre-filter applies it."""
    k = _gaussian_kernel1d(sigma)
    r = len(k) // 2
    pad = [(0, 0)] * img.ndim
    pad[axis] = (r, r)
    padded = np.pad(img, pad, mode="constant")
    out = np.zeros_like(img)
    n = img.shape[axis]
    for i, kv in enumerate(k):
        sl = [slice(None)] * img.ndim
        sl[axis] = slice(i, i + n)
        out += kv * padded[tuple(sl)]
    return out


def _bilinear_resize(image, out_h, out_w, anti_aliasing=True):
    """Bilinear resampling with pixel centres aligned and zeros outside -- the arithmetic of
    skimage.transform.resize(order=1, mode='constant', preserve_range=True) (utils.py:457-459).  When an axis shrinks, newer
    skimage versions first smooth it with a Gaussian of sigma = (in/out - 1)/2 (anti_aliasing=True is their default):
    reproduced here because real URSO / SPEED frames (1280x960, 1920x1200) are reduced by 2-3x on this path.  skimage is not
    installed in this image; its resize is scipy.ndimage.gaussian_filter + zoom(order=1, grid_mode=True, mode='grid-constant'), and
    this function is pinned to exactly those two calls (tests/test_host_cpu.py::test_resize_equals_scipy_ndimage_...)."""
    h, w = image.shape[:2]
    img = image.astype(np.float64)
    if img.ndim == 2:
        img = img[:, :, None]
    if anti_aliasing:
        for axis, (n_in, n_out) in enumerate(((h, out_h), (w, out_w))):
            if n_out < n_in:
                img = _smooth_axis(img, (n_in / n_out - 1) / 2.0, axis)
    ys = (np.arange(out_h) + 0.5) * h / out_h - 0.5
    xs = (np.arange(out_w) + 0.5) * w / out_w - 0.5
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]

    def rows(yy):                                            # gather rows, zero where outside
        return img[np.clip(yy, 0, h - 1)] * ((yy >= 0) & (yy < h))[:, None, None]

    def cols(a, xx):
        return a[:, np.clip(xx, 0, w - 1)] * ((xx >= 0) & (xx < w))[None, :, None]
    top, bot = rows(y0), rows(y0 + 1)
    out = ((cols(top, x0) * (1 - fx) + cols(top, x0 + 1) * fx) * (1 - fy) + (cols(bot, x0) * (1 - fx) + cols(bot, x0 + 1) * fx) * fy)
    return out if image.ndim == 3 else out[:, :, 0]


def _centered_pad(size, target):
    """(before, after) zero padding that centres `size` pixels inside `target`; the odd pixel goes after."""
    before = (target - size) // 2
    return before, target - size - before


def resize_geometry(h, w, min_dim=None, max_dim=None, min_scale=None, mode="square"):
    """The geometry half of utils.resize_image (utils.py:398-511) as a pure function of the image size:
    -> (scale, (new_h, new_w), ((top, bottom), (left, right)), window).  'crop' is resolved by the caller (it draws the
    crop origin from `random`); 'none' leaves the image alone."""
    if mode == "none":
        return 1, (h, w), ((0, 0), (0, 0)), (0, 0, h, w)
    scale = 1
    if min_dim:
        scale = min_dim / min(h, w)                          # the reference also scales DOWN to min_dim (no max(1, .))
    if min_scale and scale < min_scale:
        scale = min_scale
    if max_dim and mode != "crop" and round(max(h, w) * scale) > max_dim:
        scale = max_dim / max(h, w)
    nh, nw = (round(h * scale), round(w * scale)) if scale != 1 else (h, w)
    if mode == "square":
        pads = (_centered_pad(nh, max_dim), _centered_pad(nw, max_dim))
    elif mode == "pad64":
        assert min_dim % 64 == 0, "Minimum dimension must be a multiple of 64"
        pads = tuple(_centered_pad(n, -(-n // 64) * 64) for n in (nh, nw))
    elif mode == "crop":
        pads = ((0, 0), (0, 0))
    else:
        raise Exception("Mode {} not supported".format(mode))
    (top, _), (left, _) = pads
    return scale, (nh, nw), pads, (top, left, nh + top, nw + left)


def resize_image(image, min_dim=None, max_dim=None, min_scale=None, mode="square"):
    """Drop-in for utils.resize_image (utils.py:398-511): -> (image, window, scale, padding, crop); modes none / square /
    pad64 / crop.  Geometry from resize_geometry (pinned to the reference's outputs in tests/golden/meta.json)."""
    dtype = image.dtype
    h, w = image.shape[:2]
    scale, (nh, nw), pads, window = resize_geometry(h, w, min_dim, max_dim, min_scale, mode)
    padding = [pads[0], pads[1]] + ([(0, 0)] if image.ndim > 2 else [])
    if mode == "none":
        return image, window, scale, [(0, 0), (0, 0), (0, 0)], None
    if scale != 1:
        image = _bilinear_resize(image, nh, nw)
    if mode == "crop":
        y, x = random.randint(0, nh - min_dim), random.randint(0, nw - min_dim)
        return (image[y:y + min_dim, x:x + min_dim].astype(dtype), (0, 0, min_dim, min_dim), scale,
                [(0, 0), (0, 0), (0, 0)], (y, x, min_dim, min_dim))
    out = np.zeros((nh + sum(pads[0]), nw + sum(pads[1])) + image.shape[2:], dtype=dtype)
    out[window[0]:window[2], window[1]:window[3]] = image
    return out, window, scale, padding, None
