"""Pose target codec (host, vectorised NumPy) and batched soft-argmax decode / metrics (GPU).

Host side mirrors the reference's L1 helpers that datasets call when they are loaded
(utils.encode_ori utils.py:246-317, encode_ori_fast :319-346, encode_loc :349-396,
stable_softmax :26-28, se3lib.euler2quat se3lib.py:53-67); the decode replaces the per-image
Python loop of se3lib.quat_weighted_avg (se3lib.py:217-260, 40-318 ms per image) with one
kernel launch for the whole batch (urso_quat_wavg_decode).  No CPU decode path is provided.
"""
import itertools

import numpy as np


def stable_softmax(X):
    """utils.py:26-28 (1-D)."""
    e = np.exp(X - np.max(X))
    return e / np.sum(e)


def euler2quat(pitch, yaw, roll):
    """se3lib.py:53-67, vectorised over arrays of angles in degrees -> [..., 4] (x, y, z, w)."""
    p, y, r = (np.asarray(a, dtype=np.float64) * (np.pi / 360) for a in (pitch, yaw, roll))
    cp, sp, cy, sy, cr, sr = np.cos(p), np.sin(p), np.cos(y), np.sin(y), np.cos(r), np.sin(r)
    return np.stack([sy * sr * cp - cy * cr * sp, -sy * cr * cp - cy * sr * sp,
                     -cy * sr * cp + sy * cr * sp, cy * cr * cp + sy * sr * sp], axis=-1)


class OrientationCodec(object):
    """Bin -> quaternion map, redundant-bin mask and Gaussian soft assignment over the
    nr_bins^3 Euler grid (bin index = i*n^2 + j*n + k, pitch slowest)."""

    def __init__(self, nr_bins_per_dim, beta, min_lim=(-180, -90, -180), max_lim=(180, 90, 180)):
        n = int(nr_bins_per_dim)
        self.n, self.beta = n, float(beta)
        min_lim, max_lim = np.asarray(min_lim, dtype=np.float64), np.asarray(max_lim, dtype=np.float64)
        bins = np.linspace(0.0, 1.0, n)
        grid = np.asarray(list(itertools.product(bins, repeat=3)))          # same enumeration as the reference
        H_ori = grid * (max_lim - min_lim) + min_lim
        self.H_ori = H_ori
        # cos/sin through NumPy's SCALAR path per distinct angle (as the reference's per-bin euler2quat
        # calls do): the vectorised SIMD path can differ by 1 ulp, and the map must be bit-identical.
        def trig(col):
            vals = {float(v): (np.cos(float(v) * np.pi / 360), np.sin(float(v) * np.pi / 360)) for v in np.unique(col)}
            c = np.array([vals[float(v)][0] for v in col]); s_ = np.array([vals[float(v)][1] for v in col])
            return c, s_
        (cp, sp), (cy, sy), (cr, sr) = trig(H_ori[:, 0]), trig(H_ori[:, 1]), trig(H_ori[:, 2])
        self.H_quat = np.stack([sy * sr * cp - cy * cr * sp, -sy * cr * cp - cy * sr * sp,
                                -cy * sr * cp + sy * cr * sp, cy * cr * cp + sy * sr * sp], axis=-1).astype(np.float32)
        boundary = np.logical_or(H_ori[:, 0] == max_lim[0], H_ori[:, 2] == max_lim[2])
        gymbal = np.logical_and(np.abs(H_ori[:, 1]) == max_lim[1], H_ori[:, 0] != min_lim[0])
        self.redundant = np.logical_or(boundary, gymbal)
        self.var = (self.beta / n) ** 2 / 12                                  # utils.py:267-268

    def encode(self, oris, dtype=np.float32):
        """[N,4] quaternions -> [N, n^3] PMFs (utils.encode_ori rows / encode_ori_fast)."""
        oris = np.atleast_2d(np.asarray(oris, dtype=np.float64))
        d = np.abs(oris @ self.H_quat.astype(np.float64).T)
        pr = np.exp(-2 * (np.arccos(np.minimum(1.0, d)) / np.pi) ** 2 / self.var)
        pr[:, self.redundant] = 0
        return (pr / pr.sum(axis=1, keepdims=True)).astype(dtype)


def encode_ori(oris, nr_bins_per_dim, beta, min_lim, max_lim):
    """Drop-in for utils.encode_ori: (ori_encoded float32, H_quat float32, Redundant_flags)."""
    c = OrientationCodec(nr_bins_per_dim, beta, min_lim, max_lim)
    return c.encode(oris), c.H_quat, c.redundant


def encode_ori_fast(ori, beta, H_quat, Redundant_flags):
    """Drop-in for utils.encode_ori_fast (one sample, prebuilt map) -> float64 PMF."""
    n = round(len(H_quat) ** (1. / 3))
    var = (beta / n) ** 2 / 12
    d = np.abs(np.sum(np.asarray(ori) * H_quat, axis=-1))
    pr = np.exp(-2 * (np.arccos(np.minimum(1.0, d)) / np.pi) ** 2 / var)
    pr[np.asarray(Redundant_flags, dtype=bool)] = 0
    return pr / np.sum(pr)


def encode_loc(locs, nr_bins_per_dim, beta, max_lim, min_lim):
    """Drop-in for utils.encode_loc (note the reference's argument order: max_lim, then min_lim)."""
    locs = np.atleast_2d(np.asarray(locs, dtype=np.float64))
    max_lim, min_lim = np.asarray(max_lim, dtype=np.float64), np.asarray(min_lim, dtype=np.float64)
    sig2 = (beta / nr_bins_per_dim) ** 2 / 12
    bins = np.linspace(0.0, 1.0, nr_bins_per_dim)
    H = np.asarray(list(itertools.product(bins, repeat=3))) * (max_lim - min_lim) + min_lim
    H[:, 0] *= H[:, 2]
    H[:, 1] *= H[:, 2]
    Z = locs[:, 2:3]
    mean = np.concatenate([locs[:, 0:1] * Z, locs[:, 1:2] * Z, Z], axis=1)
    d2 = ((H[None, :, :] - mean[:, None, :]) ** 2).sum(-1)
    pr = np.exp(-0.5 * d2 / sig2) / np.sqrt((2 * np.pi * sig2) ** 3)
    return (pr / pr.sum(axis=1, keepdims=True)).astype(np.float32), H


def location_map(nr_bins_per_dim, max_lim, min_lim):
    """The metric bin-centre grid utils.encode_loc returns as its second value (utils.py:363-378): [m^3, 3] float64."""
    max_lim, min_lim = np.asarray(max_lim, dtype=np.float64), np.asarray(min_lim, dtype=np.float64)
    bins = np.linspace(0.0, 1.0, nr_bins_per_dim)
    H = np.asarray(list(itertools.product(bins, repeat=3))) * (max_lim - min_lim) + min_lim
    H[:, 0] *= H[:, 2]
    H[:, 1] *= H[:, 2]
    return H


def encode_locations(locs, nr_bins_per_dim, beta, max_lim, min_lim):
    """Batched utils.encode_loc on the GPU (urso_encode_loc): locs [B,3] -> (float32 CUDA tensor [B, m^3] of PMFs, map [m^3,3])."""
    import torch
    from . import hip
    H = location_map(nr_bins_per_dim, max_lim, min_lim)
    ld = torch.as_tensor(np.atleast_2d(np.asarray(locs, dtype=np.float64))).cuda().contiguous()
    hd = torch.as_tensor(H).cuda().contiguous()
    out = torch.empty(ld.shape[0], H.shape[0], dtype=torch.float32, device=ld.device)
    hip.encode_loc(ld.shape[0], H.shape[0], ld, hd, (beta / nr_bins_per_dim) ** 2 / 12, out)
    return out, H


# --------------------------------------------------------------------------- GPU decode + metrics
def decode_orientations(ori_logits, H_quat):
    """Batched probabilistic soft-argmax (pose_estimator.py:406-409) on the GPU.
    ori_logits: [B, n^3] device tensor / array; returns [B,4] float32 unit quaternions (numpy)."""
    import torch
    from . import hip
    z = torch.as_tensor(ori_logits, dtype=torch.float32).cuda().contiguous()
    hq = torch.as_tensor(np.ascontiguousarray(H_quat), dtype=torch.float32).cuda()
    q = torch.empty(z.shape[0], 4, dtype=torch.float32, device=z.device)
    hip.quat_wavg_decode(z.shape[0], z.shape[1], z, hq, q)
    return q.cpu().numpy()


def pose_errors(loc_est, q_est, loc_gt, q_gt):
    """(angular error [deg], location error, ESA score), pose_estimator.py:434-445, vectorised."""
    q_est, q_gt = np.atleast_2d(q_est).astype(np.float64), np.atleast_2d(q_gt).astype(np.float64)
    loc_est, loc_gt = np.atleast_2d(loc_est).astype(np.float64), np.atleast_2d(loc_gt).astype(np.float64)
    d = np.minimum(1.0, np.abs((q_est * q_gt).sum(-1)))
    ang = 2 * np.arccos(d)
    le = np.linalg.norm(loc_est - loc_gt, axis=-1)
    return ang * 180 / np.pi, le, le / np.linalg.norm(loc_gt, axis=-1) + ang
