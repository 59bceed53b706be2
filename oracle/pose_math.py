"""Oracle (test infrastructure): NumPy restatement of the reference's pose codec,
soft-argmax decode and evaluation metrics.

Every function cites the reference lines it follows.  Pinned against golden
vectors produced by the reference's own modules (tests/golden/make_golden.py).
Deliberately literal (Python loops where the reference loops) -- this is the
checker, not the product.
"""
import itertools
import math

import numpy as np


# --------------------------------------------------------------------------
# se3lib.py
# --------------------------------------------------------------------------
def euler2quat(pitch, yaw, roll):
    """se3lib.py:53-67 -- Euler angles (deg) -> quaternion [x,y,z,w] as a (4,) array."""
    cp, sp = np.cos(pitch * np.pi / 360), np.sin(pitch * np.pi / 360)
    cy, sy = np.cos(yaw * np.pi / 360), np.sin(yaw * np.pi / 360)
    cr, sr = np.cos(roll * np.pi / 360), np.sin(roll * np.pi / 360)
    return np.array([sy * sr * cp - cy * cr * sp,
                     -sy * cr * cp - cy * sr * sp,
                     -cy * sr * cp + sy * cr * sp,
                     cy * cr * cp + sy * sr * sp])


def euler2SO3_left(pitch, yaw, roll):
    """se3lib.py:38-51."""
    cp, sp = np.cos(pitch * np.pi / 180), np.sin(pitch * np.pi / 180)
    cy, sy = np.cos(yaw * np.pi / 180), np.sin(yaw * np.pi / 180)
    cr, sr = np.cos(roll * np.pi / 180), np.sin(roll * np.pi / 180)
    return np.array([[cy * cr, sp * sy * cr - cp * sr, cp * sy * cr + sp * sr],
                     [cy * sr, sp * sy * sr + cp * cr, cp * sy * sr - sp * cr],
                     [-sy, sp * cy, cp * cy]])


def SO32quat(R):
    """se3lib.py:77-113 (JPL convention, four-branch)."""
    q = [0, 0, 0, 0]
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        Z = math.sqrt(tr + 1) * 2
        q[3] = 0.25 * Z
        q[0] = (R[1, 2] - R[2, 1]) / Z
        q[1] = (R[2, 0] - R[0, 2]) / Z
        q[2] = (R[0, 1] - R[1, 0]) / Z
    elif (R[0, 0] > R[1, 1]) & (R[0, 0] > R[2, 2]):
        Z = math.sqrt(1.0 + 2 * R[0, 0] - tr) * 2
        q[3] = (R[1, 2] - R[2, 1]) / Z
        q[0] = 0.25 * Z
        q[1] = (R[0, 1] + R[1, 0]) / Z
        q[2] = (R[0, 2] + R[2, 0]) / Z
    elif R[1, 1] > R[2, 2]:
        Z = math.sqrt(1.0 + 2 * R[1, 1] - tr) * 2
        q[3] = (R[2, 0] - R[0, 2]) / Z
        q[0] = (R[0, 1] + R[1, 0]) / Z
        q[1] = 0.25 * Z
        q[2] = (R[1, 2] + R[2, 1]) / Z
    else:
        Z = math.sqrt(1.0 + 2 * R[2, 2] - tr) * 2
        q[3] = (R[0, 1] - R[1, 0]) / Z
        q[0] = (R[0, 2] + R[2, 0]) / Z
        q[1] = (R[1, 2] + R[2, 1]) / Z
        q[2] = 0.25 * Z
    return q


def quat2SO3(q):
    """se3lib.py:134-144."""
    return np.array([
        [1 - 2 * q[1] ** 2 - 2 * q[2] ** 2, 2 * (q[0] * q[1] + q[2] * q[3]), 2 * (q[0] * q[2] - q[1] * q[3])],
        [2 * (q[0] * q[1] - q[2] * q[3]), 1 - 2 * q[0] ** 2 - 2 * q[2] ** 2, 2 * (q[1] * q[2] + q[0] * q[3])],
        [2 * (q[0] * q[2] + q[1] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 1 - 2 * q[0] ** 2 - 2 * q[1] ** 2]])


def quat_mult(a, b):
    """se3lib.py:164-179, for 1-D length-4 ``b`` (the ``b*c.T`` branch), unit-normalised."""
    c = np.array([[a[3], a[2], -a[1], a[0]],
                  [-a[2], a[3], a[0], a[1]],
                  [a[1], -a[0], a[3], a[2]],
                  [-a[0], -a[1], -a[2], a[3]]], dtype=np.float64)
    res = np.asarray(b, dtype=np.float64) @ c.T
    return res / np.linalg.norm(res)


def angle_between_quats(q1, q2):
    """se3lib.py:213-215 -- degrees."""
    d = np.abs(np.dot(np.asarray(q1, dtype=np.float64).ravel(), np.asarray(q2, dtype=np.float64).ravel()))
    return 2 * np.arccos(np.clip(d, 0.0, 1.0)) * 180 / np.pi


def quat_weighted_avg(Q, W):
    """se3lib.py:217-260 -- A = sum_i w_i q_i q_i^T accumulated in float32 in bin
    order (se3lib.py:245-248), eigenvector of the largest eigenvalue, normalised.
    Returns (q_avg (4,), A (4,4) float32).  The reference also returns inv(A); the
    sign of q_avg is arbitrary (eigen-solver dependent) -- compare with |dot|."""
    N = np.size(Q, 0)
    A = np.zeros((4, 4), dtype=np.float32)
    for i in range(N):
        a = np.array([[Q[i, 0], Q[i, 1], Q[i, 2], Q[i, 3]]])
        A += a.T * a * W[i]
    s, v = np.linalg.eig(A)
    idx = np.argsort(s)
    q_avg = v[:, idx[-1]]
    q_avg = q_avg / np.linalg.norm(q_avg)
    return np.real(q_avg), A


# --------------------------------------------------------------------------
# utils.py
# --------------------------------------------------------------------------
def stable_softmax(X):
    """utils.py:26-28."""
    exps = np.exp(X - np.max(X))
    return exps / np.sum(exps)


def ori_histogram(nr_bins_per_dim, min_lim, max_lim):
    """utils.py:270-300 -- (H_ori [n^3,3] float64, H_quat [n^3,4] float32, Redundant_flags [n^3] bool).
    Bin index = i*n^2 + j*n + k over (pitch, yaw, roll), pitch slowest (itertools.product)."""
    d = 3
    min_lim = np.asarray(min_lim)
    max_lim = np.asarray(max_lim)
    bins = np.linspace(0.0, 1.0, nr_bins_per_dim)
    H_loc_list = list(itertools.product(bins, repeat=d))
    H_ori = np.asarray(H_loc_list * (max_lim - min_lim) + min_lim)
    nr_total = nr_bins_per_dim ** d
    H_quat = np.zeros((nr_total, 4), dtype=np.float32)
    for i in range(nr_total):
        H_quat[i, :] = euler2quat(H_ori[i, 0], H_ori[i, 1], H_ori[i, 2])
    boundary = np.logical_or(H_ori[:, 0] == max_lim[0], H_ori[:, 2] == max_lim[2])
    gymbal = np.logical_and(np.abs(H_ori[:, 1]) == max_lim[1], H_ori[:, 0] != min_lim[0])
    return H_ori, H_quat, np.logical_or(boundary, gymbal)


def ori_kernel_var(nr_bins_per_dim, beta):
    """utils.py:267-268 / :334-335 / pose_estimator.py:332-333."""
    delta = beta / nr_bins_per_dim
    return delta ** 2 / 12


def encode_ori(oris, nr_bins_per_dim, beta, min_lim, max_lim):
    """utils.py:246-317 -- returns (ori_encoded [N,n^3] float32, H_quat, Redundant_flags)."""
    oris = np.asarray(oris)
    var = ori_kernel_var(nr_bins_per_dim, beta)
    _, H_quat, red = ori_histogram(nr_bins_per_dim, min_lim, max_lim)
    out = np.zeros((np.size(oris, 0), nr_bins_per_dim ** 3), dtype=np.float32)
    for i in range(np.size(oris, 0)):
        pr = np.exp(-2 * (np.arccos(np.minimum(1.0, np.abs(np.sum(oris[i, :] * H_quat, axis=-1)))) / np.pi) ** 2 / var)
        pr[red] = 0
        out[i, :] = pr / np.sum(pr)
    return out, H_quat, red


def encode_ori_fast(ori, beta, H_quat, Redundant_flags):
    """utils.py:319-346 -- one sample, prebuilt map; returns float64 PMF (as the reference does)."""
    n = round(len(H_quat) ** (1. / 3))
    var = ori_kernel_var(n, beta)
    pr = np.exp(-2 * (np.arccos(np.minimum(1.0, np.abs(np.sum(ori * H_quat, axis=-1)))) / np.pi) ** 2 / var)
    pr[np.asarray(Redundant_flags, dtype=bool)] = 0
    return pr / np.sum(pr)


def encode_loc(locs, nr_bins_per_dim, beta, max_lim, min_lim):
    """utils.py:349-396 -- NOTE the reference's argument order (max_lim before min_lim).
    3-D isotropic Gaussian PMF over the (x/z, y/z, z) grid scaled back to metric.
    The normalising constant of multivariate_normal.pdf cancels in the division."""
    locs = np.asarray(locs, dtype=np.float64)
    max_lim = np.asarray(max_lim, dtype=np.float64)
    min_lim = np.asarray(min_lim, dtype=np.float64)
    n_ex = np.size(locs, 0)
    d = np.size(locs, 1) if locs.ndim > 1 else 1
    delta = beta / nr_bins_per_dim                      # utils.py:364 ("tmp" override)
    sig2 = delta ** 2 / 12
    bins = np.linspace(0.0, 1.0, nr_bins_per_dim)
    H = np.asarray(list(itertools.product(bins, repeat=d)) * (max_lim - min_lim) + min_lim)
    H[:, 0] = H[:, 0] * H[:, 2]
    H[:, 1] = H[:, 1] * H[:, 2]
    out = np.zeros((n_ex, nr_bins_per_dim ** d), dtype=np.float32)
    norm = 1.0 / np.sqrt((2 * np.pi * sig2) ** 3)
    for i in range(n_ex):
        Z = locs[i, 2]
        mean = np.array([locs[i, 0] * Z, locs[i, 1] * Z, Z])
        diff = H - mean
        pr = norm * np.exp(-0.5 * np.sum(diff * diff, axis=1) / sig2)
        out[i, :] = pr / np.sum(pr)
    return out, H


def resize_image_geometry(h, w, min_dim, max_dim, min_scale, mode):
    """utils.py:398-511 restricted to scale==1 (any rescale needs skimage, absent):
    returns (out_h, out_w, window, padding) for modes none/square/pad64."""
    window = (0, 0, h, w)
    padding = [(0, 0), (0, 0), (0, 0)]
    if mode == "none":
        return h, w, window, padding
    scale = 1
    if min_dim:
        scale = min_dim / min(h, w)
    if min_scale and scale < min_scale:
        scale = min_scale
    if max_dim and mode != "crop":
        if round(max(h, w) * scale) > max_dim:
            scale = max_dim / max(h, w)
    if scale != 1:
        raise NotImplementedError("rescale needs skimage (absent); oracle covers scale==1 only")
    if mode == "square":
        tp = (max_dim - h) // 2
        bp = max_dim - h - tp
        lp = (max_dim - w) // 2
        rp = max_dim - w - lp
    elif mode == "pad64":
        assert min_dim % 64 == 0, "Minimum dimension must be a multiple of 64"
        if h % 64 > 0:
            mh = h - (h % 64) + 64
            tp = (mh - h) // 2
            bp = mh - h - tp
        else:
            tp = bp = 0
        if w % 64 > 0:
            mw = w - (w % 64) + 64
            lp = (mw - w) // 2
            rp = mw - w - lp
        else:
            lp = rp = 0
    else:
        raise Exception("Mode {} not supported".format(mode))
    padding = [(tp, bp), (lp, rp), (0, 0)]
    return h + tp + bp, w + lp + rp, (tp, lp, h + tp, w + lp), padding


# --------------------------------------------------------------------------
# pose_estimator.py -- decode + metrics (evaluate)
# --------------------------------------------------------------------------
def decode_orientation(ori_logits, ori_histogram_map):
    """pose_estimator.py:406-409 -- softmax over the bins, weighted quaternion average."""
    pmf = stable_softmax(ori_logits)
    q, _ = quat_weighted_avg(ori_histogram_map, pmf)
    return q


def decode_location_classified(loc_logits, histogram_3D_map):
    """pose_estimator.py:380-383."""
    pmf = stable_softmax(loc_logits)
    return np.asarray(pmf) @ np.asarray(histogram_3D_map)


def pose_errors(loc_est, q_est, loc_gt, q_gt):
    """pose_estimator.py:434-445 -- (angular_err_deg, loc_err, esa_score)."""
    d = np.abs(np.dot(np.asarray(q_est, dtype=np.float64).ravel(), np.asarray(q_gt, dtype=np.float64).ravel()))
    ang = 2 * np.arccos(d) * 180 / np.pi
    loc_err = np.linalg.norm(np.asarray(loc_est, dtype=np.float64).ravel() - np.asarray(loc_gt, dtype=np.float64).ravel())
    esa = loc_err / np.linalg.norm(loc_gt) + 2 * np.arccos(d)
    return ang, loc_err, esa


# --------------------------------------------------------------------------
# rotation augmentation (utils.py:30-86).  The image warp is OpenCV's (cv2 is absent here and on the GPU box), restated
# from OpenCV's published warpPerspective / remap algorithm for 8-bit images.  What the reference's call runs:
# `cv2.warpPerspective(image, M, (width, height), cv2.WARP_INVERSE_MAP)` -- in the Python binding the 4th positional
# parameter is `dst`, not `flags` (signature: src, M, dsize[, dst[, flags[, borderMode[, borderValue]]]]), so the
# constant lands in the output-array slot and the call runs with the DEFAULT flags: M is the forward map
# (dst(p) = src(M^-1 p)), INTER_LINEAR, BORDER_CONSTANT 0.  This is also the only reading under which the warped image
# agrees with the updated pose (K t_new = M K t; tests/test_augment_gpu.py checks it).
# PARITY UNPINNED for the image part (no cv2 output to compare with); pinned for the pose update (se3lib goldens).
# --------------------------------------------------------------------------
def invert3x3(M):
    """cv::invert of a 3x3 (closed form: adjugate / determinant), float64."""
    M = np.asarray(M, dtype=np.float64).reshape(3, 3)
    a, b, c, d, e, f, g, h, i = M.ravel()
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    adj = np.array([[e * i - f * h, c * h - b * i, b * f - c * e],
                    [f * g - d * i, a * i - c * g, c * d - a * f],
                    [d * h - e * g, b * g - a * h, a * e - b * d]])
    return adj * (1.0 / det)


def warp_perspective(image, M, inverse_map=False, interp="linear"):
    """cv2.warpPerspective(image, M, (W, H), flags=interp | (WARP_INVERSE_MAP if inverse_map)) for uint8 images,
    constant-0 border.  Per destination pixel (Python loop over pixels: the checker, small images only).
      nearest: src(cvRound(X/W), cvRound(Y/W));
      linear : coordinates quantised to 1/32 pixel (cvRound(32 X/W)), taps weighted by the 15-bit table
               w = round((1-ax/32)(1-ay/32) 2^15) ..., result (sum + 2^14) >> 15; taps outside the image read 0."""
    image = np.asarray(image)
    assert image.dtype == np.uint8
    H, W = image.shape[:2]
    Mi = np.asarray(M, dtype=np.float64).reshape(3, 3) if inverse_map else invert3x3(M)
    out = np.zeros_like(image)
    scale = 32.0 if interp == "linear" else 1.0
    lo, hi = -2147483648.0, 2147483647.0

    def tap(yy, xx):
        if 0 <= yy < H and 0 <= xx < W:
            return image[yy, xx].astype(np.int64)
        return np.zeros(image.shape[2:], dtype=np.int64)

    for y in range(H):
        for x in range(W):
            X = Mi[0, 0] * x + Mi[0, 1] * y + Mi[0, 2]
            Y = Mi[1, 0] * x + Mi[1, 1] * y + Mi[1, 2]
            Wd = Mi[2, 0] * x + Mi[2, 1] * y + Mi[2, 2]
            w = scale / Wd if Wd != 0 else 0.0
            qx = int(np.rint(min(max(X * w, lo), hi)))          # cvRound: half to even
            qy = int(np.rint(min(max(Y * w, lo), hi)))
            if interp != "linear":
                out[y, x] = tap(qy, qx)
                continue
            sx, sy, ax, ay = qx >> 5, qy >> 5, qx & 31, qy & 31
            tab = np.array([(1 - ax / 32.0) * (1 - ay / 32.0), (ax / 32.0) * (1 - ay / 32.0),
                            (1 - ax / 32.0) * (ay / 32.0), (ax / 32.0) * (ay / 32.0)], dtype=np.float32)
            wt = np.rint(tab * np.float32(32768)).astype(np.int64)          # exact: multiples of 32 summing to 2^15
            acc = wt[0] * tap(sy, sx) + wt[1] * tap(sy, sx + 1) + wt[2] * tap(sy + 1, sx) + wt[3] * tap(sy + 1, sx + 1)
            out[y, x] = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out


def rotate_cam_given(image, t, q, K, pyr_change):
    """utils.rotate_cam (utils.py:30-58) / rotate_image (:60-86) for a GIVEN Euler perturbation (deg)."""
    R = euler2SO3_left(pyr_change[0], pyr_change[1], pyr_change[2])
    K = np.asarray(K, dtype=np.float64)
    M = K @ R @ np.linalg.inv(K)
    warped = warp_perspective(image, M)
    t_new = np.asarray(t, dtype=np.float64) @ R.T
    q_new = quat_mult(SO32quat(R), q)
    return warped, t_new, q_new


def encode_as_keypoints(ori, centroid, scale=1.0):
    """utils.py:220-227 (one pose): K = R [0,0,1]^T scale + c and R [0,1,0]^T scale + c, as 3x1 columns."""
    R = quat2SO3(ori)
    c = np.asarray(centroid, dtype=np.float64).reshape(3, 1)
    return R @ (scale * np.array([[0.0], [0.0], [1.0]])) + c, R @ (scale * np.array([[0.0], [1.0], [0.0]])) + c
