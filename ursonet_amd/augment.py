"""Rotation augmentation with the image work on the GPU ("next" scope row f-1).

Mirrors utils.rotate_cam / utils.rotate_image (utils.py:30-86) as load_image_gt calls them
(net.py:415-438): a random camera rotation R_change is applied to the image as the homography
M = K R_change K^-1 and to the pose (t' = t R^T, q' = q_change (x) q); the soft-classification
target is then re-encoded (utils.encode_ori_fast).  The reference's
`cv2.warpPerspective(image, M, (w, h), cv2.WARP_INVERSE_MAP)` passes the constant in the binding's
`dst` slot, so the call runs with default flags: forward map, INTER_LINEAR, zero border -- the only
reading under which image and pose stay consistent; that is what is implemented.  The warp and the
re-encode run as HIP kernels (urso_warp_perspective, urso_encode_ori) on whole batches that stay
in HBM; the 3x3 / quaternion algebra per sample stays on the host in float64.
"""
import math

import numpy as np


def euler2SO3_left(pitch, yaw, roll):
    """se3lib.py:38-51 (degrees) -> 3x3 ndarray."""
    p, y, r = (float(a) * np.pi / 180 for a in (pitch, yaw, roll))
    cp, sp, cy, sy, cr, sr = np.cos(p), np.sin(p), np.cos(y), np.sin(y), np.cos(r), np.sin(r)
    return np.array([[cy * cr, sp * sy * cr - cp * sr, cp * sy * cr + sp * sr],
                     [cy * sr, sp * sy * sr + cp * cr, cp * sy * sr - sp * cr],
                     [-sy, sp * cy, cp * cy]])


def SO32quat(R):
    """se3lib.py:77-113: rotation matrix -> [x, y, z, w] (largest-diagonal branch selection)."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    d = [R[0, 0], R[1, 1], R[2, 2]]
    if tr > 0:
        Z = math.sqrt(tr + 1) * 2
        return np.array([(R[1, 2] - R[2, 1]) / Z, (R[2, 0] - R[0, 2]) / Z, (R[0, 1] - R[1, 0]) / Z, 0.25 * Z])
    if d[0] > d[1] and d[0] > d[2]:
        i, j, k = 0, 1, 2
    elif d[1] > d[2]:
        i, j, k = 1, 2, 0
    else:
        i, j, k = 2, 0, 1
    Z = math.sqrt(1.0 + 2 * d[i] - tr) * 2
    q = np.zeros(4)
    q[i] = 0.25 * Z
    q[j] = (R[i, j] + R[j, i]) / Z
    q[k] = (R[i, k] + R[k, i]) / Z
    q[3] = (R[j, k] - R[k, j]) / Z
    return q


def quat_mult(a, b):
    """se3lib.py:164-179 for a 1-D b; result normalised."""
    x, y, z, w = (float(v) for v in a)
    b = np.asarray(b, dtype=np.float64).ravel()
    res = np.array([w * b[0] + z * b[1] - y * b[2] + x * b[3],
                    -z * b[0] + w * b[1] + x * b[2] + y * b[3],
                    y * b[0] - x * b[1] + w * b[2] + z * b[3],
                    -x * b[0] - y * b[1] - z * b[2] + w * b[3]])
    return res / np.linalg.norm(res)


def quat2SO3(q):
    """se3lib.py:134-144: [x, y, z, w] -> rotation matrix."""
    x, y, z, w = (float(v) for v in q)
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * (x * y + z * w), 2 * (x * z - y * w)],
                     [2 * (x * y - z * w), 1 - 2 * x * x - 2 * z * z, 2 * (y * z + x * w)],
                     [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * x * x - 2 * y * y]])


def encode_as_keypoints(oris, centroids, scale=1.0):
    """utils.py:220-244: a pose as two virtual 3-D keypoints on the body z and y axes.  One pose -> two 3x1 columns (the caller
    transposes them, net.py:456); a batch [N,4], [N,3] -> two float32 [N,3] arrays."""
    if np.ndim(oris) == 1:
        R = quat2SO3(oris)
        c = np.asarray(centroids, dtype=np.float64).reshape(3, 1)
        return R @ (scale * np.array([[0.0], [0.0], [1.0]])) + c, R @ (scale * np.array([[0.0], [1.0], [0.0]])) + c
    oris, centroids = np.asarray(oris), np.asarray(centroids)
    K1 = np.zeros((len(oris), 3), dtype=np.float32); K2 = np.zeros((len(oris), 3), dtype=np.float32)
    for i in range(len(oris)):
        k1, k2 = encode_as_keypoints(oris[i], centroids[i], scale)
        K1[i], K2[i] = k1[:, 0], k2[:, 0]
    return K1, K2


def rotation_homography(K, R_change):
    K = np.asarray(K, dtype=np.float64)
    return K @ np.asarray(R_change, dtype=np.float64) @ np.linalg.inv(K)


def rotate_pose(t, q, R_change):
    """Pose update of utils.py:53-56."""
    R_change = np.asarray(R_change, dtype=np.float64)
    return np.asarray(t, dtype=np.float64) @ R_change.T, quat_mult(SO32quat(R_change), q)


def invert_homography(M):
    """3x3 inverse in closed form (adjugate / determinant), as cv::invert does for 3x3."""
    m = np.asarray(M, dtype=np.float64).reshape(3, 3)
    c = np.array([[m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1], m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2], m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]],
                  [m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2], m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0], m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]],
                  [m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0], m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1], m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]]])
    det = m[0, 0] * c[0, 0] - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0])
    return c * (1.0 / det)


def warp_images(images, M, inverse_map=False, interp="linear"):
    """cv2.warpPerspective on a batch: images uint8 [B,H,W,C] (device tensor or array), M [B,3,3] forward
    homographies (source -> destination; destination -> source with inverse_map=True).
    Returns a uint8 CUDA tensor [B,H,W,C] (urso_warp_perspective)."""
    import torch
    from . import hip
    x = torch.as_tensor(images)
    assert x.dtype == torch.uint8 and x.dim() == 4, "uint8 [B,H,W,C] images expected"
    x = x.cuda().contiguous()
    B, H, W, C = x.shape
    M = np.asarray(M, dtype=np.float64).reshape(B, 3, 3)
    if not inverse_map:
        M = np.stack([invert_homography(m) for m in M])
    m = torch.as_tensor(np.ascontiguousarray(M.reshape(B, 9))).cuda()
    out = torch.empty_like(x)
    hip.warp_perspective(B, H, W, C, {"nearest": 0, "linear": 1}[interp], x, m, out)
    return out


def encode_orientations(q, H_quat, redundant, beta):
    """Batched utils.encode_ori_fast on the GPU: q [B,4] -> float32 CUDA tensor [B, n^3] of PMFs."""
    import torch
    from . import hip
    K = len(H_quat)
    n = round(K ** (1. / 3))
    var = (float(beta) / n) ** 2 / 12
    qd = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(np.asarray(q, dtype=np.float64)))).cuda()
    hq = torch.as_tensor(np.ascontiguousarray(H_quat, dtype=np.float32)).cuda()
    rd = torch.as_tensor(np.ascontiguousarray(np.asarray(redundant, dtype=bool).astype(np.uint8))).cuda()
    out = torch.empty(qd.shape[0], K, dtype=torch.float32, device="cuda")
    hip.encode_ori(qd.shape[0], K, qd, hq, rd, var, out)
    return out


def rotate_cam_batch(images, t, q, K, pyr_change):
    """Batched rotate_cam for GIVEN per-sample Euler perturbations pyr_change [B,3] (degrees):
    returns (warped uint8 CUDA tensor, t_new [B,3], q_new [B,4])."""
    t, q, pyr = np.atleast_2d(t), np.atleast_2d(q), np.atleast_2d(pyr_change)
    Rs = [euler2SO3_left(*p) for p in pyr]
    Ms = np.stack([rotation_homography(K, R) for R in Rs])
    poses = [rotate_pose(t[i], q[i], Rs[i]) for i in range(len(Rs))]
    return warp_images(images, Ms), np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses])


def rotate_cam(image, t, q, K, magnitude):
    """Drop-in for utils.rotate_cam (same NumPy global-RNG draw): one image [H,W,C] uint8 -> (image_warped ndarray, t_new, q_new)."""
    pyr_change = (np.random.rand(3) - 0.5) * magnitude
    w, tn, qn = rotate_cam_batch(np.asarray(image)[None], t, q, K, pyr_change[None])
    return w[0].cpu().numpy(), tn[0], qn[0]


def rotate_image(image, t, q, K):
    """Drop-in for utils.rotate_image: random in-plane rotation of up to +-85 degrees."""
    change = (np.random.rand(1) - 0.5) * 170
    w, tn, qn = rotate_cam_batch(np.asarray(image)[None], t, q, K, np.array([[0.0, 0.0, change[0]]]))
    return w[0].cpu().numpy(), tn[0], qn[0]


# --------------------------------------------------------------------------- sim2real (net.py:390-406)
SIM2REAL_OPS = ("noise", "blur", "add", "multiply", "dropout")


_PIPELINE_RNG = np.random.RandomState()      # stands for imgaug's own generator: the stage parameters never consume NumPy's global stream


def sim2real_draw(n, height, width, rng=np.random, prng=None):
    """The random decisions of the reference's sim2real branch for `n` samples, sample by sample: whether the pipeline runs (p = 0.5)
    from `rng` (NumPy's global generator by default: net.py:395 `np.random.rand(1) > 0.5`), and -- only for the samples it runs on, as
    imgaug draws when `to_deterministic()` is called (net.py:405) -- the random order of its five stages and each stage's parameter
    from `prng` (imgaug's own generator in the reference, not NumPy's global stream; default: the same generator as `rng`, which the
    distribution tests seed): AdditiveGaussianNoise(scale=0.01*255), GaussianBlur(sigma=(0, 1.5)), Add((-20, 20)), Multiply((0.5, 2.0)),
    CoarseDropout([0.0, 0.03], size_percent=(0.02, 0.1)).  Returns a dict of arrays."""
    prng = rng if prng is None else prng
    apply = np.zeros(n, dtype=bool)
    order = np.tile(np.arange(5), (n, 1))
    par = np.zeros((n, 5, 4), dtype=np.float32)
    seeds = np.zeros(n, dtype=np.uint32)
    masks = []
    for i in range(n):
        apply[i] = rng.rand(1)[0] > 0.5
        par[i, 0, 0] = 0.01 * 255
        par[i, 4, 0], par[i, 4, 1] = 1, 1
        if not apply[i]:
            masks.append(np.zeros((1, 1), dtype=bool))
            continue
        order[i] = prng.permutation(5)
        par[i, 1, 0] = prng.uniform(0.0, 1.5)
        par[i, 2, 0] = float(prng.randint(-20, 21))
        par[i, 3, 0] = prng.uniform(0.5, 2.0)
        p = (0.0, 0.03)[prng.randint(0, 2)]                  # a LIST of two values is a choice in imgaug, not a range
        sp = prng.uniform(0.02, 0.1)
        dh, dw = max(int(height * sp), 1), max(int(width * sp), 1)
        par[i, 4, 0], par[i, 4, 1] = dh, dw
        masks.append(prng.rand(dh, dw) < p)
        seeds[i] = prng.randint(0, 2 ** 31 - 1)
    return {"apply": apply, "order": order, "par": par, "seeds": seeds, "masks": masks}


def sim2real_batch(images, draw=None, rng=np.random):
    """net.py:390-406 for a uint8 batch [B,H,W,3] (array or device tensor): grey conversion, then for the samples whose dice say
    so the five imgaug stages in their drawn order -- five passes of urso_sim2real_op over the batch, ping-ponging two HBM
    buffers.  Returns a uint8 CUDA tensor."""
    import torch
    from . import hip
    x = torch.as_tensor(images)
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[-1] == 3
    x = x.cuda().contiguous()
    B, H, W, _ = x.shape
    d = draw if draw is not None else sim2real_draw(B, H, W, rng)
    a, b = torch.empty_like(x), torch.empty_like(x)
    hip.rgb_to_grey3(B, H, W, x, a)
    if not d["apply"].any():
        return a
    stride = max(m.size for m in d["masks"])
    drop = np.zeros((B, stride), dtype=np.uint8)
    for i, m in enumerate(d["masks"]):
        drop[i, :m.size] = m.reshape(-1)
    drop_d = torch.as_tensor(drop).cuda()
    seed_d = torch.as_tensor(d["seeds"].view(np.int32).copy()).cuda()
    for slot in range(5):
        ops = np.where(d["apply"], d["order"][:, slot], -1).astype(np.int32)
        par = np.stack([d["par"][i, d["order"][i, slot]] for i in range(B)]).astype(np.float32)
        hip.sim2real_op(B, H, W, a, b, torch.as_tensor(ops).cuda(), torch.as_tensor(par).cuda(), seed_d, drop_d, stride)
        a, b = b, a
    return a
