"""Shared helpers for the test-suite (CPU and GPU)."""
import numpy as np


def make_config(backbone="resnet50", h=128, w=192, batch=2, regress_ori=False, regress_loc=True, ori_bins=8, loc_bins=8,
                bottleneck=32, branch=1024, ori_param="quaternion", dtype=None, f16=False, keypoints=False,
                nr_dense=1, wd=1e-4, lr=0.001):
    """Mirror of the CLI's Config mutation (pose_estimator.py:815-872) for an H x W input."""
    from ursonet_amd.config import Config
    c = Config()
    c.ORIENTATION_PARAM = ori_param
    c.ORI_BINS_PER_DIM = ori_bins
    c.LOC_BINS_PER_DIM = loc_bins
    c.NAME = "synthetic"
    c.EPOCHS = 1
    c.NR_DENSE_LAYERS = nr_dense
    c.LEARNING_RATE = lr
    c.BOTTLENECK_WIDTH = bottleneck
    c.BRANCH_SIZE = branch
    c.BACKBONE = backbone
    c.ROT_AUG = False
    c.F16 = f16
    c.OPTIMIZER = "SGD"
    c.REGRESS_ORI = regress_ori
    c.REGRESS_LOC = regress_loc
    c.REGRESS_KEYPOINTS = keypoints
    c.WEIGHT_DECAY = wd
    c.IMAGE_RESIZE_MODE = "pad64"
    c.IMAGE_MAX_DIM = w
    c.IMAGE_MIN_DIM = h
    c.IMAGES_PER_GPU = batch
    c.update()
    if dtype is not None:
        c.COMPUTE_DTYPE = dtype
    return c


def synthetic_batch(config, batch, seed=0):
    """SPEED/URSO-like synthetic inputs: dark background + bright blob + noise, grey
    replicated to 3 channels, mean-subtracted; targets shaped for the configured heads."""
    rng = np.random.default_rng(seed)
    h, w = int(config.IMAGE_SHAPE[0]), int(config.IMAGE_SHAPE[1])
    img = rng.normal(0, 2.55, size=(batch, h, w, 1))
    yy, xx = np.mgrid[0:h, 0:w]
    for b in range(batch):
        cy, cx = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w
        r = rng.uniform(0.12, 0.3) * min(h, w)
        blob = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        img[b, :, :, 0] += blob * rng.uniform(80, 220) * (0.6 + 0.4 * np.sin(xx / 7.0) * np.cos(yy / 5.0))
    img = np.clip(img, 0, 255).round()
    img = np.repeat(img, 3, axis=-1).astype(np.float32) - np.asarray(config.MEAN_PIXEL, dtype=np.float32)
    if config.REGRESS_LOC:
        loc = np.stack([rng.uniform(-1, 1, batch), rng.uniform(-1, 1, batch), rng.uniform(3, 40, batch)], 1).astype(np.float32)
    else:
        k = config.LOC_BINS_PER_DIM ** 3
        z = rng.normal(size=(batch, k)) * 4
        loc = (np.exp(z - z.max(1, keepdims=True)) / np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)).astype(np.float32)
    q = rng.normal(size=(batch, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    if config.REGRESS_ORI:
        ori = q.astype(np.float32) if config.ORIENTATION_PARAM == "quaternion" else q[:, :3].astype(np.float32)
    else:
        k = config.ORI_BINS_PER_DIM ** 3
        z = rng.normal(size=(batch, k)) * 4
        ori = (np.exp(z - z.max(1, keepdims=True)) / np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)).astype(np.float32)
    return img, loc, ori, q
