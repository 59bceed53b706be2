"""Rotation augmentation on the GPU (scope row f-1): urso_encode_ori against the reference's own
encode_ori_fast outputs (golden), urso_warp_perspective against the oracle restatement of
cv2.warpPerspective (parity unpinned for the image part: cv2 is absent), and
net.load_image_gt(ROT_AUG / ROT_IMAGE_AUG) end to end against the oracle with the same RNG draws."""
import json
import os

import numpy as np
import pytest
import torch

from util import make_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n", [4, 8, 16])
def test_encode_ori_kernel_against_reference_golden(n):
    from ursonet_amd import augment as A
    g = np.load(os.path.join(GOLD, "ori_codec.npz"))
    beta = json.load(open(os.path.join(GOLD, "meta.json")))["beta"]
    out = A.encode_orientations(g["oris"], g["Hquat_%d" % n], g["red_%d" % n], beta).cpu().numpy()
    fast = g["fast_%d" % n]                                   # float64 PMFs from utils.encode_ori_fast
    assert out.dtype == np.float32 and out.shape == fast.shape
    assert np.abs(out - fast).max() <= 1e-7 + 1e-6 * fast.max()
    assert np.abs(out - g["enc_%d" % n]).max() <= 2e-7        # float32 rows of utils.encode_ori
    assert np.all(out[:, g["red_%d" % n].astype(bool)] == 0)
    assert np.allclose(out.sum(1), 1, atol=1e-5)
    assert np.array_equal(out.argmax(1), fast.argmax(1))


def test_encode_ori_kernel_full_resolution_properties():
    """ori_resolution 24 (13824 bins, BASELINE configs[3]): PMF properties and the peak bin = nearest bin."""
    from ursonet_amd import augment as A
    from ursonet_amd.pose import OrientationCodec
    codec = OrientationCodec(24, 6.0)
    rng = np.random.default_rng(3)
    q = rng.normal(size=(16, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    out = A.encode_orientations(q, codec.H_quat, codec.redundant, 6.0).cpu().numpy()
    ref = codec.encode(q, dtype=np.float64)
    assert np.abs(out - ref).max() <= 1e-6 * ref.max() + 1e-7
    assert np.allclose(out.sum(1), 1, atol=1e-5) and np.all(out[:, codec.redundant] == 0)


@pytest.mark.parametrize("interp", ["linear", "nearest"])
@pytest.mark.parametrize("shape", [(2, 48, 64, 3), (1, 60, 80, 1), (3, 33, 47, 3)])
def test_warp_kernel_against_oracle(shape, interp):
    from ursonet_amd import augment as A
    from ursonet_amd.dataset import Camera
    from oracle import pose_math as P
    B, H, W, C = shape
    rng = np.random.default_rng(B * 100 + H)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    cam = Camera(W, H)
    pyr = (rng.random((B, 3)) - 0.5) * np.array([20, 20, 170])
    Ms = np.stack([A.rotation_homography(cam.K, A.euler2SO3_left(*p)) for p in pyr])
    out = A.warp_images(img, Ms, interp=interp).cpu().numpy()
    for b in range(B):
        ref = P.warp_perspective(img[b], Ms[b], interp=interp)
        mism = (out[b] != ref).any(-1).mean()
        assert mism == 0.0, "sample %d: %.4f%% of pixels differ" % (b, 100 * mism)
        assert (ref == 0).all(-1).mean() < 0.9                 # the case really samples the image
        inv = A.warp_images(img[b:b + 1], P.invert3x3(Ms[b])[None], inverse_map=True, interp=interp).cpu().numpy()[0]
        assert (inv != ref).any(-1).mean() < 0.002             # same map handed over pre-inverted (1-ulp coordinate ties only)
    ident = A.warp_images(img, np.tile(np.eye(3), (B, 1, 1)), interp=interp).cpu().numpy()
    assert np.array_equal(ident, img)
    shift = np.array([[1, 0, 2.0], [0, 1, -3.0], [0, 0, 1]])   # inverse map: dst(x,y) = src(x+2, y-3), zero outside
    sh = A.warp_images(img, np.tile(shift, (B, 1, 1)), inverse_map=True, interp=interp).cpu().numpy()
    assert np.array_equal(sh[:, 3:, :W - 2], img[:, :H - 3, 2:]) and (sh[:, :3] == 0).all() and (sh[:, :, W - 2:] == 0).all()
    half = np.array([[1, 0, 0.5], [0, 1, 0.0], [0, 0, 1]])     # dst(x,y) = src(x+0.5, y): average of neighbours, rounded half up
    if interp == "linear":
        hf = A.warp_images(img, np.tile(half, (B, 1, 1)), inverse_map=True).cpu().numpy()
        exp = (img[:, :, :-1].astype(int) + img[:, :, 1:].astype(int) + 1) >> 1
        assert np.array_equal(hf[:, :, :-1], exp)


def test_rotate_cam_batch_pose_and_image_consistency():
    """Rotating the camera moves the projection of the object's centre exactly as the warp moves pixels."""
    from ursonet_amd import augment as A
    from ursonet_amd.dataset import Camera
    H, W = 240, 320
    cam = Camera(W, H)
    t = np.array([[0.4, -0.3, 8.0]])
    q = np.array([[0.0, 0.0, 0.0, 1.0]])
    # the reference's convention: pixel = K [x/z, y/z, 1] with fy < 0
    u0 = cam.K @ (t[0] / t[0, 2])
    img = np.zeros((1, H, W, 3), dtype=np.uint8)
    cx, cy = int(round(u0[0])), int(round(u0[1]))
    img[0, cy - 2:cy + 3, cx - 2:cx + 3] = 255
    pyr = np.array([[4.0, -6.0, 10.0]])
    out, tn, qn = A.rotate_cam_batch(img, t, q, cam.K, pyr)
    out = out.cpu().numpy()
    ys, xs = np.nonzero(out[0, :, :, 0])
    assert len(ys) > 0
    u1 = cam.K @ (tn[0] / tn[0, 2])
    assert abs(xs.mean() - u1[0]) < 1.5 and abs(ys.mean() - u1[1]) < 1.5
    assert abs(np.linalg.norm(qn[0]) - 1) < 1e-12


@pytest.mark.parametrize("mode", ["cam_class", "image_regress"])
def test_load_image_gt_rotation_augmentation_matches_oracle(mode):
    from ursonet_amd import net
    from ursonet_amd.dataset import SyntheticPoses
    from oracle import pose_math as P
    regress = mode == "image_regress"
    cfg = make_config("resnet18", 64, 128, batch=2, regress_ori=regress, ori_bins=8)
    cfg.ROT_AUG, cfg.ROT_IMAGE_AUG = (mode == "cam_class"), regress
    ds = SyntheticPoses(5, 64, 128, cfg, seed=2)
    hit = 0
    for seed in range(8):
        np.random.seed(seed)
        img, meta, loc, ori = net.load_image_gt(ds, cfg, 1)
        np.random.seed(seed)
        dice = np.random.rand(1)
        raw, t0, q0 = ds.load_image(1), ds.load_location(1), ds.load_quaternion(1)
        apply = (dice > 0.5) if mode == "cam_class" else (dice <= 0.5)
        if not apply:
            assert np.array_equal(img, raw) and np.array_equal(loc, t0)
            continue
        hit += 1
        pyr = (np.random.rand(3) - 0.5) * 20 if mode == "cam_class" else np.array([0, 0, ((np.random.rand(1) - 0.5) * 170)[0]])
        w, tn, qn = P.rotate_cam_given(raw, t0, q0, ds.camera.K, pyr)
        assert np.array_equal(img, w) and not np.array_equal(img, raw)
        assert np.allclose(loc, tn, atol=1e-12)
        if regress:
            assert np.allclose(ori, qn, atol=1e-12)
        else:
            ref = P.encode_ori_fast(qn, cfg.BETA, ds.ori_histogram_map, ds.ori_output_mask)
            assert np.abs(ori - ref).max() <= 1e-6 * ref.max() + 1e-7
    assert hit >= 2


def test_load_image_gt_rotation_augmentation_keypoint_mode():
    """REGRESS_KEYPOINTS + ROT_AUG (net.py:421-424): the two virtual keypoints follow the rotated pose."""
    from ursonet_amd import net
    from ursonet_amd.dataset import SyntheticPoses
    from oracle import pose_math as P
    cfg = make_config("resnet18", 64, 128, batch=2, regress_ori=True, keypoints=True)
    cfg.ROT_AUG = True
    ds = SyntheticPoses(4, 64, 128, cfg, seed=5)
    hit = 0
    for seed in range(8):
        np.random.seed(seed)
        img, meta, loc, k1, k2 = net.load_image_gt(ds, cfg, 2)
        np.random.seed(seed)
        if not (np.random.rand(1) > 0.5):
            continue
        hit += 1
        pyr = (np.random.rand(3) - 0.5) * 20
        w, tn, qn = P.rotate_cam_given(ds.load_image(2), ds.load_location(2), ds.load_quaternion(2), ds.camera.K, pyr)
        r1, r2 = P.encode_as_keypoints(qn, tn)
        assert np.array_equal(img, w) and np.allclose(loc, tn, atol=1e-12)
        assert np.asarray(k1).shape == (1, 3) and np.allclose(k1, r1.T, atol=1e-12) and np.allclose(k2, r2.T, atol=1e-12)
    assert hit >= 2


# ------------------------------------------------------------------ sim2real stages (net.py:390-406)
def _s2r_ref(img, code, par, mask):
    """NumPy restatement of one urso_sim2real_op stage on a uint8 frame [H,W,3] (noise excluded: checked statistically)."""
    f = img.astype(np.float32)
    sat = lambda v: np.clip(np.rint(v), 0, 255).astype(np.uint8)
    if code == 2:
        return sat(f + par[0])
    if code == 3:
        return sat(f * np.float32(par[0]))
    if code == 4:
        dh, dw = int(par[0]), int(par[1])
        H, W = img.shape[:2]
        my = np.minimum(np.arange(H) * dh // H, dh - 1); mx = np.minimum(np.arange(W) * dw // W, dw - 1)
        return np.where(mask[my][:, mx][:, :, None], 0, img).astype(np.uint8)
    if code == 1:
        s = float(par[0])
        if s < 1e-3:
            return img.copy()
        r = int(np.ceil(3 * s))
        k = np.exp(-0.5 * (np.arange(-r, r + 1) / s) ** 2).astype(np.float32)
        pad = np.pad(f, ((r, r), (r, r), (0, 0)), mode="reflect")
        H, W = img.shape[:2]
        acc = np.zeros_like(f); wsum = 0.0
        for dy in range(2 * r + 1):
            for dx in range(2 * r + 1):
                w = k[dy] * k[dx]
                acc += w * pad[dy:dy + H, dx:dx + W]; wsum += w
        return sat(acc / wsum)
    return img.copy()


def test_sim2real_stages_against_numpy_restatement():
    """urso_rgb_to_grey3 and the deterministic stages of urso_sim2real_op (blur / add / multiply / dropout / copy) vs NumPy;
    the Gaussian-noise stage statistically (zero mean, sigma 2.55, identical on the three channels: per_channel=False)."""
    import ursonet_amd.hip as hip
    from ursonet_amd import augment as A
    rng = np.random.default_rng(5)
    B, H, W = 6, 40, 56
    rgb = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    x = torch.as_tensor(rgb).cuda()
    g = torch.empty_like(x)
    hip.rgb_to_grey3(B, H, W, x, g)
    grey = (0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]).astype(np.uint8)
    assert np.array_equal(g.cpu().numpy(), np.repeat(grey[..., None], 3, -1))
    codes = np.array([1, 2, 3, 4, -1, 1], dtype=np.int32)
    par = np.zeros((B, 4), dtype=np.float32)
    par[0, 0], par[1, 0], par[2, 0], par[5, 0] = 1.2, -17.0, 1.7, 0.0005
    masks = rng.random((B, 3 * 5)) < 0.4
    par[3, 0], par[3, 1] = 3, 5
    out = torch.empty_like(g)
    hip.sim2real_op(B, H, W, g, out, torch.as_tensor(codes).cuda(), torch.as_tensor(par).cuda(), torch.zeros(B, dtype=torch.int32, device="cuda"),
                    torch.as_tensor(masks.astype(np.uint8)).cuda(), 15)
    torch.cuda.synchronize()
    o = out.cpu().numpy(); gi = g.cpu().numpy()
    for b in range(B):
        ref = _s2r_ref(gi[b], int(codes[b]), par[b], masks[b].reshape(3, 5))
        d = np.abs(o[b].astype(int) - ref.astype(int)).max()
        assert d <= (1 if codes[b] == 1 else 0), (b, int(codes[b]), d)       # blur: fp32 summation order may move a .5 tie by one level
    # noise
    flat = torch.full((2, 64, 64, 3), 128, dtype=torch.uint8, device="cuda"); outn = torch.empty_like(flat)
    hip.sim2real_op(2, 64, 64, flat, outn, torch.zeros(2, dtype=torch.int32, device="cuda"),
                    torch.tensor([[2.55, 0, 0, 0]] * 2, dtype=torch.float32, device="cuda"),
                    torch.tensor([11, 12], dtype=torch.int32, device="cuda"), None, 0)
    n = outn.cpu().numpy().astype(np.float64) - 128
    assert np.array_equal(n[..., 0], n[..., 1]) and np.array_equal(n[..., 0], n[..., 2])
    assert abs(n.mean()) < 0.15 and 2.2 < n[..., 0].std() < 2.9 and not np.array_equal(n[0], n[1])
    with pytest.raises(hip.UrsoHipError):
        hip.sim2real_op(2, 64, 64, flat, flat, torch.zeros(2, dtype=torch.int32, device="cuda"), torch.zeros(2, 4, device="cuda"),
                        torch.zeros(2, dtype=torch.int32, device="cuda"), None, 0)


def test_load_image_gt_sim2real_branch_end_to_end():
    """net.load_image_gt with SIM2REAL_AUG (net.py:390-413): grey always; with the dice > 0.5 the five stages in the drawn order."""
    from ursonet_amd import net, augment as A
    from ursonet_amd.dataset import SyntheticPoses
    cfg = make_config("resnet50", 64, 128, batch=2, regress_ori=True)
    cfg.SIM2REAL_AUG = True
    ds = SyntheticPoses(4, 64, 128, cfg, seed=4)
    raw = ds.load_image(1)
    np.random.seed(3)
    d = A.sim2real_draw(1, 64, 128)
    np.random.seed(3)
    img, meta, loc, ori = net.load_image_gt(ds, cfg, 1)
    assert img.shape == (64, 128, 3) and img.dtype == np.uint8 and np.array_equal(img[..., 0], img[..., 1])
    grey = (0.2126 * raw[..., 0] + 0.7152 * raw[..., 1] + 0.0722 * raw[..., 2]).astype(np.uint8)
    if not d["apply"][0]:
        assert np.array_equal(img[..., 0], grey)
    else:
        assert not np.array_equal(img[..., 0], grey)
    # the other dice outcome too
    for seed in range(4, 12):
        np.random.seed(seed)
        if A.sim2real_draw(1, 64, 128)["apply"][0] != d["apply"][0]:
            np.random.seed(seed)
            img2 = net.load_image_gt(ds, cfg, 1)[0]
            assert np.array_equal(img2[..., 0], grey) == bool(d["apply"][0])
            break


@pytest.mark.parametrize("m", [4, 8])
def test_encode_loc_kernel_matches_reference_golden(m):
    """urso_encode_loc against the reference's utils.encode_loc outputs (tests/golden, URSO-style limits) and the oracle."""
    from ursonet_amd import pose
    from oracle import pose_math as P
    g = np.load(os.path.join(GOLD, "loc_codec.npz"))
    beta = json.load(open(os.path.join(GOLD, "meta.json")))["beta"]
    out, H = pose.encode_locations(g["locs"], m, beta, g["max_lim"], g["min_lim"])
    assert out.shape == g["enc_%d" % m].shape and np.allclose(H, g["map_%d" % m], rtol=1e-12, atol=1e-12)
    assert np.abs(out.cpu().numpy() - g["enc_%d" % m]).max() < 1e-6
    assert np.abs(out.sum(1).cpu().numpy() - 1).max() < 1e-5
    enc_o, _ = P.encode_loc(g["locs"], m, beta, g["max_lim"], g["min_lim"])
    assert np.abs(out.cpu().numpy() - enc_o).max() < 1e-6
    # full-size head (cfg5: 16 bins per dimension -> 4096) on a seeded batch vs the host implementation
    rng = np.random.default_rng(3)
    xyz = np.stack([rng.uniform(-0.25, 0.25, 32), rng.uniform(-0.2, 0.2, 32), rng.uniform(4, 35, 32)], 1)
    big, _ = pose.encode_locations(xyz, 16, beta, [0.3, 0.25, 40.0], [-0.3, -0.25, 3.0])
    host, _ = pose.encode_loc(xyz, 16, beta, [0.3, 0.25, 40.0], [-0.3, -0.25, 3.0])
    assert np.abs(big.cpu().numpy() - host).max() < 1e-6
