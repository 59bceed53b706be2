"""Dataset base class with the reference's accessor names (dataset.py:5-56) and a synthetic
SPEED/URSO-shaped dataset (there is no dataset on disk and no network here): dark background,
one bright textured blob, Gaussian sensor noise, grey replicated to RGB; poses are random with
north-hemisphere quaternions (speed.py:64-68) and, in classification mode, orientation targets
encoded like speed.py:74 / urso.py:70."""
import numpy as np

from . import utils


class Dataset(object):
    def __init__(self):
        self._image_ids = []
        self.image_info = []

    def add_image(self, source, image_id, path, **kwargs):
        info = {"id": image_id, "source": source, "path": path}
        info.update(kwargs)
        self.image_info.append(info)

    @property
    def image_ids(self):
        return self._image_ids

    def source_image_link(self, image_id):
        return self.image_info[image_id]["path"]

    def load_location(self, image_id):
        return self.image_info[image_id]["location"]

    def load_keypoints(self, image_id):
        return self.image_info[image_id]["keypoints"]

    def load_quaternion(self, image_id):
        return self.image_info[image_id]["quaternion"]

    def load_euler_angles(self, image_id):
        return self.image_info[image_id]["pyr"]

    def load_angle_axis(self, image_id):
        return self.image_info[image_id]["angleaxis"]

    def load_location_encoded(self, image_id):
        return self.image_info[image_id]["location_map"]

    def load_orientation_encoded(self, image_id):
        return self.image_info[image_id]["ori_map"]


class Camera(object):
    """Pinhole intrinsics from the fields of view, as the reference's dataset modules define them
    (urso.py:12-22, speed.py:15-25): fx = W / (2 tan(fov_x/2)), fy = -H / (2 tan(fov_y/2)), principal point at the centre."""

    def __init__(self, width=1280, height=960, fov_x=90.0, fov_y=73.7):
        self.width, self.height = width, height
        self.fov_x, self.fov_y = fov_x * np.pi / 180, fov_y * np.pi / 180
        self.fx = width / (2 * np.tan(self.fov_x / 2))
        self.fy = -height / (2 * np.tan(self.fov_y / 2))
        self.K = np.array([[self.fx, 0, width / 2], [0, self.fy, height / 2], [0, 0, 1]])


class SyntheticPoses(Dataset):
    """`n` synthetic images of `height` x `width` (uint8 RGB) generated on the fly from a seed."""

    def __init__(self, n, height, width, config, seed=0):
        super(SyntheticPoses, self).__init__()
        self.name = "Synthetic"
        self.height, self.width, self.seed = height, width, seed
        self.camera = Camera(width, height)
        rng = np.random.default_rng(seed)
        q = rng.normal(size=(n, 4)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        q[q[:, 3] < 0] *= -1
        t = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(3, 40, n)], 1).astype(np.float32)
        if not config.REGRESS_ORI:
            codec = utils.OrientationCodec(config.ORI_BINS_PER_DIM, config.BETA)
            enc = codec.encode(q)
            self.ori_histogram_map, self.ori_output_mask = codec.H_quat, codec.redundant
        loc_enc = None
        if not config.REGRESS_LOC:
            # classification location head: soft labels over the (x/z, y/z, z) grid, as urso.py:85-93 builds them with utils.encode_loc
            xyz = np.stack([t[:, 0] / t[:, 2], t[:, 1] / t[:, 2], t[:, 2]], axis=1)
            loc_enc, self.loc_histogram_map = utils.encode_loc(xyz, config.LOC_BINS_PER_DIM, config.BETA, xyz.max(0), xyz.min(0))
        for i in range(n):
            self.add_image("SYN", image_id=i, path="synthetic://%d" % i, location=t[i], quaternion=q[i],
                           pyr=np.zeros(3, dtype=np.float32), angleaxis=np.zeros(3, dtype=np.float32),
                           keypoints=[np.zeros(3), np.zeros(3)], location_map=[] if loc_enc is None else loc_enc[i],
                           ori_map=[] if config.REGRESS_ORI else enc[i])
        self._image_ids = np.arange(n)

    def load_image(self, image_id):
        rng = np.random.default_rng(self.seed * 1000003 + int(image_id))
        h, w = self.height, self.width
        yy, xx = np.mgrid[0:h, 0:w]
        img = rng.normal(0, 2.55, size=(h, w))
        cy, cx, r = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w, rng.uniform(0.12, 0.3) * min(h, w)
        blob = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        img += blob * rng.uniform(80, 220) * (0.6 + 0.4 * np.sin(xx / 7.0) * np.cos(yy / 5.0))
        img = np.clip(img, 0, 255).astype(np.uint8)
        return np.repeat(img[:, :, None], 3, axis=2)
