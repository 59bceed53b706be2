"""Configuration object of the drop-in: same attribute names, defaults and derived
fields as the reference's `Config` (config.py:13-196), so that pose_estimator.py's
field-by-field mutation (pose_estimator.py:815-872) works unchanged.

Additions (all optional, never required by a reference caller):
  COMPUTE_DTYPE  "float32" (default; "float16" when F16) | "bfloat16" | "float16" --
                 storage type of activations / MFMA inputs on the MI355X; accumulation,
                 master weights, gradients of parameters and the optimizer are fp32.
  DP_EXACT_REL_LOSS  False: per-rank `rel_loss_graph` (what a tower-parallel Keras model
                 would compute); True: the two squared norms are summed over the ranks between
                 forward and backward, i.e. the loss and its gradient are those of the one global batch.
"""
import json
import os

import numpy as np


class Config(object):
    GPU_COUNT = 1
    IMAGES_PER_GPU = 2
    STEPS_PER_EPOCH = 1000
    VALIDATION_STEPS = 50
    BACKBONE = "resnet101"
    BOTTLENECK_WIDTH = 128
    BRANCH_SIZE = 1024
    IMAGE_RESIZE_MODE = "pad64"
    IMAGE_MIN_DIM = 480
    IMAGE_MAX_DIM = 512
    IMAGE_MIN_SCALE = 0
    NR_IMAGE_CHANNELS = 3
    MEAN_PIXEL = np.array([123.7, 116.8, 103.9])
    LEARNING_RATE = 0.001
    LEARNING_MOMENTUM = 0.9
    CLR = False
    MAX_LEARNING_RATE = 0.0005
    BASE_LEARNING_RATE = 0.0001
    CLR_STEP_SIZE = 4000
    REGRESS_ORI = True
    REGRESS_LOC = True
    REGRESS_KEYPOINTS = False
    ROT_AUG = True
    SIM2REAL_AUG = False
    ROT_IMAGE_AUG = False
    ORIENTATION_PARAM = 'quaternion'
    DECOUPLE_ORIENTATION = False
    LOC_BINS_PER_DIM = 16
    ORI_BINS_PER_DIM = 32
    BETA = 6.0
    OPTIMIZER = 'SGD'
    WEIGHT_DECAY = 0.0001
    F16 = False
    LEARNABLE_LOSS_WEIGHTS = False
    LOSS_WEIGHTS = {"loc_loss": 1., "ori_loss": 1., "k2_loss": 1., "k3_loss": 1.}
    TRAIN_BN = False
    GRADIENT_CLIP_NORM = 5.0

    def update(self):
        """Derived fields (config.py:151-166)."""
        self.BATCH_SIZE = self.IMAGES_PER_GPU * self.GPU_COUNT
        if self.IMAGE_RESIZE_MODE == "crop":
            self.IMAGE_SHAPE = np.array([self.IMAGE_MIN_DIM, self.IMAGE_MIN_DIM, self.NR_IMAGE_CHANNELS])
        elif self.IMAGE_RESIZE_MODE == "pad64":
            self.IMAGE_SHAPE = np.array([self.IMAGE_MIN_DIM, self.IMAGE_MAX_DIM, self.NR_IMAGE_CHANNELS])
        else:
            self.IMAGE_SHAPE = np.array([self.IMAGE_MAX_DIM, self.IMAGE_MAX_DIM, self.NR_IMAGE_CHANNELS])
        self.IMAGE_META_SIZE = 1 + self.NR_IMAGE_CHANNELS + 3 + 4 + 1

    def __init__(self):
        self.LOSS_WEIGHTS = dict(type(self).LOSS_WEIGHTS)
        self.update()

    def display(self):
        print("\nConfigurations:")
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)):
                print("{:30} {}".format(a, getattr(self, a)))
        print("\n")

    def write_to_file(self, filepath):
        """JSON dump of every non-array attribute (config.py:180-196)."""
        d = {}
        for a in dir(self):
            v = getattr(self, a)
            if not a.startswith("__") and not callable(v) and not isinstance(v, np.ndarray):
                d[a] = v
        directory = os.path.dirname(filepath)
        if directory and not os.path.isdir(directory):
            os.makedirs(directory)
        with open(filepath, 'w+') as f:
            f.write(json.dumps(d))


def compute_dtype_name(config):
    """Resolve the activation storage dtype: explicit COMPUTE_DTYPE wins, else F16 (net.py:590-593)."""
    name = getattr(config, "COMPUTE_DTYPE", None)
    if name is None:
        name = "float16" if getattr(config, "F16", False) else "float32"
    if name not in ("float32", "bfloat16", "float16"):
        raise ValueError("COMPUTE_DTYPE must be float32, bfloat16 or float16, got %r" % (name,))
    return name
