import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
for shape in (dict(h=256, w=320, batch=4), dict(h=512, w=640, batch=32)):
    cfg = make_config(backbone="resnet50", regress_ori=False, ori_bins=16, dtype="bfloat16", **shape)
    eng = Engine(cfg, "training", seed=7, randomize_bn=True)
    img, loc, ori, _ = synthetic_batch(cfg, shape["batch"], seed=3)
    eng.load_batch(img, loc, ori)
    eng.step_eager()
    for r in eng.profile_step():
        if r[0].startswith(("dgrad:res3d_branch2c", "dgrad:res3d_branch2b", "wgrad:res5a_branch2b")):
            print(shape, r[0][:50], "%.1f us" % (r[2] * 1e3), r[6][:90], "launches", r[5])
    c = eng.convs["res3d_branch2c"]
    print("  ws_d", c.ws_d, "halo_d", c.halo_d, "gd_scatter", getattr(c, "gd_scatter", None), "gd_compact", getattr(c, "gd_compact", None) is not None)
