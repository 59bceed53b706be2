import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine
B = int(os.environ.get("CHK_B", "2"))
cfg = make_config(dtype="bfloat16", backbone="resnet50", h=512, w=640, batch=B, regress_ori=False, ori_bins=16)
img, loc, ori, _ = synthetic_batch(cfg, B, seed=1)
eng = Engine(cfg, "training", seed=3, randomize_bn=True)
eng.load_batch(img, loc, ori)
labs = [l for l in eng.labels["bwd"] if l is not None]
print("mode", os.environ.get("URSO_WGRAD_STREAM"), "side stream", eng.wgrad_stream is not None, "bwd launches", len(labs), flush=True)
if os.environ.get("CHK_EAGER") == "1":
    for _ in range(3): eng.step_eager()
else:
    eng.capture(); print("captured", flush=True)
    for i in range(3):
        eng.step(); torch.cuda.synchronize(); print("replay", i, "ok", flush=True)
torch.cuda.synchronize()
print("done", float(eng.flat_g.abs().sum()), flush=True)
