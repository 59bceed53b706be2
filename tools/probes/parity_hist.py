"""Distribution of the bf16 whole-step parity errors (tests/test_model_gpu.py::test_training_step_parity_16bit_same_rounding_points) over data
seeds and kernel plans: how chaotic the 16-bit residue is, which is what the test's gates have to sit above."""
import sys, os, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
import test_model_gpu as T
import ursonet_amd.hip as hip
from oracle import graph_ref as G
dtype = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kw = dict(backbone="resnet50", h=128, w=192, batch=batch, regress_ori=False, ori_bins=8)
for hc2 in (1, 0):
    for pair in (1, 0):
        rows = []
        for seed in range(1, 7):
            cfg = make_config(dtype=dtype, **kw)
            img, loc, ori, _ = synthetic_batch(cfg, batch, seed=seed)
            with hip.options(pair=pair, hconv2=hc2):
                eng, w0 = T._run_engine(cfg, img, loc, ori)
            q = G.StorageRounding(torch.bfloat16 if dtype == "bfloat16" else torch.float16)
            dec = T.ReluDecisions(eng, tol=1.0)
            ref, newW = T._oracle_step(cfg, w0, img, loc, ori, cfg.LEARNING_RATE, relu_hook=dec, q=q)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                try:
                    T._compare_step(eng, ref, newW, 1.0, 1.0, 1.0)
                except AssertionError:
                    pass
            line = buf.getvalue().strip().split("parity: ")[-1]
            d = eval(line.split("} ")[0] + "}")
            rows.append((float(d["loc"]), float(d["ori"]), float(d["grad_big"]), float(d["grad"]), float(d["grad_norm"]), float(d["grad_l2"]), dec.flips / max(dec.total, 1), dec.worst, dec.histogram()))
            del eng
        print("hconv2=%d pair=%d  out max %.2e  grad_big %s  grad(all) max %.2e  norm max %.2e  relu flips max %.1e" % (
            hc2, pair, max(max(r[0], r[1]) for r in rows), " ".join("%.1e" % r[2] for r in rows), max(r[3] for r in rows), max(r[4] for r in rows), max(r[6] for r in rows)), "grad_l2 " + " ".join("%.1e" % r[5] for r in rows), "| worst |pre-act| at a flipped ReLU %.1e, flips by magnitude [<1e-3 <3e-3 <1e-2 <3e-2 <1e-1 >=1e-1] %s" % (
            max(r[7] for r in rows), [sum(r[8][i] for r in rows) for i in range(6)]), flush=True)
