"""The forked backward pass (Engine._fork_weight_gradients, URSO_WGRAD_STREAM=2) in a process of its own: tests/test_model_gpu.py starts this
script and reads one JSON line.  A forked hipGraph replay has segfaulted inside the ROCm 7.2 runtime in long-lived processes (engine.py); a
fresh process is where the path is known to work -- and a crash here fails one test instead of taking the suite down.
    python fork_worker.py identity '<make_config kwargs as JSON>'      three replayed steps: single chain / fork / fork rejected -> same bits
    python fork_worker.py stress                                        240 replays of the cfg2-width plan, chain comparison every 20th"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from util import make_config, synthetic_batch      # noqa: E402


def identity(kw):
    from ursonet_amd.engine import Engine
    out, res = {}, {}
    for mode in ("0", "2", "2 rejected"):
        os.environ["URSO_WGRAD_STREAM"] = mode[0]
        cfg = make_config(**kw)
        eng = Engine(cfg, "training", seed=11, randomize_bn=True)
        img, loc, ori, _ = synthetic_batch(cfg, kw["batch"], seed=4)
        eng.load_batch(img, loc, ori)
        if mode == "2 rejected":             # what Engine._verify_forked_graph does when the captured graph fails its check: the same launch list on one chain
            eng._single_chain_always = True
        for _ in range(3):
            eng.step()
        torch.cuda.synchronize()
        out[mode] = (eng.flat_w.clone(), eng.flat_g.clone(), eng.flat_v.clone())
        labs = [l for l in eng.labels["bwd"] if l is not None]
        at = next(i for i, l in enumerate(labs) if l.startswith(("dgrad:res3", "dgrad:res2")))
        r = {"side_stream": eng.wgrad_stream is not None, "forked": bool(eng.forked), "fork_checks": getattr(eng, "fork_checks", 0),
             "wgrads_before_point": sum(1 for l in labs[:at] if l.startswith("wgrad:res"))}
        tail = []
        while at - 1 - len(tail) >= 0 and labs[at - 1 - len(tail)].startswith("wgrad"):
            tail.append(labs[at - 1 - len(tail)])
        r["wgrads_right_in_front_of_point"] = len(tail)
        res[mode] = r
    res["equal"] = {other: all(torch.equal(a, b) for a, b in zip(out["0"], out[other])) for other in ("2", "2 rejected")}
    return res


def stress():
    from ursonet_amd.engine import Engine
    os.environ["URSO_WGRAD_STREAM"] = "2"
    cfg = make_config(backbone="resnet50", h=512, w=640, batch=2, regress_ori=False, ori_bins=16, dtype="bfloat16", lr=1e-3)
    eng = Engine(cfg, "training", seed=11, randomize_bn=True)
    eng.capture()                                    # nothing loaded yet: the check must not run on zeros
    res = {"forked_on_empty_batch": bool(eng.forked), "fork_checks_after_capture": getattr(eng, "fork_checks", 0),
           "buffers_left_zero": not bool(eng.in_images.any()) and not bool(eng.gt_loc.any()) and not bool(eng.gt_ori.any())}
    compared, bad = 0, []
    for k in range(240):
        if k % 20 == 0:
            img, loc, ori, _ = synthetic_batch(cfg, 2, seed=100 + k)
            eng.load_batch(img, loc, ori)
        if k % 20 == 19:
            saved = eng.save_train_state()
            eng.step()
            torch.cuda.synchronize()
            got = [t.clone() for t in (eng.flat_w, eng.flat_g, eng.flat_v)]
            eng.restore_train_state(saved)
            eng._single_chain = True
            try:
                eng.step_eager()
            finally:
                eng._single_chain = False
            torch.cuda.synchronize()
            if not all(torch.equal(a, b) for a, b in zip(got, (eng.flat_w, eng.flat_g, eng.flat_v))):
                bad.append(k)
            compared += 1
        else:
            eng.step()
    res.update(compared=compared, replays_that_differ=bad, finite=bool(torch.isfinite(eng.flat_g).all()), forked_at_end=bool(eng.forked))
    w = eng.flat_w.clone()
    res["verify_fork"] = bool(eng.verify_fork(replays=3))
    res["fork_checks_at_end"] = eng.fork_checks
    res["state_untouched_by_verify"] = bool(torch.equal(w, eng.flat_w))
    return res


if __name__ == "__main__":
    r = identity(json.loads(sys.argv[2])) if sys.argv[1] == "identity" else stress()
    print("RESULT " + json.dumps(r), flush=True)
