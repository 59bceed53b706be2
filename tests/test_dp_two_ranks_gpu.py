"""Two REAL ranks on device tensors: two processes share the one GPU of the box and run the data-parallel path through
the drop-in boundary -- `net.UrsoNet(mode='training')` under a launcher's environment (RANK / WORLD_SIZE / MASTER_*), exactly what
`python -m torch.distributed.run --nproc-per-node 2 pose_estimator.py train ...` gives it.  RCCL refuses two ranks on one device, so the
exchange runs over gloo on the same device tensors (URSO_DP_BACKEND=gloo, host-staged: ursonet_amd/dp.py); everything else -- the bucketed
hipGraph segments, the weight broadcast, the rank-sharded feeder, the rank-averaged loss log, the single checkpoint writer -- is the code
an 8-GPU run executes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import make_config, synthetic_batch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(tmp_path, loss_mode):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   URSO_DP_BACKEND="gloo", URSO_DP_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("URSO_DP_FORCE_COLLECTIVES", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "workers", "dp_two_rank_worker.py"), str(tmp_path), loss_mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                    # the exact processes this test started
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out[-4000:])
    res = [json.load(open(os.path.join(str(tmp_path), "res_rank%d_%s.json" % (r, loss_mode)))) for r in range(2)]
    w = [np.load(os.path.join(str(tmp_path), "w_rank%d_%s.npy" % (r, loss_mode))) for r in range(2)]
    return res, w, outs


def _single_process(loss_mode):
    """The same two steps in ONE process on the concatenated batch (4 = 2 ranks x 2), plain Engine."""
    from ursonet_amd.engine import Engine, initial_weights
    cfg = make_config("resnet50", 64, 128, batch=4, regress_ori=False, regress_loc=(loss_mode != "xent"), ori_bins=4, loc_bins=4,
                      dtype="float32", lr=0.01)
    eng = Engine(cfg, "training")
    eng.set_weights(initial_weights(eng.graph, seed=8, randomize_bn=True))
    w0 = eng.flat_w.cpu().numpy().copy()
    eng.hyper[0] = cfg.LEARNING_RATE
    img, loc, ori, _ = synthetic_batch(cfg, 4, seed=11)
    eng.load_batch(img, loc, ori)
    losses = []
    for _ in range(2):
        eng.step()
        losses.append(eng.losses())
    torch.cuda.synchronize()
    return w0, eng.flat_w.cpu().numpy(), losses


def _update_error(w_dp, w_one, w0):
    return float(np.abs((w_dp - w0) - (w_one - w0)).max() / np.abs(w_one - w0).max())


def test_two_ranks_equal_one_process_on_the_concatenated_batch_and_train_through_the_boundary(tmp_path):
    """Decomposable losses (soft-label cross-entropy on both heads, net.py:705-733): after two steps the replicas are bit-identical and
    their weights equal a single process on the concatenated batch to fp32 round-off.  Then UrsoNet.train(): sharded feeding, the logged
    losses are the rank average (identical lists on both ranks), ONE checkpoint per epoch written by rank 0, replicas still identical."""
    res, w, outs = _launch(tmp_path, "xent")
    assert all(r["buckets"] >= 2 for r in res), res           # the step really ran as graph segments between collectives
    assert not any(r["forked"] for r in res)
    assert np.array_equal(w[0], w[1]), "replicas diverged"
    w0, w_one, losses_one = _single_process("xent")
    err = _update_error(w[0], w_one, w0)
    print("two ranks vs one process, update error %.3e" % err)
    assert err < 2e-4, err
    # the global batch's loss is the mean of the shards' losses (equal shard sizes)
    for k in range(2):
        for key in ("loc_loss", "ori_loss"):
            mean = 0.5 * (res[0]["losses"][k][key] + res[1]["losses"][k][key])
            assert abs(mean - losses_one[k][key]) < 2e-4 * abs(losses_one[k][key]) + 1e-6, (k, key, mean, losses_one[k][key])
    # ---- train() through the boundary
    assert res[0]["hist_loc"] == res[1]["hist_loc"] and res[0]["hist_ori"] == res[1]["hist_ori"]         # rank-averaged: the same list everywhere
    assert len(res[0]["hist_loc"]) == 6 and np.isfinite(res[0]["hist_loc"]).all() and np.isfinite(res[0]["hist_ori"]).all()
    assert res[0]["epoch"] == 2 and res[1]["epoch"] == 2
    cks = sorted(f for f in os.listdir(res[0]["log_dir"]) if f.startswith("weights"))
    assert [c for c in cks if c.endswith(".npz")] == ["weights_dp2_0001.npz", "weights_dp2_0002.npz"], cks
    assert "Checkpoint Path" in outs[0] and "Checkpoint Path" not in outs[1]                             # one logger, one writer
    wt = [np.load(os.path.join(str(tmp_path), "w_trained_rank%d.npy" % r)) for r in range(2)]
    assert np.array_equal(wt[0], wt[1]) and not np.array_equal(wt[0], w[0])
    from ursonet_amd.net import read_weights_file
    ck = read_weights_file(os.path.join(res[0]["log_dir"], "weights_dp2_0002.npz"))
    assert np.isfinite(ck["conv1"]["kernel"]).all()


def test_two_ranks_rel_loss_is_per_rank_by_default_and_global_in_exact_mode(tmp_path):
    """rel_loss_graph is a ratio of batch-wide norms (net.py:750-762): with per-rank losses (default, what per-tower averaging gives) two
    ranks do NOT reproduce the one-process step; with DP_EXACT_REL_LOSS (two scalars all-reduced between forward and backward) they do."""
    w0, w_one, _ = _single_process("rel")
    res, w, _ = _launch(tmp_path, "rel")
    assert np.array_equal(w[0], w[1]) and not res[0]["rel_exact"]
    err_default = _update_error(w[0], w_one, w0)
    res, w, _ = _launch(tmp_path, "rel_exact")
    assert np.array_equal(w[0], w[1]) and res[0]["rel_exact"]
    err_exact = _update_error(w[0], w_one, w0)
    print("rel_loss: per-rank %.3e, exact %.3e" % (err_default, err_exact))
    assert err_exact < 2e-4, err_exact
    assert err_default > 20 * max(err_exact, 1e-6) and err_default > 1e-3, (err_default, err_exact)
