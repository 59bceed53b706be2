"""One rank of tests/test_dp_two_ranks_gpu.py: started twice (RANK 0 / 1, WORLD_SIZE 2, both on the one GPU of the box, URSO_DP_BACKEND=gloo
because RCCL refuses two ranks on one device) the way a launcher starts `pose_estimator.py train`.  Goes through the drop-in boundary only:
net.UrsoNet(mode='training') finds the launcher's environment, builds its engine for IMAGES_PER_GPU and wraps it in DataParallelEngine."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from util import make_config, synthetic_batch      # noqa: E402


def main():
    out_dir, loss_mode = sys.argv[1], sys.argv[2]
    rank = int(os.environ["RANK"])
    from ursonet_amd import net
    from ursonet_amd import dp as dpm
    from ursonet_amd.engine import initial_weights
    from ursonet_amd.dataset import SyntheticPoses
    import torch.distributed as dist

    cfg = make_config("resnet50", 64, 128, batch=2, regress_ori=False, regress_loc=(loss_mode != "xent"), ori_bins=4, loc_bins=4,
                      dtype="float32", lr=0.01)
    cfg.NAME = "dp2"
    cfg.DP_EXACT_REL_LOSS = loss_mode == "rel_exact"
    cfg.STEPS_PER_EPOCH, cfg.VALIDATION_STEPS = 3, 1
    model = net.UrsoNet(mode="training", config=cfg, model_dir=out_dir)
    eng, dp = model._engine, model._dp
    assert dp is not None and dp.world == 2 and eng.B == 2, "UrsoNet did not take the data-parallel path under the launcher's environment"
    assert dist.is_initialized() and dist.get_world_size() == 2 and dist.get_backend() == "gloo"
    res = {"rank": rank, "buckets": len(dp.buckets), "forked": bool(eng.forked), "rel_exact": bool(dp.rel_exact)}

    # ---- A. two steps on this rank's shard of one global batch, from weights only rank 0 holds until they are broadcast
    w0 = initial_weights(eng.graph, seed=8 + 5 * rank, randomize_bn=True)         # rank 1 starts from OTHER weights: the broadcast must replace them
    eng.set_weights(w0)
    dpm.broadcast_(eng.flat_w, 0)
    dpm.broadcast_(eng.flat_stats, 0)
    model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
    img, loc, ori, _ = synthetic_batch(cfg, 4, seed=11)
    sl = slice(2 * rank, 2 * rank + 2)
    eng.load_batch(img[sl], loc[sl], ori[sl])
    losses = []
    for _ in range(2):
        dp.step()
        losses.append(eng.losses())
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "w_rank%d_%s.npy" % (rank, loss_mode)), eng.flat_w.cpu().numpy())
    res["losses"] = losses

    # ---- B. the reference's entry point: train() with rank-sharded feeding, rank-averaged logged losses, one checkpoint writer
    if loss_mode == "xent":
        ds_train, ds_val = SyntheticPoses(16, 64, 128, cfg, seed=1), SyntheticPoses(8, 64, 128, cfg, seed=2)
        hist = model.train(ds_train, ds_val, learning_rate=cfg.LEARNING_RATE, epochs=2, layers="all")
        torch.cuda.synchronize()
        res["hist_loc"], res["hist_ori"] = hist.loc_loss_acc, hist.ori_loss_acc
        res["epoch"] = model.epoch
        res["log_dir"] = model.log_dir
        np.save(os.path.join(out_dir, "w_trained_rank%d.npy" % rank), eng.flat_w.cpu().numpy())
    with open(os.path.join(out_dir, "res_rank%d_%s.json" % (rank, loss_mode)), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dp.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
