"""Child process of test_gpu_model.py::test_one_rank_rccl_collectives_leave_the_step_unchanged: runs the 2-step
train fixture on a ONE-rank `nccl` (= RCCL) process group with every collective of the data-parallel step forced on
(ITG_FORCE_COLLECTIVES=1 must be in the environment: sync-BN statistics all-reduces in forward and backward, the two
flat gradient all-reduces) and prints the losses + a digest of the post-step parameters as one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from helpers import load, parse_flags, state  # noqa: E402


def main(out_path):
    from infinite_texture_gans_amd import dist as itg_dist, utils as U
    from infinite_texture_gans_amd.engine import Trainer
    from test_gpu_model import build
    assert itg_dist._FORCE, "run with ITG_FORCE_COLLECTIVES=1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        fx = load("train_bn_nl4_sn")
        a = parse_flags(fx["argv"])
        G, D = build(a, state(fx, "G0/"), state(fx, "D0/"))
        G.train(), D.train()
        args = U.prepare_parser().parse_args(["--smooth"])
        args.beta1 = 0.0
        # ITG_TEST_SYNC_BN=0: per-rank statistics, so the stream overlap stays on, the tail buckets are all-reduced on
        # the communication stream under the backward and D(real) of the next iteration runs beside G's head bucket
        sync_bn = os.environ.get("ITG_TEST_SYNC_BN", "1") == "1"
        tr = Trainer(G, D, args, dev, dist_group=dist.group.WORLD, sync_bn=sync_bn)
        assert tr._exchange and all(e.split > 0 for e in tr._exchange.values()), "two-bucket exchange not active"
        losses = []
        steps = int(fx["steps"])
        reals = [torch.from_numpy(fx["real_x%d" % s]).to(dev) for s in range(steps)]
        for s in range(steps):
            nxt = reals[s + 1] if not sync_bn and s + 1 < steps else None
            l = tr.step(reals[s], torch.from_numpy(fx["z%d" % s]).to(dev), None, nxt)
            losses.append([float(v) for v in l])
        torch.save({"G": {k: v.cpu() for k, v in G.state_dict().items()},
                    "D": {k: v.cpu() for k, v in D.state_dict().items()}}, out_path)
        print(json.dumps({"losses": losses, "backend": dist.get_backend(), "world": dist.get_world_size()}), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
