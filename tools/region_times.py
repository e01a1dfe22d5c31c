#!/usr/bin/env python3
"""Wall time of the phases of one overlapped (eager) train step on the main stream, from HIP events at the phase
boundaries (engine.Trainer.marks), averaged over N steps.  usage (GPU box): python tools/region_times.py [config1|config3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from infinite_texture_gans_amd import ops, utils as U  # noqa: E402
from infinite_texture_gans_amd.engine import Trainer  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "config1"
    dev = torch.device("cuda", 0)
    args = U.prepare_parser().parse_args(bench.FLAGS3 if wl == "config3" else bench.FLAGS)
    if args.bf16:
        ops.mfma_precision("bf16").set()
    args.beta1 = float(args.beta1)
    torch.manual_seed(args.seed)
    netG, netD = U.prepare_models(args, dev)
    netG.train(), netD.train()
    group = None
    if os.environ.get("ITG_FORCE_COLLECTIVES", "0") == "1":     # one-rank RCCL rehearsal: the data-parallel step's collectives
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        print("affinity before init", len(os.sched_getaffinity(0)), "threads", len(os.listdir("/proc/self/task")))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        torch.zeros(1, device=dev); dist.barrier()
        print("affinity after init", len(os.sched_getaffinity(0)), "threads", len(os.listdir("/proc/self/task")))
        group = None if os.environ.get("ITG_RT_INIT_ONLY") == "1" else dist.group.WORLD   # init only: RCCL present, no collective in the step
    tr = Trainer(netG, netD, args, dev, dist_group=group)
    g = torch.Generator().manual_seed(1)
    crop = args.random_crop
    real = (torch.rand(8, 3, crop, crop, generator=g) * 2 - 1).to(dev)
    z = torch.randn(8, 128, 14, 14, generator=g).to(dev)
    for _ in range(5):
        tr.step(real, z)
    torch.cuda.synchronize()
    n = 20
    acc = {}
    order = []
    runs = []
    import time
    t0 = time.perf_counter()
    for _ in range(n):                      # no sync inside: the host runs ahead of the GPU as in the bench
        tr.marks = []
        tr.step(real, z)
        runs.append(tr.marks)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("wall: host issue %.3f ms / step, host + drain %.3f ms / step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    for m in runs[2:]:
        for (a, ea), (b, eb) in zip(m[:-1], m[1:]):
            if b not in acc:
                acc[b] = 0.0
                order.append(b)
            acc[b] += ea.elapsed_time(eb)
    n -= 2
    gap = sum(r0[-1][1].elapsed_time(r1[0][1]) for r0, r1 in zip(runs[2:-1], runs[3:])) / (len(runs) - 3)
    print("%-70s %7.3f ms" % ("(between steps: end of step k -> start mark of step k+1)", gap))
    tr.marks = None
    tot = sum(acc.values())
    for k in order:
        print("%-70s %7.3f ms  %5.1f %%" % (k, acc[k] / n, 100 * acc[k] / tot))
    print("%-70s %7.3f ms" % ("step (main stream, start -> end)", tot / n))


if __name__ == "__main__":
    main()
