#!/usr/bin/env python3
"""Is the eager train step host-bound?  Times the host-side enqueue of N steps (no sync) against the synced wall time,
for the overlapped and the single-stream schedule, and a hipGraph replay.  usage (GPU box): python tools/host_pace.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from infinite_texture_gans_amd import utils as U  # noqa: E402
from infinite_texture_gans_amd.engine import Trainer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    args = U.prepare_parser().parse_args(bench.FLAGS)
    args.beta1 = float(args.beta1)
    torch.manual_seed(args.seed)
    netG, netD = U.prepare_models(args, dev)
    netG.train(), netD.train()
    tr = Trainer(netG, netD, args, dev)
    g = torch.Generator().manual_seed(1)
    real = (torch.rand(8, 3, 192, 192, generator=g) * 2 - 1).to(dev)
    z = torch.randn(8, 128, 14, 14, generator=g).to(dev)
    for overlap in (True, False):
        tr.set_overlap(overlap)
        for _ in range(3):
            tr.step(real, z)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(real, z)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # one isolated step: host enqueue time with an empty queue
        t3 = time.perf_counter()
        tr.step(real, z)
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        print("overlap=%d: %d steps enqueue %.2f ms/step, wall %.2f ms/step | single step: enqueue %.2f ms, wall %.2f ms" % (
            overlap, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t4 - t3) * 1e3, (t5 - t3) * 1e3), flush=True)


if __name__ == "__main__":
    main()
