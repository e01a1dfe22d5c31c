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
    tr = Trainer(netG, netD, args, dev)
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
    for _ in range(n):                      # no sync inside: the host runs ahead of the GPU as in the bench
        tr.marks = []
        tr.step(real, z)
        runs.append(tr.marks)
    torch.cuda.synchronize()
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
