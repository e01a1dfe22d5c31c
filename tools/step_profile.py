#!/usr/bin/env python3
"""Per-call view of one un-overlapped train step at the bench configuration: every convolution call in issue
order with its kernel instantiation, algorithmic GFLOP, HIP-event time and TFLOP/s (ops.PROFILE), and the
per-phase totals.  Usage (GPU box): python tools/step_profile.py [config1|config3]"""
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
    tr.set_overlap(False)
    g = torch.Generator().manual_seed(1)
    crop = args.random_crop
    real = (torch.rand(8, 3, crop, crop, generator=g) * 2 - 1).to(dev)
    z = torch.randn(8, 128, 14, 14, generator=g).to(dev)
    for _ in range(3):
        tr.step(real, z)
    torch.cuda.synchronize()
    ops.PROFILE = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tr.step(real, z)
    e1.record()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    tot_t = tot_f = 0.0
    print("%3s %-52s %9s %9s %8s" % ("#", "kernel", "GFLOP", "us", "TF"))
    for i, (tag, nl, fl, a, b, _nb) in enumerate(prof):
        us = a.elapsed_time(b) * 1e3
        tot_t += us
        tot_f += fl
        print("%3d %-52s %9.3f %9.1f %8.1f" % (i, tag, fl / 1e9, us, fl / us / 1e6))
    print("conv calls: %d, %.1f GF in %.3f ms = %.1f TF; whole un-overlapped step (host-paced) %.3f ms" % (
        len(prof), tot_f / 1e9, tot_t / 1e3, tot_f / tot_t / 1e6, e0.elapsed_time(e1)))


if __name__ == "__main__":
    main()
