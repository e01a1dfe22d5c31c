#!/usr/bin/env python3
"""Round 6: where the per-step panel job (itg_pack_multi) spends its time - every job of D's and G's PackSet alone and together
(hipGraph-timed, GPU time per launch).  Usage (GPU box): python tools/probes/pack_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench  # noqa: E402
from conv_bench import timeit  # noqa: E402
from infinite_texture_gans_amd import ops, utils as U  # noqa: E402
from infinite_texture_gans_amd.engine import PackSet  # noqa: E402

dev = torch.device("cuda", 0)
args = U.prepare_parser().parse_args(bench.FLAGS)
args.beta1 = float(args.beta1)
netG, netD = U.prepare_models(args, dev)
KIND = {0: "plain fwd", 1: "plain dgrad", 2: "up2 fwd", 3: "up2 dgrad", 4: "F44 fwd", 5: "F44 dgrad", 6: "F43 fwd", 7: "F43 dgrad",
        8: "F42 fwd", 9: "F42 dgrad^T"}
for name, net in (("D", netD), ("G", netG)):
    ps = PackSet(net)
    tot = sum(j[1].numel() for j in ps.jobs) * 4 / 1e6
    t = timeit(ps.repack, iters=20)
    print("%s: %d jobs, %.1f MB of panels, %.1f us per launch = %.2f TB/s" % (name, len(ps.jobs), tot, t * 1e6, tot / t / 1e6))
    for j in ps.jobs:
        tb = ops.pack_tables([j], dev)
        t = timeit(lambda: ops.pack_multi(tb), iters=20)
        mb = j[1].numel() * 4 / 1e6
        print("   %-12s co %4d ci %4d k %d s %d  %7.2f MB  %6.1f us  %.2f TB/s" % (KIND[j[8]], j[2], j[3], j[5], j[7], mb, t * 1e6, mb / t / 1e6))
