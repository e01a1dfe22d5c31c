#!/usr/bin/env python3
"""GPU time (hipGraph replay) of the BatchNorm forward / backward ops at the generator's tensor sizes of config 1
(8 images x 3 x 3 patches), smallest to largest: where the ~40 BatchNorm launches of a step spend their time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from infinite_texture_gans_amd import ops  # noqa: E402
from conv_bench import timeit  # noqa: E402

dev = torch.device("cuda")
tot_f = tot_b = 0.0
for (c, p) in [(416, 4), (208, 8), (104, 16), (52, 32), (26, 64), (13, 128)]:
    x = ops.GT(torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev), c)
    mb = x.t.numel() * 4 / 1e6
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    stats = torch.zeros(2 * x.ld, device=dev, dtype=torch.float64)
    xs = ops.GT(x.t, c, stats)          # statistics supplied (as the conv epilogue does in the step): finalize_apply only
    t_f = timeit(lambda: ops.bn_act(xs, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, ops.ACT_LRELU, 0.02, False))
    xr = x.t.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    dy = torch.randn_like(x.t)
    t_fb = timeit(lambda: torch.autograd.grad(ops.bn_act(ops.GT(xr, c, stats), gr, br, rm, rv, nbt, True, 1e-5, 0.1, ops.ACT_LRELU, 0.02, False).t,
                                              (xr, gr, br), dy))
    tot_f += t_f; tot_b += t_fb - t_f
    print("C=%3d P=%3d  x %6.1f MB | finalize_apply %6.1f us (%5.2f TB/s) | bwd reduce+apply %6.1f us (%5.2f TB/s)" % (
        c, p, mb, t_f * 1e6, 2 * mb / t_f / 1e6, (t_fb - t_f) * 1e6, 5 * mb / (t_fb - t_f) / 1e6), flush=True)
print("sum over the six sizes: fwd %.1f us, bwd %.1f us" % (tot_f * 1e6, tot_b * 1e6))
