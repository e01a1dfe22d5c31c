#!/usr/bin/env python3
"""Achieved HBM GB/s of the memory-bound operators at config-1 tensor sizes (algorithmic bytes /
HIP-event time): LocalPadder (standalone operator), BatchNorm stats / apply(+LeakyReLU, +x2 upsample) /
backward, activation, upsample.  Usage (GPU box): python tools/membound_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from infinite_texture_gans_amd import ops

dev = torch.device("cuda")
PEAK = 8000.0  # GB/s, MI355X HBM3E spec


def timeit(fn, iters=20):
    """GPU time per call (hipGraph replay timed with events: the host's launch cost - 30 us for a BatchNorm forward, 85 us for
    its autograd backward - would otherwise hide the kernels)."""
    from conv_bench import timeit as graph_timeit
    return graph_timeit(fn, iters)


def row(name, nbytes, t):
    print("%-46s %8.1f MB %8.1f us %8.1f GB/s  %5.1f %% of 8 TB/s" % (name, nbytes / 1e6, t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / 1e9 / PEAK), flush=True)


def main():
    only_bn13 = "bn13" in sys.argv          # profiling runs: just the largest BatchNorm tensor
    for (c, p) in ([] if only_bn13 else [(416, 8), (104, 32), (26, 128), (13, 128)]):
        NP = 72
        x = torch.randn(NP, c, p, p, device=dev)
        t = timeit(lambda: ops.local_pad_nchw(x, 3, 3, ops.PAD_REPLICATE))
        row("LocalPadder NCHW fwd  C=%d P=%d" % (c, p), 4 * NP * c * (p * p + (p + 2) ** 2), t)
        g = ops.to_grid(x, 3, 3, merged=False)
        t = timeit(lambda: ops.local_pad_grid(g, ops.PAD_REPLICATE))
        ld = g.ld
        row("LocalPadder NHWC fwd  C=%d(ld %d) P=%d" % (c, ld, p), 4 * NP * ld * (p * p + (p + 2) ** 2), t)
        # backward rows: forward + backward recorded together, the forward's time subtracted (a backward alone would run on an
        # autograd graph built outside the capture: its stale default-stream AccumulateGrad nodes break hipStreamEndCapture)
        xr = x.clone().requires_grad_(True)
        dy = torch.randn_like(ops.local_pad_nchw(x, 3, 3, ops.PAD_REPLICATE))
        t_f = timeit(lambda: ops.local_pad_nchw(x, 3, 3, ops.PAD_REPLICATE))
        t = timeit(lambda: torch.autograd.grad(ops.local_pad_nchw(xr, 3, 3, ops.PAD_REPLICATE), xr, dy)) - t_f
        row("LocalPadder NCHW bwd  C=%d P=%d" % (c, p), 4 * NP * c * (p * p + (p + 2) ** 2), t)
    for (c, p, ups) in ([(13, 128, False)] if only_bn13 else [(13, 128, False), (26, 64, True), (104, 16, True), (416, 4, False)]):
        NP = 72
        xg = ops.GT(torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev), c)
        numel = xg.t.numel()
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
        t_f = timeit(lambda: ops.bn_act(xg, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, ops.ACT_LRELU, 0.02, ups))
        out_mult = 4 if ups else 1
        row("BN train fwd (+lrelu%s) C=%d P=%d" % (",+up2" if ups else "", c, p), 4 * numel * (2 + out_mult), t_f)
        xr = xg.t.clone().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        dy = torch.randn_like(ops.bn_act(xg, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, ops.ACT_LRELU, 0.02, ups).t)
        t = timeit(lambda: torch.autograd.grad(ops.bn_act(ops.GT(xr, c), gr, br, rm, rv, nbt, True, 1e-5, 0.1, ops.ACT_LRELU, 0.02, ups).t,
                                               (xr, gr, br), dy)) - t_f
        row("BN train bwd (+lrelu%s) C=%d P=%d" % (",+up2" if ups else "", c, p), 4 * numel * (2 + 2 * out_mult + 1), t)
    if only_bn13:
        return
    xg = ops.GT(torch.randn(8, 3, 3, 128, 128, 16, device=dev), 13)
    t = timeit(lambda: ops.act(xg, ops.ACT_LRELU, 0.2))
    row("LeakyReLU C=13(ld16) P=128", 8 * xg.t.numel(), t)
    xs = ops.GT(torch.randn(8, 3, 3, 64, 64, 28, device=dev), 26)
    t = timeit(lambda: ops.upsample2x(xs))
    row("nearest x2 upsample C=26(ld28) P=64", 4 * xs.t.numel() * 5, t)


if __name__ == "__main__":
    main()
