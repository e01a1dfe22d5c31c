#!/usr/bin/env python3
"""Per-shape microbenchmark of the conv kernels (fwd / dgrad / wgrad) at the config-1 layer shapes.
Usage (GPU box):  python tools/conv_bench.py [filter]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from infinite_texture_gans_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda")
# name, n, grid, P(in patch), cin, cout, k, stride, pad, mode
SHAPES = [
    ("D0 fake 3->64 s2", 8, (3, 3), 128, 3, 64, 4, 2, 1, "zero"),
    ("D1 fake 64->128 s2", 8, (1, 1), 192, 64, 128, 4, 2, 1, "zero"),
    ("D2 fake 128->256 s2", 8, (1, 1), 96, 128, 256, 4, 2, 1, "zero"),
    ("D3 fake 256->512 s1", 8, (1, 1), 48, 256, 512, 4, 1, 1, "zero"),
    ("D4 fake 512->1 s1", 8, (1, 1), 47, 512, 1, 4, 1, 1, "zero"),
    ("D1 real 64->128 s2", 8, (1, 1), 96, 64, 128, 4, 2, 1, "zero"),
    ("D2 real 128->256 s2", 8, (1, 1), 48, 128, 256, 4, 2, 1, "zero"),
    ("D3 real 256->512 s1", 8, (1, 1), 24, 256, 512, 4, 1, 1, "zero"),
    ("G b1 416->416 P4", 8, (3, 3), 4, 416, 416, 3, 1, 1, "rep"),
    ("G b2c1 416->208 P8", 8, (3, 3), 8, 416, 208, 3, 1, 1, "rep"),
    ("G b3c1 208->104 P16", 8, (3, 3), 16, 208, 104, 3, 1, 1, "rep"),
    ("G b4c1 104->52 P32", 8, (3, 3), 32, 104, 52, 3, 1, 1, "rep"),
    ("G b5c1 52->26 P64", 8, (3, 3), 64, 52, 26, 3, 1, 1, "rep"),
    ("G b6c1 26->13 P128", 8, (3, 3), 128, 26, 13, 3, 1, 1, "rep"),
    ("G b6c2 13->13 P128", 8, (3, 3), 128, 13, 13, 3, 1, 1, "rep"),
    ("G final 13->3 P128", 8, (3, 3), 128, 13, 3, 3, 1, 1, "rep"),
]
# what-if shapes (only with a filter that names them): stride / K length / width of D's layers varied one at a time
EXTRA = [
    ("X D1 as stride 1 (97^2 in)", 8, (1, 1), 97, 64, 128, 4, 1, 1, "zero"),
    ("X D1 K=4096 (256->128 s2)", 8, (1, 1), 192, 256, 128, 4, 2, 1, "zero"),
    ("X D3 K=1024 (64->512 s1)", 8, (1, 1), 48, 64, 512, 4, 1, 1, "zero"),
    ("X D1 co=512 (64->512 s2)", 8, (1, 1), 192, 64, 512, 4, 2, 1, "zero"),
    ("X D1 K=64 (4->128 s2): fixed cost", 8, (1, 1), 192, 4, 128, 4, 2, 1, "zero"),
    ("X D1 K=256 (16->128 s2)", 8, (1, 1), 192, 16, 128, 4, 2, 1, "zero"),
    ("X D1 K=512 (32->128 s2)", 8, (1, 1), 192, 32, 128, 4, 2, 1, "zero"),
    # proxies of the upsample-folded c1 layers (conv3x3 o nearest x2 = the transpose of a 4x4 stride-2 conv from the
    # high-resolution side): dgrad column ~ folded forward, fwd column ~ folded input gradient, wgrad ~ folded wgrad
    ("X fold b2c1 (208->416 s2, 24^2)", 8, (1, 1), 24, 208, 416, 4, 2, 1, "zero"),
    ("X fold b3c1 (104->208 s2, 48^2)", 8, (1, 1), 48, 104, 208, 4, 2, 1, "zero"),
    ("X fold b4c1 (52->104 s2, 96^2)", 8, (1, 1), 96, 52, 104, 4, 2, 1, "zero"),
    ("X fold b5c1 (26->52 s2, 192^2)", 8, (1, 1), 192, 26, 52, 4, 2, 1, "zero"),
    ("X fold b6c1 (13->26 s2, 384^2)", 8, (1, 1), 384, 13, 26, 4, 2, 1, "zero"),
    ("X tile b5c2 26->26 P64", 8, (3, 3), 64, 26, 26, 3, 1, 1, "rep"),       # halo-tile kernel with 76 KB of LDS (the kernels' 80 KB cap)
    ("X tile b4c2 52->52 P32", 8, (3, 3), 32, 52, 52, 3, 1, 1, "rep"),
    # the folded-upsample layers themselves (mode "rep-up2": ops.conv(up2=True) on the half-size input; P = source patch)
    ("X up2 b6c1 26->13 P64", 8, (3, 3), 64, 26, 13, 3, 1, 1, "rep-up2"),
    ("X up2 b5c1 52->26 P32", 8, (3, 3), 32, 52, 26, 3, 1, 1, "rep-up2"),
    ("X up2 b4c1 104->52 P16", 8, (3, 3), 16, 104, 52, 3, 1, 1, "rep-up2"),
    # the generator's wide layers as the step runs them (round 6: un-split plan experiments; filter "XW")
    ("XW b1c 416->416 P4", 8, (3, 3), 4, 416, 416, 3, 1, 1, "rep"),
    ("XW b2c1 up2 416->208 P4", 8, (3, 3), 4, 416, 208, 3, 1, 1, "rep-up2"),
    ("XW b2c2 208->208 P8", 8, (3, 3), 8, 208, 208, 3, 1, 1, "rep"),
    ("XW b2sc 1x1 416->208 P4", 8, (3, 3), 4, 416, 208, 1, 1, 0, "zero"),
    ("XW b3c1 up2 208->104 P8", 8, (3, 3), 8, 208, 104, 3, 1, 1, "rep-up2"),
    ("XW b3c2 104->104 P16", 8, (3, 3), 16, 104, 104, 3, 1, 1, "rep"),
    ("XW b3sc 1x1 208->104 P8", 8, (3, 3), 8, 208, 104, 1, 1, 0, "zero"),
]


def timeit(fn, iters=10):
    """GPU time per call: the calls are recorded into a hipGraph so that host launch overhead (which
    exceeds the kernel time of the small layers) does not enter."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    if flt.startswith("X"):
        SHAPES.extend(EXTRA)
    tot = [0.0, 0.0, 0.0]
    for name, n, (gh, gw), p, ci, co, k, s, pad, mode in SHAPES:
        if flt and flt not in name:
            continue
        x = torch.randn(n, gh, gw, p, p, ops.ld_for(ci), device=dev)
        x[..., ci:] = 0
        w = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
        b = torch.zeros(co, device=dev)
        up2 = mode.endswith("-up2")
        pm = ops.PAD_REPLICATE if mode.startswith("rep") else ops.PAD_ZERO
        og = (gh, gw) if k in (1, 3) else (1, 1)
        gx = ops.GT(x.requires_grad_(True), ci)
        wq = w.requires_grad_(True)
        y = ops.conv(gx, wq, b, k, k, s, pad, pm, out_grid=og, up2=up2)
        npix = y.t.numel() // y.t.shape[-1]
        flops = 2.0 * npix * co * ci * (4 if up2 else k * k)
        dy = torch.randn_like(y.t)
        t_f = timeit(lambda: ops.conv(ops.GT(x.detach(), ci), w.detach(), b, k, k, s, pad, pm, out_grid=og, up2=up2))
        # dgrad only / wgrad only: (bias-free forward + backward) recorded together, the bias-free forward's time subtracted
        t_f0 = timeit(lambda: ops.conv(ops.GT(x.detach(), ci), w.detach(), None, k, k, s, pad, pm, out_grid=og, up2=up2))
        xg = x.detach().requires_grad_(True)
        t_d = timeit(lambda: torch.autograd.grad(
            ops.conv(ops.GT(xg, ci), w.detach(), None, k, k, s, pad, pm, out_grid=og, up2=up2).t, xg, dy)) - t_f0
        wg = w.detach().requires_grad_(True)
        t_w = timeit(lambda: torch.autograd.grad(
            ops.conv(ops.GT(x.detach(), ci), wg, None, k, k, s, pad, pm, out_grid=og, up2=up2).t, wg, dy)) - t_f0
        tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
        print("%-24s %7.2f GF | fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % (
            name, flops / 1e9, t_f * 1e6, flops / t_f / 1e12, t_d * 1e6, flops / t_d / 1e12, t_w * 1e6,
            flops / t_w / 1e12), flush=True)
    print("sum fwd %.2f ms dgrad %.2f ms wgrad %.2f ms" % tuple(t * 1e3 for t in tot))


if __name__ == "__main__":
    main()
