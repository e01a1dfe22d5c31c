#!/usr/bin/env python3
"""Rounding error and time of D's 256 -> 512 4 x 4 stride-1 layer (reference models/discriminators.py:196-206) on the GPU:
direct kernel vs Winograd F(4 x 4, 4 x 4), each against F.conv2d in fp64 on the CPU.  The switches of the Winograd GEMMs
(ITG_WINO_ACC64) are read once per process: run the tool once per variant.
Usage (GPU box):  python tools/wino_accuracy.py [n_images=8] [size=48]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from infinite_texture_gans_amd import ops  # noqa: E402
from conv_bench import timeit  # noqa: E402

dev = torch.device("cuda")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    cin, cout = 256, 512
    g = torch.Generator().manual_seed(3)
    x = F.leaky_relu(torch.randn(n, cin, size, size, generator=g), 0.2)       # what the layer sees: the previous layer's LeakyReLU output
    w = torch.randn(cout, cin, 4, 4, generator=g) / (cin * 16) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    torch.set_num_threads(16)
    ref = F.conv2d(x[:2].double(), w.double(), b.double(), padding=1)         # fp64 truth on two images
    dy = torch.randn(n, cout, size - 1, size - 1, generator=g)
    dxref = torch.autograd.functional.vjp(lambda t: F.conv2d(t, w.double(), None, padding=1), x[:1].double(), dy[:1].double())[1]
    xg, wg, bg = x.to(dev), w.to(dev), b.to(dev)
    print("switches: ITG_WINO_ACC64=%s" % os.environ.get("ITG_WINO_ACC64", "default"))
    for wino in (False, True):
        ops.WINOGRAD = wino
        gx = ops.to_grid(xg, 1, 1, merged=True)
        y = ops.to_nchw(ops.conv(gx, wg, bg, 4, 4, 1, 1, ops.PAD_ZERO, wino=wino), merged=True)
        kern = ops._lib.fn("itg_last_conv_kernel")().decode()
        xr = xg[:1].clone().requires_grad_(True)
        yy = ops.to_nchw(ops.conv(ops.to_grid(xr, 1, 1, merged=True), wg, None, 4, 4, 1, 1, ops.PAD_ZERO, wino=wino), merged=True)
        dx, = torch.autograd.grad(yy, xr, dy[:1].to(dev))
        t_f = timeit(lambda: ops.conv(ops.GT(gx.t.detach(), cin), wg, bg, 4, 4, 1, 1, ops.PAD_ZERO, wino=wino))
        xl = gx.t.detach().requires_grad_(True)
        dyl = torch.randn_like(ops.conv(ops.GT(xl, cin), wg, None, 4, 4, 1, 1, ops.PAD_ZERO, wino=wino).t)
        t_f0 = timeit(lambda: ops.conv(ops.GT(xl.detach(), cin), wg, None, 4, 4, 1, 1, ops.PAD_ZERO, wino=wino))
        t_d = timeit(lambda: torch.autograd.grad(ops.conv(ops.GT(xl, cin), wg, None, 4, 4, 1, 1, ops.PAD_ZERO, wino=wino).t, xl, dyl)) - t_f0
        print("%-8s fwd rel-L2 vs fp64 %.3e | dgrad rel-L2 vs fp64 %.3e | fwd %.1f us | dgrad %.1f us | %s" % (
            "winograd" if wino else "direct", rel(y[:2].cpu(), ref), rel(dx.cpu(), dxref), t_f * 1e6, t_d * 1e6, kern), flush=True)


if __name__ == "__main__":
    main()
