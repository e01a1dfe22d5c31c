#!/usr/bin/env python3
"""Round-3 record gpurun_out/r3al.log: a probe segfaulted inside
    torch.autograd.grad(ops.conv(GT(xg, ci), w, None, 3, 3, 1, 1, PAD_REPLICATE).t, xg, dy)
when the call was recorded into a hipGraph by conv_bench.timeit.  This script runs that call in child processes, one variant
each, with faulthandler on, and reports how each ended:
  eager        the call itself, no capture, checked against F.conv2d on the CPU (the C ABI / halo-tile input gradient)
  capture      the call under capture with NO autograd graph of xg alive from before
  stale        the probe's exact sequence: y = conv(GT(xg)) computed on the default stream and KEPT ALIVE, then the capture
Usage (GPU box): python tools/probes/capture_stale_graph.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tools"))
import torch, torch.nn.functional as F
from infinite_texture_gans_amd import ops
from conv_bench import timeit
variant = %(variant)r
dev = torch.device("cuda")
for (ci, co, P) in [(26, 26, 64), (26, 13, 128)]:
    torch.manual_seed(0)
    x = torch.randn(8, 3, 3, P, P, ops.ld_for(ci), device=dev); x[..., ci:] = 0
    w = torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5
    xg = x.clone().requires_grad_(True)
    y = ops.conv(ops.GT(xg, ci), w, None, 3, 3, 1, 1, ops.PAD_REPLICATE)
    dy = torch.randn_like(y.t)
    if variant == "eager":
        dx, = torch.autograd.grad(y.t, xg, dy)
        kern = ops._lib.fn("itg_last_conv_kernel")().decode()
        # reference: merged image, replicate frame, F.conv2d on the CPU
        xm = ops.to_nchw(ops.GT(x, ci), merged=True)[:2].cpu().double().requires_grad_(True)
        yr = F.conv2d(F.pad(xm, (1, 1, 1, 1), mode="replicate"), w.cpu().double())
        dym = ops.to_nchw(ops.GT(dy, co), merged=True)[:2].cpu().double()
        dxr, = torch.autograd.grad(yr, xm, dym)
        got = ops.to_nchw(ops.GT(dx, ci), merged=True)[:2].cpu().double()
        err = float((got - dxr).norm() / dxr.norm())
        print("%%d->%%d P%%d eager dgrad rel-L2 %%.2e (%%s)" %% (ci, co, P, err, kern), flush=True)
        assert err < 5e-6, err
        continue
    if variant == "capture":
        del y                                   # nothing of xg's autograd graph survives from the default stream
    td = timeit(lambda: torch.autograd.grad(ops.conv(ops.GT(xg, ci), w, None, 3, 3, 1, 1, ops.PAD_REPLICATE).t, xg, dy))
    print("%%d->%%d P%%d %%s: fwd+dgrad %%.1f us under capture (%%s)" %% (ci, co, P, variant, td * 1e6, ops._lib.fn("itg_last_conv_kernel")().decode()), flush=True)
print("variant %%s finished" %% variant, flush=True)
'''


def main():
    for variant in ("eager", "capture", "stale"):
        code = CHILD % {"root": ROOT, "variant": variant}
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-14:])
        print("==== %s: exit code %d\n%s\n" % (variant, r.returncode, tail), flush=True)


if __name__ == "__main__":
    main()
