#!/bin/bash
# D's 256 -> 512 layer (fake and real batch): direct kernels vs Winograd F(4 x 4, 4 x 4), forward / input gradient
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for V in 0 1 0 1; do
  echo "== ITG_WINOGRAD=$V"
  ITG_WINOGRAD=$V python3 - <<'PY'
import sys, os, torch
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
from infinite_texture_gans_amd import ops
from conv_bench import timeit
dev = torch.device("cuda")
for (n, P) in [(8, 48), (8, 24)]:
    ci, co = 256, 512
    x = torch.randn(n, 1, 1, P, P, ci, device=dev)
    w = torch.randn(co, ci, 4, 4, device=dev) / (16 * ci) ** 0.5
    b = torch.zeros(co, device=dev)
    f = lambda xx=x: ops.conv(ops.GT(xx, ci), w, b, 4, 4, 1, 1, ops.PAD_ZERO, ops.ACT_LRELU, 0.2, wino=True)
    t = timeit(f)
    xg = x.detach().requires_grad_(True)
    dy = torch.randn_like(f().t)
    f0 = lambda: ops.conv(ops.GT(xg, ci), w, None, 4, 4, 1, 1, ops.PAD_ZERO, wino=True)
    t0 = timeit(lambda: ops.conv(ops.GT(x, ci), w, None, 4, 4, 1, 1, ops.PAD_ZERO, wino=True))
    td = timeit(lambda: torch.autograd.grad(f0().t, xg, dy)) - t0
    print("D3 n=%d %dx%d: fwd %.1f us  dgrad %.1f us (per-call panel packing included)" % (n, P, P, t * 1e6, td * 1e6))
PY
done
