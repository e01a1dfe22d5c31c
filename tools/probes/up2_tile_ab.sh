#!/bin/bash
# b6c1-shaped folded-upsample forward: generic parity-class kernel vs the folded halo-tile kernel (same box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for V in 0 1 0 1; do
  echo "== ITG_UP2_TILE=$V"
  ITG_UP2_TILE=$V python3 - <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
from infinite_texture_gans_amd import ops
from conv_bench import timeit
dev = torch.device("cuda")
for (ci, co, P) in [(26, 13, 64), (52, 26, 32), (13, 13, 64)]:
    x = torch.randn(8, 3, 3, P, P, ops.ld_for(ci), device=dev); x[..., ci:] = 0
    w = torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5
    b = torch.zeros(co, device=dev)
    t = timeit(lambda: ops.conv(ops.GT(x, ci), w, b, 3, 3, 1, 1, ops.PAD_REPLICATE, up2=True, out_stats=True))
    print("up2 fwd %d->%d source P%d: %.1f us  (%s)" % (ci, co, P, t * 1e6, ops._lib.fn("itg_last_conv_kernel")().decode()))
PY
done
