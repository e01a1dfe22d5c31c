#!/usr/bin/env python3
"""Round 6: what hipGraphLaunch costs the host and whether it lets the host run ahead - host time of each replay() call of the
config-1 step (1 or 2 executable graphs, ITG_GRAPH_EXECS), then the synchronised wall time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")
import torch  # noqa: E402
import bench  # noqa: E402
from infinite_texture_gans_amd import utils as U  # noqa: E402
from infinite_texture_gans_amd.engine import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
args = U.prepare_parser().parse_args(bench.FLAGS)
args.beta1 = float(args.beta1)
torch.manual_seed(args.seed)
netG, netD = U.prepare_models(args, dev)
netG.train(), netD.train()
tr = Trainer(netG, netD, args, dev)
g = torch.Generator().manual_seed(1)
real = (torch.rand(8, 3, 192, 192, generator=g) * 2 - 1).to(dev)
z = torch.randn(8, 128, 14, 14, generator=g).to(dev)
tr.capture(real, z, warmup=3)
for _ in range(4):
    tr.step_graphed(real, z)
torch.cuda.synchronize()
n = 12
ts = []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    tr.step_graphed(real, z)
    ts.append((time.perf_counter() - a) * 1e3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("execs %s: host ms per step_graphed call: %s" % (os.environ.get("ITG_GRAPH_EXECS", "2"), " ".join("%.2f" % t for t in ts)))
print("host enqueue of %d steps %.2f ms, synchronised %.2f ms = %.3f ms / step" % (n, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / n))
