#!/usr/bin/env python3
"""Soak test: N graph-replayed train steps of config 1 on synthetic data; losses must stay finite and the
graph-replayed run must track an eager run of the same seeds (usage: python tools/soak.py [steps])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from infinite_texture_gans_amd import utils as U  # noqa: E402
from infinite_texture_gans_amd.engine import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
args = U.prepare_parser().parse_args(bench.FLAGS)
args.beta1 = float(args.beta1)


def run(graph):
    torch.manual_seed(3)
    G, D = U.prepare_models(args, dev)
    G.train(), D.train()
    tr = Trainer(G, D, args, dev)
    g = torch.Generator().manual_seed(5)
    tex = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    out = []
    if graph:
        tr.capture(torch.zeros(8, 3, 192, 192, device=dev), torch.zeros(8, 128, 14, 14, device=dev), warmup=0)
    for i in range(steps):
        ys = torch.randint(0, 512 - 192, (8,), generator=g)
        xs = torch.randint(0, 512 - 192, (8,), generator=g)
        real = torch.stack([tex[0, :, y:y + 192, x:x + 192] for y, x in zip(ys, xs)]).to(dev)
        z = torch.randn(8, 128, 14, 14, generator=g).to(dev)
        l = (tr.step_graphed if graph else tr.step)(real, z)
        if i % 20 == 0 or i == steps - 1:
            out.append([float(v) for v in l])
    return out, G


a, Ga = run(True)
b, Gb = run(False)
for i, (x, y) in enumerate(zip(a, b)):
    print(i, ["%.4f" % v for v in x], ["%.4f" % v for v in y])
assert all(all(v == v and abs(v) < 1e3 for v in x) for x in a + b), "non-finite loss"
d = max(float((p - q).abs().max()) for p, q in zip(Ga.parameters(), Gb.parameters()))
print("max |param(graph) - param(eager)| after %d steps: %.3e" % (steps, d))
print("soak ok")
