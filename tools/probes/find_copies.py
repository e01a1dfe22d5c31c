#!/usr/bin/env python3
"""Which Python lines issue the device-to-device copies / fills of one train step (torch.profiler with stacks)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from infinite_texture_gans_amd import utils as U  # noqa: E402
from infinite_texture_gans_amd.engine import Trainer  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

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
for _ in range(3):
    tr.step(real, z)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(real, z)
torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat", "aten::add", "aten::mul", "aten::add_", "aten::mul_"):
        st = [s for s in ev.stack if "infinite_texture_gans_amd" in s]
        c[(ev.name, st[0] if st else "?", str(ev.input_shapes)[:60])] += 1
for (name, where, shp), n in sorted(c.items(), key=lambda kv: -kv[1]):
    print("%3d x %-16s %-90s %s" % (n, name, where[-90:], shp))
