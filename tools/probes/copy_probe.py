import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import bench, torch
from infinite_texture_gans_amd import ops, utils as U
from infinite_texture_gans_amd.engine import Trainer
dev = torch.device("cuda", 0)
args = U.prepare_parser().parse_args(bench.FLAGS); args.beta1 = float(args.beta1)
torch.manual_seed(1)
netG, netD = U.prepare_models(args, dev); netG.train(); netD.train()
tr = Trainer(netG, netD, args, dev)
g = torch.Generator().manual_seed(2)
real = (torch.rand(8, 3, 192, 192, generator=g) * 2 - 1).to(dev)
zs = [torch.randn(8, 128, 14, 14, generator=g).to(dev) for _ in range(8)]
for i in range(4): tr.step(real, zs[i])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(real, zs[5])
torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ("aten::copy_", "aten::clone", "aten::cat", "aten::fill_", "aten::zero_", "aten::mul", "aten::add", "aten::add_", "aten::div", "aten::sum", "aten::mean")]
import collections
cnt = collections.Counter()
for e in evs:
    st = [s for s in (e.stack or []) if "infinite_texture_gans_amd" in s or "bench.py" in s]
    cnt[(e.name, st[0] if st else "?")] += 1
for k, v in cnt.most_common(40): print(v, k)
