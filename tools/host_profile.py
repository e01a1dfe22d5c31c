import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"]
import bench
import torch
from infinite_texture_gans_amd import ops, utils as U
from infinite_texture_gans_amd.engine import Trainer
dev = torch.device("cuda", 0)
args = U.prepare_parser().parse_args(bench.FLAGS)
args.beta1 = float(args.beta1)
torch.manual_seed(1)
netG, netD = U.prepare_models(args, dev)
netG.train(); netD.train()
tr = Trainer(netG, netD, args, dev)
g = torch.Generator().manual_seed(2)
reals = [(torch.rand(8, 3, 192, 192, generator=g) * 2 - 1).to(dev) for _ in range(2)]
zs = [torch.randn(8, 128, 14, 14, generator=g).to(dev) for _ in range(40)]
for i in range(5):
    tr.step(reals[i % 2], zs[i])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.step(reals[i % 2], zs[5 + i])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
