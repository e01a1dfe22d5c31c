import os, sys
sys.path.insert(0, os.getcwd())
import torch
from infinite_texture_gans_amd import ops
dev = torch.device("cuda")
ci = co = 13
x = torch.randn(8, 3, 3, 128, 128, ops.ld_for(ci), device=dev); x[..., ci:] = 0
w = torch.randn(co, ci, 3, 3, device=dev) / 10
b = torch.zeros(co, device=dev)
for _ in range(3):
    y = ops.conv(ops.GT(x, ci), w, b, 3, 3, 1, 1, ops.PAD_REPLICATE, out_grid=(3, 3))
torch.cuda.synchronize()
