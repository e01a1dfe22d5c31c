#!/usr/bin/env python3
"""On which stream does torch's ProcessGroupNCCL run a blocking (async_op=False) collective issued under
torch.cuda.stream(s)?  One-rank group, all_gather_into_tensor (a copy on one rank); run under
`rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` and compare the Stream_Id / Queue_Id columns of the
copy with those of the two fill kernels that bracket it."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29551")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
s = [torch.cuda.Stream() for _ in range(3)][2]
x = torch.zeros(1 << 20, device=dev)
out = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()
with torch.cuda.stream(s):
    for _ in range(3):
        x.fill_(1.0)
        dist.all_gather_into_tensor(out, x)
        out.fill_(2.0)
        w = dist.all_reduce(x, async_op=True)
        w.wait()
        x.fill_(3.0)
torch.cuda.synchronize()
print("stream handle", hex(s.cuda_stream))
dist.destroy_process_group()
