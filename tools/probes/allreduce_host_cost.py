#!/usr/bin/env python3
"""Host and device cost of a ONE-rank RCCL all-reduce as issued by torch.distributed (usage, GPU box:
python tools/probes/allreduce_host_cost.py).  The data-parallel step issues 2-4 of them; on one rank they move no data,
so what they cost is what torch's ProcessGroupNCCL adds per call."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.zeros(2_764_737, device=dev)
y = torch.zeros(64 * 1024 * 1024, device=dev)
for _ in range(5):
    dist.all_reduce(x)
torch.cuda.synchronize()
for n, label in ((x, "11 MB"), (y[:16], "64 B")):
    t0 = time.perf_counter()
    for _ in range(200):
        dist.all_reduce(n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.1f us per call, host+device %.1f us per call" % (label, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
# the same between kernels that keep the GPU busy: does the collective stall the stream?
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for with_ar in (False, True):
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        y.mul_(1.0)
        if with_ar:
            dist.all_reduce(x)
    e1.record()
    torch.cuda.synchronize()
    print("50 x (256 MB mul_%s): %.1f us per iteration" % (" + all_reduce" if with_ar else "", e0.elapsed_time(e1) * 1e3 / 50))
dist.destroy_process_group()
