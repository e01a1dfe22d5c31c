#!/usr/bin/env python3
"""Which pool streams run concurrently with the default stream?  (usage, GPU box: python tools/probes/stream_queues.py [rccl])
Prints the duration of one itg_stream_spin on the default stream alone and of a pair (default + candidate) started together."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from infinite_texture_gans_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if len(sys.argv) > 1 and sys.argv[1] == "rccl":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29549")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print("rccl initialised")
cur = torch.cuda.current_stream()
US = 200


def spin(st):
    _lib.call("itg_stream_spin", US, C.c_void_p(st.cuda_stream))


def timed(streams):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for st in streams:
        st.wait_event(e0)
    spin(cur)
    for st in streams:
        spin(st)
    for st in streams:
        cur.wait_stream(st)
    e1.record(cur)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


for _ in range(3):
    timed([])
print("one spin of %d us alone: %.0f us, again %.0f us" % (US, timed([]), timed([])))
cands = [torch.cuda.Stream() for _ in range(10)]
for i, c in enumerate(cands):
    timed([c])
    print("default + pool stream %d: %.0f us, again %.0f us" % (i, timed([c]), timed([c])))
print("default + streams 0,1,2: %.0f us" % timed(cands[:3]))
print("default + streams 0,1,2,3: %.0f us" % timed(cands[:4]))
print("default + streams 0..4: %.0f us" % timed(cands[:5]))
