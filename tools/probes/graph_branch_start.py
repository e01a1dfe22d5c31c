#!/usr/bin/env python3
"""Round 6: when does the SECOND branch of a replayed hipGraph start?  A graph of two independent chains of 5-us spin kernels
(n_a on the capture stream, n_b on a forked stream) is replayed a few times under `rocprofv3 --kernel-trace`; tools print the
start offset of each queue's first kernel per replay.  usage: graph_branch_start.py n_a n_b"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")
import torch  # noqa: E402
from infinite_texture_gans_amd import _lib  # noqa: E402

n_a, n_b = int(sys.argv[1]), int(sys.argv[2])
SA, SB = int(os.environ.get("SPIN_A", "5")), int(os.environ.get("SPIN_B", "9"))      # kernel durations of the two chains, us
a_first = len(sys.argv) > 3 and sys.argv[3] == "afirst"      # capture order: the capture stream's chain before the forked one
torch.cuda.set_device(0)
x = torch.zeros(8, device="cuda")


def spin(st, us=5):
    _lib.call("itg_stream_spin", int(us), C.c_void_p(st.cuda_stream))


split = len(sys.argv) > 3 and sys.argv[3] == "split"         # two single-chain graphs launched on two streams instead of one forked graph
side = torch.cuda.Stream()
cap = torch.cuda.Stream()
spin(side), spin(cap)
torch.cuda.synchronize()
if split and len(sys.argv) > 4 and sys.argv[4] == "picked":
    # the forked chain's graph on a stream that ops.concurrent_streams PROBED to run beside the default stream, the other chain's
    # graph on the default stream: the pair of hardware queues the eager step overlaps on
    from infinite_texture_gans_amd import ops
    side = ops.concurrent_streams(torch.device("cuda", 0), 1)[0]
    main = torch.cuda.current_stream()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        ga.capture_begin()
        for _ in range(n_a):
            spin(cap, SA)
        ga.capture_end()
    c2 = torch.cuda.Stream()
    with torch.cuda.stream(c2):
        gb.capture_begin()
        for _ in range(n_b):
            spin(c2, SB)
        gb.capture_end()
    torch.cuda.synchronize()
    for _ in range(6):
        x.add_(1)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gb.replay()
        ga.replay()
        main.wait_stream(side)
        x.add_(1)
    torch.cuda.synchronize()
    print("done split picked", n_a, n_b, float(x[0]))
    sys.exit(0)
if split:
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        ga.capture_begin()
        for _ in range(n_a):
            spin(cap, SA)
        ga.capture_end()
    with torch.cuda.stream(side):
        gb.capture_begin()
        for _ in range(n_b):
            spin(side, SB)
        gb.capture_end()
    torch.cuda.synchronize()
    with torch.cuda.stream(cap):
        for _ in range(6):
            x.add_(1)
            side.wait_stream(cap)
            with torch.cuda.stream(side):
                gb.replay()
            ga.replay()
            cap.wait_stream(side)
            x.add_(1)
    torch.cuda.synchronize()
    print("done split", n_a, n_b, float(x[0]))
    sys.exit(0)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    g.capture_begin()
    x.add_(1)                              # root
    side.wait_stream(cap)
    if a_first:
        for _ in range(n_a):
            spin(cap, SA)
    for _ in range(n_b):
        spin(side, SB)                      # the forked chain's kernels are the 9-us ones
    if not a_first:
        for _ in range(n_a):
            spin(cap, SA)
    cap.wait_stream(side)
    x.add_(1)                              # join
    g.capture_end()
torch.cuda.synchronize()
for _ in range(6):
    g.replay()
torch.cuda.synchronize()
print("done", n_a, n_b, float(x[0]))
