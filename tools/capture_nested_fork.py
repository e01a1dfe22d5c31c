#!/usr/bin/env python3
"""Repro / probe for the hipGraph capture topology of the train step's stream overlap (engine.Trainer):

    origin M --fork--> S (D(real) branch) --fork--> W0, W1 (weight-gradient streams) ; W* and S join M before EndCapture

Stage 1 (plain torch ops, no library of ours): M forks S, S forks W, every stream rejoins M.  If hipStreamEndCapture
is unhappy with a fork of a forked stream, this fails.
Stage 1b [skip]: the train step's WHOLE topology with plain torch ops.  It segfaults inside hipStreamEndCapture
(ROCm 7.2) unless step 1 or step 3 is left out (`b 1`, `b 3`): the trigger is waiting for a forked stream that holds no
captured node yet (W* entered the capture by waiting for the origin; S waits for them, then forks kernels onto them).
engine / ops therefore never wait for an idle weight-gradient stream (ops.wgrad_streams_join).
Stage 2: the real train step captured with the nested fork and compared with the eager step.  usage (GPU box): python tools/capture_nested_fork.py [stage]   (run under `timeout`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda")


def stage1():
    a = torch.randn(1 << 20, device=dev)
    outs = [torch.empty_like(a) for _ in range(4)]
    S, W0, W1 = (torch.cuda.Stream() for _ in range(3))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        M = torch.cuda.current_stream()
        W0.wait_stream(M), W1.wait_stream(M)          # step start: the weight-gradient streams wait for the origin
        S.wait_stream(M)
        with torch.cuda.stream(S):
            torch.mul(a, 2.0, out=outs[0])
            for w, o in ((W0, outs[1]), (W1, outs[2])):  # fork of a fork: W waits on an event recorded on S
                ev = torch.cuda.Event()
                ev.record(S)
                w.wait_event(ev)
                with torch.cuda.stream(w):
                    torch.add(outs[0], 1.0, out=o)
        torch.mul(a, 3.0, out=outs[3])                 # origin works meanwhile
        M.wait_stream(S)
        M.wait_stream(W0), M.wait_stream(W1)
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(outs[1], a * 2 + 1) and torch.allclose(outs[2], a * 2 + 1) and torch.allclose(outs[3], a * 3)
    print("stage 1 ok: a fork of a forked stream captures, instantiates and replays", flush=True)


def stage1b(alloc=False, skip=""):
    """The train step's whole stream / event topology with plain torch ops standing in for the kernels: W* wait for the
    origin at the start, the branch S first WAITS for W* (spectral-norm power iteration) and then forks work onto them,
    the origin joins S while W* still carry S-descended work, forks onto W* itself, joins them.  ``alloc``: the forked
    kernels take their scratch from the caching allocator inside the capture, as the weight-gradient calls do."""
    a = torch.randn(1 << 20, device=dev)
    o = [torch.empty_like(a) for _ in range(12)]
    S, W0, W1 = (torch.cuda.Stream() for _ in range(3))
    W = (W0, W1)
    keep = []

    def fork(src, w, inp, out):
        ev = torch.cuda.Event()
        ev.record(src)
        w.wait_event(ev)
        with torch.cuda.stream(w):
            if alloc:
                tmp = torch.empty_like(inp)
                torch.add(inp, 1.0, out=tmp)
                torch.add(tmp, 0.0, out=out)
                keep.append(tmp)
            else:
                torch.add(inp, 1.0, out=out)

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        M = torch.cuda.current_stream()
        if "1" not in skip:
            W0.wait_stream(M), W1.wait_stream(M)                # 1
        S.wait_stream(M)                                        # 2
        with torch.cuda.stream(S):
            if "3" not in skip:
                S.wait_stream(W0), S.wait_stream(W1)            # 3  (power iteration waits for the wgrad streams)
            torch.mul(a, 2.0, out=o[0])
            for i in range(4):                                  # 4  D(real) backward: wgrads leave the branch
                fork(S, W[i & 1], o[0], o[1 + i])
                torch.mul(o[0], 1.0, out=o[5])
        torch.mul(a, 3.0, out=o[6])                             # 5  generator forward on the origin
        M.wait_stream(S)                                        # 6
        if "7" not in skip:
            M.wait_stream(W0), M.wait_stream(W1)                # 7  power iteration of D(fake)
        torch.mul(o[6], 1.0, out=o[7])
        for i in range(4):
            fork(M, W[i & 1], o[7], o[8 + (i & 1)])
        M.wait_stream(W0), M.wait_stream(W1)                    # 8
        keep.clear()
        torch.mul(o[8], 1.0, out=o[10])                         # 9  Adam, G step ...
        M.wait_stream(W0), M.wait_stream(W1)
        fork(M, W0, o[10], o[11])
        M.wait_stream(W0), M.wait_stream(W1)
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(o[4], a * 2 + 1) and torch.allclose(o[11], a * 3 + 2)
    print("stage 1b ok (alloc=%s, skipped steps %r): the step's fork / join topology captures and replays" % (alloc, skip), flush=True)


def stage2():
    import bench
    from infinite_texture_gans_amd import utils as U
    from infinite_texture_gans_amd.engine import Trainer
    args = U.prepare_parser().parse_args(bench.FLAGS)
    args.beta1 = float(args.beta1)
    res = []
    for graph in (False, True):
        torch.manual_seed(3)
        G, D = U.prepare_models(args, dev)
        G.train(), D.train()
        tr = Trainer(G, D, args, dev)
        assert tr.nested_fork and tr.overlap
        gen = torch.Generator().manual_seed(5)
        real = (torch.rand(8, 3, 192, 192, generator=gen) * 2 - 1).to(dev)
        z = torch.randn(8, 128, 14, 14, generator=gen).to(dev)
        if graph:
            tr.capture(real, z, warmup=1)
            print("captured with the nested fork", flush=True)
            l = tr.step_graphed(real, z)
        else:
            tr.step(real, z)
            l = tr.step(real, z)
        torch.cuda.synchronize()
        res.append([float(v) for v in l])
    print("eager", res[0], "graph", res[1], flush=True)
    assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(*res))
    print("stage 2 ok: the captured step with nested weight-gradient forks equals the eager step", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "12"
    if "1" in which:
        stage1()
    if "b" in which:
        stage1b(False, sys.argv[2] if len(sys.argv) > 2 else "")
    if "c" in which:
        stage1b(True)
    if "2" in which:
        stage2()
