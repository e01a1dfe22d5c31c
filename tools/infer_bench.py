#!/usr/bin/env python3
"""Inference tiling throughput (BASELINE config 5 shape): one-shot patch-grid generation of a large image
vs the reference's streamed 3x3 schedule.  Random-init weights, eval-mode norm.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinite_texture_gans_amd import utils as U

dev = torch.device("cuda")


def run(norm, out, streamed):
    args = U.prepare_parser().parse_args(["--padding_mode", "local", "--type_norm", norm, "--n_layers_G", "6", "--leak_G", "0.02"])
    torch.manual_seed(0)
    G, _ = U.prepare_models(args, dev)
    G.eval()
    kw = dict(z_dim=128, base_res=4, map_dim=1, num_images=1, device=dev, output_resolution_height=out,
              output_resolution_width=out, one_shot=not streamed)
    U.sample_from_gen_PatchByPatch_test(G, **dict(kw, output_resolution_height=384, output_resolution_width=384))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = U.sample_from_gen_PatchByPatch_test(G, **kw)      # cold: includes the allocator's first hipMalloc of the activations
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    img = U.sample_from_gen_PatchByPatch_test(G, **kw)      # steady state (what a server generating image after image sees)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sh, sw, th, tw, p = U.tiling_plan(6, 4, 3, 3, out, out)
    print("%-3s %4dx%-4d %-8s grid %2dx%-2d  %7.3f s (first call %.3f s)  %8.2f Mpix/s  %8.1f patches/s  finite=%s" % (
        norm, out, out, "streamed" if streamed else "one-shot", th, tw, dt, cold, out * out / dt / 1e6,
        (sh * sw * 9 if streamed else th * tw) / dt, bool(torch.isfinite(img).all())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:                       # e.g. `infer_bench.py SSM 4096` under rocprofv3: one configuration only
        run(sys.argv[1], int(sys.argv[2]), False)
        sys.exit(0)
    for norm in ("BN", "SSM"):
        for out in (1024, 4096):
            run(norm, out, False)
        run(norm, 1024, True)
