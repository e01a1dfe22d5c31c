#!/usr/bin/env python3
"""Headline benchmark: G+D train-step throughput, real 192x192x3 crops per second, batch 8 per GPU
(BASELINE.json metric; config 1/2: n_layers_G=6, n_layers_D=4, BN, padding_mode=local, fp32).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = reference train.py:122-171 with disc_iters=1: D(real) fwd+bwd, G fwd of 8 images x 3x3
patches of 128^2, D(fake) fwd+bwd, Adam(D), D(fake) fwd, bwd through D and G, Adam(G).  Inputs
(real crops, latents) are synthetic and resident in HBM before the timed region.  With N > 1 every
rank runs batch 8 (weak scaling) and each model's flat gradient is all-reduced once per step over RCCL;
BatchNorm uses per-rank statistics like the reference's nn.DataParallel replicas (ITG_SYNC_BN=1 all-reduces
the statistics instead, which gives the single-process semantics at batch 8N).  `python bench.py --gpus N`
without a launcher starts the N ranks itself.

Prints ONE JSON line on rank 0; see DESIGN.md section "Measurement" for the roofline numerator
(855.5 GF of necessary conv MACs x2 per step of batch 8) and the cpu_baseline definition.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# hipGraph replay (config 3, or ITG_GRAPH=1): two graph queues overlap the captured branches best on this runtime
# (measured 1 / 2 / 3 / 4 queues: 679 / 763 / 722 / 749 crops/s on config 1).  Must be set before HIP initialises.
if "config3" in sys.argv or os.environ.get("ITG_GRAPH") == "1":
    os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")
# (config 3 is launch- and latency-bound - 300 launches of ~9 us: the deferred weight-gradient reduce, one launch per backward pass
# instead of 2-5 per layer, is worth +4 % there; config 1 is bound by kernel work and loses 2 %.  The decision is train.py's
# launch_plan - the CLI and the bench run the same schedule - and is recorded in the line's config block.)

import torch  # noqa: E402

NECESSARY_GF_PER_STEP = 855.5      # SURVEY.md section 8d, config 1, batch 8 / 8 images
NECESSARY_GF_CONFIG3 = 340.9       # nl_G 5 + attention, fake 192^2, real 128^2, batch 8
BF16_MFMA_PEAK_TF = 2500.0         # MI355X_MICROARCH.md dense bf16 peak
FLAGS3 = ["--n_layers_G", "5", "--n_layers_D", "4", "--type_norm", "BN", "--padding_mode", "local", "--attention",
          "--outer_padding", "replicate", "--num_images", "8", "--batch_size", "8", "--leak_G", "0.02",
          "--spec_norm_D", "--smooth", "--random_crop", "128", "--seed", "1234", "--bf16"]
NECESSARY_GF_CONFIG4 = 1463.2      # 4x4 patch grid, fake 512^2, batch 8 in total
FP32_MFMA_PEAK_TF = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CU @ 2.4 GHz
FLAGS = ["--n_layers_G", "6", "--n_layers_D", "4", "--type_norm", "BN", "--padding_mode", "local",
         "--outer_padding", "replicate", "--num_images", "8", "--batch_size", "8", "--leak_G", "0.02",
         "--spec_norm_D", "--smooth", "--random_crop", "192", "--seed", "1234"]


def gpu_leg(a):
    import torch.distributed as dist
    from infinite_texture_gans_amd import ops, utils as U
    from infinite_texture_gans_amd.engine import Trainer, BandTrainer
    from infinite_texture_gans_amd.dist import BandComm

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("ITG_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1 or os.environ.get("ITG_FORCE_COLLECTIVES", "0") == "1":      # one-rank RCCL rehearsal of the collectives
        # "nccl" is RCCL over xGMI.  ITG_DIST_BACKEND=gloo (+ ITG_FORCE_DEVICE=0) lets the multi-rank code path
        # be rehearsed with several ranks on a single GPU.
        backend = os.environ.get("ITG_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD
        if dist.get_world_size() != world:
            raise SystemExit("process group has %d ranks, WORLD_SIZE says %d" % (dist.get_world_size(), world))
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else ""
        except Exception:      # noqa: BLE001
            ver = "?"
        _RANKS.update(ranks=dist.get_world_size(), backend=("nccl (RCCL %s)" % ver) if backend == "nccl" else backend)
        if os.environ.get("ITG_RT_INIT_ONLY") == "1":       # rehearsal: RCCL present (its streams exist), no collective in the step
            group = None
    band = a.workload == "config4"
    cfg3 = a.workload == "config3"
    _WORKLOAD[0] = a.workload
    args = U.prepare_parser().parse_args(
        FLAGS3 if cfg3 else FLAGS + (["--num_patches_height", "4", "--num_patches_width", "4"] if band else []))
    if args.bf16:
        ops.mfma_precision("bf16").set()
    crop = args.random_crop
    args.beta1 = float(args.beta1)
    torch.manual_seed(args.seed)               # identical initial weights on every rank
    netG, netD = U.prepare_models(args, dev)
    netG.train(), netD.train()
    n_in = a.steps + a.warmup + 2
    if band:
        # config 4: ONE batch of 8 fake 512^2 images whose 4x4 patch grid is sharded by rows over the ranks
        # (strong scaling); latents are the same full tensors on every rank, real crops are sharded.
        tr = BandTrainer(netG, netD, args, dev, BandComm(rank, world, group))
        g = torch.Generator().manual_seed(args.seed + 1)
        k = args.batch_size // world
        reals = [(torch.rand(args.batch_size, 3, 192, 192, generator=g) * 2 - 1)[rank * k:(rank + 1) * k].to(dev)
                 for _ in range(2)]
        zs = [torch.randn(args.num_images, args.z_dim, 18, 18, generator=g).to(dev) for _ in range(n_in)]
    else:
        from infinite_texture_gans_amd.train import launch_plan
        args.launch_mode = "graph" if graph_mode() else "eager"
        _, defer = launch_plan(args, world, False)
        _LAUNCH.update(defer_reduce=bool(defer))
        tr = Trainer(netG, netD, args, dev, dist_group=group, defer_reduce=defer)
        g = torch.Generator().manual_seed(args.seed + 1 + rank)
        reals = [(torch.rand(args.batch_size, 3, crop, crop, generator=g) * 2 - 1).to(dev) for _ in range(2)]
        zs = [torch.randn(args.num_images, args.z_dim, 14, 14, generator=g).to(dev) for _ in range(n_in)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # One iteration is ~380 kernel launches (5-7 ms of host time against 11 ms on the GPU).  Default: eager
    # launches - with the step's three HIP streams the eager queue overlaps better than a replayed hipGraph
    # (measured 11.07 vs 11.47 ms); ITG_GRAPH=1 records the iteration once (engine.Trainer.capture) and replays it.
    use_graph = graph_mode()
    nec_gf = NECESSARY_GF_CONFIG4 / world if band else (NECESSARY_GF_CONFIG3 if cfg3 else NECESSARY_GF_PER_STEP)
    peak_tf = BF16_MFMA_PEAK_TF if args.bf16 else FP32_MFMA_PEAK_TF
    if use_graph:
        tr.capture(reals[0], zs[0], warmup=max(1, a.warmup))
        step = tr.step_graphed
    else:
        for i in range(a.warmup):
            tr.step(reals[i % 2], zs[i])
        step = tr.step
    # (handing the step the NEXT real batch so that its D(real) pass runs beside this iteration's generator backward -
    # engine.Trainer.step(next_real=...) - measured neutral to -2 %: the bench does not use it)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        losses = step(reals[i % 2], zs[a.warmup + i])
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    losses = [float(v) for v in losses]

    # ---- the same workload with the reference's DIRECT algorithm for every convolution the headline runs through Winograd
    # (same process, same box, same streams): the number the Winograd path has to be read against
    direct = None
    if rank == 0 and world == 1 and a.workload == "config1" and ops.WINOGRAD and not args.bf16 and not a.no_direct:
        direct = direct_leg(a, args, dev, reals, zs, use_graph)

    # ---- per-kernel launch durations (HIP events on the launching stream), one extra step
    # (every rank runs this iteration: it contains the sync-BN / gradient collectives)
    roof = None
    if rank == 0:
        ops.PROFILE = []
    if hasattr(tr, "set_overlap"):
        tr.set_overlap(False)      # kernels timed one at a time: concurrent streams would stretch each other's events
    from infinite_texture_gans_amd import _lib as _itg_lib
    _itg_lib.BYTE_LOG = {"bytes": 0, "calls": 0}      # algorithmic HBM bytes of this iteration (every tensor a launch is handed, once)
    try:
        tr.step(reals[0], zs[-1])
    finally:
        byte_log, _itg_lib.BYTE_LOG = _itg_lib.BYTE_LOG, None
    torch.cuda.synchronize()
    if rank == 0:
        agg = {}
        for tag, launches, flops, e0, e1, nbytes in ops.PROFILE:
            d = agg.setdefault(tag, [0, 0.0, 0.0, 0.0])
            d[0] += launches
            d[1] += flops
            d[2] += e0.elapsed_time(e1) * 1e-3
            d[3] += nbytes
        ops.PROFILE = None
        tag, (nl, fl, sec, nby) = max(agg.items(), key=lambda kv: kv[1][2])
        ach = fl / sec / 1e12
        traffic, traffic_note = hbm_traffic(tag)
        tot_sec = sum(x[2] for x in agg.values())
        tot_fl = sum(x[1] for x in agg.values())
        # the time-weighted figure first: ALL conv calls of the step together - the "dominant kernel" below is
        # ~13 % of the conv time
        roof = {"conv_stack": {"time_ms": round(tot_sec * 1e3, 3), "tflops": round(tot_fl / tot_sec / 1e12, 2),
                               "frac_of_peak": round(tot_fl / tot_sec / 1e12 / peak_tf, 4),
                               "what": "executed conv flops of one iteration / sum of the HIP-event spans of every conv call "
                                       "(un-overlapped), against the same MFMA peak"},
                "bound": "mfma", "kernel": tag, "achieved": round(ach, 2), "peak": peak_tf,
                "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4), "traffic": traffic,
                "kernel_time_share": round(sec / tot_sec, 3),
                "algorithmic_bytes_per_launch": int(nby / nl),
                "traffic_over_algorithmic": None if traffic is None else round(traffic / (nby / nl), 2),
                "launches_per_step": nl, "avg_launch_us": round(sec / nl * 1e6, 1),
                "flops_per_launch": round(fl / nl / 1e9, 3),
                "timing": "HIP events around each conv call of one un-overlapped iteration (a split-K call includes its "
                          "second-stage launch, a weight-gradient call its slab reduce)",
                "conv_time_share": {k: round(v[2] / tot_sec, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:8]},
                # every conv kernel instantiation with >= 1.5 % of the conv time: its own rate against the same peak, so
                # that the below-roofline tail (weight gradients, the generator's narrow layers) is in the line itself
                "kernels": [{"kernel": k, "launches": v[0], "avg_us": round(v[2] / v[0] * 1e6, 1),
                             "tflops": round(v[1] / v[2] / 1e12, 1), "frac": round(v[1] / v[2] / 1e12 / peak_tf, 3),
                             "time_share": round(v[2] / tot_sec, 3)}
                            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2]) if v[2] / tot_sec >= 0.015],
                # necessary = the reference algorithm's conv flops (every 3x3 conv at its input's resolution);
                # executed = what this build's conv launches contract in one step (the x2 upsample in front of a block's
                # first conv is folded into its filter: 4 of 9 taps' worth of multiply-adds) - the utilisation figure uses it
                "step_necessary_gflop": round(nec_gf, 1),
                "step_executed_gflop": round(sum(x[1] for x in agg.values()) / 1e9, 1),
                "step_frac_of_mfma_peak": round(sum(x[1] for x in agg.values()) / (dt / a.steps) / 1e12 / peak_tf, 4),
                # the same step priced at the REFERENCE algorithm's flop count (direct convolutions, materialised upsample): what a
                # direct implementation would have to sustain to match this step time - an equivalence figure, not a utilisation
                "step_reference_equivalent_tflops": round(nec_gf / (dt / a.steps) / 1e3, 1),
                # the whole iteration against BOTH rooflines, and which one (if any) binds it: algorithmic HBM
                # bytes = every tensor a launch is handed, counted once per launch (operands read once, results written once;
                # Winograd / split-K / slab workspaces are the implementation's traffic, not counted), flat buffers by element count
                "step_hbm": _step_hbm(byte_log, dt / a.steps, sum(x[1] for x in agg.values()) / (dt / a.steps) / 1e12 / peak_tf),
                "algorithms": _algorithms_note(),
                "direct_algorithm": direct,
                "traffic_source": traffic_note,
                "streams": dict(ops.STREAM_PLACEMENT),
                "membound": None if a.no_membound else membound_leg(dev)}
    if world > 1:
        dist.barrier()
    _PAR[0] = exchange_desc(tr)
    return rank, world, dt, args, losses, roof


HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def _step_hbm(byte_log, step_s, mfma_frac):
    """Algorithmic HBM bytes of one iteration / step time against 8 TB/s, beside the MFMA fraction of the same iteration, and the
    bound they name: `mfma` / `hbm` when the larger fraction is at least 0.25, else `launch` - the step is a chain of short
    dependent launches (average below) and neither the matrix pipe nor the memory system is what it waits for."""
    b, calls = byte_log["bytes"], byte_log["calls"]
    gbps = b / step_s / 1e9
    fh = gbps / HBM_PEAK_GBPS
    bound = "launch" if max(fh, mfma_frac) < 0.25 else ("hbm" if fh > mfma_frac else "mfma")
    return {"algorithmic_bytes_per_step": int(b), "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(fh, 4), "mfma_frac_same_step": round(mfma_frac, 4), "bound": bound,
            "abi_calls_per_step": calls, "avg_us_per_call": round(step_s / max(calls, 1) * 1e6, 2)}


def direct_leg(a, args, dev, reals, zs, use_graph):
    """crops/s of the same step with ops.WINOGRAD off (a second engine on freshly initialised models: the panels a layer
    packs are decided at construction), timed exactly like the headline."""
    from infinite_texture_gans_amd import ops, utils as U
    from infinite_texture_gans_amd.engine import Trainer
    keep = ops.WINOGRAD
    ops.WINOGRAD = False
    try:
        torch.manual_seed(args.seed)
        netG, netD = U.prepare_models(args, dev)
        netG.train(), netD.train()
        tr = Trainer(netG, netD, args, dev)
        if use_graph:
            tr.capture(reals[0], zs[0], warmup=max(1, a.warmup))
            step = tr.step_graphed
        else:
            for i in range(a.warmup):
                tr.step(reals[i % 2], zs[i])
            step = tr.step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(reals[i % 2], zs[a.warmup + i])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"crops_per_s": round(args.batch_size * a.steps / dt, 3), "ms_per_step": round(dt / a.steps * 1e3, 3),
                "steps": a.steps, "note": "every convolution direct except the folded upsample (ITG_WINOGRAD=0), same process and streams"}
    finally:
        ops.WINOGRAD = keep


_LAUNCH = {"defer_reduce": False}      # what bench.py's trainer was built with (recorded in the line's config block)
_PAR = [""]
_RANKS = {"ranks": 1, "backend": "none"}      # what the process group reports once initialised (not the environment)


def exchange_desc(tr):
    """How this trainer exchanges gradients, from the objects it actually built (not from the environment)."""
    if getattr(tr, "sync", None) is None or tr.world == 1 and not tr._exchange:
        return "no gradient exchange (one rank)"
    if tr._exchange:
        parts = []
        for name, flat in (("D", tr.flatD), ("G", tr.flatG)):
            ex = tr._exchange.get(id(flat))
            parts.append("%s: %s" % (name, "two buckets, split at %d of %d floats" % (ex.split, flat.numel) if ex is not None and ex.split
                                     else "single all-reduce"))
        return "flat gradient exchange per model (" + "; ".join(parts) + ")"
    return "flat gradient exchange, single all-reduce per model"


GRAPH_DEFAULT = {"config1": "0", "config3": "1", "config4": "0", "config5": "0"}   # config 3's 5.5 ms step is shorter than its host time
_WORKLOAD = ["config1"]


def _algorithms_note():
    """Which convolutions of the step do NOT run the reference's direct algorithm (switch state of this process)."""
    from infinite_texture_gans_amd import ops
    from infinite_texture_gans_amd.models.layers import up2_fold_enabled
    parts = []
    if ops.WINOGRAD and ops.MFMA_PRECISION == ops.PREC_F32:
        parts.append("Winograd F(4x4,4x4) for the discriminator's 256->512 layer: forward, input gradient%s (49 of 256 multiplications)"
                     % (", weight gradient" if ops.WINOGRAD_WGRAD else ""))
        if ops.WINOGRAD_S2:
            parts.append("Winograd F(4x4,2x2) over the four parity classes of the discriminator's stride-2 64->128 and 128->256 layers "
                         "(25 of 64 multiplications): forward from %d tiles per pass%s%s"
                         % (ops.WINO_S2_MIN_TILES,
                            ", weight gradient from %d input channels" % ops.WINO_S2_WGRAD_MIN_CI if ops.WINO_S2_WGRAD and ops.WINOGRAD_WGRAD else "",
                            ", input gradient (adjoint pipeline) from %d input channels" % ops.WINO_S2_DGRAD_MIN_CI if ops.WINO_S2_DGRAD else ""))
        if ops.WINOGRAD_G:
            parts.append("Winograd F(4x4,3x3) for the generator's 416- and 208-channel 3x3 layers (36 of 144)")
    if up2_fold_enabled():
        parts.append("nearest-x2 upsample folded into the generator blocks' first conv (4 of 9 taps' multiply-adds)")
    return "; ".join(parts + ["every other convolution direct"]) if parts else "direct convolutions throughout"


def _env_defaults():
    """Every ITG_* / HIP switch this process ran under that bench.py itself injected or that differs from unset (ADVICE r3:
    a committed number must say what configuration it measured)."""
    keys = sorted(k for k in os.environ if k.startswith("ITG_") or k in ("DEBUG_HIP_FORCE_GRAPH_QUEUES", "GPU_MAX_HW_QUEUES"))
    return {k: os.environ[k] for k in keys}


def graph_mode():
    return os.environ.get("ITG_GRAPH", GRAPH_DEFAULT[_WORKLOAD[0]]) == "1" and int(os.environ.get("WORLD_SIZE", 1)) == 1


def kernel_source_hash():
    """sha256 over the kernel sources and the ABI header: identifies the build a profile was taken from (the .so itself is
    rebuilt by the driver and need not be byte-identical)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "infinite_texture_gans_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(ROOT, "include", "itg.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def schedule_source_hash():
    """sha256 over the host files that decide WHICH kernels a step launches, in what form and order (ops.py, engine.py,
    models/layers.py, dist.py): a profile taken before a schedule change describes another step even when the kernel sources are
    unchanged (the r04 set trailed two such commits)."""
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "infinite_texture_gans_amd")
    for f in ("ops.py", "engine.py", os.path.join("models", "layers.py"), "dist.py"):
        h.update(f.encode())
        h.update(open(os.path.join(pkg, f), "rb").read())
    return h.hexdigest()[:16]


def hbm_traffic(kernel):
    """(HBM bytes per launch of ``kernel``, provenance note) from the committed PMC summary (separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied by tools/prof_summary.py).  The summary carries the hash
    of the kernel sources it was measured on: when that is not the build that is running, traffic is None."""
    import glob
    wl = _WORKLOAD[0]
    pat = "r[0-9][0-9]_hbm_traffic.json" if wl == "config1" else "r[0-9][0-9]_%s_hbm_traffic.json" % wl
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not files:
        return None, "no committed PMC summary under profiles/"
    name = os.path.basename(files[-1])
    try:
        js = json.load(open(files[-1]))
        meta = js.get("_meta", {})
        here, sched = kernel_source_hash(), schedule_source_hash()
        if meta.get("kernel_source_sha16") != here:
            return None, ("profiles/%s was measured on kernel sources %s (git %s), this build is %s: traffic withheld; "
                          "re-run tools/profile_all.sh" % (name, meta.get("kernel_source_sha16", "unrecorded"),
                                                           meta.get("git_head", "?"), here))
        if meta.get("schedule_source_sha16") != sched:
            return None, ("profiles/%s was measured under the launch schedule %s (git %s), this tree's ops / engine / layers / dist "
                          "hash to %s: traffic withheld; re-run tools/profile_all.sh" % (
                              name, meta.get("schedule_source_sha16", "unrecorded"), meta.get("git_head", "?"), sched))
        t = js.get(kernel)
        if t is None:
            return None, "profiles/%s has no entry for this kernel" % name
        return int(t["fetch_bytes"] + t["write_bytes"]), (
            "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of kernel sources %s and launch schedule %s = this build "
            "(git %s); counters cannot be read from inside the timed process" % (name, here, sched, meta.get("git_head", "?")))
    except Exception as e:      # noqa: BLE001
        return None, "profiles/%s unreadable: %s" % (name, e)


def host_cores():
    """Cores this process may actually use: cgroup quota, else affinity mask, capped at the 16-core
    share a one-GPU box grants (oversubscribing the host's 256 hardware threads is 10x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("ITG_CPU_THREADS", 16))))


def cpu_leg(workload="config1"):
    """The oracle (CPU restatement of the reference path) timed on the host cores: full train steps at the same
    config with the vectorised LocalPadder ('port'), plus - config 1 - ONE step with the reference-faithful Python-loop
    LocalPadder / merge / crop ('port-loops': what the reference's own CPU path spends ~60 % of its time in,
    SURVEY.md F6 / BASELINE.md section 4)."""
    from oracle import step as ostep
    from oracle.nets import GCfg, DCfg
    from infinite_texture_gans_amd import utils as U
    ncores = host_cores()
    torch.set_num_threads(ncores)
    cfg3, band = workload == "config3", workload == "config4"
    flags = [f for f in FLAGS3 if f != "--bf16"] if cfg3 else FLAGS + (["--num_patches_height", "4", "--num_patches_width", "4"] if band else [])
    args = U.prepare_parser().parse_args(flags)
    torch.manual_seed(1234)
    netG, netD = U.prepare_models(args, "cpu")          # parameter containers only (no forward on CPU)
    gsd = ostep.as_leaf_params({k: v.clone() for k, v in netG.state_dict().items()})
    dsd = ostep.as_leaf_params({k: v.clone() for k, v in netD.state_dict().items()})
    gcfg = GCfg(z_dim=128, G_ch=52, base_res=4, n_layers_G=args.n_layers_G, attention=args.attention, leak=0.02, type_norm="BN",
                num_patches_h=args.num_patches_height, num_patches_w=args.num_patches_width)
    dcfg = DCfg(img_ch=3, base_ch=64, n_layers_D=4, SN=True)
    optD = ostep.Adam([dsd[k] for k in ostep.trainable(dsd)])
    optG = ostep.Adam([gsd[k] for k in ostep.trainable(gsd)])
    g = torch.Generator().manual_seed(7)
    crop = args.random_crop
    zs = args.num_patches_height * 4 + 2
    real = torch.rand(8, 3, crop, crop, generator=g) * 2 - 1
    z = torch.randn(8, 128, zs, zs, generator=g)
    ostep.train_step(gsd, dsd, gcfg, dcfg, optG, optD, real, z, None, smooth=True)      # warm-up (thread pools, oneDNN)
    nsteps, t0 = 0, time.perf_counter()
    while nsteps < 3 or (time.perf_counter() - t0 < 10.0 and nsteps < 8):
        ostep.train_step(gsd, dsd, gcfg, dcfg, optG, optD, real, z, None, smooth=True)
        nsteps += 1
    dt = (time.perf_counter() - t0) / nsteps
    out = {"value": round(8.0 / dt, 4), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port",
           "cpu_model": cpu_model(),
           "sample": "%d full G+D train steps of %s after 1 warm-up, batch 8 / 8 images (%d G-patches), vectorised "
                     "LocalPadder, torch-CPU fp32 (%.2f s/step)" % (nsteps, workload, 8 * args.num_patches_height ** 2, dt)}
    if workload == "config1":
        t0 = time.perf_counter()
        ostep.train_step(gsd, dsd, gcfg, dcfg, optG, optD, real, z, None, smooth=True, loops=True)
        dl = time.perf_counter() - t0
        out["loops"] = {"value": round(8.0 / dl, 4), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port-loops",
                        "sample": "1 full G+D train step with the reference-faithful per-patch torch.cat / slice loops of "
                                  "LocalPadder, merge_patches_into_image and crop_images (%.2f s/step)" % dl}
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def membound_leg(dev):
    """Achieved HBM GB/s (algorithmic bytes / HIP-event time, SURVEY.md section 8d's byte counts) of the memory-bound
    operators of the generator at config-1 tensor sizes, standalone launches: LocalPadder (NHWC operator), BatchNorm
    train forward (+LeakyReLU) and backward, LeakyReLU, nearest x2 upsample.  -> list for roofline["membound"]."""
    from infinite_texture_gans_amd import ops
    rows = []

    def timeit(fn, iters=10):
        """GPU time per call: the calls are recorded into a hipGraph and the replay is timed with HIP events, so that the host's
        launch cost (30 us for a BatchNorm forward, 85 us for its autograd backward - more than the kernels take) stays out."""
        fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    def row(name, nbytes, t, t_cold=None):
        r = {"op": name, "mbytes": round(nbytes / 1e6, 1), "us": round(t * 1e6, 1), "gbps": round(nbytes / t / 1e9, 1),
             "frac_of_8tbps": round(nbytes / t / 8e12, 3)}
        if t_cold is not None:
            # the same launches rotating over enough tensor sets that the replay's working set exceeds 600 MB - past the 256 MB
            # Infinity Cache, so every byte comes from / goes to HBM (the figure above re-reads a 75-377 MB
            # set that the cache partly holds)
            r.update(us_hbm=round(t_cold * 1e6, 1), gbps_hbm=round(nbytes / t_cold / 1e9, 1), frac_of_8tbps_hbm=round(nbytes / t_cold / 8e12, 3))
        rows.append(r)

    def timeit_sets(make, nbytes_set, iters=10):
        """timeit over K = ceil(600 MB / set) (+1) independent tensor sets, call i on set i % K."""
        k = max(2, int(600e6 // max(nbytes_set, 1)) + 2)
        fns = [make() for _ in range(min(k, iters))]
        calls = iter(range(1 << 30))
        return timeit(lambda: fns[next(calls) % len(fns)](), iters=iters)

    for c, p in ((26, 128), (104, 32)):
        g = ops.GT(torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev), c)
        nb = 4 * 72 * g.ld * (p * p + (p + 2) ** 2)

        def mk_pad(c=c, p=p):
            gi = ops.GT(torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev), c)
            return lambda: ops.local_pad_grid(gi, ops.PAD_REPLICATE)
        row("LocalPadder halo gather C=%d P=%d" % (c, p), nb, timeit(lambda: ops.local_pad_grid(g, ops.PAD_REPLICATE)), timeit_sets(mk_pad, nb))
    c, p = 13, 128
    xg = ops.GT(torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev), c)
    numel = xg.t.numel()
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev)

    def mk_bn_f():
        xi = ops.GT(torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev), c)
        return lambda: ops.bn_act(xi, gamma, beta, rm, rv, nbt, True, 1e-5, 0.1, ops.ACT_LRELU, 0.02, False)

    def mk_bn_fb():
        xi = torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev).requires_grad_(True)
        gi, bi = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        di = torch.randn_like(xi)
        return lambda: torch.autograd.grad(ops.bn_act(ops.GT(xi, c), gi, bi, rm, rv, nbt, True, 1e-5, 0.1, ops.ACT_LRELU, 0.02, False).t,
                                           (xi, gi, bi), di)
    # backward: forward + backward recorded together (a backward alone would run on an autograd graph built outside the capture,
    # whose stale default-stream AccumulateGrad nodes break hipStreamEndCapture - DESIGN section 3), the forward's time subtracted
    t_f = timeit(mk_bn_f())
    t_f_cold = timeit_sets(mk_bn_f, 4 * numel * 2)
    row("BatchNorm train fwd + LeakyReLU C=13 P=128", 4 * numel * 3, t_f, t_f_cold)
    t_fb = timeit(mk_bn_fb())
    t_fb_cold = timeit_sets(mk_bn_fb, 4 * numel * 4)
    row("BatchNorm train bwd C=13 P=128", 4 * numel * 5, t_fb - t_f, t_fb_cold - t_f_cold)

    def mk_act():
        xi = ops.GT(torch.randn(8, 3, 3, p, p, ops.ld_for(c), device=dev), c)
        return lambda: ops.act(xi, ops.ACT_LRELU, 0.2)
    row("LeakyReLU C=13 P=128", 8 * numel, timeit(mk_act()), timeit_sets(mk_act, 8 * numel))

    def mk_ups():
        xi = ops.GT(torch.randn(8, 3, 3, 64, 64, 28, device=dev), 26)
        return lambda: ops.upsample2x(xi)
    nu = 4 * 8 * 9 * 64 * 64 * 28 * 5
    row("nearest x2 upsample C=26 P=64", nu, timeit(mk_ups()), timeit_sets(mk_ups, nu))
    # ... and the normalisation kernels' HBM-side rates INSIDE the train step, from the committed rocprofv3 passes of this build:
    # (FETCH_SIZE + WRITE_SIZE per launch) / the kernel's average duration in the un-overlapped step
    rows.extend(_rocprof_membound())
    return rows


def _rocprof_membound():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))
    if not files:
        return []
    try:
        js = json.load(open(files[-1]))
        meta = js.get("_meta", {})
        if meta.get("kernel_source_sha16") != kernel_source_hash():
            return [{"op": "rocprof rows withheld", "note": "profiles/%s was measured on other kernel sources" % os.path.basename(files[-1])}]
        out = []
        for k, v in js.items():
            if k.startswith(("bn_", "local_pad", "act_", "upsample_")) and "hbm_gbps" in v and v.get("fetch_bytes", 0) + v.get("write_bytes", 0) > 20e6:
                out.append({"op": "rocprof: " + k, "mbytes": round((v["fetch_bytes"] + v["write_bytes"]) / 1e6, 1), "us": v["avg_us"],
                            "gbps_hbm": v["hbm_gbps"], "frac_of_8tbps_hbm": round(v["hbm_gbps"] / 8000.0, 3), "launches": v["launches"],
                            "source": "profiles/%s: HBM counters / kernel time, averaged over the launches of one train step" % os.path.basename(files[-1])})
        return out
    except Exception as e:      # noqa: BLE001
        return [{"op": "rocprof rows unreadable", "note": str(e)}]


def relaunch(n):
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run as N
    fresh child processes (one rank per GPU over RCCL) and pass their output through.  This parent has not touched
    the GPU (torch.cuda.device_count() does not initialise HIP on this image) and never execs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("ITG_FORCE_DEVICE") is None:      # ITG_FORCE_DEVICE: several (gloo) ranks on one GPU, rehearsal only
        print("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


SSM_GF_PER_PATCH = 10.66            # SSM generator forward, one 128^2 patch (n_layers_G 6, G_ch 52)
FLAGS5 = ["--n_layers_G", "6", "--type_norm", "SSM", "--padding_mode", "local", "--outer_padding", "replicate", "--leak_G", "0.02",
          "--seed", "1234"]


def infer_leg(a):
    """BASELINE config 5: inference tiling (reference test_sample.py -> utils.py:258-397) of one ``--out``^2 image per step as
    ONE forward over the T x T patch grid, the patch rows sharded over the ranks (dist.RowHalo: one pixel row
    per conv and neighbour over RCCL).  The band's latents are resident in HBM before the timed region and the image stays on
    the device (the PCIe-inclusive figure is in DESIGN.md)."""
    import torch.distributed as dist
    from infinite_texture_gans_amd import ops, utils as U
    from infinite_texture_gans_amd.dist import RowHalo
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("ITG_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1:
        backend = os.environ.get("ITG_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else ""
        except Exception:      # noqa: BLE001
            ver = "?"
        _RANKS.update(ranks=dist.get_world_size(), backend=("nccl (RCCL %s)" % ver) if backend == "nccl" else backend)
    _WORKLOAD[0] = "config5"
    args = U.prepare_parser().parse_args(FLAGS5)
    torch.manual_seed(args.seed)               # identical weights on every rank
    netG, _ = U.prepare_models(args, dev)
    netG.eval()
    out = a.out
    sh, sw, t_h, t_w, p = U.tiling_plan(args.n_layers_G, args.base_res, 3, 3, out, out)
    halo = RowHalo(rank, world, group)
    zf = mf = None
    if a.reference_rng:
        g = torch.Generator().manual_seed(args.seed + 1)
        zf = torch.randn(1, args.z_dim, t_h * args.base_res + 2, t_w * args.base_res + 2, generator=g)
        mf = [torch.randn(1, args.map_dim, t_h * (2 ** i) * args.base_res + 4, t_w * (2 ** i) * args.base_res + 4, generator=g)
              for i in range(args.n_layers_G)]
    lat = [U.band_latents(netG, t_h, t_w, args.base_res, halo, dev, args.z_dim, args.map_dim, seed=args.seed + 1 + i,
                          z_full=zf, maps_full=mf) for i in range(2)]

    def step(i):
        z, m = lat[i % 2]
        return U.generate_band(netG, z, m, t_w, args.base_res, halo)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        img = step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        img = step(i)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    finite = bool(torch.isfinite(img).all())
    rows_mine = halo.band(t_h)
    # per-kernel HIP events of one more forward (rank 0): the dominant conv instantiation against the fp32 MFMA peak
    roof = None
    if rank == 0:
        ops.PROFILE = []
    step(0)
    torch.cuda.synchronize()
    if rank == 0:
        agg = {}
        for tag, launches, flops, e0, e1, nbytes in ops.PROFILE:
            d = agg.setdefault(tag, [0, 0.0, 0.0, 0.0])
            d[0] += launches; d[1] += flops; d[2] += e0.elapsed_time(e1) * 1e-3; d[3] += nbytes
        ops.PROFILE = None
        tag, (nl, fl, sec, nby) = max(agg.items(), key=lambda kv: kv[1][2])
        tot_sec, tot_fl = sum(x[2] for x in agg.values()), sum(x[1] for x in agg.values())
        traffic, note = hbm_traffic(tag)
        roof = {"bound": "mfma", "kernel": tag, "achieved": round(fl / sec / 1e12, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(fl / sec / 1e12 / FP32_MFMA_PEAK_TF, 4), "traffic": traffic, "traffic_source": note,
                "launches_per_step": nl, "avg_launch_us": round(sec / nl * 1e6, 1),
                "conv_stack": {"time_ms": round(tot_sec * 1e3, 3), "tflops": round(tot_fl / tot_sec / 1e12, 2),
                               "frac_of_peak": round(tot_fl / tot_sec / 1e12 / FP32_MFMA_PEAK_TF, 4),
                               "gflop_this_rank": round(tot_fl / 1e9, 1)},
                "grid_gflop": round(SSM_GF_PER_PATCH * t_h * t_w, 1),
                "step_frac_of_mfma_peak": round(SSM_GF_PER_PATCH * t_h * t_w * 1e9 / (dt / a.steps) / 1e12 / (FP32_MFMA_PEAK_TF * world), 4),
                "kernels": [{"kernel": k, "launches": v[0], "avg_us": round(v[2] / v[0] * 1e6, 1), "tflops": round(v[1] / v[2] / 1e12, 1),
                             "frac": round(v[1] / v[2] / 1e12 / FP32_MFMA_PEAK_TF, 3), "time_share": round(v[2] / tot_sec, 3)}
                            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2]) if v[2] / tot_sec >= 0.015]}
    if world > 1:
        dist.barrier()
    if rank != 0:
        return
    ms = dt / a.steps * 1e3
    line = {"metric": "inference tiling: output megapixels/sec of one-shot patch-grid generation (SSM generator, %dx%d image)" % (out, out),
            "value": round(out * out * a.steps / dt / 1e6, 3), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config 5: type_norm=SSM, n_layers_G=6, G_ch=52, random-init weights, eval-mode norm; one %dx%d image "
                                   "per step = ONE forward over the %dx%d patch grid of 128^2 patches (test_sample.py path, one-shot "
                                   "instead of %d streamed 3x3 sub-images)" % (out, out, t_h, t_w, sh * sw),
                       "patches_per_sec": round(t_h * t_w * a.steps / dt, 1),
                       "parallelism": "patch rows over %d rank(s) (rank 0: rows %d-%d of %d), one halo pixel row per 3x3 conv and "
                                      "neighbour%s" % (world, rows_mine[0], rows_mine[1] - 1, t_h, "" if world == 1 else " over RCCL send/recv"),
                       "latents": "reference full-grid CPU draw, cut per band" if a.reference_rng else "each rank draws its band's rows on the device (per-patch-row seeds)",
                       "finite": finite, "launch": "eager"},
            "roofline": roof}
    line["config"].update(_RANKS)
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_infer_leg()
    print(json.dumps(line), flush=True)


def cpu_infer_leg():
    """The oracle's one-shot generation (CPU restatement of reference utils.py:258-397 + models/generators.py:86-124) of a
    384 x 384 image (3 x 3 patch grid, same SSM generator) on the host cores: a bounded sample of the same workload."""
    from oracle import step as ostep
    from oracle.nets import GCfg
    from infinite_texture_gans_amd import utils as U
    ncores = host_cores()
    torch.set_num_threads(ncores)
    args = U.prepare_parser().parse_args(FLAGS5)
    torch.manual_seed(1234)
    netG, _ = U.prepare_models(args, "cpu")
    sd = {k: v.clone() for k, v in netG.state_dict().items()}
    cfg = GCfg(z_dim=128, G_ch=52, base_res=4, n_layers_G=6, attention=False, leak=0.02, type_norm="SSM")
    sh, sw, t_h, t_w, p = ostep.grid_size(384, 384, cfg)
    zf, maps = ostep.full_latents(cfg, t_h, t_w, torch.Generator().manual_seed(7))
    ostep.infer_oneshot(sd, cfg, zf, maps, 384, 384)
    n, t0 = 0, time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 20):
        ostep.infer_oneshot(sd, cfg, zf, maps, 384, 384)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(384 * 384 / dt / 1e6, 4), "unit": "Mpix/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(),
            "sample": "%d one-shot generations of a 384x384 image (3x3 patch grid, %d patches, same SSM generator) after 1 warm-up, "
                      "torch-CPU fp32 (%.2f s each)" % (n, t_h * t_w, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 0.7 s of config 1 (20 steps = 0.15 s was a short sample)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-membound", dest="no_membound", action="store_true",
                    help="skip the memory-bound operator probes (roofline.membound): profiler runs, so that the kernel statistics "
                         "hold the train step only")
    ap.add_argument("--no-direct", dest="no_direct", action="store_true",
                    help="config1: skip the second timed leg with the direct algorithm instead of Winograd (roofline.direct_algorithm)")
    ap.add_argument("--reference_rng", action="store_true",
                    help="config5: every rank draws the reference's FULL-grid CPU latents (seed-identical images) instead of only "
                         "the rows of its own band on the device")
    ap.add_argument("--out", type=int, default=4096, help="config5: output height = width in pixels")
    ap.add_argument("--workload", choices=["config1", "config3", "config4", "config5"], default="config1",
                    help="config1 (default, the headline metric): batch 8 per GPU, data parallel.  config3: 128^2 crops, "
                         "n_layers_G=5 + attention, convolutions on bf16-operand MFMA.  config4: 4x4 patch "
                         "grid of ONE batch sharded by patch rows over <= 4 GPUs with halo exchange (strong scaling).  config5: "
                         "inference tiling, ONE 4096^2 image (33 x 33 patch grid, SSM generator) per step, patch rows sharded "
                         "over the ranks with halo-row exchange (strong scaling); value = output megapixels per second")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(a.gpus))          # plain `python bench.py --gpus N`: start N ranks, relay rank 0's line
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world:
        sys.exit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (a.gpus, world))
    if a.workload == "config5":
        return infer_leg(a)
    rank, world, dt, args, losses, roof = gpu_leg(a)
    if rank != 0:
        return
    ms = dt / a.steps * 1e3
    ranks = dict(_RANKS)
    if a.workload == "config3":
        out = {"metric": "G+D train-step real 128x128x3 crops/sec (batch 8 per GPU, bf16 MFMA path)",
               "value": round(args.batch_size * world * a.steps / dt, 3), "unit": "crops/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "config 3: 34.jpg-shaped 128x128 crops, n_layers_G=5 n_layers_D=4, attention, BN, "
                                      "padding_mode=local, 3x3 patch grid of 64^2 (fake 192^2), conv operands bf16 / fp32 "
                                      "accumulate, everything else fp32, batch 8 + 8 generated images per GPU",
                          "global_batch": args.batch_size * world, "g_patches_per_sec": round(72 * world * a.steps / dt, 1),
                          "parallelism": "dp%d (%s BatchNorm statistics, %s)" % (
                          world, "all-reduced" if os.environ.get("ITG_SYNC_BN", "0") == "1" else "per-rank", _PAR[0]), "last_losses": losses,
                          "launch": "hipGraph replay" if graph_mode() else "eager",
                          "wgrad_reduce": "deferred (one launch per backward pass)" if _LAUNCH["defer_reduce"] else "per layer",
                          "env_defaults": _env_defaults(),
                          "cli_equivalent": "train.py --launch_mode auto --wgrad_reduce auto runs this schedule (train.launch_plan)"},
               "roofline": roof}
        out["config"].update(ranks)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_leg("config3")
        print(json.dumps(out), flush=True)
        return
    if a.workload == "config4":
        out = {"metric": "G+D train-step real 192x192x3 crops/sec (batch 8 in total, 4x4 patch grid sharded by rows)",
               "value": round(args.batch_size * a.steps / dt, 3), "unit": "crops/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "config 4: 192x192 crops, n_layers_G=6 n_layers_D=4, BN, padding_mode=local, 4x4 patch "
                                      "grid (128 G-patches, fake 512^2), batch 8 + 8 generated images in total",
                          "global_batch": args.batch_size, "g_patches_per_sec": round(128 * a.steps / dt, 1),
                          "parallelism": "patch rows over %d rank(s): halo-row exchange per conv fwd+bwd, sync-BN, "
                                         "band gather -> image-parallel D, flat grad all-reduce" % world,
                          "last_losses": losses,
                          "launch": "hipGraph replay" if graph_mode() else "eager"},
               "roofline": roof}
        out["config"].update(ranks)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_leg("config4")
        print(json.dumps(out), flush=True)
        return
    out = {"metric": "G+D train-step real 192x192x3 crops/sec (batch 8 per GPU)", "value": round(args.batch_size * world * a.steps / dt, 3),
           "unit": "crops/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "config 1/2: 241.jpg-shaped 192x192 crops, n_layers_G=6 n_layers_D=4, BN, "
                                  "padding_mode=local (replicate), G_ch=52 D_ch=64, 3x3 patch grid of 128^2, "
                                  "spec_norm_D, smooth, batch 8 + 8 generated images per GPU",
                      "global_batch": args.batch_size * world, "g_patches_per_sec": round(72 * world * a.steps / dt, 1),
                      "parallelism": "dp%d (%s BatchNorm statistics, %s)" % (
                          world, "all-reduced" if os.environ.get("ITG_SYNC_BN", "0") == "1" else "per-rank", _PAR[0]), "last_losses": losses,
                      "launch": "hipGraph replay" if graph_mode() else "eager",
                      "wgrad_reduce": "deferred (one launch per backward pass)" if _LAUNCH["defer_reduce"] else "per layer",
                      "env_defaults": _env_defaults()},
           "roofline": roof}
    out["config"].update(ranks)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_leg("config1")
    print(json.dumps(out), flush=True)


def _teardown():
    """Leave the process group cleanly (ProcessGroupNCCL warns about leaked resources otherwise); never raises."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:       # noqa: BLE001 - the JSON line is out; a failing teardown must not change the exit code
        pass


if __name__ == "__main__":
    try:
        main()
    finally:
        _teardown()
