"""Training CLI with the reference's flags and control flow (reference train.py), one process per GPU.

    python train.py --data_path datasets/241.jpg --random_crop 192 --padding_mode local --type_norm BN ...
    python -m torch.distributed.run --nproc-per-node 8 train.py ...        # data parallel over RCCL
"""
import os
import random
import time

import numpy as np
import torch

from . import utils as U
from .engine import Trainer, BandTrainer


def lr_factor(decay_lr, epochs_done):
    """Learning-rate multiplier after ``epochs_done`` scheduler steps: ExponentialLR(gamma=0.99) for 'exp',
    MultiStepLR(milestones=[40, 80, 120], gamma=0.5) for 'step' (reference train.py:60-70, stepped once per epoch
    at :183-185)."""
    if decay_lr == "exp":
        return 0.99 ** epochs_done
    if decay_lr == "step":
        return 0.5 ** sum(epochs_done >= m for m in (40, 80, 120))
    return 1.0


def plot_losses(G_losses, D_losses, path):
    """The loss curve the reference writes after the last epoch (train.py:220-227)."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception as e:      # noqa: BLE001  (matplotlib is optional on a training box)
        print("loss plot skipped:", e)
        return
    fig = plt.figure(figsize=(10, 5))
    plt.title("Generator and Discriminator Loss During Training")
    plt.plot(G_losses, label="G")
    plt.plot(D_losses, label="D")
    plt.xlabel("iterations")
    plt.ylabel("Loss")
    plt.legend()
    fig.savefig(path)
    plt.close(fig)


def argparse_copy(args, **kw):
    import copy
    a = copy.copy(args)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def launch_plan(args, world, band):
    """(use_graph, defer_reduce) of this run from --launch_mode / --wgrad_reduce (the same rule bench.py applies):
    hipGraph replay needs one rank and --disc_iters 1 (one latent set per step); the deferred weight-gradient reduce pays
    where the step is launch-bound - under replay with crops of at most 128 pixels (BASELINE config 3: +4 %), not on
    config 1 (-2 %: its one big reduce lands on the critical path)."""
    mode = getattr(args, "launch_mode", "auto")
    if mode not in ("auto", "eager", "graph"):
        raise ValueError("--launch_mode must be auto, eager or graph")
    can = world == 1 and not band and args.disc_iters == 1
    if mode == "graph" and not can:
        raise ValueError("--launch_mode graph needs a single rank, --disc_iters 1 and no --shard_patch_rows")
    env = os.environ.get("ITG_GRAPH")
    # auto: "probe" - train() times its first eager iterations and switches to replay only if the host is the bottleneck
    # (issue time > 90 % of the synchronised step time).  Measured on MI355X: config 1 issues a step in 5.6 ms against 7.4 ms
    # of GPU time (eager 1 069 vs replay 1 050 crops/s through this CLI), config 3 in ~7 ms against 3.4 ms (replay 2x).
    use_graph = can and (mode == "graph" or (mode == "auto" and env == "1"))
    if can and mode == "auto" and env is None:
        use_graph = "probe"
    wr = getattr(args, "wgrad_reduce", "auto")
    if wr not in ("auto", "layer", "deferred"):
        raise ValueError("--wgrad_reduce must be auto, layer or deferred")
    crop = args.center_crop or args.random_crop or 1 << 30
    # "probe": the engine starts with per-layer reduces (the eager schedule) and train() switches the deferred form on together
    # with the replay if the probe picks it (Trainer.set_defer_reduce)
    defer = wr == "deferred" or (wr == "auto" and use_graph is True and crop <= 128)
    if "ITG_DEFER_REDUCE" in os.environ and wr == "auto":
        defer = os.environ["ITG_DEFER_REDUCE"] == "1"
    return use_graph, defer


def defer_with_replay(args):
    """Does --wgrad_reduce auto pick the deferred reduce once --launch_mode auto has decided for replay?"""
    crop = args.center_crop or args.random_crop or 1 << 30
    return getattr(args, "wgrad_reduce", "auto") == "auto" and "ITG_DEFER_REDUCE" not in os.environ and crop <= 128


def train(args):
    device = U.prepare_device(args)
    seed = U.prepare_seed(args)
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    group = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD
    if getattr(args, "bf16", False):
        from . import ops
        ops.mfma_precision("bf16").set()
    random.seed(seed), torch.manual_seed(seed), np.random.seed(seed)
    args.beta1, args.beta2 = float(args.beta1), float(args.beta2)    # the reference's int default breaks torch>=2
    if rank == 0:
        print(args)
    band = bool(getattr(args, "shard_patch_rows", False)) and world > 1
    if band and args.batch_size % world:
        raise ValueError("--shard_patch_rows: batch_size %d does not split over %d ranks" % (args.batch_size, world))
    if band:
        args_data = argparse_copy(args, batch_size=args.batch_size // world)
    else:
        args_data = args
    data, train_data = U.prepare_data(args_data, device, seed=seed + 17 * rank)
    if rank == 0:
        print('Training samples: ', len(train_data))
    netG, netD = U.prepare_models(args, device)          # same seed -> same initial weights on every rank
    netG_ema = None
    if args.ema:
        netG_ema, _ = U.prepare_models(args, device)
        for p in netG_ema.parameters():
            p.requires_grad = False
    if rank == 0:
        print(netG), print(netD)
        print("# Params. G: ", sum(p.numel() for p in netG.parameters()))
        print("# Params. D: ", sum(p.numel() for p in netD.parameters()))
    netG.train(), netD.train()
    use_graph, defer = launch_plan(args, world, band)
    if band:      # one batch, patch rows of every fake image spread over the ranks (BASELINE config 4)
        from .dist import BandComm
        tr = BandTrainer(netG, netD, args, device, BandComm(rank, world, group), netG_ema=netG_ema)
    else:
        tr = Trainer(netG, netD, args, device, netG_ema=netG_ema, dist_group=group, defer_reduce=defer)
    if rank == 0:
        print("launch: %s, weight-gradient reduce: %s" % ({True: "hipGraph replay", False: "eager", "probe": "auto (decided after 4 iterations)"}[use_graph],
                                                           "deferred" if defer else "per layer"))
    probe = {"n": 0, "issue": 0.0, "start": 0.0}      # auto: host issue time of the first full-batch eager iterations
    filename = U.prepare_filename(args)
    start = time.time()
    G_losses, D_losses = [], []
    torch.manual_seed(seed if band else seed + 1000 * rank)   # latents: per-rank CPU RNG stream (shared when sharding rows)
    local = args.padding_mode == 'local'

    def sample():
        if local:
            return U.sample_latents_train(netG, args.z_dim, args.base_res, args.map_dim, args.num_images,
                                          args.num_patches_height, args.num_patches_width, device, merged_maps=band)
        return U.sample_latents_zeros(netG, args.z_dim, args.base_res, args.map_dim, args.num_images, device)

    recorded_lr = None          # (lr_D, lr_G) the recorded graph was captured with: the learning rates are kernel arguments
    fatal_capture = getattr(args, "launch_mode", "auto") == "graph"      # auto falls back to eager launches, graph does not

    def record(real_x, z0, m0):
        """Record the step; returns False (and says why, once) when the capture fails and the run is to go on eagerly in
        this process (an op that cannot be captured, the capture rule of ops.capture_rule, an out-of-memory on the
        graph's private pool during a re-record must not end a training that runs eagerly) - never by restarting: the GPU is
        initialised."""
        try:
            tr.capture(real_x, z0, m0, warmup=0)
            return True
        except Exception as e:      # noqa: BLE001 - whatever the capture raised, eager launches still work
            if fatal_capture:
                raise
            tr.graph = None
            if probe_deferred[0]:
                # the probe switched the deferred reduce on for the replay it had picked: eager launches keep the per-layer
                # reduces (the deferred form costs them ~2 %, launch_plan)
                tr.set_defer_reduce(False)
                probe_deferred[0] = False
            if rank == 0:
                print("launch (auto): recording the step failed (%s: %s) - continuing with eager launches, weight-gradient reduce %s"
                      % (type(e).__name__, e, "deferred" if tr.defer_reduce else "per layer"))
            return False

    # --launch_mode auto: 1 untimed + 4 timed eager iterations, carried ACROSS epochs (a run with fewer than five batches per
    # epoch still decides); the epoch-end host sync, checkpoint and print stay out of its clock
    probe = {"n": 0, "issue": 0.0, "wall": 0.0, "start": None}
    probe_deferred = [False]

    print("Starting Training Loop...")
    for epoch in range(args.epochs):
        d_run = torch.zeros((), device=device)
        g_run = torch.zeros((), device=device)
        n_d = n_g = 0
        if use_graph == "probe" and probe["n"] >= 1:
            torch.cuda.synchronize()
            probe["start"] = time.perf_counter()          # the probe goes on in this epoch: restart its clock on an idle GPU
        for data_b in data:
            real_x = data_b[0]
            b = real_x.shape[0]
            # --disc_iters D updates with fresh latents each, then ONE G update on the last fake (train.py:124-169)
            lat = [sample() for _ in range(args.disc_iters)]
            if use_graph == "probe" and b == args.batch_size:
                # auto launch mode: 1 untimed + 4 timed eager iterations, host issue time against synchronised time
                if probe["n"] == 1:
                    torch.cuda.synchronize()
                    probe["start"] = time.perf_counter()
                t0 = time.perf_counter()
                _, _, g_loss = tr.step(real_x, [l[0] for l in lat], [l[1] for l in lat])
                if probe["n"] >= 1:
                    probe["issue"] += time.perf_counter() - t0
                probe["n"] += 1
                if probe["n"] == 5:
                    torch.cuda.synchronize()
                    wall = probe["wall"] + time.perf_counter() - probe["start"]
                    use_graph = probe["issue"] > 0.9 * wall
                    if use_graph and defer_with_replay(args):
                        tr.set_defer_reduce(True)          # the schedule that pays under replay (launch_plan)
                        probe_deferred[0] = True
                    if rank == 0:
                        print("launch (auto): the host issues an iteration in %.2f ms, the GPU finishes one every %.2f ms -> %s"
                              % (probe["issue"] / 4 * 1e3, wall / 4 * 1e3, "hipGraph replay" if use_graph else "eager"))
            elif use_graph is True and b == args.batch_size:
                # the first full batch runs eagerly (every lazily created buffer then exists) and is recorded afterwards
                # (recording does not train); later batches replay.  A decayed learning rate re-records; a short last
                # batch of an epoch (another shape) runs eagerly.
                z0, m0 = lat[0]
                if recorded_lr is None:
                    _, _, g_loss = tr.step(real_x, z0, m0)
                    losses = list(tr.d_losses)
                    if record(real_x, z0, m0):
                        recorded_lr, graph_losses = (tr.optD.lr, tr.optG.lr), tr.d_losses      # the graph's static loss tensors
                    else:
                        use_graph = False
                    tr.d_losses = losses                      # this iteration's losses are the eager step's
                else:
                    if recorded_lr != (tr.optD.lr, tr.optG.lr):
                        if record(real_x, z0, m0):
                            recorded_lr, graph_losses = (tr.optD.lr, tr.optG.lr), tr.d_losses
                        else:
                            use_graph = False
                    if use_graph:
                        _, _, g_loss = tr.step_graphed(real_x, z0, m0)
                        tr.d_losses = graph_losses
                    else:
                        _, _, g_loss = tr.step(real_x, z0, m0)
            else:
                _, _, g_loss = tr.step(real_x, [l[0] for l in lat], [l[1] for l in lat])
            for d_real, d_fake in tr.d_losses:
                d_run += d_fake * args.num_images + d_real * b
            g_run += g_loss * args.num_images
            n_d += b
            n_g += args.num_images
        if use_graph == "probe" and probe["n"] >= 2:
            torch.cuda.synchronize()                      # close the probe's clock before the epoch-end work
            probe["wall"] += time.perf_counter() - probe["start"]
        if args.decay_lr:
            f = lr_factor(args.decay_lr, epoch + 1)
            tr.optD.lr, tr.optG.lr = args.lr_D * f, args.lr_G * f
        d_l, g_l = float(d_run) / n_d, float(g_run) / n_g      # the only host sync of the epoch
        if rank == 0:
            print('[%d/%d]\tLoss_D: %.4f\tLoss_G: %.4f, elapsed_time = %.4f min'
                  % (epoch + 1, args.epochs, d_l, g_l, U.elapsed_time(start) / 60))
        G_losses.append(g_l), D_losses.append(d_l)
        last = epoch + 1 == args.epochs
        if rank == 0 and args.saving_rate is not None and ((epoch + 1) % args.saving_rate == 0 or last):
            torch.save({'epoch': epoch + 1, 'netG_state_dict': netG.state_dict(), 'netD_state_dict': netD.state_dict(),
                        'Gloss': G_losses, 'Dloss': D_losses, 'args': args, 'seed': seed},
                       filename + str(epoch + 1) + ".pth")
        if rank == 0 and last and args.ema:
            torch.save({'netG_state_dict': netG_ema.state_dict(), 'args': args}, filename + "_ema.pth")
        if rank == 0 and last:
            plot_losses(G_losses, D_losses, filename + 'losses.png')


def main(argv=None):
    train(U.prepare_parser().parse_args(argv))


if __name__ == '__main__':
    main()
