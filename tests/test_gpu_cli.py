"""GPU end-to-end: the train.py / test_sample.py command lines (reference flags) on a synthetic
texture: train a tiny model for one short epoch, write the reference-format checkpoints, then
sample a non-multiple-of-patch image from the EMA checkpoint."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_then_sample_cli(tmp_path):
    from PIL import Image
    from infinite_texture_gans_amd import train as T, test_sample as S
    rng = np.random.RandomState(0)
    Image.fromarray(rng.randint(0, 255, (120, 160, 3), dtype=np.uint8)).save(tmp_path / "tex.jpg")
    out = tmp_path / "cp"
    T.main(["--data_path", str(tmp_path / "tex.jpg"), "--random_crop", "48", "--padding_mode", "local",
            "--type_norm", "BN", "--G_ch", "4", "--D_ch", "4", "--z_dim", "8", "--n_layers_G", "4",
            "--n_layers_D", "3", "--batch_size", "4", "--num_images", "2", "--sampling", "16", "--epochs", "2",
            "--saving_rate", "1", "--leak_G", "0.02", "--spec_norm_D", "--smooth", "--ema", "--seed", "3",
            "--decay_lr", "exp", "--fname", str(out)])
    for f in ("2_1.pth", "2_2.pth", "2__ema.pth"):
        assert (out / f).exists(), f
    ck = torch.load(out / "2_2.pth", map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "netG_state_dict", "netD_state_dict", "Gloss", "Dloss", "args", "seed"}
    assert "model.0.weight_orig" in ck["netD_state_dict"] and "block1.bn1.running_mean" in ck["netG_state_dict"]
    assert all(np.isfinite(ck["Gloss"])) and all(np.isfinite(ck["Dloss"]))
    # DataParallel-style prefixes must load too (reference test_sample.py:35-36)
    ema = torch.load(out / "2__ema.pth", map_location="cpu", weights_only=False)
    ema["netG_state_dict"] = {"module." + k: v for k, v in ema["netG_state_dict"].items()}
    torch.save(ema, out / "2__ema.pth")
    S.main(["--model_path", str(out / "2__ema.pth"), "--output_resolution_height", "100",
            "--output_resolution_width", "170", "--output_name", "gen.png"])
    img = Image.open(out / "gen.png")
    assert img.size == (170, 100)


def _tiny_train(tmp_path, name, extra):
    from PIL import Image
    from infinite_texture_gans_amd import train as T
    tex = tmp_path / "tex.jpg"
    if not tex.exists():
        Image.fromarray(np.random.RandomState(0).randint(0, 255, (120, 160, 3), dtype=np.uint8)).save(tex)
    out = tmp_path / name
    T.main(["--data_path", str(tex), "--random_crop", "48", "--padding_mode", "local", "--type_norm", "BN", "--G_ch", "4",
            "--D_ch", "4", "--z_dim", "8", "--n_layers_G", "4", "--n_layers_D", "3", "--batch_size", "4", "--num_images", "2",
            "--sampling", "24", "--epochs", "3", "--saving_rate", "3", "--leak_G", "0.02", "--spec_norm_D", "--smooth", "--seed", "5",
            "--decay_lr", "exp", "--fname", str(out)] + extra)
    return torch.load(out / "3_3.pth", map_location="cpu", weights_only=False)      # <epochs>_<epoch>.pth (reference utils.py:129-135)


def test_train_cli_graph_replay_equals_eager_through_a_learning_rate_decay(tmp_path, monkeypatch):
    """the same seeds through `--launch_mode graph` and `--launch_mode eager` with `--decay_lr exp`.  The graph is
    recorded after the first (eager) batch and RE-recorded after every epoch's decay - the learning rates are kernel arguments
    (checked: three recordings, at lr, 0.99 lr, 0.99^2 lr; reference train.py:60-70,183-185) - and both runs end in the same
    losses and parameters.  Not bit-identical: the replay keeps D(real)'s weight gradients on their branch stream (another
    accumulation order into the flat gradient), and 18 Adam steps with beta1 = 0 amplify that rounding - 1e-3 on the epoch
    losses (measured 2e-5), 6 % rel-L2 on the parameter UPDATES (measured 1-2 %: sign flips of Adam steps whose gradient is
    rounding noise; the learning rates themselves are checked exactly through the recordings)."""
    from infinite_texture_gans_amd import engine
    real_capture, seen = engine.Trainer.capture, []

    def counting(self, *a, **k):
        seen.append((self.optD.lr, self.optG.lr))
        return real_capture(self, *a, **k)
    monkeypatch.setattr(engine.Trainer, "capture", counting)
    # (--wgrad_reduce layer in both: the deferred reduce sums the slabs in another order, and Adam turns the rounding noise of
    # the mathematically-zero bias gradients in front of a BatchNorm into +-lr steps)
    a = _tiny_train(tmp_path, "graph", ["--launch_mode", "graph", "--wgrad_reduce", "layer"])
    monkeypatch.setattr(engine.Trainer, "capture", real_capture)
    assert len(seen) == 3, seen
    for e, (ld, lg) in enumerate(seen):
        assert abs(ld - a["args"].lr_D * 0.99 ** e) < 1e-12 and abs(lg - a["args"].lr_G * 0.99 ** e) < 1e-12, seen
    b = _tiny_train(tmp_path, "eager", ["--launch_mode", "eager", "--wgrad_reduce", "layer"])
    assert np.allclose(a["Gloss"], b["Gloss"], rtol=1e-3) and np.allclose(a["Dloss"], b["Dloss"], rtol=1e-3), (a["Gloss"], b["Gloss"], a["Dloss"], b["Dloss"])
    init = _tiny_train(tmp_path, "init", ["--launch_mode", "eager", "--lr_D", "0", "--lr_G", "0"])      # the initial parameters: a run that does not move them
    for net in ("netG_state_dict", "netD_state_dict"):
        num = den = 0.0
        for k, v in a[net].items():
            if v.is_floating_point() and "running" not in k and not k.endswith((".bias", "weight_u", "weight_v")):
                init_k = init[net][k]
                da, db = v - init_k, b[net][k] - init_k
                num += float((da - db).pow(2).sum())
                den += float(db.pow(2).sum())
            elif not v.is_floating_point():
                assert torch.equal(v, b[net][k]), (net, k)
        assert den > 0 and (num / den) ** 0.5 < 6e-2, (net, (num / den) ** 0.5)


def test_train_cli_auto_mode_falls_back_to_eager_when_recording_fails(tmp_path, monkeypatch, capsys):
    """in `--launch_mode auto` a failing Trainer.capture (here: forced) prints one warning and the SAME process
    goes on with eager launches - and ends where the eager run ends; `--launch_mode graph` still raises."""
    from infinite_texture_gans_amd import engine
    monkeypatch.setenv("ITG_GRAPH", "1")                     # auto mode told to replay (skips the timing probe)
    real_capture = engine.Trainer.capture

    def broken(self, *a, **k):
        raise RuntimeError("forced capture failure")
    monkeypatch.setattr(engine.Trainer, "capture", broken)
    a = _tiny_train(tmp_path, "auto", ["--launch_mode", "auto", "--wgrad_reduce", "layer"])
    assert "continuing with eager launches" in capsys.readouterr().out
    with pytest.raises(RuntimeError, match="forced capture failure"):
        _tiny_train(tmp_path, "forced", ["--launch_mode", "graph", "--wgrad_reduce", "layer"])
    monkeypatch.setattr(engine.Trainer, "capture", real_capture)
    monkeypatch.delenv("ITG_GRAPH")
    b = _tiny_train(tmp_path, "eager", ["--launch_mode", "eager", "--wgrad_reduce", "layer"])
    # (two eager trainings of one process differ by ~2e-5 themselves: the step's streams are placed by measurement, and the
    # accumulation order into the flat gradient follows the placement)
    assert np.allclose(a["Gloss"], b["Gloss"], rtol=1e-3) and np.allclose(a["Dloss"], b["Dloss"], rtol=1e-3)


def test_train_cli_bf16_flag(tmp_path):
    """--bf16 (BASELINE config 3's MFMA path) through the command line; the process-wide precision is restored."""
    from PIL import Image
    from infinite_texture_gans_amd import ops, train as T
    rng = np.random.RandomState(1)
    Image.fromarray(rng.randint(0, 255, (96, 96, 3), dtype=np.uint8)).save(tmp_path / "tex.jpg")
    out = tmp_path / "cp"
    prev = ops.MFMA_PRECISION
    try:
        T.main(["--data_path", str(tmp_path / "tex.jpg"), "--random_crop", "48", "--padding_mode", "local",
                "--type_norm", "BN", "--G_ch", "4", "--D_ch", "4", "--z_dim", "8", "--n_layers_G", "5", "--attention",
                "--n_layers_D", "3", "--batch_size", "4", "--num_images", "2", "--sampling", "8", "--epochs", "1",
                "--saving_rate", "1", "--spec_norm_D", "--seed", "3", "--bf16", "--fname", str(out)])
        assert ops.MFMA_PRECISION == ops.PREC_BF16
    finally:
        ops.MFMA_PRECISION = prev
    ck = torch.load(out / "1_1.pth", map_location="cpu", weights_only=False)
    assert all(np.isfinite(ck["Gloss"])) and all(np.isfinite(ck["Dloss"]))


def test_reference_style_caller_runs_on_the_drop_in_module_names():
    """A training loop written the way the reference's train.py is (top-level `utils` / `models` imports, prepare_models,
    torch.optim.Adam over module.parameters(), nn.BCEWithLogitsLoss, utils.sample_from_gen_PatchByPatch_train drawing its
    latents from the global CPU generator) runs on this build's modules and reproduces the reference's own losses and
    post-step parameters (golden train_bn_nl4_sn, 2 steps)."""
    import torch.nn as nn
    import torch.optim as optim
    import utils                                     # the reference's module names (repo root shims)
    from models import generators, discriminators    # noqa: F401
    from helpers import load, state, rel_l2
    fx = load("train_bn_nl4_sn")
    dev = torch.device("cuda")
    args = utils.prepare_parser().parse_args([str(x) for x in fx["argv"]] + [
        "--padding_mode", "local", "--G_ch", "4", "--D_ch", "4", "--z_dim", "8", "--leak_G", "0.02", "--batch_size", "2",
        "--num_images", "2", "--beta1", "0.0"])
    netG, netD = utils.prepare_models(args, dev)
    assert type(netG).__name__ == "ResidualPatchGenerator" and isinstance(netG, generators.ResidualPatchGenerator)
    netG.load_state_dict(state(fx, "G0/")), netD.load_state_dict(state(fx, "D0/"))
    netG.train(), netD.train()
    optD = optim.Adam(netD.parameters(), lr=args.lr_D, betas=(float(args.beta1), args.beta2))
    optG = optim.Adam(netG.parameters(), lr=args.lr_G, betas=(float(args.beta1), args.beta2))
    crit = nn.BCEWithLogitsLoss().to(dev)
    label_t = 0.9 if args.smooth else 1
    for s in range(int(fx["steps"])):
        real_x = torch.from_numpy(fx["real_x%d" % s]).to(dev)
        netD.zero_grad()
        real_logit = netD(real_x)
        d_real = crit(real_logit, torch.full_like(real_logit, label_t))
        d_real.backward()
        torch.manual_seed(201 + 100 + s)             # the seed the fixture's sampler call ran under
        fake_x = utils.sample_from_gen_PatchByPatch_train(netG, args.z_dim, args.base_res, args.map_dim,
                                                          num_images=args.num_images,
                                                          num_patches_height=args.num_patches_height,
                                                          num_patches_width=args.num_patches_width, device=dev)
        fake_logit = netD(fake_x.detach())
        d_fake = crit(fake_logit, torch.zeros_like(fake_logit))
        d_fake.backward()
        optD.step()
        netG.zero_grad()
        fake_logit = netD(fake_x)
        g_loss = crit(fake_logit, torch.full_like(fake_logit, label_t))
        g_loss.backward()
        optG.step()
        got = [float(d_real), float(d_fake), float(g_loss)]
        assert np.allclose(got, fx["loss%d" % s], rtol=1e-4, atol=1e-6), (s, got, fx["loss%d" % s])
    sd = netD.state_dict()
    for k, v in state(fx, "D1/").items():
        assert rel_l2(sd[k].double().cpu(), v.double()) < 2e-3, k


def test_single_image_loader_on_the_device_returns_exact_windows(tmp_path):
    """f1 on the GPU (reference datasets/datasets_classes.py:12-51 behind utils.prepare_data :158-191): the decoded texture
    lives in HBM as ToTensor -> Normalize(0.5, 0.5) bit for bit; a batch of random crops is cut on the device and every
    item is an exact window of it at the offsets the generator drew (torch.randint order: all rows, then all columns);
    center_crop wins over random_crop and uses torchvision's rounded offsets; the loader yields {0: batch} with a short
    last batch (DataLoader(drop_last=False)); the batch feeds the train step unchanged."""
    from PIL import Image
    from infinite_texture_gans_amd.data import single_image, CropLoader
    from infinite_texture_gans_amd import utils as U
    rng = np.random.RandomState(5)
    a = rng.randint(0, 256, (75, 91, 3), dtype=np.uint8)
    path = str(tmp_path / "tex.png")
    Image.fromarray(a).save(path)
    want = (torch.from_numpy(a).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5
    ds = single_image(path, "png", random_crop=32, sampling=20, device="cuda")
    assert ds.img.is_cuda and torch.equal(ds.img.cpu(), want)
    g = torch.Generator().manual_seed(11)
    b = ds.crop_batch(6, g)
    assert b.is_cuda and b.shape == (6, 3, 32, 32) and b.is_contiguous()
    g2 = torch.Generator().manual_seed(11)
    ys = torch.randint(0, 75 - 32 + 1, (6,), generator=g2).tolist()
    xs = torch.randint(0, 91 - 32 + 1, (6,), generator=g2).tolist()
    for i, (y, x) in enumerate(zip(ys, xs)):
        assert torch.equal(b[i].cpu(), want[:, y:y + 32, x:x + 32]), i
    dc = single_image(path, "png", center_crop=40, random_crop=32, device="cuda")
    t, l = int(round((75 - 40) / 2.0)), int(round((91 - 40) / 2.0))
    cb = dc.crop_batch(3)
    assert cb.is_cuda and all(torch.equal(cb[i].cpu(), want[:, t:t + 40, l:l + 40]) for i in range(3))
    assert torch.equal(dc[0][0].cpu(), want[:, t:t + 40, l:l + 40])
    batches = list(CropLoader(ds, 8, seed=3))
    assert [set(d) for d in batches] == [{0}] * 3 and [d[0].shape[0] for d in batches] == [8, 8, 4]
    assert all(d[0].is_cuda and d[0].dtype == torch.float32 for d in batches)
    # utils.prepare_data's (loader, dataset) pair on the device, as train.py uses it
    args = U.prepare_parser().parse_args(["--data_path", path, "--data_ext", "png", "--random_crop", "32", "--sampling", "10",
                                          "--batch_size", "4"])
    loader, dset = U.prepare_data(args, device="cuda", seed=1)
    first = next(iter(loader))[0]
    assert first.is_cuda and first.shape == (4, 3, 32, 32) and len(dset) == 10
    hits = [(y, x) for y in range(75 - 31) for x in range(91 - 31) if torch.equal(want[:, y:y + 32, x:x + 32], first[0].cpu())]
    assert hits


def _sample_worker(rank, world, port, ckpt, name, seed):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      ITG_DIST_BACKEND="gloo", ITG_SAMPLE_SEED=str(seed))
    from infinite_texture_gans_amd import test_sample as S
    S.main(["--model_path", ckpt, "--output_resolution_height", "150", "--output_resolution_width", "100", "--output_name", name])


def _checkpoint(tmp_path, attention):
    """A reference-format generator checkpoint with random-init weights (what test_sample.py reads: args + netG_state_dict)."""
    from infinite_texture_gans_amd import utils as U
    flags = ["--padding_mode", "local", "--type_norm", "BN", "--G_ch", "4", "--z_dim", "8", "--n_layers_G", "4", "--leak_G", "0.02"]
    args = U.prepare_parser().parse_args(flags + (["--attention"] if attention else []))
    torch.manual_seed(5)
    netG, _ = U.prepare_models(args, "cpu")
    d = tmp_path / ("att" if attention else "bn")
    d.mkdir()
    torch.save({"args": args, "netG_state_dict": netG.state_dict()}, d / "g.pth")
    return str(d / "g.pth"), d


def test_sample_cli_on_two_ranks_row_sharded_and_replica_mode(tmp_path):
    """test_sample.py under a 2-rank launch (two processes on this one GPU over gloo; a real launch is one rank per GPU over
    RCCL): a BN generator shards the ONE image by patch rows and rank 0 collects the strips as tensors (utils.gather_strips;
    round 3 pickled them) - the image equals the single-process one from the same seed; an attention checkpoint streams with
    carried state and does not shard: replica mode, rank r writes its own image <stem>_rank<r> from seed + r."""
    import torch.multiprocessing as mp
    from PIL import Image
    from test_gpu_model import free_port
    from infinite_texture_gans_amd import test_sample as S
    ckpt, d = _checkpoint(tmp_path, attention=False)
    os.environ["ITG_SAMPLE_SEED"] = "7"
    try:
        S.main(["--model_path", ckpt, "--output_resolution_height", "150", "--output_resolution_width", "100", "--output_name", "one.png"])
    finally:
        del os.environ["ITG_SAMPLE_SEED"]
    mp.spawn(_sample_worker, args=(2, free_port(), ckpt, "two.png", 7), nprocs=2, join=True)
    one, two = np.asarray(Image.open(d / "one.png")).astype(int), np.asarray(Image.open(d / "two.png")).astype(int)
    assert one.shape == two.shape == (150, 100, 3)
    assert np.abs(one - two).max() <= 1, np.abs(one - two).max()            # 8-bit rounding of equal-to-1e-6 floats
    ckpt, d = _checkpoint(tmp_path, attention=True)
    mp.spawn(_sample_worker, args=(2, free_port(), ckpt, "att.png", 11), nprocs=2, join=True)
    r0, r1 = np.asarray(Image.open(d / "att_rank0.png")).astype(int), np.asarray(Image.open(d / "att_rank1.png")).astype(int)
    assert r0.shape == r1.shape == (150, 100, 3) and np.abs(r0 - r1).max() > 8      # two different images
    os.environ["ITG_SAMPLE_SEED"] = "11"
    try:
        S.main(["--model_path", ckpt, "--output_resolution_height", "150", "--output_resolution_width", "100", "--output_name", "att_one.png"])
    finally:
        del os.environ["ITG_SAMPLE_SEED"]
    assert np.abs(np.asarray(Image.open(d / "att_one.png")).astype(int) - r0).max() <= 1      # rank 0's replica = the seed-11 image


def _bench_two_ranks(workload, extra=()):
    """`python bench.py --gpus 2 --workload ...` as the driver starts it on a multi-GPU node, rehearsed on ONE GPU: the parent
    spawns the two ranks itself (bench.relaunch -> torch.distributed.run, never an exec from a GPU process), both ranks use device
    0 (ITG_FORCE_DEVICE) and the collectives run over gloo (RCCL refuses two ranks on one device)."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ITG_DIST_BACKEND="gloo", ITG_FORCE_DEVICE="0", PYTHONPATH=ROOT)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-membound", "--no-direct", "--workload", workload] + list(extra)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["config1", "config4", "config5"])
def test_bench_two_rank_launch_prints_one_line_with_both_ranks(workload):
    """the driver's first multi-GPU run of bench.py must not fail on plumbing.  Two ranks of every sharded
    workload - config 1 (data parallel, gradient all-reduce), config 4 (patch rows of ONE batch, halo rows + band gather +
    gradient all-reduce), config 5 (inference bands, halo rows + strip gather) - run to the JSON line: n_gpus 2, ranks 2, the
    partitioning named, finite losses / a finite rate."""
    import math
    out = _bench_two_ranks(workload, ("--out", "768") if workload == "config5" else ())
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"].get("ranks") == 2 and out["config"].get("backend") == "gloo", out["config"]
    assert math.isfinite(out["value"]) and out["value"] > 0 and out["ms_per_step"] > 0
    par = out["config"]["parallelism"]
    if workload == "config1":
        assert out["scaling"] == "weak" and par.startswith("dp2") and out["config"]["global_batch"] == 16
    else:
        assert out["scaling"] == "strong" and "patch rows over 2 rank" in par, par
    if workload == "config5":
        assert out["config"]["finite"] is True
    else:
        assert len(out["config"]["last_losses"]) == 3 and all(math.isfinite(v) for v in out["config"]["last_losses"])
