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
