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
