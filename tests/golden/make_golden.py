#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  The
fixtures are data (inputs + expected outputs as .npz); no reference source or
bytecode is written anywhere (PYTHONDONTWRITEBYTECODE is forced).

    python tests/golden/make_golden.py

Import recipe: ``import utils`` first, then ``models``
(circular import, reference models/layers.py:5 <-> utils.py:12-13).  The train
step is restated here exactly as reference train.py:122-171 with injected
real_x / latents (the reference draws them from the global CPU RNG, so the
same seed is set before each sampler call and the tensors are recorded).
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import utils as R  # noqa: E402  (reference utils; must precede models)
from models import generators as RG, discriminators as RD, layers as RL  # noqa: E402

torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()


def sd_np(sd, prefix):
    return {prefix + k: npy(v) for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------- 1. merge / crop / LocalPadder (train)
def gen_patch_ops():
    out = {}
    g = torch.Generator().manual_seed(11)
    cases = [(3, 3, 4, "replicate"), (3, 3, 4, "constant"), (4, 4, 2, "replicate"),
             (2, 5, 3, "constant"), (2, 5, 3, "replicate")]
    for i, (gh, gw, p, outer) in enumerate(cases):
        n, c = 2, 3
        x = torch.randn(n * gh * gw, c, p, p, generator=g, requires_grad=True)
        RL.LocalPadder.set_attributes(gh, gw, outer, 1, 2)
        pad = RL.LocalPadder(True)
        pad.train()
        y = pad(x, "1st_row_1st_col")
        dy = torch.randn(y.shape, generator=g)
        (dx,) = torch.autograd.grad(y, x, dy)
        m = R.merge_patches_into_image(x.detach(), gh, gw)
        out.update({"lp%d_cfg" % i: np.array([gh, gw, p, outer == "replicate"]), "lp%d_x" % i: npy(x),
                    "lp%d_y" % i: npy(y), "lp%d_dy" % i: npy(dy), "lp%d_dx" % i: npy(dx),
                    "lp%d_merged" % i: npy(m)})
    # start-layer variant: already merged input, crop only (generators.py:59)
    RL.LocalPadder.set_attributes(3, 3, "replicate", 1, 2)
    pad = RL.LocalPadder(False)
    pad.train()
    z = torch.randn(2, 5, 3 * 4 + 2, 3 * 4 + 2, generator=g)
    out["start_z"] = npy(z)
    out["start_y"] = npy(pad(z, "1st_row_1st_col"))
    # generic crop with stride != size (build_z style, utils.py:232)
    img = torch.randn(1, 2, 5 * 4 + 2, 7 * 4 + 2, generator=g)
    out["crop_img"] = npy(img)
    out["crop_out"] = npy(R.crop_images(img, 3 * 4 + 2, 3 * 4 + 2, 2 * 4))
    save("patch_ops", **out)


# --------------------------------------------------------------------------- helpers for tiny models
def make_args(extra):
    base = ["--padding_mode", "local", "--G_ch", "4", "--D_ch", "4", "--z_dim", "8", "--leak_G", "0.02",
            "--batch_size", "2", "--num_images", "2", "--beta1", "0.0"]
    a = R.prepare_parser().parse_args(base + extra)
    a.beta1 = float(a.beta1)
    return a


def build(args, seed):
    torch.manual_seed(seed)
    netG, netD = R.prepare_models(args, torch.device("cpu"))
    # make BN affine / attention gamma non-trivial so the fixtures exercise them
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in netG.state_dict().items():
            if k.endswith("bn1.weight") or k.endswith("bn2.weight") or k == "bn.weight":
                v.copy_(1 + 0.1 * torch.randn(v.shape, generator=g))
            if (k.endswith("bn1.bias") or k.endswith("bn2.bias") or k == "bn.bias"
                    or k.endswith("conv.bias") or k.endswith("conv3.bias") or k.endswith("embed.bias")):
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
            if k == "attention.gamma":
                v.fill_(0.3)
            if k.endswith("embed.weight"):
                v.copy_(v + 0.05 * torch.randn(v.shape, generator=g))
        for k, v in netD.state_dict().items():
            if k.endswith(".bias"):
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
    return netG, netD


def latents(args, netG, seed):
    """Replays the RNG draws of utils.py:503-519 (local padding) / utils.py:556-566 (zeros) and records them."""
    gh, gw, b = args.num_patches_height, args.num_patches_width, args.base_res
    torch.manual_seed(seed)
    if args.padding_mode != "local":
        z = torch.randn(args.num_images, args.z_dim, b, b)
        maps_full = []
        if netG.type_norm == "SSM":
            for i in range(netG.n_layers_G):
                maps_full.append(torch.randn(args.num_images, args.map_dim, (2 ** i) * b, (2 ** i) * b))
        return z, maps_full
    z = torch.randn(args.num_images, args.z_dim, gh * b + 2, gw * b + 2)
    maps_full = []
    if netG.type_norm == "SSM":
        for i in range(netG.n_layers_G):
            r = (2 ** i) * b
            maps_full.append(torch.randn(args.num_images, args.map_dim, gh * r + 4, gw * r + 4))
    return z, maps_full


def ref_sampler(args, netG, seed):
    torch.manual_seed(seed)
    if args.padding_mode != "local":      # reference train.py:143-144
        return R.sample_from_gen(netG, args.z_dim, args.base_res, num_images=args.num_images, device="cpu")
    return R.sample_from_gen_PatchByPatch_train(
        netG, args.z_dim, args.base_res, args.map_dim, num_images=args.num_images,
        num_patches_height=args.num_patches_height, num_patches_width=args.num_patches_width, device="cpu")


# --------------------------------------------------------------------------- 2. forward fixtures
def gen_forward(tag, extra, seed):
    args = make_args(extra)
    netG, netD = build(args, seed)
    netG.train(), netD.train()
    out = dict(argv=np.array(extra))
    out.update(sd_np(netG.state_dict(), "G0/"))
    out.update(sd_np(netD.state_dict(), "D0/"))
    z, maps_full = latents(args, netG, seed + 7)
    out["z"] = npy(z)
    for i, m in enumerate(maps_full):
        out["map%d" % i] = npy(m)
    fake = ref_sampler(args, netG, seed + 7)
    out["fake"] = npy(fake)
    out["d_fake"] = npy(netD(fake))
    out.update(sd_np(netG.state_dict(), "G1/"))  # BN running stats after one training forward
    out.update(sd_np(netD.state_dict(), "D1/"))  # SN u/v after one power iteration
    save("fwd_" + tag, **out)


# --------------------------------------------------------------------------- 3. train-step fixtures
def gen_train(tag, extra, seed, steps=2):
    """The iteration of reference train.py:122-180 incl. --disc_iters > 1 (D step repeated with fresh
    latents, then ONE G step on the last fake_x) and --ema (train.py:38-45,176-180)."""
    args = make_args(extra)
    netG, netD = build(args, seed)
    netG.train(), netD.train()
    out = dict(argv=np.array(extra), steps=np.array(steps))
    out.update(sd_np(netG.state_dict(), "G0/"))
    out.update(sd_np(netD.state_dict(), "D0/"))
    netG_ema = None
    if args.ema:        # train.py:38-45
        netG_ema, _ = R.prepare_models(args, torch.device("cpu"))
        with torch.no_grad():
            for key in netG_ema.state_dict():
                netG_ema.state_dict()[key].data.copy_(netG.state_dict()[key].data)
    optD = torch.optim.Adam(netD.parameters(), lr=args.lr_D, betas=(args.beta1, args.beta2))
    optG = torch.optim.Adam(netG.parameters(), lr=args.lr_G, betas=(args.beta1, args.beta2))
    crit = torch.nn.BCEWithLogitsLoss()
    lt = 0.9 if args.smooth else 1
    crop = args.random_crop
    g = torch.Generator().manual_seed(seed + 3)
    di = args.disc_iters
    for s in range(steps):
        real_x = torch.rand(args.batch_size, 3, crop, crop, generator=g) * 2 - 1
        out["real_x%d" % s] = npy(real_x)
        d_losses = []
        for d in range(di):
            # latent seed and key names of the disc_iters == 1 fixtures are kept (z0, z1, map0_1 ...)
            lseed = seed + 100 + s if di == 1 else seed + 100 + 10 * s + d
            sfx = "%d" % s if di == 1 else "%d_%d" % (s, d)
            z, maps_full = latents(args, netG, lseed)
            out["z" + sfx] = npy(z)
            for i, m in enumerate(maps_full):
                out["map%s_%d" % (sfx, i)] = npy(m)
            # ---- reference train.py:124-153
            netD.zero_grad()
            real_logit = netD(real_x)
            lab = torch.FloatTensor(1).fill_(lt).expand_as(real_logit)
            d_real = crit(real_logit, lab)
            d_real.backward()
            fake_x = ref_sampler(args, netG, lseed)
            fake_logit = netD(fake_x.detach())
            lab = torch.FloatTensor(1).fill_(0).expand_as(fake_logit)
            d_fake = crit(fake_logit, lab)
            d_fake.backward()
            if s == 0 and d == 0:
                for k, p in netD.named_parameters():
                    out["gradD0/" + k] = npy(p.grad)
            optD.step()
            d_losses += [d_real.item(), d_fake.item()]
        # ---- reference train.py:161-169
        netG.zero_grad()
        fake_logit = netD(fake_x)
        lab = torch.FloatTensor(1).fill_(lt).expand_as(fake_logit)
        g_loss = crit(fake_logit, lab)
        g_loss.backward()
        if s == 0:
            for k, p in netG.named_parameters():
                out["gradG0/" + k] = npy(p.grad)
        optG.step()
        if netG_ema is not None:    # train.py:176-180
            with torch.no_grad():
                for key in netG.state_dict():
                    netG_ema.state_dict()[key].data.copy_(netG_ema.state_dict()[key].data * args.ema_decay
                                                          + netG.state_dict()[key].data * (1 - args.ema_decay))
        out["loss%d" % s] = np.array(d_losses[-2:] + [g_loss.item()], dtype=np.float64)
        if di > 1:
            out["dloss%d" % s] = np.array(d_losses, dtype=np.float64)    # every D iteration: real, fake, real, fake ...
    out["fake_last"] = npy(fake_x)
    out.update(sd_np(netG.state_dict(), "G1/"))
    out.update(sd_np(netD.state_dict(), "D1/"))
    if netG_ema is not None:
        out.update(sd_np(netG_ema.state_dict(), "E1/"))
    save("train_" + tag, **out)


# --------------------------------------------------------------------------- 4. inference tiling
def gen_infer(tag, extra, seed, out_h, out_w):
    args = make_args(extra)
    netG, _ = build(args, seed)
    # non-trivial running stats
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for k, v in netG.state_dict().items():
            if k.endswith("running_mean"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            if k.endswith("running_var"):
                v.copy_(1 + 0.2 * torch.rand(v.shape, generator=g))
    netG.eval()
    out = dict(argv=np.array(extra), out_hw=np.array([out_h, out_w]))
    out.update(sd_np(netG.state_dict(), "G0/"))
    # replay build_z / build_maps draws (utils.py:228,246): z first, then maps 0..nl-1
    p = (2 ** (netG.n_layers_G - 1)) * args.base_res
    sh = int(np.ceil((out_h / p - 1) / 2))
    sw = int(np.ceil((out_w / p - 1) / 2))
    th, tw = sh * 2 + 1, sw * 2 + 1
    torch.manual_seed(seed + 9)
    zf = torch.randn(1, args.z_dim, th * args.base_res + 2, tw * args.base_res + 2)
    out["z_full"] = npy(zf)
    if netG.type_norm == "SSM":
        for i in range(netG.n_layers_G):
            r = (2 ** i) * args.base_res
            out["map_full%d" % i] = npy(torch.randn(1, args.map_dim, th * r + 4, tw * r + 4))
    torch.manual_seed(seed + 9)
    with torch.no_grad():
        img = R.sample_from_gen_PatchByPatch_test(
            netG, z_dim=args.z_dim, base_res=args.base_res, map_dim=args.map_dim, num_images=1,
            device="cpu", output_resolution_height=out_h, output_resolution_width=out_w)
    out["image"] = npy(img)
    save("infer_" + tag, **out)


# --------------------------------------------------------------------------- 5. non-local baseline sampler (+ tiling)
def gen_zeros_infer(tag, extra, seed, base_res_out):
    """padding_mode='zeros' generator in eval mode through reference utils.sample_from_gen, plain and with
    --tiles (tile_process, utils.py:401-470), as test_sample.py:70-73 calls it."""
    args = make_args(extra)
    netG, _ = build(args, seed)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for k, v in netG.state_dict().items():
            if k.endswith("running_mean"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            if k.endswith("running_var"):
                v.copy_(1 + 0.2 * torch.rand(v.shape, generator=g))
    netG.eval()
    out = dict(argv=np.array(extra), base_res_out=np.array(base_res_out))
    out.update(sd_np(netG.state_dict(), "G0/"))
    torch.manual_seed(seed + 9)
    out["z"] = npy(torch.randn(1, args.z_dim, base_res_out, base_res_out))
    with torch.no_grad():
        torch.manual_seed(seed + 9)
        out["image"] = npy(R.sample_from_gen(netG, z_dim=args.z_dim, base_res=base_res_out, num_images=1,
                                             tiles=False, device="cpu"))
        torch.manual_seed(seed + 9)
        out["image_tiles"] = npy(R.sample_from_gen(netG, z_dim=args.z_dim, base_res=base_res_out, num_images=1,
                                                   tiles=True, device="cpu"))
    save("infer_" + tag, **out)


if __name__ == "__main__":
    gen_patch_ops()
    gen_forward("bn_nl4", ["--n_layers_G", "4", "--type_norm", "BN"], 101)
    gen_forward("bn_nl6_const", ["--n_layers_G", "6", "--type_norm", "BN", "--outer_padding", "constant",
                                 "--spec_norm_D", "--base_res", "2", "--num_images", "1"], 102)
    gen_forward("bn_nl5_att", ["--n_layers_G", "5", "--type_norm", "BN", "--attention", "--spec_norm_D",
                               "--base_res", "2", "--num_images", "1"], 103)
    gen_train("bn_nl4_sn", ["--n_layers_G", "4", "--type_norm", "BN", "--spec_norm_D", "--smooth",
                            "--random_crop", "32"], 201)
    gen_train("ssm_nl4", ["--n_layers_G", "4", "--type_norm", "SSM", "--random_crop", "32", "--G_ch", "2",
                          "--num_images", "1"], 202, steps=2)
    gen_train("bn_nl5_att", ["--n_layers_G", "5", "--type_norm", "BN", "--attention", "--spec_norm_D",
                             "--smooth", "--random_crop", "48", "--num_patches_height", "4",
                             "--num_patches_width", "4", "--base_res", "2", "--num_images", "1"], 203, steps=2)
    # BASELINE config 4's workload shape: 4x4 patch grid, BN, no attention; 4 images / 4 real crops so that the
    # row-sharded step can be run on 2 and on 4 ranks
    gen_train("bn_nl4_g44", ["--n_layers_G", "4", "--type_norm", "BN", "--spec_norm_D", "--smooth",
                             "--random_crop", "32", "--num_patches_height", "4", "--num_patches_width", "4",
                             "--base_res", "2", "--num_images", "4", "--batch_size", "4"], 204, steps=2)
    # --disc_iters 2 (D step twice with fresh latents, one G step on the last fake) and --ema (whole state_dict
    # incl. BatchNorm buffers and the int64 counters)
    gen_train("bn_nl4_di2_ema", ["--n_layers_G", "4", "--type_norm", "BN", "--spec_norm_D", "--smooth",
                                 "--random_crop", "32", "--disc_iters", "2", "--ema", "--ema_decay", "0.9"], 205,
              steps=3)
    # D without spectral norm (the CLI default): weight gradients of D(real) and D(fake) accumulate into one buffer
    gen_train("bn_nl4_nosn", ["--n_layers_G", "4", "--type_norm", "BN", "--random_crop", "32"], 206, steps=2)
    # non-local baseline (padding_mode zeros): train step and the eval samplers of test_sample.py:70-73
    gen_train("bn_nl4_zeros", ["--n_layers_G", "4", "--type_norm", "BN", "--padding_mode", "zeros", "--spec_norm_D",
                               "--random_crop", "32"], 207, steps=2)
    gen_zeros_infer("bn_nl4_zeros_tiles", ["--n_layers_G", "4", "--type_norm", "BN", "--padding_mode", "zeros"],
                    304, 40)
    gen_infer("bn_nl4", ["--n_layers_G", "4", "--type_norm", "BN"], 301, 150, 230)
    gen_infer("ssm_nl4", ["--n_layers_G", "4", "--type_norm", "SSM", "--G_ch", "2"], 302, 100, 164)
    gen_infer("bn_nl4_att", ["--n_layers_G", "4", "--type_norm", "BN", "--attention"], 303, 96, 160)
