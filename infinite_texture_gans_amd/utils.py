"""Host-side helpers with the reference's names and call signatures (reference utils.py):
flag parser, device/seed/model preparation, patch merge/crop, latent builders and the
patch-by-patch samplers (training and inference tiling)."""
import argparse
import math
import os
import random
import time

import torch
import torch.nn as nn

from . import ops
from .models.generators import ResidualPatchGenerator
from .models.discriminators import PatchDiscriminator

# (flag, type | 'flag', default, help) - same names, types and defaults as reference utils.py:15-132.
# argparse prefix abbreviation is kept on (README's `--type_norm BN` resolves to --type_norm_G).
_FLAGS = [
    ("data", str, "single_image", "type of data"),
    ("data_path", str, "datasets/241.jpg", "data path"),
    ("data_ext", str, "jpg", "data extension txt, png"),
    ("center_crop", int, None, "center cropping"),
    ("random_crop", int, None, "random cropping"),
    ("resize_h", int, None, "resize for h"),
    ("resize_w", int, None, "resize for w"),
    ("sampling", int, 8000, "randomly sample --sampling instances from the training data if not None"),
    ("D_model", str, "patch_GAN", "discriminator model (only patch_GAN is built)"),
    ("attention", "flag", False, "use attention in the generator"),
    ("img_ch", int, 3, "number of image channels"),
    ("G_ch", int, 52, "base channel multiplier of the generator"),
    ("D_ch", int, 64, "base channel multiplier of the discriminator"),
    ("leak_G", float, 0, "LeakyReLU slope of the generator, 0 = ReLU"),
    ("leak_D", float, 0, "unused by patch_GAN (slope is 0.2)"),
    ("z_dim", int, 128, "channels of the latent input"),
    ("map_dim", int, 1, "channels of the SSM modulation maps"),
    ("spec_norm_D", "flag", False, "spectral normalisation in the discriminator"),
    ("spec_norm_G", "flag", False, "spectral normalisation in the generator"),
    ("n_layers_D", int, 4, "number of discriminator layers"),
    ("n_layers_G", int, 6, "number of generator layers (4, 5 or 6)"),
    ("norm_layer_D", str, None, "normalisation layer in the discriminator (None | batch | instance)"),
    ("base_res", int, 4, "base resolution of G"),
    ("padding_mode", str, "zeros", "padding used in G: zeros or local"),
    ("type_norm_G", str, "BN", "normalisation used in G: BN or SSM"),
    ("lr_G", float, 2e-4, "generator learning rate"),
    ("lr_D", float, 2e-4, "discriminator learning rate"),
    ("beta1", float, 0, "Adam beta1"),
    ("beta2", float, 0.999, "Adam beta2"),
    ("batch_size", int, 64, "discriminator batch size (real crops per step)"),
    ("loss", str, "standard", "standard (BCE, what the reference always uses) | hinge (build-side extra)"),
    ("disc_iters", int, 1, "discriminator updates per generator update"),
    ("epochs", int, 1, "number of epochs"),
    ("saving_rate", int, 30, "save checkpoints every N epochs"),
    ("ema", "flag", False, "keep an EMA copy of G"),
    ("ema_decay", float, 0.999, "EMA decay rate"),
    ("decay_lr", str, None, "learning-rate decay: exp | step"),
    ("seed", int, None, "fixed seed (None = random)"),
    ("smooth", "flag", False, "one-sided label smoothing (0.9)"),
    ("num_images", int, 8, "images generated per step"),
    ("num_patches_width", int, 3, "patches along the width of an image"),
    ("num_patches_height", int, 3, "patches along the height of an image"),
    ("outer_padding", str, "replicate", "outer padding of the patch grid: replicate | constant"),
    ("padding_size", int, 1, "padding size of local padding"),
    ("conv_reduction", int, 2, "spatial reduction of the convolution"),
    ("num_gpus", int, 1, "number of GPUs (ranks are launched with torch.distributed.run)"),
    ("dev_num", int, 0, "GPU index when a single GPU is used"),
    ("num_workers", int, 0, "data loader workers"),
    ("fname", str, "models_cp", "folder for checkpoints"),
    ("bf16", "flag", False, "build-side extra: run the convolutions on bf16-operand MFMA (fp32 tensors and "
                            "accumulation; BASELINE config 3)"),
    ("sync_bn", "flag", False, "build-side extra: all-reduce the BatchNorm statistics over the data-parallel ranks "
                               "(default: per-rank statistics, as the reference's nn.DataParallel)"),
    ("launch_mode", str, "auto", "build-side extra: eager | graph | auto.  graph = the whole iteration is recorded once into a "
                                 "hipGraph and replayed (host cost ~0; single rank, --disc_iters 1); auto = graph where that applies"),
    ("wgrad_reduce", str, "auto", "build-side extra: layer | deferred | auto.  deferred = one weight-gradient reduce launch per "
                                  "backward pass instead of 2-5 per layer (pays when the step is launch-bound: auto turns it on "
                                  "under graph replay for crops <= 128)"),
    ("shard_patch_rows", "flag", False, "build-side extra: multi-GPU runs shard the patch grid of ONE batch by patch "
                                        "rows (halo exchange) instead of replicating the batch per GPU"),
]


def prepare_parser():
    parser = argparse.ArgumentParser()
    for name, typ, default, hlp in _FLAGS:
        if typ == "flag":
            parser.add_argument("--" + name, action="store_true", default=default, help=hlp)
        else:
            parser.add_argument("--" + name, type=typ, default=default, help=hlp)
    parser.add_argument("--gpu_list", nargs="+", default=None, type=int, help="kept for CLI compatibility")
    return parser


def prepare_device(args):
    """One process drives one GPU.  LOCAL_RANK (torch.distributed.run) wins over --dev_num."""
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this build runs on MI355X only (there is no CPU path)")
    idx = int(os.environ.get("LOCAL_RANK", args.dev_num))
    print("Device: ", idx)
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def prepare_seed(args):
    seed = random.randint(1, 10000) if args.seed is None else args.seed
    print("Random Seed: ", seed)
    return seed


def prepare_data(args, device=None, seed=None):
    """-> (loader, dataset) as reference utils.py:158-191.  The loader yields ``{0: batch}`` dicts with the batch
    already on ``device`` (default: the current GPU); ``len(dataset)`` is ``--sampling``."""
    from . import data as D
    print(" laoding " + args.data + " ...")
    resize = None if args.resize_h is None and args.resize_w is None else (args.resize_h, args.resize_w)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if args.data == "single_image":
        train_data = D.single_image(path=args.data_path, ext=args.data_ext, sampling=args.sampling,
                                    random_crop=args.random_crop, center_crop=args.center_crop, device=device)
    elif args.data == "multiple_images":
        train_data = D.multiple_images(path=args.data_path, ext=args.data_ext, sampling=args.sampling,
                                       random_crop=args.random_crop, center_crop=args.center_crop, resize=resize,
                                       device=device)
    else:
        print("no data named :", args.data)
        raise SystemExit(1)
    loader = D.CropLoader(train_data, args.batch_size, seed=seed)
    print("Finished data loading")
    return loader, train_data


def prepare_models(args, device="cpu"):
    netG = ResidualPatchGenerator(
        z_dim=args.z_dim, G_ch=args.G_ch, base_res=args.base_res, n_layers_G=args.n_layers_G,
        attention=args.attention, img_ch=args.img_ch, leak=args.leak_G, SN=args.spec_norm_G,
        type_norm=args.type_norm_G, map_dim=args.map_dim, padding_mode=args.padding_mode,
        outer_padding=args.outer_padding, num_patches_h=args.num_patches_height,
        num_patches_w=args.num_patches_width, padding_size=args.padding_size,
        conv_reduction=args.conv_reduction).to(device)
    if args.D_model != "patch_GAN":
        raise NotImplementedError("only --D_model patch_GAN exists on the reference's own path (utils.py:205-207)")
    netD = PatchDiscriminator(img_ch=args.img_ch, base_ch=args.D_ch, n_layers_D=args.n_layers_D, kw=4,
                              SN=args.spec_norm_D, norm_layer=args.norm_layer_D).to(device)
    return netG, netD


def prepare_filename(args):
    filename = str(args.epochs) + "_"
    if args.fname is not None:
        os.makedirs(args.fname, exist_ok=True)
        filename = args.fname + "/" + filename
    return filename


def elapsed_time(start_time):
    return time.time() - start_time


def init_weight(m):
    """Kept for API compatibility; modules of this build initialise themselves (orthogonal
    conv weights, zero biases - reference utils.py:745-762)."""
    return m


# ------------------------------------------------------------------------------- merge / crop
def merge_patches_into_image(patches, num_rows=3, num_cols=3, device="cpu"):
    """(B, C, ph, pw) patches, image-major then row then column -> (B/(rows*cols), C, rows*ph, cols*pw).
    reference utils.py:577-613.  GPU tensors go through the HIP layout kernels."""
    if patches.is_cuda:
        return ops.to_nchw(ops.to_grid(patches, num_rows, num_cols, merged=False), merged=True)
    b, c, ph, pw = patches.shape
    n = b // (num_rows * num_cols)
    return patches.reshape(n, num_rows, num_cols, c, ph, pw).permute(0, 3, 1, 4, 2, 5).reshape(
        n, c, num_rows * ph, num_cols * pw)


def crop_images(img, cropping_size_h=256, cropping_size_w=256, stride=256, device="cpu"):
    """Sliding-window crops (window h x w, one stride for both axes), row-major per image, images
    outermost.  reference utils.py:658-742.  Pure data movement (host-side latents in the samplers)."""
    n, c, h, w = img.shape
    if h < cropping_size_h or w < cropping_size_w:
        return img.new_zeros((0,))
    u = img.unfold(2, cropping_size_h, stride).unfold(3, cropping_size_w, stride)
    nh, nw = u.shape[2], u.shape[3]
    return u.permute(0, 2, 3, 1, 4, 5).reshape(n * nh * nw, c, cropping_size_h, cropping_size_w).contiguous()


def crop_image(img, cropping_size_h=256, cropping_size_w=256, stride=256, device="cpu"):
    return crop_images(img.unsqueeze(0), cropping_size_h, cropping_size_w, stride, device)


# ------------------------------------------------------------------------------- latents
def _randn_on(device, *shape):
    """torch.randn on the CPU generator (the reference's RNG stream, utils.py:503-519), delivered to ``device``.  For a GPU
    the draw lands in page-locked memory and the copy is asynchronous: a pageable host -> device copy blocks the host
    until the stream has drained, which stops the host from issuing the next train step ahead of the GPU (train.py
    end to end: 745 -> see DESIGN.md section 7)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.randn(*shape).to(dev)
    return torch.randn(*shape, pin_memory=True).to(dev, non_blocking=True)


def build_z(num_images=1, z_dim=128, base_res=4, num_patches_height=3, num_patches_width=3,
            total_num_patches_height=3, total_num_patches_width=3, device="cpu"):
    """Full-grid latent cut into overlapping sub-image latents.  reference utils.py:221-234."""
    z_full = _randn_on(device, num_images, z_dim, total_num_patches_height * base_res + 2,
                       total_num_patches_width * base_res + 2)
    return crop_images(z_full, num_patches_height * base_res + 2, num_patches_width * base_res + 2,
                       (num_patches_width - 1) * base_res, device=device)


def build_maps(num_images=1, map_dim=1, n_layers_G=4, base_res=4, num_patches_height=3, num_patches_width=3,
               total_num_patches_height=3, total_num_patches_width=3, device="cpu"):
    """Per-layer full-grid SSM maps cut into overlapping sub-image maps.  reference utils.py:237-256."""
    out = []
    for i in range(n_layers_G):
        r = (2 ** i) * base_res
        full = _randn_on(device, num_images, map_dim, total_num_patches_height * r + 4, total_num_patches_width * r + 4)
        out.append(crop_images(full, num_patches_height * r + 4, num_patches_width * r + 4,
                               (num_patches_width - 1) * r, device=device))
    return out


def _unwrap(netG):
    return netG.module if hasattr(netG, "module") else netG


def sample_latents_train(netG, z_dim=128, base_res=4, map_dim=1, num_images=1, num_patches_height=3,
                         num_patches_width=3, device="cpu", merged_maps=False):
    """z then SSM maps 0..nl-1, drawn on the CPU generator and moved to ``device``
    (the reference's RNG order, utils.py:503-519).  ``merged_maps``: leave the maps un-cropped, (N, map_dim, nph*r + 4,
    npw*r + 4) - what the row-sharded trainer (engine.BandTrainer) slices its bands from."""
    g = _unwrap(netG)
    z = _randn_on(device, num_images, z_dim, num_patches_height * base_res + 2, num_patches_width * base_res + 2)
    maps = [None] * g.n_layers_G
    if g.type_norm == "SSM":
        maps = []
        for i in range(g.n_layers_G):
            r = (2 ** i) * base_res
            m = _randn_on(device, num_images, map_dim, num_patches_height * r + 4, num_patches_width * r + 4)
            maps.append(m if merged_maps else crop_images(m, r + 4, r + 4, r, device=device))
    return z, maps


def sample_latents_zeros(netG, z_dim=128, base_res=4, map_dim=1, num_images=1, device="cpu"):
    """Latents of the non-local baseline (padding_mode='zeros'): z (N, z_dim, b, b), then the SSM maps
    (N, map_dim, r, r) of layer 0..nl-1 (reference utils.py:556-566)."""
    g = _unwrap(netG)
    z = _randn_on(device, num_images, z_dim, base_res, base_res)
    maps = [None] * g.n_layers_G
    if g.type_norm == "SSM":
        maps = [_randn_on(device, num_images, map_dim, (2 ** i) * base_res, (2 ** i) * base_res)
                for i in range(g.n_layers_G)]
    return z, maps


def sample_from_gen_PatchByPatch_train(netG, z_dim=128, base_res=4, map_dim=1, num_images=1, num_patches_height=3,
                                       num_patches_width=3, device="cpu"):
    """Generate ``num_images`` images patch by patch (training).  reference utils.py:475-527."""
    z, maps = sample_latents_train(netG, z_dim, base_res, map_dim, num_images, num_patches_height,
                                   num_patches_width, device)
    g = netG.forward_grid(z, maps, "1st_row_1st_col") if hasattr(netG, "forward_grid") else None
    if g is None:
        return merge_patches_into_image(netG(z, maps, image_location="1st_row_1st_col"), num_patches_height,
                                        num_patches_width, device)
    return ops.to_nchw(g, merged=True)


def tile_process(img, model, scale=4, tile_size=32, tile_pad=8):
    """Real-ESRGAN style tiling of the latent ``img`` (reference utils.py:401-470): every tile_size^2 block is run
    through ``model`` with up to ``tile_pad`` latent pixels of context per side and its own scale x region of the
    result is pasted into the output."""
    n, _, h, w = img.shape
    out = img.new_zeros((n, 3, h * scale, w * scale))
    with torch.no_grad():
        for y0 in range(0, h, tile_size):
            for x0 in range(0, w, tile_size):
                y1, x1 = min(y0 + tile_size, h), min(x0 + tile_size, w)
                ya, xa = max(y0 - tile_pad, 0), max(x0 - tile_pad, 0)
                yb, xb = min(y1 + tile_pad, h), min(x1 + tile_pad, w)
                t = model(img[:, :, ya:yb, xa:xb].contiguous())
                oy, ox = (y0 - ya) * scale, (x0 - xa) * scale
                out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = \
                    t[:, :, oy:oy + (y1 - y0) * scale, ox:ox + (x1 - x0) * scale]
    return out


def sample_from_gen(netG, z_dim=128, base_res=4, map_dim=1, num_images=1, tiles=False, device="cpu", z=None):
    """Non-local baseline sampler (padding_mode='zeros'), reference utils.py:530-575: one zero-padded forward over
    the whole latent, or - ``tiles`` - tile_process(z, netG, 2**(nl-1), 32, 16).  ``z`` injects the latent."""
    g = _unwrap(netG)
    if z is None:
        z = torch.randn(num_images, z_dim, base_res, base_res)
    z = z.to(device)
    maps = [None] * g.n_layers_G
    if g.type_norm == "SSM":
        maps = [torch.randn(num_images, map_dim, (2 ** i) * base_res, (2 ** i) * base_res).to(device)
                for i in range(g.n_layers_G)]
    if tiles:
        return tile_process(z, netG, 2 ** (g.n_layers_G - 1), 32, 16)
    return netG(z, maps, image_location="1st_row_1st_col")


# ------------------------------------------------------------------------------- inference tiling
def _location(ih, iw, steps_h, steps_w):
    """image_location strings of reference utils.py:321-337."""
    if steps_h == 1:
        s = "1st_row_last_row"
    elif ih == 0:
        s = "1st_row"
    elif ih == steps_h - 1:
        s = "last_row"
    else:
        s = "inter_row"
    if steps_w == 1:
        return s + "_1st_col_last_col"
    if iw == 0:
        return s + "_1st_col"
    if iw == steps_w - 1:
        return s + "_last_col"
    return s + "_inter_col"


def tiling_plan(n_layers_G, base_res, num_patches_height, num_patches_width, out_h, out_w):
    """(steps_h, steps_w, T_h, T_w, P) of reference utils.py:294-303."""
    p = (2 ** (n_layers_G - 1)) * base_res
    steps_h = math.ceil((out_h / p - 1) / (num_patches_height - 1))
    steps_w = math.ceil((out_w / p - 1) / (num_patches_width - 1))
    if steps_h < 1 or steps_w < 1:
        raise ValueError("output %dx%d must exceed one generator patch (%d px)" % (out_h, out_w, p))
    return steps_h, steps_w, steps_h * (num_patches_height - 1) + 1, steps_w * (num_patches_width - 1) + 1, p


def sample_from_gen_PatchByPatch_test(netG, z_dim=128, base_res=4, map_dim=1, num_images=1, num_patches_height=3,
                                      num_patches_width=3, device="cpu", output_resolution_height=384,
                                      output_resolution_width=384, z_full=None, maps_full=None,
                                      one_shot=None, halo=None, strip_on_device=False):
    """Generate one large image (reference utils.py:258-397).

    Default (``one_shot=None``): generators without attention run ONE forward over the whole
    T_h x T_w patch grid - bit-equivalent to the reference's streamed schedule and
    free of its 2.1x recomputation; generators with attention stream 3x3 sub-images in raster order
    with carried halos exactly as the reference does.  ``one_shot=False`` forces streaming.
    ``z_full`` / ``maps_full`` inject pre-built full-grid latents (tests); otherwise they are drawn
    in the reference's RNG order: z first, then maps 0..nl-1.
    """
    g = _unwrap(netG)
    nph, npw = num_patches_height, num_patches_width
    steps_h, steps_w, t_h, t_w, p = tiling_plan(g.n_layers_G, base_res, nph, npw, output_resolution_height,
                                                output_resolution_width)
    ssm = g.type_norm == "SSM"
    if one_shot is None:
        one_shot = not g.attention
    if z_full is None:
        # page-locked for the one-shot forward (its up to 72 MB copies then run at PCIe rate); the streamed schedule cuts
        # the latents into sub-image windows on the CPU first, which is slower on page-locked memory (SSM 1024^2: 1.2 vs 0.6 s)
        pin = torch.device(device).type == "cuda" and one_shot and halo is None
        z_full = torch.randn(num_images, z_dim, t_h * base_res + 2, t_w * base_res + 2, pin_memory=pin)
        maps_full = [torch.randn(num_images, map_dim, t_h * (2 ** i) * base_res + 4, t_w * (2 ** i) * base_res + 4,
                                 pin_memory=pin) for i in range(g.n_layers_G)] if ssm else None
    with torch.no_grad():
        if halo is not None:
            # patch grid sharded by patch rows over halo.world ranks; returns THIS rank's strip of rows
            if g.attention:
                raise ValueError("row-sharded generation needs a generator without attention (SURVEY.md F7)")
            return _generate_row_sharded(g, z_full, maps_full, t_h, t_w, base_res, device, halo, p,
                                         output_resolution_height, output_resolution_width, on_device=strip_on_device)
        if one_shot:
            return _generate_one_shot(g, z_full, maps_full, t_h, t_w, base_res, device)[
                :, :, :output_resolution_height, :output_resolution_width]
        return _generate_streamed(g, z_full, maps_full, steps_h, steps_w, p, base_res, nph, npw, device)[
            :, :, :output_resolution_height, :output_resolution_width]


def _to_host(t):
    """Device image -> CPU tensor (what the reference's sampler returns, utils.py:396) through page-locked memory: the
    4224^2 x 3 image of a 4096^2 request is 214 MB, ~50 ms as a pageable copy against ~9 ms at PCIe rate; torch's host
    allocator keeps the page-locked block for the next image."""
    if not t.is_cuda:
        return t
    out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    out.copy_(t)
    return out


def _generate_one_shot(g, z_full, maps_full, t_h, t_w, base_res, device):
    saved = (g.num_patches_h, g.num_patches_w)
    from .models.layers import LocalPadder
    pads = [m for m in g.modules() if isinstance(m, LocalPadder)]
    try:
        g.num_patches_h, g.num_patches_w = t_h, t_w
        for m in pads:
            m.pin(t_h, t_w, g.outer_padding)
            m.reset_state()
        maps = None
        if maps_full is not None:
            maps = [crop_images(maps_full[i].to(device), (2 ** i) * base_res + 4, (2 ** i) * base_res + 4,
                                (2 ** i) * base_res) for i in range(g.n_layers_G)]
        out = g.forward_grid(z_full.to(device), maps, "1st_row_1st_col_last_row_last_col")
        return _to_host(ops.to_nchw(out, merged=True))
    finally:
        g.num_patches_h, g.num_patches_w = saved
        for m in pads:
            m.pin(saved[0], saved[1], g.outer_padding)
            m.reset_state()


def band_latents(g, t_h, t_w, base_res, halo, device, z_dim=128, map_dim=1, seed=0, z_full=None, maps_full=None):
    """The latents of THIS rank's band of patch rows: z (1, z_dim, rows*b + 2, T_w*b + 2) and, for SSM, one map per layer
    (1, map_dim, rows*r + 4, T_w*r + 4), on ``device``.

    With ``z_full`` / ``maps_full`` (the reference's full-grid CPU draw, utils.py:221-256: z first, then maps 0..nl-1 from
    the global generator) the band is cut out of them - seed-identical images, but every rank draws the whole grid
    (18 M values at the finest level of a 4096^2 image: ~55 ms that do not shrink with the rank count).  Without them each
    rank draws only its own rows, on the device, from generators seeded per (seed, layer, patch row): the rows two
    neighbouring bands share (2 of z, 4 of every map) come out identical on both ranks, so the sharded image is the one a
    single rank would have drawn with the same seed."""
    a, b = halo.band(t_h)
    if z_full is not None:
        z = z_full[:, :, a * base_res:b * base_res + 2, :].to(device)
        maps = None
        if maps_full is not None:
            maps = [maps_full[i][:, :, a * (2 ** i) * base_res:b * (2 ** i) * base_res + 4, :].to(device)
                    for i in range(g.n_layers_G)]
        return z, maps

    def rows(layer, ch, r, extra, width):
        gen = torch.Generator(device=device)
        parts = []
        for k in range(a, b + 1):                      # block k = the r latent rows of patch row k; block b only lends its first rows
            gen.manual_seed((int(seed) * 1000003 + layer * 10007 + k) & 0x7fffffffffffffff)
            blk = torch.randn(1, ch, r if k < t_h else extra, width, device=device, generator=gen)
            parts.append(blk if k < b else blk[:, :, :extra])
        return torch.cat(parts, 2)

    z = rows(0, z_dim, base_res, 2, t_w * base_res + 2)
    maps = None
    if g.type_norm == "SSM":
        maps = [rows(1 + i, map_dim, (2 ** i) * base_res, 4, t_w * (2 ** i) * base_res + 4) for i in range(g.n_layers_G)]
    return z, maps


def generate_band(g, z_loc, maps_loc, t_w, base_res, halo):
    """One forward of this rank's band (latents from band_latents, on the device) -> the band's image (1, c, rows*P, T_w*P)
    on the device; every 3x3 conv exchanges one pixel row with each neighbour rank."""
    from .models.layers import LocalPadder
    rows = (z_loc.shape[2] - 2) // base_res
    pads = [m for m in g.modules() if isinstance(m, LocalPadder)]
    saved = (g.num_patches_h, g.num_patches_w)
    try:
        g.num_patches_h, g.num_patches_w = rows, t_w
        for m in pads:
            m.pin(rows, t_w, g.outer_padding)
            m.reset_state()
            m.halo = halo if m.merge_patches_into_image else None
        maps = None
        if maps_loc is not None:
            maps = [crop_images(maps_loc[i], (2 ** i) * base_res + 4, (2 ** i) * base_res + 4, (2 ** i) * base_res)
                    for i in range(g.n_layers_G)]
        with torch.no_grad():
            return ops.to_nchw(g.forward_grid(z_loc, maps, "1st_row_1st_col"), merged=True)
    finally:
        g.num_patches_h, g.num_patches_w = saved
        for m in pads:
            m.pin(saved[0], saved[1], g.outer_padding)
            m.reset_state()
            m.halo = None


def _generate_row_sharded(g, z_full, maps_full, t_h, t_w, base_res, device, halo, p, out_h, out_w, on_device=False):
    """One-shot generation with the T_h x T_w patch grid split by patch rows over the ranks of ``halo``
    (dist.RowHalo): every 3x3 conv exchanges one pixel row with each neighbour.  Latents are the SAME
    full-grid tensors on every rank (same seed); each rank cuts out its band (+ the latent's own halo).
    ``on_device``: the strip stays in HBM (gather_strips collects the strips over the process group)."""
    a, b = halo.band(t_h)
    z_loc, maps_loc = band_latents(g, t_h, t_w, base_res, halo, device, z_full=z_full, maps_full=maps_full)
    strip = generate_band(g, z_loc, maps_loc, t_w, base_res, halo)
    if not on_device:
        strip = _to_host(strip)
    lo, hi = a * p, min(b * p, out_h)
    return strip[:, :, :max(0, hi - lo), :out_w]


def strip_rows(halo, t_h, p, out_h):
    """[lo, hi) output pixel rows of every rank's strip (static: band() x patch height, cropped to the requested height)."""
    out = []
    for r in range(halo.world):
        a, b = type(halo).band(_RankView(r, halo.world), t_h)
        out.append((min(a * p, out_h), min(b * p, out_h)))
    return out


class _RankView:
    """band() of another rank (RowHalo.band only reads rank / world)."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world


def gather_strips(strip, halo, rows, dst=0):
    """The strips (1, c, h_r, W) of a row-sharded generation -> the whole image on rank ``dst`` (None elsewhere), moved as
    tensors into ONE preallocated image: rank r sends its strip, ``dst`` receives every strip into a staging buffer and
    copies it into its rows (strips are ragged, so this is point-to-point, not dist.gather; round 3 pickled the strips
    with gather_object).  Device strips travel over RCCL; under gloo (several ranks on one GPU, rehearsal) through the host."""
    import torch.distributed as dist
    if halo.world == 1:
        return strip
    via_host = strip.is_cuda and dist.get_backend(halo.group) == "gloo"
    mine = strip.contiguous().cpu() if via_host else strip.contiguous()
    if halo.rank != dst:
        if mine.shape[-2] > 0:
            dist.send(mine, dst, group=halo.group)
        return None
    n, c, _, w = strip.shape
    img = torch.empty((n, c, rows[-1][1], w), dtype=strip.dtype, device=strip.device)
    for r, (lo, hi) in enumerate(rows):
        if hi <= lo:
            continue
        if r == dst:
            img[:, :, lo:hi] = strip
            continue
        buf = torch.empty((n, c, hi - lo, w), dtype=strip.dtype, device=mine.device)
        dist.recv(buf, r, group=halo.group)
        img[:, :, lo:hi] = buf.to(strip.device)
    return img


def _generate_streamed(g, z_full, maps_full, steps_h, steps_w, p, base_res, nph, npw, device):
    g.reset_stream_state()
    z_sub = crop_images(z_full, nph * base_res + 2, npw * base_res + 2, (npw - 1) * base_res)
    m_sub = None
    if maps_full is not None:
        m_sub = [crop_images(maps_full[i], nph * (2 ** i) * base_res + 4, npw * (2 ** i) * base_res + 4,
                             (npw - 1) * (2 ** i) * base_res) for i in range(g.n_layers_G)]
    rows, k = [], 0
    for ih in range(steps_h):
        row = []
        for iw in range(steps_w):
            loc = _location(ih, iw, steps_h, steps_w)
            maps = None
            if m_sub is not None:
                maps = [crop_images(m_sub[i][[k]].to(device), (2 ** i) * base_res + 4, (2 ** i) * base_res + 4,
                                    (2 ** i) * base_res) for i in range(g.n_layers_G)]
            img = _to_host(ops.to_nchw(g.forward_grid(z_sub[[k]].to(device), maps, loc), merged=True))
            hh = img.shape[-2] if ih == steps_h - 1 else p * (nph - 1)
            ww = img.shape[-1] if iw == steps_w - 1 else p * (npw - 1)
            row.append(img[:, :, :hh, :ww])
            k += 1
        rows.append(torch.cat(row, -1))
    g.reset_stream_state()
    return torch.cat(rows, -2)
