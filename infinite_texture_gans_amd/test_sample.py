"""Inference CLI with the reference's flags (reference test_sample.py): load a checkpoint written by
either this build or the reference (optional DataParallel 'module.' prefix), generate one image of
arbitrary size patch by patch, write it with PIL."""
import argparse
import os
from collections import OrderedDict

import numpy as np
import torch

from . import utils as U
from .models import generators


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--output_resolution_height', type=int, default=384, help='output_resolution_height')
    p.add_argument('--output_resolution_width', type=int, default=384, help='output_resolution_width')
    p.add_argument('--output_name', type=str, default='241_generated.jpg', help='name of the generated image')
    p.add_argument('--model_path', type=str, default='results/241_lp_bn_outerpadRepl/300__ema.pth',
                   help='path of the generator network')
    p.add_argument('--tiles', default=False, action='store_true', help='use tiling of the input')
    return p


def load_G(state_dict_G, netG):
    sd = OrderedDict((k.replace('module.', ''), v) for k, v in state_dict_G.items())
    netG.load_state_dict(sd)
    return netG.eval()


def save_image(img, path):
    """img: (1, C, H, W) in [0, 1]."""
    from PIL import Image
    a = (img[0].clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(path)


def main(argv=None):
    a = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this build runs on MI355X only")
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    halo = None
    folder = os.path.dirname(a.model_path)
    ckpt = torch.load(a.model_path, map_location='cpu', weights_only=False)   # args is a pickled Namespace
    args = ckpt['args']
    # N ranks (torch.distributed.run --nproc-per-node N test_sample.py ...):
    #  * generators without attention: ONE image, its patch grid sharded by patch rows, one band per GPU, with a halo-row
    #    exchange (RCCL send/recv) before every 3x3 conv;
    #  * attention checkpoints (and padding_mode != local) stream their sub-images with carried state - a sequential
    #    dependency that does not shard (SURVEY.md 8e "replicas only"): REPLICA mode, rank r generates its own image from
    #    seed + r into <output_name stem>_rank<r><ext>, no collective at all.
    replicas = world > 1 and (bool(args.attention) or args.padding_mode != 'local')
    if world > 1 and not replicas:
        import torch.distributed as dist
        from .dist import RowHalo
        backend = os.environ.get("ITG_DIST_BACKEND", "nccl")      # gloo: several ranks on one GPU (rehearsal)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        halo = RowHalo(rank, world, dist.group.WORLD)
    netG = generators.ResidualPatchGenerator(
        z_dim=args.z_dim, G_ch=args.G_ch, base_res=args.base_res, n_layers_G=args.n_layers_G,
        attention=args.attention, img_ch=args.img_ch, leak=args.leak_G, SN=False, type_norm=args.type_norm_G,
        map_dim=1, padding_mode=args.padding_mode, outer_padding=args.outer_padding, num_patches_h=3,
        num_patches_w=3, padding_size=1, conv_reduction=2)
    netG = load_G(ckpt['netG_state_dict'], netG).to(device)
    if rank == 0:
        print(args)
    if world == 1 and "ITG_SAMPLE_SEED" in os.environ:
        torch.manual_seed(int(os.environ["ITG_SAMPLE_SEED"]))      # reproducible single-process images (tests)
    if replicas:
        base = int(os.environ.get("ITG_SAMPLE_SEED", 1234))
        torch.manual_seed(base + rank)                   # a different image per rank
        stem, ext = os.path.splitext(a.output_name)
        a.output_name = "%s_rank%d%s" % (stem, rank, ext)
    with torch.no_grad():
        if args.padding_mode == 'local' and halo is not None:
            seed = int(os.environ.get("ITG_SAMPLE_SEED", torch.seed() if world == 1 else 1234))
            torch.manual_seed(seed)                      # every rank draws the SAME full-grid latents
            strip = U.sample_from_gen_PatchByPatch_test(
                netG, z_dim=args.z_dim, num_images=1, output_resolution_height=a.output_resolution_height,
                output_resolution_width=a.output_resolution_width, device=device, halo=halo, strip_on_device=True)
            g_ = U._unwrap(netG)
            _, _, t_h, _, p_ = U.tiling_plan(g_.n_layers_G, args.base_res, 3, 3, a.output_resolution_height,
                                             a.output_resolution_width)
            img = U.gather_strips(strip, halo, U.strip_rows(halo, t_h, p_, a.output_resolution_height))
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
            if rank != 0:
                return
            img = img.cpu()
        elif args.padding_mode == 'local':
            img = U.sample_from_gen_PatchByPatch_test(
                netG, z_dim=args.z_dim, num_images=1, output_resolution_height=a.output_resolution_height,
                output_resolution_width=a.output_resolution_width, device=device).cpu()
        else:
            scale = 2 ** (netG.n_layers_G - 1)
            img = U.sample_from_gen(netG, z_dim=args.z_dim, base_res=a.output_resolution_height // scale,
                                    num_images=1, tiles=a.tiles, device=device).cpu()
    path = os.path.join(folder, a.output_name)
    print('The image is saved as:', path)
    save_image(img * 0.5 + 0.5, path)


if __name__ == '__main__':
    main()
