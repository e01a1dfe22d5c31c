"""Shared test helpers: golden fixture loading, flag parsing, error metrics."""
import os

import numpy as np
import torch

from oracle.nets import GCfg, DCfg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_BASE = {"G_ch": 4, "D_ch": 4, "z_dim": 8, "leak_G": 0.02, "batch_size": 2, "num_images": 2,
         "n_layers_G": 6, "n_layers_D": 4, "type_norm": "BN", "outer_padding": "replicate",
         "base_res": 4, "num_patches_height": 3, "num_patches_width": 3, "map_dim": 1,
         "attention": False, "spec_norm_D": False, "smooth": False, "random_crop": None,
         "padding_mode": "local", "disc_iters": 1, "ema": False, "ema_decay": 0.999}


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def parse_flags(argv):
    """Flags recorded in a fixture (the same spellings the reference CLI takes)."""
    a = dict(_BASE)
    argv = [str(x) for x in argv]
    i = 0
    while i < len(argv):
        k = argv[i].lstrip("-")
        if k in ("attention", "spec_norm_D", "smooth", "ema"):
            a[k] = True
            i += 1
            continue
        v = argv[i + 1]
        a[k] = type(_BASE[k])(v) if _BASE.get(k) is not None else int(v)
        i += 2
    return a


def cfgs(a):
    g = GCfg(z_dim=a["z_dim"], G_ch=a["G_ch"], base_res=a["base_res"], n_layers_G=a["n_layers_G"],
             attention=a["attention"], img_ch=3, leak=a["leak_G"], SN=False, type_norm=a["type_norm"],
             map_dim=a["map_dim"], padding_mode=a["padding_mode"], outer_padding=a["outer_padding"],
             num_patches_h=a["num_patches_height"], num_patches_w=a["num_patches_width"])
    d = DCfg(img_ch=3, base_ch=a["D_ch"], n_layers_D=a["n_layers_D"], SN=a["spec_norm_D"])
    return g, d


def state(fix, prefix):
    out = {}
    for k, v in fix.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.array(v))
    return out


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def crop_maps(gcfg, maps_full):
    """Full per-layer SSM maps (N,m,gh*r+4,gw*r+4) -> per-patch (N*gh*gw,m,r+4,r+4)."""
    from oracle import patches as P
    out = []
    for i, m in enumerate(maps_full):
        r = (2 ** i) * gcfg.base_res
        out.append(P.crop(m, r + 4, r + 4, r))
    return out
