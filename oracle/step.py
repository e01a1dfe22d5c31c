"""Oracle: latent samplers, the G+D train step, Adam, EMA and the inference tiler
(test infrastructure).

Restates:
  * sample_from_gen_PatchByPatch_train        reference utils.py:475-527
  * build_z / build_maps                      reference utils.py:221-256
  * sample_from_gen_PatchByPatch_test         reference utils.py:258-397
  * the train iteration                       reference train.py:122-180 (incl. --disc_iters > 1, --ema)
  * sample_from_gen / tile_process            reference utils.py:530-575, 401-470 (padding_mode='zeros' baseline)
  * torch.optim.Adam (lr, betas, eps 1e-8, no weight decay, bias correction) as the
    reference configures it at train.py:57-58
  * hinge loss: NOT in the reference (utils.py:85 flag is never read) - parity unpinned.
"""
import math

import torch
import torch.nn.functional as F

from . import patches as P
from .nets import g_forward, d_forward


# --------------------------------------------------------------------------- latents
def sample_latents(cfg, num_images, generator=None):
    """z first, then SSM maps layer 0..nl-1, all from the CPU generator (utils.py:503-519)."""
    gh, gw, b = cfg.num_patches_h, cfg.num_patches_w, cfg.base_res
    z = torch.randn(num_images, cfg.z_dim, gh * b + 2, gw * b + 2, generator=generator)
    maps = None
    if cfg.type_norm == "SSM":
        maps = []
        for i in range(cfg.n_layers_G):
            r = (2 ** i) * b
            m = torch.randn(num_images, cfg.map_dim, gh * r + 4, gw * r + 4, generator=generator)
            maps.append(P.crop(m, r + 4, r + 4, r))
    return z, maps


def g_sample_train(sd, cfg, z, maps, loops=False):
    """G forward on given latents + merge -> (N, C, gh*P, gw*P).  utils.py:523-527.
    padding_mode='zeros': z is (N, z_dim, b, b) and G's output already is the image (utils.py:556-573)."""
    patches = g_forward(sd, cfg, z, maps, training=True, loops=loops)
    if cfg.padding_mode != "local":
        return patches
    mg = P.merge_loops if loops else P.merge
    return mg(patches, cfg.num_patches_h, cfg.num_patches_w)


# --------------------------------------------------------------------------- losses
def bce_logits(logit, target):
    """nn.BCEWithLogitsLoss (mean) against a constant target.  train.py:81,131-132."""
    return F.binary_cross_entropy_with_logits(logit, torch.full_like(logit, float(target)))


def hinge_d(real_logit, fake_logit):
    """Hinge D loss (parity unpinned: absent from the reference)."""
    return F.relu(1.0 - real_logit).mean(), F.relu(1.0 + fake_logit).mean()


def hinge_g(fake_logit):
    return -fake_logit.mean()


# --------------------------------------------------------------------------- Adam
class Adam:
    """Plain restatement of torch.optim.Adam(params, lr, betas, eps=1e-8), single-tensor
    form: m=b1*m+(1-b1)g ; v=b2*v+(1-b2)g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t)+eps)."""

    def __init__(self, params, lr=2e-4, betas=(0.0, 0.999), eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, float(betas[0]), float(betas[1]), eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.lerp_(g, 1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


def trainable(sd):
    """Names of the nn.Parameters inside a reference state_dict (buffers excluded)."""
    skip = ("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v")
    return [k for k in sd if not k.endswith(skip)]


def as_leaf_params(sd):
    for k in trainable(sd):
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    return sd


def zero_grads(sd):
    for k in trainable(sd):
        sd[k].grad = None


# --------------------------------------------------------------------------- train step
def d_step(gsd, dsd, gcfg, dcfg, optD, real_x, z, maps, label_t, loops=False):
    """One discriminator update, reference train.py:126-153.  Returns (d_real, d_fake, fake, logits)."""
    zero_grads(dsd)
    real_logit = d_forward(dsd, dcfg, real_x, training=True)
    d_real = bce_logits(real_logit, label_t)
    d_real.backward()
    fake = g_sample_train(gsd, gcfg, z, maps, loops=loops)
    fake_logit = d_forward(dsd, dcfg, fake.detach(), training=True)
    d_fake = bce_logits(fake_logit, 0.0)
    d_fake.backward()
    optD.step()
    return d_real, d_fake, fake, real_logit, fake_logit


def g_step(gsd, dsd, dcfg, optG, fake, label_t):
    """The generator update on the LAST fake batch of the D loop, reference train.py:161-169."""
    zero_grads(gsd)
    fake_logit2 = d_forward(dsd, dcfg, fake, training=True)
    g_loss = bce_logits(fake_logit2, label_t)
    g_loss.backward()
    optG.step()
    return g_loss, fake_logit2


def train_step(gsd, dsd, gcfg, dcfg, optG, optD, real_x, z, maps, smooth=True, loops=False,
               ema_sd=None, ema_decay=0.999):
    """One iteration of reference train.py:122-180 with injected real_x / latents.  ``z`` / ``maps`` may be
    lists of equal length: --disc_iters = len(z) discriminator updates (fresh latents each, the same real_x),
    then ONE generator update on the last fake batch.  Mutates the state dicts (params, BN buffers, SN u/v) in
    place.  Returns dict(d_loss_real, d_loss_fake, g_loss, fake, ...) of the last D iteration (+ d_losses: all)."""
    label_t = 0.9 if smooth else 1.0
    zs = z if isinstance(z, (list, tuple)) else [z]
    ms = maps if isinstance(z, (list, tuple)) else [maps]
    d_losses = []
    for zi, mi in zip(zs, ms):
        d_real, d_fake, fake, real_logit, fake_logit = d_step(gsd, dsd, gcfg, dcfg, optD, real_x, zi, mi, label_t, loops)
        d_losses += [float(d_real.detach()), float(d_fake.detach())]
    # D's gradients of the D step as Adam(D) consumed them (the G step's backward accumulates into the same .grad
    # afterwards, as in the reference, train.py:161-166)
    grad_d = {k: dsd[k].grad.detach().clone() for k in trainable(dsd) if dsd[k].grad is not None}
    g_loss, fake_logit2 = g_step(gsd, dsd, dcfg, optG, fake, label_t)
    if ema_sd is not None:
        ema_update(ema_sd, gsd, ema_decay)
    return dict(d_loss_real=float(d_real.detach()), d_loss_fake=float(d_fake.detach()), g_loss=float(g_loss.detach()),
                fake=fake.detach(), real_logit=real_logit.detach(), fake_logit=fake_logit.detach(),
                fake_logit2=fake_logit2.detach(), d_losses=d_losses, gradD=grad_d)


@torch.no_grad()
def ema_update(ema_sd, gsd, decay):
    """ema = decay*ema + (1-decay)*g over every state_dict entry, int64 counters included
    (float arithmetic, truncated on copy).  reference train.py:176-180."""
    for k in gsd:
        ema_sd[k].copy_(ema_sd[k] * decay + gsd[k].detach() * (1 - decay))


# --------------------------------------------------------------------------- inference tiling
def grid_size(out_h, out_w, cfg):
    """(steps_h, steps_w, T_h, T_w, P).  reference utils.py:294-303."""
    p = (2 ** (cfg.n_layers_G - 1)) * cfg.base_res
    sh = math.ceil((out_h / p - 1) / (cfg.num_patches_h - 1))
    sw = math.ceil((out_w / p - 1) / (cfg.num_patches_w - 1))
    return sh, sw, sh * (cfg.num_patches_h - 1) + 1, sw * (cfg.num_patches_w - 1) + 1, p


def full_latents(cfg, t_h, t_w, generator=None):
    """Full-grid latents in the reference's RNG order: z, then maps 0..nl-1 (utils.py:228,246)."""
    b = cfg.base_res
    z = torch.randn(1, cfg.z_dim, t_h * b + 2, t_w * b + 2, generator=generator)
    maps = None
    if cfg.type_norm == "SSM":
        maps = [torch.randn(1, cfg.map_dim, t_h * (2 ** i) * b + 4, t_w * (2 ** i) * b + 4, generator=generator)
                for i in range(cfg.n_layers_G)]
    return z, maps


def _loc(ih, iw, sh, sw):
    """image_location string.  reference utils.py:321-337."""
    if sh == 1:
        s = "1st_row_last_row"
    elif ih == 0:
        s = "1st_row"
    elif ih == sh - 1:
        s = "last_row"
    else:
        s = "inter_row"
    if sw == 1:
        return s + "_1st_col_last_col"
    if iw == 0:
        return s + "_1st_col"
    if iw == sw - 1:
        return s + "_last_col"
    return s + "_inter_col"


@torch.no_grad()
def infer_streamed(sd, cfg, z_full, maps_full, out_h, out_w):
    """Raster-streamed generation with carried halos.  reference utils.py:306-395 (num_images=1)."""
    gh, gw, b = cfg.num_patches_h, cfg.num_patches_w, cfg.base_res
    sh, sw, t_h, t_w, p = grid_size(out_h, out_w, cfg)
    z_sub = P.crop(z_full, gh * b + 2, gw * b + 2, (gw - 1) * b)
    m_sub = None
    if cfg.type_norm == "SSM":
        m_sub = [P.crop(maps_full[i], gh * (2 ** i) * b + 4, gw * (2 ** i) * b + 4, (gw - 1) * (2 ** i) * b)
                 for i in range(cfg.n_layers_G)]
    padders = {}
    rows = []
    k = 0
    for ih in range(sh):
        row = []
        for iw in range(sw):
            loc = _loc(ih, iw, sh, sw)
            maps = None
            if m_sub is not None:
                maps = [P.crop(m_sub[i][[k]], (2 ** i) * b + 4, (2 ** i) * b + 4, (2 ** i) * b)
                        for i in range(cfg.n_layers_G)]
            patches = g_forward(sd, cfg, z_sub[[k]], maps, training=False, loc=loc, padders=padders)
            img = P.merge(patches, gh, gw)
            hh = img.shape[-2] if ih == sh - 1 else p * (gh - 1)
            ww = img.shape[-1] if iw == sw - 1 else p * (gw - 1)
            row.append(img[:, :, :hh, :ww])
            k += 1
        rows.append(torch.cat(row, -1))
    return torch.cat(rows, -2)[:, :, :out_h, :out_w]


@torch.no_grad()
def infer_oneshot(sd, cfg, z_full, maps_full, out_h, out_w):
    """One G forward over the whole T_h x T_w grid (equal to the streamed result without
    attention: SURVEY.md F7)."""
    import copy
    sh, sw, t_h, t_w, p = grid_size(out_h, out_w, cfg)
    big = copy.copy(cfg)
    big.num_patches_h, big.num_patches_w = t_h, t_w
    maps = None
    if cfg.type_norm == "SSM":
        b = cfg.base_res
        maps = [P.crop(maps_full[i], (2 ** i) * b + 4, (2 ** i) * b + 4, (2 ** i) * b)
                for i in range(cfg.n_layers_G)]
    # eval-mode BN (running stats) with the training-branch padding over the whole grid
    from . import nets
    ctx_padders = {}
    patches = nets.g_forward(sd, big, z_full, maps, training=False, loc="1st_row_1st_col_last_row_last_col",
                             padders=ctx_padders)
    return P.merge(patches, t_h, t_w)[:, :, :out_h, :out_w]


# --------------------------------------------------------------------------- non-local baseline sampler
@torch.no_grad()
def tile_process(z, model, scale, tile_size=32, tile_pad=8):
    """Real-ESRGAN style tiling of the latent: every tile_size x tile_size block of ``z`` is run through
    ``model`` together with up to ``tile_pad`` latent pixels of context per side, and the block's own
    scale x region of the result is pasted into the output.  reference utils.py:401-470."""
    n, _, h, w = z.shape
    out = z.new_zeros((n, 3, h * scale, w * scale))
    for y0 in range(0, h, tile_size):
        for x0 in range(0, w, tile_size):
            y1, x1 = min(y0 + tile_size, h), min(x0 + tile_size, w)
            ya, xa = max(y0 - tile_pad, 0), max(x0 - tile_pad, 0)
            yb, xb = min(y1 + tile_pad, h), min(x1 + tile_pad, w)
            t = model(z[:, :, ya:yb, xa:xb])
            oy, ox = (y0 - ya) * scale, (x0 - xa) * scale
            out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = \
                t[:, :, oy:oy + (y1 - y0) * scale, ox:ox + (x1 - x0) * scale]
    return out


@torch.no_grad()
def sample_zeros(sd, cfg, z, maps=None, tiles=False, training=False):
    """sample_from_gen on an injected latent (N, z_dim, b, b), reference utils.py:530-575: the whole image in
    one zero-padded forward, or - ``tiles`` - tile_process(z, G, 2**(nl-1), 32, 16) (utils.py:568-570)."""
    if tiles:
        return tile_process(z, lambda t: g_forward(sd, cfg, t, None, training=training), 2 ** (cfg.n_layers_G - 1), 32, 16)
    return g_forward(sd, cfg, z, maps, training=training)
