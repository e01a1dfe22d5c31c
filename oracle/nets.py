"""Oracle: generator / discriminator forward passes as functions over a state_dict
(test infrastructure).

The arithmetic is delegated to the same torch CPU primitives the reference calls
(conv2d, batch_norm, leaky_relu, interpolate, bmm, softmax, max_pool2d); the wiring
restates:
  * conv2d_lp                     reference models/layers.py:8-36
  * StochasticSpatialModulation   reference models/layers.py:203-234
  * Attention                     reference models/layers.py:236-258
  * ResBlockGenerator             reference models/layers.py:260-322
  * ResidualPatchGenerator        reference models/generators.py:25-124
  * PatchDiscriminator            reference models/discriminators.py:171-210
  * legacy torch.nn.utils.spectral_norm hook (layers.py:178-200; torch semantics:
    dim 0, one power iteration per training-mode forward, eps 1e-12)
state_dict keys are those of the reference modules.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import patches as P


@dataclass
class GCfg:
    z_dim: int = 128
    G_ch: int = 64
    base_res: int = 4
    n_layers_G: int = 4
    attention: bool = True
    img_ch: int = 3
    leak: float = 0.0
    SN: bool = False
    type_norm: str = "BN"
    map_dim: int = 1
    padding_mode: str = "local"
    outer_padding: str = "replicate"
    num_patches_h: int = 3
    num_patches_w: int = 3


@dataclass
class DCfg:
    img_ch: int = 3
    base_ch: int = 64
    n_layers_D: int = 4
    SN: bool = False


# --------------------------------------------------------------------------- spectral norm
def sn_weight(sd, prefix, training):
    """Effective weight of a (possibly) spectrally normalised conv.

    With SN the state_dict holds ``weight_orig``, ``weight_u``, ``weight_v``; a
    training-mode forward runs one power iteration *in place* on u, v (no grad),
    then weight = weight_orig / (u^T W v).  Without SN returns ``weight``.
    """
    if prefix + ".weight_orig" not in sd:
        return sd[prefix + ".weight"]
    w = sd[prefix + ".weight_orig"]
    u, v = sd[prefix + ".weight_u"], sd[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            vn = torch.mv(wm.t(), u)
            v.copy_(vn / vn.norm().clamp_min(1e-12))
            un = torch.mv(wm, v)
            u.copy_(un / un.norm().clamp_min(1e-12))
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


# Test hook (tests/test_gpu_fullsize.py): a list of boolean masks, consumed in call order by every piecewise-linear activation
# of g_forward / d_forward.  The activation then takes its branch from the mask instead of from the sign of its own input -
# x where the mask is set, leak * x elsewhere - so that an fp64 run can be put on the SAME side of every LeakyReLU / ReLU as
# another implementation's run of the same step: what is left between the two gradients is rounding, not the ~1e-3 per
# flipped activation of SURVEY F10.  A mask of half the extent is read through a nearest x2 upsample (the product normalises
# and activates in front of the upsample, which commutes).  None = the reference behaviour.
ACT_REPLAY = None
ACT_RECORD = None       # a list: receives the mask (x > 0) of every such activation, in call order


def _act(x, leak):
    if ACT_RECORD is not None:
        ACT_RECORD.append(x.detach() > 0)
    if ACT_REPLAY is not None:
        m = ACT_REPLAY.pop(0)
        if m.shape != x.shape:
            if m.shape[:2] == x.shape[:2] and m.shape[2] * 2 == x.shape[2] and m.shape[3] * 2 == x.shape[3]:
                m = m.repeat_interleave(2, 2).repeat_interleave(2, 3)
            else:
                raise ValueError("activation mask %s does not fit the activation input %s" % (tuple(m.shape), tuple(x.shape)))
        return torch.where(m, x, x * leak)
    return F.leaky_relu(x, leak) if leak > 0 else F.relu(x)


# --------------------------------------------------------------------------- generator pieces
class _Ctx:
    """Carries per-call mode for the generator: training flag, image location, per-layer
    streaming padders (eval), and whether to use the loop-faithful merge/crop."""

    def __init__(self, cfg, training, loc, padders, loops):
        self.cfg, self.training, self.loc, self.padders, self.loops = cfg, training, loc, padders, loops


def _conv_lp(sd, name, x, ctx, merged_input=False):
    cfg = ctx.cfg
    w = sn_weight(sd, name + ".conv", ctx.training)
    b = sd[name + ".conv.bias"]
    if cfg.padding_mode != "local":
        return F.conv2d(x, w, b, stride=1, padding=1)
    gh, gw = cfg.num_patches_h, cfg.num_patches_w
    if ctx.training:
        xp = P.local_pad(x, gh, gw, cfg.outer_padding, merged_input=merged_input, loops=ctx.loops)
    else:
        pad = ctx.padders.setdefault(
            name, P.StreamPadder(gh, gw, cfg.outer_padding, merge_input=not merged_input))
        xp = pad(x, ctx.loc)
    return F.conv2d(xp, w, b, stride=1, padding=0)


def _bn(sd, name, x, training, affine=True):
    return F.batch_norm(
        x, sd[name + ".running_mean"], sd[name + ".running_var"],
        sd[name + ".weight"] if affine else None, sd[name + ".bias"] if affine else None,
        training, 0.1, 1e-5)


def _bump(sd, name, training):
    if training and name + ".num_batches_tracked" in sd:
        sd[name + ".num_batches_tracked"] += 1


def _ssm(sd, name, x, m, ctx):
    """(1+gamma)*bn(x)+beta, [gamma,beta]=embed(relu(mlp_shared(map))).  layers.py:228-234."""
    p = 0 if ctx.cfg.padding_mode == "local" else 1
    out = _bn(sd, name + ".bn", x, ctx.training, affine=False)
    _bump(sd, name + ".bn", ctx.training)
    w0 = sn_weight(sd, name + ".mlp_shared.0", ctx.training)
    actv = F.relu(F.conv2d(m.float(), w0, sd[name + ".mlp_shared.0.bias"], padding=p))
    w1 = sn_weight(sd, name + ".embed", ctx.training)
    emb = F.conv2d(actv, w1, sd[name + ".embed.bias"], padding=p)
    gamma, beta = emb.chunk(2, dim=1)
    return (1 + gamma) * out + beta


def _norm(sd, name, x, m, ctx):
    if ctx.cfg.type_norm == "SSM":
        return _ssm(sd, name, x, m, ctx)
    y = _bn(sd, name, x, ctx.training)
    _bump(sd, name, ctx.training)
    return y


def _block(sd, name, x, m, ctx):
    """Pre-activation residual block.  reference models/layers.py:301-322."""
    cfg = ctx.cfg
    out = _act(_norm(sd, name + ".bn1", x, m, ctx), cfg.leak)
    out = _conv_lp(sd, name + ".conv1", out, ctx)
    out = _act(_norm(sd, name + ".bn2", out, m, ctx), cfg.leak)
    out = _conv_lp(sd, name + ".conv2", out, ctx)
    sc = x
    if name + ".conv3.bias" in sd:  # learnable shortcut iff in != out channels
        if cfg.type_norm == "SSM":
            sc = _ssm(sd, name + ".bn3", sc, m, ctx)
        sc = F.conv2d(sc, sn_weight(sd, name + ".conv3", ctx.training), sd[name + ".conv3.bias"])
    return out + sc


def attention(sd, name, x, training=True):
    """Per-patch SAGAN attention.  reference models/layers.py:246-258."""
    b, c, h, w = x.shape
    th = F.conv2d(x, sn_weight(sd, name + ".theta", training), sd[name + ".theta.bias"])
    ph = F.max_pool2d(F.conv2d(x, sn_weight(sd, name + ".phi", training), sd[name + ".phi.bias"]), [2, 2])
    g = F.max_pool2d(F.conv2d(x, sn_weight(sd, name + ".g", training), sd[name + ".g.bias"]), [2, 2])
    th = th.reshape(b, c // 8, -1)
    ph = ph.reshape(b, c // 8, -1)
    g = g.reshape(b, c // 2, -1)
    beta = F.softmax(torch.bmm(th.transpose(1, 2), ph), -1)
    o = torch.bmm(g, beta.transpose(1, 2)).reshape(b, c // 2, h, w)
    o = F.conv2d(o, sn_weight(sd, name + ".o", training), sd[name + ".o.bias"])
    return sd[name + ".gamma"] * o + x


def g_forward(sd, cfg, z, maps=None, training=True, loc="1st_row_1st_col", padders=None, loops=False):
    """ResidualPatchGenerator.forward.  reference models/generators.py:86-124.

    z is the merged latent (N, z_dim, gh*b+2, gw*b+2); returns patches
    (N*gh*gw, img_ch, P, P).  In training mode BN running stats in ``sd`` are updated
    in place, as the reference's modules do.
    """
    if maps is None:
        maps = [None] * cfg.n_layers_G
    ctx = _Ctx(cfg, training, loc, padders if padders is not None else {}, loops)
    h = _conv_lp(sd, "start", z, ctx, merged_input=(cfg.padding_mode == "local"))
    h = _block(sd, "block1", h, maps[0], ctx)
    for i in range(2, cfg.n_layers_G + 1):
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        h = _block(sd, "block%d" % i, h, maps[i - 1], ctx)
        if i == 3 and cfg.attention:
            h = attention(sd, "attention", h, training)
    if cfg.type_norm == "BN":
        h = _bn(sd, "bn", h, training)
        _bump(sd, "bn", training)
    h = _act(h, cfg.leak)
    h = _conv_lp(sd, "final", h, ctx)
    return torch.tanh(h)


# --------------------------------------------------------------------------- discriminator
def d_strides(n_layers_D):
    """Strides of the n_layers_D+1 4x4 convs.  reference models/discriminators.py:187-204."""
    s = [2]
    for n in range(1, n_layers_D):
        s.append(1 if n == n_layers_D - 1 else 2)
    s.append(1)
    return s


def d_forward(sd, cfg, x, training=True):
    """PatchDiscriminator.forward (no norm layer).  reference models/discriminators.py:208-210."""
    strides = d_strides(cfg.n_layers_D)
    h = x
    for i, s in enumerate(strides):
        name = "model.%d" % (2 * i)
        h = F.conv2d(h, sn_weight(sd, name, training), sd[name + ".bias"], stride=s, padding=1)
        if i < len(strides) - 1:
            h = _act(h, 0.2)
    return h
