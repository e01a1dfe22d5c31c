"""Layer modules with the reference's names, constructor signatures and state_dict keys
(reference models/layers.py), executing on the MI355X kernels of libitg_hip.so.

Every module accepts either the reference's NCHW tensors (patch batches, converted at the
module boundary) or the internal patch-grid NHWC form (:class:`ops.GT`), in which case the
result stays in that form - this is how the generator keeps its whole forward/backward in
NHWC without per-layer conversions.  There is no CPU path.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from ..ops import GT


def _pad_mode(outer_padding):
    if outer_padding == "replicate":
        return ops.PAD_REPLICATE
    if outer_padding == "constant":
        return ops.PAD_ZERO
    raise ValueError("outer_padding must be 'replicate' or 'constant', got %r" % (outer_padding,))


# read once at import (an os.environ lookup per layer call is measurable in the 6.8 ms a step takes to issue)
_ENV_RES_UPS = True
_ENV_BN_FUSE = True
_ENV_UP2_FOLD = os.environ.get("ITG_UP2_FOLD", "1") == "1"
_ENV_HALO_INTERIOR = os.environ.get("ITG_HALO_INTERIOR", "0") == "1"      # band training: interior rows first, border rows after the exchange
_ENV_BN_FORK = True      # the shortcut's gradient is added inside bn1's backward kernel


def up2_fold_enabled():
    """ITG_UP2_FOLD=0: materialise the x2 upsample in front of every block's first conv again (A/B).  Default: the conv
    runs in folded form on the half-size tensor (ops.conv(up2=True)): conv3x3(up2x(x)) is four 2 x 2 convs of x with
    phase-summed weights - 4/9 of the multiply-adds, no upsampled tensor, the upsample's backward inside the input gradient."""
    return _ENV_UP2_FOLD


def res_upsample_enabled():
    """_ENV_RES_UPS = False: materialise the upsampled shortcut again (A/B)."""
    return _ENV_RES_UPS


def _whole_image(image_location):
    return all(k in image_location for k in ("1st_row", "1st_col", "last_row", "last_col"))


# ------------------------------------------------------------------------------- parameters
class _ConvParams(nn.Module):
    """Holds a conv's parameters under the reference's names: ``weight``/``bias`` or, with
    spectral norm, ``weight_orig``/``weight_u``/``weight_v``/``bias`` (the buffers of
    torch.nn.utils.spectral_norm as wrapped at reference models/layers.py:178-200).
    Orthogonal init, zero bias (reference utils.py:745-750)."""

    def __init__(self, ch_in, ch_out, k, SN=False, stride=1, padding=0, bias=True):
        super().__init__()
        self.ch_in, self.ch_out, self.k, self.stride, self.padding, self.SN = ch_in, ch_out, k, stride, padding, SN
        w = torch.empty(ch_out, ch_in, k, k)
        nn.init.orthogonal_(w, gain=1)
        if SN:
            self.weight_orig = nn.Parameter(w)
            u = nn.functional.normalize(torch.randn(ch_out), dim=0, eps=1e-12)
            v = nn.functional.normalize(torch.randn(ch_in * k * k), dim=0, eps=1e-12)
            self.register_buffer("weight_u", u)
            self.register_buffer("weight_v", v)
        else:
            self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(torch.zeros(ch_out))
        else:
            self.register_parameter("bias", None)

    # Set by engine.Trainer: parameter gradients are accumulated by the backward kernels straight into
    # the (flat) .grad buffers instead of being handed to autograd's AccumulateGrad.
    grad_sinks = False
    _preset_inv = None      # 1/sigma computed ahead by a model-level batched power iteration
    _packed = None          # (forward panel, dgrad panel) kept current by engine.PackSet, else packed per call
    _packed_direct = None   # stride-2 Winograd layers: (direct forward panel, direct dgrad panel) for batches under the size rule
    _packed_kind = "plain"  # what the persistent panels are: "plain" | "up2" (itg_pack_up2_*) | "wino" (itg_pack_wino_*)
    up2 = False             # this conv sits behind a x2 upsample that its owner folds into it (ResBlockGenerator.conv1)

    wino_ok = True          # engine.BandTrainer clears it on the generator: its bands carry explicit halo rows (pad_h = 0)

    @property
    def wino(self):
        """Wide stride-1 layer whose forward, input gradient and weight gradient run in the Winograd domain when the call
        qualifies (ops.wino_applicable): the discriminator's 4 x 4 256 -> 512 layer (F(4 x 4, 4 x 4)) and the generator's wide
        3 x 3 layers that are not behind a folded upsample (F(4 x 4, 3 x 3): 416 -> 416, 208 -> 208)."""
        if not (ops.WINOGRAD and self.wino_ok and self.stride == 1 and self.ch_in % 16 == 0 and ops.MFMA_PRECISION == ops.PREC_F32):
            return False
        if self.k == 4:
            return self.padding == 1 and self.ch_in >= 64 and self.ch_out >= 64
        return (self.k == 3 and ops.WINOGRAD_G and not self.up2 and self.ch_in >= 96 and self.ch_out >= 96
                and ops.ld_for(self.ch_out) % 16 == 0)

    @property
    def wino_s2(self):
        """Wide 4 x 4 stride-2 layer whose FORWARD runs through F(4 x 4, 2 x 2) on its four parity classes (ops.wino_s2_applicable;
        the discriminator's 64 -> 128 and 128 -> 256 layers): 6.25 multiplications per output and input channel instead of 16."""
        return (ops.WINOGRAD and ops.WINOGRAD_S2 and self.wino_ok and self.k == 4 and self.stride == 2 and self.padding == 1
                and self.ch_in >= ops.WINO_S2_MIN_CI and self.ch_out >= 64 and ops.MFMA_PRECISION == ops.PREC_F32)

    def pack_jobs(self):
        """Allocate this layer's persistent panels and return its ops.pack_multi jobs (two, three for a stride-2 Winograd layer)."""
        w = self.weight_orig if self.SN else self.weight
        co, ci, k, st = w.shape[0], w.shape[1], self.k, self.stride
        kind = "wino" if self.wino else ("wino_s2" if self.wino_s2 else ("up2" if self.up2 else "plain"))
        nf, nd = ops.pack_sizes(co, ci, k, k, st, kind == "up2", 2 if kind == "wino_s2" else kind == "wino")
        # the stride-2 Winograd layers whose input gradient may take the adjoint pipeline (ops._wino_s2_dgrad) keep a third
        # panel: the forward panel transposed (packed per call it was 30 us on the input-gradient chain, twice per step)
        nt = (ops._lib.fn("itg_pack_wino_s2_dgrad_size")(ops.ld_for(ci), ops.ld_for(co))
              if kind == "wino_s2" and ops.WINO_S2_DGRAD and ci >= ops.WINO_S2_DGRAD_MIN_CI else 0)
        if (self._packed is None or self._packed[0].device != w.device or self._packed_kind != kind
                or self._packed[0].numel() != nf or (self._packed[2].numel() if len(self._packed) > 2 else 0) != nt):
            self._packed = (torch.empty(nf, device=w.device, dtype=torch.float32),
                            torch.empty(nd, device=w.device, dtype=torch.float32))
            if nt:
                self._packed += (torch.empty(nt, device=w.device, dtype=torch.float32),)
        self._packed_kind = kind
        kf, kd = {"plain": (0, 1), "up2": (2, 3), "wino": (6, 7) if k == 3 else (4, 5), "wino_s2": (8, 1)}[kind]
        jobs = [(w, self._packed[0], co, ci, ops.ld_for(ci), k, k, 1, kf),
                (w, self._packed[1], co, ci, ops.ld_for(co), k, k, st, kd)]
        if nt:
            jobs.append((w, self._packed[2], co, ci, ops.ld_for(co), ops.ld_for(ci), k, st, 9))      # (kh slot: ci_ld)
        self._packed_direct = None
        if kind == "wino_s2":
            # A stride-2 Winograd layer runs BOTH forms every step: the generated batch through F(4 x 4, 2 x 2), a batch under
            # ops.WINO_S2_MIN_TILES (the real 192^2 crops behind D's 128 -> 256 layer: 288 tiles) through the direct kernels.
            # The direct forward panel is persistent too, and the direct form's input gradient takes _packed[1] (kind 1 IS the
            # direct input-gradient panel): two pack launches fewer on D(real)'s chain per step.
            npf = ops.pack_sizes(co, ci, k, k, st)[0]
            fwd = torch.empty(npf, device=w.device, dtype=torch.float32)
            self._packed_direct = (fwd, self._packed[1])
            jobs.append((w, fwd, co, ci, ops.ld_for(ci), k, k, 1, 0))
        return jobs

    def weight_and_sn(self):
        """(weight tensor, sn tuple or None); runs the power iteration in training mode.  u / v are not
        copied: they only change in the next forward, which never precedes this forward's backward."""
        if not self.SN:
            return self.weight, None
        inv = self._preset_inv
        self._preset_inv = None
        if inv is None:
            inv = ops.sn_power_iter(self.weight_orig, self.weight_u, self.weight_v, training=self.training)
        return self.weight_orig, (inv, self.weight_u, self.weight_v)

    def _sinks(self, w):
        if not self.grad_sinks or not w.requires_grad or w.grad is None:
            return None
        b = self.bias
        return (w.grad, b.grad if (b is not None and b.grad is not None) else None)

    def run(self, x, pad=None, pad_mode=ops.PAD_ZERO, act=ops.ACT_NONE, slope=0.0, residual=None, out_grid=None,
            pad_h=-1, in_act=None, defer_act_bwd=False, out_stats=False, out=None, up2=False):
        w, sn = self.weight_and_sn()
        p_ = self.padding if pad is None else pad
        wino = self.wino and ops.wino_applicable(x, self.k, self.k, self.stride, p_, pad_h, pad_mode, ops.MFMA_PRECISION, up2,
                                                 out_stats, out, self.ch_out)
        if not wino and self.wino_s2 and ops.wino_s2_applicable(x, self.k, self.k, self.stride, p_, pad_h, pad_mode, ops.MFMA_PRECISION,
                                                                up2, residual, out, self.ch_out):
            wino = 2
        kind = "wino_s2" if wino == 2 else ("wino" if wino else ("up2" if up2 else "plain"))
        packed = self._packed if (self._packed is None or self._packed_kind == kind) else None    # else: packed per call
        if packed is None and self._packed is not None and kind == "plain" and self._packed_kind == "wino_s2":
            packed = self._packed_direct          # the direct form of a stride-2 Winograd layer (pack_jobs)
        return ops.conv(x, w, self.bias, self.k, self.k, self.stride, p_,
                        pad_mode, act, slope, residual, sn, out_grid, self._sinks(w), pad_h, packed=packed,
                        in_act=in_act, defer_act_bwd=defer_act_bwd, out_stats=out_stats, out=out, up2=up2, wino=wino)

    def forward(self, x):
        """Plain conv on an NCHW image batch (zero padding), as the reference's nn.Conv2d."""
        g = x if isinstance(x, GT) else ops.to_grid(x, 1, 1, merged=True)
        y = self.run(g)
        return y if isinstance(x, GT) else ops.to_nchw(y, merged=True)


def batched_power_iteration(module):
    """One batched spectral-norm power iteration for every SN conv under ``module`` (4 launches per 8
    layers instead of 5 per layer); each conv consumes its 1/sigma at its next call."""
    convs = [m for m in module.modules() if isinstance(m, _ConvParams) and m.SN]
    for i in range(0, len(convs), 8):
        grp = convs[i:i + 8]
        invs = ops.sn_power_iter_multi([(m.weight_orig, m.weight_u, m.weight_v) for m in grp],
                                       training=grp[0].training)
        for m, inv in zip(grp, invs):
            m._preset_inv = inv


def conv3x3(ch_in, ch_out, SN=False, s=1, p=1, bias=True, padding_mode="zeros"):
    return _ConvParams(ch_in, ch_out, 3, SN, s, p, bias)


def conv4x4(ch_in, ch_out, SN=False, s=2, p=1, bias=True):
    return _ConvParams(ch_in, ch_out, 4, SN, s, p, bias)


def conv1x1(ch_in, ch_out, SN=False, s=1, p=0, bias=True):
    return _ConvParams(ch_in, ch_out, 1, SN, s, p, bias)


# ------------------------------------------------------------------------------- LocalPadder
class LocalPadder(nn.Module):
    """The local-padding operator (reference models/layers.py:38-173).

    Training mode (and the first sub-image at inference): every patch receives a 1-pixel halo
    from its 8 grid neighbours; the outer frame of the merged image is replicated or zero.
    Eval mode streams sub-images in raster order and carries the left column / top row of the
    previous ones.  The class-level configuration of the reference is kept
    (:meth:`set_attributes`); a generator additionally pins a per-instance copy so that two
    generators with different grids can coexist.
    """
    num_patches_h = 3
    num_patches_w = 3
    outer_padding = "replicate"
    padding_size = 1
    conv_reduction = 2

    @classmethod
    def set_attributes(cls, num_patches_h=3, num_patches_w=3, outer_padding="replicate", padding_size=1,
                       conv_reduction=2):
        cls.num_patches_h, cls.num_patches_w = num_patches_h, num_patches_w
        cls.outer_padding, cls.padding_size, cls.conv_reduction = outer_padding, padding_size, conv_reduction

    def __init__(self, merge_patches_into_image=True):
        super().__init__()
        self.merge_patches_into_image = merge_patches_into_image
        self._cfg = None
        self.halo = None       # dist.RowHalo when the patch grid is sharded by rows across ranks
        self.reset_state()

    def pin(self, gh, gw, outer):
        self._cfg = (gh, gw, outer)

    def cfg(self):
        if self._cfg is not None:
            return self._cfg
        if self.padding_size != 1 or self.conv_reduction != 2:
            raise ValueError("only padding_size=1 / conv_reduction=2 (3x3 convs) is meaningful")
        return self.num_patches_h, self.num_patches_w, self.outer_padding

    def reset_state(self):
        self._v = self._v_next = None       # left column now / for the next sub-image: (n, gh*P, ld)
        self._h = None                      # top row for this call: (n, gw*P+2, ld)
        self._h_cur = self._h_next = None   # row buffers: current sub-image row / being assembled

    def first_position(self, image_location):
        return self.training or ("1st_row" in image_location and "1st_col" in image_location)

    # ---- eval-mode state machine (reference layers.py:103-143) on patch-grid tensors
    def _update(self, x, loc):
        t = x.t
        n, gh, gw, ph, pw, ld = t.shape
        outer = self.cfg()[2]
        if self._v_next is not None:
            self._v = self._v_next
        self._v_next = None if "last_col" in loc else t[:, :, gw - 2, :, pw - 1, :].reshape(n, gh * ph, ld).contiguous()
        row = t[:, gh - 2, :, ph - 1, :, :].reshape(n, gw * pw, ld)
        s = row if "last_col" in loc else row[:, :(gw - 1) * pw]
        if "1st_col" in loc:
            if "1st_row" not in loc:
                hn = self._h_next
                if outer == "replicate":
                    self._h_cur = torch.cat((hn[:, :1], hn, hn[:, -1:]), 1)
                else:
                    z = torch.zeros_like(hn[:, :1])
                    self._h_cur = torch.cat((z, hn, z), 1)
            self._h_next = s.contiguous()
        else:
            self._h_next = torch.cat((self._h_next, s), 1)
        if self._h_cur is not None:
            self._h = self._h_cur[:, :gw * pw + 2].contiguous()
            self._h_cur = None if "last_col" in loc else self._h_cur[:, (gw - 1) * pw:]

    def halo_sources(self, x, image_location):
        """Eval mode: update the carried halos from this sub-image and return (left, top) to use."""
        self._update(x, image_location)
        if "1st_row" in image_location and "1st_col" in image_location:
            return None, None
        left = None if "1st_col" in image_location else self._v
        top = None if "1st_row" in image_location else self._h
        return left, top

    def forward(self, input, image_location="1st_row_1st_col"):
        gh, gw, outer = self.cfg()
        pm = _pad_mode(outer)
        if not self.merge_patches_into_image:
            # already merged + randomly padded latent: crop only (reference layers.py:152-155,165-170)
            return ops.local_pad_nchw(input, gh, gw, pm, merged=True)
        if self.training:
            return ops.local_pad_nchw(input, gh, gw, pm, merged=False)
        x = ops.to_grid(input, gh, gw, merged=False)
        left, top = self.halo_sources(x, image_location)
        return ops.to_nchw(ops.local_pad_grid(x, pm, left, top), merged=False)


class conv2d_lp(nn.Module):
    """3x3 convolution with local padding (reference models/layers.py:8-36)."""

    def __init__(self, ch_in, ch_out, SN=False, padding_mode="zeros", merge_patches_into_image=True):
        super().__init__()
        self.padding_mode = padding_mode
        if padding_mode == "local":
            self.local_padder = LocalPadder(merge_patches_into_image)
            self.conv = conv3x3(ch_in, ch_out, SN, 1, 0)
        else:
            self.conv = conv3x3(ch_in, ch_out, SN, 1, 1)

    def forward_grid(self, x, image_location="1st_row_1st_col", act=ops.ACT_NONE, slope=0.0, residual=None, out_stats=False,
                     bn=None, bn_act=(ops.ACT_NONE, 0.0), upsample=False, fork_out=None):
        """x: GT patches (or, for the generator's ``start`` layer, the merged latent as a 1x1-grid GT).
        ``out_stats``: the output feeds a training-mode BatchNorm - its statistics are taken in this conv's epilogue.
        ``bn`` (a _BNParams) + ``bn_act`` = (activation, slope) + ``upsample``: the conv's input is
        up2x?(act(bn(x))) (reference models/layers.py:301-311): the normalisation runs as its own pass (at the block input's
        resolution when the conv folds the upsample).  (A loader-side form - BatchNorm-apply inside the conv kernels' tile loaders,
        `itg_in_norm` - was built in round 3, measured slower in every configuration and deleted in round 5: DESIGN section 3.)"""
        # the x2 upsample in front of a block's first conv is folded into the conv (ops.conv(up2=True)) on the paths that hand
        # the patch grid to the kernel as it is; the normalisation then stays at the block input's resolution
        lp_ = self.local_padder if self.padding_mode == "local" else None
        direct = (lp_ is None
                  or (lp_.merge_patches_into_image and lp_.halo is None and (self.training or _whole_image(image_location)))
                  or (lp_.merge_patches_into_image and lp_.halo is not None and lp_.training))      # band training: halo rows of SOURCE pixels
        fold = bool(upsample) and bn is not None and up2_fold_enabled() and direct
        if bn is not None:
            # fork_out (a list): the caller reads the BatchNorm's input a second time (residual shortcut); it gets an alias of
            # it whose gradient the BatchNorm backward absorbs (ops.bn_act(fork=True))
            # a row-sharded band (training): the BatchNorm writes straight into the halo-row layout the band conv reads
            band = (lp_ is not None and lp_.merge_patches_into_image and lp_.halo is not None and lp_.training
                    and not (upsample and not fold) and not (_ENV_HALO_INTERIOR and lp_.halo.world > 1)
                    and x.t.shape[1] == 1 and x.t.shape[2] == 1)
            x = bn.run(x, act=bn_act[0], slope=bn_act[1], upsample=upsample and not fold, consumer_upsamples=fold,
                       fork=fork_out is not None, pad_rows=band)
            if fork_out is not None and x.fork is not None:
                fork_out.append(x.fork)
        # a residual at half the output's patch extent (the un-upsampled shortcut) is read through the x2 upsample by the conv
        # epilogue on the paths that hand the patch grid to the kernel as it is; the reshaping paths materialise it
        out_ph = x.t.shape[3] * (2 if (fold and upsample) else 1)
        half_res = residual is not None and residual.t.shape[3] * 2 == out_ph
        native = (self.padding_mode == "local" and self.local_padder.merge_patches_into_image
                  and (self.training or (_whole_image(image_location) and self.local_padder.halo is None)))
        if half_res and not native:
            residual = ops.upsample2x(residual)
        if self.padding_mode != "local":
            # per-patch zero padding: every patch is an independent image
            n, gh, gw, ph, pw, ld = x.t.shape
            flat = GT(x.t.reshape(n * gh * gw, 1, 1, ph, pw, ld), x.c, x.stats)
            s_ = 2 if (fold and upsample) else 1
            r = None if residual is None else GT(residual.t.reshape(n * gh * gw, 1, 1, ph * s_, pw * s_, -1), residual.c)
            y = self.conv.run(flat, pad=1, pad_mode=ops.PAD_ZERO, act=act, slope=slope, residual=r, up2=fold)
            return GT(y.t.reshape(n, gh, gw, ph * s_, pw * s_, -1), y.c)
        lp = self.local_padder
        gh, gw, outer = lp.cfg()
        if not lp.merge_patches_into_image:
            # valid conv over the pre-padded merged latent == crop(b+2, stride b) + valid conv per patch
            return self.conv.run(x, pad=0, act=act, slope=slope, residual=residual, out_grid=(gh, gw), out_stats=out_stats)
        if lp.halo is not None:
            if lp.training:
                return self._forward_band_train(x, lp, outer, act, slope, residual, up2=fold)
            return self._forward_row_sharded(x, lp, outer, act, slope, residual)
        if lp.training or fold:     # (fold outside training: ONE sub-image that is the whole picture, no state is carried on)
            # halo + outer padding are resolved inside the conv's tile loader
            return self.conv.run(x, pad=1, pad_mode=_pad_mode(outer), act=act, slope=slope, residual=residual,
                                 out_stats=out_stats, up2=fold)
        left, top = lp.halo_sources(x, image_location)
        if left is None and top is None and "1st_row" in image_location and "1st_col" in image_location:
            return self.conv.run(x, pad=1, pad_mode=_pad_mode(outer), act=act, slope=slope, residual=residual)
        xp = ops.local_pad_grid(x, _pad_mode(outer), left, top)
        n, g1, g2, ph, pw, ld = xp.t.shape
        flat = GT(xp.t.reshape(n * g1 * g2, 1, 1, ph, pw, ld), xp.c)
        r = None if residual is None else GT(residual.t.reshape(n * g1 * g2, 1, 1, ph - 2, pw - 2, -1), residual.c)
        y = self.conv.run(flat, pad=0, act=act, slope=slope, residual=r)
        return GT(y.t.reshape(n, g1, g2, ph - 2, pw - 2, -1), y.c)

    def _forward_band_train(self, x, lp, outer, act, slope, residual, up2=False):
        """Training with the patch grid sharded by rows: x is this rank's band in image layout
        (n, 1, 1, H, W, ld).  The band is extended by the neighbours' boundary rows (differentiable RCCL
        exchange; at the image border the outer padding row) and convolved with vertical padding 0.
        ``up2``: x is the band BEFORE the block's x2 upsample; its boundary rows are what the neighbours need (the
        upsampled halo row is the nearest source row), and the conv folds the upsample (half the exchanged bytes)."""
        t = x.t
        n, g1, g2, H, W, ld = t.shape
        if g1 != 1 or g2 != 1:
            raise ValueError("band training expects the image layout (1x1 grid), got %r" % (x,))
        padded = getattr(x, "padded", False)
        interior_first = _ENV_HALO_INTERIOR and lp.halo.world > 1 and H >= 3 and not padded
        if not interior_first:
            # the band in the halo-row layout (written there by the BatchNorm in front, or copied once for the producers that
            # cannot), its two halo rows filled in place (ops.band_halo: exchange + itg_band_halo_fill), vertical padding 0
            ext = ops.band_halo(x if padded else ops.band_extend(x), lp.halo, outer == "replicate")
            return self.conv.run(ext, pad=1, pad_mode=_pad_mode(outer), act=act, slope=slope, residual=residual, pad_h=0, up2=up2)
        rows = t[:, 0, 0]
        first, last = rows[:, 0], rows[:, H - 1]
        top, bottom = ops.halo_exchange(first, last, lp.halo, defer_wait=True)
        if top is None:
            top = first if outer == "replicate" else torch.zeros_like(first)
        if bottom is None:
            bottom = last if outer == "replicate" else torch.zeros_like(last)
        # Interior first (ITG_HALO_INTERIOR=1, SURVEY section 7 / VERDICT r3 item 7): the exchange has been POSTED (on the halo's
        # communication stream with ITG_HALO_STREAM=1); the output rows that need no neighbour row - all but the first and the
        # last k (k = 1, or 2 behind the folded upsample) - are convolved from the band itself while the rows travel, then this
        # stream waits for them and the 2 k border rows follow as ONE small conv over [top, row 0, row 1] and
        # [row H-2, row H-1, bottom] of every image.  Three differentiable pieces, concatenated.
        k = 2 if up2 else 1
        r_full = None
        if residual is not None:
            r_full = residual.t[:, 0, 0]
            if r_full.shape[1] * 2 == (2 * H if up2 else H) and not up2:
                r_full = ops.upsample2x(residual).t[:, 0, 0]      # a half-size shortcut cannot be cut at single output rows
        def res_rows(a, b):             # output rows [a, b) of the residual (read through the x2 upsample when it is half-size)
            if r_full is None:
                return None
            half = r_full.shape[1] * 2 == (2 * H if up2 else H)
            ra, rb = (a // 2, b // 2) if half else (a, b)
            return GT(r_full[:, ra:rb].contiguous().unsqueeze(1).unsqueeze(1), residual.c)
        Hout = 2 * H if up2 else H
        y_int = self.conv.run(GT(rows.reshape(n, 1, 1, H, W, ld), x.c), pad=1, pad_mode=_pad_mode(outer), act=act, slope=slope,
                              residual=res_rows(k, Hout - k), pad_h=0, up2=up2)
        lp.halo.wait()                  # the neighbours' rows are needed from here on
        b_in = torch.cat((torch.stack((top, rows[:, 0], rows[:, 1]), 1), torch.stack((rows[:, H - 2], rows[:, H - 1], bottom), 1)), 0)
        r_b = None
        if r_full is not None:
            rt, rb_ = res_rows(0, k), res_rows(Hout - k, Hout)
            r_b = GT(torch.cat((rt.t, rb_.t), 0), residual.c)
        y_b = self.conv.run(GT(b_in.reshape(2 * n, 1, 1, 3, W, ld), x.c), pad=1, pad_mode=_pad_mode(outer), act=act, slope=slope,
                            residual=r_b, pad_h=0, up2=up2)
        yt = torch.cat((y_b.t[:n], y_int.t, y_b.t[n:]), 3)
        return GT(yt, y_int.c)

    def _forward_row_sharded(self, x, lp, outer, act, slope, residual):
        """This rank owns a band of patch rows: fetch the neighbours' boundary pixel rows (RCCL send/recv),
        build the padded patches with them (outer padding only at the true image border), valid conv."""
        t = x.t
        n, gh, gw, ph, pw, ld = t.shape
        first = t[:, 0, :, 0, :, :].reshape(n, gw * pw, ld).contiguous()
        last = t[:, gh - 1, :, ph - 1, :, :].reshape(n, gw * pw, ld).contiguous()
        top, bottom = lp.halo.exchange(first, last)

        def widen(row):     # + the two corner pixels: the columns are not sharded, so plain outer padding
            if row is None:
                return None
            if outer == "replicate":
                return torch.cat((row[:, :1], row, row[:, -1:]), 1).contiguous()
            z = torch.zeros_like(row[:, :1])
            return torch.cat((z, row, z), 1).contiguous()

        xp = ops.local_pad_grid(x, _pad_mode(outer), top=widen(top), bottom=widen(bottom))
        flat = GT(xp.t.reshape(n * gh * gw, 1, 1, ph + 2, pw + 2, ld), xp.c)
        r = None if residual is None else GT(residual.t.reshape(n * gh * gw, 1, 1, ph, pw, -1), residual.c)
        y = self.conv.run(flat, pad=0, act=act, slope=slope, residual=r)
        return GT(y.t.reshape(n, gh, gw, ph, pw, -1), y.c)

    def forward(self, x, image_location="1st_row_1st_col"):
        if isinstance(x, GT):
            return self.forward_grid(x, image_location)
        if self.padding_mode != "local":
            return ops.to_nchw(self.conv.run(ops.to_grid(x, 1, 1, True), pad=1), True)
        lp = self.local_padder
        gh, gw, _ = lp.cfg()
        if lp.merge_patches_into_image:
            g = ops.to_grid(x, gh, gw, merged=False)
        else:
            g = ops.to_grid(x, 1, 1, merged=True)
        return ops.to_nchw(self.forward_grid(g, image_location), merged=False)


# ------------------------------------------------------------------------------- normalisation
class _BNParams(nn.BatchNorm2d):
    """nn.BatchNorm2d as a parameter/buffer container (same state_dict keys); the arithmetic
    runs in the HIP kernels.  ``sync`` (an ops.SyncGroup) makes the statistics global."""
    sync = None
    grad_sinks = False

    def _sinks(self):
        if self.grad_sinks and self.weight is not None and self.weight.requires_grad and self.weight.grad is not None:
            return (self.weight.grad, self.bias.grad)
        return None

    def run(self, x, act=ops.ACT_NONE, slope=0.0, upsample=False, consumer_upsamples=False, fork=False, pad_rows=False):
        sinks = self._sinks()
        return ops.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                          training=self.training, eps=self.eps, momentum=self.momentum, act=act, slope=slope,
                          upsample=upsample, sync=self.sync, sinks=sinks, consumer_upsamples=consumer_upsamples, fork=fork,
                          pad_rows=pad_rows)

    def forward(self, x):
        if isinstance(x, GT):
            return self.run(x)
        return ops.to_nchw(self.run(ops.to_grid(x, 1, 1, True)), True)


class _INParams(nn.Module):
    """nn.InstanceNorm2d(affine=False) (reference models/discriminators.py:183-185, `--norm_layer_D instance`): every image
    is normalised with its own per-channel statistics - the BatchNorm kernels on one image at a time, no parameters, no
    buffers (so the state_dict is the reference's).  Unused by the BASELINE configurations: correctness, not speed."""

    def __init__(self, num_features, eps=1e-5, affine=False):
        super().__init__()
        if affine:
            raise NotImplementedError("the reference builds its instance norm with affine=False")
        self.num_features, self.eps = num_features, eps

    def run(self, x, act=ops.ACT_NONE, slope=0.0):
        outs = [ops.bn_act(GT(x.t[i:i + 1], x.c), None, None, None, None, None, training=True, eps=self.eps, momentum=0.0,
                           act=act, slope=slope).t for i in range(x.t.shape[0])]
        return GT(torch.cat(outs, 0), x.c)

    def forward(self, x):
        if isinstance(x, GT):
            return self.run(x)
        return ops.to_nchw(self.run(ops.to_grid(x, 1, 1, True)), True)


class StochasticSpatialModulation(nn.Module):
    """(1+gamma)*BN(x)+beta with [gamma,beta] = embed(ReLU(mlp_shared(map))).
    reference models/layers.py:203-234."""

    def __init__(self, in_channel, map_dim, SN=False, padding_mode="zeros"):
        super().__init__()
        self.in_channel = in_channel
        self.out_channels = in_channel * 2
        self.p = 1 if padding_mode == "zeros" else 0
        self.bn = _BNParams(in_channel, affine=False)
        self.mlp_shared = nn.Sequential(conv3x3(map_dim, 128, SN=SN, p=self.p), nn.ReLU())
        self.embed = conv3x3(128, self.out_channels, bias=True, SN=SN, p=self.p)
        w = self.embed.weight_orig if SN else self.embed.weight
        with torch.no_grad():      # the reference's (odd) init: re-orthogonalise dim-1 slice, zero the rest
            nn.init.orthogonal_(w.data[:, :in_channel], gain=1)
            w.data[:, in_channel:].zero_()

    def run(self, x, maps, act=ops.ACT_NONE, slope=0.0):
        """x: GT (n,gh,gw,r,r); maps: NCHW (n*gh*gw, map_dim, r+4, r+4) or an equivalent 1x1-grid GT."""
        m = maps if isinstance(maps, GT) else ops.to_grid(maps.float(), 1, 1, merged=True)
        nb = m.t.shape[0]
        # the 128-channel hidden map is the largest tensor of the model: at inference (thousands of patches)
        # the per-patch map convs run in chunks so that it stays below the 32-bit offset range of the kernels
        per_patch = 128 * m.t.shape[3] * m.t.shape[4]
        chunk = max(1, min(nb, (1 << 29) // max(1, per_patch)))
        if chunk >= nb or torch.is_grad_enabled():
            a = self.mlp_shared[0].run(m, pad=self.p, act=ops.ACT_LRELU, slope=0.0)
            e = self.embed.run(a, pad=self.p)
        else:
            # every chunk's modulation goes straight into its slice of ONE result tensor (no concatenation copy afterwards)
            red = 2 * (1 - self.p)                       # a valid 3x3 conv shrinks the map by 2
            hh, ww = m.t.shape[3] - 2 * red, m.t.shape[4] - 2 * red
            et = torch.empty((nb, 1, 1, hh, ww, ops.ld_for(2 * self.in_channel)), device=m.t.device, dtype=torch.float32)
            for i in range(0, nb, chunk):
                mi = GT(m.t[i:i + chunk], m.c)
                ai = self.mlp_shared[0].run(mi, pad=self.p, act=ops.ACT_LRELU, slope=0.0)
                self.embed.run(ai, pad=self.p, out=et[i:i + chunk])
                del ai
            e = GT(et, 2 * self.in_channel)
        n, gh, gw, ph, pw, _ = x.t.shape
        if e.t.shape[3] != ph or e.t.shape[4] != pw or e.t.shape[0] != n * gh * gw:
            raise ValueError("modulation map does not match the activation: %r vs %r" % (e, x))
        e = GT(e.t.reshape(n, gh, gw, ph, pw, -1), e.c)
        bn = self.bn
        return ops.ssm_modulate(x, e, bn.running_mean, bn.running_var, bn.num_batches_tracked, training=bn.training,
                                eps=bn.eps, momentum=bn.momentum, act=act, slope=slope, sync=bn.sync)

    def forward(self, inputs, maps):
        if isinstance(inputs, GT):
            return self.run(inputs, maps)
        g = ops.to_grid(inputs, 1, 1, merged=True)
        g = GT(g.t.reshape(1, g.t.shape[0], 1, *g.t.shape[3:]), g.c)   # patches as a (nb x 1) grid
        y = self.run(g, maps)
        return ops.to_nchw(GT(y.t.reshape(y.t.shape[1], 1, 1, *y.t.shape[3:]), y.c), True)


class Attention(nn.Module):
    """SAGAN-style self-attention inside each patch (reference models/layers.py:236-258)."""

    def __init__(self, channels, SN=False):
        super().__init__()
        self.channels = channels
        self.theta = conv1x1(channels, channels // 8, SN=SN)
        self.phi = conv1x1(channels, channels // 8, SN=SN)
        self.g = conv1x1(channels, channels // 2, SN=SN)
        self.o = conv1x1(channels // 2, channels, SN=SN)
        self.gamma = nn.Parameter(torch.tensor(0.0), requires_grad=True)

    def run(self, x):
        theta = self.theta.run(x)
        phi = ops.maxpool2(self.phi.run(x))
        g = ops.maxpool2(self.g.run(x))
        mid = ops.attention_core(theta, phi, g)
        o = self.o.run(mid)
        # gamma*o + x: the scalar gate is applied with a tensor op on the NHWC buffer (pad lanes stay 0)
        return GT(_ScaleAdd.apply(o.t, x.t, self.gamma), x.c)

    def forward(self, inputs):
        if isinstance(inputs, GT):
            return self.run(inputs)
        nb = inputs.shape[0]
        g = ops.to_grid(inputs, nb, 1, merged=False)
        return ops.to_nchw(self.run(g), merged=False)


class _ScaleAdd(torch.autograd.Function):
    """gamma*o + x (reference layers.py:258); gamma is a 0-dim parameter read on the device."""

    @staticmethod
    def forward(ctx, o, x, gamma):
        ctx.save_for_backward(o, gamma)
        return ops.axpby(o, x, 1.0, 1.0, a_dev=gamma)

    @staticmethod
    def backward(ctx, g):
        o, gamma = ctx.saved_tensors
        g = g.contiguous()
        return ops.axpby(g, g, 1.0, 0.0, a_dev=gamma), g, ops.dot(g, o).to(torch.float32).reshape(())


# ------------------------------------------------------------------------------- residual block
class ResBlockGenerator(nn.Module):
    """norm -> act -> conv2d_lp -> norm -> act -> conv2d_lp (+ 1x1 shortcut iff in != out).
    reference models/layers.py:260-322."""

    def __init__(self, generator, in_channels, out_channels, hidden_channels=None, padding_mode="zeros"):
        super().__init__()
        hidden_channels = out_channels if hidden_channels is None else hidden_channels
        self.learnable_sc = in_channels != out_channels
        self.type_norm = generator.type_norm
        self.leak = generator.leak
        self.map_dim = generator.map_dim
        self.SN = generator.SN
        self.conv1 = conv2d_lp(in_channels, hidden_channels, self.SN, padding_mode)
        self.conv2 = conv2d_lp(hidden_channels, out_channels, self.SN, padding_mode)
        if self.learnable_sc:
            self.conv3 = conv1x1(in_channels, out_channels, SN=self.SN)
        if self.type_norm == "BN":
            self.bn1 = _BNParams(in_channels)
            self.bn2 = _BNParams(hidden_channels)
        elif self.type_norm == "SSM":
            self.bn1 = StochasticSpatialModulation(in_channels, self.map_dim, self.SN, padding_mode)
            self.bn2 = StochasticSpatialModulation(hidden_channels, self.map_dim, self.SN, padding_mode)
            if self.learnable_sc:
                self.bn3 = StochasticSpatialModulation(in_channels, self.map_dim, self.SN, padding_mode)
        else:
            raise ValueError("type_norm must be 'BN' or 'SSM', got %r" % (self.type_norm,))
        # nn.LeakyReLU(leak) if leak > 0 else nn.ReLU(): both are ACT_LRELU with slope = leak
        self.activation = nn.LeakyReLU(self.leak) if self.leak > 0 else nn.ReLU()

    def forward_grid(self, x, map=None, image_location="1st_row_1st_col", upsample_input=False, out_stats=False):
        """x: GT.  ``upsample_input``: x is the block input BEFORE the generator's nearest x2
        upsample; the upsample is then folded into bn1 (BN mode) and moved behind the 1x1 shortcut.
        ``out_stats``: the block output goes straight into a training-mode BatchNorm (next block's bn1 / the final bn)."""
        fuse = self.type_norm == "BN" and self.training and _ENV_BN_FUSE
        A, s = ops.ACT_LRELU, float(self.leak)
        if self.type_norm == "SSM":
            if upsample_input:
                x = ops.upsample2x(x)
                upsample_input = False
        # (the learnable 1x1 shortcut on a HIP stream of its own measured 784 vs 785 crops/s: dropped in round 4)
        if self.type_norm == "SSM":
            out = self.bn1.run(x, map, act=A, slope=s)
            out = self.conv1.forward_grid(out, image_location, out_stats=fuse)
            out = self.bn2.run(out, map, act=A, slope=s)
        else:
            # BatchNorm-apply + LeakyReLU (+ the x2 upsample) happen in conv1's / conv2's tile loaders where the conv path
            # gathers its halo itself (conv2d_lp.forward_grid); otherwise as their own pass
            fo = [] if (_ENV_BN_FORK and self.training and torch.is_grad_enabled()) else None
            out = self.conv1.forward_grid(x, image_location, out_stats=fuse, bn=self.bn1, bn_act=(A, s), upsample=upsample_input,
                                          fork_out=fo)
            if fo:
                x = fo[0]          # the shortcut below reads the block input through bn1's alias (its gradient joins bn1's)
        if self.learnable_sc:
            sc = self._shortcut(x, map, upsample_input)
        else:
            sc = ops.upsample2x(x) if (upsample_input and not res_upsample_enabled()) else x
        if self.type_norm == "SSM":
            return self.conv2.forward_grid(out, image_location, residual=sc, out_stats=fuse and out_stats)
        return self.conv2.forward_grid(out, image_location, residual=sc, out_stats=fuse and out_stats, bn=self.bn2, bn_act=(A, s))

    def _shortcut(self, x, map, upsample_input):
        """The 1x1 shortcut at the block INPUT's resolution: when the block upsamples, conv2's epilogue reads it through the
        nearest x2 upsample (the 1x1 conv commutes with it), so the shortcut is never materialised at the output's size."""
        sc = x
        if self.type_norm == "SSM":
            sc = self.bn3.run(sc, map)
        sc = self.conv3.run(sc)
        if upsample_input and not res_upsample_enabled():
            return ops.upsample2x(sc)
        return sc

    def forward(self, x, map=None, image_location="1st_row_1st_col"):
        if isinstance(x, GT):
            return self.forward_grid(x, map, image_location)
        gh, gw, _ = self.conv1.local_padder.cfg() if self.conv1.padding_mode == "local" else (x.shape[0], 1, None)
        g = ops.to_grid(x, gh, gw, merged=False)
        return ops.to_nchw(self.forward_grid(g, map, image_location), merged=False)
