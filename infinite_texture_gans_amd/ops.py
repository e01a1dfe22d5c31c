"""torch.autograd bindings over the C ABI (include/itg.h).

PyTorch supplies device memory, the stream and the autograd tape; every piece of arithmetic
on the hot path is a call into libitg_hip.so.  Activations travel as *patch-grid NHWC*
tensors: a contiguous fp32 CUDA tensor of shape (n, gh, gw, ph, pw, ld) plus the logical
channel count ``c`` (ld = c rounded up to 4, pad channels are zero) - see :class:`GT`.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import Tensor as _T, ConvGeom as _G, PAD_ZERO, PAD_REPLICATE, ACT_NONE, ACT_LRELU, ACT_TANH  # noqa: F401


def ld_for(c):
    return (int(c) + 3) // 4 * 4


# Optional launch profiler (bench.py): a list that receives (kernel_tag, n_launches, flops, ev0, ev1)
# per convolution call, with HIP events recorded on the launching stream.  None = off.
PROFILE = None

# MFMA operand type of every convolution launched without an explicit ``precision``: PREC_F32 is the
# reference's arithmetic; PREC_BF16 rounds the operands to bf16 on their way into LDS (fp32 tensors,
# fp32 accumulation) - BASELINE config 3's "bf16 MFMA path", parity tolerance 1e-2 per conv.
from ._lib import PREC_F32, PREC_BF16  # noqa: E402
MFMA_PRECISION = PREC_BF16 if os.environ.get("ITG_MFMA", "f32").lower() == "bf16" else PREC_F32


class mfma_precision:
    """``with ops.mfma_precision("bf16"): ...`` (also usable as a plain setter: ``ops.mfma_precision("bf16").set()``)."""

    def __init__(self, name):
        self.value = {"f32": PREC_F32, "fp32": PREC_F32, "bf16": PREC_BF16}[str(name).lower()]

    def set(self):
        global MFMA_PRECISION
        MFMA_PRECISION = self.value
        return self

    def __enter__(self):
        global MFMA_PRECISION
        self.prev, MFMA_PRECISION = MFMA_PRECISION, self.value
        return self

    def __exit__(self, *a):
        global MFMA_PRECISION
        MFMA_PRECISION = self.prev



def _nt_tag(rows):
    r = (int(rows) + 15) // 16 * 16
    return "conv_nt<%s>" % ("16,256" if r <= 16 else "32,256" if r <= 32 else "64,256" if r <= 64 else "128,128")


class _Prof:
    def __init__(self, tag, launches, flops, nbytes=0):
        """``nbytes``: algorithmic HBM bytes of the call (operands read once + result written once)."""
        self.rec = PROFILE is not None
        if self.rec:
            self.item = [tag, launches, flops, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), nbytes]

    def __enter__(self):
        if self.rec:
            self.item[3].record()

    def __exit__(self, *a):
        if self.rec:
            self.item[4].record()
            name = _lib.fn("itg_last_conv_kernel")()
            if name:
                self.item[0] = name.decode()
            PROFILE.append(tuple(self.item))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream as a void*.  torch.cuda.current_stream() costs ~10 us of Python per call (device-index plumbing,
    a Stream object) and every op asks once or twice: ~0.7 ms of the 6.8 ms a train step takes to issue; the raw getter is
    a single C call."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _desc(t, c):
    n, gh, gw, ph, pw, ld = t.shape
    return _T(t.data_ptr(), n, gh, gw, ph, pw, int(c), ld)


def _null_desc():
    return _T(None, 1, 1, 1, 1, 1, 1, 4)


def _req_cuda(t, what):
    if not t.is_cuda:
        raise _lib.ItgError("%s must live on the GPU: this build has no CPU path" % what)
    if t.dtype != torch.float32:
        raise _lib.ItgError("%s must be float32 (the reference path is fp32 end to end)" % what)


class GT:
    """Patch-grid NHWC activation: ``t`` is (n, gh, gw, ph, pw, ld), ``c`` logical channels."""
    __slots__ = ("t", "c", "stats", "fork", "padded")

    def __init__(self, t, c, stats=None):
        # stats: fp64 [2 * ld] per-channel (sum, sum of squares) of ``t`` when the conv that produced it accumulated
        # them in its epilogue (conv(..., out_stats=True)); the BatchNorm that consumes ``t`` then skips its stats pass
        # fork: set by bn_act(fork=True): the alias of the BatchNorm's input whose gradient the BatchNorm backward absorbs
        # padded: ``t`` is a row-sharded band in the halo-row layout (n, 1, 1, H + 2, W, ld), rows 1 .. H written (bn_act(pad_rows=True))
        self.t, self.c, self.stats, self.fork, self.padded = t, int(c), stats, None, False

    n = property(lambda s: s.t.shape[0])
    gh = property(lambda s: s.t.shape[1])
    gw = property(lambda s: s.t.shape[2])
    ph = property(lambda s: s.t.shape[3])
    pw = property(lambda s: s.t.shape[4])
    ld = property(lambda s: s.t.shape[5])

    def detach(self):
        return GT(self.t.detach(), self.c)

    def __repr__(self):
        return "GT(n=%d grid=%dx%d patch=%dx%d c=%d ld=%d)" % (self.n, self.gh, self.gw, self.ph, self.pw, self.c, self.ld)


# ------------------------------------------------------------------------------- layout
class _ToGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gh, gw, merged):
        _req_cuda(x, "input")
        x = x.contiguous()
        if merged:
            n, c, H, W = x.shape
            ph, pw = H // gh, W // gw
            if ph * gh != H or pw * gw != W:
                raise _lib.ItgError("image %dx%d does not divide into a %dx%d grid" % (H, W, gh, gw))
        else:
            nb, c, ph, pw = x.shape
            n = nb // (gh * gw)
            if n * gh * gw != nb:
                raise _lib.ItgError("patch batch %d is not a multiple of the %dx%d grid" % (nb, gh, gw))
        out = torch.empty((n, gh, gw, ph, pw, ld_for(c)), device=x.device, dtype=torch.float32)
        d = _desc(out, c)
        _lib.call("itg_nchw_to_grid", _ptr(x), C.byref(d), int(merged), _stream())
        ctx.meta = (tuple(x.shape), c, merged)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, c, merged = ctx.meta
        g = g.contiguous()
        out = torch.empty(shape, device=g.device, dtype=torch.float32)
        d = _desc(g, c)
        _lib.call("itg_grid_to_nchw", C.byref(d), _ptr(out), int(merged), _stream())
        return out, None, None, None


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, c, merged):
        t = t.contiguous()
        n, gh, gw, ph, pw, ld = t.shape
        shape = (n, c, gh * ph, gw * pw) if merged else (n * gh * gw, c, ph, pw)
        out = torch.empty(shape, device=t.device, dtype=torch.float32)
        d = _desc(t, c)
        _lib.call("itg_grid_to_nchw", C.byref(d), _ptr(out), int(merged), _stream())
        ctx.meta = (tuple(t.shape), c, merged)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, c, merged = ctx.meta
        g = g.contiguous()
        out = torch.empty(shape, device=g.device, dtype=torch.float32)
        d = _desc(out, c)
        _lib.call("itg_nchw_to_grid", _ptr(g), C.byref(d), int(merged), _stream())
        return out, None, None


def to_grid(x, gh=1, gw=1, merged=True):
    """NCHW tensor -> GT.  merged: x is (n, c, gh*ph, gw*pw); else a patch batch (n*gh*gw, c, ph, pw)."""
    return GT(_ToGrid.apply(x, gh, gw, merged), x.shape[1])


def to_nchw(g, merged=True):
    return _ToNCHW.apply(g.t, g.c, merged)


# ------------------------------------------------------------------------------- spectral norm
def sn_power_iter(w_orig, u, v, training=True, eps=1e-12):
    """One power iteration in place on (u, v) (training) and 1/sigma as a 1-element tensor.
    torch.nn.utils.spectral_norm semantics as used at reference models/layers.py:190-194."""
    wgrad_streams_join()
    rows = w_orig.shape[0]
    cols = w_orig.numel() // rows
    ws = torch.empty(rows + 8 * cols + 2, device=w_orig.device, dtype=torch.float32)
    inv = ws[rows + 8 * cols:rows + 8 * cols + 1]
    with torch.no_grad():
        _lib.call("itg_spectral_norm_power_iter", _ptr(w_orig), _ptr(u), _ptr(v), rows, cols, int(training),
                  float(eps), None, _ptr(inv), _ptr(ws), _stream())
    return inv


def sn_power_iter_multi(layers, training=True, eps=1e-12):
    """Power iteration for several (w_orig, u, v) triples in 4 launches; returns the list of 1/sigma
    tensors.  Falls back to per-layer calls above 8 layers."""
    wgrad_streams_join()
    if len(layers) > 8:
        return [sn_power_iter(w, u, v, training, eps) for (w, u, v) in layers]
    n = len(layers)
    rows = [w.shape[0] for (w, _, _) in layers]
    cols = [w.numel() // w.shape[0] for (w, _, _) in layers]
    dev = layers[0][0].device
    sizes = [8 * c + r + 2 for r, c in zip(rows, cols)]
    ws = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
    works, invs, o = [], [], 0
    for sz in sizes:
        works.append(ws[o:o + sz - 2])
        invs.append(ws[o + sz - 2:o + sz - 1])
        o += sz
    PA = C.c_void_p * n
    IA = C.c_int * n
    with torch.no_grad():
        _lib.call("itg_spectral_norm_power_iter_multi", n, PA(*[w.data_ptr() for (w, _, _) in layers]),
                  PA(*[u.data_ptr() for (_, u, _) in layers]), PA(*[v.data_ptr() for (_, _, v) in layers]),
                  IA(*rows), IA(*cols), int(training), float(eps), PA(*[t.data_ptr() for t in invs]),
                  PA(*[t.data_ptr() for t in works]), _stream())
    return invs


# ------------------------------------------------------------------------------- convolution
def _residual_grad(dy, c, ups):
    """Gradient of a conv's residual input: dy itself, or its 2x2 sums when the residual was read through a nearest x2
    upsample (itg_conv2d_fwd with a half-size residual)."""
    if not ups:
        return dy
    n, gh, gw, ph, pw, ld = dy.shape
    dx = torch.empty((n, gh, gw, ph // 2, pw // 2, ld), device=dy.device, dtype=torch.float32)
    a, b = _desc(dy, c), _desc(dx, c)
    _lib.call("itg_upsample2x_bwd", C.byref(a), C.byref(b), _stream())
    return dx


_WS_SIZE = {}        # workspace sizes by (kind, shapes, geometry): the plan is a pure function of them (one ctypes call less per launch)


class _Conv(torch.autograd.Function):
    """out = act(conv(x, w*scale) + bias [+ residual]) on patch-grid tensors (merged-image
    coordinates).  ``sn`` = (inv_sigma, u, v) makes ``w`` the spectral-norm ``weight_orig``."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, sn, c_in, geom, act, slope, out_grid, sinks=None, packed=None, fuse=(None, False),
                stats=None, out_buf=None):
        kh, kw, stride, pad, pad_mode, pad_h, prec, up2, wino = geom
        pv = pad_h if pad_h >= 0 else pad
        n, gh, gw, ph, pw, ld = x.shape
        co, ci = w.shape[0], w.shape[1]
        if ci != c_in or w.shape[2] != kh or w.shape[3] != kw:
            raise _lib.ItgError("weight %s does not match conv geometry (c_in=%d, k=%dx%d)" % (tuple(w.shape), c_in, kh, kw))
        H, W = (gh * ph) << up2, (gw * pw) << up2        # up2: the conv runs on the x2 upsample of x (folded into the filter)
        Ho, Wo = (H + 2 * pv - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        if up2 and pv == 0:                              # x carries one explicit halo row of SOURCE pixels above and below
            Ho = 2 * (gh * ph - 2)
        ogh, ogw = out_grid
        if Ho % ogh or Wo % ogw:
            raise _lib.ItgError("conv output %dx%d does not divide into a %dx%d grid" % (Ho, Wo, ogh, ogw))
        x = x.contiguous()
        st = _stream()
        inv_sigma = sn[0] if sn is not None else None
        if packed is not None:      # panels packed once per optimizer step (engine.PackSet): 1/sigma rides in the epilogue
            wp, out_scale = packed[0], inv_sigma
        elif wino == 2:             # 4 x 4 stride 2: F(4 x 4, 2 x 2) on the four parity classes, forward only
            wp, out_scale = torch.empty(_lib.fn("itg_pack_wino_s2_size")(co, ld), device=x.device, dtype=torch.float32), None
            _lib.call("itg_pack_wino_s2_fwd", _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, ld, st)
        elif wino:
            sfx = "wino3" if kh == 3 else "wino"
            wp, out_scale = torch.empty(_lib.fn("itg_pack_%s_size" % sfx)(co, ld), device=x.device, dtype=torch.float32), None
            _lib.call("itg_pack_%s_fwd" % sfx, _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, ld, st)
        elif up2:
            wp, out_scale = torch.empty(_lib.fn("itg_pack_up2_fwd_size")(co, ld), device=x.device, dtype=torch.float32), None
            _lib.call("itg_pack_up2_fwd", _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, ld, st)
        else:
            wp, out_scale = torch.empty(_lib.fn("itg_pack_fwd_size")(co, ld, kh, kw), device=x.device, dtype=torch.float32), None
            _lib.call("itg_pack_fwd", _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, ld, kh, kw, st)
        oshape = (n, ogh, ogw, Ho // ogh, Wo // ogw, ld_for(co))
        if out_buf is not None:       # inference: the caller's (contiguous) slice of a larger result - no concatenation copy later
            if tuple(out_buf.shape) != oshape or not out_buf.is_contiguous() or torch.is_grad_enabled():
                raise _lib.ItgError("conv(out=...): expected a contiguous %s buffer under no_grad" % (oshape,))
            out = out_buf
        else:
            out = torch.empty(oshape, device=x.device, dtype=torch.float32)
        dx_, do_ = _desc(x, c_in), _desc(out, co)
        dr_ = _desc(residual, co) if residual is not None else _null_desc()
        g = _G(kh, kw, stride, pad, pad_mode, pad_h, prec, stats.data_ptr() if stats is not None else None, None, up2,
               _lib.GEOM_WINO if wino else 0)
        key = ("f", tuple(x.shape), tuple(out.shape), geom, c_in, co)
        nws = _WS_SIZE.get(key)
        if nws is None:
            nws = _WS_SIZE[key] = _lib.fn("itg_conv2d_fwd_workspace")(C.byref(dx_), C.byref(do_), C.byref(g))
        ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
        taps = 4 if up2 else (6.25 if wino == 2 else ((kh + 3) ** 2 / 16.0 if wino else kh * kw))      # multiply-adds per output element and input channel
        with _Prof(_nt_tag(co), 1, 2.0 * n * Ho * Wo * co * ci * taps, 4 * (x.numel() + out.numel() + wp.numel())):
            _lib.call("itg_conv2d_fwd", C.byref(dx_), _ptr(wp), _ptr(bias), _ptr(out_scale), C.byref(dr_), C.byref(do_),
                      C.byref(g), act, float(slope), _ptr(ws), nws, st)
        ctx.geom, ctx.act, ctx.slope, ctx.c_in, ctx.co = geom, act, slope, c_in, co
        _sink_act(out, co, act, "conv")
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.res_ups = residual is not None and residual.shape[3] * 2 == out.shape[3]     # read through a nearest x2 upsample
        ctx.sn = sn
        ctx.sinks = sinks          # (weight.grad, bias.grad) buffers to accumulate into, or None
        ctx.packed = packed
        # fuse = (in_act, defer): in_act = (act, slope) of the layer that produced x -> this backward returns the
        # gradient w.r.t. that layer's PRE-activation; defer = the consumer of this layer's output does the same
        # for us, so our own activation backward is skipped.  The two always come in pairs (see callers).
        ctx.in_act, ctx.defer_act = fuse
        # Winograd layers: the transformed input V sits at the start of the forward's workspace; the weight gradient of the
        # same x takes it from there instead of transforming x again (itg_conv_geom.wino_v) - kept only when w takes a gradient
        # (only where the weight gradient can consume it - fp32 operands, Winograd weight gradient on; the workspace
        # is V followed by the GEMM result M, one allocation behind one C-ABI pointer: M's ~40 MB per pass stay alive with V)
        ctx.wino_ws = ws if (wino and (wino == 1 or _wino_s2_wgrad(x.shape, ci)) and WINO_KEEP_V and WINOGRAD_WGRAD and prec == PREC_F32 and w.requires_grad) else None       # (grad mode reads off inside forward())
        ctx.save_for_backward(x, w, out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        kh, kw, stride, pad, pad_mode, pad_h, prec, up2, wino = ctx.geom
        wino_s2 = wino == 2                 # the stride-2 form: forward and weight gradient in the transformed domain, the input
        wino = 1 if wino == 1 else 0        # gradient on the direct kernel (per class it would move 4 x dx's bytes)
        taps = 4 if up2 else kh * kw
        co, ci = ctx.co, ctx.c_in
        if BACKWARD_ENTRY_HOOK is not None and ctx.sinks is not None:
            BACKWARD_ENTRY_HOOK(ctx.sinks)
        st = _stream()
        dout = dout.contiguous()
        if ctx.act != ACT_NONE and not ctx.defer_act:
            dy = torch.empty_like(dout)
            a, b, c_ = _desc(out, co), _desc(dout, co), _desc(dy, co)
            _lib.call("itg_act_bwd", C.byref(a), C.byref(b), C.byref(c_), ctx.act, float(ctx.slope), st)
        else:
            dy = dout
        s2_dgrad = wino_s2 and dy.shape[5] % 16 == 0 and _wino_s2_dgrad(x.shape, ci)        # the input gradient as the adjoint of the stride-2 Winograd forward
        g = _G(kh, kw, stride, pad, pad_mode, pad_h, prec, None, None, up2, _lib.GEOM_WINO if (wino or s2_dgrad) else 0)
        ddy = _desc(dy, co)
        inv_sigma = ctx.sn[0] if ctx.sn is not None else None
        gx = gw_ = gb = None
        if ctx.needs_input_grad[0]:
            if s2_dgrad and ctx.packed is not None and len(ctx.packed) > 2:
                wp, out_scale = ctx.packed[2], inv_sigma      # the layer's third persistent panel (layers._ConvParams.pack_jobs)
            elif s2_dgrad:                  # its transposed panel packed per call (the layer's persistent input-gradient panel is the
                                            # direct kernel's: the real batch of the same layer takes that one)
                wp, out_scale = torch.empty(_lib.fn("itg_pack_wino_s2_dgrad_size")(x.shape[5], dy.shape[5]), device=x.device,
                                            dtype=torch.float32), None
                _lib.call("itg_pack_wino_s2_dgrad", _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, x.shape[5], dy.shape[5], st)
            elif ctx.packed is not None:
                wp, out_scale = ctx.packed[1], inv_sigma
            elif wino:
                sfx = "wino3" if kh == 3 else "wino"
                wp, out_scale = torch.empty(_lib.fn("itg_pack_%s_size" % sfx)(ci, dy.shape[5]), device=x.device,
                                            dtype=torch.float32), None
                _lib.call("itg_pack_%s_dgrad" % sfx, _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, dy.shape[5], st)
            elif up2:
                wp, out_scale = torch.empty(_lib.fn("itg_pack_up2_dgrad_size")(ci, dy.shape[5]), device=x.device,
                                            dtype=torch.float32), None
                _lib.call("itg_pack_up2_dgrad", _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, dy.shape[5], st)
            else:
                wp, out_scale = torch.empty(_lib.fn("itg_pack_dgrad_size")(ci, dy.shape[5], kh, kw, stride), device=x.device,
                                            dtype=torch.float32), None
                _lib.call("itg_pack_dgrad", _ptr(w), _ptr(inv_sigma), _ptr(wp), co, ci, dy.shape[5], kh, kw, stride, st)
            gx = None
            if FRAMES is not None and pad_mode == PAD_REPLICATE and stride == 1 and (pad > 0 or pad_h > 0):
                # replicate-padded layer inside a step engine's backward pass: its dx buffer is kept across steps and its
                # frame was zeroed with every other layer's in one launch (begin_frames) - no zeroing launch in front of
                # this input gradient
                fkey = (w.data_ptr(), tuple(x.shape))
                if fkey in _FRAMES_READY:
                    _FRAMES_READY.discard(fkey)
                    gx = FRAMES[fkey][0]
                    g.flags |= _lib.GEOM_FRAME_ZEROED
                elif fkey not in FRAMES:
                    gx = torch.empty_like(x)
                    FRAMES[fkey] = (gx, ci)              # from the next pass on
            if gx is None:
                gx = torch.empty_like(x)
            ddx = _desc(gx, ci)
            npix_out = dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * dy.shape[4]
            key = ("d", tuple(dy.shape), tuple(gx.shape), ctx.geom, ci, co, s2_dgrad)
            nws = _WS_SIZE.get(key)
            if nws is None:
                nws = _WS_SIZE[key] = _lib.fn("itg_conv2d_dgrad_workspace")(C.byref(ddy), C.byref(ddx), C.byref(g))
            ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
            fuse = bn = None
            if _BN_FUSE:
                bn = _BN_FUSE.pop(x.data_ptr(), None)
                bx, bstat, by = (bn[0](), bn[1](), bn[5]()) if bn is not None else (None, None, None)
                if (by is not None and bx is not None and bstat is not None and ctx.in_act is None and not wino and not s2_dgrad
                        and by.shape == x.shape and bn[4] == ci):
                    bact, bslope = bn[2], bn[3]
                    bsums = _zeros_f64(2 * x.shape[5], x.device)
                    bxd = _desc(bx, ci)
                    fuse = _lib.BnBwdFuse(C.pointer(bxd), bstat[2 * x.shape[5]:].data_ptr(), bstat.data_ptr(), int(bact), float(bslope),
                                          bsums.data_ptr(), 0, 0)
                    g.bn_bwd = C.pointer(fuse)
            with _Prof(_nt_tag(ci), 1, 2.0 * npix_out * co * ci * (6.25 if s2_dgrad else ((kh + 3) ** 2 / 16.0 if wino else taps)),
                       4 * (dy.numel() + gx.numel() + wp.numel())):
                ia = ctx.in_act
                dact = _desc(x, ci) if ia is not None else _null_desc()
                _lib.call("itg_conv2d_dgrad", C.byref(ddy), _ptr(wp), _ptr(out_scale), C.byref(ddx), C.byref(dact),
                          ia[0] if ia is not None else ACT_NONE, float(ia[1]) if ia is not None else 0.0, C.byref(g),
                          _ptr(ws), nws, st)
            if fuse is not None:
                g.bn_bwd = None
                if fuse.taken:                 # the BatchNorm's backward (next on the chain) finds its sums here
                    _BN_SUMS[gx.data_ptr()] = (bsums, gx, bx.data_ptr())
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if need_w or need_b:
            if wino_s2:                       # (its own geometry: the input gradient's flag does not carry over)
                wg_w = _wino_s2_wgrad(x.shape, ci)
                g = _G(kh, kw, stride, pad, pad_mode, pad_h, prec, None, None, up2, _lib.GEOM_WINO if wg_w else 0,
                       ctx.wino_ws.data_ptr() if (wg_w and ctx.wino_ws is not None) else None)
            elif ctx.wino_ws is not None and not wino_s2:       # a geometry of its own for the weight gradient: + the forward's V
                g = _G(kh, kw, stride, pad, pad_mode, pad_h, prec, None, None, up2, _lib.GEOM_WINO, ctx.wino_ws.data_ptr())
            wsink, bsink = ctx.sinks if ctx.sinks is not None else (None, None)
            # Everything below only writes into the flat gradient buffers when sinks cover the requested
            # gradients: such a weight-gradient can run on the side stream, next to the input-gradient chain.
            side = None
            if not ((need_w and wsink is None) or (need_b and bsink is None)):
                side = wgrad_stream_for((wsink if wsink is not None else bsink).data_ptr())
            if side is not None and _lib.CAPTURE_LOG is not None and not capture_rule(torch.cuda.current_stream(), "weight-gradient fork"):
                side = None                # (recording) the forking stream holds no node yet: run the weight gradient in line
            if side is not None:
                ev = torch.cuda.Event()
                ev.record()
                # the operands were allocated on the main stream: keep them referenced until the owner of the side
                # stream has joined it (engine.Trainer._join clears the list), so the allocator cannot hand their
                # memory to a later main-stream kernel while the weight-gradient still reads it
                WGRAD_KEEPALIVE.append((x, dy, inv_sigma, ctx.wino_ws))
                side.wait_event(ev)
                if side not in _wgrad_dirty:
                    _wgrad_dirty.append(side)
            # The kernels take their stream as an argument: torch's current stream is NOT switched to the side stream (the
            # torch.cuda.stream context manager costs ~25 us of host time per layer).  Scratch allocated here therefore belongs
            # to the main stream and joins the keep-alive list like the operands.
            if True:
                st = C.c_void_p(side.cuda_stream) if side is not None else _stream()
                dxd = _desc(x, ci)
                sinks_ok = not ((need_w and wsink is None) or (need_b and bsink is None))
                one = []
                if WGRAD_DEFER is not None and sinks_ok and _queue_wgrad(x, dxd, dy, ddy, g, w, wsink, bsink, need_w, need_b,
                                                                         ctx.sn, st):
                    gw_ = gb = None         # the slabs are computed; ops.flush_deferred() finishes the layer
                elif (SN_FUSED_REDUCE and ctx.sn is not None and need_w and sinks_ok
                      and _queue_wgrad(x, dxd, dy, ddy, g, w, wsink, bsink, need_w, need_b, ctx.sn, st, queue=one)):
                    # spectrally normalised layer: the reduce launch also accumulates <G, W> and one apply launch finishes the
                    # layer - 3-4 launches instead of 5 (contraction, [group stage], reduce, dot, apply), same stream
                    _finish_jobs(one, st)
                    if side is not None:
                        WGRAD_KEEPALIVE.append(one[0][2])
                    gw_ = gb = None
                else:
                    key = ("w", tuple(x.shape), tuple(dy.shape), ctx.geom, ci, co)
                    nws = _WS_SIZE.get(key)
                    if nws is None:
                        nws = _WS_SIZE[key] = _lib.fn("itg_conv2d_wgrad_workspace")(C.byref(dxd), C.byref(ddy), C.byref(g))
                    # every tensor the side-stream kernels of this layer touch that torch allocated HERE (main stream) goes
                    # through `scratch`, which also puts it on the keep-alive list below (one place, not per branch)
                    keep = []

                    def scratch(t):
                        keep.append(t)
                        return t

                    ws = scratch(torch.empty(nws, device=x.device, dtype=torch.float32))
                    direct_w = wsink is not None and ctx.sn is None and need_w
                    direct_b = bsink is not None and need_b
                    gw_ = wsink if direct_w else scratch(torch.empty_like(w))
                    gb = bsink if direct_b else (scratch(torch.empty_like(w[:, 0, 0, 0])) if need_b else None)
                    npix_out = dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * dy.shape[4]
                    wino_wg = bool(g.flags & _lib.GEOM_WINO) and WINOGRAD_WGRAD
                    with _Prof(_nt_tag(co).replace("nt", "tn(+reduce)"), 1,
                               2.0 * npix_out * co * ci * ((6.25 if stride == 2 else (kh + 3) ** 2 / 16.0) if wino_wg else taps),
                               4 * (x.numel() + dy.numel() + w.numel())):
                        # separate accumulate flags: a spectrally normalised layer sinks its bias gradient but takes dW into a
                        # temporary (no zero-fill launch for it)
                        _lib.call("itg_conv2d_wgrad", C.byref(dxd), C.byref(ddy), _ptr(gw_), _ptr(gb), C.byref(g),
                                  (ACC_DW if direct_w else 0) | (ACC_DB if direct_b else 0), _ptr(ws), nws, st)
                    if ctx.sn is not None and need_w:
                        _, u, v = ctx.sn
                        rows, cols = co, w.numel() // co
                        d_orig = wsink if wsink is not None else scratch(torch.empty_like(w))
                        # the <G, W> accumulator: a slice of the step's zeroed arena when there is one (no memset launch)
                        ws2 = ARENA.take(1) if ARENA is not None and ARENA.buf.device == x.device else None
                        zeroed = ws2 is not None
                        if ws2 is None:
                            ws2 = scratch(torch.empty(2, device=x.device, dtype=torch.float64))
                        _lib.call("itg_spectral_norm_bwd", _ptr(gw_), _ptr(w), _ptr(u), _ptr(v), _ptr(inv_sigma), rows, cols,
                                  _ptr(d_orig), (ACC_DW if wsink is not None else 0) | (WS_ZEROED if zeroed else 0), _ptr(ws2), st)
                        gw_ = None if wsink is not None else d_orig
                    elif direct_w:
                        gw_ = None
                    if direct_b:
                        gb = None
                    if not need_w:
                        gw_ = None
                    if side is not None:
                        WGRAD_KEEPALIVE.append(tuple(keep))
        gres = _residual_grad(dy, co, ctx.res_ups) if ctx.has_res and ctx.needs_input_grad[3] else None
        return gx, gw_, gb, gres, None, None, None, None, None, None, None, None, None, None, None


WINOGRAD = os.environ.get("ITG_WINOGRAD", "1") == "1"
# F(4 x 4, 3 x 3) for the generator's wide 3 x 3 layers (416 -> 416, 208 -> 208 on 12 x 12 / 24 x 24 images): built, parity-tested,
# and measured neutral (1 104.8 vs 1 108.8 crops/s, conv time 7.44 vs 7.48 ms per step): a quarter of the multiplications, but
# three to five dependent launches of 8-20 us where the direct path has one or two - these layers are launch-latency-bound.  Opt-in.
WINOGRAD_G = os.environ.get("ITG_WINOGRAD_G", "0") == "1"
WINOGRAD_WGRAD = os.environ.get("ITG_WINOGRAD_WGRAD", "1") == "1"     # read by the library itself; here for the flop accounting
WINO_KEEP_V = True           # the weight gradient re-uses the forward's transformed input


def wino_applicable(x, kh, kw, stride, pad, pad_h, pad_mode, prec, up2=False, out_stats=False, out=None, co=None):
    """Whether conv(..., wino=True) takes the Winograd pipeline for this call (else the direct kernels): F(4 x 4, 4 x 4) for
    4 x 4 zero-padded convs, F(4 x 4, 3 x 3) for 3 x 3 zero- or replicate-padded ones; stride 1, pad 1, fp32, channel pitches
    that are multiples of 16 on both sides (the input gradient runs the same pipeline on dy)."""
    if not (WINOGRAD and kh == kw and stride == 1 and pad == 1 and pad_h in (-1, 1) and prec == PREC_F32 and not up2
            and x.t.shape[5] % 16 == 0 and out is None and (co is None or ld_for(co) % 16 == 0)):
        return False
    if kh == 4:
        return pad_mode == PAD_ZERO
    return kh == 3 and WINOGRAD_G and pad_mode in (PAD_ZERO, PAD_REPLICATE)


WINO_S2_MIN_CI = 64      # narrower layers stay direct (64: both of D's stride-2 layers)
WINO_S2_MIN_TILES = 1024      # (module constants: tests and tools set them; no environment variables since round 5)
WINO_S2_WGRAD_MIN_CI = 128
WINO_S2_DGRAD = True
WINO_S2_DGRAD_MIN_CI = 128      # (64 -> 128 layer: 164 vs 171 us incl. its per-call panel: even)
WINO_S2_WGRAD = True      # ... and its weight gradient (25 contractions over the tiles, the forward's V re-used)
WINOGRAD_S2 = os.environ.get("ITG_WINOGRAD_S2", "1") == "1"      # F(4 x 4, 2 x 2) forward for 4 x 4 stride-2 layers (conv_wino.hip wino_conv_s2)


def wino_s2_applicable(x, kh, kw, stride, pad, pad_h, pad_mode, prec, up2=False, residual=None, out=None, co=None):
    """Whether conv(..., wino=2) takes the forward-only stride-2 Winograd form: 4 x 4, stride 2, pad 1, zero padding, fp32, no
    residual (the discriminator's 64 -> 128 / 128 -> 256 layers, reference models/discriminators.py:190-195)."""
    if not (WINOGRAD and WINOGRAD_S2 and kh == 4 and kw == 4 and stride == 2 and pad == 1 and pad_h in (-1, 1) and pad_mode == PAD_ZERO
            and prec == PREC_F32 and not up2 and residual is None and out is None and x.t.shape[5] % 4 == 0):
        return False
    # 25 GEMMs over the 4 x 4 output tiles: below ~1 000 tiles (the real batch behind the 128 -> 256 layer: 288) they under-fill
    # the chip and the direct kernel wins (94 vs 80 us)
    return wino_s2_tiles(x) >= WINO_S2_MIN_TILES


def wino_s2_tiles(x):
    n, gh, gw, ph, pw, _ = x.t.shape
    ho, wo = (gh * ph) // 2, (gw * pw) // 2
    return n * ((ho + 3) // 4) * ((wo + 3) // 4)


def _wino_s2_dgrad(x_shape, ci):
    """The stride-2 form's input gradient through the adjoint pipeline (conv_wino.hip wino_conv_s2_dgrad)."""
    n, gh, gw, ph, pw, _ = x_shape
    tiles = n * (((gh * ph) // 2 + 3) // 4) * (((gw * pw) // 2 + 3) // 4)
    return WINO_S2_DGRAD and ci >= WINO_S2_DGRAD_MIN_CI and tiles >= WINO_S2_MIN_TILES


def _wino_s2_wgrad(x_shape, ci):
    """The stride-2 form's weight gradient in the transformed domain: where the transforms are cheap next to the contraction -
    >= 128 input channels on >= 1 000 tiles (D's 128 -> 256 layer on the generated batch: 196 -> 143 us; its 64 -> 128 layer
    191 vs 194 us, the real batches 72 / 106 vs 70 / 72 us: direct)."""
    n, gh, gw, ph, pw, _ = x_shape
    tiles = n * (((gh * ph) // 2 + 3) // 4) * (((gw * pw) // 2 + 3) // 4)
    return WINOGRAD_WGRAD and WINO_S2_WGRAD and ci >= WINO_S2_WGRAD_MIN_CI and tiles >= WINO_S2_MIN_TILES


def conv(x, w, bias=None, kh=3, kw=3, stride=1, pad=0, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, residual=None,
         sn=None, out_grid=None, sinks=None, pad_h=-1, precision=None, packed=None, in_act=None, defer_act_bwd=False,
         out_stats=False, out=None, up2=False, wino=False):
    """x: GT.  Returns GT with ``out_grid`` (default: the input grid).  ``sinks`` = (weight.grad, bias.grad)
    buffers: the backward then accumulates the parameter gradients straight into them (and reports no
    gradient to autograd), which removes one AccumulateGrad add kernel per parameter.
    ``up2``: the result is conv3x3(nearest_up2x(x)) (stride 1, pad 1) computed in folded form from the half-size ``x``
    (itg_conv_geom.up2); ``packed`` panels must then be the itg_pack_up2_* ones."""
    og = out_grid if out_grid is not None else (x.gh, x.gw)
    r = residual.t if residual is not None else None
    prec = MFMA_PRECISION if precision is None else precision
    # out_stats: the epilogue also accumulates the per-channel (sum, sum of squares) of the output - the statistics pass
    # of the BatchNorm that consumes it (at most 512 padded channels; single-output-channel taps-as-rows convs excluded)
    stats = None
    if out_stats and ld_for(w.shape[0]) <= 512 and w.shape[0] > 1:
        stats = _zeros_f64(2 * ld_for(w.shape[0]), x.t.device)
    # wino: Winograd F(4 x 4, 4 x 4) for the forward and the input gradient (4 x 4, stride 1, pad 1, zero padding, plain images,
    # fp32; itg_conv_geom.flags & ITG_GEOM_WINO); ``packed`` panels must then be the itg_pack_wino_* ones
    if wino == 2:
        wino = 2 if wino_s2_applicable(x, kh, kw, stride, pad, pad_h, pad_mode, prec, up2, residual, out, w.shape[0]) else 0
    else:
        wino = 1 if (bool(wino) and wino_applicable(x, kh, kw, stride, pad, pad_h, pad_mode, prec, up2, stats is not None, out, w.shape[0])) else 0
    t = _Conv.apply(x.t, w, bias, r, sn, x.c, (kh, kw, stride, pad, pad_mode, pad_h, prec, 1 if up2 else 0, wino), act, slope, og, sinks, packed,
                    (in_act, bool(defer_act_bwd)), stats, out)
    return GT(t, w.shape[0], stats)


# ------------------------------------------------------------------------------- batch norm (+act, +upsample)
from .dist import SyncGroup, _active  # noqa: E402,F401  (sync-BN statistics exchange)


# Side stream for weight-gradient kernels (set by engine.Trainer for the duration of a step).  The power
# iteration of a spectrally normalised layer rewrites u / v in place, which a still-running weight-gradient of
# the previous pass reads: sn_power_iter* therefore waits for this stream first.
BACKWARD_ENTRY_HOOK = None   # callable((weight.grad, bias.grad)) at the entry of every conv backward that owns gradient sinks
                             # (engine.GradExchange: the first layer of the model's head starting its backward means the
                             # gradients of everything behind it are enqueued)
ACC_DW, ACC_DB, WS_ZEROED = 1, 2, 4      # include/itg.h: ITG_ACC_DW, ITG_ACC_DB, ITG_WS_ZEROED
WGRAD_STREAM = None          # one stream or a list of streams used round-robin (consecutive layers overlap each other too)
WGRAD_KEEPALIVE = []
_wgrad_rr = [0]


_wgrad_slot = {}

# Deferred weight-gradient reduce (engine.Trainer sets WGRAD_DEFER to a list for the duration of a step): a conv backward
# then only runs the contraction into its slabs (itg_conv2d_wgrad_slabs, on the weight-gradient stream) and queues the rest;
# flush_deferred() finishes every queued layer of the backward pass in ONE launch (itg_wgrad_reduce_multi: slab sums, OIHW
# transposition, bias gradients, the <G, W> dots of spectrally normalised layers) plus one itg_spectral_norm_bwd_multi -
# instead of 2-5 small launches per layer (79 second-stage + 20 spectral-norm launches per train step before).
# Frames of the replicate-padded layers' input gradients (see _Conv.backward): a step engine sets FRAMES to its dict for the
# duration of a backward pass (begin_frames / end_frames)
FRAMES = None
_FRAMES_READY = set()


def begin_frames(frames):
    """Zero the 1-pixel frames of every dx buffer registered in ``frames`` with one launch per ZERO_FRAMES_MAX buffers and
    make them available to the backward pass that follows."""
    global FRAMES
    FRAMES = frames
    _FRAMES_READY.clear()
    items = list(frames.items())
    st = _stream()
    for i in range(0, len(items), _lib.ZERO_FRAMES_MAX):
        chunk = items[i:i + _lib.ZERO_FRAMES_MAX]
        arr = (_T * len(chunk))(*[_desc(t, c) for _, (t, c) in chunk])
        _lib.call("itg_zero_frames", arr, len(chunk), st)
        _FRAMES_READY.update(k for k, _ in chunk)


def end_frames():
    global FRAMES
    FRAMES = None
    _FRAMES_READY.clear()


WGRAD_DEFER = None
SN_FUSED_REDUCE = os.environ.get("ITG_SN_FUSED_REDUCE", "0") == "1"      # see _Conv.backward: 10 launches fewer per step, 0.2 % slower
_WGRAD_WS = {}           # persistent slab workspaces / spectral-norm temporaries, keyed by layer and shape (never freed: the
                         # slabs must outlive the conv call, and a captured hipGraph replays their addresses)


def _persistent(key, numel, device, dtype=torch.float32):
    t = _WGRAD_WS.get(key)
    if t is None or t.numel() < numel or t.device != device:
        t = _WGRAD_WS[key] = torch.empty(numel, device=device, dtype=dtype)
    return t


def _queue_wgrad(x, dxd, dy, ddy, gwg, w, wsink, bsink, need_w, need_b, sn, st, queue=None):
    """Deferred form of the weight gradient of one conv (appended to ``queue``, default WGRAD_DEFER); False when the
    layer cannot be deferred."""
    if gwg.up2:           # folded-upsample layers reduce their class slabs through a kernel of their own
        return False
    key = ((wsink if wsink is not None else bsink).data_ptr(), tuple(x.shape), tuple(dy.shape))
    pending = WGRAD_DEFER if queue is None else queue
    # (a private queue - the spectral-norm fused reduce - shares the persistent slabs with the step's deferred queue: a sink
    # that is still pending THERE must not have its slabs overwritten either)
    also = WGRAD_DEFER if (queue is not None and WGRAD_DEFER is not None and WGRAD_DEFER is not queue) else ()
    if any(j[3][0] == key[0] for j in pending) or any(j[3][0] == key[0] for j in also):
        # the same gradient sink queued twice before a flush (a weight-shared conv, a module applied twice in one backward
        # pass - e.g. the interior and border pieces of an interior-first band conv): two jobs of one reduce launch must not
        # share dw / db (itg.h: they would read-modify-write the same words concurrently), and with equal shapes the second
        # contraction would also overwrite the first one's persistent slabs - this call takes the direct path
        return False
    nws = _lib.fn("itg_conv2d_wgrad_workspace")(C.byref(dxd), C.byref(ddy), C.byref(gwg))
    ws = _persistent(("ws",) + key, nws, x.device)
    job = _lib.WgradJob()
    co, ci, kh, kw = w.shape
    wino_wg = bool(gwg.flags & _lib.GEOM_WINO) and WINOGRAD_WGRAD
    with _Prof(_nt_tag(co).replace("nt", "tn"), 1,
               2.0 * (dy.numel() // dy.shape[5]) * co * ci * ((6.25 if gwg.stride == 2 else (kh + 3) ** 2 / 16.0) if wino_wg else kh * kw),
               4 * (x.numel() + dy.numel() + w.numel())):
        if _lib.CAPTURE_LOG is not None:          # (this entry point is called through fn(): its error code is inspected below)
            _lib.CAPTURE_LOG.add(getattr(st, "value", st) or 0)
        rc = _lib.fn("itg_conv2d_wgrad_slabs")(C.byref(dxd), C.byref(ddy), C.byref(gwg), _ptr(ws), nws, C.byref(job), st)
    if rc == -1:                    # ITG_ERR_ARG: a path without slabs (single-output-channel taps-as-rows layer)
        return False
    if rc:
        raise _lib.ItgError("itg_conv2d_wgrad_slabs failed with %d" % rc)
    snjob = None
    if sn is not None and need_w:
        inv_sigma, u, v = sn
        gtmp = _persistent(("g",) + key, w.numel(), x.device)
        dot = ARENA.take(1) if ARENA is not None and ARENA.buf.device == x.device else None
        if dot is None:
            dot = torch.zeros(1, device=x.device, dtype=torch.float64)
        job.dw, job.w_orig, job.dot = gtmp.data_ptr(), w.data_ptr(), dot.data_ptr()
        job.accumulate = ACC_DB if (need_b and bsink is not None) else 0
        snjob = _lib.SnJob(gtmp.data_ptr(), u.data_ptr(), v.data_ptr(), inv_sigma.data_ptr(), dot.data_ptr(), wsink.data_ptr(),
                           w.shape[0], w.numel() // w.shape[0], ACC_DW, 0)
        keep = (gtmp, dot, u, v, inv_sigma, w)
    else:
        job.dw = wsink.data_ptr() if need_w else _persistent(("g",) + key, w.numel(), x.device).data_ptr()
        job.w_orig, job.dot = None, None
        job.accumulate = (ACC_DW if need_w else 0) | (ACC_DB if need_b else 0)
        keep = ()
    job.db = bsink.data_ptr() if (need_b and bsink is not None) else None
    pending.append((job, snjob, (ws, x, dy) + keep, key, getattr(st, "value", st) or 0))
    return True


def _finish_jobs(jobs, st):
    """The reduce stage of queued weight gradients on stream ``st``: itg_wgrad_reduce_multi (+ the <G, W> dots) and
    itg_spectral_norm_bwd_multi."""
    n = _lib.WGRAD_MAX_JOBS
    for i in range(0, len(jobs), n):
        chunk = [j[0] for j in jobs[i:i + n]]
        _lib.call("itg_wgrad_reduce_multi", (_lib.WgradJob * len(chunk))(*chunk), len(chunk), st)
    sn = [j[1] for j in jobs if j[1] is not None]
    for i in range(0, len(sn), n):
        chunk = sn[i:i + n]
        _lib.call("itg_spectral_norm_bwd_multi", (_lib.SnJob * len(chunk))(*chunk), len(chunk), st)


DEFER_PER_STREAM = True


def flush_deferred(per_stream=False):
    """Finish every queued weight gradient.  Default: on the CURRENT stream, which must have been ordered behind the streams
    the slabs were computed on (wgrad_streams_join).  ``per_stream`` (call it BEFORE the join): every stream finishes the
    layers whose slabs it computed itself - one reduce (+ one spectral-norm) launch per weight-gradient stream and backward
    pass, issued behind that stream's last contraction, so the reduces run beside the tail of the input-gradient chain
    instead of behind the join (round 3's single reduce at the join sat on the critical path: 765 vs 780 crops/s)."""
    jobs = WGRAD_DEFER
    if not jobs:
        return
    if per_stream and not DEFER_PER_STREAM:
        return                           # the plain flush behind the join finishes them
    if per_stream:
        by = {}
        for j in jobs:
            by.setdefault(j[4], []).append(j)
        for handle, group in by.items():
            _finish_jobs(group, C.c_void_p(handle))
    else:
        _finish_jobs(jobs, _stream())
    if not torch.cuda.is_current_stream_capturing():
        WGRAD_KEEPALIVE.append(tuple(j[2] for j in jobs))       # operands stay referenced until the owner's join clears the list
    del jobs[:]


def wgrad_stream_for(sink_key):
    """The weight-gradient stream of the layer whose gradient sink starts at ``sink_key``.  A layer keeps its
    stream for the life of the process (slots are dealt round-robin on first sight): the read-modify-write
    accumulations of D(real)'s and D(fake)'s weight gradients into the same flat .grad slice are then ordered
    by the stream itself, whatever else the step overlaps."""
    w = WGRAD_STREAM
    if w is None:
        return None
    if isinstance(w, (list, tuple)):
        slot = _wgrad_slot.get(sink_key)
        if slot is None:
            _wgrad_rr[0] += 1
            slot = _wgrad_slot[sink_key] = _wgrad_rr[0]
        return w[slot % len(w)]
    return w


_STREAM_DEBUG = bool(int(os.environ.get("ITG_DEBUG", "0"), 0) & 2)


def concurrent_streams(device, want, spin_us=150, candidates=12):
    """``want`` streams from torch's pool that run CONCURRENTLY with the current stream and with each other.

    HIP multiplexes streams onto at most GPU_MAX_HW_QUEUES (default 4) hardware queues, dealt in creation order, and a
    hardware queue executes in order: two streams that landed on one queue do not overlap at all, and a cross-queue
    wait parked in a shared queue blocks the other stream's kernels behind it.  Which pool stream sits on which queue
    depends on every stream created before (torch's pools, RCCL's internal streams: with a process group initialised
    first the same code lost the D(real) || G-forward overlap, 785 -> 709 crops/s).  So the mapping is measured: a
    one-wave spin kernel (itg_stream_spin) on the already chosen streams and on a candidate at once; the candidate is
    kept when the whole thing takes one spin, not two.  At most three are kept beside the current stream: more than four
    busy hardware queues is the one configuration that is far worse than no overlap (533 crops/s, measured with
    GPU_MAX_HW_QUEUES=5..12).  If fewer than ``want`` concurrent streams exist the last ones repeat (correct, just
    in order)."""
    cur = torch.cuda.current_stream(device)
    ckey = (str(torch.device(device)), cur.cuda_stream, int(want))
    if ckey in _PLACED:            # a second engine of the process (bench.py's direct-algorithm leg) gets the same queues
        return list(_PLACED[ckey])

    def spin(st):
        _lib.call("itg_stream_spin", int(spin_us), C.c_void_p(st.cuda_stream))

    def together(streams):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for st in streams:
            st.wait_event(e0)
        spin(cur)
        for st in streams:
            spin(st)
        for st in streams:
            cur.wait_stream(st)
        e1.record(cur)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        if _STREAM_DEBUG:
            print("[streams] %d spins of %d us on %d streams at once: %.0f us" % (1 + len(streams), spin_us, 1 + len(streams), us),
                  flush=True)
        return us < 1.6 * spin_us

    spin(cur)
    torch.cuda.synchronize(device)
    chosen = []
    for _ in range(candidates):
        if len(chosen) >= min(want, 3):
            break
        cand = torch.cuda.Stream(device=device)      # (stream priorities: the range here is (0, -1); high-priority side or main streams change nothing)
        together([cand])                                 # first use of a stream: its queue is set up now, not timed
        if together(chosen + [cand]):
            chosen.append(cand)
    found = len(chosen)
    if not chosen:
        chosen = [torch.cuda.Stream(device=device)]
    while len(chosen) < want:
        chosen.append(chosen[-1])
    STREAM_PLACEMENT.update(want=min(want, 3), concurrent=found, probed=True)
    if found < min(want, 3):
        # a loaded host or a tracing profiler stretches the spin timings and makes every candidate fail: the step is then
        # correct but partly serialised - say so once instead of silently benchmarking another schedule
        import sys
        print("[itg] stream placement: only %d of %d side streams run concurrently with the main stream "
              "(ITG_DEBUG=2 prints the probe timings)" % (found, min(want, 3)), file=sys.stderr, flush=True)
    else:
        _PLACED[ckey] = list(chosen)
    return chosen


_PLACED = {}               # (device, origin stream, want) -> the streams a successful probe chose
STREAM_PLACEMENT = {"want": 0, "concurrent": 0, "probed": False}    # what the last probe found (bench.py reports it)


_wgrad_dirty = []          # weight-gradient streams that carry forked work nobody has waited for yet

# ---- the capture rule (ROCm 7.2): hipStreamEndCapture segfaults when a forked stream that holds NO node of the capture yet
# (it entered only by waiting) is itself waited for (tools/capture_nested_fork.py; rounds 1 and 3 both hit it through a
# schedule change: a join of the weight-gradient streams into the D(real) branch while they were idle, 82b2aa7).  While an
# engine records (engine.Trainer.capture sets _lib.CAPTURE_LOG to the streams that received a launch, seeded with the origin),
# every cross-stream wait of the package asks capture_rule() first; an illegal wait is NOT recorded (the capture stays
# valid) and is reported as a Python error after the capture has been closed cleanly.
CAPTURE_ERRORS = []


def capture_rule(waited, what):
    """True if waiting for ``waited`` (a torch stream) may be recorded.  Outside an engine capture: always."""
    log = _lib.CAPTURE_LOG
    if log is None or (waited.cuda_stream or 0) in log:
        return True
    CAPTURE_ERRORS.append("%s: stream %#x is waited for but holds no node of this capture" % (what, waited.cuda_stream))
    return False


def join_stream(waited, what, waiter=None):
    """``waiter`` (default: the current stream) waits for everything queued on ``waited`` - subject to the capture rule."""
    if capture_rule(waited, what):
        (waiter if waiter is not None else torch.cuda.current_stream()).wait_stream(waited)


def wgrad_streams_join():
    """The current stream waits for every weight-gradient stream that carries un-joined work (and only for those).

    Waiting for an IDLE forked stream is not just useless: under hipGraph capture it is what crashed
    hipStreamEndCapture on ROCm 7.2 (tools/capture_nested_fork.py: a stream that entered the capture by waiting for the
    origin and holds no node yet, waited for by a second forked stream which then forks kernels back onto it -
    reproducible with plain torch ops).  A stream is therefore only ever waited for after work was forked onto it, and
    it enters a capture only through the event of the kernel that forks onto it."""
    if not _wgrad_dirty:
        return
    cur = torch.cuda.current_stream()
    for s_ in _wgrad_dirty:
        join_stream(s_, "wgrad_streams_join", cur)
    del _wgrad_dirty[:]


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ZeroArena:
    """fp64 scratch that is zeroed ONCE per train step (engine.Trainer) and handed out in slices to the
    BatchNorm statistics kernels, replacing one memset launch per BatchNorm call (26 per step).  The
    slices of a step are taken in a fixed order, so the layout is identical under hipGraph replay."""

    def __init__(self, device, doubles=1 << 16):
        self.buf = torch.zeros(doubles, device=device, dtype=torch.float64)
        self.cur = 0

    def reset(self):
        self.buf.zero_()
        self.cur = 0

    def take(self, n):
        if self.cur + n > self.buf.numel():
            return None
        t = self.buf[self.cur:self.cur + n]
        self.cur += n
        return t


ARENA = None      # set by engine.Trainer for the duration of a step

# Round 6 (itg.h itg_bn_bwd_fuse): the conv that consumed y = act(BatchNorm(x)) hands the BatchNorm to its input gradient; when
# that launch has a split-K second stage (the generator's wide layers) the stage also accumulates the BatchNorm's backward sums and
# the BatchNorm's backward skips its reduce launch - one dependent launch fewer per BatchNorm on the latency-bound backward chain.
# ITG_STATS_PATHS bit 3 (default on).  _BN_FUSE: y.data_ptr() -> weak references to (x, stat, y) + (act, slope, c), registered
# by _BNAct.forward and consumed by _Conv.backward (an entry whose y has died - its pointer may belong to another tensor by
# then - dereferences to None and is ignored); _BN_SUMS: dx.data_ptr() -> (sums, dx, x.data_ptr()), left by _Conv.backward for the
# _BNAct.backward that follows it; it holds dx, so that pointer cannot be re-used while the entry exists.
BN_BWD_FUSE = (int(os.environ.get("ITG_STATS_PATHS", "13")) & 8) != 0
_BN_FUSE, _BN_SUMS = {}, {}

# Test hook (tests/test_gpu_fullsize.py): a list that receives (output tensor, channels, op, halo_rows) of every LeakyReLU / ReLU this module
# applies - BatchNorm + activation, the conv epilogues, the pointwise op - in host issue order.  The signs of these tensors are
# the branches the step took; the test puts the fp64 oracle on the same branches (oracle.nets.ACT_REPLAY).
ACT_SINK = None


def _sink_act(y, c, act, op, halo_rows=False):
    if ACT_SINK is not None and act == ACT_LRELU:
        ACT_SINK.append((y, c, op, halo_rows))      # halo_rows: band layout, rows 0 and -1 are the neighbours' (not this op's output)


def _zeros_f64(n, device):
    t = ARENA.take(n) if ARENA is not None and ARENA.buf.device == device else None
    return t if t is not None else torch.zeros(n, device=device, dtype=torch.float64)


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, c, training, eps, momentum, act, slope, ups, sync, sinks=None, pre_sums=None,
                virt_ups=False, fork=False, pad_rows=False):
        ctx.set_materialize_grads(False)
        x_in = x
        x = x.contiguous()
        n, gh, gw, ph, pw, ld = x.shape
        dev = x.device
        st = _stream()
        dx_ = _desc(x, c)
        count = float(x.numel() // ld)
        sums = None
        if training:
            if pre_sums is not None:      # accumulated by the epilogue of the conv that produced x
                sums = pre_sums
            else:
                sums = _zeros_f64(2 * ld, dev)
                _lib.call("itg_bn_stats", C.byref(dx_), _ptr(sums), st)
            if sync is not None and _active(sync):
                sync.all_reduce(sums)
                count = sync.global_count(count)
        stat = torch.empty(4 * ld, device=dev, dtype=torch.float32)
        mean_rstd, ab = stat[:2 * ld], stat[2 * ld:]
        s = 2 if ups else 1
        if pad_rows:
            # the band layout of a row-sharded grid (include/itg.h "halo rows of a row-sharded band"): rows 1 .. ph are written
            # here, rows 0 / ph + 1 by ops.band_halo; the backward receives the gradient in the same layout
            if ups or gh != 1 or gw != 1:
                raise _lib.ItgError("pad_rows: image-layout bands only, no materialised upsample")
            y = torch.empty((n, 1, 1, ph + 2, pw, ld), device=dev, dtype=torch.float32)
        else:
            y = torch.empty((n, gh, gw, ph * s, pw * s, ld), device=dev, dtype=torch.float32)
        dy_ = _desc(y, c)
        if training:      # coefficients, running statistics and the normalised output in one launch
            _lib.call("itg_bn_finalize_apply", C.byref(dx_), _ptr(sums), count, 4.0 if (ups or virt_ups) else 1.0, _ptr(gamma), _ptr(beta),
                      float(eps), float(momentum), _ptr(rm), _ptr(rv), _ptr(nbt), _ptr(mean_rstd), _ptr(ab), C.byref(dy_),
                      act, float(slope), st)
        else:
            _lib.call("itg_bn_finalize", _ptr(sums), count, 4.0 if (ups or virt_ups) else 1.0, _ptr(gamma), _ptr(beta), float(eps),
                      float(momentum), _ptr(rm), _ptr(rv), _ptr(nbt), _ptr(mean_rstd), _ptr(ab), c, ld, int(training), st)
            _lib.call("itg_bn_apply", C.byref(dx_), _ptr(ab), C.byref(dy_), act, float(slope), st)
        ctx.meta = (c, act, slope, count, sync, training, gamma is not None)
        ctx.sinks = sinks
        ctx.save_for_backward(x, stat)
        if BN_BWD_FUSE and training and not ups and not pad_rows and ctx.needs_input_grad[0]:
            if len(_BN_FUSE) > 256:        # outputs that no conv's backward consumed: drop the stale (weak) entries
                _BN_FUSE.clear()
            _BN_FUSE[y.data_ptr()] = (weakref.ref(x), weakref.ref(stat), act, slope, c, weakref.ref(y))
        _sink_act(y, c, act, "bn", pad_rows)
        if fork:
            # second output: x itself (an alias).  Whatever consumes it (the block's residual shortcut) hands its gradient to
            # THIS backward, which adds it inside itg_bn_bwd_apply_add - instead of autograd summing the two gradients of x
            # with an add launch of its own
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    def backward(ctx, dy, g_alias=None):
        x, stat = ctx.saved_tensors
        if dy is None:          # only the alias was used downstream
            return (g_alias,) + (None,) * 18
        c, act, slope, count, sync, training, affine = ctx.meta
        if not training:
            raise _lib.ItgError("BatchNorm backward is only implemented for training-mode statistics")
        ld = x.shape[5]
        mean_rstd, ab = stat[:2 * ld], stat[2 * ld:]
        dy = dy.contiguous()
        st = _stream()
        dx_, ddy_ = _desc(x, c), _desc(dy, c)
        ent = _BN_SUMS.pop(dy.data_ptr(), None) if _BN_SUMS else None
        if ent is not None and ent[1].shape == x.shape and ent[2] == x.data_ptr():
            sums = ent[0]                   # accumulated by the split-K second stage of the input gradient that produced dy
        else:
            sums = _zeros_f64(2 * ld, x.device)
            _lib.call("itg_bn_bwd_reduce", C.byref(dx_), C.byref(ddy_), _ptr(ab), _ptr(mean_rstd), act, float(slope),
                      _ptr(sums), st)
        if sync is not None and _active(sync):
            # dgamma/dbeta are the LOCAL sums (the gradient all-reduce adds the ranks up later);
            # the dx formula needs the GLOBAL ones.
            local = sums.clone()
            sync.all_reduce(sums)
        else:
            local = sums
        gx = torch.empty_like(x)
        dgx = _desc(gx, c)
        dg = db = None
        acc = 0
        if affine:
            if ctx.sinks is not None:
                dg, db = ctx.sinks
                acc = 1
            else:
                dg = torch.empty(c, device=x.device, dtype=torch.float32)
                db = torch.empty(c, device=x.device, dtype=torch.float32)
        if g_alias is not None:
            g_alias = g_alias.contiguous()
            dadd = _desc(g_alias, c)
            _lib.call("itg_bn_bwd_apply_add", C.byref(dx_), C.byref(ddy_), _ptr(ab), _ptr(mean_rstd), _ptr(local), _ptr(sums),
                      count, act, float(slope), C.byref(dgx), _ptr(dg), _ptr(db), acc, C.byref(dadd), st)
        else:
            _lib.call("itg_bn_bwd_apply", C.byref(dx_), C.byref(ddy_), _ptr(ab), _ptr(mean_rstd), _ptr(local), _ptr(sums),
                      count, act, float(slope), C.byref(dgx), _ptr(dg), _ptr(db), acc, st)
        if acc:
            dg = db = None
        return gx, dg, db, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def bn_act(x, gamma, beta, rm, rv, nbt, training=True, eps=1e-5, momentum=0.1, act=ACT_NONE, slope=0.0,
           upsample=False, sync=None, sinks=None, consumer_upsamples=False, fork=False, pad_rows=False):
    """y = act(BatchNorm(x)) [nearest-upsampled x2 when ``upsample``]: statistics are taken on x
    (identical to those of the upsampled tensor), the unbiased running_var uses the x4 count.
    ``consumer_upsamples``: y stays at x's size and the conv that reads it folds the x2 upsample into its filter
    (ops.conv(up2=True)); the reference normalises the upsampled tensor, so running_var still takes the x4 count."""
    if fork and training and torch.is_grad_enabled() and x.t.requires_grad:
        # ``fork``: the result carries ``.fork``, an alias of x; a consumer that reads x through it (the residual shortcut of
        # a generator block) gets its gradient summed into the BatchNorm's own input gradient by the backward kernel
        t, alias = _BNAct.apply(x.t, gamma, beta, rm, rv, nbt, x.c, training, eps, momentum, act, slope, upsample, sync, sinks,
                                x.stats if training else None, bool(consumer_upsamples), True, bool(pad_rows))
        y = GT(t, x.c)
        y.fork = GT(alias, x.c)
        y.padded = bool(pad_rows)
        return y
    t = _BNAct.apply(x.t, gamma, beta, rm, rv, nbt, x.c, training, eps, momentum, act, slope, upsample, sync, sinks,
                     x.stats if training else None, bool(consumer_upsamples), False, bool(pad_rows))
    y = GT(t, x.c)
    y.padded = bool(pad_rows)         # ``pad_rows``: y.t is (n, 1, 1, ph + 2, pw, ld) with its halo rows still to be filled (band_halo)
    return y


def bn_stats_only(x, rm, rv, nbt, training=True, eps=1e-5, momentum=0.1, sync=None):
    """Affine-free BN statistics for SSM: returns mean_rstd (2*ld) and updates running stats."""
    t = x.t.contiguous()
    ld = t.shape[5]
    st = _stream()
    dx_ = _desc(t, x.c)
    count = float(t.numel() // ld)
    sums = None
    if training:
        sums = torch.zeros(2 * ld, device=t.device, dtype=torch.float64)
        _lib.call("itg_bn_stats", C.byref(dx_), _ptr(sums), st)
        if sync is not None and _active(sync):
            sync.all_reduce(sums)
            count = sync.global_count(count)
    stat = torch.empty(4 * ld, device=t.device, dtype=torch.float32)
    _lib.call("itg_bn_finalize", _ptr(sums), count, 1.0, None, None, float(eps), float(momentum), _ptr(rm), _ptr(rv),
              _ptr(nbt), _ptr(stat[:2 * ld]), _ptr(stat[2 * ld:]), x.c, ld, int(training), st)
    return stat, count


# ------------------------------------------------------------------------------- SSM modulation
class _SSM(torch.autograd.Function):
    """y = act((1+gamma)*xhat + beta), xhat = affine-free BN(x), [gamma,beta] = emb halves.
    reference models/layers.py:228-234."""

    @staticmethod
    def forward(ctx, x, emb, rm, rv, nbt, c, training, eps, momentum, act, slope, sync):
        x, emb = x.contiguous(), emb.contiguous()
        stat, count = bn_stats_only(GT(x, c), rm, rv, nbt, training, eps, momentum, sync)
        ld = x.shape[5]
        y = torch.empty_like(x)
        a, e, o = _desc(x, c), _desc(emb, 2 * c), _desc(y, c)
        _lib.call("itg_ssm_modulate_fwd", C.byref(a), _ptr(stat[:2 * ld]), C.byref(e), C.byref(o), act, float(slope),
                  _stream())
        ctx.meta = (c, act, slope, count, sync, training)
        ctx.save_for_backward(x, emb, stat)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, emb, stat = ctx.saved_tensors
        c, act, slope, count, sync, training = ctx.meta
        ld = x.shape[5]
        st = _stream()
        dy = dy.contiguous()
        dxhat = torch.empty_like(x)
        demb = torch.zeros_like(emb)
        a, e, g = _desc(x, c), _desc(emb, 2 * c), _desc(dy, c)
        h, de = _desc(dxhat, c), _desc(demb, 2 * c)
        mean_rstd = stat[:2 * ld]
        _lib.call("itg_ssm_modulate_bwd", C.byref(a), _ptr(mean_rstd), C.byref(e), C.byref(g), act, float(slope),
                  C.byref(h), C.byref(de), st)
        if not training:
            raise _lib.ItgError("SSM backward is only implemented for training-mode statistics")
        # affine-free BN backward on dxhat: alpha = rstd, beta' = -mean*rstd are stat[2ld:]
        ab = stat[2 * ld:]
        sums = torch.zeros(2 * ld, device=x.device, dtype=torch.float64)
        _lib.call("itg_bn_bwd_reduce", C.byref(a), C.byref(h), _ptr(ab), _ptr(mean_rstd), ACT_NONE, 0.0, _ptr(sums), st)
        if sync is not None and _active(sync):
            sync.all_reduce(sums)
        gx = torch.empty_like(x)
        dgx = _desc(gx, c)
        _lib.call("itg_bn_bwd_apply", C.byref(a), C.byref(h), _ptr(ab), _ptr(mean_rstd), None, _ptr(sums), count,
                  ACT_NONE, 0.0, C.byref(dgx), None, None, 0, st)
        return gx, demb, None, None, None, None, None, None, None, None, None, None


def ssm_modulate(x, emb, rm, rv, nbt, training=True, eps=1e-5, momentum=0.1, act=ACT_NONE, slope=0.0, sync=None):
    t = _SSM.apply(x.t, emb.t, rm, rv, nbt, x.c, training, eps, momentum, act, slope, sync)
    return GT(t, x.c)


# ------------------------------------------------------------------------------- pointwise
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, act, slope):
        x = x.contiguous()
        y = torch.empty_like(x)
        a, b = _desc(x, c), _desc(y, c)
        _lib.call("itg_act_fwd", C.byref(a), C.byref(b), act, float(slope), _stream())
        ctx.meta = (c, act, slope)
        ctx.save_for_backward(y)
        _sink_act(y, c, act, "act")
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        c, act, slope = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        a, b, d = _desc(y, c), _desc(dy, c), _desc(dx, c)
        _lib.call("itg_act_bwd", C.byref(a), C.byref(b), C.byref(d), act, float(slope), _stream())
        return dx, None, None, None


def act(x, kind, slope=0.0):
    return GT(_Act.apply(x.t, x.c, kind, slope), x.c)


class _Up(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        x = x.contiguous()
        n, gh, gw, ph, pw, ld = x.shape
        y = torch.empty((n, gh, gw, 2 * ph, 2 * pw, ld), device=x.device, dtype=torch.float32)
        a, b = _desc(x, c), _desc(y, c)
        _lib.call("itg_upsample2x_fwd", C.byref(a), C.byref(b), _stream())
        ctx.c = c
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        n, gh, gw, ph, pw, ld = dy.shape
        dx = torch.empty((n, gh, gw, ph // 2, pw // 2, ld), device=dy.device, dtype=torch.float32)
        a, b = _desc(dy, ctx.c), _desc(dx, ctx.c)
        _lib.call("itg_upsample2x_bwd", C.byref(a), C.byref(b), _stream())
        return dx, None


def upsample2x(x):
    return GT(_Up.apply(x.t, x.c), x.c)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        a, b = a.contiguous(), b.contiguous()
        o = torch.empty_like(a)
        da, db, do = _desc(a, c), _desc(b, c), _desc(o, c)
        _lib.call("itg_add", C.byref(da), C.byref(db), C.byref(do), _stream())
        return o

    @staticmethod
    def backward(ctx, g):
        return g, g, None


def add(a, b):
    return GT(_Add.apply(a.t, b.t, a.c), a.c)


class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        x = x.contiguous()
        n, gh, gw, ph, pw, ld = x.shape
        y = torch.empty((n, gh, gw, ph // 2, pw // 2, ld), device=x.device, dtype=torch.float32)
        a, b = _desc(x, c), _desc(y, c)
        _lib.call("itg_maxpool2_fwd", C.byref(a), C.byref(b), _stream())
        ctx.c = c
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        a, b, g, d = _desc(x, ctx.c), _desc(y, ctx.c), _desc(dy, ctx.c), _desc(dx, ctx.c)
        _lib.call("itg_maxpool2_bwd", C.byref(a), C.byref(b), C.byref(g), C.byref(d), _stream())
        return dx, None


def maxpool2(x):
    return GT(_Pool.apply(x.t, x.c), x.c)


# ------------------------------------------------------------------------------- attention core
class _Att(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, phi, g, c8, c2):
        theta, phi, g = theta.contiguous(), phi.contiguous(), g.contiguous()
        n, gh, gw, ph, pw, _ = theta.shape
        nb, hw, j = n * gh * gw, ph * pw, phi.shape[3] * phi.shape[4]
        o = torch.empty((n, gh, gw, ph, pw, g.shape[5]), device=theta.device, dtype=torch.float32)
        a, b, c_, d = _desc(theta, c8), _desc(phi, c8), _desc(g, c2), _desc(o, c2)
        beta = torch.empty(_lib.fn("itg_attention_scratch_floats")(C.byref(a), C.byref(b), C.byref(c_)), device=theta.device,
                           dtype=torch.float32)
        _lib.call("itg_attention_fwd", C.byref(a), C.byref(b), C.byref(c_), C.byref(d), _ptr(beta), _stream())
        ctx.meta = (c8, c2)
        ctx.save_for_backward(theta, phi, g, beta)
        return o

    @staticmethod
    def backward(ctx, do):
        theta, phi, g, beta = ctx.saved_tensors
        c8, c2 = ctx.meta
        do = do.contiguous()
        dth, dph, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
        a, b, c_, d = _desc(theta, c8), _desc(phi, c8), _desc(g, c2), _desc(do, c2)
        e, f, h = _desc(dth, c8), _desc(dph, c8), _desc(dg, c2)
        _lib.call("itg_attention_bwd", C.byref(a), C.byref(b), C.byref(c_), _ptr(beta), C.byref(d), C.byref(e),
                  C.byref(f), C.byref(h), _stream())
        return dth, dph, dg, None, None


def attention_core(theta, phi_pooled, g_pooled):
    t = _Att.apply(theta.t, phi_pooled.t, g_pooled.t, theta.c, g_pooled.c)
    return GT(t, g_pooled.c)


# ------------------------------------------------------------------------------- LocalPadder (NCHW API form)
class _LocalPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gh, gw, pad_mode, merged):
        _req_cuda(x, "LocalPadder input")
        x = x.contiguous()
        if merged:
            n, c = x.shape[0], x.shape[1]
            p = x.shape[3] // gw
        else:
            n, c, p = x.shape[0] // (gh * gw), x.shape[1], x.shape[3]
        y = torch.empty((n * gh * gw, c, p + 2, p + 2), device=x.device, dtype=torch.float32)
        _lib.call("itg_local_pad_fwd", _ptr(x), _ptr(y), n, c, gh, gw, p, pad_mode, int(merged), _stream())
        ctx.meta = (tuple(x.shape), n, c, gh, gw, p, pad_mode, merged)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, n, c, gh, gw, p, pad_mode, merged = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty(shape, device=dy.device, dtype=torch.float32)
        _lib.call("itg_local_pad_bwd", _ptr(dy), _ptr(dx), n, c, gh, gw, p, pad_mode, int(merged), _stream())
        return dx, None, None, None, None


def local_pad_nchw(x, gh, gw, pad_mode=PAD_REPLICATE, merged=False):
    return _LocalPad.apply(x, gh, gw, pad_mode, merged)


def local_pad_grid(x, pad_mode=PAD_REPLICATE, left=None, top=None, bottom=None):
    """Inference-side halo gather on GT tensors with optional carried left column / top row (streaming)
    and top / bottom halo rows received from neighbouring ranks (row-sharded patch grid)."""
    t = x.t.contiguous()
    n, gh, gw, ph, pw, ld = t.shape
    y = torch.empty((n, gh, gw, ph + 2, pw + 2, ld), device=t.device, dtype=torch.float32)
    a, b = _desc(t, x.c), _desc(y, x.c)
    _lib.call("itg_local_pad_stream_fwd", C.byref(a), _ptr(left), _ptr(top), _ptr(bottom), C.byref(b), pad_mode,
              _stream())
    return GT(y, x.c)


def pack_tables(jobs, device):
    """jobs: list of (w OIHW tensor, out buffer, co, ci, ld, kh, kw, stride, dgrad) -> list of
    (device int64 table, n, total) for itg_pack_multi, one per _lib.PACK_MAX_JOBS panels."""
    tables = []
    for i in range(0, len(jobs), _lib.PACK_MAX_JOBS):
        rows, start = [], 0
        for (w, o, co, ci, ld, kh, kw, stride, dg) in jobs[i:i + _lib.PACK_MAX_JOBS]:
            rows.append([w.data_ptr(), o.data_ptr(), co, ci, ld, kh, kw, stride, dg, start])
            start += o.numel()
        tables.append((torch.tensor(rows, dtype=torch.int64).to(device), len(rows), start))
    return tables


def pack_multi(tables):
    st = _stream()
    for t, n, total in tables:
        _lib.call("itg_pack_multi", _ptr(t), n, total, st)


def pack_sizes(co, ci, kh, kw, stride, up2=False, wino=False):
    """(floats of the forward panel, floats of the dgrad panel) for a conv between patch-grid tensors."""
    if wino == 2:
        return (_lib.fn("itg_pack_wino_s2_size")(co, ld_for(ci)), _lib.fn("itg_pack_dgrad_size")(ci, ld_for(co), kh, kw, stride))
    if wino:
        f = _lib.fn("itg_pack_wino3_size" if kh == 3 else "itg_pack_wino_size")
        return (f(co, ld_for(ci)), f(ci, ld_for(co)))
    if up2:
        return (_lib.fn("itg_pack_up2_fwd_size")(co, ld_for(ci)), _lib.fn("itg_pack_up2_dgrad_size")(ci, ld_for(co)))
    return (_lib.fn("itg_pack_fwd_size")(co, ld_for(ci), kh, kw),
            _lib.fn("itg_pack_dgrad_size")(ci, ld_for(co), kh, kw, stride))


# ------------------------------------------------------------------------------- row-sharded grids (training)
class _BandHalo(torch.autograd.Function):
    """In place: fill the two halo rows of a band in the padded image layout (n, 1, 1, H + 2, W, ld) - the neighbour ranks'
    boundary rows over ``comm`` (dist.RowHalo), the replicate / zero padding at the image's outer border - reference
    LocalPadder.forward, models/layers.py:145-173, for a patch grid sharded by patch rows.  Backward, in place on the gradient
    of that tensor: the halo rows' gradients travel back the same way and are added onto rows 1 / H (itg_band_halo_grad).
    One small launch each way on one rank, three with neighbours (rows out, exchange, rows in): no concatenated copy of the
    band, no slicing of its gradient (the torch.cat path of rounds 2-4 spent 12.8 % of config 4's kernel time on both)."""

    @staticmethod
    def forward(ctx, ext, c, comm, replicate):
        _req_cuda(ext, "ext")
        n, _, _, H2, W, ld = ext.shape
        st = _stream()
        d = _desc(ext, c)
        top = bottom = None
        if comm is not None and comm.world > 1:
            first, last = torch.empty((n, W, ld), device=ext.device), torch.empty((n, W, ld), device=ext.device)
            _lib.call("itg_band_rows_get", C.byref(d), 1, H2 - 2, _ptr(first), _ptr(last), st)
            top, bottom = comm.exchange(first, last)
        border = 1 if replicate else 2
        ctx.modes = (0 if top is not None else border, 0 if bottom is not None else border)
        _lib.call("itg_band_halo_fill", C.byref(d), _ptr(top), _ptr(bottom), ctx.modes[0], ctx.modes[1], st)
        ctx.comm, ctx.c = comm, c
        ctx.mark_dirty(ext)
        return ext

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n, _, _, H2, W, ld = g.shape
        st = _stream()
        d = _desc(g, ctx.c)
        above = below = None
        if 0 in ctx.modes:
            up, down = torch.zeros((n, W, ld), device=g.device), torch.zeros((n, W, ld), device=g.device)
            _lib.call("itg_band_rows_get", C.byref(d), 0, H2 - 1, _ptr(up) if ctx.modes[0] == 0 else None,
                      _ptr(down) if ctx.modes[1] == 0 else None, st)
            above, below = ctx.comm.exchange(up, down)
        _lib.call("itg_band_halo_grad", C.byref(d), _ptr(above), _ptr(below), ctx.modes[0], ctx.modes[1], st)
        return g, None, None, None


def band_halo(x, comm, replicate):
    """x: GT whose tensor is a band in the padded layout with rows 1 .. H written (ops.bn_act(pad_rows=True), band_extend)."""
    return GT(_BandHalo.apply(x.t, x.c, comm, bool(replicate)), x.c)


class _BandExtend(torch.autograd.Function):
    """(n, 1, 1, H, W, ld) -> the padded layout with rows 1 .. H copied (halo rows unwritten): for producers that cannot write
    the padded layout themselves (SSM modulation, attention); BatchNorm writes it directly (bn_act(pad_rows=True))."""

    @staticmethod
    def forward(ctx, x, c):
        _req_cuda(x, "x")
        x = x.contiguous()
        n, g1, g2, H, W, ld = x.shape
        if g1 != 1 or g2 != 1:
            raise _lib.ItgError("band_extend expects the image layout (1x1 grid)")
        ext = torch.empty((n, 1, 1, H + 2, W, ld), device=x.device, dtype=torch.float32)
        _lib.call("itg_band_interior_copy", C.byref(_desc(x, c)), C.byref(_desc(ext, c)), 1, _stream())
        ctx.c = c
        return ext

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n, _, _, H2, W, ld = g.shape
        gx = torch.empty((n, 1, 1, H2 - 2, W, ld), device=g.device, dtype=torch.float32)
        _lib.call("itg_band_interior_copy", C.byref(_desc(gx, ctx.c)), C.byref(_desc(g, ctx.c)), 0, _stream())
        return gx, None


def band_extend(x):
    y = GT(_BandExtend.apply(x.t, x.c), x.c)
    y.padded = True
    return y


class _HaloExchange(torch.autograd.Function):
    """(first_row, last_row) of this rank's band -> (row above, row below).  The backward is the same
    exchange applied to the halo gradients: the gradient of the row this rank sent up comes back as
    the upper neighbour's bottom-halo gradient, and vice versa."""

    @staticmethod
    def forward(ctx, first, last, comm, defer_wait=False):
        ctx.comm = comm
        if defer_wait:      # the consumer calls comm.wait() before it reads the rows (interior-first convs)
            top, bottom = comm.exchange(first.contiguous(), last.contiguous(), defer_wait=True)
        else:
            top, bottom = comm.exchange(first.contiguous(), last.contiguous())
        ctx.has = (top is not None, bottom is not None)
        ctx.shape = tuple(first.shape)
        return (top if top is not None else first.new_zeros(0), bottom if bottom is not None else first.new_zeros(0))

    @staticmethod
    def backward(ctx, dtop, dbottom):
        z = lambda: torch.zeros(ctx.shape, device=dtop.device if dtop.numel() else dbottom.device, dtype=torch.float32)  # noqa: E731
        up = dtop.contiguous() if ctx.has[0] else z()
        down = dbottom.contiguous() if ctx.has[1] else z()
        from_above, from_below = ctx.comm.exchange(up, down)
        return (from_above if from_above is not None else z(), from_below if from_below is not None else z(), None, None)


def halo_exchange(first, last, comm, defer_wait=False):
    """-> (top, bottom) with None at the grid's outer border; differentiable.  ``defer_wait``: the transfers are only posted;
    the caller runs whatever does not need the rows and then calls ``comm.wait()``."""
    top, bottom = _HaloExchange.apply(first, last, comm, defer_wait)
    return (top if top.numel() else None), (bottom if bottom.numel() else None)


class _GatherRows(torch.autograd.Function):
    """Bands (n, c, h_r, W) of all ranks -> full images (n, c, H, W) on every rank.  Backward: the
    gradient of this rank's band summed over ranks (a reduce-scatter when the bands are equal)."""

    @staticmethod
    def forward(ctx, band, comm, heights):
        parts = comm.all_gather(band.contiguous(), heights)
        ctx.comm = comm
        ctx.heights = [p.shape[-2] for p in parts]
        return torch.cat(parts, -2)

    @staticmethod
    def backward(ctx, g):
        return ctx.comm.reduce_scatter_rows(g.contiguous(), ctx.heights), None, None


def gather_rows(band, comm, heights=None):
    """``heights``: every rank's band height when static (no size exchange / host sync per step)."""
    return _GatherRows.apply(band, comm, heights)


# ------------------------------------------------------------------------------- losses
class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        _req_cuda(logits, "logits")
        logits = logits.contiguous()
        out = torch.empty((), device=logits.device, dtype=torch.float32)
        _lib.call("itg_bce_logits_fwd", _ptr(logits), logits.numel(), float(target), _ptr(out), _stream())
        ctx.target = float(target)
        ctx.save_for_backward(logits)
        return out

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        g = g.contiguous()
        d = torch.empty_like(logits)
        _lib.call("itg_bce_logits_bwd", _ptr(logits), logits.numel(), ctx.target, _ptr(g), _ptr(d), _stream())
        return d, None


def bce_with_logits(logits, target):
    """Mean BCE-with-logits against a constant target (reference train.py:81,131-132)."""
    return _BCE.apply(logits, target)


class _Hinge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, mode):
        _req_cuda(logits, "logits")
        logits = logits.contiguous()
        out = torch.empty((), device=logits.device, dtype=torch.float32)
        _lib.call("itg_hinge_fwd", _ptr(logits), logits.numel(), int(mode), _ptr(out), _stream())
        ctx.mode = int(mode)
        ctx.save_for_backward(logits)
        return out

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        d = torch.empty_like(logits)
        _lib.call("itg_hinge_bwd", _ptr(logits), logits.numel(), ctx.mode, _ptr(g.contiguous()), _ptr(d), _stream())
        return d, None


def hinge(logits, mode):
    """mode: 'd_real' | 'd_fake' | 'g' (build-side extra; not in the reference)."""
    return _Hinge.apply(logits, {"d_real": 0, "d_fake": 1, "g": 2}[mode])


UNIT_GRAD = None       # engine.Trainer: the device tensor 1.0 it seeds every backward pass with (recognised by address)


class _LogitLossGrid(torch.autograd.Function):
    """The loss head on the discriminator's logit map in patch-grid layout (c == 1): mean BCE-with-logits / hinge and its
    derivative in ONE launch (itg_logit_loss_grid) - reference train.py:131-132,148-149,164-165 on a (n, 1, h, w) map; the mean
    does not care about the layout.  The generic heads (bce_with_logits / hinge on NCHW logits) cost grid_to_nchw + a
    single-workgroup sum + the backward + nchw_to_grid: four latency-bound launches on every D pass's critical path."""

    @staticmethod
    def forward(ctx, t, c, kind, target):
        _req_cuda(t, "logits")
        if c != 1:
            raise _lib.ItgError("logit_loss: a one-channel logit map expected, got %d channels" % c)
        t = t.contiguous()
        out = torch.empty((), device=t.device, dtype=torch.float32)
        dl = torch.empty_like(t)
        ws = _zeros_f64(int(_lib.fn("itg_logit_loss_grid_workspace")()), t.device)
        a, b = _desc(t, 1), _desc(dl, 1)
        _lib.call("itg_logit_loss_grid", C.byref(a), int(kind), float(target), _ptr(out), C.byref(b), _ptr(ws), _stream())
        ctx.save_for_backward(dl)
        return out

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if UNIT_GRAD is not None and g.data_ptr() == UNIT_GRAD.data_ptr():
            return dl, None, None, None            # seeded with 1.0: the stored derivative is the gradient
        return dl * g, None, None, None


_LOSS_KINDS = {"bce": 0, "d_real": 1, "d_fake": 2, "g": 3}


def logit_loss(logits, kind, target=0.0):
    """logits: GT with one channel; kind: 'bce' (against the constant ``target``) or a hinge mode 'd_real' | 'd_fake' | 'g'."""
    return _LogitLossGrid.apply(logits.t, logits.c, _LOSS_KINDS[kind], target)


# ------------------------------------------------------------------------------- flat helpers
def axpby(x, y, a, b, a_dev=None, out=None):
    """out = a*(a_dev)*x + b*y over flat fp32 buffers."""
    x, y = x.contiguous(), y.contiguous()
    if out is None:
        out = torch.empty_like(x)
    _lib.call("itg_axpby", _ptr(x), _ptr(y), _ptr(out), float(a), _ptr(a_dev), float(b), x.numel(), _stream())
    return out


def dot(x, y):
    """<x, y> accumulated in fp64 -> 1-element fp64 tensor."""
    out = torch.empty(1, device=x.device, dtype=torch.float64)
    _lib.call("itg_dot", _ptr(x.contiguous()), _ptr(y.contiguous()), x.numel(), _ptr(out), _stream())
    return out


# ------------------------------------------------------------------------------- optimiser
def adam_ema_step(p, g, m, v, ema, lr, beta1, beta2, eps, step, ema_decay=0.999, step_dev=None):
    """Fused Adam (+EMA) over flat fp32 buffers, in place.  ``step_dev``: int32 device scalar holding the
    step count (graph-replay safe); else the host integer ``step``."""
    _lib.call("itg_adam_ema_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), int(step), _ptr(step_dev), float(ema_decay), _stream())
