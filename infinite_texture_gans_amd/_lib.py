"""ctypes binding of libitg_hip.so (the C ABI declared in include/itg.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/build.sh``; there is
NO fallback: if it is missing, or a call is rejected, the product raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ITG_LIB", os.path.join(_HERE, "libitg_hip.so"))   # ITG_LIB: A/B another build of the same ABI

PAD_ZERO, PAD_REPLICATE = 0, 1
ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2


class Tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int32), ("gh", C.c_int32), ("gw", C.c_int32),
                ("ph", C.c_int32), ("pw", C.c_int32), ("c", C.c_int32), ("ld", C.c_int32)]


PREC_F32, PREC_BF16 = 0, 1      # itg.h ITG_PREC_*


class BnBwdFuse(C.Structure):
    """itg_bn_bwd_fuse: the BatchNorm in front of a conv, handed to the conv's input gradient (itg.h)."""
    _fields_ = [("x", C.POINTER(Tensor)), ("ab", C.c_void_p), ("mean_rstd", C.c_void_p), ("act", C.c_int32), ("slope", C.c_float),
                ("sums", C.c_void_p), ("taken", C.c_int32), ("reserved", C.c_int32)]


class ConvGeom(C.Structure):
    _fields_ = [("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("pad_mode", C.c_int32), ("pad_h", C.c_int32), ("precision", C.c_int32), ("up2", C.c_int32),
                ("out_stats", C.c_void_p), ("bn_bwd", C.POINTER(BnBwdFuse)), ("flags", C.c_int32), ("reserved", C.c_int32),
                ("wino_v", C.c_void_p)]

    def __init__(self, kh, kw, stride, pad, pad_mode, pad_h=-1, precision=0, out_stats=None, bn_bwd=None, up2=0, flags=0,
                 wino_v=None):
        super().__init__(kh, kw, stride, pad, pad_mode, pad_h, precision, int(up2), out_stats,
                         C.pointer(bn_bwd) if bn_bwd is not None else None, int(flags), 0, wino_v)


GEOM_FRAME_ZEROED = 1
GEOM_WINO = 2
ZERO_FRAMES_MAX = 32
PACK_MAX_JOBS = 48
WGRAD_MAX_JOBS = 24


class WgradJob(C.Structure):
    """itg_wgrad_job: one layer's deferred weight-gradient reduce."""
    _fields_ = [("slab", C.c_void_p), ("dbslab", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("w_orig", C.c_void_p),
                ("dot", C.c_void_p), ("splits", C.c_int32), ("dbsplits", C.c_int32), ("co", C.c_int32), ("ci", C.c_int32),
                ("ci_ld", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("co_rows", C.c_int32), ("Kpad", C.c_int32),
                ("accumulate", C.c_int32), ("stage", C.c_void_p), ("ngroups", C.c_int32), ("group", C.c_int32)]


class SnJob(C.Structure):
    """itg_sn_job: the apply half of a spectral-norm backward."""
    _fields_ = [("g_w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("inv_sigma", C.c_void_p), ("dot", C.c_void_p),
                ("d_w_orig", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("accumulate", C.c_int32),
                ("reserved", C.c_int32)]

_P = C.c_void_p
_TP = C.POINTER(Tensor)
_GP = C.POINTER(ConvGeom)
_i, _f, _d, _l = C.c_int, C.c_float, C.c_double, C.c_int64

# name -> (restype, argtypes); mirrors include/itg.h one to one
SIGNATURES = {
    "itg_version": (_i, []),
    "itg_last_conv_kernel": (C.c_char_p, []),
    "itg_pack_fwd_size": (_l, [_i, _i, _i, _i]),
    "itg_pack_dgrad_size": (_l, [_i, _i, _i, _i, _i]),
    "itg_pack_fwd": (_i, [_P, _P, _P, _i, _i, _i, _i, _i, _P]),
    "itg_pack_dgrad": (_i, [_P, _P, _P, _i, _i, _i, _i, _i, _i, _P]),
    "itg_pack_up2_fwd_size": (_l, [_i, _i]),
    "itg_pack_up2_dgrad_size": (_l, [_i, _i]),
    "itg_pack_up2_fwd": (_i, [_P, _P, _P, _i, _i, _i, _P]),
    "itg_pack_up2_dgrad": (_i, [_P, _P, _P, _i, _i, _i, _P]),
    "itg_zero_frames": (_i, [_TP, _i, _P]),
    "itg_pack_wino_size": (_l, [_i, _i]),
    "itg_pack_wino_s2_size": (_l, [_i, _i]),
    "itg_pack_wino_s2_fwd": (_i, [_P, _P, _P, _i, _i, _i, _P]),
    "itg_pack_wino_s2_dgrad_size": (_l, [_i, _i]),
    "itg_pack_wino_s2_dgrad": (_i, [_P, _P, _P, _i, _i, _i, _i, _P]),
    "itg_pack_wino3_size": (_l, [_i, _i]),
    "itg_pack_wino_fwd": (_i, [_P, _P, _P, _i, _i, _i, _P]),
    "itg_pack_wino3_fwd": (_i, [_P, _P, _P, _i, _i, _i, _P]),
    "itg_pack_wino3_dgrad": (_i, [_P, _P, _P, _i, _i, _i, _P]),
    "itg_pack_wino_dgrad": (_i, [_P, _P, _P, _i, _i, _i, _P]),
    "itg_conv2d_fwd_workspace": (_l, [_TP, _TP, _GP]),
    "itg_conv2d_dgrad_workspace": (_l, [_TP, _TP, _GP]),
    "itg_pack_multi": (_i, [_P, _i, _l, _P]),
    "itg_conv2d_fwd": (_i, [_TP, _P, _P, _P, _TP, _TP, _GP, _i, _f, _P, _l, _P]),
    "itg_conv2d_dgrad": (_i, [_TP, _P, _P, _TP, _TP, _i, _f, _GP, _P, _l, _P]),
    "itg_conv2d_wgrad_workspace": (_l, [_TP, _TP, _GP]),
    "itg_conv2d_wgrad": (_i, [_TP, _TP, _P, _P, _GP, _i, _P, _l, _P]),
    "itg_conv2d_wgrad_slabs": (_i, [_TP, _TP, _GP, _P, _l, C.POINTER(WgradJob), _P]),
    "itg_wgrad_reduce_multi": (_i, [C.POINTER(WgradJob), _i, _P]),
    "itg_spectral_norm_bwd_multi": (_i, [C.POINTER(SnJob), _i, _P]),
    "itg_local_pad_fwd": (_i, [_P, _P, _i, _i, _i, _i, _i, _i, _i, _P]),
    "itg_local_pad_bwd": (_i, [_P, _P, _i, _i, _i, _i, _i, _i, _i, _P]),
    "itg_local_pad_nhwc_fwd": (_i, [_TP, _TP, _i, _P]),
    "itg_local_pad_stream_fwd": (_i, [_TP, _P, _P, _P, _TP, _i, _P]),
    "itg_band_halo_fill": (_i, [_TP, _P, _P, _i, _i, _P]),
    "itg_band_halo_grad": (_i, [_TP, _P, _P, _i, _i, _P]),
    "itg_band_rows_get": (_i, [_TP, _i, _i, _P, _P, _P]),
    "itg_band_interior_copy": (_i, [_TP, _TP, _i, _P]),
    "itg_nchw_to_grid": (_i, [_P, _TP, _i, _P]),
    "itg_grid_to_nchw": (_i, [_TP, _P, _i, _P]),
    "itg_bn_stats": (_i, [_TP, _P, _P]),
    "itg_bn_finalize": (_i, [_P, _d, _d, _P, _P, _f, _f, _P, _P, _P, _P, _P, _i, _i, _i, _P]),
    "itg_bn_finalize_apply": (_i, [_TP, _P, _d, _d, _P, _P, _f, _f, _P, _P, _P, _P, _P, _TP, _i, _f, _P]),
    "itg_bn_apply": (_i, [_TP, _P, _TP, _i, _f, _P]),
    "itg_bn_bwd_reduce": (_i, [_TP, _TP, _P, _P, _i, _f, _P, _P]),
    "itg_bn_bwd_apply": (_i, [_TP, _TP, _P, _P, _P, _P, _d, _i, _f, _TP, _P, _P, _i, _P]),
    "itg_bn_bwd_apply_add": (_i, [_TP, _TP, _P, _P, _P, _P, _d, _i, _f, _TP, _P, _P, _i, _TP, _P]),
    "itg_ssm_modulate_fwd": (_i, [_TP, _P, _TP, _TP, _i, _f, _P]),
    "itg_ssm_modulate_bwd": (_i, [_TP, _P, _TP, _TP, _i, _f, _TP, _TP, _P]),
    "itg_act_fwd": (_i, [_TP, _TP, _i, _f, _P]),
    "itg_act_bwd": (_i, [_TP, _TP, _TP, _i, _f, _P]),
    "itg_upsample2x_fwd": (_i, [_TP, _TP, _P]),
    "itg_upsample2x_bwd": (_i, [_TP, _TP, _P]),
    "itg_add": (_i, [_TP, _TP, _TP, _P]),
    "itg_colsum": (_i, [_TP, _P, _P, _P]),
    "itg_axpby": (_i, [_P, _P, _P, _f, _P, _f, _l, _P]),
    "itg_dot": (_i, [_P, _P, _l, _P, _P]),
    "itg_stream_spin": (_i, [_i, _P]),
    "itg_attention_scratch_floats": (_l, [_TP, _TP, _TP]),
    "itg_attention_fwd": (_i, [_TP, _TP, _TP, _TP, _P, _P]),
    "itg_attention_bwd": (_i, [_TP, _TP, _TP, _P, _TP, _TP, _TP, _TP, _P]),
    "itg_maxpool2_fwd": (_i, [_TP, _TP, _P]),
    "itg_maxpool2_bwd": (_i, [_TP, _TP, _TP, _TP, _P]),
    "itg_bce_logits_fwd": (_i, [_P, _l, _f, _P, _P]),
    "itg_bce_logits_bwd": (_i, [_P, _l, _f, _P, _P, _P]),
    "itg_logit_loss_grid_workspace": (_l, []),
    "itg_logit_loss_grid": (_i, [_TP, _i, _f, _P, _TP, _P, _P]),
    "itg_hinge_fwd": (_i, [_P, _l, _i, _P, _P]),
    "itg_hinge_bwd": (_i, [_P, _l, _i, _P, _P, _P]),
    "itg_spectral_norm_power_iter": (_i, [_P, _P, _P, _i, _i, _i, _f, _P, _P, _P, _P]),
    "itg_spectral_norm_power_iter_multi": (_i, [_i, _P, _P, _P, _P, _P, _i, _f, _P, _P, _P]),
    "itg_spectral_norm_bwd": (_i, [_P, _P, _P, _P, _P, _i, _i, _P, _i, _P, _P]),
    "itg_adam_ema_step": (_i, [_P, _P, _P, _P, _P, _l, _f, _f, _f, _f, _i, _P, _f, _P]),
}

_ERR = {-1: "ITG_ERR_ARG (inconsistent shapes / unsupported geometry)",
        -2: "ITG_ERR_ALIGN (pointer or ld misaligned)",
        -3: "ITG_ERR_LAUNCH (HIP launch failed)",
        -4: "ITG_ERR_WORKSPACE (workspace too small)"}

_lib = None


class ItgError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ItgError(
            "libitg_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI and the header drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


CAPTURE_LOG = None      # a set while an engine records a hipGraph: raw handles of the streams that received a launch (ops.capture_rule)


BYTE_LOG = None         # {"bytes": 0, "calls": 0} while bench.py counts the algorithmic HBM bytes of a step (roofline.step_hbm)

# entry points whose operands are raw pointers: (index of the element count, bytes moved per element)
_PTR_BYTES = {"itg_adam_ema_step": (5, 7 * 4), "itg_axpby": (6, 3 * 4), "itg_dot": (2, 2 * 4), "itg_pack_multi": (2, 4),
              "itg_bce_logits_fwd": (1, 4), "itg_bce_logits_bwd": (1, 2 * 4), "itg_hinge_fwd": (1, 4), "itg_hinge_bwd": (1, 2 * 4)}


def _algorithmic_bytes(name, args):
    """Bytes a launch must move at least: every itg_tensor it is handed once (operands read once, results written once - a
    Winograd / split-K / slab workspace is NOT counted: that traffic is the implementation's), flat buffers by their count."""
    n = 0
    for a in args:
        t = getattr(a, "_obj", None)
        if isinstance(t, Tensor) and t.ptr:
            n += 4 * t.n * t.gh * t.gw * t.ph * t.pw * t.ld
    pb = _PTR_BYTES.get(name)
    if pb is not None:
        n += int(args[pb[0]]) * pb[1]
    if name in ("itg_conv2d_fwd", "itg_conv2d_dgrad", "itg_conv2d_wgrad", "itg_conv2d_wgrad_slabs"):
        # + the filter (panel read / gradient written once): kh * kw * ci * co floats; the first and last tensors are the two sides
        ts = [a._obj for a in args if isinstance(getattr(a, "_obj", None), Tensor) and a._obj.ptr]
        g = next((a._obj for a in args if isinstance(getattr(a, "_obj", None), ConvGeom)), None)
        if g is not None and len(ts) >= 2:
            n += 4 * g.kh * g.kw * ts[0].ld * ts[1 if name != "itg_conv2d_fwd" else -1].ld
    return n


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    if BYTE_LOG is not None:
        BYTE_LOG["calls"] += 1
        BYTE_LOG["bytes"] += _algorithmic_bytes(name, args)
    if CAPTURE_LOG is not None and args:
        st = args[-1]          # every launching entry point takes its stream last
        CAPTURE_LOG.add(getattr(st, "value", st) or 0)
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise ItgError("%s failed: %s" % (name, _ERR.get(rc, rc)))


def fn(name):
    return getattr(load(), name)
