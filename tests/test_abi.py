"""CPU-side checks of the C-ABI boundary: the in-tree library loads, exports every symbol that
include/itg.h declares, the ctypes table mirrors the header, and the host-only helpers (no kernel
launch) behave.  No compute call is made here - there is no GPU in this tier."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "itg.h")


def declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(itg_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from infinite_texture_gans_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(lib):
    names = declared()
    assert len(names) >= 40
    so = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_ctypes_table_mirrors_header(lib):
    assert sorted(lib.SIGNATURES) == declared()


def test_argument_counts_match_header(lib):
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(args), (name, n, len(args))


def test_host_only_helpers(lib):
    so = lib.load()
    assert so.itg_version() >= 100
    # packed sizes: rows rounded to 16, K rounded to 16
    assert so.itg_pack_fwd_size(13, 28, 3, 3) == 16 * 256
    assert so.itg_pack_dgrad_size(3, 64, 4, 4, 2) == 4 * 16 * 256
    assert so.itg_pack_dgrad_size(26, 16, 3, 3, 1) == 32 * 144


def test_rejected_calls_return_error_codes_not_crashes(lib):
    so = lib.load()
    # null / inconsistent arguments must come back as ITG_ERR_* (negative), never touch the device
    assert so.itg_bce_logits_fwd(None, 4, ctypes.c_float(1.0), None, None) < 0
    assert so.itg_local_pad_fwd(None, None, 1, 1, 3, 3, 4, 1, 0, None) < 0
    t = lib.Tensor(None, 1, 1, 1, 4, 4, 3, 4)
    assert so.itg_bn_stats(ctypes.byref(t), None, None) < 0
    # the conv family with null tensors / panels / geometry, and a tensor whose pointer is null (the no-gradient-requested
    # shapes a caller can produce: bias None, no weight gradient, no act_out) - VERDICT r3 item 2a
    f0 = ctypes.c_float(0.0)
    assert so.itg_conv2d_fwd(None, None, None, None, None, None, None, 0, f0, None, 0, None) < 0
    assert so.itg_conv2d_dgrad(None, None, None, None, None, 0, f0, None, None, 0, None) < 0
    assert so.itg_conv2d_dgrad(ctypes.byref(t), None, None, ctypes.byref(t), None, 0, f0, None, None, 0, None) < 0
    assert so.itg_conv2d_wgrad(None, None, None, None, None, 0, None, 0, None) < 0
    assert so.itg_conv2d_wgrad(ctypes.byref(t), ctypes.byref(t), None, None, None, 0, None, 0, None) < 0
    ok = lib.Tensor(ctypes.c_void_p(64), 1, 1, 1, 4, 4, 3, 4)           # a well-formed descriptor (never dereferenced: the
    assert so.itg_conv2d_dgrad(ctypes.byref(ok), None, None, ctypes.byref(ok), None, 0, f0, None, None, 0, None) < 0   # panel is null)
    assert so.itg_conv2d_wgrad(ctypes.byref(ok), ctypes.byref(ok), None, None, None, 0, None, 0, None) < 0               # geometry is null
    # round 5's entry points: the band halo rows (a band is one 1 x 1-grid image with >= 3 rows; buffer mode needs the buffer),
    # the fused loss head (one-channel logits, a zeroed workspace), the stride-2 Winograd panels
    assert so.itg_band_halo_fill(None, None, None, 1, 1, None) < 0
    assert so.itg_band_halo_fill(ctypes.byref(ok), None, None, 0, 1, None) < 0          # rows from a buffer, no buffer
    assert so.itg_band_halo_fill(ctypes.byref(ok), None, None, 3, 1, None) < 0          # unknown mode
    assert so.itg_band_halo_grad(ctypes.byref(ok), None, None, 1, 0, None) < 0
    assert so.itg_band_rows_get(ctypes.byref(ok), 0, 9, None, None, None) < 0
    assert so.itg_band_interior_copy(ctypes.byref(ok), ctypes.byref(ok), 1, None) < 0   # the band must be two rows shorter
    assert so.itg_logit_loss_grid(None, 0, f0, None, None, None, None) < 0
    assert so.itg_logit_loss_grid(ctypes.byref(ok), 0, f0, ctypes.c_void_p(64), ctypes.byref(ok), ctypes.c_void_p(64), None) < 0   # 3 channels
    one = lib.Tensor(ctypes.c_void_p(64), 1, 1, 1, 4, 4, 1, 4)
    assert so.itg_logit_loss_grid(ctypes.byref(one), 7, f0, ctypes.c_void_p(64), ctypes.byref(one), ctypes.c_void_p(64), None) < 0  # unknown kind
    assert so.itg_logit_loss_grid_workspace() >= 9
    assert so.itg_pack_wino_s2_fwd(None, None, None, 128, 64, 64, None) < 0
    assert so.itg_pack_wino_s2_dgrad(None, None, None, 128, 64, 64, 128, None) < 0
    assert so.itg_pack_wino_s2_size(256, 128) == 25 * 256 * 512 and so.itg_pack_wino_s2_dgrad_size(128, 256) == 25 * 512 * 256
    bad_ld = lib.Tensor(ctypes.c_void_p(64), 1, 1, 1, 4, 4, 3, 3)      # ld not a multiple of 4
    assert so.itg_act_fwd(ctypes.byref(bad_ld), ctypes.byref(bad_ld), 1, ctypes.c_float(0.2), None) == -2


def test_product_has_no_cpu_fallback_and_does_not_import_oracle():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import infinite_texture_gans_amd.utils, infinite_texture_gans_amd.engine; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imported the oracle'; "
            "import torch; from infinite_texture_gans_amd import ops, _lib\n"
            "try:\n    ops.to_grid(torch.zeros(1,3,4,4), 1, 1, True)\n    raise SystemExit('CPU tensor was accepted')\n"
            "except _lib.ItgError: pass") % ROOT
    subprocess.check_call([sys.executable, "-c", code])


def test_struct_layouts_match_the_header(lib, tmp_path):
    """The ctypes mirrors of itg_tensor / itg_conv_geom / itg_bn_bwd_fuse (round 6) have the C structs' sizes and field offsets:
    a C program that includes include/itg.h prints them (gcc is enough - the header is plain C)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "itg.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu\\n", sizeof(itg_tensor), sizeof(itg_conv_geom), sizeof(itg_bn_bwd_fuse));\n'
                   '  printf("%zu %zu %zu %zu\\n", offsetof(itg_conv_geom, out_stats), offsetof(itg_conv_geom, bn_bwd), '
                   'offsetof(itg_conv_geom, flags), offsetof(itg_conv_geom, wino_v));\n'
                   '  printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(itg_bn_bwd_fuse, x), offsetof(itg_bn_bwd_fuse, ab), '
                   'offsetof(itg_bn_bwd_fuse, mean_rstd), offsetof(itg_bn_bwd_fuse, act), offsetof(itg_bn_bwd_fuse, sums), '
                   'offsetof(itg_bn_bwd_fuse, taken));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    got = [int(v) for v in out]
    G, F = lib.ConvGeom, lib.BnBwdFuse
    want = [ctypes.sizeof(lib.Tensor), ctypes.sizeof(G), ctypes.sizeof(F),
            G.out_stats.offset, G.bn_bwd.offset, G.flags.offset, G.wino_v.offset,
            F.x.offset, F.ab.offset, F.mean_rstd.offset, F.act.offset, F.sums.offset, F.taken.offset]
    assert got == want, (got, want)


def test_bn_bwd_fuse_is_an_input_gradient_argument_only(lib):
    """itg_conv_geom.bn_bwd (round 6): the forward and the weight gradient reject a geometry that carries it; the input gradient
    validates it (a null BatchNorm input is an argument error, not a crash) and clears `taken` before anything else."""
    so = lib.load()
    f0 = ctypes.c_float(0.0)
    ok = lib.Tensor(ctypes.c_void_p(64), 1, 1, 1, 4, 4, 3, 4)
    fuse = lib.BnBwdFuse(None, None, None, 1, 0.02, None, 7, 0)
    g = lib.ConvGeom(3, 3, 1, 1, 0, bn_bwd=fuse)
    panel = ctypes.c_void_p(64)
    assert so.itg_conv2d_fwd(ctypes.byref(ok), panel, None, None, None, ctypes.byref(ok), ctypes.byref(g), 0, f0, None, 0, None) < 0
    assert so.itg_conv2d_wgrad(ctypes.byref(ok), ctypes.byref(ok), panel, None, ctypes.byref(g), 0, None, 0, None) < 0
    assert so.itg_conv2d_dgrad(ctypes.byref(ok), panel, None, ctypes.byref(ok), None, 0, f0, ctypes.byref(g), None, 0, None) < 0
    assert g.bn_bwd.contents.taken == 0
