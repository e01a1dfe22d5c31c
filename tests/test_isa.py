"""Build-time ISA checks (CPU: hipcc cross-compiles gfx950 without a GPU).

gfx950 store hazard of the strip kernels (gpurun_out/r5_full2.log): a `buffer_store_dwordx4` whose data
registers a VALU instruction overwrites in the next cycle needs a wait state.  LLVM's hazard recognizer inserts it only for
buffer stores WITHOUT an SGPR offset (the GFX9 rule); on gfx950 the SGPR-offset form is exposed as well and stored zeros.
The source keeps the row offset in the vector offset; this test pins what the compiler made of it, so that a compiler which
re-scalarises the wave-uniform part into soffset, or a second buffer store added elsewhere, fails here and not on the GPU."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "infinite_texture_gans_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# buffer_store_dwordx4 vdata, voffset, srsrc, soffset [modifiers]
STORE = re.compile(r"^\s*buffer_store_dwordx4\s+(v\[\d+:\d+\]),\s*(\S+),\s*(s\[\d+:\d+\]),\s*(\S+)(.*)$")


def _isa(src, tmp_path):
    out = os.path.join(str(tmp_path), os.path.basename(src)[:-4] + ".s")
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out]
    subprocess.run(cmd, cwd=CSRC, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(out) as f:
        return f.read().splitlines()


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not installed")
def test_strip_kernel_buffer_stores_carry_no_sgpr_offset(tmp_path):
    lines = _isa(os.path.join(CSRC, "conv_strip.hip"), tmp_path)
    stores = [(i, STORE.match(l)) for i, l in enumerate(lines) if "buffer_store_dwordx4" in l]
    assert len(stores) >= 24, "the strip kernels' epilogue stores were not found in the ISA (%d)" % len(stores)
    for i, m in stores:
        assert m is not None, "unparsed store: %r" % lines[i]
        soffset = m.group(4).rstrip(",")
        assert soffset == "0", "line %d: buffer store with soffset %s (gfx950 store hazard): %s" % (i + 1, soffset, lines[i].strip())
        assert "offen" in m.group(5), lines[i]


def test_buffer_stores_exist_in_the_strip_kernels_only():
    """Every other kernel stores through flat / global instructions, whose > 64-bit data hazard LLVM always covers
    (GCNHazardRecognizer::createsVALUHazard has no soffset exemption for FLAT); a new raw buffer store must come with
    its own ISA check above."""
    users = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        with open(src) as f:
            if "raw_buffer_store" in f.read():
                users.append(os.path.basename(src))
    assert users == ["conv_strip.hip"], users


def test_strip_drop_sentinel_survives_any_row_offset():
    """lanes whose channel group lies past out.ld must stay out of range for EVERY row of a tensor up to the
    admitted 0xFFFF0000 bytes; the old sentinel (= tensor bytes) wrapped back into range once sentinel + row offset
    passed 2^32.  The arithmetic of the kernel, restated: the offset is selected, never added to."""
    with open(os.path.join(CSRC, "conv_strip.hip")) as f:
        src = f.read()
    m = re.search(r"constexpr unsigned STRIP_DROP = (0x[0-9A-Fa-f]+)u;", src)
    assert m, "STRIP_DROP not found"
    drop = int(m.group(1), 16)
    limit = 0xFFFF0000                                   # try_conv_strip admits tensors below this many bytes
    assert re.search(r"ob >= 0xFFFF0000LL", src), "size limit of try_conv_strip changed: revisit the sentinel"
    assert limit <= drop and drop + 16 <= 2 ** 32        # out of range without a 32-bit wrap in (offset + 16)
    assert re.search(r"vo\[i\]\[f\] == STRIP_DROP \? STRIP_DROP : vo\[i\]\[f\] \+ ob", src), "the store offset must select the sentinel"
    # the old form, for the record: wraps into range for a 3.2 GB tensor's upper rows
    out_bytes, ob = 3_200_000_000, 2_000_000_000
    assert (out_bytes + ob) % 2 ** 32 < out_bytes
