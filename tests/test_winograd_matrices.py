"""The Winograd matrices the HIP kernels are compiled with (csrc/winograd_f44.h, winograd_f43.h) are the generator's
(tools/gen_winograd.py), and they satisfy the 2-D identity  y = A^T [(G g G^T) .* (B^T d B)] A  = the direct correlation -
in fp64 to 1e-10, i.e. the algorithm itself is exact and what the GPU tests measure is fp32 rounding only."""
import importlib.util
import os
import re
from fractions import Fraction as Fr

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_winograd", os.path.join(ROOT, "tools", "gen_winograd.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _header(fname, prefix):
    text = open(os.path.join(ROOT, "infinite_texture_gans_amd", "csrc", fname)).read()
    out = {}
    for name in ("AT", "G", "BT"):
        m = re.search(r"%s_%s\[(\d+)\]\[(\d+)\] = \{(.*?)\};" % (prefix, name), text, re.S)
        rows, cols = int(m.group(1)), int(m.group(2))
        vals = [float(v.rstrip("f")) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?f", m.group(3))]
        out[name] = np.array(vals).reshape(rows, cols)
    return out


CASES = [(4, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1, 2)], "winograd_f44.h", "WINO"),
         (3, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2)], "winograd_f43.h", "WINO3"),
         (2, [Fr(0), Fr(1), Fr(-1), Fr(2)], "winograd_f42.h", "WINO2")]


@pytest.mark.parametrize("R,pts,fname,prefix", CASES, ids=["F(4x4,4x4)", "F(4x4,3x3)", "F(4x4,2x2)"])
def test_compiled_matrices_are_the_generators_and_exact(R, pts, fname, prefix):
    at, g, bt = _gen().matrices(R, pts)
    h = _header(fname, prefix)
    for name, ref in (("AT", at), ("G", g), ("BT", bt)):
        assert h[name].shape == ref.shape
        assert np.array_equal(h[name], ref.astype(np.float32).astype(np.float64)), name     # the header holds the fp32 roundings
    # 2-D identity in fp64 with the exact (unrounded) matrices
    n = 4 + R - 1
    rng = np.random.default_rng(5)
    for _ in range(4):
        d, w = rng.standard_normal((n, n)), rng.standard_normal((R, R))
        y = at @ ((g @ w @ g.T) * (bt @ d @ bt.T)) @ at.T
        ref = np.array([[np.sum(d[k:k + R, l:l + R] * w) for l in range(4)] for k in range(4)])
        assert np.allclose(y, ref, atol=1e-10)
    # the bias gradient reads the transformed dy at the point 1: its A row must be all ones (conv_wino.hip)
    assert np.array_equal(at[:, 1], np.ones(4))
