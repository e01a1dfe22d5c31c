#!/usr/bin/env python3
"""CPU study of the rounding error of Winograd F(m x m, 4 x 4) for D's 256 -> 512 layer (round 4).
Emulates the GPU pipeline in numpy: input transform V = B^T d B, filter transform U = G g G^T, NP^2 GEMMs over the input
channels (fp32 operands, fp32 accumulation), output transform A^T M A.  Each of the three transforms can run in fp32 or in
fp64 (operands / results rounded to fp32 once).  Reports rel-L2 against the fp64 direct correlation for candidate point sets.
Run on the CPU: python tools/wino_error_study.py"""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np


def matrices(M, R, PTS):
    """Cook-Toom matrices with exact rational arithmetic (A^T: M x N, G: N x R, B^T: N x N), last point = infinity."""
    N = M + R - 1
    assert len(PTS) == N - 1
    at = [[Fr(0)] * N for _ in range(M)]
    g = [[Fr(0)] * R for _ in range(N)]
    for i, a in enumerate(PTS):
        ni = Fr(1)
        for j, b in enumerate(PTS):
            if j != i:
                ni *= a - b
        for k in range(M):
            at[k][i] = a ** k
        for j in range(R):
            g[i][j] = a ** j / ni
    at[M - 1][N - 1] = Fr(1)
    g[N - 1][R - 1] = Fr(1)
    # B^T: rows are the coefficients of prod_{j != i}(x - p_j) (Lagrange numerators), last row: prod over all points
    def polymul(p, q):
        r = [Fr(0)] * (len(p) + len(q) - 1)
        for i, a in enumerate(p):
            for j, b in enumerate(q):
                r[i + j] += a * b
        return r
    bt = [[Fr(0)] * N for _ in range(N)]
    for i in range(N - 1):
        p = [Fr(1)]
        for j, b in enumerate(PTS):
            if j != i:
                p = polymul(p, [-b, Fr(1)])
        for k, c in enumerate(p):
            bt[i][k] = c
    p = [Fr(1)]
    for b in PTS:
        p = polymul(p, [-b, Fr(1)])
    for k, c in enumerate(p):
        bt[N - 1][k] = c
    f = lambda m: np.array([[float(v) for v in r] for r in m])
    return f(at), f(g), f(bt)


def check(M, R, at, g, bt, rng):
    N = M + R - 1
    for _ in range(4):
        gg, d = rng.standard_normal(R), rng.standard_normal(N)
        y = at @ ((g @ gg) * (bt @ d))
        ref = np.array([sum(gg[j] * d[k + j] for j in range(R)) for k in range(M)])
        assert np.allclose(y, ref, atol=1e-9), (y, ref)


def run(M, R, PTS, x, w, tf_in, tf_w, tf_out, rng):
    """x [tiles, N, N, Cin] fp64 input tiles, w [Cout, Cin, R, R]; returns rel-L2 of the emulated pipeline vs fp64 direct."""
    at, g, bt = matrices(M, R, PTS)
    check(M, R, at, g, bt, rng)
    N = M + R - 1
    f32 = np.float32
    # reference, fp64
    ref = np.zeros((x.shape[0], M, M, w.shape[0]))
    for k in range(M):
        for l in range(M):
            ref[:, k, l, :] = np.einsum("tijc,ocij->to", x[:, k:k + R, l:l + R, :], w)
    x32, w32 = x.astype(f32), w.astype(f32)

    def tf2(mat, arr, axes, dt):
        """arr <- mat applied along two axes in dtype dt, sequentially (rows then columns), rounding after each pass as the kernel does"""
        m = mat.astype(dt)
        a = arr.astype(dt)
        a = np.moveaxis(np.tensordot(m, a, axes=(1, axes[0])), 0, axes[0]).astype(dt)
        a = np.moveaxis(np.tensordot(m, a, axes=(1, axes[1])), 0, axes[1]).astype(dt)
        return a

    V = tf2(bt, x32, (1, 2), tf_in).astype(f32)               # [tiles, N, N, Cin]
    U = tf2(g, w32, (2, 3), tf_w).astype(f32)                 # [Cout, Cin, N, N]
    Mm = np.empty((x.shape[0], N, N, w.shape[0]), f32)
    for a in range(N):
        for b in range(N):
            # fp32 GEMM, sequential-ish accumulation in chunks of 4 (the MFMA K step), fp32 accumulator
            acc = np.zeros((x.shape[0], w.shape[0]), f32)
            Va, Ub = V[:, a, b, :], U[:, :, a, b]
            for c0 in range(0, Va.shape[1], 4):
                acc = (acc + (Va[:, c0:c0 + 4].astype(np.float64) @ Ub[:, c0:c0 + 4].astype(np.float64).T).astype(f32)).astype(f32)
            Mm[:, a, b, :] = acc
    y = tf2(at, Mm, (1, 2), tf_out)
    return float(np.linalg.norm(y.astype(np.float64) - ref) / np.linalg.norm(ref))


def direct_f32(x, w, M, R):
    f32 = np.float32
    ref = np.zeros((x.shape[0], M, M, w.shape[0]))
    got = np.zeros((x.shape[0], M, M, w.shape[0]), f32)
    x32, w32 = x.astype(f32), w.astype(f32)
    for k in range(M):
        for l in range(M):
            ref[:, k, l, :] = np.einsum("tijc,ocij->to", x[:, k:k + R, l:l + R, :], w)
            acc = np.zeros((x.shape[0], w.shape[0]), f32)
            for i in range(R):
                for j in range(R):
                    for c0 in range(0, x.shape[3], 4):
                        acc = (acc + (x32[:, k + i, l + j, c0:c0 + 4].astype(np.float64) @ w32[:, c0:c0 + 4, i, j].astype(np.float64).T).astype(f32)).astype(f32)
            got[:, k, l, :] = acc
    return float(np.linalg.norm(got.astype(np.float64) - ref) / np.linalg.norm(ref))


def main():
    rng = np.random.default_rng(0)
    R, Cin, Cout, tiles = 4, 256, 32, 24
    F = Fr
    sets = {
        "F44 0,1,-1,2,-2,1/2 (round 3)": (4, [F(0), F(1), F(-1), F(2), F(-2), F(1, 2)]),
        "F44 0,1,-1,1/2,-1/2,2": (4, [F(0), F(1), F(-1), F(1, 2), F(-1, 2), F(2)]),
        "F44 0,1,-1,1/2,-2,2": (4, [F(0), F(1), F(-1), F(1, 2), F(-2), F(2)]),
        "F44 0,1,-1,1/2,-1/2,-2": (4, [F(0), F(1), F(-1), F(1, 2), F(-1, 2), F(-2)]),
        "F44 0,1,-1,1/2,-2,-1/2 ": (4, [F(0), F(1), F(-1), F(1, 2), F(-2), F(-1, 2)]),
        "F44 0,1,-1,3/4,-4/3,1/2": (4, [F(0), F(1), F(-1), F(3, 4), F(-4, 3), F(1, 2)]),
        "F44 0,1,-1,2/3,-3/2,1/2": (4, [F(0), F(1), F(-1), F(2, 3), F(-3, 2), F(1, 2)]),
        "F44 0,1,-1,2/3,-3/2,-1/2": (4, [F(0), F(1), F(-1), F(2, 3), F(-3, 2), F(-1, 2)]),
        "F34 0,1,-1,1/2,-1/2": (3, [F(0), F(1), F(-1), F(1, 2), F(-1, 2)]),
        "F34 0,1,-1,2,-1/2": (3, [F(0), F(1), F(-1), F(2), F(-1, 2)]),
        "F34 0,1,-1,2,-2": (3, [F(0), F(1), F(-1), F(2), F(-2)]),
        "F24 0,1,-1,1/2": (2, [F(0), F(1), F(-1), F(1, 2)]),
        "F24 0,1,-1,2": (2, [F(0), F(1), F(-1), F(2)]),
    }
    only = sys.argv[1:]
    for name, (M, pts) in sets.items():
        if only and not any(o in name for o in only):
            continue
        N = M + R - 1
        x = rng.standard_normal((tiles, N, N, Cin))
        # activations behind a LeakyReLU are not zero-mean: add the positive bias the real layer sees
        x = np.where(x > 0, x, 0.2 * x)
        w = rng.standard_normal((Cout, Cin, R, R)) * 0.02
        out = []
        for (ti, tw, to) in [(np.float32,) * 3, (np.float64, np.float64, np.float32), (np.float32, np.float32, np.float64),
                             (np.float64,) * 3]:
            out.append(run(M, R, pts, x, w, ti, tw, to, rng))
        d = direct_f32(x, w, M, R)
        print("%-34s all-f32 %.2e | in,w f64 %.2e | out f64 %.2e | all f64 %.2e | direct f32 %.2e   (mults/output %.2f)"
              % (name, out[0], out[1], out[2], out[3], d, N * N / (M * M)), flush=True)


if __name__ == "__main__":
    main()
