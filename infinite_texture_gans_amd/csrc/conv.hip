// Weight packing, the taps-as-rows paths of the single-output-channel / 3-input-channel layers, and the C ABI of the
// convolution family for gfx950 (MI355X).  Kernels: conv_nt.hip, conv_tile.hip, conv_wgrad.hip.
#include "conv_common.h"
#include "winograd_f44.h"
#include "winograd_f43.h"
#include "winograd_f42.h"

using namespace itgk;

namespace {

inline int pad_v_raw(const itg_conv_geom* g) { return g->pad_h >= 0 ? g->pad_h : g->pad; }

// ------------------------------------------------------------------------------- packing
// fwd: out[co][k], k = (ky*kw+kx)*ci_ld + ci, rows co >= co zero, k >= K zero
__global__ void pack_fwd_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                int co, int ci, int ci_ld, int kh, int kw, int co_pad, int Kpad) {
  int64_t total = (int64_t)co_pad * Kpad;
  float sc = scale ? *scale : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(i % Kpad);
    int o = (int)(i / Kpad);
    int tap = k / ci_ld, c = k - tap * ci_ld;
    float v = 0.f;
    if (o < co && tap < kh * kw && c < ci) {
      int y = tap / kw, x = tap - y * kw;
      v = w[(((size_t)o * ci + c) * kh + y) * kw + x] * sc;
    }
    out[i] = v;
  }
}

// dgrad stride 1: out[ci][k], k = (ky'*kw+kx')*co_ld + co, W[co][ci][kh-1-ky'][kw-1-kx']
// dgrad stride 2: class (ry,rx) major; taps (jy',jx') of the (kh/2 x kw/2) sub-kernel,
//                 ky = ay + 2*(kh/2-1-jy'), ay = (ry+pad)&1 (same for x)
__global__ void pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                  int co, int ci, int co_ld, int kh, int kw, int stride, int pad, int ci_pad,
                                  int Kpad) {
  int ncls = stride * stride;
  int skh = kh / stride, skw = kw / stride;
  int64_t total = (int64_t)ncls * ci_pad * Kpad;
  float sc = scale ? *scale : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(i % Kpad);
    int64_t r = i / Kpad;
    int c_in = (int)(r % ci_pad);
    int cls = (int)(r / ci_pad);
    int ry = cls / stride, rx = cls - ry * stride;
    int tap = k / co_ld, o = k - tap * co_ld;
    float v = 0.f;
    if (c_in < ci && o < co && tap < skh * skw) {
      int jy = tap / skw, jx = tap - jy * skw;
      int ay = (ry + pad) % stride, ax = (rx + pad) % stride;
      int y = ay + stride * (skh - 1 - jy), x = ax + stride * (skw - 1 - jx);
      v = w[(((size_t)o * ci + c_in) * kh + y) * kw + x] * sc;
    }
    out[i] = v;
  }
}


// Panels of conv3x3(up2x(x)) (itg_conv_geom.up2).  Forward: class (ry, rx) major, out[cls][o][k], k = (jy*2+jx)*ci_ld + c;
// tap jj of parity r sums the 3x3 taps lo..hi per axis:  lo = jj ? 1 + r : 0,  hi = jj ? 2 : r.
__device__ __forceinline__ float up2_fwd_elem(const float* __restrict__ w, int co, int ci, int ci_ld, int co_pad, int Kpad,
                                              unsigned e) {
  const int k = (int)(e % (unsigned)Kpad);
  const unsigned r = e / (unsigned)Kpad;
  const int o = (int)(r % (unsigned)co_pad), cls = (int)(r / (unsigned)co_pad);
  const int tap = k / ci_ld, c = k - tap * ci_ld;
  if (o >= co || tap >= 4 || c >= ci) return 0.f;
  const int ry = cls >> 1, rx = cls & 1, jy = tap >> 1, jx = tap & 1;
  const int ylo = jy ? 1 + ry : 0, yhi = jy ? 2 : ry, xlo = jx ? 1 + rx : 0, xhi = jx ? 2 : rx;
  const float* q = w + ((size_t)o * ci + c) * 9;
  float v = 0.f;
  for (int y = ylo; y <= yhi; ++y)
    for (int x = xlo; x <= xhi; ++x) v += q[y * 3 + x];
  return v;
}
// Input gradient: out[c_in][k], k = (ty*4+tx)*co_ld + o - the forward-layout panel of the 4x4 stride-2 pad-1 conv of dy;
// tap t sums the 3x3 taps max(0, 2-t) .. min(2, 3-t) per axis.
__device__ __forceinline__ float up2_dgrad_elem(const float* __restrict__ w, int co, int ci, int co_ld, int Kpad, unsigned e) {
  const int k = (int)(e % (unsigned)Kpad);
  const int c_in = (int)(e / (unsigned)Kpad);
  const int tap = k / co_ld, o = k - tap * co_ld;
  if (c_in >= ci || tap >= 16 || o >= co) return 0.f;
  const int ty = tap >> 2, tx = tap & 3;
  const int ylo = max(0, 2 - ty), yhi = min(2, 3 - ty), xlo = max(0, 2 - tx), xhi = min(2, 3 - tx);
  const float* q = w + ((size_t)o * ci + c_in) * 9;
  float v = 0.f;
  for (int y = ylo; y <= yhi; ++y)
    for (int x = xlo; x <= xhi; ++x) v += q[y * 3 + x];
  return v;
}

// Winograd F(4 x 4, R x R) panels (conv_wino.hip), NP = 4 + R - 1: U[xi = a * NP + b][row][k] = sum_ij G[a][i] G[b][j] g[i][j].
// forward: row = co, k = ci, g = w[co][ci];  input gradient: row = ci, k = co, g[i][j] = w[co][ci][R - 1 - i][R - 1 - j]
template <int R> __device__ __forceinline__ constexpr float wino_g(int i, int j) { return R == 4 ? WINO_G[i][j < 4 ? j : 0] : WINO3_G[i < 6 ? i : 0][j < 3 ? j : 0]; }

template <int R>
__device__ __forceinline__ float wino_elem(const float* __restrict__ w, int co, int ci, int ld, int dgrad, unsigned e) {
  constexpr int NP = 4 + R - 1;
  const int rows = dgrad ? ci : co, kdim = dgrad ? co : ci;
  const unsigned Kpad = (unsigned)round_up_d(ld, BK), rows_pad = (unsigned)round_up_d(rows, 16);
  const int k = (int)(e % Kpad);
  const unsigned r = e / Kpad;
  const int row = (int)(r % rows_pad), xi = (int)(r / rows_pad);
  if (row >= rows || k >= kdim || xi >= NP * NP) return 0.f;
  const int a = xi / NP, b = xi - a * NP;
  const float* g = w + ((size_t)(dgrad ? k : row) * ci + (dgrad ? row : k)) * (R * R);
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < R; ++j) {             // the order wino_pack4 sums in: the two agree to the bit
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float ga = 0.f;                        // G[a][i] with a run-time a: select over the NP rows
#pragma unroll
      for (int aa = 0; aa < NP; ++aa) ga = a == aa ? wino_g<R>(aa, i) : ga;
      t = fmaf(ga, dgrad ? g[R * R - 1 - (i * R + j)] : g[i * R + j], t);
    }
    float gb = 0.f;
#pragma unroll
    for (int bb = 0; bb < NP; ++bb) gb = b == bb ? wino_g<R>(bb, j) : gb;
    v = fmaf(gb, t, v);
  }
  return v;
}

// The same panel, one thread = FOUR consecutive k of one row and all NP^2 classes of them: the taps of a filter are
// read once instead of NP^2 times, the stores are 16 bytes wide.  item < rows_pad * Kpad / 4.
template <int R>
__device__ __forceinline__ void wino_pack4(const float* __restrict__ w, float* __restrict__ out, int co, int ci, int ld,
                                           int dgrad, unsigned item) {
  constexpr int NP = 4 + R - 1, RR = R * R;
  const int rows = dgrad ? ci : co, kdim = dgrad ? co : ci;
  const unsigned Kpad = (unsigned)round_up_d(ld, BK), rows_pad = (unsigned)round_up_d(rows, 16);
  const unsigned kq = Kpad >> 2;
  const int row = (int)(item / kq), k0 = (int)(item - (unsigned)row * kq) * 4;
  float g[4][RR];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = k0 + q;
    if (row < rows && k < kdim) {
      const float* src = w + ((size_t)(dgrad ? k : row) * ci + (dgrad ? row : k)) * RR;
      if constexpr (R == 4) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const f32x4 x = reinterpret_cast<const f32x4*>(src)[h];
#pragma unroll
          for (int u = 0; u < 4; ++u) g[q][dgrad ? 15 - (h * 4 + u) : h * 4 + u] = x[u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < RR; ++u) g[q][dgrad ? RR - 1 - u : u] = src[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < RR; ++u) g[q][u] = 0.f;
    }
  }
  const size_t plane = (size_t)rows_pad * Kpad;
  float* o = out + (size_t)row * Kpad + k0;
#pragma unroll
  for (int a = 0; a < NP; ++a) {
    float t[4][R];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < R; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) s = fmaf(wino_g<R>(a, i), g[q][i * R + j], s);
        t[q][j] = s;
      }
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j) s = fmaf(wino_g<R>(b, j), t[q][j], s);
        v[q] = s;
      }
      *reinterpret_cast<f32x4*>(o + (size_t)(a * NP + b) * plane) = v;
    }
  }
}

// F(4 x 4, 2 x 2) panel of a 4 x 4 STRIDE-2 convolution (conv_wino.hip wino_conv_s2): the four parity classes (a, b) of the
// filter, g_ab(jy, jx) = w(2 jy + a, 2 jx + b), concatenated along K: U[xi = al * 5 + be][co][k = cls * ld + c] =
// sum_jy,jx G[al][jy] G[be][jx] g_ab[jy][jx]
__device__ __forceinline__ float wino_s2_elem(const float* __restrict__ w, int co, int ci, int ld, unsigned e) {
  const unsigned Kpad = (unsigned)round_up_d(4 * ld, BK), rows_pad = (unsigned)round_up_d(co, 16);
  const int k = (int)(e % Kpad);
  const unsigned r = e / Kpad;
  const int row = (int)(r % rows_pad), xi = (int)(r / rows_pad);
  const int cls = k / ld, c = k - cls * ld;
  if (row >= co || cls >= 4 || c >= ci || xi >= 25) return 0.f;
  const int a = cls >> 1, b = cls & 1, al = xi / 5, be = xi - al * 5;
  const float* g = w + ((size_t)row * ci + c) * 16;
  float v = 0.f;
#pragma unroll
  for (int jx = 0; jx < 2; ++jx) {
    float t = 0.f;
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) t = fmaf(WINO2_G[al][jy], g[(2 * jy + a) * 4 + 2 * jx + b], t);
    v = fmaf(WINO2_G[be][jx], t, v);
  }
  return v;
}

// ... and its transpose for the input gradient (wino_conv_s2_dgrad): UT[xi][row = cls * ci_ld + c][k = co] = U[xi][co][cls * ci_ld + c]
__device__ __forceinline__ float wino_s2_dgrad_elem(const float* __restrict__ w, int co, int ci, int ci_ld, int co_ld, unsigned e) {
  const unsigned Kpad = (unsigned)round_up_d(co_ld, BK), rows_pad = (unsigned)round_up_d(4 * ci_ld, 16);
  const int k = (int)(e % Kpad);
  const unsigned r = e / Kpad;
  const int row = (int)(r % rows_pad), xi = (int)(r / rows_pad);
  const int cls = row / ci_ld, c = row - cls * ci_ld;
  if (k >= co || cls >= 4 || c >= ci || xi >= 25) return 0.f;
  const int a = cls >> 1, b = cls & 1, al = xi / 5, be = xi - al * 5;
  const float* g = w + ((size_t)k * ci + c) * 16;
  float v = 0.f;
#pragma unroll
  for (int jx = 0; jx < 2; ++jx) {
    float t = 0.f;
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) t = fmaf(WINO2_G[al][jy], g[(2 * jy + a) * 4 + 2 * jx + b], t);
    v = fmaf(WINO2_G[be][jx], t, v);
  }
  return v;
}

// Both F(4 x 4, 2 x 2) panels, one thread = FOUR consecutive k of one row and all 25 classes of them (round 6; as wino_pack4 does
// for F(4 x 4, 4 x 4)): the element form above reads a filter's taps 25 times and stores 4 bytes at a time.  Bit-identical to
// wino_s2_elem / wino_s2_dgrad_elem (same fma order: t over jy, v over jx).
//   forward  (transposed = 0): row = co, k = cls * ld + c        - the four k are four input channels of one parity class
//   adjoint  (transposed = 1): row = cls * ci_ld + c, k = co     - the four k are four filters
// item < rows_pad * Kpad / 4.
__device__ __forceinline__ void wino_s2_pack4(const float* __restrict__ w, float* __restrict__ out, int co, int ci, int ci_ld,
                                              int co_ld, int transposed, unsigned item) {
  const unsigned Kpad = (unsigned)round_up_d(transposed ? co_ld : 4 * ci_ld, BK);
  const unsigned rows_pad = (unsigned)round_up_d(transposed ? 4 * ci_ld : co, 16);
  const unsigned kq = Kpad >> 2;
  const int row = (int)(item / kq), k0 = (int)(item - (unsigned)row * kq) * 4;
  const int rc = transposed ? row : k0;                    // the index that carries (cls, c)
  const int cls = rc / ci_ld, c0 = rc - cls * ci_ld;
  const int a = cls >> 1, b = cls & 1;
  float g[4][2][2];                                        // [q][jy][jx]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = transposed ? k0 + q : row, c = transposed ? c0 : c0 + q;
    const bool ok = cls < 4 && o < co && c < ci;
    const float* src = w + ((size_t)(ok ? o : 0) * ci + (ok ? c : 0)) * 16;
#pragma unroll
    for (int jy = 0; jy < 2; ++jy)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) g[q][jy][jx] = ok ? src[(2 * jy + a) * 4 + 2 * jx + b] : 0.f;
  }
  const size_t plane = (size_t)rows_pad * Kpad;
  float* o = out + (size_t)row * Kpad + k0;
#pragma unroll
  for (int al = 0; al < 5; ++al) {
    float t[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) {
        float s = 0.f;
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) s = fmaf(WINO2_G[al][jy], g[q][jy][jx], s);
        t[q][jx] = s;
      }
#pragma unroll
    for (int be = 0; be < 5; ++be) {
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = 0.f;
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) s = fmaf(WINO2_G[be][jx], t[q][jx], s);
        v[q] = s;
      }
      *reinterpret_cast<f32x4*>(o + (size_t)(al * 5 + be) * plane) = v;
    }
  }
}

__global__ void pack_wino_s2_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out, int co,
                                          int ci, int ci_ld, int co_ld, long long total) {
  const float sc = scale ? *scale : 1.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = sc * wino_s2_dgrad_elem(w, co, ci, ci_ld, co_ld, (unsigned)i);
}

__global__ void pack_wino_s2_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out, int co, int ci,
                                    int ld, long long total) {
  const float sc = scale ? *scale : 1.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = sc * wino_s2_elem(w, co, ci, ld, (unsigned)i);
}

template <int R>
__global__ void pack_wino_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out, int co,
                                 int ci, int ld, int dgrad, long long total) {
  const float sc = scale ? *scale : 1.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = sc * wino_elem<R>(w, co, ci, ld, dgrad, (unsigned)i);
}

__global__ void pack_up2_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out, int co,
                                int ci, int ld, int dgrad, long long total) {
  const float sc = scale ? *scale : 1.f;
  const int Kpad = dgrad ? round_up_d(16 * ld, BK) : round_up_d(4 * ld, BK);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = sc * (dgrad ? up2_dgrad_elem(w, co, ci, ld, Kpad, (unsigned)i)
                         : up2_fwd_elem(w, co, ci, ld, round_up_d(co, 16), Kpad, (unsigned)i));
}

// every packed panel of a model in one launch.  The job table lives in DEVICE memory (it is static for a
// model: built once, no per-launch upload, capturable in a hipGraph): row j = 10 x int64
// {w_oihw, out, co, ci, ld, kh, kw, stride, dgrad, start}; job j owns elements [start_j, start_{j+1}).
constexpr int PACK_ROW = 10;

// One thread packs FOUR consecutive panel elements (panel sizes and therefore job starts are multiples of 16, ld % 4 == 0:
// the four share their filter tap and differ in the channel that is contiguous in the panel) with 32-bit index
// arithmetic and one 16-byte store - the 64-bit divisions per element of the first version were most of its 60 us.
__global__ void pack_multi_kernel(const long long* __restrict__ table, int n, long long total) {
  __shared__ long long T[ITG_PACK_MAX_JOBS * PACK_ROW];
  for (int i = threadIdx.x; i < n * PACK_ROW; i += blockDim.x) T[i] = table[i];
  __syncthreads();
  const long long items = total >> 2;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
    const long long i = it << 2;
    int lo = 0, hi = n - 1;
    while (lo < hi) {                       // last job with start <= i
      int mid = (lo + hi + 1) >> 1;
      if (T[mid * PACK_ROW + 9] <= i) lo = mid; else hi = mid - 1;
    }
    const long long* b = T + lo * PACK_ROW;
    const float* __restrict__ w = reinterpret_cast<const float*>(b[0]);
    float* out = reinterpret_cast<float*>(b[1]);
    const int co = (int)b[2], ci = (int)b[3], ld = (int)b[4], kh = (int)b[5], kw = (int)b[6], stride = (int)b[7];
    const int kind = (int)b[8];
    const unsigned e = (unsigned)(i - b[9]);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kind == 2) {
      const int co_pad = round_up_d(co, 16), Kpad = round_up_d(4 * ld, BK);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = up2_fwd_elem(w, co, ci, ld, co_pad, Kpad, e + j);
    } else if (kind == 3) {
      const int Kpad = round_up_d(16 * ld, BK);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = up2_dgrad_elem(w, co, ci, ld, Kpad, e + j);
    } else if (kind == 4 || kind == 5) {
      // the job's 49 * units items are dealt out in runs of 256: run 49 * u does unit run u (wino_pack4), the other 48 of
      // every 49 runs have nothing to do - the working wavefronts are full and spread over the grid
      const unsigned L = e >> 2, run = L >> 8;
      const unsigned units = (unsigned)(round_up_d(kind == 5 ? ci : co, 16) * round_up_d(ld, BK)) >> 2;
      const unsigned unit = (run / 49u) * 256u + (L & 255u);
      if (run % 49u == 0 && unit < units) wino_pack4<4>(w, out, co, ci, ld, kind == 5, unit);
      continue;
    } else if (kind == 6 || kind == 7) {      // F(4 x 4, 3 x 3): 36 classes
      const unsigned L = e >> 2, run = L >> 8;
      const unsigned units = (unsigned)(round_up_d(kind == 7 ? ci : co, 16) * round_up_d(ld, BK)) >> 2;
      const unsigned unit = (run / 36u) * 256u + (L & 255u);
      if (run % 36u == 0 && unit < units) wino_pack4<3>(w, out, co, ci, ld, kind == 7, unit);
      continue;
    } else if (kind == 8 || kind == 9) {
      // F(4 x 4, 2 x 2) panel of a stride-2 4 x 4 layer (8: forward, row ld = ci_ld) and its transposed panel (9: the adjoint
      // input gradient; row: ld = co_ld, kh = ci_ld): runs of 256 items as for kinds 4 / 5, 25 classes per unit
      const unsigned L = e >> 2, run = L >> 8;
      const int ci_ld = kind == 9 ? kh : ld, co_ld = kind == 9 ? ld : 0;
      const unsigned units = (kind == 9 ? (unsigned)(round_up_d(4 * ci_ld, 16) * round_up_d(co_ld, BK))
                                        : (unsigned)(round_up_d(co, 16) * round_up_d(4 * ci_ld, BK))) >> 2;
      const unsigned unit = (run / 25u) * 256u + (L & 255u);
      if (run % 25u == 0 && unit < units) wino_s2_pack4(w, out, co, ci, ci_ld, co_ld, kind == 9, unit);
      continue;
    } else if (kind == 0) {
      const unsigned Kpad = (unsigned)round_up_d(kh * kw * ld, BK);
      const int k = (int)(e % Kpad), o = (int)(e / Kpad);
      const int tap = k / ld, c = k - tap * ld;
      if (o < co && tap < kh * kw) {
        const float* q = w + ((size_t)o * ci + c) * (kh * kw) + tap;      // [o][c][tap]: the four channels are kh * kw apart
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < ci) v[j] = q[(size_t)j * kh * kw];
      }
    } else {
      const int skh = kh / stride, skw = kw / stride;
      const unsigned Kpad = (unsigned)round_up_d(skh * skw * ld, BK);
      const unsigned ci_pad = (unsigned)round_up_d(ci, 16);
      const int k = (int)(e % Kpad);
      const unsigned r = e / Kpad;
      const int c_in = (int)(r % ci_pad), cls = (int)(r / ci_pad);
      const int ry = cls / stride, rx = cls - ry * stride;
      const int tap = k / ld, o = k - tap * ld;
      if (c_in < ci && tap < skh * skw) {
        const int jy = tap / skw, jx = tap - jy * skw;
        const int ay = (ry + 1) % stride, ax = (rx + 1) % stride;     // pad = 1 for stride-2 convs (ABI)
        const int y = ay + stride * (skh - 1 - jy), x = ax + stride * (skw - 1 - jx);
        const float* q = w + (((size_t)o * ci + c_in) * kh + y) * kw + x;    // the four output channels are ci * kh * kw apart
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (o + j < co) v[j] = q[(size_t)j * ci * kh * kw];
      }
    }
    *reinterpret_cast<f32x4*>(out + e) = v;
  }
}

// ------------------------------------------------------------------------------- single-output-channel convs
// The discriminator's logit layer (512 -> 1, 4x4, stride 1) has ONE output channel: the implicit GEMM
// above would spend 15 of its 16 MFMA rows on zeros (and the weight-gradient 15 of 16 columns).  Here the
// 16 filter taps take the place of the 16 rows:
//   forward   P[i][t] = x[i] . w[t]  for every INPUT pixel i (a 1x1 conv with 16 "channels", K = cin), then
//             y[o] = b + sum_t P[o + t][t]                              (tap_gather_fwd_kernel)
//   wgrad     Q[i][t] = dy[i - t]  (tap_scatter_dy_kernel), dW[t][c] = sum_i Q[i][t] x[i][c] (1x1 wgrad),
//             transposed into the OIHW gradient; the bias gradient is the column of the tap that sees
//             every dy pixel exactly once (ky = pad_h, kx = pad).
__global__ void tap_gather_fwd_kernel(const float* __restrict__ P, int H, int W, GridT out, const float* __restrict__ bias,
                                      const float* __restrict__ scale, int kh, int kw, int pad_h, int pad_w, int act,
                                      float slope) {
  const int64_t total = (int64_t)out.n * out.H * out.W;
  const int ntaps = kh * kw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int ox = (int)(i % out.W);
    int64_t r = i / out.W;
    int oy = (int)(r % out.H);
    int n = (int)(r / out.H);
    float v = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      int iy = oy - pad_h + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        int ix = ox - pad_w + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        v += P[(((size_t)n * H + iy) * W + ix) * ntaps + ky * kw + kx];
      }
    }
    if (scale) v *= *scale;
    if (bias) v += bias[0];
    v = act_apply(v, act, slope);
    *reinterpret_cast<f32x4*>(out.p + grid_off(out, n, oy, ox)) = f32x4{v, 0.f, 0.f, 0.f};
  }
}

__global__ void tap_scatter_dy_kernel(GridT dy, float* __restrict__ Q, int H, int W, int kh, int kw, int pad_h, int pad_w) {
  const int ntaps = kh * kw;
  const int64_t total = (int64_t)dy.n * H * W * ntaps;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int t = (int)(i % ntaps);
    int64_t r = i / ntaps;
    int ix = (int)(r % W); r /= W;
    int iy = (int)(r % H);
    int n = (int)(r / H);
    int ky = t / kw, kx = t - ky * kw;
    int oy = iy + pad_h - ky, ox = ix + pad_w - kx;
    float v = 0.f;
    if ((unsigned)oy < (unsigned)dy.H && (unsigned)ox < (unsigned)dy.W) v = dy.p[grid_off(dy, n, oy, ox)];
    Q[i] = v;
  }
}

// dw[c][t] (+)= tmp[t][c]; db[0] (+)= dbtmp[tdb]
__global__ void tap_wgrad_finish_kernel(const float* __restrict__ tmp, const float* __restrict__ dbtmp, float* __restrict__ dw,
                                        float* __restrict__ db, int ci, int ntaps, int tdb, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ci * ntaps) {
    int c = i / ntaps, t = i - c * ntaps;
    float v = tmp[(size_t)t * ci + c];
    dw[i] = (accumulate & ITG_ACC_DW) ? dw[i] + v : v;
  }
  if (i == 0 && db) db[0] = (accumulate & ITG_ACC_DB) ? db[0] + dbtmp[tdb] : dbtmp[tdb];
}


// ------------------------------------------------------------------------------- few-input-channel stride-2 convs
// Input gradient of the discriminator's first layer (3 -> 64, 4x4, stride 2): the parity-class implicit
// GEMM would fill 3 of its 16 MFMA rows.  Instead Q[o][(class, c, tap')] = dy[o] . w[., c, tap] for every
// OUTPUT pixel o (a 1x1 conv with 4*c*4 <= 64 rows, K = cout), then each input pixel gathers its 4 taps.
// compact panel row r = (cls * 4 + tap') * 4 + c  <-  dgrad panel row (cls * ci_pad + c), columns tap' * co_ld ..
// (c padded to 4: the gather then reads one 16-byte vector per tap)
__global__ void thin_dgrad_panel_kernel(const float* __restrict__ wd, float* __restrict__ out, int cin, int ci_pad, int co_ld,
                                        int rows_pad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_pad * co_ld) return;
  int o = i % co_ld, r = i / co_ld;
  float v = 0.f;
  if (r < 64) {
    int c = r & 3, tap = (r >> 2) & 3, cls = r >> 4;
    if (c < cin) v = wd[((size_t)(cls * ci_pad + c) * 4 + tap) * co_ld + o];
  }
  out[i] = v;
}

__global__ void thin_dgrad_gather_kernel(const float* __restrict__ Q, int Ho, int Wo, int qld, GridT dx, GridT act_out,
                                         int act, float slope) {
  const int64_t total = (int64_t)dx.n * dx.H * dx.W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int ix = (int)(i % dx.W);
    int64_t r = i / dx.W;
    int iy = (int)(r % dx.H);
    int n = (int)(r / dx.H);
    const int ry = iy & 1, rx = ix & 1, t = iy >> 1, u = ix >> 1;
    const int by = (ry + 1 - ((ry + 1) & 1)) >> 1, bx = (rx + 1 - ((rx + 1) & 1)) >> 1;
    const int cls = ry * 2 + rx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) {
      int oy = t + by - 1 + jy;
      if ((unsigned)oy >= (unsigned)Ho) continue;
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) {
        int ox = u + bx - 1 + jx;
        if ((unsigned)ox >= (unsigned)Wo) continue;
        v += *reinterpret_cast<const f32x4*>(Q + (((size_t)n * Ho + oy) * Wo + ox) * qld + (cls * 4 + jy * 2 + jx) * 4);
      }
    }
    const int off = grid_off(dx, n, iy, ix);
    if (act_out.p) v *= act_deriv(*reinterpret_cast<const f32x4*>(act_out.p + grid_off(act_out, n, iy, ix)), act, slope);
    *reinterpret_cast<f32x4*>(dx.p + off) = v;
  }
}

// ------------------------------------------------------------------------------- single-input-channel 3x3 convs
// The first conv of an SSM modulation MLP (reference models/layers.py:212-215,231: conv3x3(map_dim = 1 -> 128) on the noise
// map of every patch).  As an implicit GEMM it has K = 9 (padded to 48): 8 TF at 5 % of the MFMA peak, and at config 5's
// size it is 15 % of the whole generation.  It is a WRITE-bound operator (1 input float, 128 outputs per pixel): 32 lanes
// own one output pixel's 128 channels (4 each, their 36 filter taps in registers), a workgroup an 8 x 32 output tile whose
// (8+2) x (32+2) input window sits in LDS; every lane slides a 3 x 3 register window along its row: per pixel 3 LDS
// broadcasts, 36 fma, one 16-byte store - 1 KB contiguous per wave-instruction.
constexpr int C1_TH = 8, C1_TW = 32;

__global__ __launch_bounds__(256) void conv_cin1_kernel(GridT in, GridT out, const float* __restrict__ w, int Kpad,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        int act, float slope, int tiles_x, int tiles_y) {
  __shared__ float xt[(C1_TH + 2) * (C1_TW + 2)];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx_i = b % tiles_x; b /= tiles_x;
  const int ty_i = b % tiles_y;
  const int n = b / tiles_y;
  const int y0 = ty_i * C1_TH, x0 = tx_i * C1_TW;
  for (int e = tid; e < (C1_TH + 2) * (C1_TW + 2); e += 256) {
    const int r = e / (C1_TW + 2), c = e - r * (C1_TW + 2);
    const int iy = y0 + r, ix = x0 + c;                      // valid conv: input pixel (y + ky, x + kx)
    xt[e] = (iy < in.H && ix < in.W) ? in.p[grid_off(in, n, iy, ix)] : 0.f;
  }
  const int cg = tid & 31, row = tid >> 5;                   // channel group (4 channels), tile row
  const int co = cg * 4;
  const bool live = co < out.ld;
  float wr[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[t][e] = (live && co + e < out.c) ? w[(size_t)(co + e) * Kpad + t * in.ld] : 0.f;
  const float osc = scale ? *scale : 1.f;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) bv[e] = (bias && live && co + e < out.c) ? bias[co + e] : 0.f;
  __syncthreads();
  const int oy = y0 + row;
  if (!live || oy >= out.H) return;
  const float* xr = xt + row * (C1_TW + 2);
  float w0[3], w1[3], w2[3];                                   // the window's three columns (rows ky = 0..2)
#pragma unroll
  for (int k = 0; k < 3; ++k) { w0[k] = xr[k * (C1_TW + 2)]; w1[k] = xr[k * (C1_TW + 2) + 1]; }
  // output addresses: one division per row when the tile lies in one patch column (always for a 1 x 1 grid)
  const bool linear = out.gw == 1 || (x0 % out.pw) + C1_TW <= out.pw;
  float* const orow = out.p + grid_off(out, n, oy, min(x0, out.W - 1)) + co;
#pragma unroll 4
  for (int c = 0; c < C1_TW; ++c) {
#pragma unroll
    for (int k = 0; k < 3; ++k) w2[k] = xr[k * (C1_TW + 2) + c + 2];
    const int ox = x0 + c;
    if (ox < out.W) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          a = fmaf(w0[k], wr[3 * k][e], a);
          a = fmaf(w1[k], wr[3 * k + 1][e], a);
          a = fmaf(w2[k], wr[3 * k + 2][e], a);
        }
        v[e] = act_apply(a * osc + bv[e], act, slope);
        if (co + e >= out.c) v[e] = 0.f;
      }
      float* const dst = linear ? orow + (size_t)c * out.ld : out.p + grid_off(out, n, oy, ox) + co;
      *reinterpret_cast<f32x4*>(dst) = v;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { w0[k] = w1[k]; w1[k] = w2[k]; }
  }
}

inline bool cin1_conv(const itg_tensor* in, const itg_tensor* out, const itg_conv_geom* g) {
  const int enable = kernel_on(KM_CIN1);
  return enable && in->c == 1 && in->ld == 4 && g->kh == 3 && g->kw == 3 && g->stride == 1 && g->pad == 0 && pad_v_raw(g) == 0 &&
         out->ld <= 128 && out->c >= 16 && g->precision != ITG_PREC_BF16 && !g->out_stats;
}

// Round 6: the same input gradient in ONE launch.  The three-launch form above writes Q (one 64-float row per dy pixel: 75 MB on the
// generated 384^2 batch) and reads it back - 3 x the bytes of an operation that has to read dy once (75 MB) and write dx (19 MB): 80 us
// where HBM needs ~20.  Here a workgroup owns an 8 x 16 tile of dy pixels plus a 1-pixel halo (180 pixels), computes their Q rows on the
// MFMA pipe straight from global memory (B operand = dy pixels, 16-byte loads in the K-permuted fragment layout; A operand = the 64 x
// co_ld panel, built in LDS once per workgroup from the packed input-gradient panel), parks Q in LDS (pitch 68: conflict-free for the
// 16-lane groups of a ds_read_b128) and gathers the 16 x 32 dx pixels of its tile from it: dy is read once from HBM (the halo from L2),
// Q never leaves the CU.  Out-of-image dy pixels load as zeros through the buffer range check.  fp32 operands whatever the launch's MFMA
// precision (2.4 GF: not worth a second instantiation).
constexpr int TD_TH = 8, TD_TW = 16, TD_HP = TD_TH + 2, TD_WP = TD_TW + 2, TD_PIX = TD_HP * TD_WP, TD_ROWS = 48, TD_QP = TD_ROWS + 4;
template <int NKB>      // dy.ld / 16
__global__ __launch_bounds__(256) void thin_dgrad_fused_kernel(const GridT dy, const float* __restrict__ wd, const float* __restrict__ scale,
                                                               const GridT dx, const GridT act_out, int act, float slope, int cin,
                                                               int ci_pad, int tiles_x, int tiles_y, int ntiles, unsigned dy_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds_td[];
  constexpr int co_ld = 16 * NKB, WP = co_ld + 4;
  float* Wl = lds_td;                    // [48][WP]: row (c, cls, tap) - one 16-row MFMA fragment per input channel - k = dy channel
  float* Ql = lds_td + TD_ROWS * WP;     // [TD_PIX][TD_QP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < TD_ROWS * co_ld; e += 256) {
    const int o = e % co_ld, r = e / co_ld;
    const int c = r >> 4, cls = (r >> 2) & 3, tap = r & 3;
    Wl[r * WP + o] = c < cin ? wd[((size_t)(cls * ci_pad + c) * 4 + tap) * co_ld + o] : 0.f;
  }
  const float sc = scale ? *scale : 1.f;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dy.p, 0, dy_bytes, 0x00020000);
  const int fr = lane & 15, g = lane >> 4;
  // the tile's 3 x NKB operand loads of this lane (all in flight at once; the NEXT tile's are issued before this tile's gather)
  f32x4 bq[3][NKB];
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * TD_TH - 1, ox0 = tx * TD_TW - 1;          // dy coordinates of the tile's local (0, 0)
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const int slot = (wave * 3 + f) * 16 + fr;
      const int ly = slot / TD_WP, lx = slot - ly * TD_WP;
      const int oy = oy0 + ly, ox = ox0 + lx;
      const bool ok = slot < TD_PIX && (unsigned)oy < (unsigned)dy.H && (unsigned)ox < (unsigned)dy.W;
      const unsigned po = ok ? (unsigned)grid_off(dy, n, oy, ox) * 4u + (unsigned)g * 16u : dy_bytes;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        bq[f][kb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, po + (unsigned)kb * 64u, 0, 0));
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int f = 0; f < 3; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      f32x4 a[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const f32x4*>(Wl + (16 * i + fr) * WP + kb * 16 + 4 * g);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int f = 0; f < 3; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s4], bq[f][kb][s4], acc[i][f], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const int slot = (wave * 3 + f) * 16 + fr;
      if (slot < TD_PIX) {
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(Ql + slot * TD_QP + 16 * i + 4 * g) = acc[i][f];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);      // in flight under the gather below
    for (int e = tid; e < 4 * TD_TH * TD_TW; e += 256) {
      const int lyy = e / (2 * TD_TW), lxx = e - lyy * (2 * TD_TW);
      const int Y = 2 * ty * TD_TH + lyy, X = 2 * tx * TD_TW + lxx;
      if (Y >= dx.H || X >= dx.W) continue;
      const int ry = Y & 1, rx = X & 1, t = lyy >> 1, u = lxx >> 1;
      const int by = (ry + 1 - ((ry + 1) & 1)) >> 1, bx = (rx + 1 - ((rx + 1) & 1)) >> 1;
      const int cls = ry * 2 + rx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jy = 0; jy < 2; ++jy)
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) {
          const float* q = Ql + ((t + by + jy) * TD_WP + (u + bx + jx)) * TD_QP + cls * 4 + jy * 2 + jx;
          v[0] += q[0]; v[1] += q[16]; v[2] += q[32];           // rows (c, cls, tap)
        }
      v *= sc;
      const int off = grid_off(dx, n, Y, X);
      if (act_out.p) v *= act_deriv(*reinterpret_cast<const f32x4*>(act_out.p + grid_off(act_out, n, Y, X)), act, slope);
      *reinterpret_cast<f32x4*>(dx.p + off) = v;
    }
    __syncthreads();                     // Ql is rewritten by the next tile
  }
}

// Round 6: input gradient of a SINGLE-output-channel 4 x 4 stride-1 pad-1 layer (D's logit layer, 512 -> 1).  dx[pix][c] = sum over the 16
// taps of dy[pix + tap] w[c][tap] (x act'(x)): 16 multiply-adds per output element, 72 MB of x read + dx written on the generated batch - a
// streaming operation that the implicit GEMM ran at 2.1 TB/s (K = 16 taps padded to 64, four K stages of prologue per 128 x 96 tile).  Here the
// MFMA does the 16-deep contraction with the roles swapped to keep the stores wide: A = the filter (16 channels per fragment, K = taps, 40 KB
// of LDS for 512 channels), B = a row of 16 pixels' tap windows gathered from a 7 x 19 window of dy in LDS; a wave owns 16 pixels x CH channels
// and streams x in / dx out as 16-byte accesses with the next fragments' loads in flight.  A workgroup = 4 image rows x 16 pixels x CH (<= 128)
// channels, so that the grid has >= 1 000 workgroups on the 47 x 47 maps.
constexpr int LG_TW = 16, LG_TH = 4, LG_WW = LG_TW + 3, LG_WH = LG_TH + 3;
__global__ __launch_bounds__(256) void logit_dgrad_kernel(const GridT dy, const float* __restrict__ panel, int Kpad, const float* __restrict__ scale,
                                                          const GridT dx, const GridT act_out, int act, float slope, int CH, int tiles_x,
                                                          int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float lds_lg[];
  float* Wl = lds_lg;                         // [CH][20]: taps of channel c0 + r
  float* Dl = lds_lg + CH * 20;               // [LG_WH][LG_WW] window of dy (its one channel)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int c0 = blockIdx.y * CH;
  for (int e = tid; e < CH * 16; e += 256) {
    const int r = e >> 4, t = e & 15;
    Wl[r * 20 + t] = c0 + r < dx.c ? panel[(size_t)(c0 + r) * Kpad + t * 4] : 0.f;
  }
  const int y0 = ty * LG_TH, x0 = tx * LG_TW;
  for (int e = tid; e < LG_WH * LG_WW; e += 256) {
    const int wy = e / LG_WW, wx = e - wy * LG_WW;
    const int iy = y0 + wy - 2, ix = x0 + wx - 2;
    Dl[e] = ((unsigned)iy < (unsigned)dy.H && (unsigned)ix < (unsigned)dy.W) ? dy.p[grid_off(dy, n, iy, ix)] : 0.f;
  }
  __syncthreads();
  const int fr = lane & 15, g = lane >> 4;
  const int Y = y0 + wave, X = x0 + fr;
  if (Y >= dx.H) return;                       // (wave-uniform)
  // B fragment: pixel fr of row `wave`, K group g = tap row g: dy[Y - 2 + g][X - 2 + 0..3]
  f32x4 bq;
#pragma unroll
  for (int e = 0; e < 4; ++e) bq[e] = Dl[(wave + g) * LG_WW + fr + e];
  const float sc = scale ? *scale : 1.f;
  const bool ok = X < dx.W;
  const int off = ok ? grid_off(dx, n, Y, X) : 0;
  const int nf = CH >> 4;
  for (int i0 = 0; i0 < nf; i0 += 4) {         // four fragments (64 channels) at a time: their act_out loads in flight together
    f32x4 r[4], acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = c0 + 16 * (i0 + k) + 4 * g;
      r[k] = (act_out.p && ok && i0 + k < nf && co < dx.ld) ? *reinterpret_cast<const f32x4*>(act_out.p + grid_off(act_out, n, Y, X) + co)
                                                            : f32x4{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i0 + k < nf) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(Wl + (16 * (i0 + k) + fr) * 20 + 4 * g);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], bq[s4], acc[k], 0, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = c0 + 16 * (i0 + k) + 4 * g;
      if (!ok || i0 + k >= nf || co >= dx.ld) continue;
      f32x4 v = acc[k] * sc;
      if (act_out.p) v *= act_deriv(r[k], act, slope);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (co + e >= dx.c) v[e] = 0.f;
      *reinterpret_cast<f32x4*>(dx.p + off + co) = v;
    }
  }
}

inline bool logit_dgrad_ok(const itg_tensor* dy, const itg_tensor* dx, const itg_conv_geom* g) {
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  return kernel_on(KM_THIN_CONV) && dy->c == 1 && dy->ld == 4 && g->kh == 4 && g->kw == 4 && g->stride == 1 && g->pad == 1 && ph == 1 &&
         g->pad_mode == ITG_PAD_ZERO && !g->up2 && !(g->flags & ITG_GEOM_WINO) && (dx->ld % 16) == 0 && dy->gh == 1 && dy->gw == 1;
}

inline bool thin_in_conv(const itg_tensor* dy, const itg_tensor* dx, const itg_conv_geom* g) {
  const int enable = kernel_on(KM_THIN_CONV);
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  return enable && dx->c <= 4 && dx->ld == 4 && g->kh == 4 && g->kw == 4 && g->stride == 2 && g->pad == 1 && ph == 1 &&
         g->pad_mode == ITG_PAD_ZERO && (dy->ld % 16) == 0;
}

inline bool thin_out_conv(const itg_tensor* in, const itg_tensor* out, const itg_conv_geom* g) {
  const int enable = kernel_on(KM_THIN_CONV);
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  return enable && out->c == 1 && g->kh * g->kw == 16 && g->stride == 1 && g->pad_mode == ITG_PAD_ZERO &&
         (in->ld % 16) == 0 && ph < g->kh && g->pad < g->kw && 2 * ph <= g->kh - 1 && 2 * g->pad <= g->kw - 1;
}

int conv_out_dim(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }
// vertical padding may differ from the horizontal one (row-sharded patch grids carry their halo rows
// explicitly and pad only the columns): pad_h < 0 means "same as pad"
inline int pad_v(const itg_conv_geom* g) { return g->pad_h >= 0 ? g->pad_h : g->pad; }
inline int prec_of(const itg_conv_geom* g) { return g->precision == ITG_PREC_BF16 ? ITG_PREC_BF16 : ITG_PREC_F32; }

// itg_conv_geom.up2: 3x3, stride 1, pad 1 on the x2 upsample of `lo` (half the patch extent of `hi`, same grid).
// pad_h = 0 (row-sharded bands, 1-row grids): `lo` carries one explicit halo row of SOURCE pixels above and below.
inline int up2_check(const itg_conv_geom* g, const itg_tensor* lo, const itg_tensor* hi) {
  const int pv = pad_v_raw(g);
  if (g->kh != 3 || g->kw != 3 || g->stride != 1 || g->pad != 1 || (pv != 0 && pv != 1)) return ITG_ERR_ARG;
  if (lo->n != hi->n || lo->gh != hi->gh || lo->gw != hi->gw || 2 * lo->pw != hi->pw) return ITG_ERR_ARG;
  if (pv == 1 ? 2 * lo->ph != hi->ph : (lo->gh != 1 || 2 * (lo->ph - 2) != hi->ph)) return ITG_ERR_ARG;
  return ITG_OK;
}
inline void clear_xf(ConvP& p) {
  p.ucls = 0; p.u_in = p.u_w = p.u_out = 0; p.u_acc = 0;
}
// itg_bn_bwd_fuse of an input gradient: validate and hand to the launch (dispatch_nt decides whether its second stage takes the sums)
inline int set_bnb(ConvP& p, const itg_conv_geom* g, const itg_tensor* dx) {
  itg_bn_bwd_fuse* f = g->bn_bwd;
  if (!f) return ITG_OK;
  f->taken = 0;
  int rc;
  if (!f->x || (rc = check_tensor(f->x))) return ITG_ERR_ARG;
  if (!same_shape(f->x, dx) || !f->ab || !f->mean_rstd || !f->sums || (((uintptr_t)f->sums) & 7)) return ITG_ERR_ARG;
  p.bnb_x = (const float*)f->x->ptr; p.bnb_ab = f->ab; p.bnb_mr = f->mean_rstd; p.bnb_sums = f->sums;
  p.bnb_act = f->act; p.bnb_slope = f->slope;
  return ITG_OK;
}


}  // namespace


// =============================================================================== C ABI
extern "C" {

int itg_version(void) { return 100; }

const char* itg_last_conv_kernel(void) { return g_last_launch; }

int64_t itg_pack_fwd_size(int co, int ci_ld, int kh, int kw) {
  return (int64_t)round_up(co, 16) * round_up(kh * kw * ci_ld, BK);
}

int64_t itg_pack_dgrad_size(int ci, int co_ld, int kh, int kw, int stride) {
  int taps = (kh / stride) * (kw / stride);
  return (int64_t)stride * stride * round_up(ci, 16) * round_up(taps * co_ld, BK);
}

int itg_pack_fwd(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, int kh, int kw,
                 void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || ci_ld < ci || (ci_ld & 3)) return ITG_ERR_ARG;
  int co_pad = round_up(co, 16), Kpad = round_up(kh * kw * ci_ld, BK);
  int64_t total = (int64_t)co_pad * Kpad;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci, ci_ld,
                     kh, kw, co_pad, Kpad);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_pack_dgrad(const float* w, const float* scale, float* out, int co, int ci, int co_ld, int kh, int kw,
                   int stride, void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || co_ld < co || (co_ld & 3)) return ITG_ERR_ARG;
  if (stride != 1 && stride != 2) return ITG_ERR_ARG;
  if (stride == 2 && ((kh & 1) || (kw & 1))) return ITG_ERR_ARG;
  // pad enters only through parity (ry+pad)&1 for stride 2; the ABI fixes pad = 1 for stride-2 convs
  int pad = 1;
  int ci_pad = round_up(ci, 16);
  int taps = (kh / stride) * (kw / stride);
  int Kpad = round_up(taps * co_ld, BK);
  int64_t total = (int64_t)stride * stride * ci_pad * Kpad;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci,
                     co_ld, kh, kw, stride, pad, ci_pad, Kpad);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int64_t itg_pack_up2_fwd_size(int co, int ci_ld) { return (int64_t)4 * round_up(co, 16) * round_up(4 * ci_ld, BK); }

int64_t itg_pack_up2_dgrad_size(int ci, int co_ld) { return (int64_t)round_up(ci, 16) * round_up(16 * co_ld, BK); }

static int pack_up2(const float* w, const float* scale, float* out, int co, int ci, int ld, int dgrad, void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || (ld & 3) || ld < (dgrad ? co : ci)) return ITG_ERR_ARG;
  const int64_t total = dgrad ? itg_pack_up2_dgrad_size(ci, ld) : itg_pack_up2_fwd_size(co, ld);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_up2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci, ld, dgrad,
                     (long long)total);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_pack_up2_fwd(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, void* stream) {
  return pack_up2(w, scale, out, co, ci, ci_ld, 0, stream);
}

int itg_pack_up2_dgrad(const float* w, const float* scale, float* out, int co, int ci, int co_ld, void* stream) {
  return pack_up2(w, scale, out, co, ci, co_ld, 1, stream);
}

int64_t itg_pack_wino_size(int rows, int k_ld) { return (int64_t)49 * round_up(rows, 16) * round_up(k_ld, BK); }
int64_t itg_pack_wino3_size(int rows, int k_ld) { return (int64_t)36 * round_up(rows, 16) * round_up(k_ld, BK); }

static int pack_wino(const float* w, const float* scale, float* out, int co, int ci, int ld, int dgrad, int R, void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || (ld & 3) || ld < (dgrad ? co : ci)) return ITG_ERR_ARG;
  const int64_t total = R == 4 ? itg_pack_wino_size(dgrad ? ci : co, ld) : itg_pack_wino3_size(dgrad ? ci : co, ld);
  if (total >= ((int64_t)1 << 32)) return ITG_ERR_ARG;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (R == 4)
    hipLaunchKernelGGL(pack_wino_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci, ld, dgrad,
                       (long long)total);
  else
    hipLaunchKernelGGL(pack_wino_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci, ld, dgrad,
                       (long long)total);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int64_t itg_pack_wino_s2_size(int co, int ci_ld) { return (int64_t)25 * round_up(co, 16) * round_up(4 * ci_ld, BK); }

int itg_pack_wino_s2_fwd(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || (ci_ld & 3) || ci_ld < ci) return ITG_ERR_ARG;
  const int64_t total = itg_pack_wino_s2_size(co, ci_ld);
  if (total >= ((int64_t)1 << 32)) return ITG_ERR_ARG;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_wino_s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci, ci_ld, (long long)total);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int64_t itg_pack_wino_s2_dgrad_size(int ci_ld, int co_ld) { return (int64_t)25 * round_up(4 * ci_ld, 16) * round_up(co_ld, BK); }

int itg_pack_wino_s2_dgrad(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, int co_ld, void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || (ci_ld & 3) || ci_ld < ci || (co_ld & 3) || co_ld < co) return ITG_ERR_ARG;
  const int64_t total = itg_pack_wino_s2_dgrad_size(ci_ld, co_ld);
  if (total >= ((int64_t)1 << 32)) return ITG_ERR_ARG;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_wino_s2_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci, ci_ld, co_ld,
                     (long long)total);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_pack_wino3_fwd(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, void* stream) {
  return pack_wino(w, scale, out, co, ci, ci_ld, 0, 3, stream);
}

int itg_pack_wino3_dgrad(const float* w, const float* scale, float* out, int co, int ci, int co_ld, void* stream) {
  return pack_wino(w, scale, out, co, ci, co_ld, 1, 3, stream);
}

int itg_pack_wino_fwd(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, void* stream) {
  return pack_wino(w, scale, out, co, ci, ci_ld, 0, 4, stream);
}

int itg_pack_wino_dgrad(const float* w, const float* scale, float* out, int co, int ci, int co_ld, void* stream) {
  return pack_wino(w, scale, out, co, ci, co_ld, 1, 4, stream);
}

int itg_pack_multi(const int64_t* table_dev, int n, int64_t total, void* stream) {
  if (!table_dev || n <= 0 || n > ITG_PACK_MAX_JOBS || total <= 0) return ITG_ERR_ARG;
  if (total & 3) return ITG_ERR_ARG;
  const int64_t items = total >> 2;
  int blocks = (int)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)table_dev, n,
                     (long long)total);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// ITG_GEOM_WINO: geometry the Winograd pipeline takes (conv_wino.hip)
// 4 x 4 stride 1 pad 1 with zero padding (R = 4), or 3 x 3 stride 1 pad 1 with zero or replicate padding (R = 3)
static inline bool wino_geom(const itg_conv_geom* g) {
  if (!(g->flags & ITG_GEOM_WINO) || g->kh != g->kw || g->stride != 1 || g->pad != 1 || pad_v_raw(g) != 1 || g->up2)
    return false;
  return (g->kh == 4 && g->pad_mode == ITG_PAD_ZERO) || (g->kh == 3 && (g->pad_mode == ITG_PAD_ZERO || g->pad_mode == ITG_PAD_REPLICATE));
}
// ... and the FORWARD-only stride-2 form: 4 x 4, stride 2, pad 1, zero padding (F(4 x 4, 2 x 2) on the parity classes)
static inline bool wino_s2_geom(const itg_conv_geom* g) {
  return (g->flags & ITG_GEOM_WINO) && g->kh == 4 && g->kw == 4 && g->stride == 2 && g->pad == 1 && pad_v_raw(g) == 1 && !g->up2 &&
         g->pad_mode == ITG_PAD_ZERO && prec_of(g) == ITG_PREC_F32;
}
// the input gradient of a replicate-padded layer is evaluated on the padded extent and folded (conv_wino.hip)
static inline int wino_fold(const itg_conv_geom* g) { return g->pad_mode == ITG_PAD_REPLICATE ? 1 : 0; }

int64_t itg_conv2d_fwd_workspace(const itg_tensor* in, const itg_tensor* out, const itg_conv_geom* g) {
  if (!in || !out || !g) return 0;
  if (wino_s2_geom(g)) return wino_s2_workspace_floats(in, out);
  if (g->flags & ITG_GEOM_WINO) return wino_geom(g) ? wino_workspace_floats(in, out, g->kh, 0) : 0;
  if (g->up2) return plan_nt(grid_pixels(out), round_up(out->c, 16), round_up(4 * in->ld, BK), 4, prec_of(g)).ws_floats;
  if (thin_out_conv(in, out, g)) {
    int64_t Min = grid_pixels(in);
    return Min * 16 + plan_nt(Min, 16, round_up(in->ld, BK), 1, prec_of(g)).ws_floats;
  }
  return plan_nt(grid_pixels(out), round_up(out->c, 16), round_up(g->kh * g->kw * in->ld, BK), 1, prec_of(g)).ws_floats;
}

int64_t itg_conv2d_dgrad_workspace(const itg_tensor* dy, const itg_tensor* dx, const itg_conv_geom* g) {
  if (!dy || !dx || !g) return 0;
  if (wino_s2_geom(g)) return wino_s2_dgrad_workspace_floats(dy, dx);
  if (g->flags & ITG_GEOM_WINO) return wino_geom(g) ? wino_workspace_floats(dy, dx, g->kh, wino_fold(g)) : 0;
  if (thin_in_conv(dy, dx, g)) {
    const int rows = 64;
    const int64_t Mo = grid_pixels(dy);
    return Mo * rows + (int64_t)rows * dy->ld + plan_nt(Mo, rows, round_up(dy->ld, BK), 1, prec_of(g)).ws_floats;
  }
  int co_rows = round_up(dx->c, 16);
  int64_t H = (int64_t)dx->gh * dx->ph, W = (int64_t)dx->gw * dx->pw;
  if (g->up2) {
    const int e = g->pad_mode == ITG_PAD_REPLICATE ? 2 : 0;
    return plan_nt((int64_t)dx->n * (H + (pad_v(g) ? e : 0)) * (W + e), co_rows, round_up(16 * dy->ld, BK), 1, prec_of(g)).ws_floats;
  }
  if (g->stride == 1) {
    int eh = (g->pad_mode == ITG_PAD_REPLICATE) ? 2 * pad_v(g) : 0, ew = (g->pad_mode == ITG_PAD_REPLICATE) ? 2 * g->pad : 0;
    return plan_nt((int64_t)dx->n * (H + eh) * (W + ew), co_rows, round_up(g->kh * g->kw * dy->ld, BK), 1, prec_of(g)).ws_floats;
  }
  int Kpad = round_up((g->kh / 2) * (g->kw / 2) * dy->ld, BK);
  int64_t Mmax = (int64_t)dx->n * ((H + 1) / 2) * ((W + 1) / 2);
  return plan_nt(Mmax * 4, co_rows, Kpad, 4, prec_of(g)).ws_floats;
}

int itg_conv2d_fwd(const itg_tensor* in, const float* w_packed, const float* bias, const float* out_scale,
                   const itg_tensor* residual, const itg_tensor* out, const itg_conv_geom* g, int act, float slope,
                   float* workspace, int64_t workspace_floats, void* stream) {
  int rc;
  if ((rc = check_tensor(in)) || (rc = check_tensor(out))) return rc;
  if (!w_packed || !g || g->kh <= 0 || g->kw <= 0 || g->stride <= 0 || g->pad < 0) return ITG_ERR_ARG;
  if (g->bn_bwd) return ITG_ERR_ARG;
  if (wino_s2_geom(g)) {
    if (residual && residual->ptr) return ITG_ERR_ARG;
    if ((rc = wino_conv_s2(in, w_packed, bias, out_scale, out, act, slope, prec_of(g), workspace, workspace_floats, (hipStream_t)stream)))
      return rc;
    return g->out_stats ? itg_bn_stats(out, g->out_stats, stream) : ITG_OK;
  }
  if (g->flags & ITG_GEOM_WINO) {
    if (!wino_geom(g)) return ITG_ERR_ARG;
    const itg_tensor* r = (residual && residual->ptr) ? residual : nullptr;
    if (r && (rc = check_tensor(r))) return rc;
    int rups = 0;
    if (r && !same_shape(r, out)) {          // half-extent residual: read through a nearest x2 upsample (see below)
      if (r->gh != out->gh || r->gw != out->gw || r->c != out->c || 2 * r->ph != out->ph || 2 * r->pw != out->pw) return ITG_ERR_ARG;
      rups = 1;
    }
    if ((rc = wino_conv(in, w_packed, bias, out_scale, r, rups, 0, 0.f, out, g->kh, 1, g->pad_mode, 0, act, slope, prec_of(g), workspace,
                        workspace_floats, (hipStream_t)stream))) return rc;
    // the consumer BatchNorm's statistics: their own pass over the finished output
    return g->out_stats ? itg_bn_stats(out, g->out_stats, stream) : ITG_OK;
  }
  if (g->up2) {
    // four output-parity classes in one grid: class (ry, rx) = output pixels (2y + ry, 2x + rx), a 2 x 2 conv of the source
    // tensor whose taps start at (y + ry - 1, x + rx - 1); the frame clamps / predicates in SOURCE coordinates, which is
    // the upsampled image's replicate / zero padding (its row -1 is source row -1)
    if ((rc = up2_check(g, in, out))) return rc;
    ConvP p;
    clear_xf(p);
    p.prec = prec_of(g);
    p.in = make_grid(in); p.out = make_grid(out);
    p.res = null_grid(); p.res_ups = 0; p.res_mode = 0; p.res_slope = 0.f;
    if (residual && residual->ptr) {
      if ((rc = check_tensor(residual))) return rc;
      if (!same_shape(residual, out)) return ITG_ERR_ARG;
      p.res = make_grid(residual);
    }
    p.w = w_packed; p.bias = bias; p.scale = out_scale;
    p.stats = g->out_stats;
    if (p.stats && out->ld > 512) return ITG_ERR_ARG;
    p.ntaps = 4; p.kw = 2; p.cin_ld = in->ld;
    p.Kpad = round_up(4 * in->ld, BK);
    p.co_rows = round_up(out->c, 16);
    p.isy = p.isx = 1; p.osy = p.osx = 2;
    p.pad_mode = g->pad_mode; p.out_mode = 0; p.act = act; p.slope = slope;
    const int pv = pad_v(g), Hs = p.out.H / 2;                  // source rows that produce output (pv = 0: in.H - 2)
    const int64_t M = (int64_t)in->n * Hs * p.in.W;
    if (M >= ((int64_t)1 << 29)) return ITG_ERR_ARG;
    p.ncls = 4;
    for (int c = 0; c < 4; ++c) {
      const int ry = c >> 1, rx = c & 1;
      p.cMT[c] = Hs; p.cMU[c] = p.in.W; p.cM[c] = (int)M;
      p.cioy[c] = ry - pv; p.ciox[c] = rx - 1; p.cooy[c] = ry; p.coox[c] = rx;
      p.cwoff[c] = (unsigned)((size_t)c * p.co_rows * p.Kpad);
    }
    p.M = (int)M; p.MT = Hs; p.MU = p.in.W;
    p.ioy = p.cioy[0]; p.iox = p.ciox[0]; p.ooy = 0; p.oox = 0;
    return dispatch_nt(p, workspace, workspace_floats, (hipStream_t)stream);
  }
  if (cin1_conv(in, out, g) && !(residual && residual->ptr)) {
    const int H = in->gh * in->ph, W = in->gw * in->pw;
    if (in->n != out->n || H - 2 != out->gh * out->ph || W - 2 != out->gw * out->pw) return ITG_ERR_ARG;
    const GridT gi = make_grid(in), go = make_grid(out);
    const int tiles_x = (go.W + C1_TW - 1) / C1_TW, tiles_y = (go.H + C1_TH - 1) / C1_TH;
    const int64_t blocks = (int64_t)out->n * tiles_x * tiles_y;
    if (blocks > 0x7fffffff) return ITG_ERR_ARG;
    snprintf(g_last_launch, sizeof(g_last_launch), "conv_cin1_kernel");
    hipLaunchKernelGGL(conv_cin1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gi, go, w_packed,
                       round_up(9 * in->ld, BK), bias, out_scale, act, slope, tiles_x, tiles_y);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  if (thin_out_conv(in, out, g) && !(residual && residual->ptr)) {
    if (g->out_stats) return ITG_ERR_ARG;       // single-output-channel layers have no BatchNorm consumer on this path
    // taps-as-rows path (see tap_gather_fwd_kernel): a 1x1 conv into P[pixel][16], then the tap gather
    const int H = in->gh * in->ph, W = in->gw * in->pw;
    if (in->n != out->n || conv_out_dim(H, g->kh, 1, pad_v(g)) != out->gh * out->ph ||
        conv_out_dim(W, g->kw, 1, g->pad) != out->gw * out->pw) return ITG_ERR_ARG;
    const int64_t Min = grid_pixels(in), pf = Min * 16;
    if (!workspace || workspace_floats < pf || pf >= ((int64_t)1 << 31)) return ITG_ERR_WORKSPACE;
    itg_tensor P = {workspace, in->n, 1, 1, H, W, 16, 16};
    itg_conv_geom g1 = {1, 1, 1, 0, ITG_PAD_ZERO, 0, prec_of(g)};
    // row 0 of the packed filter is (tap, ci)-ordered with ci_ld a multiple of 16: read as 16 rows of ci_ld
    if ((rc = itg_conv2d_fwd(in, w_packed, nullptr, nullptr, nullptr, &P, &g1, ITG_ACT_NONE, 0.f, workspace + pf,
                             workspace_floats - pf, stream))) return rc;
    GridT og = make_grid(out);
    int64_t total = (int64_t)og.n * og.H * og.W;
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(tap_gather_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, H, W,
                       og, bias, out_scale, g->kh, g->kw, pad_v(g), g->pad, act, slope);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  ConvP p;
  clear_xf(p);
  p.ncls = 1;
  p.prec = prec_of(g);
  p.in = make_grid(in);
  p.out = make_grid(out);
  p.res = null_grid();
  p.res_ups = 0;
  if (residual && residual->ptr) {
    if ((rc = check_tensor(residual))) return rc;
    if (!same_shape(residual, out)) {
      // a residual of half the patch extent is read through a nearest x2 upsample (the generator's shortcut branch,
      // reference models/layers.py:294-299 behind generators.py:52: nothing is materialised at the output's resolution)
      if (residual->n != out->n || residual->gh != out->gh || residual->gw != out->gw || residual->c != out->c ||
          residual->ld != out->ld || 2 * residual->ph != out->ph || 2 * residual->pw != out->pw)
        return ITG_ERR_ARG;
      p.res_ups = 1;
    }
    p.res = make_grid(residual);
  }
  if (in->n != out->n) return ITG_ERR_ARG;
  int Ho = conv_out_dim(p.in.H, g->kh, g->stride, pad_v(g)), Wo = conv_out_dim(p.in.W, g->kw, g->stride, g->pad);
  if (Ho != p.out.H || Wo != p.out.W) return ITG_ERR_ARG;
  if (g->pad_mode == ITG_PAD_REPLICATE && g->stride != 1) return ITG_ERR_ARG;
  p.w = w_packed; p.bias = bias; p.scale = out_scale; p.res_mode = 0; p.res_slope = 0.f;
  p.stats = g->out_stats;
  if (p.stats && out->ld > 512) return ITG_ERR_ARG;
  p.ntaps = g->kh * g->kw; p.kw = g->kw; p.cin_ld = in->ld;
  p.Kpad = round_up(p.ntaps * in->ld, BK);
  p.MT = Ho; p.MU = Wo;
  int64_t M = (int64_t)in->n * Ho * Wo;
  if (M >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
  p.M = (int)M;
  p.isy = p.isx = g->stride; p.ioy = -pad_v(g); p.iox = -g->pad;
  p.osy = p.osx = 1; p.ooy = p.oox = 0;
  p.pad_mode = g->pad_mode; p.out_mode = 0; p.act = act; p.slope = slope;
  p.co_rows = round_up(out->c, 16);
  return dispatch_nt(p, workspace, workspace_floats, (hipStream_t)stream);
}

int itg_conv2d_dgrad(const itg_tensor* dy, const float* w_packed_dgrad, const float* out_scale, const itg_tensor* dx,
                     const itg_tensor* act_out, int act, float slope, const itg_conv_geom* g, float* workspace,
                     int64_t workspace_floats, void* stream) {
  int rc;
  if ((rc = check_tensor(dy)) || (rc = check_tensor(dx))) return rc;
  if (!w_packed_dgrad || !g) return ITG_ERR_ARG;
  if (dy->n != dx->n) return ITG_ERR_ARG;
  if (g->bn_bwd) g->bn_bwd->taken = 0;       // the kernel families below that do not take the sums leave it at 0
  hipStream_t s = (hipStream_t)stream;
  if (wino_s2_geom(g)) {
    // the 4 x 4 stride-2 layer's input gradient as the ADJOINT of its F(4 x 4, 2 x 2) forward (conv_wino.hip): panel from
    // itg_pack_wino_s2_dgrad
    const itg_tensor* r = (act_out && act_out->ptr && act != ITG_ACT_NONE) ? act_out : nullptr;
    if (r && ((rc = check_tensor(r)) || !same_shape(r, dx))) return rc ? rc : ITG_ERR_ARG;
    return wino_conv_s2_dgrad(dy, w_packed_dgrad, out_scale, dx, r, r ? act : 0, slope, prec_of(g), workspace, workspace_floats, s);
  }
  if (g->flags & ITG_GEOM_WINO) {
    // the input gradient of an R x R stride-1 pad-1 conv is the pad-(R - 2) correlation of dy with the flipped, transposed
    // filter; replicate padding: evaluated on the padded extent (pad R - 1), the frame folded onto dx's zeroed border
    if (!wino_geom(g)) return ITG_ERR_ARG;
    const itg_tensor* r = (act_out && act_out->ptr && act != ITG_ACT_NONE) ? act_out : nullptr;
    if (r && ((rc = check_tensor(r)) || !same_shape(r, dx))) return rc ? rc : ITG_ERR_ARG;
    const int fold = wino_fold(g);
    if (fold && !(g->flags & ITG_GEOM_FRAME_ZEROED) && (rc = launch_zero_border(make_grid(dx), s))) return rc;
    return wino_conv(dy, w_packed_dgrad, nullptr, out_scale, r, 0, r ? act : 0, slope, dx, g->kh, g->kh - 2 + fold, ITG_PAD_ZERO, fold,
                     ITG_ACT_NONE, 0.f, prec_of(g), workspace, workspace_floats, s, true);
  }
  if (g->up2) {
    // dx(z) = sum_t W4(t) dy(2 z + t - 1): the 4 x 4 stride-2 conv of dy with the phase-summed taps (itg_pack_up2_dgrad).
    // Replicate padding: the frame's source pixels are x(-1) := x(0), ..., so the domain is extended by one source pixel
    // on every side and the gradients of the frame fold onto the edge pixels (atomics on the zeroed 1-pixel border).
    if ((rc = up2_check(g, dx, dy))) return rc;
    ConvP p;
    clear_xf(p);
    p.stats = nullptr; p.ncls = 1; p.prec = prec_of(g);
    p.in = make_grid(dy); p.out = make_grid(dx);
    p.res = null_grid(); p.res_mode = 0; p.res_slope = 0.f; p.res_ups = 0;
    if (act_out && act_out->ptr && act != ITG_ACT_NONE) {
      if ((rc = check_tensor(act_out))) return rc;
      if (!same_shape(act_out, dx)) return ITG_ERR_ARG;
      p.res = make_grid(act_out); p.res_mode = act; p.res_slope = slope;
    }
    p.w = w_packed_dgrad; p.bias = nullptr; p.scale = out_scale; p.act = ITG_ACT_NONE; p.slope = 0.f;
    p.cin_ld = dy->ld; p.co_rows = round_up(dx->c, 16);
    p.ntaps = 16; p.kw = 4; p.Kpad = round_up(16 * dy->ld, BK);
    p.isy = p.isx = 2; p.osy = p.osx = 1;
    p.pad_mode = ITG_PAD_ZERO;
    if (g->pad_mode == ITG_PAD_REPLICATE) {
      p.MT = p.out.H + 2; p.MU = p.out.W + 2;
      p.ioy = p.iox = -3; p.ooy = p.oox = -1; p.out_mode = 1;
      if (!(g->flags & ITG_GEOM_FRAME_ZEROED) && (rc = launch_zero_border(p.out, s))) return rc;
    } else {
      p.MT = p.out.H; p.MU = p.out.W;
      p.ioy = p.iox = -1; p.ooy = p.oox = 0; p.out_mode = 0;
    }
    if (pad_v(g) == 0) {       // explicit halo rows: dx row e (halo rows included) gathers dy rows 2 (e - 1) + t - 1, no vertical fold
      p.MT = p.out.H; p.ioy = -3; p.ooy = 0;
    }
    const int64_t M = (int64_t)dx->n * p.MT * p.MU;
    if (M >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
    p.M = (int)M;
    if (!p.res.p && (rc = set_bnb(p, g, dx))) return rc;
    return dispatch_nt(p, workspace, workspace_floats, s, g->bn_bwd ? &g->bn_bwd->taken : nullptr);
  }
  if (logit_dgrad_ok(dy, dx, g)) {
    const int Ho = dy->ph, Wo = dy->pw, H = dx->gh * dx->ph, W = dx->gw * dx->pw;
    if (conv_out_dim(H, 4, 1, 1) != Ho || conv_out_dim(W, 4, 1, 1) != Wo) return ITG_ERR_ARG;
    GridT ao = null_grid();
    if (act_out && act_out->ptr && act != ITG_ACT_NONE) {
      if ((rc = check_tensor(act_out))) return rc;
      if (!same_shape(act_out, dx)) return ITG_ERR_ARG;
      ao = make_grid(act_out);
    }
    const int CH = dx->ld >= 128 ? 128 : dx->ld;
    const int tiles_x = (W + LG_TW - 1) / LG_TW, tiles_y = (H + LG_TH - 1) / LG_TH;
    const int64_t nb = (int64_t)dx->n * tiles_x * tiles_y;
    if (nb <= 0 || nb >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
    const size_t lds = (size_t)(CH * 20 + LG_WH * LG_WW) * sizeof(float);
    hipLaunchKernelGGL(logit_dgrad_kernel, dim3((unsigned)nb, (unsigned)((dx->ld + CH - 1) / CH)), dim3(256), lds, s, make_grid(dy),
                       w_packed_dgrad, round_up(16 * dy->ld, BK), out_scale, make_grid(dx), ao, act, slope, CH, tiles_x, tiles_y);
    ITG_CHECK_LAUNCH();
    snprintf(g_last_launch, sizeof(g_last_launch), "logit_dgrad_kernel");
    return ITG_OK;
  }
  if (thin_in_conv(dy, dx, g)) {
    const int Ho = dy->gh * dy->ph, Wo = dy->gw * dy->pw, H = dx->gh * dx->ph, W = dx->gw * dx->pw;
    if (conv_out_dim(H, 4, 2, 1) != Ho || conv_out_dim(W, 4, 2, 1) != Wo) return ITG_ERR_ARG;
    const int rows = 64;     // 4 parity classes x 4 taps x 4 (padded) input channels
    if (dx->c <= 3 && (dy->ld == 16 || dy->ld == 32 || dy->ld == 64 || dy->ld == 128) && grid_pixels(dy) * dy->ld * 4 < 0xFFFF0000LL) {
      // one launch: Q stays in LDS (thin_dgrad_fused_kernel)
      GridT ao = null_grid();
      if (act_out && act_out->ptr && act != ITG_ACT_NONE) {
        if ((rc = check_tensor(act_out))) return rc;
        if (!same_shape(act_out, dx)) return ITG_ERR_ARG;
        ao = make_grid(act_out);
      }
      const int tiles_x = ((W + 1) / 2 + TD_TW - 1) / TD_TW, tiles_y = ((H + 1) / 2 + TD_TH - 1) / TD_TH;
      const int64_t nt = (int64_t)dx->n * tiles_x * tiles_y;
      if (nt <= 0 || nt >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
      const size_t lds = (size_t)(TD_ROWS * (dy->ld + 4) + TD_PIX * TD_QP) * sizeof(float);
      const int64_t rounds = (nt + 767) / 768;                   // three workgroups fit a CU (50 KB of LDS each at 64 channels): an even deal over 768
      const int blocks = (int)((nt + rounds - 1) / rounds);
      const unsigned dyb = (unsigned)(grid_pixels(dy) * dy->ld * 4);
      const GridT gdy = make_grid(dy), gdx = make_grid(dx);
      const int cpad = round_up(dx->c, 16);
#define ITG_TDF(NKB)                                                                                                                     \
  do {                                                                                                                                   \
    if (lds > 64 * 1024) {                                                                                                               \
      static bool attr_done = false;                                                                                                     \
      if (!attr_done) {                                                                                                                  \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_dgrad_fused_kernel<NKB>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                96 * 1024) != hipSuccess) { (void)hipGetLastError(); return ITG_ERR_LAUNCH; }                              \
        attr_done = true;                                                                                                                \
      }                                                                                                                                  \
    }                                                                                                                                    \
    hipLaunchKernelGGL(thin_dgrad_fused_kernel<NKB>, dim3(blocks), dim3(256), lds, s, gdy, w_packed_dgrad, out_scale, gdx, ao, act, slope, \
                       dx->c, cpad, tiles_x, tiles_y, (int)nt, dyb);                                                                     \
  } while (0)
      if (dy->ld == 16) ITG_TDF(1); else if (dy->ld == 32) ITG_TDF(2); else if (dy->ld == 64) ITG_TDF(4); else ITG_TDF(8);
#undef ITG_TDF
      ITG_CHECK_LAUNCH();
      snprintf(g_last_launch, sizeof(g_last_launch), "thin_dgrad_fused_kernel");
      return ITG_OK;
    }
    const int64_t Mo = grid_pixels(dy), qf = Mo * rows, pf = (int64_t)rows * dy->ld;
    if (!workspace || workspace_floats < qf + pf || qf >= ((int64_t)1 << 31)) return ITG_ERR_WORKSPACE;
    float* Q = workspace;
    float* panel = workspace + qf;
    {
      int tot = rows * dy->ld;
      hipLaunchKernelGGL(thin_dgrad_panel_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, w_packed_dgrad, panel, dx->c,
                         round_up(dx->c, 16), dy->ld, rows);
      ITG_CHECK_LAUNCH();
    }
    itg_tensor Qt = {Q, dy->n, 1, 1, Ho, Wo, rows, rows};
    itg_conv_geom g1 = {1, 1, 1, 0, ITG_PAD_ZERO, 0, prec_of(g)};
    if ((rc = itg_conv2d_fwd(dy, panel, nullptr, out_scale, nullptr, &Qt, &g1, ITG_ACT_NONE, 0.f, workspace + qf + pf,
                             workspace_floats - qf - pf, stream))) return rc;
    GridT ao = null_grid();
    if (act_out && act_out->ptr && act != ITG_ACT_NONE) {
      if ((rc = check_tensor(act_out))) return rc;
      if (!same_shape(act_out, dx)) return ITG_ERR_ARG;
      ao = make_grid(act_out);
    }
    GridT gx = make_grid(dx);
    int64_t total = (int64_t)gx.n * gx.H * gx.W;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(thin_dgrad_gather_kernel, dim3(blocks), dim3(256), 0, s, (const float*)Q, Ho, Wo, rows, gx, ao, act,
                       slope);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  ConvP p;
  clear_xf(p);
  p.stats = nullptr;
  p.ncls = 1;
  p.prec = prec_of(g);
  p.in = make_grid(dy);
  p.out = make_grid(dx);
  p.res = null_grid();
  p.res_mode = 0; p.res_slope = 0.f; p.res_ups = 0;
  if (act_out && act_out->ptr && act != ITG_ACT_NONE) {
    // dx is the gradient w.r.t. act(.)'s output `act_out`: hand back the gradient w.r.t. its input instead
    if ((rc = check_tensor(act_out))) return rc;
    if (!same_shape(act_out, dx)) return ITG_ERR_ARG;
    p.res = make_grid(act_out); p.res_mode = act; p.res_slope = slope;
  }
  const int padh = pad_v(g);
  int Ho = conv_out_dim(p.out.H, g->kh, g->stride, padh), Wo = conv_out_dim(p.out.W, g->kw, g->stride, g->pad);
  if (Ho != p.in.H || Wo != p.in.W) return ITG_ERR_ARG;
  p.bias = nullptr; p.scale = out_scale; p.act = ITG_ACT_NONE; p.slope = 0.f;
  p.cin_ld = dy->ld;
  p.co_rows = round_up(dx->c, 16);
  p.pad_mode = ITG_PAD_ZERO;
  if (g->stride == 1) {
    p.w = w_packed_dgrad;
    p.ntaps = g->kh * g->kw; p.kw = g->kw;
    p.Kpad = round_up(p.ntaps * dy->ld, BK);
    p.isy = p.isx = 1; p.osy = p.osx = 1;
    if (g->pad_mode == ITG_PAD_REPLICATE && (g->pad > 0 || padh > 0)) {
      // padded domain, gradients of the replicated frame fold onto the edge pixels
      p.MT = p.out.H + 2 * padh; p.MU = p.out.W + 2 * g->pad;
      p.ioy = -(g->kh - 1); p.iox = -(g->kw - 1);
      p.ooy = -padh; p.oox = -g->pad;
      p.out_mode = 1;
      if (!(g->flags & ITG_GEOM_FRAME_ZEROED) && (rc = launch_zero_border(p.out, s))) return rc;
    } else {
      p.MT = p.out.H; p.MU = p.out.W;
      p.ioy = -(g->kh - 1 - padh); p.iox = -(g->kw - 1 - g->pad);
      p.ooy = p.oox = 0;
      p.out_mode = 0;
    }
    int64_t M = (int64_t)dx->n * p.MT * p.MU;
    if (M >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
    p.M = (int)M;
    if (!p.res.p && (rc = set_bnb(p, g, dx))) return rc;
    return dispatch_nt(p, workspace, workspace_floats, s, g->bn_bwd ? &g->bn_bwd->taken : nullptr);
  }
  if (g->stride != 2 || g->pad != 1 || padh != 1 || (g->kh & 1) || (g->kw & 1) || g->pad_mode != ITG_PAD_ZERO)
    return ITG_ERR_ARG;
  int skh = g->kh / 2, skw = g->kw / 2;
  p.ntaps = skh * skw; p.kw = skw;
  p.Kpad = round_up(p.ntaps * dy->ld, BK);
  p.isy = p.isx = 1; p.osy = p.osx = 2; p.out_mode = 0;
  int ci_pad = round_up(dx->c, 16);
  p.ncls = 4;
  int Mmax = 0;
  for (int ry = 0; ry < 2; ++ry)
    for (int rx = 0; rx < 2; ++rx) {
      const int c = ry * 2 + rx;
      int ay = (ry + g->pad) & 1, ax = (rx + g->pad) & 1;
      int by = (ry + g->pad - ay) / 2, bx = (rx + g->pad - ax) / 2;
      p.cMT[c] = (p.out.H - ry + 1) / 2; p.cMU[c] = (p.out.W - rx + 1) / 2;
      int64_t M = (int64_t)dx->n * p.cMT[c] * p.cMU[c];
      if (M >= ((int64_t)1 << 29)) return ITG_ERR_ARG;
      p.cM[c] = (int)M;
      if (p.cM[c] > Mmax) Mmax = p.cM[c];
      p.cioy[c] = by - (skh - 1); p.ciox[c] = bx - (skw - 1);
      p.cooy[c] = ry; p.coox[c] = rx;
      p.cwoff[c] = (unsigned)((size_t)c * ci_pad * p.Kpad);
    }
  if (Mmax <= 0) return ITG_OK;
  p.w = w_packed_dgrad;
  p.M = Mmax; p.MT = p.cMT[0]; p.MU = p.cMU[0];
  p.ioy = p.cioy[0]; p.iox = p.ciox[0]; p.ooy = 0; p.oox = 0;
  return dispatch_nt(p, workspace, workspace_floats, s);
}

// the weight gradient of an ITG_GEOM_WINO layer also goes through the transformed domain (conv_wino.hip) when both tensors
// are plain fp32 images with 16-aligned pitches; ITG_WINOGRAD_WGRAD=0 keeps the direct contraction (A/B switch)
// (the stride-2 form - wino_s2_geom - takes R = 2 in the plan: its four parity classes are 2 x 2 convolutions)
static inline int wino_wg_R(const itg_conv_geom* g) { return wino_s2_geom(g) ? 2 : g->kh; }
static bool wino_wgrad_ok(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g) {
  static const int on = env_int("ITG_WINOGRAD_WGRAD", 1);
  return on && (wino_geom(g) || wino_s2_geom(g)) && prec_of(g) == ITG_PREC_F32 && !(x->ld & 15) && !(dy->ld & 15) && x->n == dy->n &&
         plan_wino_wgrad(x, dy, wino_wg_R(g)).tn.ngroups == 0;
}

int64_t itg_conv2d_wgrad_workspace(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g) {
  if (!x || !dy || !g) return 0;
  if (wino_wgrad_ok(x, dy, g)) return plan_wino_wgrad(x, dy, wino_wg_R(g)).ws_floats;
  if (thin_out_conv(x, dy, g)) {
    int64_t Min = grid_pixels(x);
    TnPlan t = plan_tn(Min, 16, x->ld, prec_of(g));
    return Min * 16 + 16 * (int64_t)x->c + 16 + t.ws_floats + (int64_t)t.splits * t.co_rows;
  }
  if (g->up2) {
    const TileWgPlan tw = plan_wgrad_up2_tile(x, dy, g);
    TnPlan t = tw.ok ? tn_plan_for_up2_tiles(tw, dy->ld, x->ld) : plan_tn(grid_pixels(dy) / 4, dy->ld, 4 * x->ld, prec_of(g), 4);
    return t.ws_floats + (int64_t)4 * t.splits * t.co_rows;
  }
  int64_t M = grid_pixels(dy);
  TileWgPlan tw = plan_wgrad_tile(x, dy, g);
  TnPlan t = tw.ok ? tn_plan_for_tiles(tw, dy->ld, g->kh * g->kw * x->ld) : plan_tn(M, dy->ld, g->kh * g->kw * x->ld, prec_of(g));
  return t.ws_floats + (int64_t)t.splits * t.co_rows;
}

// shared by itg_conv2d_wgrad / itg_conv2d_wgrad_slabs: geometry checks, plan, WgP
static int wgrad_setup(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g, bool want_db, float* workspace,
                       int64_t workspace_floats, WgP& p, TnPlan& t_out, TileWgPlan& tw_out, int& prec_out) {
  int rc;
  p.x = make_grid(x);
  p.dy = make_grid(dy);
  p.up2 = 0;
  p.ucls = 0; p.u_x = 0; p.u_dy = 0;
  if (g->up2) {
    // conv3x3(up2x(x)): four output-parity classes of 2 x 2 taps over the SOURCE pixel domain (WgP.up2); the replicate
    // clamp of the gathered operand in source coordinates is the upsampled image's replicate padding
    if ((rc = up2_check(g, x, dy))) return rc;
    p.up2 = 1;
    const int64_t M = grid_pixels(dy) / 4;              // source pixels that produce output (pad_h = 0: x carries 2 halo rows more)
    if (M >= ((int64_t)1 << 29)) return ITG_ERR_ARG;
    p.ntaps = 4; p.kw = 2; p.cin_ld = x->ld; p.Ktot = 4 * x->ld;
    const int prec = prec_of(g);
    const TileWgPlan tw = plan_wgrad_up2_tile(x, dy, g);        // narrow layers on large images: folded halo-tile kernel
    TnPlan t = tw.ok ? tn_plan_for_up2_tiles(tw, dy->ld, x->ld) : plan_tn(M, dy->ld, p.Ktot, prec, 4);
    if (t.ws_floats + (int64_t)4 * t.splits * t.co_rows > workspace_floats) return ITG_ERR_WORKSPACE;
    p.Kpad = t.Kpad; p.co_rows = t.co_rows;
    p.slab = workspace;
    p.dbslab = want_db ? workspace + t.ws_floats : nullptr;       // [splits][4][co_rows]
    p.MT = p.dy.H / 2; p.MU = p.x.W; p.M = (int)M;
    p.stride = 1; p.pad = 1; p.pad_h = pad_v(g); p.pad_mode = g->pad_mode;
    p.chunks_per_split = t.chunks_per_split; p.nchunks = t.nchunks;
    int64_t xb = grid_pixels(x) * x->ld * 4, yb = grid_pixels(dy) * dy->ld * 4;
    if (xb >= 0xFFFF0000LL || yb >= 0xFFFF0000LL) return ITG_ERR_ARG;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
    tw_out = tw;
    t_out = t; prec_out = prec;
    return ITG_OK;
  }
  if (g->bn_bwd) return ITG_ERR_ARG;
  int Ho = conv_out_dim(p.x.H, g->kh, g->stride, pad_v(g)), Wo = conv_out_dim(p.x.W, g->kw, g->stride, g->pad);
  if (Ho != p.dy.H || Wo != p.dy.W) return ITG_ERR_ARG;
  if (g->pad_mode == ITG_PAD_REPLICATE && g->stride != 1) return ITG_ERR_ARG;
  int64_t M = (int64_t)x->n * Ho * Wo;
  if (M >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
  p.ntaps = g->kh * g->kw; p.kw = g->kw; p.cin_ld = x->ld;
  p.Ktot = p.ntaps * x->ld;
  const int prec = prec_of(g);
  const TileWgPlan tw = plan_wgrad_tile(x, dy, g);
  TnPlan t = tw.ok ? tn_plan_for_tiles(tw, dy->ld, p.Ktot) : plan_tn(M, dy->ld, p.Ktot, prec);
  if (t.ws_floats + (int64_t)t.splits * t.co_rows > workspace_floats) return ITG_ERR_WORKSPACE;
  p.Kpad = t.Kpad; p.co_rows = t.co_rows;
  p.slab = workspace;
  p.dbslab = want_db ? workspace + t.ws_floats : nullptr;       // [splits][co_rows] after the slabs
  p.MT = Ho; p.MU = Wo; p.M = (int)M;
  p.stride = g->stride; p.pad = g->pad; p.pad_h = pad_v(g); p.pad_mode = g->pad_mode;
  p.chunks_per_split = t.chunks_per_split; p.nchunks = t.nchunks;
  {
    int64_t xb = grid_pixels(x) * x->ld * 4, yb = grid_pixels(dy) * dy->ld * 4;
    if (xb >= 0xFFFF0000LL || yb >= 0xFFFF0000LL) return ITG_ERR_ARG;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
  }
  static const int plan_debug = env_int("ITG_DEBUG", 0) & 1;
  if (plan_debug)
    fprintf(stderr, "[tn] M=%lld co_rows=%d Kpad=%d -> bcol=%d bco=%d splits=%d ngroups=%d tile=%d\n", (long long)M, t.co_rows,
            t.Kpad, t.bcol, t.bco, t.splits, t.ngroups, tw.ok);
  t_out = t; tw_out = tw; prec_out = prec;
  return ITG_OK;
}

int itg_conv2d_wgrad(const itg_tensor* x, const itg_tensor* dy, float* dw, float* db, const itg_conv_geom* g,
                     int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(dy))) return rc;
  if (!dw || !g || !workspace) return ITG_ERR_ARG;
  if (x->n != dy->n) return ITG_ERR_ARG;
  if (accumulate & ~(ITG_ACC_DW | ITG_ACC_DB)) return ITG_ERR_ARG;     // a bit set, not a boolean (ABI note)
  hipStream_t s = (hipStream_t)stream;
  if (thin_out_conv(x, dy, g)) {
    const int H = x->gh * x->ph, W = x->gw * x->pw;
    if (conv_out_dim(H, g->kh, 1, pad_v(g)) != dy->gh * dy->ph || conv_out_dim(W, g->kw, 1, g->pad) != dy->gw * dy->pw)
      return ITG_ERR_ARG;
    const int64_t Min = grid_pixels(x), qf = Min * 16, tf = 16 * (int64_t)x->c + 16;
    if (workspace_floats < qf + tf || qf >= ((int64_t)1 << 31)) return ITG_ERR_WORKSPACE;
    float* Q = workspace;
    float* tmp = workspace + qf;            // [16][ci] then [16] bias partial
    int blocks = (int)((qf + 255) / 256 < 8192 ? (qf + 255) / 256 : 8192);
    hipLaunchKernelGGL(tap_scatter_dy_kernel, dim3(blocks), dim3(256), 0, s, make_grid(dy), Q, H, W, g->kh, g->kw, pad_v(g),
                       g->pad);
    ITG_CHECK_LAUNCH();
    itg_tensor Qt = {Q, x->n, 1, 1, H, W, 16, 16};
    itg_conv_geom g1 = {1, 1, 1, 0, ITG_PAD_ZERO, 0, prec_of(g)};
    if ((rc = itg_conv2d_wgrad(x, &Qt, tmp, tmp + 16 * (int64_t)x->c, &g1, 0, workspace + qf + tf,
                               workspace_floats - qf - tf, stream))) return rc;
    int n = x->c * 16;
    hipLaunchKernelGGL(tap_wgrad_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)tmp,
                       (const float*)(tmp + 16 * (int64_t)x->c), dw, db, x->c, 16, pad_v(g) * g->kw + g->pad, accumulate);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  if (wino_wgrad_ok(x, dy, g)) {
    const WinoWgPlan w = plan_wino_wgrad(x, dy, wino_wg_R(g));
    if (workspace_floats < w.ws_floats) return ITG_ERR_WORKSPACE;
    if ((rc = wino_wgrad_slabs(x, dy, g->pad, g->pad_mode, prec_of(g), w, workspace, db != nullptr, s, g->wino_v))) return rc;
    return launch_wgrad_reduce(workspace + w.slab_off, 1, workspace + w.db_off, w.Rr, dw, db, dy->c, x->c, x->ld, g->kh, g->kw, w.co_rows,
                               w.Kpad, accumulate, s);
  }
  WgP p;
  TnPlan t;
  TileWgPlan tw;
  int prec;
  if ((rc = wgrad_setup(x, dy, g, db != nullptr, workspace, workspace_floats, p, t, tw, prec))) return rc;
  return run_wgrad(p, t, tw, prec, x, dy, g, dw, db, accumulate, workspace, s);
}


int itg_conv2d_wgrad_slabs(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g, float* workspace,
                           int64_t workspace_floats, itg_wgrad_job* job, void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(dy))) return rc;
  if (!g || !workspace || !job) return ITG_ERR_ARG;
  if (x->n != dy->n || g->kh * g->kw > 49) return ITG_ERR_ARG;
  if (thin_out_conv(x, dy, g)) return ITG_ERR_ARG;      // taps-as-rows path: not deferrable
  if (g->up2) return ITG_ERR_ARG;                                       // folded-upsample layers reduce through their own kernel
  if (wino_wgrad_ok(x, dy, g)) {
    const WinoWgPlan w = plan_wino_wgrad(x, dy, wino_wg_R(g));
    if (workspace_floats < w.ws_floats) return ITG_ERR_WORKSPACE;
    if ((rc = wino_wgrad_slabs(x, dy, g->pad, g->pad_mode, prec_of(g), w, workspace, true, (hipStream_t)stream, g->wino_v))) return rc;
    job->slab = workspace + w.slab_off; job->dbslab = workspace + w.db_off;
    job->splits = 1; job->dbsplits = w.Rr;
    job->co = dy->c; job->ci = x->c; job->ci_ld = x->ld; job->kh = g->kh; job->kw = g->kw;
    job->co_rows = w.co_rows; job->Kpad = w.Kpad;
    job->ngroups = 0; job->group = red_group(); job->stage = nullptr;
    return ITG_OK;
  }
  WgP p;
  TnPlan t;
  TileWgPlan tw;
  int prec;
  if ((rc = wgrad_setup(x, dy, g, true, workspace, workspace_floats, p, t, tw, prec))) return rc;
  if ((rc = run_wgrad_slabs(p, t, tw, prec, (hipStream_t)stream))) return rc;
  job->slab = workspace; job->dbslab = p.dbslab;
  job->splits = t.splits; job->dbsplits = t.splits;
  job->co = dy->c; job->ci = x->c; job->ci_ld = x->ld; job->kh = g->kh; job->kw = g->kw;
  job->co_rows = t.co_rows; job->Kpad = t.Kpad;
  job->ngroups = t.ngroups; job->group = red_group();
  job->stage = t.ngroups > 0 ? workspace + t.slab_floats : nullptr;
  return ITG_OK;
}

int itg_zero_frames(const itg_tensor* tensors, int n, void* stream) { return launch_zero_frames(tensors, n, (hipStream_t)stream); }

int itg_wgrad_reduce_multi(const itg_wgrad_job* jobs, int n, void* stream) {
  return launch_reduce_multi(jobs, n, (hipStream_t)stream);
}

}  // extern "C"
