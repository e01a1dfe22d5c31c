// Memory-bound pointwise / data-movement kernels (gfx950): activations, nearest x2 upsample,
// 2x2 max-pool, NCHW <-> patch-grid NHWC conversion, and the LocalPadder operator itself
// (reference models/layers.py:38-173, utils.py:577-613,658-742) in its standalone forms.
#include "itg_common.h"

namespace {

inline int blocks_for(int64_t total, int per_block = 256, int cap = 16384) {
  int64_t b = (total + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

#define GRID_STRIDE(i, total) \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

__global__ void act_fwd_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int64_t n4, int act, float slope) {
  GRID_STRIDE(i, n4) {
    f32x4 v = x[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act, slope);
    y[i] = v;
  }
}

// derivative from the forward OUTPUT: lrelu: out>0 ? 1 : slope ; tanh: 1-out^2
__global__ void act_bwd_kernel(const f32x4* __restrict__ out, const f32x4* __restrict__ dout, f32x4* __restrict__ dx,
                               int64_t n4, int act, float slope) {
  GRID_STRIDE(i, n4) {
    f32x4 o = out[i], g = dout[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (act == ITG_ACT_LRELU) g[e] = o[e] > 0.f ? g[e] : g[e] * slope;
      else if (act == ITG_ACT_TANH) g[e] = g[e] * (1.f - o[e] * o[e]);
    }
    dx[i] = g;
  }
}

__global__ void add_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ o, int64_t n4) {
  GRID_STRIDE(i, n4) o[i] = a[i] + b[i];
}

__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t npix, int q4, int ph,
                                    int pw) {
  int64_t total = npix * q4;
  const int ld = q4 * 4;
  GRID_STRIDE(i, total) {
    int c4 = (int)(i % q4);
    int64_t pix = i / q4;
    int64_t blk = pix / (ph * pw);
    int r = (int)(pix - blk * ph * pw);
    int yy = r / pw, xx = r - yy * pw;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ld + c4 * 4);
    float* o = y + ((blk * 2 * ph + 2 * yy) * (2 * pw) + 2 * xx) * (int64_t)ld + c4 * 4;
    *reinterpret_cast<f32x4*>(o) = v;
    *reinterpret_cast<f32x4*>(o + ld) = v;
    *reinterpret_cast<f32x4*>(o + (int64_t)2 * pw * ld) = v;
    *reinterpret_cast<f32x4*>(o + (int64_t)2 * pw * ld + ld) = v;
  }
}

__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t npix, int q4, int ph,
                                    int pw) {
  int64_t total = npix * q4;
  const int ld = q4 * 4;
  GRID_STRIDE(i, total) {
    int c4 = (int)(i % q4);
    int64_t pix = i / q4;
    int64_t blk = pix / (ph * pw);
    int r = (int)(pix - blk * ph * pw);
    int yy = r / pw, xx = r - yy * pw;
    const float* o = dy + ((blk * 2 * ph + 2 * yy) * (2 * pw) + 2 * xx) * (int64_t)ld + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(o);
    v += *reinterpret_cast<const f32x4*>(o + ld);
    v += *reinterpret_cast<const f32x4*>(o + (int64_t)2 * pw * ld);
    v += *reinterpret_cast<const f32x4*>(o + (int64_t)2 * pw * ld + ld);
    *reinterpret_cast<f32x4*>(dx + pix * ld + c4 * 4) = v;
  }
}

// 2x2 max pool inside each patch (y.ph = x.ph/2)
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t npix_out, int q4, int oph,
                                   int opw) {
  int64_t total = npix_out * q4;
  const int ld = q4 * 4;
  GRID_STRIDE(i, total) {
    int c4 = (int)(i % q4);
    int64_t pix = i / q4;
    int64_t blk = pix / (oph * opw);
    int r = (int)(pix - blk * oph * opw);
    int yy = r / opw, xx = r - yy * opw;
    const float* s = x + ((blk * 2 * oph + 2 * yy) * (2 * opw) + 2 * xx) * (int64_t)ld + c4 * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(s);
    f32x4 b = *reinterpret_cast<const f32x4*>(s + ld);
    f32x4 c = *reinterpret_cast<const f32x4*>(s + (int64_t)2 * opw * ld);
    f32x4 d = *reinterpret_cast<const f32x4*>(s + (int64_t)2 * opw * ld + ld);
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
    *reinterpret_cast<f32x4*>(y + pix * ld + c4 * 4) = m;
  }
}

// routes dy to the first element (row-major window order) equal to the max
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                   int64_t npix_out, int q4, int oph, int opw) {
  int64_t total = npix_out * q4;
  const int ld = q4 * 4;
  GRID_STRIDE(i, total) {
    int c4 = (int)(i % q4);
    int64_t pix = i / q4;
    int64_t blk = pix / (oph * opw);
    int r = (int)(pix - blk * oph * opw);
    int yy = r / opw, xx = r - yy * opw;
    int64_t base = ((blk * 2 * oph + 2 * yy) * (2 * opw) + 2 * xx) * (int64_t)ld + c4 * 4;
    int64_t o1 = ld, o2 = (int64_t)2 * opw * ld, o3 = o2 + ld;
    f32x4 a = *reinterpret_cast<const f32x4*>(x + base);
    f32x4 b = *reinterpret_cast<const f32x4*>(x + base + o1);
    f32x4 c = *reinterpret_cast<const f32x4*>(x + base + o2);
    f32x4 d = *reinterpret_cast<const f32x4*>(x + base + o3);
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + pix * ld + c4 * 4);
    f32x4 ga, gb, gc, gd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int k = 0; float m = a[e];
      if (b[e] > m) { m = b[e]; k = 1; }
      if (c[e] > m) { m = c[e]; k = 2; }
      if (d[e] > m) { m = d[e]; k = 3; }
      ga[e] = k == 0 ? g[e] : 0.f; gb[e] = k == 1 ? g[e] : 0.f;
      gc[e] = k == 2 ? g[e] : 0.f; gd[e] = k == 3 ? g[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dx + base) = ga;
    *reinterpret_cast<f32x4*>(dx + base + o1) = gb;
    *reinterpret_cast<f32x4*>(dx + base + o2) = gc;
    *reinterpret_cast<f32x4*>(dx + base + o3) = gd;
  }
}

// ---- NCHW <-> patch-grid NHWC -----------------------------------------------------------
// dst-ordered: element (pix, ch) of the grid tensor
__global__ void nchw_to_grid_kernel(const float* __restrict__ src, GridT d, int merged) {
  int64_t total = (int64_t)d.n * d.gh * d.gw * d.ph * d.pw * d.ld;
  GRID_STRIDE(i, total) {
    int ch = (int)(i % d.ld);
    int64_t pix = i / d.ld;
    int x = (int)(pix % d.pw); pix /= d.pw;
    int y = (int)(pix % d.ph); pix /= d.ph;
    int gc = (int)(pix % d.gw); pix /= d.gw;
    int gr = (int)(pix % d.gh);
    int n = (int)(pix / d.gh);
    float v = 0.f;
    if (ch < d.c) {
      int64_t s;
      if (merged) s = (((int64_t)n * d.c + ch) * d.H + gr * d.ph + y) * d.W + gc * d.pw + x;
      else s = (((((int64_t)n * d.gh + gr) * d.gw + gc) * d.c + ch) * d.ph + y) * d.pw + x;
      v = src[s];
    }
    d.p[i] = v;
  }
}

// dst-ordered over the NCHW tensor
__global__ void grid_to_nchw_kernel(GridT s, float* __restrict__ dst, int merged) {
  int64_t total = (int64_t)s.n * s.c * s.H * s.W;
  GRID_STRIDE(i, total) {
    int64_t r = i;
    int n, ch, Y, X;
    if (merged) {
      X = (int)(r % s.W); r /= s.W;
      Y = (int)(r % s.H); r /= s.H;
      ch = (int)(r % s.c);
      n = (int)(r / s.c);
    } else {
      int x = (int)(r % s.pw); r /= s.pw;
      int y = (int)(r % s.ph); r /= s.ph;
      ch = (int)(r % s.c); r /= s.c;
      int gc = (int)(r % s.gw); r /= s.gw;
      int gr = (int)(r % s.gh);
      n = (int)(r / s.gh);
      Y = gr * s.ph + y; X = gc * s.pw + x;
    }
    dst[i] = s.p[grid_off(s, n, Y, X) + ch];
  }
}

// ---- LocalPadder, NCHW patches in / out (the reference module's own tensor format) -------
// x: (n*gh*gw, c, p, p) or merged (n, c, gh*p+2, gw*p+2); y: (n*gh*gw, c, p+2, p+2)
// Small patches (p + 2 < 48): one whole (patch, channel) plane per wave iteration, lanes sweep the
// flattened (p+2)^2 outputs with float-reciprocal index arithmetic.
__global__ __launch_bounds__(256) void local_pad_fwd_flat_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                                 int c, int gh, int gw, int p, int pad_mode, int merged) {
  const int q = p + 2, qq = q * q;
  const int H = gh * p, W = gw * p;
  const float inv_q = 1.0f / (float)q, inv_p = 1.0f / (float)p;
  const int64_t planes = (int64_t)n * gh * gw * c;
  const int lane = threadIdx.x & 63;
  const int pp = p * p;
  for (int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); plane < planes; plane += (int64_t)gridDim.x * 4) {
    int64_t r = plane;
    const int ch = (int)(r % c); r /= c;
    const int gc = (int)(r % gw); r /= gw;
    const int gr = (int)(r % gh);
    const int nn = (int)(r / gh);
    float* dst = y + plane * qq;
    if (merged) {
      const float* src = x + (((int64_t)nn * c + ch) * (H + 2) + gr * p) * (W + 2) + gc * p;
      for (int idx = lane; idx < qq; idx += 64) {
        int i = fdiv_small(idx, inv_q), j = idx - i * q;
        dst[idx] = src[(int64_t)i * (W + 2) + j];
      }
      continue;
    }
    const int64_t img = (int64_t)nn * gh * gw;
    for (int idx = lane; idx < qq; idx += 64) {
      int i = fdiv_small(idx, inv_q), j = idx - i * q;
      int Y = gr * p + i - 1, X = gc * p + j - 1;
      bool ok = true;
      if (pad_mode == ITG_PAD_REPLICATE) { Y = min(max(Y, 0), H - 1); X = min(max(X, 0), W - 1); }
      else ok = (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
      float v = 0.f;
      if (ok) {
        int sr = fdiv_small(Y, inv_p), sc = fdiv_small(X, inv_p);
        v = x[((img + sr * gw + sc) * c + ch) * pp + (Y - sr * p) * p + (X - sc * p)];
      }
      dst[idx] = v;
    }
  }
}

// Work item = 8 output rows of one (patch, channel) plane, one wave per item: the plane's grid position
// is wave-uniform, lanes sweep columns, so there is no per-element division at all and every read / write
// is a contiguous run of a row.
constexpr int LP_RB = 8;
__global__ __launch_bounds__(256) void local_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int c,
                                                            int gh, int gw, int p, int pad_mode, int merged) {
  const int q = p + 2, qq = q * q;
  const int H = gh * p, W = gw * p;
  const int nb = (q + LP_RB - 1) / LP_RB;
  const int64_t items = (int64_t)n * gh * gw * c * nb;
  const int lane = threadIdx.x & 63;
  const int pp = p * p;
  for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < items; item += (int64_t)gridDim.x * 4) {
    const int band = (int)(item % nb);
    int64_t plane = item / nb, r = plane;
    const int ch = (int)(r % c); r /= c;
    const int gc = (int)(r % gw); r /= gw;
    const int gr = (int)(r % gh);
    const int nn = (int)(r / gh);
    float* dst = y + plane * qq;
    const int i1 = min(q, (band + 1) * LP_RB);
    if (merged) {
      const float* src = x + (((int64_t)nn * c + ch) * (H + 2) + gr * p) * (W + 2) + gc * p;
      for (int i = band * LP_RB; i < i1; ++i)
        for (int j = lane; j < q; j += 64) dst[i * q + j] = src[(int64_t)i * (W + 2) + j];
      continue;
    }
    const int64_t img = (int64_t)nn * gh * gw;
    for (int i = band * LP_RB; i < i1; ++i) {
      int Y = gr * p + i - 1;
      bool oky = true;
      if (pad_mode == ITG_PAD_REPLICATE) Y = min(max(Y, 0), H - 1);
      else oky = (unsigned)Y < (unsigned)H;
      const int Yc = min(max(Y, 0), H - 1);
      const int sr = Yc / p;                                 // wave-uniform
      const int64_t rowbase = ((img + (int64_t)sr * gw) * c + ch) * pp + (int64_t)(Yc - sr * p) * p;
      for (int j = lane; j < q; j += 64) {
        int X = gc * p + j - 1;
        bool ok = oky;
        if (pad_mode == ITG_PAD_REPLICATE) X = min(max(X, 0), W - 1);
        else ok = ok && (unsigned)X < (unsigned)W;
        const int Xc = min(max(X, 0), W - 1);
        const int sc = (Xc >= (gc + 1) * p) ? gc + 1 : (Xc < gc * p ? gc - 1 : gc);
        float v = x[rowbase + (int64_t)sc * c * pp + (Xc - sc * p)];
        dst[i * q + j] = ok ? v : 0.f;
      }
    }
  }
}

// windows (grid index r, in-window index i) that read padded coordinate Yp along one axis
__device__ __forceinline__ int windows_of(int Yp, int p, int g, int* rr, int* ii, int cnt) {
  // 0 <= Yp + 1 - r*p <= p + 1
  int q = (Yp + 1 >= 0) ? (Yp + 1) / p : -1;
  for (int r = q - 2; r <= q; ++r) {
    if (r < 0 || r >= g) continue;
    int i = Yp + 1 - r * p;
    if (i >= 0 && i <= p + 1) { rr[cnt] = r; ii[cnt] = i; ++cnt; }
  }
  return cnt;
}

__device__ __forceinline__ int sources_of(int Y, int H, int p, int g, int pad_mode, int* rr, int* ii) {
  int cnt = windows_of(Y, p, g, rr, ii, 0);
  if (pad_mode == ITG_PAD_REPLICATE) {
    if (Y == 0) cnt = windows_of(-1, p, g, rr, ii, cnt);
    if (Y == H - 1) cnt = windows_of(H, p, g, rr, ii, cnt);
  }
  return cnt;
}

// Readers of merged coordinate Y along one axis as <= 4 (window, in-window index) pairs held in
// registers (fixed slots, fully unrolled use - runtime-indexed local arrays would live in scratch).
struct AxisReaders { int r[4], i[4], n; };
__device__ __forceinline__ void axis_add(AxisReaders& a, int Yp, int p, int g) {
  const int q0 = (Yp + 1 >= 0) ? (Yp + 1) / p : -1;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int r = q0 - d, i = Yp + 1 - r * p;
    const bool ok = r >= 0 && r < g && i >= 0 && i <= p + 1 && a.n < 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ok && a.n == k) { a.r[k] = r; a.i[k] = i; }
    a.n += ok ? 1 : 0;
  }
}
__device__ __forceinline__ AxisReaders axis_readers(int Y, int H, int p, int g, int pad_mode) {
  AxisReaders a;
  a.n = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { a.r[k] = 0; a.i[k] = 0; }
  axis_add(a, Y, p, g);
  if (pad_mode == ITG_PAD_REPLICATE) {
    if (Y == 0) axis_add(a, -1, p, g);
    if (Y == H - 1) axis_add(a, H, p, g);
  }
  return a;
}
__device__ __forceinline__ float gather_readers(const float* __restrict__ dy, int64_t img_base, int ch, int c, int gh,
                                                int gw, int p, int q, int H, int W, int Y, int X, int pad_mode) {
  const AxisReaders ay = axis_readers(Y, H, p, gh, pad_mode), ax = axis_readers(X, W, p, gw, pad_mode);
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (a < ay.n && b < ax.n)
        s += dy[((((img_base + ay.r[a]) * gw + ax.r[b]) * c + ch) * q + ay.i[a]) * q + ax.i[b]];
  return s;
}

__global__ void local_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n, int c, int gh, int gw,
                                     int p, int pad_mode, int merged) {
  const int q = p + 2;
  const int H = gh * p, W = gw * p;
  if (merged) {
    int64_t total = (int64_t)n * c * (H + 2) * (W + 2);
    GRID_STRIDE(i, total) {
      int64_t r = i;
      int X = (int)(r % (W + 2)); r /= (W + 2);
      int Y = (int)(r % (H + 2)); r /= (H + 2);
      int ch = (int)(r % c);
      int nn = (int)(r / c);
      int ry[4], iy[4], rx[4], ix[4];
      int cy = windows_of(Y - 1, p, gh, ry, iy, 0), cx = windows_of(X - 1, p, gw, rx, ix, 0);
      float s = 0.f;
      for (int a = 0; a < cy; ++a)
        for (int b = 0; b < cx; ++b)
          s += dy[(((((int64_t)nn * gh + ry[a]) * gw + rx[b]) * c + ch) * q + iy[a]) * q + ix[b]];
      dx[i] = s;
    }
    return;
  }
  // work item = 8 rows of one (patch, channel) plane; interior pixels have exactly one reader (own window)
  const int pp = p * p, qq = q * q;
  const int nb = (p + LP_RB - 1) / LP_RB;
  const int64_t items = (int64_t)n * gh * gw * c * nb;
  const int lane = threadIdx.x & 63;
  for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < items; item += (int64_t)gridDim.x * 4) {
    const int band = (int)(item % nb);
    int64_t plane = item / nb, r = plane;
    const int ch = (int)(r % c); r /= c;
    const int gc = (int)(r % gw); r /= gw;
    const int gr = (int)(r % gh);
    const int nn = (int)(r / gh);
    const float* own = dy + plane * qq;
    float* dst = dx + plane * pp;
    const int y1 = min(p, (band + 1) * LP_RB);
    for (int y = band * LP_RB; y < y1; ++y) {
      const bool rowin = y >= 1 && y <= p - 2;
      if (rowin)   // interior columns: one reader, no divergence
        for (int x = 1 + lane; x <= p - 2; x += 64) dst[y * p + x] = own[(y + 1) * q + x + 1];
      // patch-border pixels: whole rows 0 / p-1, else just the two edge columns (lanes 0 and 1)
      for (int k = lane; k < (rowin ? 2 : p); k += 64) {
        const int x = rowin ? (k == 0 ? 0 : p - 1) : k;
        if (rowin && p == 1 && k == 1) continue;
        dst[y * p + x] = gather_readers(dy, (int64_t)nn * gh, ch, c, gh, gw, p, q, H, W, gr * p + y, gc * p + x, pad_mode);
      }
    }
  }
}

// patch-grid NHWC -> patch-grid NHWC with a 1-px halo; left/top carried halos optional
__global__ void local_pad_nhwc_kernel(GridT s, GridT d, const float* __restrict__ left, const float* __restrict__ top,
                                      const float* __restrict__ bottom, int pad_mode) {
  const int q4 = d.ld >> 2;
  int64_t total = (int64_t)d.n * d.gh * d.gw * d.ph * d.pw * q4;
  GRID_STRIDE(i, total) {
    int c4 = (int)(i % q4);
    int64_t pix = i / q4;
    int x = (int)(pix % d.pw); pix /= d.pw;
    int y = (int)(pix % d.ph); pix /= d.ph;
    int gc = (int)(pix % d.gw); pix /= d.gw;
    int gr = (int)(pix % d.gh);
    int n = (int)(pix / d.gh);
    int Y = gr * s.ph + y - 1, X = gc * s.pw + x - 1;   // padded coords in [-1, H], [-1, W]
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    bool done = false;
    if (Y == -1 && top) {
      v = *reinterpret_cast<const f32x4*>(top + ((int64_t)n * (s.W + 2) + X + 1) * s.ld + c4 * 4);
      done = true;
    }
    if (Y == s.H && bottom) {   // halo row received from the rank that owns the patch row below
      v = *reinterpret_cast<const f32x4*>(bottom + ((int64_t)n * (s.W + 2) + X + 1) * s.ld + c4 * 4);
      done = true;
    }
    if (!done) {
      bool zero = false;
      if (Y < 0 || Y >= s.H) { if (pad_mode == ITG_PAD_REPLICATE) Y = min(max(Y, 0), s.H - 1); else zero = true; }
      if (!zero && X == -1 && left) {
        v = *reinterpret_cast<const f32x4*>(left + ((int64_t)n * s.H + Y) * s.ld + c4 * 4);
        done = true;
      }
      if (!done && !zero) {
        if (X < 0 || X >= s.W) { if (pad_mode == ITG_PAD_REPLICATE) X = min(max(X, 0), s.W - 1); else zero = true; }
        if (!zero) v = *reinterpret_cast<const f32x4*>(s.p + grid_off(s, n, Y, X) + c4 * 4);
      }
    }
    *reinterpret_cast<f32x4*>(d.p + (i / q4) * d.ld + c4 * 4) = v;
  }
}

}  // namespace

// One wave that keeps its hardware queue busy for `ticks` of the 100 MHz constant clock and then exits (a time bound
// every wave reaches): the step engine launches it on two streams at once to learn whether they were mapped to
// different hardware queues (see itg_stream_spin in itg.h).
__global__ void stream_spin_kernel(long long ticks, unsigned* sink) {
  const long long t0 = wall_clock64();
  unsigned n = 0;
  while (wall_clock64() - t0 < ticks && n < 0x40000000u) { __builtin_amdgcn_s_sleep(8); ++n; }
  if (sink && threadIdx.x == 0) *sink = n;
}

// ------------------------------------------------------------------------------- halo rows of a row-sharded band (training)
// A rank of a patch grid sharded by patch ROWS holds its band in image layout with one halo row above and below every image:
// ext = (n, 1, 1, H + 2, W, ld), rows 1 .. H its own pixels (written there by the producing BatchNorm, norm.hip BN_PADROWS),
// rows 0 and H + 1 what LocalPadder (reference models/layers.py:145-173) puts around them: the neighbour rank's boundary row
// (arrived over RCCL as an (n, W, ld) buffer), or at the image's outer border the replicate / zero padding of layers.py:82.
enum { BAND_FROM_BUFFER = 0, BAND_REPLICATE = 1, BAND_ZERO = 2 };

__global__ void band_halo_fill_kernel(float* __restrict__ ext, const float* __restrict__ top, const float* __restrict__ bottom, int n,
                                      int H2, int W, int ld, int mode_top, int mode_bottom) {
  const int q4 = ld >> 2;
  const int64_t total = (int64_t)n * 2 * W * q4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % q4);
    int64_t r = i / q4;
    const int x = (int)(r % W); r /= W;
    const int which = (int)(r & 1), img = (int)(r >> 1);
    const int mode = which ? mode_bottom : mode_top;
    float* const row0 = ext + (((int64_t)img * H2) * W + x) * ld + c4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (mode == BAND_FROM_BUFFER) v = *reinterpret_cast<const f32x4*>((which ? bottom : top) + ((int64_t)img * W + x) * ld + c4 * 4);
    else if (mode == BAND_REPLICATE) v = *reinterpret_cast<const f32x4*>(row0 + (int64_t)(which ? H2 - 2 : 1) * W * ld);
    *reinterpret_cast<f32x4*>(row0 + (int64_t)(which ? H2 - 1 : 0) * W * ld) = v;
  }
}

// backward of the fill on the gradient of ext: the halo rows' gradients belong to the rows they were copies of.  Row 1 takes
// what came back from the rank above (`above`: the gradient that rank found in ITS bottom halo row), or - replicate border -
// this tensor's own row 0; row H likewise from below.  (One thread handles both ends of a column: with H = 1 they are one row.)
__global__ void band_halo_grad_kernel(float* __restrict__ g, const float* __restrict__ above, const float* __restrict__ below, int n,
                                      int H2, int W, int ld, int mode_top, int mode_bottom) {
  const int q4 = ld >> 2;
  const int64_t total = (int64_t)n * W * q4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % q4);
    int64_t r = i / q4;
    const int x = (int)(r % W);
    const int img = (int)(r / W);
    float* const col = g + (((int64_t)img * H2) * W + x) * ld + c4 * 4;
    const int64_t rs = (int64_t)W * ld;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int mode = which ? mode_bottom : mode_top;
      if (mode == BAND_ZERO) continue;
      f32x4 a;
      if (mode == BAND_FROM_BUFFER) a = *reinterpret_cast<const f32x4*>((which ? below : above) + ((int64_t)img * W + x) * ld + c4 * 4);
      else a = *reinterpret_cast<const f32x4*>(col + (which ? H2 - 1 : 0) * rs);
      float* const dst = col + (which ? H2 - 2 : 1) * rs;
      *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(dst) + a;
    }
  }
}

// rows r0 and r1 of every image -> two compact (n, W, ld) buffers (what the exchange sends: forward rows 1 / H, backward 0 / H + 1)
__global__ void band_rows_get_kernel(const float* __restrict__ ext, float* __restrict__ out0, float* __restrict__ out1, int n, int H2,
                                     int W, int ld, int r0, int r1) {
  const int q4 = ld >> 2;
  const int64_t total = (int64_t)n * 2 * W * q4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % q4);
    int64_t r = i / q4;
    const int x = (int)(r % W); r /= W;
    const int which = (int)(r & 1), img = (int)(r >> 1);
    float* const out = which ? out1 : out0;
    if (!out) continue;
    *reinterpret_cast<f32x4*>(out + ((int64_t)img * W + x) * ld + c4 * 4) =
        *reinterpret_cast<const f32x4*>(ext + (((int64_t)img * H2 + (which ? r1 : r0)) * W + x) * ld + c4 * 4);
  }
}

// rows 1 .. H of ext <-> a compact band (producers that cannot write the padded layout themselves, e.g. SSM modulation)
template <bool TO_EXT>
__global__ void band_interior_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int H, int W, int ld) {
  const int q4 = ld >> 2;
  const int64_t per = (int64_t)H * W * q4, total = (int64_t)n * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t img = i / per, w = i - img * per;
    const int64_t compact = i * 4, padded = (img * (H + 2) * W * q4 + (int64_t)W * q4 + w) * 4;
    *reinterpret_cast<f32x4*>(dst + (TO_EXT ? padded : compact)) = *reinterpret_cast<const f32x4*>(src + (TO_EXT ? compact : padded));
  }
}

extern "C" {

static int flat_check(const itg_tensor* a, const itg_tensor* b) {
  int rc;
  if ((rc = check_tensor(a)) || (rc = check_tensor(b))) return rc;
  return same_shape(a, b) ? ITG_OK : ITG_ERR_ARG;
}

int itg_act_fwd(const itg_tensor* x, const itg_tensor* y, int act, float slope, void* stream) {
  int rc = flat_check(x, y);
  if (rc) return rc;
  int64_t n4 = grid_pixels(x) * (x->ld >> 2);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x->ptr,
                     (f32x4*)y->ptr, n4, act, slope);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_act_bwd(const itg_tensor* out, const itg_tensor* dout, const itg_tensor* dx, int act, float slope,
                void* stream) {
  int rc = flat_check(out, dout);
  if (rc || (rc = flat_check(out, dx))) return rc;
  int64_t n4 = grid_pixels(out) * (out->ld >> 2);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)out->ptr,
                     (const f32x4*)dout->ptr, (f32x4*)dx->ptr, n4, act, slope);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_add(const itg_tensor* a, const itg_tensor* b, const itg_tensor* out, void* stream) {
  int rc = flat_check(a, b);
  if (rc || (rc = flat_check(a, out))) return rc;
  int64_t n4 = grid_pixels(a) * (a->ld >> 2);
  hipLaunchKernelGGL(add_kernel, dim3(blocks_for(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)a->ptr,
                     (const f32x4*)b->ptr, (f32x4*)out->ptr, n4);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

static int half_check(const itg_tensor* small, const itg_tensor* big) {
  int rc;
  if ((rc = check_tensor(small)) || (rc = check_tensor(big))) return rc;
  if (small->n != big->n || small->gh != big->gh || small->gw != big->gw || small->c != big->c ||
      small->ld != big->ld || 2 * small->ph != big->ph || 2 * small->pw != big->pw)
    return ITG_ERR_ARG;
  return ITG_OK;
}

int itg_upsample2x_fwd(const itg_tensor* x, const itg_tensor* y, void* stream) {
  int rc = half_check(x, y);
  if (rc) return rc;
  int64_t npix = grid_pixels(x);
  int q4 = x->ld >> 2;
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(blocks_for(npix * q4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x->ptr, (float*)y->ptr, npix, q4, x->ph, x->pw);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_upsample2x_bwd(const itg_tensor* dy, const itg_tensor* dx, void* stream) {
  int rc = half_check(dx, dy);
  if (rc) return rc;
  int64_t npix = grid_pixels(dx);
  int q4 = dx->ld >> 2;
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(blocks_for(npix * q4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)dy->ptr, (float*)dx->ptr, npix, q4, dx->ph, dx->pw);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_maxpool2_fwd(const itg_tensor* x, const itg_tensor* y, void* stream) {
  int rc = half_check(y, x);
  if (rc) return rc;
  int64_t npix = grid_pixels(y);
  int q4 = y->ld >> 2;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks_for(npix * q4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x->ptr, (float*)y->ptr, npix, q4, y->ph, y->pw);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_maxpool2_bwd(const itg_tensor* x, const itg_tensor* y, const itg_tensor* dy, const itg_tensor* dx,
                     void* stream) {
  (void)y;
  int rc = half_check(dy, x);
  if (rc || (rc = flat_check(x, dx))) return rc;
  int64_t npix = grid_pixels(dy);
  int q4 = dy->ld >> 2;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks_for(npix * q4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x->ptr, (const float*)dy->ptr, (float*)dx->ptr, npix, q4, dy->ph, dy->pw);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_nchw_to_grid(const float* src, const itg_tensor* dst, int merged_src, void* stream) {
  int rc;
  if (!src || (rc = check_tensor(dst))) return src ? rc : ITG_ERR_ARG;
  GridT d = make_grid(dst);
  int64_t total = grid_pixels(dst) * dst->ld;
  hipLaunchKernelGGL(nchw_to_grid_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, d,
                     merged_src);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_grid_to_nchw(const itg_tensor* src, float* dst, int merged_dst, void* stream) {
  int rc;
  if (!dst || (rc = check_tensor(src))) return dst ? rc : ITG_ERR_ARG;
  GridT s = make_grid(src);
  int64_t total = grid_pixels(src) * src->c;
  hipLaunchKernelGGL(grid_to_nchw_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, s, dst,
                     merged_dst);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_local_pad_fwd(const float* x, float* y, int n, int c, int gh, int gw, int p, int pad_mode, int merged,
                      void* stream) {
  if (!x || !y || n <= 0 || c <= 0 || gh <= 0 || gw <= 0 || p <= 0) return ITG_ERR_ARG;
  if (p + 2 > 2000) return ITG_ERR_ARG;
  if (p + 2 < 48) {
    int64_t planes = (int64_t)n * gh * gw * c;
    hipLaunchKernelGGL(local_pad_fwd_flat_kernel, dim3(blocks_for(planes, 4, 65536)), dim3(256), 0, (hipStream_t)stream, x,
                       y, n, c, gh, gw, p, pad_mode, merged);
  } else {
    int64_t items = (int64_t)n * gh * gw * c * ((p + 2 + LP_RB - 1) / LP_RB);
    hipLaunchKernelGGL(local_pad_fwd_kernel, dim3(blocks_for(items, 4, 65536)), dim3(256), 0, (hipStream_t)stream, x, y,
                       n, c, gh, gw, p, pad_mode, merged);
  }
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_local_pad_bwd(const float* dy, float* dx, int n, int c, int gh, int gw, int p, int pad_mode, int merged,
                      void* stream) {
  if (!dy || !dx || n <= 0 || c <= 0 || gh <= 0 || gw <= 0 || p <= 0) return ITG_ERR_ARG;
  if (p + 2 > 2000) return ITG_ERR_ARG;
  int64_t total = merged ? (int64_t)n * c * (gh * p + 2) * (gw * p + 2) : (int64_t)n * gh * gw * c * ((p + LP_RB - 1) / LP_RB);
  int nb = merged ? blocks_for(total) : blocks_for(total, 4, 65536);
  hipLaunchKernelGGL(local_pad_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, dx, n, c, gh, gw, p,
                     pad_mode, merged);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

static int pad_check(const itg_tensor* x, const itg_tensor* y) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(y))) return rc;
  if (x->n != y->n || x->gh != y->gh || x->gw != y->gw || x->c != y->c || x->ld != y->ld ||
      y->ph != x->ph + 2 || y->pw != x->pw + 2)
    return ITG_ERR_ARG;
  return ITG_OK;
}

int itg_local_pad_nhwc_fwd(const itg_tensor* x, const itg_tensor* y, int pad_mode, void* stream) {
  return itg_local_pad_stream_fwd(x, nullptr, nullptr, nullptr, y, pad_mode, stream);
}

int itg_local_pad_stream_fwd(const itg_tensor* x, const float* left, const float* top, const float* bottom,
                             const itg_tensor* y, int pad_mode, void* stream) {
  int rc = pad_check(x, y);
  if (rc) return rc;
  GridT s = make_grid(x), d = make_grid(y);
  int64_t total = grid_pixels(y) * (y->ld >> 2);
  hipLaunchKernelGGL(local_pad_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, s, d, left, top,
                     bottom, pad_mode);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

static int band_check(const itg_tensor* ext) {
  int rc = check_tensor(ext);
  if (rc) return rc;
  if (ext->gh != 1 || ext->gw != 1 || ext->ph < 3) return ITG_ERR_ARG;      // image layout, at least one own row
  return ITG_OK;
}

int itg_band_halo_fill(const itg_tensor* ext, const float* top, const float* bottom, int mode_top, int mode_bottom, void* stream) {
  int rc = band_check(ext);
  if (rc) return rc;
  if (mode_top < 0 || mode_top > 2 || mode_bottom < 0 || mode_bottom > 2) return ITG_ERR_ARG;
  if ((mode_top == BAND_FROM_BUFFER && !top) || (mode_bottom == BAND_FROM_BUFFER && !bottom)) return ITG_ERR_ARG;
  const int64_t total = (int64_t)ext->n * 2 * ext->pw * (ext->ld >> 2);
  hipLaunchKernelGGL(band_halo_fill_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (float*)ext->ptr, top, bottom,
                     ext->n, ext->ph, ext->pw, ext->ld, mode_top, mode_bottom);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_band_halo_grad(const itg_tensor* gext, const float* above, const float* below, int mode_top, int mode_bottom, void* stream) {
  int rc = band_check(gext);
  if (rc) return rc;
  if (mode_top < 0 || mode_top > 2 || mode_bottom < 0 || mode_bottom > 2) return ITG_ERR_ARG;
  if ((mode_top == BAND_FROM_BUFFER && !above) || (mode_bottom == BAND_FROM_BUFFER && !below)) return ITG_ERR_ARG;
  const int64_t total = (int64_t)gext->n * gext->pw * (gext->ld >> 2);
  hipLaunchKernelGGL(band_halo_grad_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (float*)gext->ptr, above, below,
                     gext->n, gext->ph, gext->pw, gext->ld, mode_top, mode_bottom);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_band_rows_get(const itg_tensor* ext, int r0, int r1, float* out0, float* out1, void* stream) {
  int rc = band_check(ext);
  if (rc) return rc;
  if (r0 < 0 || r0 >= ext->ph || r1 < 0 || r1 >= ext->ph || (!out0 && !out1)) return ITG_ERR_ARG;
  const int64_t total = (int64_t)ext->n * 2 * ext->pw * (ext->ld >> 2);
  hipLaunchKernelGGL(band_rows_get_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)ext->ptr, out0, out1,
                     ext->n, ext->ph, ext->pw, ext->ld, r0, r1);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_band_interior_copy(const itg_tensor* band, const itg_tensor* ext, int to_ext, void* stream) {
  int rc = band_check(ext);
  if (rc || (rc = check_tensor(band))) return rc;
  if (band->gh != 1 || band->gw != 1 || band->n != ext->n || band->ph + 2 != ext->ph || band->pw != ext->pw || band->ld != ext->ld ||
      band->c != ext->c)
    return ITG_ERR_ARG;
  const int64_t total = grid_pixels(band) * (band->ld >> 2);
  if (to_ext)
    hipLaunchKernelGGL(band_interior_copy_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)band->ptr,
                       (float*)ext->ptr, band->n, band->ph, band->pw, band->ld);
  else
    hipLaunchKernelGGL(band_interior_copy_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)ext->ptr,
                       (float*)band->ptr, band->n, band->ph, band->pw, band->ld);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_stream_spin(int microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 100000) return ITG_ERR_ARG;
  hipLaunchKernelGGL(stream_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100, (unsigned*)nullptr);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

}  // extern "C"
