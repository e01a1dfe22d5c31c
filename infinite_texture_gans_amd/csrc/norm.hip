// BatchNorm2d / SSM normalisation kernels for patch-grid NHWC tensors (gfx950).
// Memory-bound: every kernel streams float4 (4 channels of one pixel) with a grid whose
// total thread count is a multiple of ld/4, so a thread's channel group - and therefore its
// alpha/beta/mean/rstd registers - is fixed for the whole sweep; every thread keeps four 16-byte loads in flight.
// Per-channel sums are kept in fp64.  Round 4 (2.6 TB/s, 38 % LDS bank conflicts, 1.1 waves/SIMD):
// the reductions ended with one atomic per channel and workgroup onto the SAME 2*ld doubles, which serialise in L2 - few, wide
// workgroups (round 4: 128 x 1 024 threads; round 6: 512 x 256, see plan_reduce - what counts is the step, not the kernel alone), thread partials
// go wave butterfly (when ld/4 divides 64) -> plain LDS stores -> a fixed-order sum by the thread that owns the channel (no
// ds_add_f64 scatter: 2 048 same-address LDS atomics per group at ld = 16 before), LDS is sized by ld instead of 32 KB flat;
// the apply kernels derive the affine coefficients once per workgroup in LDS (before: two fp64 divisions and a square root
// per channel in EVERY thread, for ~5 pixels of work).  The fp64 (sum, sumsq) pair is what ranks all-reduce for sync-BN.
// Replaces nn.BatchNorm2d at reference models/layers.py:218,279-280 and generators.py:78,115,
// and the modulation arithmetic of StochasticSpatialModulation.forward (layers.py:228-234).
#include <cstdlib>
#include "itg_common.h"

namespace {

constexpr int MAX_LD = 2048;

inline int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// number of 256-thread workgroups: multiple of q4/gcd(256,q4), about `total/(256*per_thread)`
inline int sweep_blocks(int64_t total_f4, int q4, int per_thread, int cap) {
  int g0 = q4 / gcd_i(256, q4);
  int64_t want = (total_f4 + 256LL * per_thread - 1) / (256LL * per_thread);
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  int64_t b = (want + g0 - 1) / g0 * g0;
  return (int)b;
}

// Reductions: at most 512 workgroups of 256 threads.  Every workgroup ends with one fp64 atomic per channel onto the SAME
// 2 * ld doubles, and those serialise at ~13 ns per workgroup; ALONE on the chip, 128 workgroups of 1 024 threads were therefore
// the fastest form (round 4, rocprofv3 on the 75 MB tensor: bn_stats 15.4 / 18.2 / 22.4 / 23.8 us with 128 / 256 / 512 / 1 024
// workgroups against 14.8 - 17.0 us with the atomics compiled out).  Round 6: INSIDE the train step these kernels sit on the
// latency-bound generator chain while weight-gradient kernels on the side streams keep three 168-register workgroups resident
// on every CU - a 16-wave workgroup then waits until two of them have retired on one CU (kernel trace of the replayed step:
// bn_bwd_reduce 81 - 118 us on the 128^2 layers where it takes 30 us alone, 35 - 52 us where it takes 10), a 4-wave workgroup
// takes the first slot that frees.  256 threads everywhere, A/B on one box (tools/probes/multi_ab.sh, medians of 3 x 60 steps):
// cap 128 / 256 / 512 / 1 024 / 2 048 workgroups = 0.999 / 1.002 / 1.010 / 1.000 / 0.994 of the 1 024-thread form; twice.
struct RedPlan { int nt, blocks; size_t lds; };
inline int sweep_blocks_nt(int64_t total_f4, int q4, int per_thread, int cap, int nt) {
  int g0 = q4 / gcd_i(nt, q4);
  int64_t want = (total_f4 + (int64_t)nt * per_thread - 1) / ((int64_t)nt * per_thread);
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)((want + g0 - 1) / g0 * g0);
}
inline RedPlan plan_reduce(int64_t total_f4, int q4, int per_thread) {
  constexpr int cap = 512;
  RedPlan r;
  r.nt = 256;
  r.blocks = sweep_blocks_nt(total_f4, q4, per_thread, cap, r.nt);
  r.lds = (size_t)8 * r.nt * sizeof(double);
  return r;
}

// Workgroup reduction of the per-thread partials a[4] / b[4] (the thread's 4 channels) into gsum[2 * ld] (fp64 atomics, one
// per channel and workgroup).  lds: [8][NT] doubles.  q4 | 64: the lanes of a wave that share a channel group are lane, lane +
// q4, ...: butterfly over those strides, then entry (k, lane < q4) of every wave holds the wave's sum; otherwise every thread
// stores its partials and the owner of a channel adds the NT / q4 threads of its group.  Fixed order either way.
template <int NT>
__device__ __forceinline__ void block_flush(double* lds, const double (&a)[4], const double (&b)[4], int q4, int ld,
                                            double* gsum) {
  const int t = threadIdx.x;
  double v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  const bool pow2 = (64 % q4) == 0;
  if (pow2) {
    for (int o = q4; o < 64; o <<= 1)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += __shfl_xor(v[k], o, 64);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) lds[k * NT + t] = v[k];
  __syncthreads();
  const int stride = pow2 ? 64 : q4;
  const int base = (int)(((int64_t)blockIdx.x * NT) % q4);      // channel group of thread 0 (0 when q4 | 64)
  for (int j = t; j < 2 * ld; j += NT) {
    const int which = j >= ld ? 1 : 0, ch = j - which * ld, cgj = ch >> 2, e = ch & 3;
    int t0 = cgj - base;
    if (t0 < 0) t0 += q4;
    const double* row = lds + (which * 4 + e) * NT;
    double s0 = 0.0, s1 = 0.0;
    int tt = t0;
    for (; tt + stride < NT; tt += 2 * stride) { s0 += row[tt]; s1 += row[tt + stride]; }
    if (tt < NT) s0 += row[tt];
    atomicAdd(&gsum[j], s0 + s1);
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void bn_stats_kernel(const float* __restrict__ x, int64_t npix, int ld,
                                                      double* __restrict__ sums) {
  extern __shared__ double lds[];
  const int q4 = ld >> 2;
  const int64_t T = (int64_t)gridDim.x * NT;
  const int64_t gt = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int cg = (int)(gt % q4);
  const int64_t step = T / q4;
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  int64_t pix = gt / q4;
  for (; pix + 3 * step < npix; pix += 4 * step) {          // four loads in flight per thread (the sweep is latency-bound otherwise)
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (pix + u * step) * ld + cg * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) { double d = v[u][e]; s[e] += d; ss[e] += d * d; }
  }
  for (; pix < npix; pix += step) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ld + cg * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { double d = v[e]; s[e] += d; ss[e] += d * d; }
  }
  block_flush<NT>(lds, s, ss, q4, ld, sums);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, double count_scale,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* running_mean, float* running_var, int64_t* nbt,
                                   float* __restrict__ mean_rstd, float* __restrict__ ab, int c, int ld, int training) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && training && nbt) *nbt += 1;
  if (i >= ld) return;
  float mean = 0.f, rstd = 0.f, a = 0.f, b = 0.f;
  if (i < c) {
    if (training) {
      double m = sums[i] / count;
      double var = sums[ld + i] / count - m * m;
      if (var < 0) var = 0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        double n = count * count_scale;
        double unb = n > 1 ? var * n / (n - 1) : var;
        running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mean;
        running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)unb;
      }
    } else {
      mean = running_mean[i];
      rstd = 1.0f / sqrtf(running_var[i] + eps);
    }
    float g = gamma ? gamma[i] : 1.f;
    a = g * rstd;
    b = (beta ? beta[i] : 0.f) - mean * a;
  }
  mean_rstd[i] = mean; mean_rstd[ld + i] = rstd;
  ab[i] = a; ab[ld + i] = b;
}

// store one normalised pixel group; UPS = 1: y has 2x the patch extent (nearest); UPS = 2 (BN_PADROWS): y is a row-sharded
// band in image layout with ONE HALO ROW above and below every image ((ph + 2) x pw pixels per image, rows 1 .. ph written:
// the halo rows are the neighbour ranks' pixels, reference models/layers.py:145-173, filled by itg_band_halo_fill)
constexpr int BN_PADROWS = 2;
__device__ __forceinline__ int64_t padrows_pix(int64_t pix, int ph, int pw) {
  const int64_t img = pix / ((int64_t)ph * pw);
  return pix + (2 * img + 1) * pw;
}
template <int UPS>
__device__ __forceinline__ void store_y(float* __restrict__ y, int64_t pix, int ld, int cg, int ph, int pw, f32x4 v) {
  if (UPS == BN_PADROWS) {
    *reinterpret_cast<f32x4*>(y + padrows_pix(pix, ph, pw) * ld + cg * 4) = v;
  } else if (!UPS) {
    *reinterpret_cast<f32x4*>(y + pix * ld + cg * 4) = v;
  } else {
    int64_t blk = pix / (ph * pw);
    int r = (int)(pix - blk * ph * pw);
    int yy = r / pw, xx = r - yy * pw;
    float* o = y + ((blk * 2 * ph + 2 * yy) * (2 * pw) + 2 * xx) * (int64_t)ld + cg * 4;
    *reinterpret_cast<f32x4*>(o) = v;
    *reinterpret_cast<f32x4*>(o + ld) = v;
    *reinterpret_cast<f32x4*>(o + (int64_t)2 * pw * ld) = v;
    *reinterpret_cast<f32x4*>(o + (int64_t)2 * pw * ld + ld) = v;
  }
}

// the sweep y = act(a * x + b) of a thread's channel group: four 16-byte loads in flight
template <int UPS>
__device__ __forceinline__ void apply_sweep(const float* __restrict__ x, float* __restrict__ y, f32x4 a, f32x4 b, int64_t pix,
                                            int64_t step, int64_t npix, int ld, int cg, int ph, int pw, int act, float slope) {
  constexpr int UN = 4;
  for (; pix + (UN - 1) * step < npix; pix += UN * step) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (pix + u * step) * ld + cg * 4);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = act_apply(fmaf(v[u][e], a[e], b[e]), act, slope);
      store_y<UPS>(y, pix + u * step, ld, cg, ph, pw, v[u]);
    }
  }
  for (; pix < npix; pix += step) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ld + cg * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(fmaf(v[e], a[e], b[e]), act, slope);
    store_y<UPS>(y, pix, ld, cg, ph, pw, v);
  }
}

// y = act(alpha*x + beta); ups: y has 2x the patch extent (nearest)
template <int UPS>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ ab, int64_t npix, int ld, int ph,
                                                       int pw, int act, float slope) {
  const int q4 = ld >> 2;
  const int64_t T = (int64_t)gridDim.x * 256;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gt % q4);
  const f32x4 a = *reinterpret_cast<const f32x4*>(ab + cg * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(ab + ld + cg * 4);
  apply_sweep<UPS>(x, y, a, b, gt / q4, T / q4, npix, ld, cg, ph, pw, act, slope);
}

// bn_finalize + bn_apply in one launch (training): every WORKGROUP derives the affine coefficients of all channels from the
// fp64 sums once, into LDS (one thread per channel); workgroup 0 also publishes mean / rstd / (a, b) for the backward and
// updates the running statistics.
template <int UPS>
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                const double* __restrict__ sums, double count,
                                                                double count_scale, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, float momentum,
                                                                float* running_mean, float* running_var, int64_t* nbt,
                                                                float* __restrict__ mean_rstd, float* __restrict__ ab,
                                                                int64_t npix, int c, int ld, int ph, int pw, int act,
                                                                float slope) {
  extern __shared__ float coef[];                 // a[ld] | b[ld]
  const int q4 = ld >> 2;
  const int64_t T = (int64_t)gridDim.x * 256;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gt % q4);
  for (int i = threadIdx.x; i < ld; i += 256) {
    float mean = 0.f, rstd = 0.f, av = 0.f, bv = 0.f;
    if (i < c) {
      double m = sums[i] / count;                  // the arithmetic of bn_finalize_kernel, bit for bit
      double var = sums[ld + i] / count - m * m;
      if (var < 0) var = 0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)eps));
      if (blockIdx.x == 0 && running_mean) {
        double n = count * count_scale;
        double unb = n > 1 ? var * n / (n - 1) : var;
        running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mean;
        running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)unb;
      }
      float g = gamma ? gamma[i] : 1.f;
      av = g * rstd;
      bv = (beta ? beta[i] : 0.f) - mean * av;
    }
    coef[i] = av; coef[ld + i] = bv;
    if (blockIdx.x == 0) {
      mean_rstd[i] = mean; mean_rstd[ld + i] = rstd;
      ab[i] = av; ab[ld + i] = bv;
    }
  }
  if (gt == 0 && nbt) *nbt += 1;
  __syncthreads();
  const f32x4 a = *reinterpret_cast<const f32x4*>(coef + cg * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(coef + ld + cg * 4);
  apply_sweep<UPS>(x, y, a, b, gt / q4, T / q4, npix, ld, cg, ph, pw, act, slope);
}

template <int UPS>
__device__ __forceinline__ f32x4 load_dy(const float* __restrict__ dy, int64_t pix, int ld, int cg, int ph, int pw) {
  if (UPS == BN_PADROWS) return *reinterpret_cast<const f32x4*>(dy + padrows_pix(pix, ph, pw) * ld + cg * 4);
  if (!UPS) return *reinterpret_cast<const f32x4*>(dy + pix * ld + cg * 4);
  int64_t blk = pix / (ph * pw);
  int r = (int)(pix - blk * ph * pw);
  int yy = r / pw, xx = r - yy * pw;
  const float* o = dy + ((blk * 2 * ph + 2 * yy) * (2 * pw) + 2 * xx) * (int64_t)ld + cg * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(o);
  v += *reinterpret_cast<const f32x4*>(o + ld);
  v += *reinterpret_cast<const f32x4*>(o + (int64_t)2 * pw * ld);
  v += *reinterpret_cast<const f32x4*>(o + (int64_t)2 * pw * ld + ld);
  return v;
}

__device__ __forceinline__ float act_grad(float pre, int act, float slope) {
  if (act == ITG_ACT_LRELU) return pre > 0.f ? 1.f : slope;
  if (act == ITG_ACT_TANH) { float t = tanhf(pre); return 1.f - t * t; }
  return 1.f;
}

template <int UPS, int NT>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ ab,
                                                           const float* __restrict__ mean_rstd, int64_t npix, int ld,
                                                           int ph, int pw, int act, float slope,
                                                           double* __restrict__ sums) {
  extern __shared__ double lds[];
  const int q4 = ld >> 2;
  const int64_t T = (int64_t)gridDim.x * NT;
  const int64_t gt = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int cg = (int)(gt % q4);
  const int64_t step = T / q4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(ab + cg * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(ab + ld + cg * 4);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean_rstd + cg * 4);
  const f32x4 rs = *reinterpret_cast<const f32x4*>(mean_rstd + ld + cg * 4);
  double s[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0};
  int64_t pix = gt / q4;
  constexpr int UN = UPS == 1 ? 2 : 4;                            // pixels in flight per thread (an upsampled dy is 4 loads per pixel)
  for (; pix + (UN - 1) * step < npix; pix += UN * step) {
    f32x4 v[UN], g[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      v[u] = *reinterpret_cast<const f32x4*>(x + (pix + u * step) * ld + cg * 4);
      g[u] = load_dy<UPS>(dy, pix + u * step, ld, cg, ph, pw);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float ge = g[u][e] * act_grad(fmaf(v[u][e], a[e], b[e]), act, slope);
        float xh = (v[u][e] - mu[e]) * rs[e];
        s[e] += (double)ge;
        sx[e] += (double)ge * (double)xh;
      }
  }
  for (; pix < npix; pix += step) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ld + cg * 4);
    f32x4 g = load_dy<UPS>(dy, pix, ld, cg, ph, pw);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = g[e] * act_grad(fmaf(v[e], a[e], b[e]), act, slope);
      float xh = (v[e] - mu[e]) * rs[e];
      s[e] += (double)ge;
      sx[e] += (double)ge * (double)xh;
    }
  }
  block_flush<NT>(lds, s, sx, q4, ld, sums);
}

// ADD: dx also takes a second gradient of the BatchNorm's input (`addend`, x's shape): the tensor feeds the residual
// shortcut as well (reference models/layers.py:313-322), and the sum of its two gradients is otherwise a launch of its own
template <int UPS, bool ADD>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ ab,
                                                           const float* __restrict__ mean_rstd,
                                                           const double* __restrict__ sums,
                                                           const double* __restrict__ sums_local, double inv_count,
                                                           int64_t npix, int ld, int c, int ph, int pw, int act,
                                                           float slope, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int accumulate, const float* __restrict__ addend) {
  const int q4 = ld >> 2;
  const int64_t T = (int64_t)gridDim.x * 256;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gt % q4);
  const int64_t step = T / q4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(ab + cg * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(ab + ld + cg * 4);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean_rstd + cg * 4);
  const f32x4 rs = *reinterpret_cast<const f32x4*>(mean_rstd + ld + cg * 4);
  f32x4 m1, m2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m1[e] = (float)(sums[cg * 4 + e] * inv_count);
    m2[e] = (float)(sums[ld + cg * 4 + e] * inv_count);
  }
  if (gt < ld) {
    int i = (int)gt;
    if (dgamma && i < c) dgamma[i] = (accumulate ? dgamma[i] : 0.f) + (float)sums_local[ld + i];
    if (dbeta && i < c) dbeta[i] = (accumulate ? dbeta[i] : 0.f) + (float)sums_local[i];
  }
  auto one = [&](int64_t pix, f32x4 v, f32x4 g, f32x4 ad) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = g[e] * act_grad(fmaf(v[e], a[e], b[e]), act, slope);
      float xh = (v[e] - mu[e]) * rs[e];
      o[e] = a[e] * (ge - m1[e] - xh * m2[e]);
    }
    if constexpr (ADD) o += ad;
    *reinterpret_cast<f32x4*>(dx + pix * ld + cg * 4) = o;
  };
  // one pixel group per iteration: with ~100 registers the unrolled form (3-4 pixels in flight) lost occupancy and measured
  // slower (38.3 vs 33.9 us on the 75 MB tensor); the grid supplies the loads in flight instead
  for (int64_t pix = gt / q4; pix < npix; pix += step) {
    f32x4 ad = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (ADD) ad = *reinterpret_cast<const f32x4*>(addend + pix * ld + cg * 4);
    one(pix, *reinterpret_cast<const f32x4*>(x + pix * ld + cg * 4), load_dy<UPS>(dy, pix, ld, cg, ph, pw), ad);
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t npix, int ld,
                                                     double* __restrict__ sums) {
  __shared__ double lds[MAX_LD];
  const int q4 = ld >> 2;
  for (int i = threadIdx.x; i < ld; i += 256) lds[i] = 0.0;
  __syncthreads();
  const int64_t T = (int64_t)gridDim.x * 256;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gt % q4);
  const int64_t step = T / q4;
  double s[4] = {0, 0, 0, 0};
  for (int64_t pix = gt / q4; pix < npix; pix += step) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ld + cg * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
  }
  for (int e = 0; e < 4; ++e) atomicAdd(&lds[cg * 4 + e], s[e]);
  __syncthreads();
  for (int i = threadIdx.x; i < ld; i += 256) atomicAdd(&sums[i], lds[i]);
}

// SSM modulation: y = act((1+gamma)*xhat + beta)
__global__ void ssm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                               const float* __restrict__ emb, float* __restrict__ y, int64_t npix, int c, int ld,
                               int ld_e, int act, float slope) {
  int64_t total = npix * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % ld);
    int64_t pix = i / ld;
    float o = 0.f;
    if (ch < c) {
      float xh = (x[i] - mean_rstd[ch]) * mean_rstd[ld + ch];
      float g = emb[pix * ld_e + ch], b = emb[pix * ld_e + c + ch];
      o = act_apply(fmaf(1.f + g, xh, b), act, slope);
    }
    y[i] = o;
  }
}

__global__ void ssm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                               const float* __restrict__ emb, const float* __restrict__ dy, float* __restrict__ dxhat,
                               float* __restrict__ demb, int64_t npix, int c, int ld, int ld_e, int act, float slope) {
  int64_t total = npix * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % ld);
    int64_t pix = i / ld;
    float o = 0.f;
    if (ch < c) {
      float xh = (x[i] - mean_rstd[ch]) * mean_rstd[ld + ch];
      float g = emb[pix * ld_e + ch], b = emb[pix * ld_e + c + ch];
      float ge = dy[i] * act_grad(fmaf(1.f + g, xh, b), act, slope);
      demb[pix * ld_e + ch] = ge * xh;
      demb[pix * ld_e + c + ch] = ge;
      o = ge * (1.f + g);
    }
    dxhat[i] = o;
  }
}

__global__ void d2f_kernel(const double* __restrict__ s, float* __restrict__ o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = (float)s[i];
}

}  // namespace

extern "C" {

int itg_bn_stats(const itg_tensor* x, double* sums, void* stream) {
  int rc;
  if ((rc = check_tensor(x))) return rc;
  if (!sums || x->ld > MAX_LD) return ITG_ERR_ARG;
  int64_t npix = grid_pixels(x);
  int q4 = x->ld >> 2;
  const RedPlan rp = plan_reduce(npix * q4, q4, 8);
  hipLaunchKernelGGL(bn_stats_kernel<256>, dim3(rp.blocks), dim3(256), rp.lds, (hipStream_t)stream, (const float*)x->ptr, npix,
                       x->ld, sums);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_bn_finalize(const double* sums, double count, double count_scale, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var, int64_t* nbt,
                    float* mean_rstd, float* ab, int c, int ld, int training, void* stream) {
  if (!mean_rstd || !ab || c <= 0 || ld < c) return ITG_ERR_ARG;
  if (training && !sums) return ITG_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return ITG_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((ld + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums, count,
                     count_scale, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean_rstd, ab, c, ld,
                     training);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// 0: same extent, 1: `big` has twice the patch extent (nearest x2 upsample), 2 (BN_PADROWS): `big` is the image-layout band
// with one halo row above and below (ph + 2 rows)
static int ups_mode(const itg_tensor* small, const itg_tensor* big, int* ups) {
  if (small->n != big->n || small->gh != big->gh || small->gw != big->gw || small->c != big->c ||
      small->ld != big->ld)
    return ITG_ERR_ARG;
  if (small->ph == big->ph && small->pw == big->pw) { *ups = 0; return ITG_OK; }
  if (2 * small->ph == big->ph && 2 * small->pw == big->pw) { *ups = 1; return ITG_OK; }
  if (small->gh == 1 && small->gw == 1 && small->ph + 2 == big->ph && small->pw == big->pw) { *ups = BN_PADROWS; return ITG_OK; }
  return ITG_ERR_ARG;
}

int itg_bn_finalize_apply(const itg_tensor* x, const double* sums, double count, double count_scale, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          int64_t* nbt, float* mean_rstd, float* ab, const itg_tensor* y, int act, float slope,
                          void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(y))) return rc;
  int ups;
  if (!sums || !mean_rstd || !ab || count <= 0 || (rc = ups_mode(x, y, &ups))) return rc ? rc : ITG_ERR_ARG;
  int64_t npix = grid_pixels(x);
  int q4 = x->ld >> 2;
  if (x->ld > MAX_LD) return ITG_ERR_ARG;
  int blocks = sweep_blocks(npix * q4, q4, 8, 2048);
  const size_t lds = (size_t)2 * x->ld * sizeof(float);
  if (ups == BN_PADROWS)
    hipLaunchKernelGGL(bn_finalize_apply_kernel<BN_PADROWS>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, (const float*)x->ptr,
                       (float*)y->ptr, sums, count, count_scale, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                       mean_rstd, ab, npix, x->c, x->ld, x->ph, x->pw, act, slope);
  else if (ups)
    hipLaunchKernelGGL(bn_finalize_apply_kernel<1>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, (const float*)x->ptr,
                       (float*)y->ptr, sums, count, count_scale, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                       mean_rstd, ab, npix, x->c, x->ld, x->ph, x->pw, act, slope);
  else
    hipLaunchKernelGGL(bn_finalize_apply_kernel<0>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, (const float*)x->ptr,
                       (float*)y->ptr, sums, count, count_scale, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                       mean_rstd, ab, npix, x->c, x->ld, x->ph, x->pw, act, slope);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_bn_apply(const itg_tensor* x, const float* ab, const itg_tensor* y, int act, float slope, void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(y))) return rc;
  int ups;
  if (!ab || (rc = ups_mode(x, y, &ups))) return rc ? rc : ITG_ERR_ARG;
  int64_t npix = grid_pixels(x);
  int q4 = x->ld >> 2;
  int blocks = sweep_blocks(npix * q4, q4, 8, 2048);
  if (ups == BN_PADROWS)
    hipLaunchKernelGGL(bn_apply_kernel<BN_PADROWS>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr,
                       (float*)y->ptr, ab, npix, x->ld, x->ph, x->pw, act, slope);
  else if (ups)
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr,
                       (float*)y->ptr, ab, npix, x->ld, x->ph, x->pw, act, slope);
  else
    hipLaunchKernelGGL(bn_apply_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr,
                       (float*)y->ptr, ab, npix, x->ld, x->ph, x->pw, act, slope);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_bn_bwd_reduce(const itg_tensor* x, const itg_tensor* dy, const float* ab, const float* mean_rstd, int act,
                      float slope, double* sums, void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(dy))) return rc;
  int ups;
  if (!ab || !mean_rstd || !sums || x->ld > MAX_LD) return ITG_ERR_ARG;
  if ((rc = ups_mode(x, dy, &ups))) return rc;
  int64_t npix = grid_pixels(x);
  int q4 = x->ld >> 2;
  const RedPlan rp = plan_reduce(npix * q4, q4, ups == 1 ? 4 : 8);
#define ITG_BWD_RED(U, N)                                                                                              \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<U, N>), dim3(rp.blocks), dim3(N), rp.lds, (hipStream_t)stream,               \
                     (const float*)x->ptr, (const float*)dy->ptr, ab, mean_rstd, npix, x->ld, x->ph, x->pw, act, slope, sums)
  if (ups == BN_PADROWS) ITG_BWD_RED(BN_PADROWS, 256); else if (ups) ITG_BWD_RED(1, 256); else ITG_BWD_RED(0, 256);
#undef ITG_BWD_RED
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_bn_bwd_apply_add(const itg_tensor* x, const itg_tensor* dy, const float* ab, const float* mean_rstd,
                         const double* sums_local, const double* sums, double count, int act, float slope,
                         const itg_tensor* dx, float* dgamma, float* dbeta, int accumulate, const itg_tensor* addend,
                         void* stream) {
  if (!sums_local) sums_local = sums;
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(dy)) || (rc = check_tensor(dx))) return rc;
  int ups;
  if (!ab || !mean_rstd || !sums || count <= 0 || !same_shape(x, dx)) return ITG_ERR_ARG;
  if ((rc = ups_mode(x, dy, &ups))) return rc;
  const float* ad = nullptr;
  if (addend && addend->ptr) {
    if ((rc = check_tensor(addend))) return rc;
    if (!same_shape(x, addend) || addend->ptr == dx->ptr) return ITG_ERR_ARG;
    ad = (const float*)addend->ptr;
  }
  int64_t npix = grid_pixels(x);
  int q4 = x->ld >> 2;
  int blocks = sweep_blocks(npix * q4, q4, 4, 4096);
  if ((int64_t)blocks * 256 < x->ld) blocks = sweep_blocks((int64_t)x->ld * q4, q4, 1, 4096);
#define ITG_BWD_APPLY(U, A)                                                                                                     \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<U, A>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr,        \
                     (const float*)dy->ptr, ab, mean_rstd, sums, sums_local, 1.0 / count, npix, x->ld, x->c, x->ph, x->pw, act, \
                     slope, (float*)dx->ptr, dgamma, dbeta, accumulate, ad)
  if (ups == BN_PADROWS) { if (ad) ITG_BWD_APPLY(BN_PADROWS, true); else ITG_BWD_APPLY(BN_PADROWS, false); }
  else if (ups) { if (ad) ITG_BWD_APPLY(1, true); else ITG_BWD_APPLY(1, false); }
  else { if (ad) ITG_BWD_APPLY(0, true); else ITG_BWD_APPLY(0, false); }
#undef ITG_BWD_APPLY
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_bn_bwd_apply(const itg_tensor* x, const itg_tensor* dy, const float* ab, const float* mean_rstd,
                     const double* sums_local, const double* sums, double count, int act, float slope,
                     const itg_tensor* dx, float* dgamma, float* dbeta, int accumulate, void* stream) {
  return itg_bn_bwd_apply_add(x, dy, ab, mean_rstd, sums_local, sums, count, act, slope, dx, dgamma, dbeta, accumulate, nullptr,
                              stream);
}

// per-channel sum over all pixels -> out (fp32[ld]); acc is fp64[ld] scratch owned by the caller
int itg_colsum(const itg_tensor* x, float* out, double* acc, void* stream) {
  int rc;
  if ((rc = check_tensor(x))) return rc;
  if (!out || !acc || x->ld > MAX_LD) return ITG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int ld = x->ld;
  if (hipMemsetAsync(acc, 0, sizeof(double) * ld, s) != hipSuccess) return ITG_ERR_LAUNCH;
  int64_t npix = grid_pixels(x);
  int q4 = ld >> 2;
  int blocks = sweep_blocks(npix * q4, q4, 8, 2048);
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, s, (const float*)x->ptr, npix, ld, acc);
  ITG_CHECK_LAUNCH();
  hipLaunchKernelGGL(d2f_kernel, dim3((x->c + 127) / 128), dim3(128), 0, s, (const double*)acc, out, x->c);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_ssm_modulate_fwd(const itg_tensor* x, const float* mean_rstd, const itg_tensor* emb, const itg_tensor* y,
                         int act, float slope, void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(emb)) || (rc = check_tensor(y))) return rc;
  if (!mean_rstd || !same_shape(x, y) || emb->c != 2 * x->c || grid_pixels(emb) != grid_pixels(x)) return ITG_ERR_ARG;
  int64_t npix = grid_pixels(x);
  int64_t total = npix * x->ld;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(ssm_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, mean_rstd,
                     (const float*)emb->ptr, (float*)y->ptr, npix, x->c, x->ld, emb->ld, act, slope);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_ssm_modulate_bwd(const itg_tensor* x, const float* mean_rstd, const itg_tensor* emb, const itg_tensor* dy,
                         int act, float slope, const itg_tensor* dxhat, const itg_tensor* demb, void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(emb)) || (rc = check_tensor(dy)) || (rc = check_tensor(dxhat)) ||
      (rc = check_tensor(demb)))
    return rc;
  if (!mean_rstd || !same_shape(x, dy) || !same_shape(x, dxhat) || !same_shape(emb, demb) || emb->c != 2 * x->c)
    return ITG_ERR_ARG;
  int64_t npix = grid_pixels(x);
  int64_t total = npix * x->ld;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(ssm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, mean_rstd,
                     (const float*)emb->ptr, (const float*)dy->ptr, (float*)dxhat->ptr, (float*)demb->ptr, npix,
                     x->c, x->ld, emb->ld, act, slope);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

}  // extern "C"
