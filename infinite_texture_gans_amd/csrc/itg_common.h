// Internal helpers shared by the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/itg.h"

#define ITG_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return ITG_ERR_LAUNCH;            \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Device-side view of an itg_tensor with the derived merged-image extent.
struct GridT {
  float* p;
  int n, gh, gw, ph, pw, c, ld;
  int H, W;            // merged image extent (gh*ph, gw*pw)
  float inv_ph, inv_pw;
};

static inline GridT make_grid(const itg_tensor* t) {
  GridT g;
  g.p = (float*)t->ptr;
  g.n = t->n; g.gh = t->gh; g.gw = t->gw; g.ph = t->ph; g.pw = t->pw; g.c = t->c; g.ld = t->ld;
  g.H = t->gh * t->ph; g.W = t->gw * t->pw;
  g.inv_ph = 1.0f / (float)t->ph; g.inv_pw = 1.0f / (float)t->pw;
  return g;
}

static inline GridT null_grid() {
  GridT g; g.p = nullptr; g.n = g.gh = g.gw = g.ph = g.pw = g.c = g.ld = 1; g.H = g.W = 1;
  g.inv_ph = g.inv_pw = 1.f; return g;
}

static inline int check_tensor(const itg_tensor* t) {
  if (!t || !t->ptr) return ITG_ERR_ARG;
  if (t->n <= 0 || t->gh <= 0 || t->gw <= 0 || t->ph <= 0 || t->pw <= 0 || t->c <= 0) return ITG_ERR_ARG;
  if (t->ld < t->c || (t->ld & 3)) return ITG_ERR_ALIGN;
  if (((uintptr_t)t->ptr) & 15) return ITG_ERR_ALIGN;
  // 32-bit float offsets inside kernels
  int64_t elems = (int64_t)t->n * t->gh * t->gw * t->ph * t->pw * t->ld;
  if (elems >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
  return ITG_OK;
}

static inline int64_t grid_pixels(const itg_tensor* t) {
  return (int64_t)t->n * t->gh * t->gw * t->ph * t->pw;
}

static inline bool same_shape(const itg_tensor* a, const itg_tensor* b) {
  return a->n == b->n && a->gh == b->gh && a->gw == b->gw && a->ph == b->ph && a->pw == b->pw &&
         a->c == b->c && a->ld == b->ld;
}

// exact floor(x / d) for 0 <= x < 2^22 via one fp multiply (inv = 1/d)
__device__ __forceinline__ int fdiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// float offset of merged-image pixel (n, Y, X), 0 <= Y < H, 0 <= X < W
__device__ __forceinline__ int grid_off(const GridT& g, int n, int Y, int X) {
  int r = fdiv_small(Y, g.inv_ph);
  int c = fdiv_small(X, g.inv_pw);
  int y = Y - r * g.ph;
  int x = X - c * g.pw;
  return ((((n * g.gh + r) * g.gw + c) * g.ph + y) * g.pw + x) * g.ld;
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  if (act == ITG_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == ITG_ACT_TANH) return tanhf(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
