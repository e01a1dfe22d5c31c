// Losses, spectral-norm power iteration and the fused Adam(+EMA) step (gfx950).
// Reductions are per-wave __shfl_xor trees (64-wide) feeding one LDS slot per wave.
//   BCE-with-logits : reference train.py:81,131-132,148-149,164-165 (nn.BCEWithLogitsLoss, mean)
//   hinge           : build-side extra, absent from the reference (utils.py:85 is never read)
//   spectral norm   : torch.nn.utils.spectral_norm as wrapped at reference models/layers.py:190-194
//   Adam / EMA      : reference train.py:57-58,153,169,176-180
#include <algorithm>
#include "itg_common.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v) {
  __shared__ double part[16];
  v = wave_sum_d(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) part[w] = v;
  __syncthreads();
  double t = 0.0;
  int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) t += part[i];
  return t;
}

__device__ __forceinline__ float bce_elem(float x, float t) {
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

// single workgroup: deterministic
__global__ __launch_bounds__(1024) void bce_fwd_kernel(const float* __restrict__ x, int64_t n, float t,
                                                       float* __restrict__ out) {
  // four independent partial sums per thread: four loads in flight (the loop is latency-bound: 17 dependent rounds for
  // D(fake)'s 17 672 logits took 18 us), fixed summation order
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int64_t i = threadIdx.x;
  const int64_t st = blockDim.x;
  for (; i + 3 * st < n; i += 4 * st) {
    const float a = x[i], b = x[i + st], c = x[i + 2 * st], d = x[i + 3 * st];
    s0 += (double)bce_elem(a, t); s1 += (double)bce_elem(b, t); s2 += (double)bce_elem(c, t); s3 += (double)bce_elem(d, t);
  }
  for (; i < n; i += st) s0 += (double)bce_elem(x[i], t);
  double s = block_sum_d((s0 + s1) + (s2 + s3));
  if (threadIdx.x == 0) *out = (float)(s / (double)n);
}

__global__ void bce_bwd_kernel(const float* __restrict__ x, int64_t n, float t, const float* __restrict__ up,
                               float* __restrict__ dx) {
  float sc = (up ? *up : 1.f) / (float)n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = x[i];
    float sg = 1.f / (1.f + expf(-v));
    dx[i] = (sg - t) * sc;
  }
}

// hinge family as max(a + b*x, lo): D real (1,-1,0), D fake (1,1,0), G (0,-1,-inf)
struct HingeC { float a, b, lo; };
__device__ __forceinline__ HingeC hinge_consts(int mode) {
  HingeC c;
  c.a = mode == 2 ? 0.f : 1.f;
  c.b = mode == 1 ? 1.f : -1.f;
  c.lo = mode == 2 ? -INFINITY : 0.f;
  return c;
}

// The loss head on the discriminator's logit map AS THE LAST CONV LEFT IT (patch-grid layout, one channel in an ld-wide
// pixel): the mean loss and d loss / d logit in the same layout, one launch (the NCHW round trip of the generic heads above -
// grid_to_nchw, the single-workgroup forward sum, the backward, nchw_to_grid - was four latency-bound launches, ~32 us of
// every D pass's critical path).  kind 0: BCE-with-logits against t; 1..3: the hinge modes 0..2.
// Blocks leave fp64 partial sums; the last one to finish adds them in block order (deterministic) - `counter` must be zero
// at launch (a slice of the step's zeroed arena).
constexpr int LOSS_GRID_MAX_BLOCKS = 64;
__global__ __launch_bounds__(256) void logit_loss_grid_kernel(const float* __restrict__ x, long long npix, int ld, int kind, float t,
                                                              float* __restrict__ dl, double* __restrict__ partial,
                                                              unsigned* __restrict__ counter, float* __restrict__ out) {
  const float inv_n = 1.f / (float)npix;
  const HingeC c = hinge_consts(kind > 0 ? kind - 1 : 0);
  double s = 0.0;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const float v = x[i * ld];
    float f, d;
    if (kind == 0) {
      f = bce_elem(v, t);
      d = 1.f / (1.f + expf(-v)) - t;
    } else {
      const float a = fmaf(c.b, v, c.a);
      f = fmaxf(a, c.lo);
      d = a > c.lo ? c.b : 0.f;
    }
    s += (double)f;
    dl[i * ld] = d * inv_n;
    for (int k = 1; k < ld; ++k) dl[i * ld + k] = 0.f;
  }
  s = block_sum_d(s);
  __shared__ bool last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partial[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    double tot = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) tot += __hip_atomic_load(&partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out = (float)(tot / (double)npix);
  }
}

__global__ __launch_bounds__(1024) void hinge_fwd_kernel(const float* __restrict__ x, int64_t n, int mode,
                                                         float* __restrict__ out) {
  const HingeC c = hinge_consts(mode);
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)fmaxf(fmaf(c.b, x[i], c.a), c.lo);
  s = block_sum_d(s);
  if (threadIdx.x == 0) *out = (float)(s / (double)n);
}

__global__ void hinge_bwd_kernel(const float* __restrict__ x, int64_t n, int mode, const float* __restrict__ up,
                                 float* __restrict__ dx) {
  const HingeC c = hinge_consts(mode);
  float sc = (up ? *up : 1.f) / (float)n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = (fmaf(c.b, x[i], c.a) > c.lo) ? c.b * sc : 0.f;
}

// ---- spectral norm ------------------------------------------------------------------------
// t[j] = sum_i W[i][j] * u[i]   (one thread per column, coalesced across columns)
// (64 columns x 4 row lanes per workgroup, blockIdx.y = one of SN_RS row slices; partial sums per slice)
constexpr int SN_RS = 8;
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                      float* __restrict__ part, int rows, int cols) {
  __shared__ double red[4][64];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int per = (rows + SN_RS - 1) / SN_RS;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0;
  if (j < cols)
    for (int i = r0 + rl; i < r1; i += 4) s += (double)w[(size_t)i * cols + j] * (double)u[i];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && j < cols)
    part[(size_t)blockIdx.y * cols + j] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                                  red[3][threadIdx.x]);
}

// dst = src / max(||src||, eps), src = sum of `nparts` partial vectors    (single workgroup)
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const float* __restrict__ src, float* __restrict__ dst, int n,
                                                            int nparts, float eps) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v = 0.f;
    for (int p = 0; p < nparts; ++p) v += src[(size_t)p * n + i];
    dst[i] = v;
    s += (double)v * (double)v;
  }
  s = block_sum_d(s);
  float nrm = fmaxf((float)sqrt(s), eps);
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = dst[i] / nrm;
}

// t[i] = sum_j W[i][j] * v[j]   (one wave per row)
__global__ void sn_w_v_kernel(const float* __restrict__ w, const float* __restrict__ v, float* __restrict__ t, int rows,
                              int cols) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  double s = 0.0;
  for (int j = lane; j < cols; j += 64) s += (double)w[(size_t)row * cols + j] * (double)v[j];
  s = wave_sum_d(s);
  if (lane == 0) t[row] = (float)s;
}

// sigma = dot(u, wv); inv = 1/sigma
__global__ __launch_bounds__(1024) void sn_sigma_kernel(const float* __restrict__ u, const float* __restrict__ wv, int rows,
                                                        float* __restrict__ sigma, float* __restrict__ inv) {
  double s = 0.0;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) s += (double)u[i] * (double)wv[i];
  s = block_sum_d(s);
  if (threadIdx.x == 0) {
    float sg = (float)s;
    if (sigma) *sigma = sg;
    if (inv) *inv = 1.f / sg;
  }
}

// ---- all spectrally-normalised layers of a model in 4 launches ---------------------------------
constexpr int SN_MAXL = 8;
struct SnBatch {
  const float* w[SN_MAXL]; float* u[SN_MAXL]; float* v[SN_MAXL]; float* work[SN_MAXL]; float* inv[SN_MAXL];
  int rows[SN_MAXL], cols[SN_MAXL];
  int blk0[SN_MAXL + 1];   // prefix of per-layer workgroup counts for the flattened grids
  int n;
  float eps;
};

__device__ __forceinline__ int sn_find_layer(const SnBatch& b, int blk) {
  int l = 0;
  while (l + 1 < b.n && blk >= b.blk0[l + 1]) ++l;
  return l;
}

// partial W^T u : blockIdx.x flattened over layers x column blocks of 64, blockIdx.y = row slice
__global__ __launch_bounds__(256) void snb_wt_u_kernel(SnBatch b) {
  __shared__ double red[4][64];
  const int l = sn_find_layer(b, blockIdx.x);
  const int rows = b.rows[l], cols = b.cols[l];
  const float* w = b.w[l]; const float* u = b.u[l]; float* part = b.work[l];
  const int j = (blockIdx.x - b.blk0[l]) * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int per = (rows + SN_RS - 1) / SN_RS;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0;
  if (j < cols)
    for (int i = r0 + rl; i < r1; i += 4) s += (double)w[(size_t)i * cols + j] * (double)u[i];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && j < cols)
    part[(size_t)blockIdx.y * cols + j] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                                  red[3][threadIdx.x]);
}

// v = normalize(sum of partials) : one workgroup per layer
__global__ __launch_bounds__(1024) void snb_norm_v_kernel(SnBatch b) {
  const int l = blockIdx.x;
  const int n = b.cols[l];
  const float* src = b.work[l]; float* dst = b.v[l];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v = 0.f;
    for (int p = 0; p < SN_RS; ++p) v += src[(size_t)p * n + i];
    dst[i] = v;
    s += (double)v * (double)v;
  }
  s = block_sum_d(s);
  float nrm = fmaxf((float)sqrt(s), b.eps);
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = dst[i] / nrm;
}

// t = W v : one workgroup per row (blockIdx.x flattened over layers x rows), 16-byte loads, fp64 accumulation, fixed
// reduction tree.  (One wave per row with scalar loads left 961 rows on 240 workgroups: 35 us per call, 3 calls a step.)
__global__ __launch_bounds__(256) void snb_w_v_kernel(SnBatch b) {
  const int l = sn_find_layer(b, blockIdx.x);
  const int cols = b.cols[l];
  const int row = blockIdx.x - b.blk0[l];
  const float* w = b.w[l] + (size_t)row * cols; const float* v = b.v[l];
  float* t = b.work[l] + (size_t)SN_RS * cols;
  double s = 0.0;
  if ((cols & 3) == 0) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    for (int j = threadIdx.x; j < (cols >> 2); j += 256) {
      const f32x4 a = w4[j], c = v4[j];
      s += (double)a[0] * (double)c[0] + (double)a[1] * (double)c[1] + (double)a[2] * (double)c[2] + (double)a[3] * (double)c[3];
    }
  } else {
    for (int j = threadIdx.x; j < cols; j += 256) s += (double)w[j] * (double)v[j];
  }
  s = block_sum_d(s);
  if (threadIdx.x == 0) t[row] = (float)s;
}

// u = normalize(t) (training), sigma = <u, t>, inv = 1/sigma : one workgroup per layer
__global__ __launch_bounds__(1024) void snb_norm_u_sigma_kernel(SnBatch b, int do_iter) {
  const int l = blockIdx.x;
  const int n = b.rows[l];
  const float* t = b.work[l] + (size_t)SN_RS * b.cols[l];
  float* u = b.u[l];
  if (do_iter) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)t[i] * (double)t[i];
    s = block_sum_d(s);
    float nrm = fmaxf((float)sqrt(s), b.eps);
    for (int i = threadIdx.x; i < n; i += blockDim.x) u[i] = t[i] / nrm;
    __syncthreads();
  }
  double d = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) d += (double)u[i] * (double)t[i];
  d = block_sum_d(d);
  if (threadIdx.x == 0) *b.inv[l] = 1.f / (float)d;
}

__global__ void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += (double)a[i] * (double)b[i];
  s = block_sum_d(s);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// d_w_orig = (G - (<G,W_orig>/sigma) * u v^T) / sigma
__global__ void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v,
                              const float* __restrict__ inv_sigma, const double* __restrict__ gdotw, int rows, int cols,
                              float* __restrict__ d, int accumulate) {
  float is = *inv_sigma;
  float coef = (float)(*gdotw) * is;
  int64_t n = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    float val = (g[i] - coef * u[r] * v[c]) * is;
    d[i] = accumulate ? d[i] + val : val;
  }
}

// ---- Adam + EMA ------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ ema, int64_t n, float lr, float b1, float b2, float eps, int step_host,
                            const int* __restrict__ step_dev, float ema_decay) {
  // step count from device memory when given (stays correct under hipGraph replay)
  const int t = step_dev ? *step_dev : step_host;
  const float bc1 = (float)(1.0 - pow((double)b1, (double)t));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
  const float step = lr / bc1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    float mi = b1 == 0.f ? gi : m[i] + (gi - m[i]) * (1.f - b1);
    float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    float pi = p[i] - step * (mi / denom);
    p[i] = pi;
    if (ema) ema[i] = ema[i] * ema_decay + pi * (1.f - ema_decay);
  }
}

__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ o, float a,
                             const float* __restrict__ a_dev, float b, int64_t n) {
  float aa = a_dev ? a * (*a_dev) : a;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    o[i] = aa * x[i] + b * y[i];
}

inline int nblocks(int64_t n, int cap = 4096) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

struct SnJobs {
  itg_sn_job j[ITG_WGRAD_MAX_JOBS];
  long long start[ITG_WGRAD_MAX_JOBS + 1];     // first flat element of each job
  int n;
};

// dW_orig (+)= (G - <G, W> / sigma * u v^T) / sigma for every job; the dots were accumulated by itg_wgrad_reduce_multi
__global__ void sn_bwd_multi_kernel(const SnJobs jb) {
  const long long total = jb.start[jb.n];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int ji = 0;
    while (ji + 1 < jb.n && i >= jb.start[ji + 1]) ++ji;
    const itg_sn_job& J = jb.j[ji];
    const long long e = i - jb.start[ji];
    const float is = *J.inv_sigma;
    const float coef = (float)(*J.dot) * is;
    const int r = (int)(e / J.cols), c = (int)(e - (long long)r * J.cols);
    const float val = (J.g_w[e] - coef * J.u[r] * J.v[c]) * is;
    J.d_w_orig[e] = (J.accumulate & ITG_ACC_DW) ? J.d_w_orig[e] + val : val;
  }
}

}  // namespace

extern "C" {

int itg_bce_logits_fwd(const float* logits, int64_t count, float target, float* loss_out, void* stream) {
  if (!logits || !loss_out || count <= 0) return ITG_ERR_ARG;
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, count, target, loss_out);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_bce_logits_bwd(const float* logits, int64_t count, float target, const float* upstream, float* dlogits,
                       void* stream) {
  if (!logits || !dlogits || count <= 0) return ITG_ERR_ARG;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(nblocks(count)), dim3(256), 0, (hipStream_t)stream, logits, count, target,
                     upstream, dlogits);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int64_t itg_logit_loss_grid_workspace(void) { return 8 + LOSS_GRID_MAX_BLOCKS; }      // doubles, zeroed

int itg_logit_loss_grid(const itg_tensor* logits, int kind, float target, float* loss_out, itg_tensor* dlogits, void* workspace_zeroed,
                        void* stream) {
  if (!logits || !logits->ptr || !dlogits || !dlogits->ptr || !loss_out || !workspace_zeroed) return ITG_ERR_ARG;
  if (logits->c != 1 || kind < 0 || kind > 3 || logits->ld < 1) return ITG_ERR_ARG;
  if (dlogits->n != logits->n || dlogits->gh != logits->gh || dlogits->gw != logits->gw || dlogits->ph != logits->ph ||
      dlogits->pw != logits->pw || dlogits->ld != logits->ld || dlogits->c != 1)
    return ITG_ERR_ARG;
  const long long npix = (long long)logits->n * logits->gh * logits->gw * logits->ph * logits->pw;
  if (npix <= 0) return ITG_ERR_ARG;
  const int blocks = (int)std::min<long long>(LOSS_GRID_MAX_BLOCKS, (npix + 255) / 256);
  double* ws = (double*)workspace_zeroed;
  hipLaunchKernelGGL(logit_loss_grid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)logits->ptr, npix,
                     (int)logits->ld, kind, target, (float*)dlogits->ptr, ws + 8, (unsigned*)ws, loss_out);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_hinge_fwd(const float* logits, int64_t count, int mode, float* loss_out, void* stream) {
  if (!logits || !loss_out || count <= 0 || mode < 0 || mode > 2) return ITG_ERR_ARG;
  hipLaunchKernelGGL(hinge_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, count, mode, loss_out);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_hinge_bwd(const float* logits, int64_t count, int mode, const float* upstream, float* dlogits, void* stream) {
  if (!logits || !dlogits || count <= 0 || mode < 0 || mode > 2) return ITG_ERR_ARG;
  hipLaunchKernelGGL(hinge_bwd_kernel, dim3(nblocks(count)), dim3(256), 0, (hipStream_t)stream, logits, count, mode,
                     upstream, dlogits);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// workspace: 8*cols + rows floats
int itg_spectral_norm_power_iter(const float* w, float* u, float* v, int rows, int cols, int do_iter, float eps,
                                 float* sigma_out, float* inv_sigma_out, float* workspace, void* stream) {
  if (!w || !u || !v || rows <= 0 || cols <= 0 || !workspace) return ITG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* t_cols = workspace;
  float* t_rows = workspace + (size_t)SN_RS * cols;
  if (do_iter) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3((cols + 63) / 64, SN_RS), dim3(256), 0, s, w, (const float*)u, t_cols, rows,
                       cols);
    ITG_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, s, (const float*)t_cols, v, cols, SN_RS, eps);
    ITG_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sn_w_v_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, w, (const float*)v, t_rows, rows, cols);
  ITG_CHECK_LAUNCH();
  if (do_iter) {
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, s, (const float*)t_rows, u, rows, 1, eps);
    ITG_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(1024), 0, s, (const float*)u, (const float*)t_rows, rows, sigma_out,
                     inv_sigma_out);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// Same iteration for up to 8 layers at once (4 launches in total); host arrays of length n.
int itg_spectral_norm_power_iter_multi(int n, const float* const* w, float* const* u, float* const* v, const int* rows,
                                       const int* cols, int do_iter, float eps, float* const* inv_sigma_out,
                                       float* const* workspace, void* stream) {
  if (n <= 0 || n > SN_MAXL || !w || !u || !v || !rows || !cols || !inv_sigma_out || !workspace) return ITG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  SnBatch b;
  b.n = n; b.eps = eps;
  for (int l = 0; l < n; ++l) {
    if (!w[l] || !u[l] || !v[l] || !inv_sigma_out[l] || !workspace[l] || rows[l] <= 0 || cols[l] <= 0) return ITG_ERR_ARG;
    b.w[l] = w[l]; b.u[l] = u[l]; b.v[l] = v[l]; b.work[l] = workspace[l]; b.inv[l] = inv_sigma_out[l];
    b.rows[l] = rows[l]; b.cols[l] = cols[l];
  }
  if (do_iter) {
    b.blk0[0] = 0;
    for (int l = 0; l < n; ++l) b.blk0[l + 1] = b.blk0[l] + (cols[l] + 63) / 64;
    hipLaunchKernelGGL(snb_wt_u_kernel, dim3(b.blk0[n], SN_RS), dim3(256), 0, s, b);
    ITG_CHECK_LAUNCH();
    hipLaunchKernelGGL(snb_norm_v_kernel, dim3(n), dim3(1024), 0, s, b);
    ITG_CHECK_LAUNCH();
  }
  b.blk0[0] = 0;
  for (int l = 0; l < n; ++l) b.blk0[l + 1] = b.blk0[l] + rows[l];
  hipLaunchKernelGGL(snb_w_v_kernel, dim3(b.blk0[n]), dim3(256), 0, s, b);
  ITG_CHECK_LAUNCH();
  hipLaunchKernelGGL(snb_norm_u_sigma_kernel, dim3(n), dim3(1024), 0, s, b, do_iter);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// workspace: 2 floats (one fp64 accumulator, 8-byte aligned)
int itg_spectral_norm_bwd_multi(const itg_sn_job* jobs, int n, void* stream) {
  if (!jobs || n <= 0 || n > ITG_WGRAD_MAX_JOBS) return ITG_ERR_ARG;
  SnJobs jb;
  jb.n = n;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    const itg_sn_job& J = jobs[i];
    if (!J.g_w || !J.u || !J.v || !J.inv_sigma || !J.dot || !J.d_w_orig || J.rows <= 0 || J.cols <= 0) return ITG_ERR_ARG;
    jb.j[i] = J;
    jb.start[i] = total;
    total += (long long)J.rows * J.cols;
  }
  jb.start[n] = total;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sn_bwd_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jb);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_spectral_norm_bwd(const float* g_w, const float* w_orig, const float* u, const float* v,
                          const float* inv_sigma, int rows, int cols, float* d_w_orig, int accumulate, float* workspace,
                          void* stream) {
  if (!g_w || !w_orig || !u || !v || !inv_sigma || !d_w_orig || !workspace || (((uintptr_t)workspace) & 7))
    return ITG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  double* acc = reinterpret_cast<double*>(workspace);
  if (!(accumulate & ITG_WS_ZEROED) && hipMemsetAsync(acc, 0, sizeof(double), s) != hipSuccess) return ITG_ERR_LAUNCH;
  int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(dot_kernel, dim3(nblocks(n, 512)), dim3(256), 0, s, g_w, w_orig, n, acc);
  ITG_CHECK_LAUNCH();
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, s, g_w, u, v, inv_sigma, (const double*)acc, rows,
                     cols, d_w_orig, accumulate & ITG_ACC_DW);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_axpby(const float* x, const float* y, float* out, float a, const float* a_dev, float b, int64_t n,
              void* stream) {
  if (!x || !y || !out || n <= 0) return ITG_ERR_ARG;
  hipLaunchKernelGGL(axpby_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, out, a, a_dev, b, n);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_dot(const float* x, const float* y, int64_t n, double* out, void* stream) {
  if (!x || !y || !out || n <= 0) return ITG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(double), s) != hipSuccess) return ITG_ERR_LAUNCH;
  hipLaunchKernelGGL(dot_kernel, dim3(nblocks(n, 512)), dim3(256), 0, s, x, y, n, out);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, int64_t count, float lr, float beta1,
                      float beta2, float eps, int step, const int32_t* step_dev, float ema_decay, void* stream) {
  if (!p || !g || !m || !v || count <= 0 || (!step_dev && step < 1)) return ITG_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(nblocks(count)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, count, lr,
                     beta1, beta2, eps, step, (const int*)step_dev, ema_decay);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

}  // extern "C"
