// Per-patch SAGAN attention core (gfx950), one workgroup per patch.
// reference models/layers.py:246-258:  beta = softmax_j(theta_i . phi_j),  o_i = sum_j beta_ij g_j
// (theta: HW x C/8, phi/g: 2x2 max-pooled to HW/4 keys, g: C/2 channels).  The four 1x1
// convolutions around it run through the implicit-GEMM conv kernels; the 2x2 pools through
// itg_maxpool2_*.  Keys/values of a patch live in LDS; each thread owns one query row, the
// softmax is a two-pass (max/sum, then normalise) sweep over LDS-broadcast keys.
#include <cstdlib>
#include "itg_common.h"

namespace {

struct AttP {
  const float* theta; const float* phi; const float* g;
  float* o; float* beta;   // beta: [NB][HW][J] (fwd) followed by dS [NB][HW][J] (bwd scratch)
  int NB, HW, J, c8, ld8, c2, ld2;
};

__global__ __launch_bounds__(256) void attention_fwd_kernel(AttP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sphi = lds;                    // [J][ld8]
  float* sg = lds + p.J * p.ld8;        // [J][ld2]
  const int b = blockIdx.x;
  const float* phi = p.phi + (size_t)b * p.J * p.ld8;
  const float* g = p.g + (size_t)b * p.J * p.ld2;
  for (int i = threadIdx.x; i < p.J * p.ld8; i += blockDim.x) sphi[i] = phi[i];
  for (int i = threadIdx.x; i < p.J * p.ld2; i += blockDim.x) sg[i] = g[i];
  __syncthreads();
  for (int q = threadIdx.x; q < p.HW; q += blockDim.x) {
    const float* th = p.theta + ((size_t)b * p.HW + q) * p.ld8;
    float* brow = p.beta + ((size_t)b * p.HW + q) * p.J;
    float mx = -INFINITY;
    for (int j = 0; j < p.J; ++j) {
      float s = 0.f;
      for (int c = 0; c < p.c8; ++c) s = fmaf(th[c], sphi[j * p.ld8 + c], s);
      brow[j] = s;
      mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int j = 0; j < p.J; ++j) { float e = expf(brow[j] - mx); brow[j] = e; sum += e; }
    float inv = 1.f / sum;
    for (int j = 0; j < p.J; ++j) brow[j] *= inv;
    float* orow = p.o + ((size_t)b * p.HW + q) * p.ld2;
    for (int c0 = 0; c0 < p.ld2; c0 += 32) {
      float acc[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) acc[c] = 0.f;
      for (int j = 0; j < p.J; ++j) {
        float bj = brow[j];
#pragma unroll
        for (int c = 0; c < 32; ++c)
          if (c0 + c < p.ld2) acc[c] = fmaf(bj, sg[j * p.ld2 + c0 + c], acc[c]);
      }
#pragma unroll
      for (int c = 0; c < 32; ++c)
        if (c0 + c < p.ld2) orow[c0 + c] = (c0 + c < p.c2) ? acc[c] : 0.f;
    }
  }
}

struct AttBP {
  const float* theta; const float* phi; const float* g; const float* beta; float* dS;
  const float* d_o; float* d_theta; float* d_phi; float* d_g;
  int NB, HW, J, c8, ld8, c2, ld2;
};

__global__ __launch_bounds__(256) void attention_bwd_kernel(AttBP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sphi = lds;
  float* sg = lds + p.J * p.ld8;
  const int b = blockIdx.x;
  const float* phi = p.phi + (size_t)b * p.J * p.ld8;
  const float* g = p.g + (size_t)b * p.J * p.ld2;
  for (int i = threadIdx.x; i < p.J * p.ld8; i += blockDim.x) sphi[i] = phi[i];
  for (int i = threadIdx.x; i < p.J * p.ld2; i += blockDim.x) sg[i] = g[i];
  __syncthreads();
  // phase A: one query per thread -> dS row, d_theta row
  for (int q = threadIdx.x; q < p.HW; q += blockDim.x) {
    const float* dorow = p.d_o + ((size_t)b * p.HW + q) * p.ld2;
    const float* brow = p.beta + ((size_t)b * p.HW + q) * p.J;
    float* srow = p.dS + ((size_t)b * p.HW + q) * p.J;
    float delta = 0.f;
    for (int j = 0; j < p.J; ++j) {
      float db = 0.f;
      for (int c = 0; c < p.c2; ++c) db = fmaf(dorow[c], sg[j * p.ld2 + c], db);
      srow[j] = db;
      delta = fmaf(brow[j], db, delta);
    }
    for (int j = 0; j < p.J; ++j) srow[j] = brow[j] * (srow[j] - delta);
    float* dth = p.d_theta + ((size_t)b * p.HW + q) * p.ld8;
    for (int c = 0; c < p.ld8; ++c) {
      float s = 0.f;
      if (c < p.c8)
        for (int j = 0; j < p.J; ++j) s = fmaf(srow[j], sphi[j * p.ld8 + c], s);
      dth[c] = s;
    }
  }
  __threadfence_block();
  __syncthreads();
  // phase B: one (key, channel) per thread, reduction over queries (deterministic)
  for (int e = threadIdx.x; e < p.J * p.ld2; e += blockDim.x) {
    int j = e / p.ld2, c = e - j * p.ld2;
    float s = 0.f;
    if (c < p.c2)
      for (int q = 0; q < p.HW; ++q)
        s = fmaf(p.beta[((size_t)b * p.HW + q) * p.J + j], p.d_o[((size_t)b * p.HW + q) * p.ld2 + c], s);
    p.d_g[(size_t)b * p.J * p.ld2 + e] = s;
  }
  for (int e = threadIdx.x; e < p.J * p.ld8; e += blockDim.x) {
    int j = e / p.ld8, c = e - j * p.ld8;
    float s = 0.f;
    if (c < p.c8)
      for (int q = 0; q < p.HW; ++q)
        s = fmaf(p.dS[((size_t)b * p.HW + q) * p.J + j], p.theta[((size_t)b * p.HW + q) * p.ld8 + c], s);
    p.d_phi[(size_t)b * p.J * p.ld8 + e] = s;
  }
}


// ------------------------------------------------------------------------------- LDS-tiled kernels (J <= 64 keys)
// The kernels above keep score rows in global memory and give one thread a whole query (forward) or a whole
// (key, channel) reduction over all queries (backward): 0.28 ms / 1.05 ms per call at BASELINE config 3's shape
// (72 patches, 256 queries x 64 keys, 13 / 52 channels) - a fifth of that configuration's step.  Here a workgroup
// stages keys, values and a chunk of 64 queries in LDS, the 64 x J score tile lives in LDS (pitch J + 1: conflict-free
// by row and by column) and every phase is spread over all 256 threads.  Forward: one workgroup per (patch, 64-query
// chunk) in both directions; the backward workgroups write their chunk's share of dK / dV to a scratch slab and a small
// second launch adds the shares in chunk order - a fixed summation order, so the result is deterministic.
constexpr int QC = 64;

__device__ __forceinline__ void stage_rows(float* dst, const float* src, int rows, int ld, int c) {
  const int q4 = ld >> 2;
  for (int e = threadIdx.x; e < rows * q4; e += blockDim.x) {
    const int c4 = e % q4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)e * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c4 * 4 + k >= c) v[k] = 0.f;                   // pad lanes never carry data
    *reinterpret_cast<f32x4*>(dst + e * 4) = v;
  }
}

__global__ __launch_bounds__(256) void attention_fwd_tiled_kernel(AttP p, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int J = p.J, SP = J + 1;
  float* sphi = lds;                       // [J][ld8]
  float* sg = sphi + J * p.ld8;            // [J][ld2]
  float* sth = sg + J * p.ld2;             // [QC][ld8]
  float* sS = sth + QC * p.ld8;            // [QC][J + 1]
  const int b = blockIdx.x / nchunk, q0 = (blockIdx.x - b * nchunk) * QC;
  const int nq = min(QC, p.HW - q0);
  const int tid = threadIdx.x;
  stage_rows(sphi, p.phi + (size_t)b * J * p.ld8, J, p.ld8, p.c8);
  stage_rows(sg, p.g + (size_t)b * J * p.ld2, J, p.ld2, p.c2);
  stage_rows(sth, p.theta + ((size_t)b * p.HW + q0) * p.ld8, nq, p.ld8, p.c8);
  __syncthreads();
  const int q8 = p.ld8 >> 2;
  for (int e = tid; e < nq * J; e += 256) {              // scores
    const int q = e / J, j = e - q * J;
    float s = 0.f;
    for (int c4 = 0; c4 < q8; ++c4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(sth + q * p.ld8 + c4 * 4);
      const f32x4 k = *reinterpret_cast<const f32x4*>(sphi + j * p.ld8 + c4 * 4);
      s = fmaf(a[0], k[0], s); s = fmaf(a[1], k[1], s); s = fmaf(a[2], k[2], s); s = fmaf(a[3], k[3], s);
    }
    sS[q * SP + j] = s;
  }
  __syncthreads();
  if (tid < nq) {                                          // softmax of one row per thread (rows are conflict-free)
    float* row = sS + tid * SP;
    float mx = -INFINITY;
    for (int j = 0; j < J; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < J; ++j) { const float e_ = expf(row[j] - mx); row[j] = e_; sum += e_; }
    const float inv = 1.f / sum;
    for (int j = 0; j < J; ++j) row[j] *= inv;
  }
  __syncthreads();
  float* bout = p.beta + ((size_t)b * p.HW + q0) * J;     // saved for the backward pass (coalesced rows)
  for (int e = tid; e < nq * J; e += 256) bout[e] = sS[(e / J) * SP + (e % J)];
  const int q2 = p.ld2 >> 2;
  for (int e = tid; e < nq * q2; e += 256) {              // O = beta . g
    const int q = e / q2, c4 = e - q * q2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* row = sS + q * SP;
    for (int j = 0; j < J; ++j) acc += row[j] * *reinterpret_cast<const f32x4*>(sg + j * p.ld2 + c4 * 4);
    *reinterpret_cast<f32x4*>(p.o + ((size_t)b * p.HW + q0 + q) * p.ld2 + c4 * 4) = acc;
  }
}

__global__ __launch_bounds__(256) void attention_bwd_tiled_kernel(AttBP p, int nchunk, float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int J = p.J, SP = J + 1;
  float* sphi = lds;                       // [J][ld8]
  float* sg = sphi + J * p.ld8;            // [J][ld2]
  float* sth = sg + J * p.ld2;             // [QC][ld8]
  float* sdo = sth + QC * p.ld8;           // [QC][ld2]
  float* sB = sdo + QC * p.ld2;            // [QC][J + 1]  beta
  float* sD = sB + QC * SP;                // [QC][J + 1]  d beta, then dS
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk, tid = threadIdx.x;
  const int q0 = chunk * QC, nq = min(QC, p.HW - q0);
  const int q8 = p.ld8 >> 2, q2 = p.ld2 >> 2;
  stage_rows(sphi, p.phi + (size_t)b * J * p.ld8, J, p.ld8, p.c8);
  stage_rows(sg, p.g + (size_t)b * J * p.ld2, J, p.ld2, p.c2);
  stage_rows(sth, p.theta + ((size_t)b * p.HW + q0) * p.ld8, nq, p.ld8, p.c8);
  stage_rows(sdo, p.d_o + ((size_t)b * p.HW + q0) * p.ld2, nq, p.ld2, p.c2);
  const float* bin = p.beta + ((size_t)b * p.HW + q0) * J;
  for (int e = tid; e < nq * J; e += 256) sB[(e / J) * SP + (e % J)] = bin[e];
  __syncthreads();
  // this chunk's share of d_g[j][c] = sum_q beta[q][j] dO[q][c] and (below) of d_phi -> slab[b][chunk][J*ld2 | J*ld8]
  float* myslab = slab + (size_t)blockIdx.x * J * (p.ld2 + p.ld8);
  for (int e = tid; e < J * q2; e += 256) {
    const int j = e / q2, c4 = e - j * q2;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nq; ++q) a += sB[q * SP + j] * *reinterpret_cast<const f32x4*>(sdo + q * p.ld2 + c4 * 4);
    *reinterpret_cast<f32x4*>(myslab + (size_t)e * 4) = a;
  }
  // d beta[q][j] = dO[q] . g[j]
  for (int e = tid; e < nq * J; e += 256) {
    const int q = e / J, j = e - q * J;
    float s = 0.f;
    for (int c4 = 0; c4 < q2; ++c4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(sdo + q * p.ld2 + c4 * 4);
      const f32x4 k = *reinterpret_cast<const f32x4*>(sg + j * p.ld2 + c4 * 4);
      s = fmaf(a[0], k[0], s); s = fmaf(a[1], k[1], s); s = fmaf(a[2], k[2], s); s = fmaf(a[3], k[3], s);
    }
    sD[q * SP + j] = s;
  }
  __syncthreads();
  if (tid < nq) {                                          // dS = beta (d beta - sum_j beta d beta), one row per thread
    float* dr = sD + tid * SP;
    const float* br = sB + tid * SP;
    float delta = 0.f;
    for (int j = 0; j < J; ++j) delta = fmaf(br[j], dr[j], delta);
    for (int j = 0; j < J; ++j) dr[j] = br[j] * (dr[j] - delta);
  }
  __syncthreads();
  // d_theta[q][c] = sum_j dS[q][j] phi[j][c]
  for (int e = tid; e < nq * q8; e += 256) {
    const int q = e / q8, c4 = e - q * q8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* dr = sD + q * SP;
    for (int j = 0; j < J; ++j) acc += dr[j] * *reinterpret_cast<const f32x4*>(sphi + j * p.ld8 + c4 * 4);
    *reinterpret_cast<f32x4*>(p.d_theta + ((size_t)b * p.HW + q0 + q) * p.ld8 + c4 * 4) = acc;
  }
  // this chunk's share of d_phi[j][c] = sum_q dS[q][j] theta[q][c]
  float* pslab = myslab + (size_t)J * p.ld2;
  for (int e = tid; e < J * q8; e += 256) {
    const int j = e / q8, c4 = e - j * q8;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nq; ++q) a += sD[q * SP + j] * *reinterpret_cast<const f32x4*>(sth + q * p.ld8 + c4 * 4);
    *reinterpret_cast<f32x4*>(pslab + (size_t)e * 4) = a;
  }
}

// d_g / d_phi = the chunks' partial sums added in chunk order (deterministic)
__global__ void attention_bwd_reduce_kernel(const float* __restrict__ slab, float* __restrict__ d_g, float* __restrict__ d_phi,
                                            int NB, int nchunk, int n2, int n8) {
  const int per = n2 + n8;
  const int64_t total = (int64_t)NB * (per >> 2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / (per >> 2)), e = (int)(i - (int64_t)b * (per >> 2)) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunk; ++c) v += *reinterpret_cast<const f32x4*>(slab + ((size_t)b * nchunk + c) * per + e);
    float* dst = e < n2 ? d_g + (size_t)b * n2 + e : d_phi + (size_t)b * n8 + (e - n2);
    *reinterpret_cast<f32x4*>(dst) = v;
  }
}

// shapes the tiled kernels cover (everything the generator produces with base_res <= 4)
inline bool att_tiled_ok(int J, int ld8, int ld2) {
  static const char* km = getenv("ITG_KERNEL_MASK");      // bit 7 (conv_common.h: kernel_on)
  if (km && !((strtol(km, nullptr, 0) >> 7) & 1)) return false;
  return J <= 64 && ld2 <= 128 && ld8 <= 32;
}

}  // namespace

extern "C" {

static int att_check(const itg_tensor* theta, const itg_tensor* phi, const itg_tensor* g, const itg_tensor* o) {
  int rc;
  if ((rc = check_tensor(theta)) || (rc = check_tensor(phi)) || (rc = check_tensor(g)) || (rc = check_tensor(o)))
    return rc;
  int nb = theta->n * theta->gh * theta->gw;
  if (phi->n * phi->gh * phi->gw != nb || g->n * g->gh * g->gw != nb || o->n * o->gh * o->gw != nb) return ITG_ERR_ARG;
  if (phi->ph * phi->pw != g->ph * g->pw || o->ph * o->pw != theta->ph * theta->pw) return ITG_ERR_ARG;
  if (phi->c != theta->c || phi->ld != theta->ld || o->c != g->c || o->ld != g->ld) return ITG_ERR_ARG;
  size_t lds = (size_t)phi->ph * phi->pw * (phi->ld + g->ld) * sizeof(float);
  if (lds > 64 * 1024) return ITG_ERR_ARG;
  return ITG_OK;
}

int64_t itg_attention_scratch_floats(const itg_tensor* theta, const itg_tensor* phi_pooled, const itg_tensor* g_pooled) {
  if (!theta || !phi_pooled || !g_pooled) return ITG_ERR_ARG;
  const int64_t nb = (int64_t)theta->n * theta->gh * theta->gw, hw = (int64_t)theta->ph * theta->pw;
  const int64_t j = (int64_t)phi_pooled->ph * phi_pooled->pw;
  const int64_t beta = nb * hw * j;                                        // the saved softmax
  const int64_t ds = nb * hw * j;                                          // generic backward: dS
  const int64_t slabs = nb * ((hw + QC - 1) / QC) * j * (phi_pooled->ld + g_pooled->ld);   // tiled backward: dK / dV shares
  return beta + (ds > slabs ? ds : slabs);
}

int itg_attention_fwd(const itg_tensor* theta, const itg_tensor* phi_pooled, const itg_tensor* g_pooled,
                      const itg_tensor* o_mid, float* beta_save, void* stream) {
  int rc = att_check(theta, phi_pooled, g_pooled, o_mid);
  if (rc) return rc;
  if (!beta_save) return ITG_ERR_ARG;
  AttP p;
  p.theta = (const float*)theta->ptr; p.phi = (const float*)phi_pooled->ptr; p.g = (const float*)g_pooled->ptr;
  p.o = (float*)o_mid->ptr; p.beta = beta_save;
  p.NB = theta->n * theta->gh * theta->gw; p.HW = theta->ph * theta->pw; p.J = phi_pooled->ph * phi_pooled->pw;
  p.c8 = theta->c; p.ld8 = theta->ld; p.c2 = g_pooled->c; p.ld2 = g_pooled->ld;
  if (att_tiled_ok(p.J, p.ld8, p.ld2)) {
    const int nchunk = (p.HW + QC - 1) / QC;
    size_t lt = ((size_t)p.J * (p.ld8 + p.ld2) + (size_t)QC * p.ld8 + (size_t)QC * (p.J + 1)) * sizeof(float);
    hipLaunchKernelGGL(attention_fwd_tiled_kernel, dim3(p.NB * nchunk), dim3(256), lt, (hipStream_t)stream, p, nchunk);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  size_t lds = (size_t)p.J * (p.ld8 + p.ld2) * sizeof(float);
  hipLaunchKernelGGL(attention_fwd_kernel, dim3(p.NB), dim3(256), lds, (hipStream_t)stream, p);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_attention_bwd(const itg_tensor* theta, const itg_tensor* phi_pooled, const itg_tensor* g_pooled,
                      float* beta_save, const itg_tensor* d_o_mid, const itg_tensor* d_theta,
                      const itg_tensor* d_phi_pooled, const itg_tensor* d_g_pooled, void* stream) {
  int rc = att_check(theta, phi_pooled, g_pooled, d_o_mid);
  if (rc) return rc;
  if (!beta_save || (rc = check_tensor(d_theta)) || (rc = check_tensor(d_phi_pooled)) || (rc = check_tensor(d_g_pooled)))
    return rc ? rc : ITG_ERR_ARG;
  if (!same_shape(theta, d_theta) || !same_shape(phi_pooled, d_phi_pooled) || !same_shape(g_pooled, d_g_pooled))
    return ITG_ERR_ARG;
  AttBP p;
  p.theta = (const float*)theta->ptr; p.phi = (const float*)phi_pooled->ptr; p.g = (const float*)g_pooled->ptr;
  p.NB = theta->n * theta->gh * theta->gw; p.HW = theta->ph * theta->pw; p.J = phi_pooled->ph * phi_pooled->pw;
  p.beta = beta_save;
  p.dS = beta_save + (size_t)p.NB * p.HW * p.J;
  p.d_o = (const float*)d_o_mid->ptr; p.d_theta = (float*)d_theta->ptr; p.d_phi = (float*)d_phi_pooled->ptr;
  p.d_g = (float*)d_g_pooled->ptr;
  p.c8 = theta->c; p.ld8 = theta->ld; p.c2 = g_pooled->c; p.ld2 = g_pooled->ld;
  if (att_tiled_ok(p.J, p.ld8, p.ld2)) {
    const int nchunk = (p.HW + QC - 1) / QC;
    size_t lt = ((size_t)p.J * (p.ld8 + p.ld2) + (size_t)QC * (p.ld8 + p.ld2) + (size_t)2 * QC * (p.J + 1)) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_tiled_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_done = true;
    }
    float* slab = p.dS;                                    // second part of beta_save (itg_attention_scratch_floats)
    hipLaunchKernelGGL(attention_bwd_tiled_kernel, dim3(p.NB * nchunk), dim3(256), lt, (hipStream_t)stream, p, nchunk, slab);
    ITG_CHECK_LAUNCH();
    const int n2 = p.J * p.ld2, n8 = p.J * p.ld8;
    const int64_t tot = (int64_t)p.NB * ((n2 + n8) >> 2);
    hipLaunchKernelGGL(attention_bwd_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)slab, p.d_g, p.d_phi, p.NB, nchunk, n2, n8);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  size_t lds = (size_t)p.J * (p.ld8 + p.ld2) * sizeof(float);
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(p.NB), dim3(256), lds, (hipStream_t)stream, p);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

}  // extern "C"
