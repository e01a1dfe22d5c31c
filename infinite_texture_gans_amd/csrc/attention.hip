// Per-patch SAGAN attention core (gfx950), one workgroup per patch.
// reference models/layers.py:246-258:  beta = softmax_j(theta_i . phi_j),  o_i = sum_j beta_ij g_j
// (theta: HW x C/8, phi/g: 2x2 max-pooled to HW/4 keys, g: C/2 channels).  The four 1x1
// convolutions around it run through the implicit-GEMM conv kernels; the 2x2 pools through
// itg_maxpool2_*.  Keys/values of a patch live in LDS; each thread owns one query row, the
// softmax is a two-pass (max/sum, then normalise) sweep over LDS-broadcast keys.
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
  size_t lds = (size_t)p.J * (p.ld8 + p.ld2) * sizeof(float);
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(p.NB), dim3(256), lds, (hipStream_t)stream, p);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

}  // extern "C"
