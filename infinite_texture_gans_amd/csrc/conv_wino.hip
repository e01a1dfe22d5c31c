// Winograd F(4 x 4, 4 x 4) for wide 4 x 4 stride-1 convolutions on plain images (gfx950).
//
// The discriminator's 256 -> 512 layer (reference models/discriminators.py:196-206: conv4x4, stride 1, padding 1 on the
// 48 x 48 maps) is the one layer of the model that is MFMA-bound AND has a fast algorithm: 49 multiplications per 4 x 4
// output tile instead of 256.  y = A^T [ (G g G^T) .* (B^T d B) ] A per (output channel, input channel) pair, summed over
// input channels in the transformed domain - i.e. 49 independent GEMMs [tiles x Cin] x [Cin x Cout]:
//   wino_in_kernel    d (7 x 7 input tiles, stride 4, zero padding)  ->  V[xi][tile][ci]        (B^T d B per channel)
//   conv_nt_kernel    49 uniform classes of a 1 x 1 convolution        ->  M[xi][tile][co]        (ConvP.ucls)
//   wino_out_kernel   A^T m A per tile and channel, 1/sigma, bias, activation (or the producing layer's activation
//                     derivative for an input gradient), cropped to the output extent
// The panel U[xi][co][ci] = G g G^T comes from itg_pack_wino_* (conv.hip).  The input gradient of the layer is the same
// pipeline on dy with the flipped, transposed filter and padding 2.  fp32 throughout; measured error 4.6e-6 rel-L2 against
// fp64 (direct fp32: 3e-7), tools/gen_winograd.py has the matrices.
#include <algorithm>
#include <cstring>
#include "conv_common.h"
#include "winograd_f44.h"

namespace itgk {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// one thread = (tile, channel pair): 7 x 7 loads of 8 bytes, V = B^T d B, 49 coalesced 8-byte stores
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x, int n, int H, int W, int ld, int pad, int T,
                                                       float* __restrict__ V) {
  const int cpairs = ld >> 1;
  const int64_t tiles = (int64_t)n * T * T;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= tiles * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int64_t tile = gt / cpairs;
  const int tx = (int)(tile % T), ty = (int)((tile / T) % T), img = (int)(tile / ((int64_t)T * T));
  const int y0 = 4 * ty - pad, x0 = 4 * tx - pad;
  f32x2 tmp[7][7];                       // tmp[i][b] = sum_j BT[b][j] d[i][j]
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    f32x2 d[7];
    const int iy = y0 + i;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int ix = x0 + j;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      d[j] = ok ? *reinterpret_cast<const f32x2*>(x + (((size_t)img * H + iy) * W + ix) * ld + 2 * cp) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (WINO_BT[b][j] != 0.f) a += WINO_BT[b][j] * d[j];
      tmp[i][b] = a;
    }
  }
  const size_t plane = (size_t)tiles * ld;
  float* const vb = V + (size_t)tile * ld + 2 * cp;
#pragma unroll
  for (int a = 0; a < 7; ++a)
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 7; ++i)
        if (WINO_BT[a][i] != 0.f) v += WINO_BT[a][i] * tmp[i][b];
      *reinterpret_cast<f32x2*>(vb + (size_t)(a * 7 + b) * plane) = v;
    }
}

// one thread = (tile, output-channel pair): 49 loads, Y = A^T m A (4 x 4), epilogue, <= 16 stores
// res_mode 0: y += res;  ITG_ACT_*: y *= act'(res) (res = the activation's OUTPUT, as itg_conv2d_dgrad's act_out)
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ Mm, int n, int Ho, int Wo, int c, int ld, int T,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const float* __restrict__ res, int res_mode, float res_slope, int act,
                                                        float slope, float* __restrict__ y) {
  const int cpairs = ld >> 1;
  const int64_t tiles = (int64_t)n * T * T;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= tiles * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int64_t tile = gt / cpairs;
  const int tx = (int)(tile % T), ty = (int)((tile / T) % T), img = (int)(tile / ((int64_t)T * T));
  const size_t plane = (size_t)tiles * ld;
  const float* const mb = Mm + (size_t)tile * ld + 2 * cp;
  f32x2 tmp[7][4];                        // tmp[a][l] = sum_b m[a][b] AT[l][b]
#pragma unroll
  for (int a = 0; a < 7; ++a) {
    f32x2 m[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) m[b] = *reinterpret_cast<const f32x2*>(mb + (size_t)(a * 7 + b) * plane);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 7; ++b)
        if (WINO_AT[l][b] != 0.f) v += WINO_AT[l][b] * m[b];
      tmp[a][l] = v;
    }
  }
  const float osc = scale ? *scale : 1.f;
  f32x2 bv = {0.f, 0.f};
  if (bias) {
    if (2 * cp < c) bv[0] = bias[2 * cp];
    if (2 * cp + 1 < c) bv[1] = bias[2 * cp + 1];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oy = 4 * ty + k;
    if (oy >= Ho) continue;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int ox = 4 * tx + l;
      if (ox >= Wo) continue;
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 7; ++a)
        if (WINO_AT[k][a] != 0.f) v += WINO_AT[k][a] * tmp[a][l];
      v = v * osc + bv;
      const size_t off = (((size_t)img * Ho + oy) * Wo + ox) * ld + 2 * cp;
      if (res) {
        const f32x2 r = *reinterpret_cast<const f32x2*>(res + off);
        if (res_mode == 0) v += r;
        else {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            v[e] *= res_mode == ITG_ACT_LRELU ? (r[e] > 0.f ? 1.f : res_slope) : (res_mode == ITG_ACT_TANH ? 1.f - r[e] * r[e] : 1.f);
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        v[e] = act_apply(v[e], act, slope);
        if (2 * cp + e >= c) v[e] = 0.f;
      }
      *reinterpret_cast<f32x2*>(y + off) = v;
    }
  }
}

// ---- weight gradient: dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G per (output channel, input channel) pair
// one thread = (tile, output-channel pair): the 4 x 4 tile of dy (zero past the output extent) -> dM = A dY A^T, 49 stores
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int n, int Ho, int Wo, int ld, int T,
                                                       float* __restrict__ dM) {
  const int cpairs = ld >> 1;
  const int64_t tiles = (int64_t)n * T * T;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= tiles * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int64_t tile = gt / cpairs;
  const int tx = (int)(tile % T), ty = (int)((tile / T) % T), img = (int)(tile / ((int64_t)T * T));
  f32x2 tmp[4][7];                        // tmp[k][b] = sum_l dY[k][l] AT[l][b]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x2 d[4];
    const int oy = 4 * ty + k;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int ox = 4 * tx + l;
      d[l] = (oy < Ho && ox < Wo) ? *reinterpret_cast<const f32x2*>(dy + (((size_t)img * Ho + oy) * Wo + ox) * ld + 2 * cp)
                                  : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int l = 0; l < 4; ++l)
        if (WINO_AT[l][b] != 0.f) v += WINO_AT[l][b] * d[l];
      tmp[k][b] = v;
    }
  }
  const size_t plane = (size_t)tiles * ld;
  float* const mb = dM + (size_t)tile * ld + 2 * cp;
#pragma unroll
  for (int a = 0; a < 7; ++a)
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (WINO_AT[k][a] != 0.f) v += WINO_AT[k][a] * tmp[k][b];
      *reinterpret_cast<f32x2*>(mb + (size_t)(a * 7 + b) * plane) = v;
    }
}

// bias-gradient partials: the interpolation point 1 has A row (1, 1, 1, 1), so class (1, 1) of dM holds every tile's sum of
// dy; dbslab[r][c] = sum of the r-th run of tiles, in a fixed order (the generic reduce stage adds the R rows)
static_assert(WINO_AT[0][1] == 1.f && WINO_AT[1][1] == 1.f && WINO_AT[2][1] == 1.f && WINO_AT[3][1] == 1.f, "point 1 is column 1");
__global__ __launch_bounds__(256) void wino_db_kernel(const float* __restrict__ dM8, int64_t tiles, int ld, int co_rows, int R,
                                                       float* __restrict__ dbslab) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= co_rows) return;
  const int64_t per = (tiles + R - 1) / R, t0 = r * per, t1 = t0 + per < tiles ? t0 + per : tiles;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < ld) {
    int64_t t = t0;
    for (; t + 4 <= t1; t += 4) {
      s0 += dM8[t * ld + c]; s1 += dM8[(t + 1) * ld + c]; s2 += dM8[(t + 2) * ld + c]; s3 += dM8[(t + 3) * ld + c];
    }
    for (; t < t1; ++t) s0 += dM8[t * ld + c];
  }
  dbslab[(size_t)r * co_rows + c] = (s0 + s1) + (s2 + s3);
}

// one thread = (co, ci): dU[z][xi][co][ci] summed over the splits z in order, dg = G^T dU G, written as the generic slab
// [co][(i * 4 + j) * ci_ld + ci] (wgrad_reduce_kernel's input)
__global__ __launch_bounds__(256) void wino_wg_out_kernel(const float* __restrict__ dU, int splits, int co_rows, int Kp, int ci_ld,
                                                           float* __restrict__ slab) {
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= (int64_t)co_rows * ci_ld) return;
  const int ci = (int)(gt % ci_ld), co = (int)(gt / ci_ld);
  const size_t plane = (size_t)co_rows * Kp;
  const float* src = dU + (size_t)co * Kp + ci;
  float tmp[7][4];                        // tmp[a][j] = sum_b dU[a][b] G[b][j]
#pragma unroll
  for (int a = 0; a < 7; ++a) {
    float u[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      float v = 0.f;
      for (int z = 0; z < splits; ++z) v += src[((size_t)z * 49 + a * 7 + b) * plane];
      u[b] = v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
#pragma unroll
      for (int b = 0; b < 7; ++b)
        if (WINO_G[b][j] != 0.f) v += WINO_G[b][j] * u[b];
      tmp[a][j] = v;
    }
  }
  float* dst = slab + (size_t)co * (16 * ci_ld) + ci;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
#pragma unroll
      for (int a = 0; a < 7; ++a)
        if (WINO_G[a][i] != 0.f) v += WINO_G[a][i] * tmp[a][j];
      dst[(size_t)(i * 4 + j) * ci_ld] = v;
    }
}

}  // namespace

// workspace: V[49][tiles][in.ld] | M[49][tiles][out.ld]
int64_t wino_workspace_floats(const itg_tensor* in, const itg_tensor* out) {
  const int Ho = out->gh * out->ph, Wo = out->gw * out->pw;
  const int T = (std::max(Ho, Wo) + 3) / 4;
  const int64_t tiles = (int64_t)in->n * T * T;
  return 49 * tiles * ((int64_t)in->ld + out->ld);
}

// in / out / res: plain images (1 x 1 grids); pad: zero padding of the 4 x 4 stride-1 correlation in -> out
int wino_conv(const itg_tensor* in, const float* u_panel, const float* bias, const float* out_scale, const itg_tensor* res, int res_mode,
              float res_slope, const itg_tensor* out, int pad, int act, float slope, int prec, float* workspace,
              int64_t workspace_floats, hipStream_t s) {
  const int H = in->gh * in->ph, W = in->gw * in->pw, Ho = out->gh * out->ph, Wo = out->gw * out->pw;
  if (in->gh != 1 || in->gw != 1 || out->gh != 1 || out->gw != 1 || in->n != out->n) return ITG_ERR_ARG;
  if (Ho != H + 2 * pad - 3 || Wo != W + 2 * pad - 3 || (in->ld & 15) || (out->ld & 3)) return ITG_ERR_ARG;
  if (res && (res->n != out->n || res->gh != 1 || res->gw != 1 || res->ph != out->ph || res->pw != out->pw || res->ld != out->ld))
    return ITG_ERR_ARG;
  const int T = (std::max(Ho, Wo) + 3) / 4;
  const int64_t tiles = (int64_t)in->n * T * T;
  const int64_t vf = 49 * tiles * in->ld, mf = 49 * tiles * out->ld;
  if (!workspace || workspace_floats < vf + mf) return ITG_ERR_WORKSPACE;
  if (tiles * std::max(in->ld, out->ld) * 4 >= 0xFFFF0000LL) return ITG_ERR_ARG;
  float* V = workspace;
  float* Mm = workspace + vf;
  {
    const int64_t th = tiles * (in->ld >> 1);
    hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, (const float*)in->ptr, in->n, H, W,
                       in->ld, pad, T, V);
    ITG_CHECK_LAUNCH();
  }
  {
    // the 49 GEMMs: a 1 x 1 convolution over the tile "images" [n][T][T][ci] with 49 uniform classes
    ConvP p;
    memset(&p, 0, sizeof(p));
    itg_tensor vin = {V, in->n, 1, 1, T, T, in->c, in->ld};
    itg_tensor vout = {Mm, in->n, 1, 1, T, T, out->c, out->ld};
    p.in = make_grid(&vin); p.out = make_grid(&vout); p.res = null_grid(); p.bnx = null_grid();
    p.w = u_panel; p.bias = nullptr; p.scale = nullptr;
    p.ntaps = 1; p.kw = 1; p.cin_ld = in->ld; p.Kpad = round_up(in->ld, BK);
    p.MT = T; p.MU = T; p.M = (int)tiles;
    p.isy = p.isx = 1; p.osy = p.osx = 1;
    p.pad_mode = ITG_PAD_ZERO; p.act = ITG_ACT_NONE;
    p.co_rows = round_up(out->c, 16);
    p.prec = prec;
    p.ncls = 1;
    p.ucls = 49;
    p.u_in = (unsigned)(tiles * in->ld); p.u_out = (unsigned)(tiles * out->ld); p.u_w = (unsigned)((size_t)p.co_rows * p.Kpad);
    int rc = dispatch_nt(p, nullptr, 0, s);
    if (rc) return rc;
  }
  {
    const int64_t th = tiles * (out->ld >> 1);
    hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, (const float*)Mm, out->n, Ho, Wo, out->c,
                       out->ld, T, bias, out_scale, res ? (const float*)res->ptr : nullptr, res_mode, res_slope, act, slope,
                       (float*)out->ptr);
    ITG_CHECK_LAUNCH();
  }
  return ITG_OK;
}

// ---- weight gradient (x, dy plain images; 4 x 4 stride-1 correlation with zero padding `pad`)
// workspace: V[49][tiles][x.ld] | dM[49][tiles][dy.ld] | dU[splits][49][co_rows][Kp] | slab[co_rows][16 * x.ld] | db[R][co_rows]
WinoWgPlan plan_wino_wgrad(const itg_tensor* x, const itg_tensor* dy) {
  WinoWgPlan w;
  const int Ho = dy->gh * dy->ph, Wo = dy->gw * dy->pw;
  w.T = (std::max(Ho, Wo) + 3) / 4;
  w.tiles = (int64_t)x->n * w.T * w.T;
  w.tn = plan_tn(w.tiles, dy->ld, x->ld, ITG_PREC_F32, 49);
  w.co_rows = w.tn.co_rows;
  w.Kpad = 16 * x->ld;
  w.R = (int)std::min<int64_t>(16, std::max<int64_t>(1, w.tiles / 32));
  w.v_off = 0;
  w.m_off = w.v_off + 49 * w.tiles * x->ld;
  w.u_off = w.m_off + 49 * w.tiles * dy->ld;
  w.slab_off = w.u_off + w.tn.slab_floats;
  w.db_off = w.slab_off + (int64_t)w.co_rows * w.Kpad;
  w.ws_floats = w.db_off + (int64_t)w.R * w.co_rows;
  return w;
}

int wino_wgrad_slabs(const itg_tensor* x, const itg_tensor* dy, int pad, int prec, const WinoWgPlan& w, float* workspace,
                     bool want_db, hipStream_t s) {
  const int H = x->gh * x->ph, W = x->gw * x->pw, Ho = dy->gh * dy->ph, Wo = dy->gw * dy->pw;
  if (x->gh != 1 || x->gw != 1 || dy->gh != 1 || dy->gw != 1 || x->n != dy->n) return ITG_ERR_ARG;
  if (Ho != H + 2 * pad - 3 || Wo != W + 2 * pad - 3 || (x->ld & 15) || (dy->ld & 15) || prec != ITG_PREC_F32) return ITG_ERR_ARG;
  if (w.tiles * std::max(x->ld, dy->ld) * 4 >= 0xFFFF0000LL || w.tn.ngroups > 0) return ITG_ERR_ARG;
  float* V = workspace + w.v_off;
  float* dM = workspace + w.m_off;
  float* dU = workspace + w.u_off;
  {
    const int64_t th = w.tiles * (x->ld >> 1);
    hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, (const float*)x->ptr, x->n, H, W, x->ld,
                       pad, w.T, V);
    ITG_CHECK_LAUNCH();
  }
  {
    const int64_t th = w.tiles * (dy->ld >> 1);
    hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, (const float*)dy->ptr, dy->n, Ho, Wo,
                       dy->ld, w.T, dM);
    ITG_CHECK_LAUNCH();
  }
  if (want_db) {
    hipLaunchKernelGGL(wino_db_kernel, dim3((unsigned)((w.co_rows + 255) / 256), (unsigned)w.R), dim3(256), 0, s,
                       (const float*)(dM + (size_t)8 * w.tiles * dy->ld), w.tiles, dy->ld, w.co_rows, w.R, workspace + w.db_off);
    ITG_CHECK_LAUNCH();
  }
  {
    // the 49 contractions over the tiles: the weight gradient of a 1 x 1 convolution between the tile "images" (one row of
    // `tiles` pixels), one uniform class per transformed point
    WgP p;
    memset(&p, 0, sizeof(p));
    itg_tensor vx = {V, 1, 1, 1, 1, (int)w.tiles, x->c, x->ld};
    itg_tensor vdy = {dM, 1, 1, 1, 1, (int)w.tiles, dy->c, dy->ld};
    p.x = make_grid(&vx); p.dy = make_grid(&vdy);
    p.slab = dU; p.dbslab = nullptr;
    p.ntaps = 1; p.kw = 1; p.cin_ld = x->ld; p.Ktot = x->ld; p.Kpad = w.tn.Kpad;
    p.MT = 1; p.MU = (int)w.tiles; p.M = (int)w.tiles;
    p.stride = 1; p.pad = 0; p.pad_h = 0; p.pad_mode = ITG_PAD_ZERO;
    p.co_rows = w.tn.co_rows;
    p.chunks_per_split = w.tn.chunks_per_split; p.nchunks = w.tn.nchunks;
    p.x_bytes = (unsigned)(w.tiles * x->ld * 4); p.dy_bytes = (unsigned)(w.tiles * dy->ld * 4);
    p.in_ab = nullptr; p.in_act = ITG_ACT_NONE;
    p.ucls = 49; p.u_x = (unsigned)(w.tiles * x->ld); p.u_dy = (unsigned)(w.tiles * dy->ld);
    TileWgPlan none;
    memset(&none, 0, sizeof(none));
    int rc = run_wgrad_slabs(p, w.tn, none, prec, s);
    if (rc) return rc;
  }
  {
    const int64_t th = (int64_t)w.co_rows * x->ld;
    hipLaunchKernelGGL(wino_wg_out_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, (const float*)dU, w.tn.splits,
                       w.co_rows, w.tn.Kpad, x->ld, workspace + w.slab_off);
    ITG_CHECK_LAUNCH();
  }
  return ITG_OK;
}

}  // namespace itgk
