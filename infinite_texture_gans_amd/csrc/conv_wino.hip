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

}  // namespace itgk
