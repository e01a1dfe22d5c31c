// Winograd F(4 x 4, R x R), R = 4 and 3, for wide stride-1 convolutions (gfx950).
//
// R = 4: the discriminator's 256 -> 512 layer (reference models/discriminators.py:196-206: conv4x4, stride 1, padding 1 on the
// 48 x 48 maps) is the one layer of the model that is MFMA-bound AND has a fast algorithm: 49 multiplications per 4 x 4
// output tile instead of 256.  R = 3: the generator's wide blocks (416 / 208 channels on 4 x 4 / 8 x 8 patches, reference
// models/layers.py:25-34,301-311): 36 instead of 144, and a 6 x 6 input tile IS one locally padded 4 x 4 patch.
// y = A^T [ (G g G^T) .* (B^T d B) ] A per (output channel, input channel) pair, summed over input channels in the
// transformed domain - i.e. NP^2 independent GEMMs [tiles x Cin] x [Cin x Cout] (NP = 4 + R - 1):
//   wino_in_kernel    d (NP x NP input tiles, stride 4, gathered in merged patch-grid coordinates, zero or replicate frame)
//                                                                      ->  V[xi][tile][ci]        (B^T d B per channel)
//   conv_nt_kernel    NP^2 uniform classes of a 1 x 1 convolution      ->  M[xi][tile][co]        (ConvP.ucls)
//   wino_out_kernel   A^T m A per tile and channel, 1/sigma, bias, residual (also through a x2 upsample), activation (or the
//                     producing layer's activation derivative for an input gradient), cropped to the output extent; for the
//                     input gradient of a replicate-padded layer the (H + 2) x (W + 2) result folds its frame onto the
//                     border pixels with atomics (the caller has zeroed dx's frame)
// The panel U[xi][co][ci] = G g G^T comes from itg_pack_wino*_ (conv.hip).  The input gradient of the layer is the same
// pipeline on dy with the flipped, transposed filter and padding R - 1 - pad.  fp32 throughout; measured error of R = 4:
// 4.6e-6 rel-L2 against fp64 (direct fp32: 3e-7); tools/gen_winograd.py has the matrices.
#include <algorithm>
#include <cstring>
#include "conv_common.h"
#include "winograd_f44.h"
#include "winograd_f43.h"
#include "winograd_f42.h"

namespace itgk {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int R> struct WM;
template <> struct WM<4> {
  static constexpr int NP = 7;
  static __device__ constexpr float at(int k, int i) { return WINO_AT[k][i]; }
  static __device__ constexpr float g(int i, int j) { return WINO_G[i][j]; }
  static __device__ constexpr float bt(int i, int j) { return WINO_BT[i][j]; }
};
template <> struct WM<3> {
  static constexpr int NP = 6;
  static __device__ constexpr float at(int k, int i) { return WINO3_AT[k][i]; }
  static __device__ constexpr float g(int i, int j) { return WINO3_G[i][j]; }
  static __device__ constexpr float bt(int i, int j) { return WINO3_BT[i][j]; }
};

template <> struct WM<2> {
  static constexpr int NP = 5;
  static __device__ constexpr float at(int k, int i) { return WINO2_AT[k][i]; }
  static __device__ constexpr float g(int i, int j) { return WINO2_G[i][j]; }
  static __device__ constexpr float bt(int i, int j) { return WINO2_BT[i][j]; }
};

// one thread = (tile, channel pair): NP x NP loads of 8 bytes, V = B^T d B, NP^2 coalesced 8-byte stores
template <int R>
__global__ __launch_bounds__(256) void wino_in_kernel(const GridT x, int pad, int pad_mode, int Ty, int Tx, float* __restrict__ V) {
  constexpr int NP = WM<R>::NP;
  const int ld = x.ld, cpairs = ld >> 1;
  const int64_t tiles = (int64_t)x.n * Ty * Tx;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= tiles * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int64_t tile = gt / cpairs;
  const int tx = (int)(tile % Tx), ty = (int)((tile / Tx) % Ty), img = (int)(tile / ((int64_t)Ty * Tx));
  const int y0 = 4 * ty - pad, x0 = 4 * tx - pad;
  const bool rep = pad_mode == ITG_PAD_REPLICATE;
  f32x2 tmp[NP][NP];                     // tmp[i][b] = sum_j BT[b][j] d[i][j]
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    f32x2 d[NP];
    int iy = y0 + i;
    const bool oky = (unsigned)iy < (unsigned)x.H;
    iy = min(max(iy, 0), x.H - 1);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      int ix = x0 + j;
      const bool ok = rep || (oky && (unsigned)ix < (unsigned)x.W);
      ix = min(max(ix, 0), x.W - 1);
      d[j] = ok ? *reinterpret_cast<const f32x2*>(x.p + grid_off(x, img, iy, ix) + 2 * cp) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (WM<R>::bt(b, j) != 0.f) a += WM<R>::bt(b, j) * d[j];
      tmp[i][b] = a;
    }
  }
  const size_t plane = (size_t)tiles * ld;
  float* const vb = V + (size_t)tile * ld + 2 * cp;
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (WM<R>::bt(a, i) != 0.f) v += WM<R>::bt(a, i) * tmp[i][b];
      *reinterpret_cast<f32x2*>(vb + (size_t)(a * NP + b) * plane) = v;
    }
}

// 4 x 4 STRIDE-2 convolutions (pad 1, zero padding; the discriminator's 64 -> 128 and 128 -> 256 layers, reference
// models/discriminators.py:190-195): with ky = 2 jy + a, kx = 2 jx + b the layer is the SUM over the four parities (a, b) of a
// 2 x 2 stride-1 correlation of the parity-decimated image x_ab(s, r) = x(2 s + a - 1, 2 r + b - 1) with g_ab(jy, jx) =
// w(2 jy + a, 2 jx + b) - so F(4 x 4, 2 x 2) applies: 25 multiplications per 4 x 4 output tile and parity class instead of 64,
// i.e. 25 GEMMs [tiles x 4 Cin] x [4 Cin x Cout] (the classes are concatenated along K: M = sum_ab U_ab .* V_ab) - 6.25 Cin
// multiplications per output where the direct form has 16 Cin.  One thread = (tile, class, channel pair): 5 x 5 loads at pixel
// stride 2, V[xi][tile][cls * ld + c] = B^T d B.
__global__ __launch_bounds__(256) void wino_in_s2_kernel(const GridT x, int Ty, int Tx, float* __restrict__ V) {
  constexpr int NP = 5;
  const int ld = x.ld, cpairs = ld >> 1;
  const int64_t tiles = (int64_t)x.n * Ty * Tx;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= tiles * 4 * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int cls = (int)((gt / cpairs) & 3);
  const int64_t tile = gt / (4 * cpairs);
  const int tx = (int)(tile % Tx), ty = (int)((tile / Tx) % Ty), img = (int)(tile / ((int64_t)Ty * Tx));
  const int y0 = 8 * ty + (cls >> 1) - 1, x0 = 8 * tx + (cls & 1) - 1;
  f32x2 tmp[NP][NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    f32x2 d[NP];
    const int iy = y0 + 2 * i;
    const bool oky = (unsigned)iy < (unsigned)x.H;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int ix = x0 + 2 * j;
      const bool ok = oky && (unsigned)ix < (unsigned)x.W;
      d[j] = ok ? *reinterpret_cast<const f32x2*>(x.p + grid_off(x, img, ok ? iy : 0, ok ? ix : 0) + 2 * cp) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (WM<2>::bt(b, j) != 0.f) a += WM<2>::bt(b, j) * d[j];
      tmp[i][b] = a;
    }
  }
  const size_t plane = (size_t)tiles * 4 * ld;
  float* const vb = V + (size_t)tile * 4 * ld + cls * ld + 2 * cp;
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (WM<2>::bt(a, i) != 0.f) v += WM<2>::bt(a, i) * tmp[i][b];
      *reinterpret_cast<f32x2*>(vb + (size_t)(a * NP + b) * plane) = v;
    }
}

// The ADJOINT of wino_in_s2_kernel: dx from dV[xi][tile][cls * ld + c] (the gradient w.r.t. V).  V = B^T d B per tile and class,
// so the gradient of the class's decimated pixel (s, r) is the sum over the tiles that cover it of (B dV B^T)[s - 4 ty][r - 4 tx]
// - tiles are 5 x 5 at stride 4, so a pixel on a tile's first row / column also lies on the LAST row / column of the tile above /
// to the left.  One thread = (4 x 4 block of the class image, class, channel pair): its own tile's 25 values, row 4 of the tile
// above, column 4 of the tile to the left, the corner of the diagonal one (L2 hits: the neighbouring threads' tiles) - a gather,
// so every dx pixel is written once with its complete sum (1/sigma and the producing layer's activation derivative applied).
__global__ __launch_bounds__(256) void wino_in_s2_adjoint_kernel(const float* __restrict__ dV, const GridT dx, int Ty, int Tx,
                                                                 const float* __restrict__ scale, const GridT res, int res_mode,
                                                                 float res_slope) {
  constexpr int NP = 5;
  const int ld = dx.ld, cpairs = ld >> 1;
  const int64_t tiles = (int64_t)dx.n * Ty * Tx;
  // the block grid has one more row / column of blocks than tiles: the last row / column of the last tiles lands there
  const int By = Ty + 1, Bx = Tx + 1;
  const int64_t blocks = (int64_t)dx.n * By * Bx;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= blocks * 4 * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int cls = (int)((gt / cpairs) & 3);
  const int64_t blk = gt / (4 * cpairs);
  const int bx = (int)(blk % Bx), by = (int)((blk / Bx) % By), img = (int)(blk / ((int64_t)By * Bx));
  const size_t plane = (size_t)tiles * 4 * ld;
  f32x2 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) d[i][j] = f32x2{0.f, 0.f};
  // contribution of tile (ty, tx) to local rows i0 .. and columns j0 .. of this block: rows `ri` / columns `rj` of B dV B^T
  auto add_tile = [&](int ty, int tx, bool last_row, bool last_col) {
    if ((unsigned)ty >= (unsigned)Ty || (unsigned)tx >= (unsigned)Tx) return;
    const int64_t tile = ((int64_t)img * Ty + ty) * Tx + tx;
    const float* vb = dV + (size_t)tile * 4 * ld + cls * ld + 2 * cp;
    // t[al][j] = sum_be dV[al][be] BT[be][j] for the needed columns j, then rows
    f32x2 t[NP][4];
    f32x2 tc[NP];                            // column 4 (last_col)
#pragma unroll
    for (int al = 0; al < NP; ++al) {
      f32x2 v[NP];
#pragma unroll
      for (int be = 0; be < NP; ++be) v[be] = *reinterpret_cast<const f32x2*>(vb + (size_t)(al * NP + be) * plane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 a = {0.f, 0.f};
#pragma unroll
        for (int be = 0; be < NP; ++be)
          if (WM<2>::bt(be, j) != 0.f) a += WM<2>::bt(be, j) * v[be];
        t[al][j] = a;
      }
      f32x2 a4 = {0.f, 0.f};
#pragma unroll
      for (int be = 0; be < NP; ++be)
        if (WM<2>::bt(be, 4) != 0.f) a4 += WM<2>::bt(be, 4) * v[be];
      tc[al] = a4;
    }
    if (!last_row && !last_col) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x2 a = {0.f, 0.f};
#pragma unroll
          for (int al = 0; al < NP; ++al)
            if (WM<2>::bt(al, i) != 0.f) a += WM<2>::bt(al, i) * t[al][j];
          d[i][j] += a;
        }
    } else if (last_row && !last_col) {      // the tile above: its row 4 lands on this block's row 0
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 a = {0.f, 0.f};
#pragma unroll
        for (int al = 0; al < NP; ++al)
          if (WM<2>::bt(al, 4) != 0.f) a += WM<2>::bt(al, 4) * t[al][j];
        d[0][j] += a;
      }
    } else if (!last_row && last_col) {      // the tile to the left: its column 4 lands on this block's column 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 a = {0.f, 0.f};
#pragma unroll
        for (int al = 0; al < NP; ++al)
          if (WM<2>::bt(al, i) != 0.f) a += WM<2>::bt(al, i) * tc[al];
        d[i][0] += a;
      }
    } else {
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int al = 0; al < NP; ++al)
        if (WM<2>::bt(al, 4) != 0.f) a += WM<2>::bt(al, 4) * tc[al];
      d[0][0] += a;
    }
  };
  add_tile(by, bx, false, false);
  add_tile(by - 1, bx, true, false);
  add_tile(by, bx - 1, false, true);
  add_tile(by - 1, bx - 1, true, true);
  const float osc = scale ? *scale : 1.f;
  const int a_ = cls >> 1, b_ = cls & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int Y = 2 * (4 * by + i) + a_ - 1;
    if ((unsigned)Y >= (unsigned)dx.H) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int X = 2 * (4 * bx + j) + b_ - 1;
      if ((unsigned)X >= (unsigned)dx.W) continue;
      f32x2 v = d[i][j] * osc;
      const int off = grid_off(dx, img, Y, X) + 2 * cp;
      if (res.p) {
        const f32x2 r = *reinterpret_cast<const f32x2*>(res.p + grid_off(res, img, Y, X) + 2 * cp);
#pragma unroll
        for (int e = 0; e < 2; ++e)
          v[e] *= res_mode == ITG_ACT_LRELU ? (r[e] > 0.f ? 1.f : res_slope) : (res_mode == ITG_ACT_TANH ? 1.f - r[e] * r[e] : 1.f);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (2 * cp + e >= dx.c) v[e] = 0.f;
      *reinterpret_cast<f32x2*>(dx.p + off) = v;
    }
  }
}

// one thread = (tile, output-channel pair): NP^2 loads, Y = A^T m A (4 x 4), epilogue, <= 16 stores
// res_mode 0: y += res (read at (oy >> res_ups, ox >> res_ups));  ITG_ACT_*: y *= act'(res) (res = the activation's OUTPUT, as
// itg_conv2d_dgrad's act_out).  fold = 1: the tiles cover the (H + 2) x (W + 2) gradient of a replicate-padded tensor; element
// (py, px) lands on pixel (clamp(py - 1), clamp(px - 1)), frame pixels of `out` by atomic adds onto their zeroed start.
template <int R>
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ Mm, const GridT out, int Ty, int Tx, int fold,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const GridT res, int res_ups, int res_mode, float res_slope, int act,
                                                        float slope) {
  constexpr int NP = WM<R>::NP;
  const int ld = out.ld, c = out.c, cpairs = ld >> 1;
  const int64_t tiles = (int64_t)out.n * Ty * Tx;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= tiles * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int64_t tile = gt / cpairs;
  const int tx = (int)(tile % Tx), ty = (int)((tile / Tx) % Ty), img = (int)(tile / ((int64_t)Ty * Tx));
  const size_t plane = (size_t)tiles * ld;
  const float* const mb = Mm + (size_t)tile * ld + 2 * cp;
  f32x2 tmp[NP][4];                       // tmp[a][l] = sum_b m[a][b] AT[l][b]
#pragma unroll
  for (int a = 0; a < NP; ++a) {
    f32x2 m[NP];
#pragma unroll
    for (int b = 0; b < NP; ++b) m[b] = *reinterpret_cast<const f32x2*>(mb + (size_t)(a * NP + b) * plane);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NP; ++b)
        if (WM<R>::at(l, b) != 0.f) v += WM<R>::at(l, b) * m[b];
      tmp[a][l] = v;
    }
  }
  const float osc = scale ? *scale : 1.f;
  f32x2 bv = {0.f, 0.f};
  if (bias) {
    if (2 * cp < c) bv[0] = bias[2 * cp];
    if (2 * cp + 1 < c) bv[1] = bias[2 * cp + 1];
  }
  const int He = out.H + 2 * fold, We = out.W + 2 * fold;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int py = 4 * ty + k;
    if (py >= He) continue;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int px = 4 * tx + l;
      if (px >= We) continue;
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int a = 0; a < NP; ++a)
        if (WM<R>::at(k, a) != 0.f) v += WM<R>::at(k, a) * tmp[a][l];
      v = v * osc + bv;
      const int oy = min(max(py - fold, 0), out.H - 1), ox = min(max(px - fold, 0), out.W - 1);
      const bool border = fold && ((oy == 0) | (oy == out.H - 1) | (ox == 0) | (ox == out.W - 1));
      if (res.p) {
        const f32x2 r = *reinterpret_cast<const f32x2*>(res.p + grid_off(res, img, oy >> res_ups, ox >> res_ups) + 2 * cp);
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
      float* dst = out.p + grid_off(out, img, oy, ox) + 2 * cp;
      if (border) {
        atomicAdd(dst, v[0]);
        atomicAdd(dst + 1, v[1]);
      } else {
        *reinterpret_cast<f32x2*>(dst) = v;
      }
    }
  }
}

// ---- weight gradient: dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G per (output channel, input channel) pair
// one thread = (tile, output-channel pair): the 4 x 4 tile of dy (zero past the output extent) -> dM = A dY A^T, NP^2 stores
template <int R>
__global__ __launch_bounds__(256) void wino_dy_kernel(const GridT dy, int Ty, int Tx, float* __restrict__ dM) {
  constexpr int NP = WM<R>::NP;
  const int ld = dy.ld, cpairs = ld >> 1;
  const int64_t tiles = (int64_t)dy.n * Ty * Tx;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= tiles * cpairs) return;
  const int cp = (int)(gt % cpairs);
  const int64_t tile = gt / cpairs;
  const int tx = (int)(tile % Tx), ty = (int)((tile / Tx) % Ty), img = (int)(tile / ((int64_t)Ty * Tx));
  f32x2 tmp[4][NP];                        // tmp[k][b] = sum_l dY[k][l] AT[l][b]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x2 d[4];
    const int oy = 4 * ty + k;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int ox = 4 * tx + l;
      d[l] = (oy < dy.H && ox < dy.W)
                 ? *reinterpret_cast<const f32x2*>(dy.p + grid_off(dy, img, min(oy, dy.H - 1), min(ox, dy.W - 1)) + 2 * cp)
                 : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int l = 0; l < 4; ++l)
        if (WM<R>::at(l, b) != 0.f) v += WM<R>::at(l, b) * d[l];
      tmp[k][b] = v;
    }
  }
  const size_t plane = (size_t)tiles * ld;
  float* const mb = dM + (size_t)tile * ld + 2 * cp;
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      f32x2 v = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (WM<R>::at(k, a) != 0.f) v += WM<R>::at(k, a) * tmp[k][b];
      *reinterpret_cast<f32x2*>(mb + (size_t)(a * NP + b) * plane) = v;
    }
}

// bias-gradient partials: the interpolation point 1 has A row (1, 1, 1, 1), so class (1, 1) of dM holds every tile's sum of
// dy; dbslab[r][c] = sum of the r-th run of tiles, in a fixed order (the generic reduce stage adds the Rr rows)
static_assert(WINO_AT[0][1] == 1.f && WINO_AT[1][1] == 1.f && WINO_AT[2][1] == 1.f && WINO_AT[3][1] == 1.f, "point 1 is column 1");
static_assert(WINO3_AT[0][1] == 1.f && WINO3_AT[1][1] == 1.f && WINO3_AT[2][1] == 1.f && WINO3_AT[3][1] == 1.f, "point 1 is column 1");
static_assert(WINO2_AT[0][1] == 1.f && WINO2_AT[1][1] == 1.f && WINO2_AT[2][1] == 1.f && WINO2_AT[3][1] == 1.f, "point 1 is column 1");
__global__ __launch_bounds__(256) void wino_db_kernel(const float* __restrict__ dM11, int64_t tiles, int ld, int co_rows, int Rr,
                                                       float* __restrict__ dbslab) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= co_rows) return;
  const int64_t per = (tiles + Rr - 1) / Rr, t0 = r * per, t1 = t0 + per < tiles ? t0 + per : tiles;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < ld) {
    int64_t t = t0;
    for (; t + 4 <= t1; t += 4) {
      s0 += dM11[t * ld + c]; s1 += dM11[(t + 1) * ld + c]; s2 += dM11[(t + 2) * ld + c]; s3 += dM11[(t + 3) * ld + c];
    }
    for (; t < t1; ++t) s0 += dM11[t * ld + c];
  }
  dbslab[(size_t)r * co_rows + c] = (s0 + s1) + (s2 + s3);
}

// one thread = (co, ci): dU[z][xi][co][ci] summed over the splits z in order, dg = G^T dU G, written as the generic slab
// [co][(i * R + j) * ci_ld + ci] (wgrad_reduce_kernel's input)
template <int R>
__global__ __launch_bounds__(256) void wino_wg_out_kernel(const float* __restrict__ dU, int splits, int co_rows, int Kp, int ci_ld,
                                                           float* __restrict__ slab) {
  constexpr int NP = WM<R>::NP;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= (int64_t)co_rows * ci_ld) return;
  const int ci = (int)(gt % ci_ld), co = (int)(gt / ci_ld);
  const size_t plane = (size_t)co_rows * Kp;
  const float* src = dU + (size_t)co * Kp + ci;
  float tmp[NP][R];                       // tmp[a][j] = sum_b dU[a][b] G[b][j]
#pragma unroll
  for (int a = 0; a < NP; ++a) {
    float u[NP];
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      float v = 0.f;
      for (int z = 0; z < splits; ++z) v += src[((size_t)z * NP * NP + a * NP + b) * plane];
      u[b] = v;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float v = 0.f;
#pragma unroll
      for (int b = 0; b < NP; ++b)
        if (WM<R>::g(b, j) != 0.f) v += WM<R>::g(b, j) * u[b];
      tmp[a][j] = v;
    }
  }
  float* dst = slab + (size_t)co * (R * R * ci_ld) + ci;
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float v = 0.f;
#pragma unroll
      for (int a = 0; a < NP; ++a)
        if (WM<R>::g(a, i) != 0.f) v += WM<R>::g(a, i) * tmp[a][j];
      dst[(size_t)(i * R + j) * ci_ld] = v;
    }
}

// ... of the stride-2 form: one thread = (co, parity class, ci): dU[z][xi][co][cls * ci_ld + ci] summed over the splits,
// dg_ab = G^T dU G (2 x 2), written at the class's taps (2 jy + a, 2 jx + b) of the generic 4 x 4 slab
__global__ __launch_bounds__(256) void wino_wg_out_s2_kernel(const float* __restrict__ dU, int splits, int co_rows, int Kp, int ci_ld,
                                                              float* __restrict__ slab) {
  constexpr int NP = 5;
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt >= (int64_t)co_rows * 4 * ci_ld) return;
  const int ci = (int)(gt % ci_ld), cls = (int)((gt / ci_ld) & 3), co = (int)(gt / (4 * ci_ld));
  const int a = cls >> 1, b = cls & 1;
  const size_t plane = (size_t)co_rows * Kp;
  const float* src = dU + (size_t)co * Kp + cls * ci_ld + ci;
  float tmp[NP][2];                       // tmp[al][jx] = sum_be dU[al][be] G[be][jx]
#pragma unroll
  for (int al = 0; al < NP; ++al) {
    float u[NP];
#pragma unroll
    for (int be = 0; be < NP; ++be) {
      float v = 0.f;
      for (int z = 0; z < splits; ++z) v += src[((size_t)z * NP * NP + al * NP + be) * plane];
      u[be] = v;
    }
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
      float v = 0.f;
#pragma unroll
      for (int be = 0; be < NP; ++be)
        if (WM<2>::g(be, jx) != 0.f) v += WM<2>::g(be, jx) * u[be];
      tmp[al][jx] = v;
    }
  }
  float* dst = slab + (size_t)co * (16 * ci_ld) + ci;
#pragma unroll
  for (int jy = 0; jy < 2; ++jy)
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
      float v = 0.f;
#pragma unroll
      for (int al = 0; al < NP; ++al)
        if (WM<2>::g(al, jy) != 0.f) v += WM<2>::g(al, jy) * tmp[al][jx];
      dst[(size_t)((2 * jy + a) * 4 + 2 * jx + b) * ci_ld] = v;
    }
}

// tiles per image of the transformed problem whose outputs cover He x We pixels
inline void wino_tiles(int He, int We, int& Ty, int& Tx) { Ty = (He + 3) / 4; Tx = (We + 3) / 4; }

}  // namespace

// workspace: V[NP^2][tiles][in.ld] | M[NP^2][tiles][out.ld]; fold: see wino_conv
int64_t wino_workspace_floats(const itg_tensor* in, const itg_tensor* out, int R, int fold) {
  const int He = out->gh * out->ph + 2 * fold, We = out->gw * out->pw + 2 * fold;
  int Ty, Tx;
  wino_tiles(He, We, Ty, Tx);
  const int NP = 4 + R - 1;
  const int64_t tiles = (int64_t)in->n * Ty * Tx;
  return (int64_t)NP * NP * tiles * ((int64_t)in->ld + out->ld);
}

// in -> out: R x R stride-1 correlation with padding `pad` (zero or replicate frame, gathered in merged patch-grid coordinates).
// fold = 1 (input gradient of a replicate-padded layer): out has in's extent, the correlation is evaluated on (H + 2) x (W + 2)
// with pad = R - 1 and its frame folded onto out's border pixels, whose 1-pixel frame the caller has zeroed.
int wino_conv(const itg_tensor* in, const float* u_panel, const float* bias, const float* out_scale, const itg_tensor* res, int res_ups,
              int res_mode, float res_slope, const itg_tensor* out, int R, int pad, int pad_mode, int fold, int act, float slope,
              int prec, float* workspace, int64_t workspace_floats, hipStream_t s, bool input_gradient) {
  const int H = in->gh * in->ph, W = in->gw * in->pw, Ho = out->gh * out->ph, Wo = out->gw * out->pw;
  if (in->n != out->n || (R != 3 && R != 4)) return ITG_ERR_ARG;
  const int He = Ho + 2 * fold, We = Wo + 2 * fold;
  if (He != H + 2 * pad - (R - 1) || We != W + 2 * pad - (R - 1) || (in->ld & 15) || (out->ld & 3)) return ITG_ERR_ARG;
  if (fold && pad_mode != ITG_PAD_ZERO) return ITG_ERR_ARG;       // the folded form gathers dy with zeros outside
  if (res && (res->n != out->n || res->ld != out->ld || (res->gh * res->ph) << res_ups != Ho || (res->gw * res->pw) << res_ups != Wo))
    return ITG_ERR_ARG;
  int Ty, Tx;
  wino_tiles(He, We, Ty, Tx);
  const int NP = 4 + R - 1, NC = NP * NP;
  const int64_t tiles = (int64_t)in->n * Ty * Tx;
  const int64_t vf = NC * tiles * in->ld, mf = NC * tiles * out->ld;
  if (!workspace || workspace_floats < vf + mf) return ITG_ERR_WORKSPACE;
  if (tiles * std::max(in->ld, out->ld) * 4 >= 0xFFFF0000LL) return ITG_ERR_ARG;
  float* V = workspace;
  float* Mm = workspace + vf;
  {
    const int64_t th = tiles * (in->ld >> 1);
    const dim3 grid((unsigned)((th + 255) / 256));
    if (R == 4) hipLaunchKernelGGL(wino_in_kernel<4>, grid, dim3(256), 0, s, make_grid(in), pad, pad_mode, Ty, Tx, V);
    else hipLaunchKernelGGL(wino_in_kernel<3>, grid, dim3(256), 0, s, make_grid(in), pad, pad_mode, Ty, Tx, V);
    ITG_CHECK_LAUNCH();
  }
  {
    // the NP^2 GEMMs: a 1 x 1 convolution over the tile "images" [n][Ty][Tx][ci] with NP^2 uniform classes
    ConvP p;
    memset(&p, 0, sizeof(p));
    itg_tensor vin = {V, in->n, 1, 1, Ty, Tx, in->c, in->ld};
    itg_tensor vout = {Mm, in->n, 1, 1, Ty, Tx, out->c, out->ld};
    p.in = make_grid(&vin); p.out = make_grid(&vout); p.res = null_grid();
    p.w = u_panel; p.bias = nullptr; p.scale = nullptr;
    p.ntaps = 1; p.kw = 1; p.cin_ld = in->ld; p.Kpad = round_up(in->ld, BK);
    p.MT = Ty; p.MU = Tx; p.M = (int)tiles;
    p.isy = p.isx = 1; p.osy = p.osx = 1;
    p.pad_mode = ITG_PAD_ZERO; p.act = ITG_ACT_NONE;
    p.co_rows = round_up(out->c, 16);
    p.prec = prec;
    p.ncls = 1;
    p.ucls = NC;
    p.u_acc = input_gradient ? 0 : 2;
    p.u_in = (unsigned)(tiles * in->ld); p.u_out = (unsigned)(tiles * out->ld); p.u_w = (unsigned)((size_t)p.co_rows * p.Kpad);
    int rc = dispatch_nt(p, nullptr, 0, s);
    if (rc) return rc;
  }
  {
    const int64_t th = tiles * (out->ld >> 1);
    const dim3 grid((unsigned)((th + 255) / 256));
    const GridT rg = res ? make_grid(res) : null_grid();
    if (R == 4)
      hipLaunchKernelGGL(wino_out_kernel<4>, grid, dim3(256), 0, s, (const float*)Mm, make_grid(out), Ty, Tx, fold, bias, out_scale, rg,
                         res_ups, res_mode, res_slope, act, slope);
    else
      hipLaunchKernelGGL(wino_out_kernel<3>, grid, dim3(256), 0, s, (const float*)Mm, make_grid(out), Ty, Tx, fold, bias, out_scale, rg,
                         res_ups, res_mode, res_slope, act, slope);
    ITG_CHECK_LAUNCH();
  }
  return ITG_OK;
}

// workspace of the stride-2 form: V[25][tiles][4 * in.ld] | M[25][tiles][out.ld]
int64_t wino_s2_workspace_floats(const itg_tensor* in, const itg_tensor* out) {
  int Ty, Tx;
  wino_tiles(out->gh * out->ph, out->gw * out->pw, Ty, Tx);
  const int64_t tiles = (int64_t)in->n * Ty * Tx;
  return 25 * tiles * (4 * (int64_t)in->ld + out->ld);
}

// in -> out: 4 x 4 stride-2 pad-1 convolution with zero padding through F(4 x 4, 2 x 2) on the four parity classes (forward
// only: the layer's input and weight gradients stay on the direct kernels - in the transformed domain they would move 4 x the
// bytes of dx per class, which costs what the multiplications save).  u_panel: itg_pack_wino_s2_fwd.
int wino_conv_s2(const itg_tensor* in, const float* u_panel, const float* bias, const float* out_scale, const itg_tensor* out, int act,
                 float slope, int prec, float* workspace, int64_t workspace_floats, hipStream_t s) {
  const int H = in->gh * in->ph, W = in->gw * in->pw, Ho = out->gh * out->ph, Wo = out->gw * out->pw;
  if (in->n != out->n || Ho != (H + 2 - 4) / 2 + 1 || Wo != (W + 2 - 4) / 2 + 1 || (in->ld & 3) || (out->ld & 3)) return ITG_ERR_ARG;
  int Ty, Tx;
  wino_tiles(Ho, Wo, Ty, Tx);
  constexpr int NC = 25;
  const int64_t tiles = (int64_t)in->n * Ty * Tx;
  const int kld = 4 * in->ld;
  const int64_t vf = NC * tiles * kld, mf = NC * tiles * out->ld;
  if (!workspace || workspace_floats < vf + mf) return ITG_ERR_WORKSPACE;
  if (tiles * std::max(kld, (int)out->ld) * 4 >= 0xFFFF0000LL) return ITG_ERR_ARG;
  float* V = workspace;
  float* Mm = workspace + vf;
  {
    const int64_t th = tiles * 4 * (in->ld >> 1);
    hipLaunchKernelGGL(wino_in_s2_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, make_grid(in), Ty, Tx, V);
    ITG_CHECK_LAUNCH();
  }
  {
    ConvP p;
    memset(&p, 0, sizeof(p));
    itg_tensor vin = {V, in->n, 1, 1, Ty, Tx, kld, kld};
    itg_tensor vout = {Mm, in->n, 1, 1, Ty, Tx, out->c, out->ld};
    p.in = make_grid(&vin); p.out = make_grid(&vout); p.res = null_grid();
    p.w = u_panel; p.bias = nullptr; p.scale = nullptr;
    p.ntaps = 1; p.kw = 1; p.cin_ld = kld; p.Kpad = round_up(kld, BK);
    p.MT = Ty; p.MU = Tx; p.M = (int)tiles;
    p.isy = p.isx = 1; p.osy = p.osx = 1;
    p.pad_mode = ITG_PAD_ZERO; p.act = ITG_ACT_NONE;
    p.co_rows = round_up(out->c, 16);
    p.prec = prec;
    p.ncls = 1;
    p.ucls = NC;
    // blocked accumulation as for F(4 x 4, 4 x 4) (ITG_WINO_ACC64): these are FORWARD GEMMs in front of a LeakyReLU - their
    // rounding decides sign flips that move every upstream gradient by ~1e-3.  Plain fp32 chains of 64 - 128 MFMA
    // steps gave 1.1 - 2.5e-6 per layer and G's full-size gradients 2.1e-3 / 2.9e-3 from the fp64 truth (direct: 1.3 / 1.8);
    // blocks of 16 summed in a second fp32 accumulator (NT_W32) halve that at the plain kernel's occupancy
    p.u_acc = 1;
    p.u_in = (unsigned)(tiles * kld); p.u_out = (unsigned)(tiles * out->ld); p.u_w = (unsigned)((size_t)p.co_rows * p.Kpad);
    int rc = dispatch_nt(p, nullptr, 0, s);
    if (rc) return rc;
  }
  {
    const int64_t th = tiles * (out->ld >> 1);
    hipLaunchKernelGGL(wino_out_kernel<2>, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, (const float*)Mm, make_grid(out), Ty, Tx, 0, bias,
                       out_scale, null_grid(), 0, 0, 0.f, act, slope);
    ITG_CHECK_LAUNCH();
  }
  return ITG_OK;
}

// workspace of the stride-2 input gradient: dM[25][tiles][dy.ld] | dV[25][tiles][4 * dx.ld]
int64_t wino_s2_dgrad_workspace_floats(const itg_tensor* dy, const itg_tensor* dx) {
  int Ty, Tx;
  wino_tiles(dy->gh * dy->ph, dy->gw * dy->pw, Ty, Tx);
  const int64_t tiles = (int64_t)dy->n * Ty * Tx;
  return 25 * tiles * ((int64_t)dy->ld + 4 * dx->ld);
}

// dy -> dx of the 4 x 4 stride-2 pad-1 layer, as the adjoint of wino_conv_s2: dM = A dY A^T, dV[xi] = dM[xi] . U[xi] (25 GEMMs
// [tiles x Cout] x [Cout x 4 Cin], panel itg_pack_wino_s2_dgrad), dx = the gathered B dV B^T of the parity classes.
int wino_conv_s2_dgrad(const itg_tensor* dy, const float* ut_panel, const float* out_scale, const itg_tensor* dx, const itg_tensor* act_out,
                       int act, float slope, int prec, float* workspace, int64_t workspace_floats, hipStream_t s) {
  const int H = dx->gh * dx->ph, W = dx->gw * dx->pw, Ho = dy->gh * dy->ph, Wo = dy->gw * dy->pw;
  if (dx->n != dy->n || Ho != (H + 2 - 4) / 2 + 1 || Wo != (W + 2 - 4) / 2 + 1 || (dx->ld & 3) || (dy->ld & 15)) return ITG_ERR_ARG;
  int Ty, Tx;
  wino_tiles(Ho, Wo, Ty, Tx);
  // every dx pixel must lie inside the tiles' 5 x 5 windows of its class: 2 (4 Ty + 4) + 1 - 1 >= H - 1
  if (8 * Ty + 8 < H || 8 * Tx + 8 < W) return ITG_ERR_ARG;
  constexpr int NC = 25;
  const int64_t tiles = (int64_t)dy->n * Ty * Tx;
  const int kld = 4 * dx->ld;
  const int64_t mf = NC * tiles * dy->ld, vf = NC * tiles * kld;
  if (!workspace || workspace_floats < mf + vf) return ITG_ERR_WORKSPACE;
  if (tiles * std::max(kld, (int)dy->ld) * 4 >= 0xFFFF0000LL) return ITG_ERR_ARG;
  float* dM = workspace;
  float* dV = workspace + mf;
  {
    const int64_t th = tiles * (dy->ld >> 1);
    hipLaunchKernelGGL(wino_dy_kernel<2>, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, make_grid(dy), Ty, Tx, dM);
    ITG_CHECK_LAUNCH();
  }
  {
    ConvP p;
    memset(&p, 0, sizeof(p));
    itg_tensor vin = {dM, dy->n, 1, 1, Ty, Tx, dy->c, dy->ld};
    itg_tensor vout = {dV, dy->n, 1, 1, Ty, Tx, kld, kld};
    p.in = make_grid(&vin); p.out = make_grid(&vout); p.res = null_grid();
    p.w = ut_panel; p.bias = nullptr; p.scale = nullptr;
    p.ntaps = 1; p.kw = 1; p.cin_ld = dy->ld; p.Kpad = round_up(dy->ld, BK);
    p.MT = Ty; p.MU = Tx; p.M = (int)tiles;
    p.isy = p.isx = 1; p.osy = p.osx = 1;
    p.pad_mode = ITG_PAD_ZERO; p.act = ITG_ACT_NONE;
    p.co_rows = round_up(kld, 16);
    p.prec = prec;
    p.ncls = 1;
    p.ucls = NC;
    p.u_acc = 0;
    p.u_in = (unsigned)(tiles * dy->ld); p.u_out = (unsigned)(tiles * kld); p.u_w = (unsigned)((size_t)p.co_rows * p.Kpad);
    int rc = dispatch_nt(p, nullptr, 0, s);
    if (rc) return rc;
  }
  {
    const int64_t th = (int64_t)dx->n * (Ty + 1) * (Tx + 1) * 4 * (dx->ld >> 1);
    hipLaunchKernelGGL(wino_in_s2_adjoint_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, (const float*)dV, make_grid(dx), Ty, Tx,
                       out_scale, act_out ? make_grid(act_out) : null_grid(), act, slope);
    ITG_CHECK_LAUNCH();
  }
  return ITG_OK;
}

// ---- weight gradient (R x R stride-1 correlation with padding `pad`, zero or replicate frame)
// workspace: V[NP^2][tiles][x.ld] | dM[NP^2][tiles][dy.ld] | dU[splits][NP^2][co_rows][Kp] | slab[co_rows][R^2 * x.ld] | db[Rr][co_rows]
WinoWgPlan plan_wino_wgrad(const itg_tensor* x, const itg_tensor* dy, int R) {
  // R = 2: the 4 x 4 STRIDE-2 layer as four parity classes of 2 x 2 convolutions (wino_conv_s2): V carries the classes side by
  // side (4 * x.ld values per tile), dU likewise, and the slab is the 4 x 4 filter's
  WinoWgPlan w;
  const int Ho = dy->gh * dy->ph, Wo = dy->gw * dy->pw;
  const int NC = (4 + R - 1) * (4 + R - 1);
  const int kld = R == 2 ? 4 * x->ld : x->ld;
  w.R = R;
  wino_tiles(Ho, Wo, w.Ty, w.Tx);
  w.tiles = (int64_t)x->n * w.Ty * w.Tx;
  w.tn = plan_tn(w.tiles, dy->ld, kld, ITG_PREC_F32, NC);
  // pixel-range splits of the NP^2 contractions: 49 classes x 8 tiles fill the chip on their own, and every split is one
  // more slab for the output transform to read (256 -> 512 layer: 274 -> 258 us on the generated batch, 120 -> 107 on the real one)
  {
    // (the stride-2 form on D's first two layers: 25 classes x 1-2 x 2-4 tiles = 50-200 workgroups - split the tiles so that
    // ~512 workgroups run; the output transform sums the splits)
    int sp = 1;
    if (R == 2) {
      const int64_t wg = (int64_t)NC * ((kld + w.tn.bcol - 1) / w.tn.bcol) * ((w.tn.co_rows + w.tn.bco - 1) / w.tn.bco);
      sp = (int)std::min<int64_t>(8, std::max<int64_t>(1, (512 + wg - 1) / wg));
      if (sp > w.tn.nchunks) sp = w.tn.nchunks > 0 ? w.tn.nchunks : 1;
    }
    w.tn.chunks_per_split = (w.tn.nchunks + sp - 1) / sp;
    w.tn.splits = (w.tn.nchunks + w.tn.chunks_per_split - 1) / w.tn.chunks_per_split;
    w.tn.slab_floats = (int64_t)w.tn.splits * NC * w.tn.co_rows * w.tn.Kpad;
    w.tn.ngroups = 0;
    w.tn.ws_floats = w.tn.slab_floats;
  }
  w.co_rows = w.tn.co_rows;
  w.Kpad = (R == 2 ? 16 : R * R) * x->ld;
  w.Rr = (int)std::min<int64_t>(16, std::max<int64_t>(1, w.tiles / 32));
  w.v_off = 0;
  w.m_off = w.v_off + NC * w.tiles * kld;
  w.u_off = w.m_off + NC * w.tiles * dy->ld;
  w.slab_off = w.u_off + w.tn.slab_floats;
  w.db_off = w.slab_off + (int64_t)w.co_rows * w.Kpad;
  w.ws_floats = w.db_off + (int64_t)w.Rr * w.co_rows;
  return w;
}

int wino_wgrad_slabs(const itg_tensor* x, const itg_tensor* dy, int pad, int pad_mode, int prec, const WinoWgPlan& w, float* workspace,
                     bool want_db, hipStream_t s, const float* v_fwd) {
  const int H = x->gh * x->ph, W = x->gw * x->pw, Ho = dy->gh * dy->ph, Wo = dy->gw * dy->pw;
  const int R = w.R, NP = 4 + R - 1, NC = NP * NP;
  const int kld = R == 2 ? 4 * x->ld : x->ld;
  if (x->n != dy->n || (R != 2 && R != 3 && R != 4)) return ITG_ERR_ARG;
  if (R == 2 ? (pad != 1 || pad_mode != ITG_PAD_ZERO || Ho != (H + 2 - 4) / 2 + 1 || Wo != (W + 2 - 4) / 2 + 1)
             : (Ho != H + 2 * pad - (R - 1) || Wo != W + 2 * pad - (R - 1)))
    return ITG_ERR_ARG;
  if ((x->ld & 15) || (dy->ld & 15) || prec != ITG_PREC_F32) return ITG_ERR_ARG;
  if (w.tiles * std::max(kld, (int)dy->ld) * 4 >= 0xFFFF0000LL || w.tn.ngroups > 0) return ITG_ERR_ARG;
  const float* V = v_fwd ? v_fwd : workspace + w.v_off;       // v_fwd: the forward's transformed input (itg_conv_geom.wino_v)
  float* dM = workspace + w.m_off;
  float* dU = workspace + w.u_off;
  if (!v_fwd) {
    const int64_t th = w.tiles * (kld >> 1);
    const dim3 grid((unsigned)((th + 255) / 256));
    if (R == 2) hipLaunchKernelGGL(wino_in_s2_kernel, grid, dim3(256), 0, s, make_grid(x), w.Ty, w.Tx, workspace + w.v_off);
    else if (R == 4) hipLaunchKernelGGL(wino_in_kernel<4>, grid, dim3(256), 0, s, make_grid(x), pad, pad_mode, w.Ty, w.Tx, workspace + w.v_off);
    else hipLaunchKernelGGL(wino_in_kernel<3>, grid, dim3(256), 0, s, make_grid(x), pad, pad_mode, w.Ty, w.Tx, workspace + w.v_off);
    ITG_CHECK_LAUNCH();
  }
  {
    const int64_t th = w.tiles * (dy->ld >> 1);
    const dim3 grid((unsigned)((th + 255) / 256));
    if (R == 2) hipLaunchKernelGGL(wino_dy_kernel<2>, grid, dim3(256), 0, s, make_grid(dy), w.Ty, w.Tx, dM);
    else if (R == 4) hipLaunchKernelGGL(wino_dy_kernel<4>, grid, dim3(256), 0, s, make_grid(dy), w.Ty, w.Tx, dM);
    else hipLaunchKernelGGL(wino_dy_kernel<3>, grid, dim3(256), 0, s, make_grid(dy), w.Ty, w.Tx, dM);
    ITG_CHECK_LAUNCH();
  }
  if (want_db) {
    hipLaunchKernelGGL(wino_db_kernel, dim3((unsigned)((w.co_rows + 255) / 256), (unsigned)w.Rr), dim3(256), 0, s,
                       (const float*)(dM + (size_t)(NP + 1) * w.tiles * dy->ld), w.tiles, dy->ld, w.co_rows, w.Rr, workspace + w.db_off);
    ITG_CHECK_LAUNCH();
  }
  {
    // the NP^2 contractions over the tiles: the weight gradient of a 1 x 1 convolution between the tile "images" (one row of
    // `tiles` pixels), one uniform class per transformed point
    WgP p;
    memset(&p, 0, sizeof(p));
    itg_tensor vx = {const_cast<float*>(V), 1, 1, 1, 1, (int)w.tiles, R == 2 ? kld : x->c, kld};
    itg_tensor vdy = {dM, 1, 1, 1, 1, (int)w.tiles, dy->c, dy->ld};
    p.x = make_grid(&vx); p.dy = make_grid(&vdy);
    p.slab = dU; p.dbslab = nullptr;
    p.ntaps = 1; p.kw = 1; p.cin_ld = kld; p.Ktot = kld; p.Kpad = w.tn.Kpad;
    p.MT = 1; p.MU = (int)w.tiles; p.M = (int)w.tiles;
    p.stride = 1; p.pad = 0; p.pad_h = 0; p.pad_mode = ITG_PAD_ZERO;
    p.co_rows = w.tn.co_rows;
    p.chunks_per_split = w.tn.chunks_per_split; p.nchunks = w.tn.nchunks;
    p.x_bytes = (unsigned)(w.tiles * kld * 4); p.dy_bytes = (unsigned)(w.tiles * dy->ld * 4);
    p.ucls = NC; p.u_x = (unsigned)(w.tiles * kld); p.u_dy = (unsigned)(w.tiles * dy->ld);
    TileWgPlan none;
    memset(&none, 0, sizeof(none));
    int rc = run_wgrad_slabs(p, w.tn, none, prec, s);
    if (rc) return rc;
  }
  {
    const int64_t th = (int64_t)w.co_rows * kld;
    const dim3 grid((unsigned)((th + 255) / 256));
    if (R == 2)
      hipLaunchKernelGGL(wino_wg_out_s2_kernel, grid, dim3(256), 0, s, (const float*)dU, w.tn.splits, w.co_rows, w.tn.Kpad, x->ld,
                         workspace + w.slab_off);
    else if (R == 4)
      hipLaunchKernelGGL(wino_wg_out_kernel<4>, grid, dim3(256), 0, s, (const float*)dU, w.tn.splits, w.co_rows, w.tn.Kpad, x->ld,
                         workspace + w.slab_off);
    else
      hipLaunchKernelGGL(wino_wg_out_kernel<3>, grid, dim3(256), 0, s, (const float*)dU, w.tn.splits, w.co_rows, w.tn.Kpad, x->ld,
                         workspace + w.slab_off);
    ITG_CHECK_LAUNCH();
  }
  return ITG_OK;
}

}  // namespace itgk
