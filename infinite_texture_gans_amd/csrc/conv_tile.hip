// Halo-tile kernels of the narrow stride-1 3x3 layers (forward / input gradient) for gfx950.
#include "conv_common.h"

namespace itgk {

// Persistent workgroups: the filter bank is staged into LDS once per workgroup, then the workgroup walks
// over tiles; the global loads of the NEXT tile are issued into registers before the current tile's MFMA
// phase and drained into the (single) LDS tile buffer after it.
// LDS: Wl[nch][16*FI][20] (K chunk q of the packed panel, 16 k per row) | koff[nch][4] | Xt[TT_PIX][cpt].
// K index k = 16 q + 4 g + e maps to (tap, c) = divmod(k, cin_ld); cin_ld % 4 == 0 keeps the four e of a lane
// in one tap, so lane group g of chunk q reads 16 B at pixel * cpt + koff[q][g].
// NLD = b128 loads per thread and tile = ceil(340 * (cin_ld / 4) / 256): 6 up to cin_ld 16, 11 up to 32
// ST: TS_STATS = also accumulate the consumer BatchNorm's statistics (p.stats); a separate instantiation, so that the plain
// launches keep the register budget and the code they were tuned to.  (Round 3's loader-side BatchNorm - an XF / TS_BNS pair of
// instantiations per kernel - was measured slower twice and removed in round 5.)
enum { TS_NONE = 0, TS_STATS = 1 };

template <int FI, int NLD, int ST>
__global__ __launch_bounds__(256, (NLD <= 6 ? 4 : 3)) void conv_tile_kernel(const ConvP p, int tiles_x, int tiles_y, int ntiles, int cpt, int nch) {
  constexpr bool STATS = ST != TS_NONE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_rows = 16 * FI;
  float* Wl = lds;
  int* koff = reinterpret_cast<int*>(lds + nch * co_rows * 20);
  float* biasl = lds + nch * co_rows * 20 + ((nch * 4 + 3) & ~3);       // [32] bias per output row (zero past out.c)
  double* lstat = reinterpret_cast<double*>(biasl + 32);                // [2][32] BatchNorm sums of this workgroup (p.stats)
  float* Xt = biasl + 32 + 128 + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = p.cin_ld >> 2;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in.p, 0, p.in_bytes, 0x00020000);
  // what a thread fetches is the same for every tile: (halo pixel row / column, channel group) and its LDS slot
  int e_r[NLD], e_c[NLD], e_lds[NLD];
  unsigned e_cb[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + i * 256;
    bool live = e < TT_PIX * q4;
    int pix = live ? e / q4 : 0, c4 = live ? e - pix * q4 : 0;
    e_r[i] = live ? pix / (TT_W + 2) : -1;
    e_c[i] = pix - (pix / (TT_W + 2)) * (TT_W + 2);
    e_lds[i] = pix * cpt + c4 * 4;
    e_cb[i] = (unsigned)c4 * 16u;
  }
  // ---- filter bank + K-chunk offset table (once)
  for (int e = tid; e < nch * co_rows * 16; e += 256) {
    int k16 = e & 15, r = e >> 4;
    int row = r % co_rows, q = r / co_rows;
    int k = q * 16 + k16;
    float v = (row < p.co_rows && k < p.Kpad) ? p.w[(size_t)row * p.Kpad + k] : 0.f;
    Wl[(q * co_rows + row) * 20 + k16] = v;
  }
  for (int e = tid; e < nch * 4; e += 256) {
    int k = (e >> 2) * 16 + (e & 3) * 4;
    int tap = k / p.cin_ld, c = k - tap * p.cin_ld;
    int ky = tap / 3, kx = tap - ky * 3;
    koff[e] = tap < 9 ? (ky * (TT_W + 2) + kx) * cpt + c : 0;      // K padding: weights are zero, read something finite
  }
  // ---- tile loader (global -> registers -> LDS)
  f32x4 rt[NLD];
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int y0 = ty_i * TT_H + p.ioy, x0 = tx_i * TT_W + p.iox;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int iy = y0 + e_r[i], ix = x0 + e_c[i];
      bool ok = e_r[i] >= 0;
      if (p.pad_mode != ITG_PAD_REPLICATE) ok = ok && (unsigned)iy < (unsigned)p.in.H && (unsigned)ix < (unsigned)p.in.W;
      iy = min(max(iy, 0), p.in.H - 1); ix = min(max(ix, 0), p.in.W - 1);
      unsigned o = (unsigned)grid_off(p.in, n, iy, ix) * 4u + e_cb[i];
      rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? o : p.in_bytes, 0, 0));
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (e_r[i] < 0) continue;
      *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = rt[i];
    }
  };
  const int fj = lane & 15, g = lane >> 4;
  int pbase[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) pbase[f] = ((2 * wave + (f >> 1)) * (TT_W + 2) + 16 * (f & 1) + fj) * cpt;
  // per-workgroup constants of the epilogue
  const float osc = p.scale ? *p.scale : 1.f;
  if (tid < 32) biasl[tid] = (p.bias && tid < p.out.c) ? p.bias[tid] : 0.f;
  if (STATS && tid < 64) lstat[tid] = 0.0;
  const bool has_res = p.res.p != nullptr;
  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  __syncthreads();                                    // Wl / koff visible
  for (; tile < ntiles; tile += gridDim.x) {
    store_tile();
    __syncthreads();
    constexpr bool PRE = FI == 1;                     // wider tiles have no registers to hold the residual across the MFMA phase
    f32x4 resv[PRE ? FI : 1][4];
    if (PRE && has_res) {                             // this tile's residual values, requested ahead of the prefetch
      int b = tile;
      const int tx_i = b % tiles_x; b /= tiles_x;
      const int ty_i = b % tiles_y;
      const int n = b / tiles_y;
      const int t0 = ty_i * TT_H, u0 = tx_i * TT_W;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int t = min(t0 + 2 * wave + (f >> 1), p.MT - 1), u = min(u0 + 16 * (f & 1) + fj, p.MU - 1);
        int oy = t * p.osy + p.ooy, ox = u * p.osx + p.oox;
        if (p.out_mode == 1) { oy = min(max(oy, 0), p.out.H - 1); ox = min(max(ox, 0), p.out.W - 1); }
        const float* rp = p.res.p + grid_off(p.res, n, oy >> p.res_ups, ox >> p.res_ups) + g * 4;
#pragma unroll
        for (int i = 0; i < FI; ++i)
          resv[i][f] = (16 * i + g * 4 < p.res.ld) ? *reinterpret_cast<const f32x4*>(rp + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const int next = tile + gridDim.x;
    if (next < ntiles) load_tile(next);               // in flight during the MFMA phase AND the epilogue
    f32x4 acc[FI][4];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nch; ++q) {
      const int ko = koff[q * 4 + g];
      f32x4 a[FI], bq[4];
#pragma unroll
      for (int i = 0; i < FI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Wl + (q * co_rows + 16 * i + fj) * 20 + g * 4);
#pragma unroll
      for (int f = 0; f < 4; ++f) bq[f] = *reinterpret_cast<const f32x4*>(Xt + pbase[f] + ko);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int f = 0; f < 4; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], bq[f][s], acc[i][f], 0, 0, 0);
    }
    // ---- epilogue
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int t0 = ty_i * TT_H, u0 = tx_i * TT_W;
    f32x4 ts1[STATS ? FI : 1], ts2[STATS ? FI : 1];   // this tile's BatchNorm partial sums (p.stats)
    if constexpr (STATS) {
#pragma unroll
      for (int i = 0; i < FI; ++i) { ts1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ts2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      int t = t0 + 2 * wave + (f >> 1), u = u0 + 16 * (f & 1) + fj;
      if (t >= p.MT || u >= p.MU) continue;
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        int co = 16 * i + g * 4;
        if (co >= p.out.ld) continue;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) {
          r = resv[i][f];
        } else if (has_res) {
          int oy = t * p.osy + p.ooy, ox = u * p.osx + p.oox;
          if (p.out_mode == 1) { oy = min(max(oy, 0), p.out.H - 1); ox = min(max(ox, 0), p.out.W - 1); }
          r = *reinterpret_cast<const f32x4*>(p.res.p + grid_off(p.res, n, oy >> p.res_ups, ox >> p.res_ups) + co);
        }
        const f32x4 v = store_out(p, n, t * p.osy + p.ooy, u * p.osx + p.oox, co, acc[i][f], osc,
                                  *reinterpret_cast<const f32x4*>(biasl + co), has_res, r);
        if constexpr (ST == TS_STATS) { ts1[i] += v; ts2[i] += v * v; }
      }
    }
    if constexpr (STATS) {      // pixel lanes -> one lane per channel group -> fp64 in LDS
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            ts1[i][e] += __shfl_xor(ts1[i][e], o, 64);
            ts2[i][e] += __shfl_xor(ts2[i][e], o, 64);
          }
          if (fj == 0) {
            atomicAdd(&lstat[16 * i + g * 4 + e], (double)ts1[i][e]);
            atomicAdd(&lstat[32 + 16 * i + g * 4 + e], (double)ts2[i][e]);
          }
        }
    }
    __syncthreads();                                  // every wave is done reading Xt
  }
  if constexpr (STATS) {
    __syncthreads();
    double* const so = p.stats;
    if (tid < 32 && tid < p.out.ld) {
      atomicAdd(&so[tid], lstat[tid]);
      atomicAdd(&so[p.out.ld + tid], lstat[32 + tid]);
    }
  }
}

// ------------------------------------------------------------------------------- thin 3x3 convs on the vector ALU
// Stride-1 3x3 convolutions whose (padded input channels) x (padded output channels) is at most 64: the generator's
// `final` layer (13 -> 3, tanh; reference models/generators.py:83,119-121) and its input gradient (3 -> 13).  On the
// MFMA kernels such a layer fills 3 of 16 rows (or 4 of 16 K lanes): 67 us / 44 us for 94 MB of traffic.  Here a
// workgroup stages one (8+2) x (32+2) halo tile in LDS, every thread owns ONE output pixel and all its output
// channels, reads its 9 neighbours as 16-byte LDS vectors and takes the filter taps through the scalar cache
// (wave-uniform addresses -> s_load), i.e. <= 576 v_fma per pixel and nothing else in the loop: HBM-bound.
template <int CI4, int CO4>
__global__ __launch_bounds__(256, 5) void conv_valu_kernel(const ConvP p, int tiles_x, int tiles_y, int cpt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xt = lds;
  constexpr int NLD = (TT_PIX * CI4 + 255) / 256;
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx_i = b % tiles_x; b /= tiles_x;
  const int ty_i = b % tiles_y;
  const int n = b / tiles_y;
  const int t0 = ty_i * TT_H, u0 = tx_i * TT_W;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in.p, 0, p.in_bytes, 0x00020000);
  const int y0 = t0 + p.ioy, x0 = u0 + p.iox;
  f32x4 rt[NLD];                                           // every load of the tile in flight before the first LDS store
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = min(tid + i * 256, TT_PIX * CI4 - 1);
    const int pix = e / CI4, c4 = e - pix * CI4;
    const int r = pix / (TT_W + 2), c = pix - r * (TT_W + 2);
    int iy = y0 + r, ix = x0 + c;
    bool ok = true;
    if (p.pad_mode != ITG_PAD_REPLICATE) ok = (unsigned)iy < (unsigned)p.in.H && (unsigned)ix < (unsigned)p.in.W;
    iy = min(max(iy, 0), p.in.H - 1); ix = min(max(ix, 0), p.in.W - 1);
    const unsigned o = (unsigned)grid_off(p.in, n, iy, ix) * 4u + (unsigned)c4 * 16u;
    rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? o : p.in_bytes, 0, 0));
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + i * 256;
    if (e < TT_PIX * CI4) {
      const int pix = e / CI4, c4 = e - pix * CI4;
      *reinterpret_cast<f32x4*>(Xt + pix * cpt + c4 * 4) = rt[i];
    }
  }
  __syncthreads();
  const int ty = tid >> 5, tx = tid & 31;
  f32x4 acc[CO4];
#pragma unroll
  for (int o = 0; o < CO4; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ w = p.w;                      // [16 rows (co)][Kpad], k = tap * cin_ld + c
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const float* xp = Xt + ((ty + ky) * (TT_W + 2) + tx + kx) * cpt;
    f32x4 xv[CI4];
#pragma unroll
    for (int c4 = 0; c4 < CI4; ++c4) xv[c4] = *reinterpret_cast<const f32x4*>(xp + c4 * 4);
#pragma unroll
    for (int co = 0; co < CO4 * 4; ++co) {
      const float* wr = w + co * p.Kpad + tap * (CI4 * 4);  // wave-uniform: scalar loads
      float a = acc[co >> 2][co & 3];
#pragma unroll
      for (int c4 = 0; c4 < CI4; ++c4) {
        a = fmaf(xv[c4][0], wr[c4 * 4 + 0], a);
        a = fmaf(xv[c4][1], wr[c4 * 4 + 1], a);
        a = fmaf(xv[c4][2], wr[c4 * 4 + 2], a);
        a = fmaf(xv[c4][3], wr[c4 * 4 + 3], a);
      }
      acc[co >> 2][co & 3] = a;
    }
  }
  const int t = t0 + ty, u = u0 + tx;
  if (t >= p.MT || u >= p.MU) return;
  const float osc = p.scale ? *p.scale : 1.f;
  const bool has_res = p.res.p != nullptr;
  int oy = t * p.osy + p.ooy, ox = u * p.osx + p.oox;
  int ry = oy, rx = ox;
  if (p.out_mode == 1) { ry = min(max(oy, 0), p.out.H - 1); rx = min(max(ox, 0), p.out.W - 1); }
#pragma unroll
  for (int o = 0; o < CO4; ++o) {
    if (o * 4 >= p.out.ld) continue;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (o * 4 + e < p.out.c) bv[e] = p.bias[o * 4 + e];
    }
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (has_res) r = *reinterpret_cast<const f32x4*>(p.res.p + grid_off(p.res, n, ry >> p.res_ups, rx >> p.res_ups) + o * 4);
    store_out(p, n, oy, ox, o * 4, acc[o], osc, bv, has_res, r);
  }
}

// eligibility + launch of the vector-ALU kernel; returns 1 when it handled the call
int try_conv_valu(const ConvP& p, hipStream_t s, int* rc) {
  const int enable = kernel_on(KM_VALU);
  if (!enable || p.ncls > 1 || p.ntaps != 9 || p.kw != 3 || p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1) return 0;
  if (p.prec != ITG_PREC_F32 || p.stats || p.co_rows != 16) return 0;
  const int ci4 = p.cin_ld >> 2, co4 = p.out.ld >> 2;
  // (4 input groups, 1 output group) = the forward of `final`: 44 us against 68 us on the halo-tile MFMA kernel; its
  // input gradient (1, 4) measured slower here (61 vs 41 us) and stays on the tile kernel
  if (!(ci4 == 4 && co4 == 1)) return 0;
  if ((int64_t)p.MT * p.MU < 64 * 64) return 0;
  ConvP q = p;
  int64_t ib = (int64_t)p.in.n * p.in.gh * p.in.gw * p.in.ph * p.in.pw * p.in.ld * 4;
  if (ib >= 0xFFFF0000LL) return 0;
  q.in_bytes = (unsigned)ib;
  const int cpt = (p.cin_ld % 8 == 4) ? p.cin_ld : p.cin_ld + 4;
  const int tiles_x = (p.MU + TT_W - 1) / TT_W, tiles_y = (p.MT + TT_H - 1) / TT_H;
  const int64_t ntiles = (int64_t)p.in.n * tiles_x * tiles_y;
  if (ntiles > 0x7fffffff) return 0;
  const size_t lds = (size_t)TT_PIX * cpt * sizeof(float);
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_valu_kernel<%d, %d>", ci4, co4);
  const dim3 grid((unsigned)ntiles);
  if (ci4 == 4) hipLaunchKernelGGL((conv_valu_kernel<4, 1>), grid, dim3(256), lds, s, q, tiles_x, tiles_y, cpt);
  else if (co4 == 4) hipLaunchKernelGGL((conv_valu_kernel<1, 4>), grid, dim3(256), lds, s, q, tiles_x, tiles_y, cpt);
  else hipLaunchKernelGGL((conv_valu_kernel<1, 1>), grid, dim3(256), lds, s, q, tiles_x, tiles_y, cpt);
  *rc = hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
  return 1;
}

// ------------------------------------------------------------------------------- 4x4 stride-2 convs with <= 4 input channels
// The discriminator's first layer (reference models/discriminators.py:187-189: conv4x4(img_ch = 3 -> 64, stride 2, pad 1)
// on the 384^2 fakes / 192^2 reals).  As an implicit GEMM it has K = 48: three pipeline stages, all prologue and epilogue -
// 60 us for 94 MB of traffic (0.20 of HBM).  Same recipe as conv_tile_kernel: persistent workgroups, the 16 x cin_ld filter
// bank in LDS, an 8 x 32 OUTPUT tile whose (2*8+2) x (2*32+2) input window (4 floats per pixel: 19 KB) is staged once and
// prefetched a tile ahead; K chunk q = kernel row q, lane group g = kernel column g, so a fragment is one ds_read_b128 at
// (2 y + q, 2 x + g).  Zero padding, bias + 1/sigma + activation in the epilogue.
constexpr int S2_IH = 2 * TT_H + 2, S2_IW = 2 * TT_W + 2, S2_PIX = S2_IH * S2_IW;     // 18 x 66
constexpr int S2_NLD = (S2_PIX + 255) / 256;                                          // 5 loads per thread and tile

// NW = waves per workgroup: 4 (two output rows of the tile per wave) or 8 (one row per wave: twice the waves behind the same
// LDS tile - the tile loop is latency-bound, more workgroups / waves per CU is what speeds it up)
template <int FI, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : 3)) void conv_s2k4_kernel(const ConvP p, int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int co_rows = 16 * FI;
  float* Wl = lds;                                   // [4][co_rows][20]
  float* biasl = lds + 4 * co_rows * 20;             // [co_rows]
  float* Xt = biasl + co_rows;                       // [S2_PIX][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in.p, 0, p.in_bytes, 0x00020000);
  for (int e = tid; e < 4 * co_rows * 16; e += 64 * NW) {
    const int k16 = e & 15, r = e >> 4;
    const int row = r % co_rows, q = r / co_rows;
    const int k = q * 16 + k16;
    Wl[(q * co_rows + row) * 20 + k16] = (row < p.co_rows && k < p.Kpad) ? p.w[(size_t)row * p.Kpad + k] : 0.f;
  }
  for (int e = tid; e < co_rows; e += 64 * NW) biasl[e] = (p.bias && e < p.out.c) ? p.bias[e] : 0.f;
  constexpr int NT = 64 * NW, NLDW = (S2_PIX + NT - 1) / NT, NF = 16 / NW;      // threads, loads per thread and tile, fragments per wave
  f32x4 rt[NLDW];
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int y0 = 2 * ty_i * TT_H - 1, x0 = 2 * tx_i * TT_W - 1;
#pragma unroll
    for (int i = 0; i < NLDW; ++i) {
      const int e = tid + i * NT;
      const int r = e / S2_IW, c = e - r * S2_IW;
      const int iy = y0 + r, ix = x0 + c;
      const bool ok = e < S2_PIX && (unsigned)iy < (unsigned)p.in.H && (unsigned)ix < (unsigned)p.in.W;
      const unsigned o = ok ? (unsigned)grid_off(p.in, n, iy, ix) * 4u : p.in_bytes;
      rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, o, 0, 0));
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NLDW; ++i) {
      const int e = tid + i * NT;
      if (e < S2_PIX) *reinterpret_cast<f32x4*>(Xt + e * 4) = rt[i];
    }
  };
  const int fj = lane & 15, g = lane >> 4;
  // fragment f of wave w: tile row (NF * w + f) / 2, column half (NF * w + f) % 2
  int pbase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) pbase[f] = (2 * ((NF * wave + f) >> 1) * S2_IW + 2 * (16 * ((NF * wave + f) & 1) + fj) + g) * 4;
  const float osc = p.scale ? *p.scale : 1.f;
  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    store_tile();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < ntiles) load_tile(next);
    f32x4 acc[FI][NF];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // kernel row q: 16 k = 4 columns x 4 channels
      f32x4 a[FI], bq[NF];
#pragma unroll
      for (int i = 0; i < FI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Wl + (q * co_rows + 16 * i + fj) * 20 + g * 4);
#pragma unroll
      for (int f = 0; f < NF; ++f) bq[f] = *reinterpret_cast<const f32x4*>(Xt + pbase[f] + q * S2_IW * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int f = 0; f < NF; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], bq[f][s], acc[i][f], 0, 0, 0);
    }
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int t = ty_i * TT_H + ((NF * wave + f) >> 1), u = tx_i * TT_W + 16 * ((NF * wave + f) & 1) + fj;
      if (t >= p.MT || u >= p.MU) continue;
      float* const dst = p.out.p + grid_off(p.out, n, t, u);
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const int co = 16 * i + g * 4;
        if (co >= p.out.ld) continue;
        f32x4 v = acc[i][f] * osc + *reinterpret_cast<const f32x4*>(biasl + co);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = act_apply(v[e], p.act, p.slope);
          if (co + e >= p.out.c) v[e] = 0.f;
        }
        *reinterpret_cast<f32x4*>(dst + co) = v;
      }
    }
    __syncthreads();                                  // every wave is done reading Xt
  }
}

// ------------------------------------------------------------------------------- folded-upsample forward (halo tiles)
// conv3x3(nearest x2 (x)) for the narrow layers behind an upsample (itg_conv_geom.up2; the generator's last blocks): a
// workgroup stages one (8+2) x (32+2) halo tile of the HALF-SIZE tensor and produces the 16 x 64 output pixels above it as
// four parity classes (ry, rx) of 2 x 2 taps: class tap (jy, jx) of output (2t + ry, 2u + rx) reads source pixel
// (t + ry - 1 + jy, u + rx - 1 + jx), i.e. halo pixel (lt + ry + jy, lu + rx + jx).  Four filter banks (one per class, K = 4 taps x
// cin_ld) stay in LDS; per tile 4 x nch K chunks instead of the 4 x (9/4) nch of the unfolded tile kernel on four times the
// tiles, and the source tensor is read once at a quarter of the size.  Same persistent structure as conv_tile_kernel.
template <int FI, int NLD, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_up2_tile_kernel(const ConvP p, int tiles_x, int tiles_y, int ntiles, int cpt, int nch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int co_rows = 16 * FI;
  float* Wl = lds;                                                       // [4][nch][co_rows][20]
  int* koff = reinterpret_cast<int*>(lds + 4 * nch * co_rows * 20);      // [4][nch][4]
  float* biasl = lds + 4 * nch * co_rows * 20 + 16 * nch;                // [32]
  double* lstat = reinterpret_cast<double*>(biasl + 32);                 // [2][32]
  float* Xt = biasl + 32 + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = p.cin_ld >> 2;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in.p, 0, p.in_bytes, 0x00020000);
  int e_r[NLD], e_c[NLD], e_lds[NLD];
  unsigned e_cb[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + i * 256;
    bool live = e < TT_PIX * q4;
    int pix = live ? e / q4 : 0, c4 = live ? e - pix * q4 : 0;
    e_r[i] = live ? pix / (TT_W + 2) : -1;
    e_c[i] = pix - (pix / (TT_W + 2)) * (TT_W + 2);
    e_lds[i] = pix * cpt + c4 * 4;
    e_cb[i] = (unsigned)c4 * 16u;
  }
  for (int e = tid; e < 4 * nch * co_rows * 16; e += 256) {
    const int k16 = e & 15, r = e >> 4;
    const int row = r % co_rows, qc = r / co_rows;                       // qc = cls * nch + q
    const int cls = qc / nch, q = qc - cls * nch;
    const int k = q * 16 + k16;
    Wl[(qc * co_rows + row) * 20 + k16] = k < p.Kpad ? p.w[((size_t)cls * co_rows + row) * p.Kpad + k] : 0.f;
  }
  for (int e = tid; e < 4 * nch * 4; e += 256) {
    const int cls = e / (nch * 4), r = e - cls * nch * 4;
    const int k = (r >> 2) * 16 + (r & 3) * 4;
    const int tap = k / p.cin_ld, c = k - tap * p.cin_ld;
    const int ry = cls >> 1, rx = cls & 1, jy = tap >> 1, jx = tap & 1;
    koff[e] = tap < 4 ? ((ry + jy) * (TT_W + 2) + rx + jx) * cpt + c : 0;     // K padding: weights are zero, read something finite
  }
  f32x4 rt[NLD];
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int y0 = ty_i * TT_H - 1, x0 = tx_i * TT_W - 1;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int iy = y0 + e_r[i], ix = x0 + e_c[i];
      bool ok = e_r[i] >= 0;
      if (p.pad_mode != ITG_PAD_REPLICATE) ok = ok && (unsigned)iy < (unsigned)p.in.H && (unsigned)ix < (unsigned)p.in.W;
      iy = min(max(iy, 0), p.in.H - 1); ix = min(max(ix, 0), p.in.W - 1);
      unsigned o = (unsigned)grid_off(p.in, n, iy, ix) * 4u + e_cb[i];
      rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? o : p.in_bytes, 0, 0));
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (e_r[i] >= 0) *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = rt[i];
  };
  const int fj = lane & 15, g = lane >> 4;
  int pbase[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) pbase[f] = ((2 * wave + (f >> 1)) * (TT_W + 2) + 16 * (f & 1) + fj) * cpt;
  const float osc = p.scale ? *p.scale : 1.f;
  if (tid < 32) biasl[tid] = (p.bias && tid < p.out.c) ? p.bias[tid] : 0.f;
  if (STATS && tid < 64) lstat[tid] = 0.0;
  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  __syncthreads();
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (; tile < ntiles; tile += gridDim.x) {
    store_tile();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < ntiles) load_tile(next);               // in flight during the four classes' MFMA phases and epilogues
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int t0 = ty_i * TT_H, u0 = tx_i * TT_W;
    f32x4 ts1[STATS ? FI : 1], ts2[STATS ? FI : 1];
    if constexpr (STATS) {
#pragma unroll
      for (int i = 0; i < FI; ++i) { ts1[i] = zero4; ts2[i] = zero4; }
    }
#pragma unroll 1
    for (int cls = 0; cls < 4; ++cls) {
      f32x4 acc[FI][4];
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[i][f] = zero4;
      const float* Wc = Wl + cls * nch * co_rows * 20;
      const int* kc = koff + cls * nch * 4;
      for (int q = 0; q < nch; ++q) {
        const int ko = kc[q * 4 + g];
        f32x4 a[FI], bq[4];
#pragma unroll
        for (int i = 0; i < FI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Wc + (q * co_rows + 16 * i + fj) * 20 + g * 4);
#pragma unroll
        for (int f = 0; f < 4; ++f) bq[f] = *reinterpret_cast<const f32x4*>(Xt + pbase[f] + ko);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f)
              acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], bq[f][s], acc[i][f], 0, 0, 0);
      }
      const int ry = cls >> 1, rx = cls & 1;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int t = t0 + 2 * wave + (f >> 1), u = u0 + 16 * (f & 1) + fj;
        if (t >= p.MT || u >= p.MU || 2 * t + ry >= p.out.H || 2 * u + rx >= p.out.W) continue;     // (odd extents: the classes differ in size)
#pragma unroll
        for (int i = 0; i < FI; ++i) {
          const int co = 16 * i + g * 4;
          if (co >= p.out.ld) continue;
          const f32x4 v = store_out(p, n, 2 * t + ry, 2 * u + rx, co, acc[i][f], osc, *reinterpret_cast<const f32x4*>(biasl + co),
                                    false, zero4);
          if constexpr (STATS) { ts1[i] += v; ts2[i] += v * v; }
        }
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            ts1[i][e] += __shfl_xor(ts1[i][e], o, 64);
            ts2[i][e] += __shfl_xor(ts2[i][e], o, 64);
          }
          if (fj == 0) {
            atomicAdd(&lstat[16 * i + g * 4 + e], (double)ts1[i][e]);
            atomicAdd(&lstat[32 + 16 * i + g * 4 + e], (double)ts2[i][e]);
          }
        }
    }
    __syncthreads();                                  // every wave is done reading Xt
  }
  if constexpr (STATS) {
    __syncthreads();
    if (tid < 32 && tid < p.out.ld) {
      atomicAdd(&p.stats[tid], lstat[tid]);
      atomicAdd(&p.stats[p.out.ld + tid], lstat[32 + tid]);
    }
  }
}

// eligibility + launch; p is the 4-class ConvP of itg_conv2d_fwd's up2 branch (class geometry in the c* arrays)
int try_conv_up2_tile(const ConvP& p, hipStream_t s, int* rc) {
  const int enable = kernel_on(KM_UP2_TILE);
  if (!enable || p.ncls != 4 || p.ntaps != 4 || p.kw != 2 || p.cioy[0] != -1 || p.ciox[0] != -1 || p.out_mode != 0) return 0;
  if (p.prec != ITG_PREC_F32 || p.cin_ld > 32 || p.co_rows > 32 || p.res.p) return 0;
  if ((int64_t)p.MT * p.MU < 48 * 48) return 0;
  const int FI = p.co_rows / 16;
  if (FI == 2 && p.stats) return 0;                           // (two row tiles + statistics: register budget; separate pass by the caller)
  ConvP q = p;
  const int64_t ib = (int64_t)p.in.n * p.in.gh * p.in.gw * p.in.ph * p.in.pw * p.in.ld * 4;
  if (ib >= 0xFFFF0000LL) return 0;
  q.in_bytes = (unsigned)ib;
  const int cpt = (p.cin_ld % 8 == 4) ? p.cin_ld : p.cin_ld + 4;
  const int nch = p.Kpad / 16;
  const int tiles_x = (p.MU + TT_W - 1) / TT_W, tiles_y = (p.MT + TT_H - 1) / TT_H;
  const int64_t ntiles = (int64_t)p.in.n * tiles_x * tiles_y;
  const size_t lds = ((size_t)4 * nch * 16 * FI * 20 + 16 * nch + 32 + 128 + (size_t)TT_PIX * cpt) * sizeof(float);
  const int nld = (TT_PIX * (p.cin_ld >> 2) + 255) / 256;
  if (ntiles > 0x7fffffff || lds > 80 * 1024 || nld > 11) return 0;
  const bool st = p.stats != nullptr, small = nld <= 6;
  const void* kern = FI == 1 ? (small ? (st ? (const void*)&conv_up2_tile_kernel<1, 6, true> : (const void*)&conv_up2_tile_kernel<1, 6, false>)
                                      : (st ? (const void*)&conv_up2_tile_kernel<1, 11, true> : (const void*)&conv_up2_tile_kernel<1, 11, false>))
                             : (small ? (const void*)&conv_up2_tile_kernel<2, 6, false> : (const void*)&conv_up2_tile_kernel<2, 11, false>);
  static const void* attr_set[8] = {nullptr};
  {
    bool seen = false;
    int slot = 0;
    for (; slot < 8 && attr_set[slot]; ++slot) seen = seen || attr_set[slot] == kern;
    if (!seen && slot < 8) {
      if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) { (void)hipGetLastError(); return 0; }
      attr_set[slot] = kern;
    }
  }
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  const int64_t want = 256 * (int64_t)per_cu;
  const unsigned blocks = (unsigned)(ntiles < want ? ntiles : want);
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_up2_tile_kernel<%d, %d, %s>", FI, small ? 6 : 11, st ? "true" : "false");
  int a_tx = tiles_x, a_ty = tiles_y, a_nt = (int)ntiles, a_cpt = cpt, a_nch = nch;
  void* args[] = {(void*)&q, &a_tx, &a_ty, &a_nt, &a_cpt, &a_nch};
  (void)hipLaunchKernel(kern, dim3(blocks), dim3(256), args, lds, s);
  *rc = hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
  return 1;
}

int try_conv_s2k4(const ConvP& p, hipStream_t s, int* rc) {
  const int enable = kernel_on(KM_S2K4);
  if (!enable || p.ncls > 1 || p.ntaps != 16 || p.kw != 4 || p.isy != 2 || p.isx != 2 || p.ioy != -1 || p.iox != -1) return 0;
  if (p.osy != 1 || p.osx != 1 || p.ooy != 0 || p.oox != 0 || p.out_mode != 0 || p.pad_mode != ITG_PAD_ZERO) return 0;
  if (p.prec != ITG_PREC_F32 || p.cin_ld != 4 || p.stats || p.res.p) return 0;
  if (p.co_rows != 16 && p.co_rows != 32 && p.co_rows != 64) return 0;
  if ((int64_t)p.MT * p.MU < 64 * 64) return 0;
  ConvP q = p;
  const int64_t ib = (int64_t)p.in.n * p.in.gh * p.in.gw * p.in.ph * p.in.pw * p.in.ld * 4;
  if (ib >= 0xFFFF0000LL) return 0;
  q.in_bytes = (unsigned)ib;
  const int tiles_x = (p.MU + TT_W - 1) / TT_W, tiles_y = (p.MT + TT_H - 1) / TT_H;
  const int64_t ntiles = (int64_t)p.in.n * tiles_x * tiles_y;
  if (ntiles > 0x7fffffff) return 0;
  const int FI = p.co_rows / 16;
  const size_t lds = ((size_t)4 * p.co_rows * 20 + p.co_rows + (size_t)S2_PIX * 4) * sizeof(float);
  constexpr int nw = 8;         // 8 waves per tile: 43.9 -> 41.0 us on D's first layer (1 / 3 / 4 workgroups per CU instead of 2: worse)
  const int64_t want = 256 * 2;
  const unsigned blocks = (unsigned)(ntiles < want ? ntiles : want);
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_s2k4_kernel<%d, %d>", FI, nw);
  if (FI == 4) hipLaunchKernelGGL((conv_s2k4_kernel<4, 8>), dim3(blocks), dim3(512), lds, s, q, tiles_x, tiles_y, (int)ntiles);
  else if (FI == 2) hipLaunchKernelGGL((conv_s2k4_kernel<2, 8>), dim3(blocks), dim3(512), lds, s, q, tiles_x, tiles_y, (int)ntiles);
  else hipLaunchKernelGGL((conv_s2k4_kernel<1, 8>), dim3(blocks), dim3(512), lds, s, q, tiles_x, tiles_y, (int)ntiles);
  *rc = hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
  return 1;
}

// eligibility + launch of the halo-tile kernel; returns 1 when it handled the call
int try_conv_tile(ConvP& p, hipStream_t s, int* rc) {
  const int enable = kernel_on(KM_CONV_TILE);
  if (!enable || p.ncls > 1 || p.ntaps != 9 || p.kw != 3 || p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1)
    return 0;
  if (p.prec != ITG_PREC_F32 || p.cin_ld > 32 || p.co_rows > 32) return 0;
  if ((int64_t)p.MT * p.MU < 64 * 64) return 0;          // tiny images: the gather kernel with split-K wins
  const int FI = p.co_rows / 16;
  if (FI == 2) p.stats = nullptr;   // two row tiles + statistics spill (17-24 VGPRs): the caller runs the separate pass
  ConvP q = p;
  int64_t ib = (int64_t)p.in.n * p.in.gh * p.in.gw * p.in.ph * p.in.pw * p.in.ld * 4;
  if (ib >= 0xFFFF0000LL) return 0;
  q.in_bytes = (unsigned)ib;
  q.scale = p.scale;
  const int cpt = (p.cin_ld % 8 == 4) ? p.cin_ld : p.cin_ld + 4;     // conflict-free b128 fragment reads
  const int nch = (9 * p.cin_ld + 15) / 16;
  const int tiles_x = (p.MU + TT_W - 1) / TT_W, tiles_y = (p.MT + TT_H - 1) / TT_H;
  const int64_t ntiles = (int64_t)p.in.n * tiles_x * tiles_y;
  const size_t lds = ((size_t)nch * 16 * FI * 20 + ((nch * 4 + 3) & ~3) + 32 + 128 + 128 + (size_t)TT_PIX * cpt) * sizeof(float);
  const int nld = (TT_PIX * (p.cin_ld >> 2) + 255) / 256;
  constexpr int lds_cap = 80 * 1024;       // 26 -> 26 channels (b5c2) needs 76 KB: two workgroups per CU
  if (ntiles > 0x7fffffff || lds > (size_t)lds_cap || nld > 11) return 0;
  const int stm = p.stats ? TS_STATS : TS_NONE;
  const bool small = nld <= 6;
  const void* kern = nullptr;
#define ITG_TILE_PICK(FI_, NLD_)                                                                                        \
  kern = stm == TS_STATS ? (const void*)&conv_tile_kernel<FI_, NLD_, TS_STATS> : (const void*)&conv_tile_kernel<FI_, NLD_, TS_NONE>
  if (FI == 1 && small) { ITG_TILE_PICK(1, 6); }
  else if (FI == 1) { ITG_TILE_PICK(1, 11); }
  else if (small) { ITG_TILE_PICK(2, 6); }
  else { ITG_TILE_PICK(2, 11); }
#undef ITG_TILE_PICK
  static const void* attr_set[24] = {nullptr};
  {
    bool seen = false;
    int slot = 0;
    for (; slot < 24 && attr_set[slot]; ++slot) seen = seen || attr_set[slot] == kern;
    if (!seen && slot < 24) {
      (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_set[slot] = kern;
    }
  }
  // persistent grid = what is resident at once (register budget: 4 workgroups per CU with 6 loads, 3 with 11)
  int per_cu = (int)((160 * 1024) / lds);
  const int reg_cu = nld <= 6 ? 4 : 3;
  if (per_cu > reg_cu) per_cu = reg_cu;
  if (per_cu < 1) per_cu = 1;
  const int64_t want = 256 * (int64_t)per_cu;
  const unsigned blocks = (unsigned)(ntiles < want ? ntiles : want);      // (equal tile counts per workgroup measured worse: b6c2 forward 69 -> 76 us)
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_tile_kernel<%d, %d, %d>", FI, nld <= 6 ? 6 : 11, stm);
  int a_tx = tiles_x, a_ty = tiles_y, a_nt = (int)ntiles, a_cpt = cpt, a_nch = nch;
  void* args[] = {(void*)&q, &a_tx, &a_ty, &a_nt, &a_cpt, &a_nch};
  (void)hipLaunchKernel(kern, dim3(blocks), dim3(256), args, lds, s);
  *rc = hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
  return 1;
}


}  // namespace itgk
