// Weight-gradient kernels (pixels are the K dimension) for gfx950.
//   conv_tn_kernel : C[(tap,ci)][co] = sum_pixel X[pixel+tap][ci] * dY[pixel][co], split over pixel ranges (grid.z)
//                    into fp32 slabs, reduced by wgrad_reduce_kernel into the OIHW gradient.
#include "conv_common.h"

namespace itgk {

// BF = false: fp32 operands, v_mfma_f32_16x16x4_f32, fragments read element-wise from pixel-major tiles.
// BF = true : operands rounded to bf16 when staged (pixel-major rows of bf16), fragments fetched with the
//             gfx950 transposing LDS read (ds_read_b64_tr_b16: a 4-pixel x 16-column block arrives
//             column-major, i.e. as the K-contiguous MFMA operand) and contracted 32 pixels at a time by
//             v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
// DEPTH = pixel stages whose global loads are in flight while one stage is computed (1 or 2).
// register budget: 3 workgroups per CU with two stages in flight, 4 with one - except the 256 x 64 tile, whose 16 row x column
// fragments per wave plus 5 prefetch vectors need the 168-register budget in either form
constexpr int tn_min_blocks(int bcol, int wcol, int wco, bool bf, int depth) {
  return depth == 2 ? 3 : (bf ? 2 : ((wcol / 16) * (wco / 16) >= 16 && bcol >= 256 ? 3 : 4));
}

// FLAT: both tensors are plain images (1 x 1 patch grids: the discriminator's layers), zero padding, no parity classes, at
// most 15 taps per column tile and MU >= the pixels of a stage: the offset producer is then straight-line code (no patch
// arithmetic, no clamp, one pass, selects instead of branches) that the scheduler may place between the MFMAs of the stage
// - the generic producer cost 10 % of D's weight-gradient time (timing experiment with the producer switched off after the pipeline had filled: D3 678 -> 608 us).
template <int BCOL, int BCO, int WCOL, int WCO, bool BF, int DEPTH, bool FLAT = false>
__global__ __launch_bounds__(256, tn_min_blocks(BCOL, WCOL, WCO, BF, DEPTH)) void conv_tn_kernel(const WgP p, int otp) {
  constexpr int FI = WCOL / 16, FJ = WCO / 16;
  constexpr int WAVES_COL = BCOL / WCOL;
  static_assert(WAVES_COL * (BCO / WCO) == 4, "4 waves per workgroup");
  constexpr int KP = BF ? 32 : BKP;                  // pixels per stage
  constexpr int LDX = BCOL + 16, LDY = BCO + 16;     // fp32 tiles: row pitch in floats
  constexpr int LHX = BCOL + 8, LHY = BCO + 8;       // bf16 tiles: row pitch in halfwords (8-B aligned rows)
  constexpr int XG = BCOL / 4, YG = BCO / 4;         // float4 groups per pixel row
  constexpr int XL = (KP * XG + 255) / 256, YL = (KP * YG + 255) / 256;
  constexpr bool XFULL = (KP * XG) % 256 == 0, YFULL = (KP * YG) % 256 == 0;      // every thread stages a row in every pass: no row guards
  __shared__ __attribute__((aligned(16))) float smem[2 * BKP * (LDX + LDY)];
  static_assert(2 * 32 * (LHX + LHY) * 2 <= 2 * BKP * (LDX + LDY) * 4, "bf16 tiles fit the fp32 allocation");
  float* Xs = smem;
  float* Ys = smem + 2 * BKP * LDX;
  unsigned short* Xh = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Yh = Xh + 2 * KP * LHX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // up2: the output-parity class is the fastest index of the tile id (the four classes of a tile gather the same source
  // pixels: adjacent workgroups share them through one L2)
  const int ncls = p.up2 ? 4 : 1;
  const int cls = (int)blockIdx.x % ncls;
  const int tile_id = (int)blockIdx.x / ncls;
  const int col_tile = tile_id % p.ncol_tiles;
  const int co_tile = tile_id / p.ncol_tiles;
  const int col0 = col_tile * BCOL, co0 = co_tile * BCO;
  const int wcol0 = (wave % WAVES_COL) * WCOL, wco0 = (wave / WAVES_COL) * WCO;
  const int split = blockIdx.z;
  const int ry = p.up2 ? cls >> 1 : 0, rx = p.up2 ? cls & 1 : 0;
  const int chunk_begin = split * p.chunks_per_split;
  const int chunk_end = min(p.nchunks, chunk_begin + p.chunks_per_split);

  // X loads: thread -> (pixel row xr[i], column group); the column (tap, ci) is fixed, the pixel moves.
  // Raw buffer loads: an offset equal to the buffer size reads zeros (padding, rows past M, columns past K).
  // The bias gradient sum_pixel dY[pixel][co] is accumulated on the side by the col_tile 0 workgroups from
  // the dY values they stage anyway (dbslab[split][co]).
  //
  // Gather addresses come from a per-stage offset table in LDS: otab[stage & 1][pixel row][slot], slot j < nt = byte
  // offset of input pixel (pixel + tap tap_lo + j) incl. padding / validity, slot nt = byte offset of the dY pixel.
  // One thread per (pixel row, slot) - spread over the four waves - tracks its pixel and does the clamp / patch-grid
  // address arithmetic ONCE per stage; a load is then a ds_read + add instead of ~40 VALU per load and stage
  // (the kernel issued 2.3 VALU per MFMA that way and kept the MFMA pipe 58 % busy).
  const int ucl = (int)blockIdx.y;                            // uniform class (WgP.ucls), else 0
  const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x.p + (size_t)ucl * p.u_x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy.p + (size_t)ucl * p.u_dy), 0, p.dy_bytes, 0x00020000);
  extern __shared__ unsigned otab[];                           // [DEPTH + 1][KP][otp], otp = taps of a column tile + 1
  const int OTP = otp;
  const int tap_lo = col0 / p.cin_ld;
  const int col_hi = min(col0 + BCOL, p.Ktot) - 1;
  const int nt = col_hi >= col0 ? min(col_hi / p.cin_ld, p.ntaps - 1) - tap_lo + 1 : 0;
  int xr[XL], xcol[XL], xj[XL];
  unsigned xcb[XL];
  bool xok[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    int idx = tid + i * 256;
    xr[i] = idx / XG;
    int g = idx - xr[i] * XG;
    xcol[i] = g * 4;
    int col = col0 + g * 4;
    int tap = col / p.cin_ld;
    xcb[i] = (unsigned)(col - tap * p.cin_ld) * 4u;
    xj[i] = tap - tap_lo;
    xok[i] = (xr[i] < KP) && (col < p.Ktot);
  }
  int yr[YL], yc[YL];
  unsigned ycb[YL];
  bool yok[YL];
#pragma unroll
  for (int i = 0; i < YL; ++i) {
    int idx = tid + i * 256;
    yr[i] = idx / YG;
    int g = idx - yr[i] * YG;
    yc[i] = g * 4;
    ycb[i] = (unsigned)(co0 + g * 4) * 4u;
    yok[i] = (yr[i] < KP) && (co0 + g * 4 < p.dy.ld);
  }
  // ---- offset producers: entry e = (pixel row e % KP, slot e / KP); lane l of wave w owns e = 4 l + w (+ 256 ...)
  constexpr int PE = (KP * 17 + 255) / 256;                   // producer passes (entries e, e + 256, ...): <= 16 taps + dY
  const int pe0 = lane * 4 + wave;
  const int prow = pe0 % KP;                                  // 256 % KP == 0: every pass of a thread has the same pixel row
  int pky[PE], pkx[PE];
  bool pact[PE], pdy[PE];
#pragma unroll
  for (int i = 0; i < PE; ++i) {
    int j = (pe0 + i * 256) / KP;
    pact[i] = j <= nt;
    pdy[i] = j == nt;
    int tap = tap_lo + min(j, nt > 0 ? nt - 1 : 0);
    pky[i] = tap / p.kw;
    pkx[i] = tap - pky[i] * p.kw;
  }
  int pn, pt_, pu;
  {
    int m = chunk_begin * KP + prow;
    decode_m(m < p.M ? m : 0, p.MT, p.MU, pn, pt_, pu);
    if (m >= p.M) pn = p.x.n;      // marks invalid
  }
  // FLAT: this thread's single entry (pass 0); inactive threads park their store in the spare word behind the table
  const unsigned ld4x = (unsigned)p.x.ld * 4u, ld4y = (unsigned)p.dy.ld * 4u;
  const int fslot = pact[0] ? prow * OTP + pe0 / KP : (DEPTH + 1) * KP * OTP;
  auto produce = [&](int buf) {
    if constexpr (FLAT) {
      const bool live = pn < p.x.n;
      const int iy = pt_ * p.stride - p.pad_h + pky[0], ix = pu * p.stride - p.pad + pkx[0];
      const bool okx = live && (unsigned)iy < (unsigned)p.x.H && (unsigned)ix < (unsigned)p.x.W;
      const unsigned ox = okx ? (unsigned)((pn * p.x.H + iy) * p.x.W + ix) * ld4x : p.x_bytes;
      const unsigned oy = live ? (unsigned)((pn * p.MT + pt_) * p.MU + pu) * ld4y : p.dy_bytes;
      otab[pact[0] ? buf * KP * OTP + fslot : fslot] = pdy[0] ? oy : ox;
      pu += KP;
      const bool w = pu >= p.MU;                                   // MU >= KP: at most one row wrap per stage
      pu -= w ? p.MU : 0;
      pt_ += w ? 1 : 0;
      const bool w2 = pt_ >= p.MT;
      pt_ = w2 ? 0 : pt_;
      pn += w2 ? 1 : 0;
      return;
    }
    const bool live = pn < p.x.n;
#pragma unroll
    for (int i = 0; i < PE; ++i) {
      if (!pact[i]) continue;
      unsigned o;
      if (pdy[i]) {
        o = live ? (unsigned)grid_off(p.dy, pn, (pt_ << p.up2) + ry, (pu << p.up2) + rx) * 4u : p.dy_bytes;
      } else {
        int iy = pt_ * p.stride - p.pad_h + pky[i] + ry, ix = pu * p.stride - p.pad + pkx[i] + rx;
        bool ok = live;
        if (p.pad_mode != ITG_PAD_REPLICATE) ok = ok && (unsigned)iy < (unsigned)p.x.H && (unsigned)ix < (unsigned)p.x.W;
        iy = min(max(iy, 0), p.x.H - 1); ix = min(max(ix, 0), p.x.W - 1);
        o = ok ? (unsigned)grid_off(p.x, live ? pn : 0, iy, ix) * 4u : p.x_bytes;
      }
      otab[(buf * KP + prow) * OTP + (pe0 + i * 256) / KP] = o;
    }
    pu += KP;
    while (pu >= p.MU) { pu -= p.MU; if (++pt_ == p.MT) { pt_ = 0; ++pn; } }
  };

  f32x4 rxs[DEPTH][XL], rys[DEPTH][YL];
  auto load_tiles = [&](int slot, int set, f32x4 (&rx)[XL], f32x4 (&ry)[YL]) {
    {
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        // table read unconditional (clamped index), validity by select: no exec-mask branch around the ds_read
        const unsigned t = otab[(slot * KP + min(xr[i], KP - 1)) * OTP + min(xj[i], OTP - 1)];
        const unsigned o = xok[i] ? t + xcb[i] : p.x_bytes;       // (an invalid pixel's entry is x_bytes: adding the column stays out of range)
        rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxr, o, 0, 0));
      }
    }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
      const unsigned t = otab[(slot * KP + min(yr[i], KP - 1)) * OTP + nt];
      const unsigned o = yok[i] ? t + ycb[i] : p.dy_bytes;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ryr, o, 0, 0));
    }
  };
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
  const bool do_db = p.dbslab != nullptr && col_tile == 0;
  auto store_tiles = [&](int buf, int set, const f32x4 (&rx)[XL], const f32x4 (&ry)[YL]) {
#pragma unroll
    for (int i = 0; i < XL; ++i)
      if (XFULL || xr[i] < KP) {
        f32x4 v = rx[i];
        if constexpr (BF) *reinterpret_cast<uint2*>(Xh + (buf * KP + xr[i]) * LHX + xcol[i]) = pack_bf16x4(v);
        else *reinterpret_cast<f32x4*>(Xs + (buf * KP + xr[i]) * LDX + xcol[i]) = v;
      }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
      if (YFULL || yr[i] < KP) {
        if constexpr (BF) *reinterpret_cast<uint2*>(Yh + (buf * KP + yr[i]) * LHY + yc[i]) = pack_bf16x4(ry[i]);
        else *reinterpret_cast<f32x4*>(Ys + (buf * KP + yr[i]) * LDY + yc[i]) = ry[i];
      }
      if (do_db) dbacc += ry[i];
    }
  };

  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = chunk_end - chunk_begin;
  const int fr = lane & 15, fkk = lane >> 4;
  // transposing read: lane 4q+pp of 16-lane group g addresses pixel row 8g+q (then 8g+4+q), columns 4pp..4pp+3
  const int trq = (lane & 15) >> 2, trp = lane & 3;
  auto compute = [&](int buf) {
      if constexpr (BF) {
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      bf16x8 a[FI], b[FJ];
      const unsigned short* xb = Xh + (buf * KP + 8 * fkk + trq) * LHX + wcol0 + 4 * trp;
      const unsigned short* yb = Yh + (buf * KP + 8 * fkk + trq) * LHY + wco0 + 4 * trp;
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xb + 16 * i));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xb + 16 * i + 4 * LHX));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb + 16 * j));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb + 16 * j + 4 * LHY));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        b[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float a[FI], b[FJ];
        const float* xrow = Xs + (buf * KP + 4 * s + fkk) * LDX + wcol0 + fr;
        const float* yrow = Ys + (buf * KP + 4 * s + fkk) * LDY + wco0 + fr;
#pragma unroll
        for (int i = 0; i < FI; ++i) a[i] = xrow[16 * i];
#pragma unroll
        for (int j = 0; j < FJ; ++j) b[j] = yrow[16 * j];
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int j = 0; j < FJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  if (nk > 0) {
    if constexpr (DEPTH == 1) {
      produce(0);
      if (nk > 1) produce(1);
      __syncthreads();
      load_tiles(0, 0, rxs[0], rys[0]);
      store_tiles(0, 0, rxs[0], rys[0]);
      __syncthreads();
      for (int kk = 0; kk < nk; ++kk) {
        const int buf = kk & 1;
        if (kk + 1 < nk) load_tiles(buf ^ 1, 0, rxs[0], rys[0]);   // table of stage kk + 1: written one barrier ago
        if (FLAT || kk + 2 < nk) produce(buf);                   // stage kk + 2 -> the slot stage kk's loads have finished with
        compute(buf);
        if (kk + 1 < nk) store_tiles(buf ^ 1, 0, rxs[0], rys[0]);
        __syncthreads();
      }
    } else {
      // two stages in flight: register set A holds stage kk + 2 while set B (stage kk + 1) drains into LDS; the offset
      // table runs three stages ahead in a ring of three slots (slot of stage s = s % 3)
      produce(0);
      if (nk > 1) produce(1);
      if (nk > 2) produce(2);
      __syncthreads();
      load_tiles(0, 0, rxs[0], rys[0]);
      if (nk > 1) load_tiles(1, DEPTH - 1, rxs[DEPTH - 1], rys[DEPTH - 1]);
      store_tiles(0, 0, rxs[0], rys[0]);
      __syncthreads();
      int s0 = 0;                                                // kk % 3
      for (int kk = 0; kk < nk; kk += 2) {
        const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        if (kk + 2 < nk) load_tiles(s2, 0, rxs[0], rys[0]);
        if (FLAT || kk + 3 < nk) produce(s0);                    // stage kk + 3 (FLAT: unconditional - entries past the last stage are never read)
        compute(0);
        if (kk + 1 < nk) store_tiles(1, DEPTH - 1, rxs[DEPTH - 1], rys[DEPTH - 1]);
        __syncthreads();
        if (kk + 1 >= nk) break;
        if (kk + 3 < nk) load_tiles(s0, DEPTH - 1, rxs[DEPTH - 1], rys[DEPTH - 1]);
        if (FLAT || kk + 4 < nk) produce(s1);                    // stage kk + 4
        compute(1);
        if (kk + 2 < nk) store_tiles(0, 0, rxs[0], rys[0]);
        __syncthreads();
        s0 = s2;                                                 // (kk + 2) % 3
      }
    }
  }
  if (do_db) {   // deterministic reduction of the per-thread dY sums over the staged pixel rows
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    red[tid] = dbacc;
    __syncthreads();
    if (tid < BCO && co0 + tid < p.co_rows) {
      // thread t staged column group (t % YG) in each of its YL passes: every thread with that group holds
      // a partial of the same 4 channels
      float sdb = 0.f;
      for (int r = (tid >> 2); r < 256; r += YG) sdb += red[r][tid & 3];
      p.dbslab[((size_t)split * ncls + cls) * p.co_rows + co0 + tid] = sdb;
    }
  }
  // D[row = column index (4 consecutive per lane)][col = co]
  float* slab = p.slab + (((size_t)split * ncls + cls) * (p.ucls ? p.ucls : 1) + ucl) * p.co_rows * p.Kpad;
  const int cq = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    int co = co0 + wco0 + 16 * j + (lane & 15);
    if (co >= p.co_rows) continue;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      int col = col0 + wcol0 + 16 * i + cq;
      if (col >= p.Kpad) continue;
      *reinterpret_cast<f32x4*>(slab + (size_t)co * p.Kpad + col) = acc[i][j];
    }
  }
}


// ------------------------------------------------------------------------------- narrow 3x3 weight gradient (halo tiles)
// dW[(tap, c)][co] = sum_pixel X[pixel + tap][c] * dY[pixel][co] for stride-1 3x3 convs with <= 32 input and <= 16
// output channels (the generator's last block and `final`).  The generic kernel above spends most of its
// issue slots on gather addresses (16 MFMAs per 16-pixel stage); here a persistent workgroup stages an
// (8+2) x (32+2) halo tile of X and the 8 x 32 tile of dY in LDS and every wave contracts its 64 pixels
// against ALL 9 * cin_ld (tap, c) rows: per 4 pixels MF ds_read_b32 + 1 and MF MFMAs, no address arithmetic.
// The 4 waves' accumulators are summed in a fixed order through LDS; one slab per workgroup, reduced by the
// same two-stage reduction as the generic path.
template <int NJ, int NLD>
__global__ __launch_bounds__(256, 2) void wgrad_tile_kernel(const WgP p, int tiles_x, int tiles_y, int ntiles, int cpt, int coef_off) {
  constexpr int MF = 4 * NJ;                               // 16-row MFMA tiles of the (tap, c) dimension
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CPD = 16;                                  // dY tile pitch (co_rows = 16)
  float* Xt = lds;                                         // [TT_PIX][cpt]
  float* Yt = lds + TT_PIX * cpt;                          // [TT_H * TT_W][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = p.cin_ld >> 2, yq4 = p.dy.ld >> 2;
  const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x.p, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy.p, 0, p.dy_bytes, 0x00020000);
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
  // dY: thread -> (pixel, channel group); yq4 in {1, 2, 4} divides 256, so a thread's channel group is fixed
  constexpr int YLD = 4;
  int y_r[YLD], y_c[YLD], y_lds[YLD];
  const unsigned y_cb = (unsigned)(tid % yq4) * 16u;
#pragma unroll
  for (int i = 0; i < YLD; ++i) {
    int e = tid + i * 256;
    bool live = e < TT_H * TT_W * yq4;
    int pix = live ? e / yq4 : 0;
    y_r[i] = live ? pix / TT_W : -1;
    y_c[i] = pix - (pix / TT_W) * TT_W;
    y_lds[i] = pix * CPD + (e % yq4) * 4;
  }
  for (int e = tid; e < TT_H * TT_W * CPD; e += 256) Yt[e] = 0.f;     // channel groups >= dy.ld stay zero
  f32x4 rt[NLD], ry[YLD];
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int t0 = ty_i * TT_H, u0 = tx_i * TT_W;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int iy = t0 - p.pad_h + e_r[i], ix = u0 - p.pad + e_c[i];
      bool ok = e_r[i] >= 0;
      if (p.pad_mode != ITG_PAD_REPLICATE) ok = ok && (unsigned)iy < (unsigned)p.x.H && (unsigned)ix < (unsigned)p.x.W;
      iy = min(max(iy, 0), p.x.H - 1); ix = min(max(ix, 0), p.x.W - 1);
      unsigned o = (unsigned)grid_off(p.x, n, iy, ix) * 4u + e_cb[i];
      rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxr, ok ? o : p.x_bytes, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < YLD; ++i) {
      int t = t0 + y_r[i], u = u0 + y_c[i];
      bool ok = y_r[i] >= 0 && t < p.MT && u < p.MU;
      unsigned o = (unsigned)grid_off(p.dy, n, ok ? t : 0, ok ? u : 0) * 4u + y_cb;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ryr, ok ? o : p.dy_bytes, 0, 0));
    }
  };
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (e_r[i] < 0) continue;
      f32x4 v = rt[i];
      *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = v;
    }
#pragma unroll
    for (int i = 0; i < YLD; ++i)
      if (y_r[i] >= 0) { *reinterpret_cast<f32x4*>(Yt + y_lds[i]) = ry[i]; dbacc += ry[i]; }
  };
  // MFMA rows.  A lane's ds_read_b128 of pixel (x + g) at (tap, channels 4 c4 .. 4 c4 + 3) feeds FOUR row tiles at
  // once: row fr of tile 4 j + e is (tap, c = 4 c4 + e) with (tap, c4) = divmod(16 j + fr, cin_ld / 4) - a permutation
  // of the (tap, c) rows that the epilogue undoes.  (One ds_read_b32 per MFMA before: the kernel was LDS-latency bound
  // at a quarter of the MFMA rate.)
  const int fr = lane & 15, g = lane >> 4;
  const int nq = 9 * q4;
  int qoff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int q = min(16 * j + fr, nq - 1);                      // rows past the last (tap, c4) group: computed, never stored
    int tap = q / q4, c4 = q - tap * q4;
    int ky = tap / 3, kx = tap - ky * 3;
    qoff[j] = (ky * (TT_W + 2) + kx) * cpt + 4 * c4;
  }
  f32x4 acc[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    store_tile();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < ntiles) load_tile(next);
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const float* xrow = Xt + ((2 * wave + rr) * (TT_W + 2) + g) * cpt;
      const float* yrow = Yt + ((2 * wave + rr) * TT_W + g) * CPD + fr;
#pragma unroll 2
      for (int s4 = 0; s4 < TT_W / 4; ++s4) {
        const float bv = yrow[s4 * 4 * CPD];
        const float* xs = xrow + s4 * 4 * cpt;
        f32x4 av[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) av[j] = *reinterpret_cast<const f32x4*>(xs + qoff[j]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[4 * j + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], bv, acc[4 * j + e], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- sum the 4 waves' accumulators in wave order through LDS: R[m = tap * cin_ld + c][16]
  float* R = lds;
  int rrow[NJ][4];                                         // R row of D row 4 g + e of the tiles 4 j .. 4 j + 3 (their c differs by the tile)
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int q = 16 * j + 4 * g + e;
      int tap = q / q4, c4 = q - tap * q4;
      rrow[j][e] = q < nq ? tap * p.cin_ld + 4 * c4 : -1;
    }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (rrow[j][e] < 0) continue;
            float* dst = R + (rrow[j][e] + t) * 16 + fr;
            *dst = (w == 0 ? 0.f : *dst) + acc[4 * j + t][e];
          }
    }
    __syncthreads();
  }
  float* slab = p.slab + (size_t)blockIdx.x * p.co_rows * p.Kpad;
  const int mrows = 9 * p.cin_ld;
  for (int idx = tid; idx < p.co_rows * p.Kpad; idx += 256) {
    int co = idx / p.Kpad, m = idx - co * p.Kpad;
    slab[idx] = m < mrows ? R[m * 16 + co] : 0.f;
  }
  if (p.dbslab) {       // bias gradient: per-thread sums of the staged dY rows -> fixed-order sum per channel
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds) + (9 * 32 * 16 + 3) / 4;
    red[tid] = dbacc;
    __syncthreads();
    if (tid < 16) {
      float sdb = 0.f;
      if ((tid >> 2) < yq4)
        for (int r = (tid >> 2); r < 256; r += yq4) sdb += red[r][tid & 3];
      p.dbslab[(size_t)blockIdx.x * p.co_rows + tid] = sdb;
    }
  }
}

// ------------------------------------------------------------------------ thin 3x3 weight gradient (<= 4 output channels)
// `final` (13 -> 3 channels on the full 192 x 192 crops): with 16-row MFMA tiles 13 of the 16 dY columns are padding.
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4 x 4 outer products: block b (lanes 4 b .. 4 b + 3) computes
// D_b[i][j] += A[lane 4 b + i] * B[lane 4 b + j], lane 4 b + j holding column j in its 4 registers (measured,
// tools/probes/mfma4x4_probe.hip).  Here i = output channel (A = dY[pixel][lane % 4], the same in every block) and
// block b = one (tap, 4-channel group) of X, j = the channel in the group: one instruction per pixel contracts 16
// (tap, c4) groups with no padding, and a lane's registers are dW[co = 0..3][(tap, c)] of its own (tap, c).
// Same persistent halo tiles, slabs and reduction as wgrad_tile_kernel.  The (tap, c4) groups are dealt GPP per
// pass so that, with 16 channels, the three taps of a pass fall in different LDS banks (the pixel is shared).
template <int NP, int NLD>
__global__ __launch_bounds__(256, 2) void wgrad_thin_kernel(const WgP p, int tiles_x, int tiles_y, int ntiles, int gpp, int coef_off) {
  constexpr int cpt = 16;                                  // X tile pitch: compile-time, so that every LDS read below has an immediate offset
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int YP = TT_H * TT_W + 4;                      // dY tile is channel-major: [4][YP]
  float* Xt = lds;                                         // [TT_PIX][cpt]
  float* Yt = lds + TT_PIX * cpt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = p.cin_ld >> 2;
  const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x.p, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy.p, 0, p.dy_bytes, 0x00020000);
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
  const int y_r = tid / TT_W, y_c = tid % TT_W;            // one dY pixel (4 channels) per thread
  f32x4 rtA[NLD], rtB[NLD], ryA, ryB;                    // two tiles in flight: a tile's MFMA work is shorter than a load
  unsigned okA = 0, okB = 0;
  auto load_tile = [&](int tile, f32x4 (&rt)[NLD], f32x4& ry, unsigned& okm) {
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int t0 = ty_i * TT_H, u0 = tx_i * TT_W;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int iy = t0 - p.pad_h + e_r[i], ix = u0 - p.pad + e_c[i];
      bool ok = e_r[i] >= 0;
      if (p.pad_mode != ITG_PAD_REPLICATE) ok = ok && (unsigned)iy < (unsigned)p.x.H && (unsigned)ix < (unsigned)p.x.W;
      iy = min(max(iy, 0), p.x.H - 1); ix = min(max(ix, 0), p.x.W - 1);
      unsigned o = (unsigned)grid_off(p.x, n, iy, ix) * 4u + e_cb[i];
      rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxr, ok ? o : p.x_bytes, 0, 0));
    }
    int t = t0 + y_r, u = u0 + y_c;
    bool ok = t < p.MT && u < p.MU;
    unsigned o = (unsigned)grid_off(p.dy, n, ok ? t : 0, ok ? u : 0) * 4u;
    ry = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ryr, ok ? o : p.dy_bytes, 0, 0));
  };
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
  auto store_tile = [&](const f32x4 (&rt)[NLD], const f32x4& ry, unsigned okm) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (e_r[i] < 0) continue;
      f32x4 v = rt[i];
      *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = v;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) Yt[c * YP + tid] = ry[c];
    dbacc += ry;
  };
  const int ch = lane & 3, blk = lane >> 2;
  const int nq = 9 * q4;
  int xoff[NP], mrow[NP];                                  // this lane's (tap, c) per pass: LDS offset and dW row
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int q = gpp * i + blk;
    bool live = blk < gpp && q < nq;
    q = live ? q : min(gpp * i, nq - 1);                   // idle blocks repeat the pass's first group (same address: a broadcast)
    int tap = q / q4, c4 = q - tap * q4;
    int ky = tap / 3, kx = tap - ky * 3;
    xoff[i] = (ky * (TT_W + 2) + kx) * cpt + 4 * c4 + ch;
    mrow[i] = live ? tap * p.cin_ld + 4 * c4 + ch : -1;
  }
  f32x4 acc[NP];
  const float* xp[NP];                                     // this wave's two pixel rows, at the lane's (tap, c)
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    xp[i] = Xt + 2 * wave * (TT_W + 2) * cpt + xoff[i];
  }

  auto contract = [&]() {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const float* yrow = Yt + ch * YP + (2 * wave + rr) * TT_W;
#pragma unroll
      for (int s4 = 0; s4 < TT_W / 4; ++s4) {
        const f32x4 ya = *reinterpret_cast<const f32x4*>(yrow + 4 * s4);
        float xb[4][NP];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < NP; ++i) xb[k][i] = xp[i][(rr * (TT_W + 2) + 4 * s4 + k) * cpt];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < NP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(ya[k], xb[k][i], acc[i], 0, 0, 0);
      }
    }
  };
  int tile = blockIdx.x;
  const int step = gridDim.x;
  if (tile < ntiles) load_tile(tile, rtA, ryA, okA);
  if (tile + step < ntiles) load_tile(tile + step, rtB, ryB, okB);
  for (; tile < ntiles; tile += 2 * step) {
    store_tile(rtA, ryA, okA);
    __syncthreads();
    if (tile + 2 * step < ntiles) load_tile(tile + 2 * step, rtA, ryA, okA);
    contract();
    __syncthreads();
    if (tile + step >= ntiles) break;
    store_tile(rtB, ryB, okB);
    __syncthreads();
    if (tile + 3 * step < ntiles) load_tile(tile + 3 * step, rtB, ryB, okB);
    contract();
    __syncthreads();
  }
  // ---- sum the 4 waves' accumulators in wave order through LDS: R[m = tap * cin_ld + c][4]
  float* R = lds;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (mrow[i] < 0) continue;
        f32x4* dst = reinterpret_cast<f32x4*>(R) + mrow[i];
        *dst = w == 0 ? acc[i] : *dst + acc[i];
      }
    }
    __syncthreads();
  }
  float* slab = p.slab + (size_t)blockIdx.x * p.co_rows * p.Kpad;
  const int mrows = 9 * p.cin_ld;
  for (int idx = tid; idx < p.co_rows * p.Kpad; idx += 256) {
    int co = idx / p.Kpad, m = idx - co * p.Kpad;
    slab[idx] = (m < mrows && co < 4) ? R[m * 4 + co] : 0.f;
  }
  if (p.dbslab) {       // bias gradient: per-thread sums of the staged dY pixels -> fixed-order sum per channel
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds) + 9 * 32;
    red[tid] = dbacc;
    __syncthreads();
    if (tid < 16) {
      float sdb = 0.f;
      if (tid < 4)
        for (int r = 0; r < 256; ++r) sdb += red[r][tid];
      p.dbslab[(size_t)blockIdx.x * p.co_rows + tid] = sdb;
    }
  }
}


// ------------------------------------------------------------------------ folded-upsample weight gradient (halo tiles)
// Weight gradient of conv3x3(nearest x2 (x)) (itg_conv_geom.up2) for the narrow layers behind an upsample: per parity
// class (ry, rx) the 2 x 2 tap gradients  dWc[co][(jy, jx), c] = sum over source pixels (t, u) of
// dY(2t + ry, 2u + rx)[co] * X(t + ry - 1 + jy, u + rx - 1 + jx)[c].  A persistent workgroup keeps the (8+2) x (32+2)
// halo tile of the HALF-SIZE x in LDS and walks the four classes over it: the class's 8 x 32 dY pixels (every other
// pixel of a 16 x 64 block) are staged into one of two LDS buffers while the previous class runs on the MFMA pipe; the
// accumulators of all four classes (4 x 4 NJ row tiles) stay in registers across tiles.  Slabs, bias partials and reduce
// stages as the generic parity-class path ([block][class][co][Kpad], wgrad_up2_reduce_kernel).
template <int NJ, int NLD>
__global__ __launch_bounds__(256, 2) void wgrad_up2_tile_kernel(const WgP p, int tiles_x, int tiles_y, int ntiles, int cpt) {
  constexpr int MF = 4 * NJ;
  constexpr int CPD = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xt = lds;                                         // [TT_PIX][cpt]
  float* Yt = lds + TT_PIX * cpt;                          // [2][TT_H * TT_W][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = p.cin_ld >> 2, yq4 = p.dy.ld >> 2;
  const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x.p, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy.p, 0, p.dy_bytes, 0x00020000);
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
  constexpr int YLD = 4;                                   // dY: 256 pixels x yq4 (<= 4) channel groups / 256 threads
  int y_r[YLD], y_c[YLD], y_lds[YLD];
  const unsigned y_cb = (unsigned)(tid % yq4) * 16u;
#pragma unroll
  for (int i = 0; i < YLD; ++i) {
    int e = tid + i * 256;
    bool live = e < TT_H * TT_W * yq4;
    int pix = live ? e / yq4 : 0;
    y_r[i] = live ? pix / TT_W : -1;
    y_c[i] = pix - (pix / TT_W) * TT_W;
    y_lds[i] = pix * CPD + (e % yq4) * 4;
  }
  for (int e = tid; e < 2 * TT_H * TT_W * CPD; e += 256) Yt[e] = 0.f;    // channel groups >= dy.ld stay zero
  f32x4 rt[NLD], ry[YLD];
  auto tile_origin = [&](int tile, int& n, int& t0, int& u0) {
    int b = tile;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    n = b / tiles_y;
    t0 = ty_i * TT_H; u0 = tx_i * TT_W;
  };
  auto load_x = [&](int tile) {
    int n, t0, u0;
    tile_origin(tile, n, t0, u0);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int iy = t0 - 1 + e_r[i], ix = u0 - 1 + e_c[i];
      bool ok = e_r[i] >= 0;
      if (p.pad_mode != ITG_PAD_REPLICATE) ok = ok && (unsigned)iy < (unsigned)p.x.H && (unsigned)ix < (unsigned)p.x.W;
      iy = min(max(iy, 0), p.x.H - 1); ix = min(max(ix, 0), p.x.W - 1);
      unsigned o = (unsigned)grid_off(p.x, n, iy, ix) * 4u + e_cb[i];
      rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxr, ok ? o : p.x_bytes, 0, 0));
    }
  };
  auto load_y = [&](int tile, int cls) {
    int n, t0, u0;
    tile_origin(tile, n, t0, u0);
    const int ryc = cls >> 1, rxc = cls & 1;
#pragma unroll
    for (int i = 0; i < YLD; ++i) {
      int t = t0 + y_r[i], u = u0 + y_c[i];
      bool ok = y_r[i] >= 0 && t < p.MT && u < p.MU;
      unsigned o = (unsigned)grid_off(p.dy, n, ok ? 2 * t + ryc : 0, ok ? 2 * u + rxc : 0) * 4u + y_cb;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ryr, ok ? o : p.dy_bytes, 0, 0));
    }
  };
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (e_r[i] >= 0) *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = rt[i];
  };
  auto store_y = [&](int buf) {
#pragma unroll
    for (int i = 0; i < YLD; ++i)
      if (y_r[i] >= 0) { *reinterpret_cast<f32x4*>(Yt + buf * TT_H * TT_W * CPD + y_lds[i]) = ry[i]; dbacc += ry[i]; }
  };
  // MFMA rows: (tap, 4-channel group) q = 16 j + fr of the class's 4 taps; a lane's ds_read_b128 feeds four row tiles
  const int fr = lane & 15, g = lane >> 4;
  const int nq = 4 * q4;
  int qbase[NJ];                                             // class (0, 0) offsets; class (ry, rx) adds (ry * (TT_W + 2) + rx) * cpt
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int q = min(16 * j + fr, nq - 1);
    int tap = q / q4, c4 = q - tap * q4;
    qbase[j] = ((tap >> 1) * (TT_W + 2) + (tap & 1)) * cpt + 4 * c4;
  }
  // Two passes over the tiles, two classes (ry = half; rx = 0, 1) each: the accumulators of FOUR classes plus the next
  // tile's prefetch registers do not fit the 256-register budget of two workgroups per CU (28 spills); the half-size x
  // tile is read twice instead (33 MB more for the 26-channel layer).
  int rrow[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int q = 16 * j + 4 * g + e;
      int tap = q / q4, c4 = q - tap * q4;
      rrow[j][e] = q < nq ? tap * p.cin_ld + 4 * c4 : -1;
    }
  const int mrows = 4 * p.cin_ld;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    f32x4 acc[2][MF];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < MF; ++i) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int tile = blockIdx.x;
    if (tile < ntiles) { load_x(tile); load_y(tile, 2 * half); }
    __syncthreads();                                     // (first pass: Yt zeroed; second: the slab writers are done with lds)
    for (; tile < ntiles; tile += gridDim.x) {
      store_x();
      store_y(0);
      __syncthreads();
      const int next = tile + gridDim.x;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c == 0) load_y(tile, 2 * half + 1);
        else if (next < ntiles) { load_x(next); load_y(next, 2 * half); }
        const int coff = (half * (TT_W + 2) + c) * cpt;
        const float* Yb = Yt + c * TT_H * TT_W * CPD;
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
          const float* xrow = Xt + ((2 * wave + rr) * (TT_W + 2) + g) * cpt + coff;
          const float* yrow = Yb + ((2 * wave + rr) * TT_W + g) * CPD + fr;
#pragma unroll 2
          for (int s4 = 0; s4 < TT_W / 4; ++s4) {
            const float bv = yrow[s4 * 4 * CPD];
            const float* xs = xrow + s4 * 4 * cpt;
            f32x4 av[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) av[j] = *reinterpret_cast<const f32x4*>(xs + qbase[j]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[c][4 * j + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], bv, acc[c][4 * j + e], 0, 0, 0);
          }
        }
        if (c == 0) store_y(1);
        __syncthreads();
      }
    }
    // ---- per class: sum the 4 waves' accumulators in wave order through LDS, R[m = tap * cin_ld + c][16], write the slab
    float* R = lds;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (rrow[j][e] < 0) continue;
                float* dst = R + (rrow[j][e] + t) * 16 + fr;
                *dst = (w == 0 ? 0.f : *dst) + acc[c][4 * j + t][e];
              }
        }
        __syncthreads();
      }
      float* slab = p.slab + ((size_t)blockIdx.x * 4 + 2 * half + c) * p.co_rows * p.Kpad;
      for (int idx = tid; idx < p.co_rows * p.Kpad; idx += 256) {
        int co = idx / p.Kpad, m = idx - co * p.Kpad;
        slab[idx] = m < mrows ? R[m * 16 + co] : 0.f;
      }
      __syncthreads();
    }
    if (half == 0) {                                     // the R rows overwrote the tiles' space: dY groups >= dy.ld are zero again
      for (int e = tid; e < 2 * TT_H * TT_W * CPD; e += 256) Yt[e] = 0.f;
    }
  }
  if (p.dbslab) {       // bias gradient: the dY pixels of all four classes, summed per channel in a fixed order (class 0's row)
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    red[tid] = dbacc;
    __syncthreads();
    if (tid < 16) {
      float sdb = 0.f;
      if ((tid >> 2) < yq4)
        for (int r = (tid >> 2); r < 256; r += yq4) sdb += red[r][tid & 3];
      for (int cls = 0; cls < 4; ++cls) p.dbslab[((size_t)blockIdx.x * 4 + cls) * p.co_rows + tid] = cls == 0 ? sdb : 0.f;
    }
  }
}

// plan of the folded halo-tile weight gradient (narrow layers on large images), or ok = 0
TileWgPlan plan_wgrad_up2_tile(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g) {
  TileWgPlan t;
  t.ok = 0; t.thin = 0; t.gpp = 16; t.coef_off = 0;
  const int enable = kernel_on(KM_UP2_WTILE);
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  if (!enable || !g->up2 || ph != 1 || g->precision == ITG_PREC_BF16 || x->ld > 32 || dy->ld > 16 || (dy->ld & 3)) return t;
  const int H = x->gh * x->ph, W = x->gw * x->pw;          // source domain
  if ((int64_t)H * W < 48 * 48) return t;
  const int nq = 4 * (x->ld >> 2);
  t.mf = (nq + 15) / 16;                                   // NJ
  if (t.mf > 2) return t;
  const int nld = (TT_PIX * (x->ld >> 2) + 255) / 256;
  if (nld > 11) return t;
  t.nld = nld <= 6 ? 6 : 11;
  t.cpt = (x->ld % 8 == 4) ? x->ld : x->ld + 4;
  t.tiles_x = (W + TT_W - 1) / TT_W; t.tiles_y = (H + TT_H - 1) / TT_H;
  t.ntiles = (int64_t)x->n * t.tiles_x * t.tiles_y;
  size_t fl = (size_t)TT_PIX * t.cpt + (size_t)2 * TT_H * TT_W * 16;
  const size_t red = (size_t)4 * 32 * 16 + 256 * 4;               // R buffer / bias partials reuse the tiles' space
  if (red > fl) fl = red;
  t.lds = fl * sizeof(float);
  if (t.lds > 80 * 1024 || t.ntiles > 0x7fffffff) return t;
  int per_cu = (int)((160 * 1024) / t.lds);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  const int64_t want = 256 * (int64_t)per_cu;
  t.blocks = (int)(t.ntiles < want ? t.ntiles : want);
  t.ok = 1;
  return t;
}

// TnPlan of the folded halo-tile weight gradient: one slab per (persistent workgroup, class)
TnPlan tn_plan_for_up2_tiles(const TileWgPlan& tw, int co_ld, int cin_ld) {
  TnPlan t;
  t.bcol = -1; t.bco = 16;
  t.co_rows = round_up(co_ld, 16);
  t.Kpad = round_up(4 * cin_ld, 16);
  t.splits = tw.blocks; t.chunks_per_split = 0; t.nchunks = 0;
  t.slab_floats = (int64_t)t.splits * 4 * t.co_rows * t.Kpad;
  t.ngroups = t.splits > red_group() ? (t.splits + red_group() - 1) / red_group() : 0;
  t.ws_floats = t.slab_floats + (int64_t)t.ngroups * 4 * t.co_rows * t.Kpad;
  return t;
}

int launch_wgrad_up2_tile(const WgP& p, const TileWgPlan& t, hipStream_t s) {
  const void* kern = t.mf == 1 ? (t.nld <= 6 ? (const void*)&wgrad_up2_tile_kernel<1, 6> : (const void*)&wgrad_up2_tile_kernel<1, 11>)
                               : (t.nld <= 6 ? (const void*)&wgrad_up2_tile_kernel<2, 6> : (const void*)&wgrad_up2_tile_kernel<2, 11>);
  static const void* attr_set[4] = {nullptr};
  {
    bool seen = false;
    int slot = 0;
    for (; slot < 4 && attr_set[slot]; ++slot) seen = seen || attr_set[slot] == kern;
    if (!seen && slot < 4) {
      if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) { (void)hipGetLastError(); return ITG_ERR_LAUNCH; }
      attr_set[slot] = kern;
    }
  }
  snprintf(g_last_launch, sizeof(g_last_launch), "wgrad_up2_tile_kernel<%d, %d>", t.mf, t.nld <= 6 ? 6 : 11);
  WgP q = p;
  int a_tx = t.tiles_x, a_ty = t.tiles_y, a_nt = (int)t.ntiles, a_cpt = t.cpt;
  void* args[] = {(void*)&q, &a_tx, &a_ty, &a_nt, &a_cpt};
  (void)hipLaunchKernel(kern, dim3((unsigned)t.blocks), dim3(256), args, t.lds, s);
  return hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
}


TileWgPlan plan_wgrad_tile(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g) {
  TileWgPlan t;
  t.ok = 0;
  const int enable = kernel_on(KM_WGRAD_TILE);
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  if (!enable || g->up2 || g->kh != 3 || g->kw != 3 || g->stride != 1 || g->pad != 1 || ph != 1) return t;
  if (g->precision == ITG_PREC_BF16 || x->ld > 32 || dy->ld > 16 || (dy->ld != 4 && dy->ld != 8 && dy->ld != 16)) return t;
  const int H = dy->gh * dy->ph, W = dy->gw * dy->pw;
  if ((int64_t)H * W < 64 * 64) return t;
  const int nq = 9 * (x->ld >> 2);                       // (tap, 4-channel group) rows; 16 per group of 4 MFMA tiles
  t.mf = (nq + 15) / 16;                                  // NJ
  if (t.mf > 5) return t;
  t.nld = (TT_PIX * (x->ld >> 2) + 255) / 256;
  t.nld = t.nld <= 6 ? 6 : 11;
  if ((TT_PIX * (x->ld >> 2) + 255) / 256 > 11) return t;
  t.cpt = (x->ld % 8 == 4) ? x->ld : x->ld + 4;
  const int thin_en = kernel_on(KM_WGRAD_THIN);
  t.thin = thin_en && dy->ld == 4 && x->ld <= 16;         // <= 36 (tap, c4) groups: three passes of 12 or 16
  t.gpp = 16;
  if (t.thin) { t.cpt = 16; t.gpp = x->ld == 16 ? 12 : 16; }   // 16 channels: three taps per pass, bank-conflict free at pitch 16
  t.tiles_x = (W + TT_W - 1) / TT_W; t.tiles_y = (H + TT_H - 1) / TT_H;
  t.ntiles = (int64_t)dy->n * t.tiles_x * t.tiles_y;
  size_t fl = (size_t)TT_PIX * t.cpt + (size_t)TT_H * TT_W * 16;
  size_t red = (size_t)9 * 32 * 16 + 4 + 256 * 4;               // reduction buffer + bias partials reuse the tiles' space
  if (red > fl) fl = red;
  fl = (fl + 3) & ~(size_t)3;
  t.coef_off = (int)fl;                                         // alpha | beta' of the input transform behind everything else
  fl += 64;
  t.lds = fl * sizeof(float);
  if (t.lds > 64 * 1024 || t.ntiles > 0x7fffffff) return t;
  // one persistent workgroup per CU: alone the kernel is 13 % faster with two, but it runs beside the input-gradient chain
  // of the same backward pass and two would crowd that out of LDS (step: 780 vs 774 crops/s)
  int per_cu = (int)((160 * 1024) / t.lds);
  if (per_cu > 1) per_cu = 1;
  int64_t want = 256 * (int64_t)(per_cu < 1 ? 1 : per_cu);
  t.blocks = (int)(t.ntiles < want ? t.ntiles : want);
  t.ok = 1;
  return t;
}

template <int NJ, int NLD>
void launch_wgrad_tile(const WgP& p, const TileWgPlan& t, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tile_kernel<NJ, NLD>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr_done = true;
  }
  snprintf(g_last_launch, sizeof(g_last_launch), "wgrad_tile_kernel<%d, %d>", NJ, NLD);
  hipLaunchKernelGGL((wgrad_tile_kernel<NJ, NLD>), dim3((unsigned)t.blocks), dim3(256), t.lds, s, p, t.tiles_x, t.tiles_y,
                       (int)t.ntiles, t.cpt, t.coef_off);
}

void launch_wgrad_thin(const WgP& p, const TileWgPlan& t, hipStream_t s) {
  snprintf(g_last_launch, sizeof(g_last_launch), "wgrad_thin_kernel<3, 6>");
  hipLaunchKernelGGL((wgrad_thin_kernel<3, 6>), dim3((unsigned)t.blocks), dim3(256), t.lds, s, p, t.tiles_x, t.tiles_y,
                       (int)t.ntiles, t.gpp, t.coef_off);
}

// dW[co][ci][ky][kx] (+)= sum_z slab[z][co][(ky*kw+kx)*ci_ld + ci]
// One workgroup per (o, 64-channel chunk): slab reads are coalesced along ci, the (ci, tap) tile is
// transposed through LDS so that the OIHW store is one contiguous run of 64*taps floats.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           float* __restrict__ db, const float* __restrict__ dbslab,
                                                           int dbsplits, int splits, int co, int ci, int ci_ld, int kh,
                                                           int kw, int co_rows, int Kpad, int accumulate) {
  __shared__ float tile[64 * 50];
  const int taps = kh * kw;
  const int pitch = taps | 1;           // odd pitch: the 16-tap layers were 16-way bank conflicted at pitch 16 (73 % of the LDS cycles)
  const int nchunk = (ci + 63) / 64;
  const int o = blockIdx.x / nchunk;
  const int c0 = (blockIdx.x - o * nchunk) * 64;
  const int cn = min(64, ci - c0);
  const size_t zstride = (size_t)co_rows * Kpad;
  const float* src = slab + (size_t)o * Kpad + c0;
  if (db && c0 == 0) {      // bias gradient: sum of the per-split partials (fixed order -> deterministic)
    __shared__ float part[256];
    float sdb = 0.f;
    for (int z = threadIdx.x; z < dbsplits; z += 256) sdb += dbslab[(size_t)z * co_rows + o];
    part[threadIdx.x] = sdb;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {               // fixed-shape tree: deterministic
      if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) db[o] = (accumulate & ITG_ACC_DB) ? db[o] + part[0] : part[0];
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < 64 * taps; idx += 256) {
    int c = idx & 63, t = idx >> 6;
    float s = 0.f;
    if (c < cn) {
      const float* q = src + (size_t)t * ci_ld + c;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four loads in flight, summed in a fixed order
      int z = 0;
      for (; z + 4 <= splits; z += 4) {
        s0 += q[(size_t)z * zstride];
        s1 += q[(size_t)(z + 1) * zstride];
        s2 += q[(size_t)(z + 2) * zstride];
        s3 += q[(size_t)(z + 3) * zstride];
      }
      for (; z < splits; ++z) s0 += q[(size_t)z * zstride];
      s = (s0 + s1) + (s2 + s3);
    }
    tile[c * pitch + t] = s;
  }
  __syncthreads();
  float* dst = dw + ((size_t)o * ci + c0) * taps;
  for (int idx = threadIdx.x; idx < cn * taps; idx += 256) {
    const int c = idx / taps, t = idx - c * taps;
    const float v = tile[c * pitch + t];
    dst[idx] = (accumulate & ITG_ACC_DW) ? dst[idx] + v : v;
  }
}

// itg_conv_geom.up2: slab z = [class (ry, rx)][co][(jy*2+jx)*ci_ld + ci] holds the class's 2 x 2 tap gradients; tap (i, j) of
// the 3 x 3 filter belongs to tap (jy, jx) = ((i + 1 - ry) >> 1, (j + 1 - rx) >> 1) of every class (the adjoint of the
// phase sums of itg_pack_up2_fwd):  dW[o][c][i][j] (+)= sum_z sum_cls slab[z][cls][o][(jy*2+jx)*ci_ld + c]
__global__ __launch_bounds__(256) void wgrad_up2_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                               float* __restrict__ db, const float* __restrict__ dbslab,
                                                               int dbsplits, int splits, int co, int ci, int ci_ld, int co_rows,
                                                               int Kpad, int accumulate) {
  __shared__ float t16[64 * 17];
  __shared__ float part[256];
  const int nchunk = (ci + 63) / 64;
  const int o = blockIdx.x / nchunk;
  const int c0 = (blockIdx.x - o * nchunk) * 64;
  const int cn = min(64, ci - c0);
  const size_t cstride = (size_t)co_rows * Kpad, zstride = 4 * cstride;
  if (db && c0 == 0) {
    float sdb = 0.f;
    for (int z = threadIdx.x; z < dbsplits; z += 256) sdb += dbslab[(size_t)z * co_rows + o];
    part[threadIdx.x] = sdb;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) db[o] = (accumulate & ITG_ACC_DB) ? db[o] + part[0] : part[0];
  }
  for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
    const int c = idx & 63, q = idx >> 6;                  // q = class * 4 + tap
    float s = 0.f;
    if (c < cn) {
      const float* src = slab + (size_t)(q >> 2) * cstride + (size_t)o * Kpad + (size_t)(q & 3) * ci_ld + c0 + c;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int z = 0;
      for (; z + 4 <= splits; z += 4) {
        s0 += src[(size_t)z * zstride];
        s1 += src[(size_t)(z + 1) * zstride];
        s2 += src[(size_t)(z + 2) * zstride];
        s3 += src[(size_t)(z + 3) * zstride];
      }
      for (; z < splits; ++z) s0 += src[(size_t)z * zstride];
      s = (s0 + s1) + (s2 + s3);
    }
    t16[c * 17 + q] = s;
  }
  __syncthreads();
  float* dst = dw + ((size_t)o * ci + c0) * 9;
  for (int idx = threadIdx.x; idx < cn * 9; idx += 256) {
    const int c = idx / 9, t = idx - c * 9, i = t / 3, j = t - i * 3;
    float v = 0.f;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int jy = (i + 1 - (cls >> 1)) >> 1, jx = (j + 1 - (cls & 1)) >> 1;
      v += t16[c * 17 + cls * 4 + jy * 2 + jx];
    }
    dst[idx] = (accumulate & ITG_ACC_DW) ? dst[idx] + v : v;
  }
}

// out[zo][e] = sum over the zo-th group of `group` slabs
__global__ void slab_group_reduce_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int64_t e4, int splits,
                                         int group, int ngroups) {
  int64_t total = e4 * ngroups;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t e = i % e4;
    int zo = (int)(i / e4);
    int z1 = min(splits, (zo + 1) * group);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = zo * group; z < z1; ++z) v += in[(size_t)z * e4 + e];
    out[i] = v;
  }
}

template <int BCOL, int BCO, int WCOL, int WCO>
int launch_tn(WgP p, int splits, int prec, hipStream_t s) {
  p.ncol_tiles = (p.Kpad + BCOL - 1) / BCOL;
  p.nco_tiles = (p.co_rows + BCO - 1) / BCO;
  dim3 grid((unsigned)(p.ncol_tiles * p.nco_tiles * (p.up2 ? 4 : 1)), (unsigned)(p.ucls ? p.ucls : 1), (unsigned)splits);
  // offset-table pitch: the taps one column tile can touch (+ the dY slot)
  int taps_tile = (BCOL + p.cin_ld - 1) / p.cin_ld + 1;
  if (taps_tile > p.ntaps) taps_tile = p.ntaps;
  const int otp = taps_tile + 1;
  // long per-workgroup pixel loops run best with the occupancy of the single-prefetch variant (4 waves per SIMD), short
  // ones with two stages in flight (measured on D's 256->512 layer vs its 64->128 / 128->256 layers)
  const int depth = p.chunks_per_split >= 128 ? 1 : 2;
  const int kp = prec == ITG_PREC_BF16 ? 32 : BKP;
  const int flat_env = kernel_on(KM_TN_FLAT);
  const bool flat = flat_env && prec != ITG_PREC_BF16 && !p.up2 && p.x.gh == 1 && p.x.gw == 1 && p.dy.gh == 1 &&
                    p.dy.gw == 1 && p.pad_mode != ITG_PAD_REPLICATE && otp <= 16 && p.MU >= kp;
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_tn_kernel<%d, %d, %d, %d, %s, %d, %s>", BCOL, BCO, WCOL, WCO,
           prec == ITG_PREC_BF16 ? "true" : "false", prec == ITG_PREC_BF16 ? 1 : depth, flat ? "true" : "false");
  if (flat) {                                                      // (+ the spare word inactive producer threads write to)
    if (depth == 2)
      hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, false, 2, true>), grid, dim3(256), (size_t)3 * kp * otp * 4 + 16, s, p, otp);
    else
      hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, false, 1, true>), grid, dim3(256), (size_t)2 * kp * otp * 4 + 16, s, p, otp);
  } else if (prec == ITG_PREC_BF16)
    hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, true, 1>), grid, dim3(256), (size_t)2 * kp * otp * 4, s, p, otp);
  else if (depth == 2)
    hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, false, 2>), grid, dim3(256), (size_t)3 * kp * otp * 4, s, p, otp);
  else
    hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, false, 1>), grid, dim3(256), (size_t)2 * kp * otp * 4, s, p, otp);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}


TnPlan plan_tn(int64_t M, int co_ld, int Ktot, int prec, int ncls) {
  const int kp = prec == ITG_PREC_BF16 ? 32 : BKP;
  TnPlan t;
  t.co_rows = round_up(co_ld, 16);
  t.Kpad = round_up(Ktot, 16);
  if (t.Kpad <= 64 && t.co_rows > 16 && t.co_rows <= 64) { t.bco = 64; t.bcol = 64; }   // 3-channel input layer: K = taps * 4
  else if (t.co_rows <= 16) { t.bco = 16; t.bcol = 256; }
  else if (t.co_rows <= 32) { t.bco = 32; t.bcol = 256; }
  else if (t.co_rows <= 64) { t.bco = 64; t.bcol = 256; }
  else { t.bco = 128; t.bcol = 128; }
  int tiles = ((t.Kpad + t.bcol - 1) / t.bcol) * ((t.co_rows + t.bco - 1) / t.bco);
  t.nchunks = (int)((M + kp - 1) / kp);
  // workgroup target: ~3 per CU; with bf16 operands a split's MFMA work is a quarter as long and the slab round trip
  // weighs more: 2 per CU (config 3: 1996 -> 2026 crops/s; config 1 loses 1 % with it)
  const int want_blocks = prec == ITG_PREC_BF16 ? 512 : 768;        // (384 / 512 / 640 / 1024 / 1152 / 1536 with fp32 operands: -3.8 / +0.2 / 0.0 / -1.1 / -1.0 / -1.1 %)
  int want = (want_blocks / ncls + tiles - 1) / tiles;      // ncls grids of (tiles x splits) workgroups run as one launch
  int max_splits = (t.nchunks + 7) / 8;             // at least 8 chunks per split
  int splits = want < max_splits ? want : max_splits;
  if (splits < 1) splits = 1;
  t.chunks_per_split = (t.nchunks + splits - 1) / splits;
  t.splits = (t.nchunks + t.chunks_per_split - 1) / t.chunks_per_split;
  t.slab_floats = (int64_t)t.splits * ncls * t.co_rows * t.Kpad;
  t.ngroups = t.splits > red_group() ? (t.splits + red_group() - 1) / red_group() : 0;
  t.ws_floats = t.slab_floats + (int64_t)t.ngroups * ncls * t.co_rows * t.Kpad;
  return t;
}


// TnPlan of the halo-tile weight gradient: one slab per persistent workgroup, same reduction stages
TnPlan tn_plan_for_tiles(const TileWgPlan& tw, int co_ld, int Ktot) {
  TnPlan t;
  t.bcol = -1; t.bco = 16;
  t.co_rows = round_up(co_ld, 16);
  t.Kpad = round_up(Ktot, 16);
  t.splits = tw.blocks; t.chunks_per_split = 0; t.nchunks = 0;
  t.slab_floats = (int64_t)t.splits * t.co_rows * t.Kpad;
  t.ngroups = t.splits > red_group() ? (t.splits + red_group() - 1) / red_group() : 0;
  t.ws_floats = t.slab_floats + (int64_t)t.ngroups * t.co_rows * t.Kpad;
  return t;
}


// the contraction of one layer into its slabs (no reduce stage)
int run_wgrad_slabs(WgP& p, const TnPlan& t, const TileWgPlan& tw, int prec, hipStream_t s) {
  int rc;
  if (tw.ok && p.up2) {
    rc = launch_wgrad_up2_tile(p, tw, s);
  } else if (tw.ok) {
    rc = ITG_OK;
    const bool small = tw.nld <= 6;
    if (tw.thin) launch_wgrad_thin(p, tw, s);
    else if (tw.mf == 1) launch_wgrad_tile<1, 6>(p, tw, s);
    else if (tw.mf == 2) launch_wgrad_tile<2, 6>(p, tw, s);
    else if (tw.mf == 3 && small) launch_wgrad_tile<3, 6>(p, tw, s);
    else if (tw.mf == 3) launch_wgrad_tile<3, 11>(p, tw, s);
    else if (tw.mf == 4) launch_wgrad_tile<4, 11>(p, tw, s);
    else launch_wgrad_tile<5, 11>(p, tw, s);
    ITG_CHECK_LAUNCH();
  } else if (t.bcol == 64) rc = launch_tn<64, 64, 32, 32>(p, t.splits, prec, s);
  else if (t.bco == 16) rc = launch_tn<256, 16, 64, 16>(p, t.splits, prec, s);
  else if (t.bco == 32) rc = launch_tn<256, 32, 64, 32>(p, t.splits, prec, s);
  else if (t.bco == 64) rc = launch_tn<256, 64, 64, 64>(p, t.splits, prec, s);
  else rc = launch_tn<128, 128, 64, 64>(p, t.splits, prec, s);
  return rc;
}

int run_wgrad(WgP& p, const TnPlan& t, const TileWgPlan& tw, int prec, const itg_tensor* x, const itg_tensor* dy,
              const itg_conv_geom* g, float* dw, float* db, int accumulate, float* workspace, hipStream_t s) {
  int rc = run_wgrad_slabs(p, t, tw, prec, s);
  if (rc) return rc;
  const float* red_src = workspace;
  int red_n = t.splits;
  const int ncls = p.up2 ? 4 : 1;
  if (t.ngroups > 0) {
    int64_t e4 = (int64_t)ncls * t.co_rows * t.Kpad / 4;
    float* stage = workspace + t.slab_floats;
    int64_t tot4 = e4 * t.ngroups;
    int b2 = (int)((tot4 + 255) / 256 < 8192 ? (tot4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(slab_group_reduce_kernel, dim3(b2), dim3(256), 0, s, (const f32x4*)workspace, (f32x4*)stage, e4,
                       t.splits, red_group(), t.ngroups);
    ITG_CHECK_LAUNCH();
    red_src = stage; red_n = t.ngroups;
  }
  if (g->kh * g->kw > 49) return ITG_ERR_ARG;
  int blocks = dy->c * ((x->c + 63) / 64);
  if (p.up2) {
    hipLaunchKernelGGL(wgrad_up2_reduce_kernel, dim3(blocks), dim3(256), 0, s, red_src, dw, db, (const float*)p.dbslab,
                       t.splits * 4, red_n, dy->c, x->c, x->ld, t.co_rows, t.Kpad, accumulate);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, red_src, dw, db, (const float*)p.dbslab,
                     t.splits, red_n, dy->c, x->c, x->ld, g->kh, g->kw, t.co_rows, t.Kpad, accumulate);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int launch_wgrad_reduce(const float* slab, int splits, const float* dbslab, int dbsplits, float* dw, float* db, int co, int ci,
                        int ci_ld, int kh, int kw, int co_rows, int Kpad, int accumulate, hipStream_t s) {
  if (kh * kw > 49) return ITG_ERR_ARG;
  const int blocks = co * ((ci + 63) / 64);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, slab, dw, db, dbslab, dbsplits, splits, co, ci, ci_ld, kh,
                     kw, co_rows, Kpad, accumulate);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// ------------------------------------------------------------------------------- every layer's reduce stage in one launch
// One workgroup per (layer, output channel o, 64-input-channel chunk) as in wgrad_reduce_kernel, but the slab range is
// also spread over the threads (zp slab phases per element, summed phase by phase in a fixed order: deterministic), so a
// layer with hundreds of slabs needs no group stage, and the (ci, tap) tile is padded to an odd pitch (the 16-tap layers
// were 16-way bank conflicted at pitch 16: 73 % of the LDS cycles).  Spectrally normalised layers also accumulate <G, W>.
struct RedJobs {
  itg_wgrad_job j[ITG_WGRAD_MAX_JOBS];
  int start[ITG_WGRAD_MAX_JOBS + 1];       // first workgroup of each job
  int n;
};

// first stage for the layers with many slabs: stage[zo][e] = sum of the zo-th group of slabs, every such layer in one launch
struct GroupJobs {
  const f32x4* in[ITG_WGRAD_MAX_JOBS];
  f32x4* out[ITG_WGRAD_MAX_JOBS];
  long long start[ITG_WGRAD_MAX_JOBS + 1];     // first flat float4 item of each job; items of a job = e4 * ngroups
  long long e4[ITG_WGRAD_MAX_JOBS];
  int splits[ITG_WGRAD_MAX_JOBS], group[ITG_WGRAD_MAX_JOBS];
  int n;
};

__global__ void slab_group_reduce_multi_kernel(const GroupJobs gj) {
  const long long total = gj.start[gj.n];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int ji = 0;
    while (ji + 1 < gj.n && i >= gj.start[ji + 1]) ++ji;
    const long long w = i - gj.start[ji];
    const long long e = w % gj.e4[ji];
    const int zo = (int)(w / gj.e4[ji]);
    const int z1 = min(gj.splits[ji], (zo + 1) * gj.group[ji]);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = zo * gj.group[ji]; z < z1; ++z) v += gj.in[ji][(size_t)z * gj.e4[ji] + e];
    gj.out[ji][w] = v;
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const RedJobs jb) {
  __shared__ float part[8192];
  __shared__ float tile[64 * 50];
  __shared__ double dpart[4];
  int ji = 0;
  while (ji + 1 < jb.n && (int)blockIdx.x >= jb.start[ji + 1]) ++ji;        // n <= 24: a short scalar scan
  const itg_wgrad_job& J = jb.j[ji];
  const int b = blockIdx.x - jb.start[ji];
  const int taps = J.kh * J.kw;
  const int nchunk = (J.ci + 63) / 64;
  const int o = b / nchunk;
  const int c0 = (b - o * nchunk) * 64;
  const int cn = min(64, J.ci - c0);
  const size_t zstride = (size_t)J.co_rows * J.Kpad;
  const float* src = J.slab + (size_t)o * J.Kpad + c0;
  const int tid = threadIdx.x;
  if (J.db && c0 == 0) {      // bias gradient: fixed-order sum of the per-split partials
    float sdb = 0.f;
    for (int z = tid; z < J.dbsplits; z += 256) sdb += J.dbslab[(size_t)z * J.co_rows + o];
    part[tid] = sdb;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) part[tid] += part[tid + w];
      __syncthreads();
    }
    if (tid == 0) J.db[o] = (J.accumulate & ITG_ACC_DB) ? J.db[o] + part[0] : part[0];
    __syncthreads();
  }
  const int nelem = 64 * taps;
  int zp = J.splits / 8;
  zp = max(1, min(zp, 8192 / nelem));
  for (int w = tid; w < nelem * zp; w += 256) {
    const int e = w % nelem, zi = w / nelem;
    const int c = e & 63, t = e >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cn) {
      const float* q = src + (size_t)t * J.ci_ld + c;
      int z = zi;
      for (; z + 3 * zp < J.splits; z += 4 * zp) {       // four loads in flight, fixed order
        s0 += q[(size_t)z * zstride];
        s1 += q[(size_t)(z + zp) * zstride];
        s2 += q[(size_t)(z + 2 * zp) * zstride];
        s3 += q[(size_t)(z + 3 * zp) * zstride];
      }
      for (; z < J.splits; z += zp) s0 += q[(size_t)z * zstride];
    }
    part[w] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  const int pitch = taps | 1;
  for (int e = tid; e < nelem; e += 256) {
    float sum = 0.f;
    for (int zi = 0; zi < zp; ++zi) sum += part[zi * nelem + e];
    tile[(e & 63) * pitch + (e >> 6)] = sum;
  }
  __syncthreads();
  float* dst = J.dw + ((size_t)o * J.ci + c0) * taps;
  double dsum = 0.0;
  for (int idx = tid; idx < cn * taps; idx += 256) {
    const int c = idx / taps, t = idx - c * taps;
    const float v = tile[c * pitch + t];
    if (J.w_orig) dsum += (double)v * (double)J.w_orig[((size_t)o * J.ci + c0) * taps + idx];
    dst[idx] = (J.accumulate & ITG_ACC_DW) ? dst[idx] + v : v;
  }
  if (J.w_orig) {             // workgroup-uniform
    dsum = wave_sum_d(dsum);
    if ((tid & 63) == 0) dpart[tid >> 6] = dsum;
    __syncthreads();
    if (tid == 0) atomicAdd(J.dot, (dpart[0] + dpart[1]) + (dpart[2] + dpart[3]));
  }
}

int launch_reduce_multi(const itg_wgrad_job* jobs, int n, hipStream_t s) {
  if (!jobs || n <= 0 || n > ITG_WGRAD_MAX_JOBS) return ITG_ERR_ARG;
  RedJobs jb;
  jb.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const itg_wgrad_job& J = jobs[i];
    if (!J.slab || !J.dw || J.splits <= 0 || J.co <= 0 || J.ci <= 0 || J.kh * J.kw > 49 || (J.w_orig && !J.dot)) return ITG_ERR_ARG;
    if (J.db && !J.dbslab) return ITG_ERR_ARG;
    jb.j[i] = J;
    jb.start[i] = total;
    total += J.co * ((J.ci + 63) / 64);
  }
  jb.start[n] = total;
  // layers with many slabs: group stage first (one launch for all of them); their final stage then sums the groups
  GroupJobs gj;
  gj.n = 0;
  long long items = 0;
  for (int i = 0; i < n; ++i) {
    const itg_wgrad_job& J = jobs[i];
    if (J.ngroups <= 0) continue;
    if (!J.stage || J.group < 2) return ITG_ERR_ARG;
    const int k = gj.n++;
    gj.in[k] = reinterpret_cast<const f32x4*>(J.slab);
    gj.out[k] = reinterpret_cast<f32x4*>(J.stage);
    gj.e4[k] = (long long)J.co_rows * J.Kpad / 4;
    gj.splits[k] = J.splits; gj.group[k] = J.group;
    gj.start[k] = items;
    items += gj.e4[k] * J.ngroups;
    jb.j[i].slab = J.stage;
    jb.j[i].splits = J.ngroups;
  }
  if (gj.n > 0) {
    gj.start[gj.n] = items;
    const int gb = (int)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    hipLaunchKernelGGL(slab_group_reduce_multi_kernel, dim3(gb), dim3(256), 0, s, gj);
    ITG_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)total), dim3(256), 0, s, jb);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

}  // namespace itgk
