// conv_nt_kernel: the implicit-GEMM forward / input-gradient kernel template and its launcher, shared by conv_nt.hip
// (plain instantiations) and conv_nt_w64.hip (blocked-accumulation instantiations for the Winograd GEMMs).
#pragma once
#include "conv_common.h"

namespace itgk {

enum { NT_PLAIN = 0, NT_W32 = 2, NT_W64 = 3 };

// NT_W64 (fp32 operands): blocked accumulation for the Winograd GEMMs.  v_mfma_f32_16x16x4_f32 adds its products to the
// accumulator one after the other, so a K loop of length L is ONE chain of L fp32 roundings: error ~ eps * sqrt(L / 2) of the
// result.  A direct convolution does not care (3e-7 at L = 4096), but the Winograd output transform A^T M A amplifies the
// error of M ~16 x (cancellation between the 49 classes): 4.6e-6 per layer, which flips the LeakyReLU behind D's 256 -> 512
// layer on 15 x as many activations (tools/wino_error_study.py reproduces the number on the CPU: the transforms' own
// rounding is 8 x smaller).  Here every K stage (16 values = 4 MFMAs) starts from a zero accumulator and the stage sums are
// added in fp64 on the vector ALU (v_cvt_f64_f32 + v_add_f64 per element and stage, in the shadow of the next stage's MFMAs):
// the chain is 16 long whatever L is.
// NT_W32: the same blocks, summed in a SECOND fp32 accumulator (v_pk_add_f32, one register per element instead of two): the
// chain is 4 MFMA steps + L / 16 additions instead of L / 4 steps - error ~ sqrt((4 + L / 16) / (L / 4)) of the plain chain's
// (0.56 at L = 256, 0.53 at L = 512).  For the F(4 x 4, 2 x 2) GEMMs, whose output transform amplifies less.
#ifndef ITG_W64_BLOCK
#define ITG_W64_BLOCK 16
#endif


// DEPTH = number of K stages whose global loads are in flight while one stage is computed.
// TBK = K elements per stage: 16 -> fp32 operands on v_mfma_f32_16x16x4_f32; 32 -> operands rounded to
// bf16 when they are staged into LDS (tensors stay fp32 in HBM) and contracted by ONE
// v_mfma_f32_16x16x32_bf16 per fragment pair and stage, fp32 accumulation (BASELINE config 3's path).
// Either way a tile row occupies 16 dwords of a 20-dword LDS row and lane group g reads dwords 4g..4g+3.
// Workgroups per CU the register budget is pinned to: 3 (168 VGPRs) for the wide tiles, 5 (96) for the
// medium fp32 tiles, 4 (128) for the medium bf16 tiles (their stage holds twice the prefetch registers).
constexpr int nt_min_blocks(int bco, int bpix, int wco, int wpix, int tbk, int mode = 0) {
  if (mode == 3) return (wco / 16) * (wpix / 16) <= 4 ? 3 : 2;       // NT_W64: fp64 accumulators (2 registers per element)
  if (mode == 2) return (wco / 16) * (wpix / 16) <= 4 ? 4 : ((wco / 16) * (wpix / 16) <= 8 ? 3 : 2);       // NT_W32: a second fp32 set
  return ((wco / 16) * (wpix / 16) <= 8 && (bco + bpix) <= 192) ? 4 : 3;
}

// MODE (NT_PLAIN | NT_W64).  (Round 3's NT_XF / NT_BNS - BatchNorm-apply in the loader, BatchNorm backward sums in the
// input-gradient epilogue - were measured slower twice and removed in round 5.)
template <int BCO, int BPIX, int WCO, int WPIX, int TBK, int DEPTH, bool TAB, int MODE>
__global__ __launch_bounds__(256, nt_min_blocks(BCO, BPIX, WCO, WPIX, TBK, MODE)) void conv_nt_kernel(const ConvP p) {
  // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2): hand every XCD one contiguous run of
  // tile ids instead, so that an L2 serves neighbouring pixel tiles (shared halo rows, all output-channel tiles of a
  // pixel tile) and not a 1-in-8 sample of the whole image.  Bijective for any grid size; speed only.
  int bx = blockIdx.x;
  if (p.xcd_remap) {
    const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = bx & 7;
    bx = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bx >> 3);
  }
  // parity classes of a multi-class launch (stride-2 input gradient, folded upsample) are the FASTEST index of the tile id:
  // the four classes of a pixel tile gather the same source pixels, so they run on one XCD at the same time and share them
  // through its L2 (as blockIdx.y they ran a quarter of the launch apart: D2's input gradient fetched 447 MB for 19 MB of dy)
  const int ncls = p.ncls > 1 ? p.ncls : 1;
  int cls = bx % ncls;
  int ucl = 0;                     // uniform classes (ConvP.ucls): the class is the slowest index, geometry of slot 0
  if (p.ucls) {
    const int per = (int)gridDim.x / p.ucls;
    ucl = bx / per;
    bx -= ucl * per;
    cls = 0;
  } else {
    bx /= ncls;
  }
  // per-class geometry (class 0 for ordinary launches); all wave-uniform scalars
  const int cMT = p.cMT[cls], cMU = p.cMU[cls], cM = p.cM[cls];
  const int cioy = p.cioy[cls], ciox = p.ciox[cls], cooy = p.cooy[cls], coox = p.coox[cls];
  const float* const cw = p.w + (p.ucls ? (size_t)ucl * p.u_w : (size_t)p.cwoff[cls]);
  float* const cpartial = p.partial + p.cpoff[cls];
  const float* const in_base = p.in.p + (size_t)ucl * p.u_in;
  float* const out_base = p.out.p + (size_t)ucl * p.u_out;
  if ((int)(bx / p.nco_tiles) * BPIX >= cM) return;
  constexpr int FI = WCO / 16, FJ = WPIX / 16;
  constexpr int WAVES_CO = BCO / WCO;
  static_assert(WAVES_CO * (BPIX / WPIX) == 4, "4 waves per workgroup");
  constexpr bool BF = TBK == 32;
  static_assert(TBK == 16 || TBK == 32, "fp32 stages hold 16 K elements, bf16 stages 32");
  // LDS rows hold 16 dwords with NO padding; the four 16-byte K groups of a row are XOR-swizzled with bit 3 of the
  // row index (group g of row r sits at slot g ^ 2*((r >> 3) & 1)), which makes every 16-lane group of a
  // ds_read_b128 fragment read (rows r..r+15 of one K group pair, MI355X_MICROARCH.md LDS table) hit 64 distinct
  // banks.  The padded pitch-20 layout this replaces was 2-way conflicted on every read (SQ_LDS_BANK_CONFLICT =
  // 50 % of the LDS cycles) and 25 % larger.
  constexpr int LDT = 16;
  constexpr int KG = TBK / 4;                // float4 groups per tile row
  constexpr int RPP = 256 / KG;              // tile rows covered per load pass
  constexpr int PL = (BPIX + RPP - 1) / RPP;         // the last pass may cover rows past the tile (96-pixel tiles)
  constexpr int WL = (BCO + RPP - 1) / RPP;
  __shared__ __attribute__((aligned(16))) float smem[2 * (BCO + BPIX) * LDT];
  float* Ws = smem;
  float* Ps = smem + 2 * BCO * LDT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int co_tile = bx % p.nco_tiles;
  const int pix_tile = bx / p.nco_tiles;
  const int co0 = co_tile * BCO;
  const int m0 = pix_tile * BPIX;
  const int wco0 = (wave % WAVES_CO) * WCO;
  const int wpix0 = (wave / WAVES_CO) * WPIX;
  const int kg = tid % KG;
  const int lrow = tid / KG;
  // dword offset of this thread's K group inside its (swizzled) LDS row; RPP is a multiple of 16, so bit 3 of the row
  // index is the same in every load pass.  bf16 stages: a thread holds half of a 16-byte group (kg & 1).
  const int swz = BF ? (((kg >> 1) ^ (((lrow >> 3) & 1) << 1)) * 4 + (kg & 1) * 2) : ((kg ^ (((lrow >> 3) & 1) << 1)) * 4);

  // ---- loader state.  Both operands are fetched with raw buffer loads: a lane's byte offset is
  // (pixel offset + channel offset); rows that read padding / lie past M carry an offset equal to the
  // buffer size, so the hardware range check returns zeros - no branches, no selects in the K loop.
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in_base, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)cw, 0, p.w_bytes, 0x00020000);
  int pn[PL], py[PL], px[PL];
  bool pv[PL];
  unsigned poff[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    int m = m0 + lrow + i * RPP;
    pv[i] = m < cM && lrow + i * RPP < BPIX;
    int n, t, u;
    decode_m(pv[i] ? m : 0, cMT, cMU, n, t, u);
    pn[i] = n;
    py[i] = t * p.isy + cioy;
    px[i] = u * p.isx + ciox;
  }
  unsigned woff[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    int row = lrow + i * RPP;
    woff[i] = (row < BCO && co0 + row < p.co_rows) ? (unsigned)(((size_t)(co0 + row) * p.Kpad + kg * 4) * 4) : p.w_bytes;
  }
  const int nk_total = (p.Kpad + TBK - 1) / TBK;
  const int kk0 = blockIdx.z * p.kchunks;
  const int kk1 = min(nk_total, kk0 + p.kchunks);
  int tap = (kk0 * TBK + kg * 4) / p.cin_ld;
  int cc = kk0 * TBK + kg * 4 - tap * p.cin_ld;
  // Per-row byte offsets of EVERY filter tap, computed once (the rows of a workgroup never change) and
  // kept in LDS: a tap change in the K loop is then one ds_read per row instead of ~35 VALU of clamp /
  // patch-grid address arithmetic.  Slot [ntaps] holds the out-of-range marker for the K padding.
  extern __shared__ unsigned taptab[];
  const int TS = p.ntaps + 1;
  constexpr bool use_tab = TAB;               // narrow layers only: wide ones change tap rarely and need the LDS
  auto tap_offset = [&](int i, int tt) -> unsigned {
    const int tky = tt / p.kw, tkx = tt - tky * p.kw;
    int iy = py[i] + tky, ix = px[i] + tkx;
    bool ok = pv[i] && tt < p.ntaps;
    if (p.pad_mode != ITG_PAD_REPLICATE) ok = ok && (unsigned)iy < (unsigned)p.in.H && (unsigned)ix < (unsigned)p.in.W;
    iy = min(max(iy, 0), p.in.H - 1);
    ix = min(max(ix, 0), p.in.W - 1);
    unsigned o = (unsigned)grid_off(p.in, pn[i], iy, ix) * 4u;
    return ok ? o : p.in_bytes;
  };
  if constexpr (use_tab) {
    for (int tt = kg; tt <= p.ntaps; tt += KG) {
#pragma unroll
      for (int i = 0; i < PL; ++i)
        if (lrow + i * RPP < BPIX) taptab[(lrow + i * RPP) * TS + tt] = tap_offset(i, tt);
    }
    __syncthreads();
  }
  auto locate = [&]() {
    if constexpr (use_tab) {
#pragma unroll
      for (int i = 0; i < PL; ++i) poff[i] = lrow + i * RPP < BPIX ? taptab[(lrow + i * RPP) * TS + min(tap, p.ntaps)] : p.in_bytes;
    } else {
#pragma unroll
      for (int i = 0; i < PL; ++i) poff[i] = tap_offset(i, tap);
    }
  };
  locate();

  f32x4 rp[DEPTH][PL], rw[DEPTH][WL];
  auto load_tiles = [&](int kk, int set, f32x4 (&rp_)[PL], f32x4 (&rw_v)[WL]) {
#pragma unroll
    for (int i = 0; i < PL; ++i)
      rp_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, poff[i] + (unsigned)cc * 4u, 0, 0));
    const int ksoff = kk * TBK * 4;
#pragma unroll
    for (int i = 0; i < WL; ++i)
      rw_v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw_, woff[i], ksoff, 0));
    cc += TBK;
    if constexpr (use_tab) {           // narrow layers: (almost) every stage crosses a tap, no wave-uniform test
      while (cc >= p.cin_ld) { cc -= p.cin_ld; ++tap; }
      locate();
    } else if (__any(cc >= p.cin_ld)) {       // wave-uniform: some lane moves on to the next filter tap
      while (cc >= p.cin_ld) { cc -= p.cin_ld; ++tap; }
      locate();
    }
  };
  auto store_tiles = [&](int buf, int set, const f32x4 (&rp_)[PL], const f32x4 (&rw_v)[WL]) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      if (BPIX % RPP != 0 && lrow + i * RPP >= BPIX) continue;
      float* dst = Ps + (buf * BPIX + lrow + i * RPP) * LDT + swz;
      const f32x4 v = rp_[i];
      if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pack_bf16x4(v);
      else *reinterpret_cast<f32x4*>(dst) = v;
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      int row = lrow + i * RPP;
      if (row < BCO) {
        float* dst = Ws + (buf * BCO + row) * LDT + swz;
        if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pack_bf16x4(rw_v[i]);
        else *reinterpret_cast<f32x4*>(dst) = rw_v[i];
      }
    }
  };

  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr bool W64 = MODE == NT_W64 || MODE == NT_W32;      // blocked accumulation; the block sums in fp64 or fp32
  static_assert(!W64 || !BF, "blocked accumulation is an fp32-operand mode");
  using wide4 = std::conditional_t<MODE == NT_W64, f64x4, f32x4>;
  using wide1 = std::conditional_t<MODE == NT_W64, double, float>;
  wide4 acc64[W64 ? FI : 1][W64 ? FJ : 1];
  if constexpr (W64) {
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc64[i][j] = wide4{0, 0, 0, 0};
  }

  const int frow = lane & 15, fk = ((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 4;   // swizzled slot of K group lane >> 4 in row frow
  auto compute = [&](int buf) {
    {
      f32x4 a[FI], b[FJ];
#pragma unroll
      for (int i = 0; i < FI; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(Ws + (buf * BCO + wco0 + 16 * i + frow) * LDT + fk);
#pragma unroll
      for (int j = 0; j < FJ; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(Ps + (buf * BPIX + wpix0 + 16 * j + frow) * LDT + fk);
      if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int j = 0; j < FJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
      } else if constexpr (W64) {
        // chains of ITG_W64_BLOCK / 4 MFMAs from a zero accumulator, summed in fp64; two fragments' chains are written
        // side by side so that consecutive MFMAs are independent, and only their 8 temporaries are live
        constexpr int CH = ITG_W64_BLOCK / 4;
        static_assert(CH == 1 || CH == 2 || CH == 4, "ITG_W64_BLOCK is 4, 8 or 16");
        static_assert(FJ % 2 == 0 || FJ == 1, "fragment columns are paired");
        constexpr int JP = FJ == 1 ? 1 : 2;
#pragma unroll
        for (int s0 = 0; s0 < 4; s0 += CH)
#pragma unroll
          for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j0 = 0; j0 < FJ; j0 += JP) {
              f32x4 t[JP];
#pragma unroll
              for (int s = s0; s < s0 + CH; ++s)
#pragma unroll
                for (int jj = 0; jj < JP; ++jj)
                  t[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j0 + jj][s], s == s0 ? f32x4{0.f, 0.f, 0.f, 0.f} : t[jj], 0, 0, 0);
#pragma unroll
              for (int jj = 0; jj < JP; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc64[i][j0 + jj][e] += (wide1)t[jj][e];
            }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      }
    }
  };
  if constexpr (DEPTH == 1) {
    load_tiles(kk0, 0, rp[0], rw[0]);
    store_tiles(0, 0, rp[0], rw[0]);
    __syncthreads();
    for (int kk = kk0; kk < kk1; ++kk) {
      const int buf = (kk - kk0) & 1;
      if (kk + 1 < kk1) load_tiles(kk + 1, 0, rp[0], rw[0]);
      compute(buf);
      if (kk + 1 < kk1) store_tiles(buf ^ 1, 0, rp[0], rw[0]);
      __syncthreads();
    }
  } else {
    // two stages in flight: register set A holds stage kk+2 while set B (stage kk+1) drains into LDS
    load_tiles(kk0, 0, rp[0], rw[0]);
    if (kk0 + 1 < kk1) load_tiles(kk0 + 1, DEPTH - 1, rp[DEPTH - 1], rw[DEPTH - 1]);
    store_tiles(0, 0, rp[0], rw[0]);
    __syncthreads();
    for (int kk = kk0; kk < kk1; kk += 2) {
      if (kk + 2 < kk1) load_tiles(kk + 2, 0, rp[0], rw[0]);
      compute(0);
      if (kk + 1 < kk1) store_tiles(1, DEPTH - 1, rp[DEPTH - 1], rw[DEPTH - 1]);
      __syncthreads();
      if (kk + 1 >= kk1) break;
      if (kk + 3 < kk1) load_tiles(kk + 3, DEPTH - 1, rp[DEPTH - 1], rw[DEPTH - 1]);
      compute(1);
      if (kk + 2 < kk1) store_tiles(0, 0, rp[0], rw[0]);
      __syncthreads();
    }
  }

  if constexpr (W64) {
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = (float)acc64[i][j][e];
  }
  // ---- epilogue: lane holds 4 consecutive output channels of one pixel per fragment
  const int cq = (lane >> 4) * 4;
  if (p.ksplit > 1) {
    float* slab = cpartial + (size_t)blockIdx.z * cM * p.co_rows;
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
      int m = m0 + wpix0 + 16 * j + (lane & 15);
      if (m >= cM) continue;
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        int co = co0 + wco0 + 16 * i + cq;
        if (co < p.co_rows) *reinterpret_cast<f32x4*>(slab + (size_t)m * p.co_rows + co) = acc[i][j];
      }
    }
    return;                 // splitk_epilogue_kernel (conv_nt.hip) sums the slabs and runs the epilogue
  }
  if (p.scale) {          // 1/sigma of an unscaled panel (two-launch split-K: applied by the second stage)
    const float osc = *p.scale;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] *= osc;
  }
  // BatchNorm statistics of the consumer layer, taken from the values as they are stored (p.stats)
  f32x4 st1[FI], st2[FI];
#pragma unroll
  for (int i = 0; i < FI; ++i) { st1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    int m = m0 + wpix0 + 16 * j + (lane & 15);
    if (m >= cM) continue;
    int n, t, u;
    decode_m(m, cMT, cMU, n, t, u);
    int oy = t * p.osy + cooy, ox = u * p.osx + coox;
    bool border = false;
    if (p.out_mode == 1) {
      int ty = min(max(oy, 0), p.out.H - 1), tx = min(max(ox, 0), p.out.W - 1);
      border = (ty == 0) | (ty == p.out.H - 1) | (tx == 0) | (tx == p.out.W - 1);
      oy = ty; ox = tx;
    }
    const int off = grid_off(p.out, n, oy, ox);
    const int roff = p.res.p ? grid_off(p.res, n, oy >> p.res_ups, ox >> p.res_ups) : 0;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      int co = co0 + wco0 + 16 * i + cq;
      if (co >= p.out.ld) continue;
      f32x4 v = acc[i][j];
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < p.out.c) v[e] += p.bias[co + e];
      }
      if (p.res.p) {
        f32x4 r = *reinterpret_cast<const f32x4*>(p.res.p + roff + co);
        if (p.res_mode == 0) v += r;
        else v *= act_deriv(r, p.res_mode, p.res_slope);
      }
      if (p.act != ITG_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act, p.slope);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (co + e >= p.out.c) v[e] = 0.f;
      st1[i] += v; st2[i] += v * v;
      float* dst = out_base + off + co;
      if (border) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, v[e]);
      } else {
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  }
  double* const sums_out = p.stats;
  if (sums_out) {      // workgroup-uniform
    // lanes that share lane >> 4 hold the same 4 channels of different pixels: butterfly over the pixel lanes, then
    // fp64 per workgroup in LDS (the K loop's buffers are free: it ended with a barrier), one global atomic per channel
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          st1[i][e] += __shfl_xor(st1[i][e], o, 64);
          st2[i][e] += __shfl_xor(st2[i][e], o, 64);
        }
    double* ls = reinterpret_cast<double*>(smem);            // [2][BCO]
    for (int t = tid; t < 2 * BCO; t += 256) ls[t] = 0.0;
    __syncthreads();
    if ((lane & 15) == 0) {
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(&ls[wco0 + 16 * i + cq + e], (double)st1[i][e]);
          atomicAdd(&ls[BCO + wco0 + 16 * i + cq + e], (double)st2[i][e]);
        }
    }
    __syncthreads();
    for (int t = tid; t < BCO; t += 256) {
      const int co = co0 + t;
      if (co < p.out.ld) {
        atomicAdd(&sums_out[co], ls[t]);
        atomicAdd(&sums_out[p.out.ld + co], ls[BCO + t]);
      }
    }
  }
}

template <int BCO, int BPIX, int WCO, int WPIX, int MODE>
int launch_nt(const ConvP& p, int tbk, hipStream_t s) {
  ConvP q = p;
  q.nco_tiles = (p.co_rows + BCO - 1) / BCO;
  int64_t npix = ((int64_t)p.M + BPIX - 1) / BPIX;
  int64_t blocks = npix * q.nco_tiles;
  if (blocks <= 0 || blocks > 0x7fffffff) return ITG_ERR_ARG;
  const int64_t gx = blocks * (p.ucls ? p.ucls : (p.ncls > 1 ? p.ncls : 1));      // classes inside the tile id (see the kernel)
  if (gx > 0x7fffffff) return ITG_ERR_ARG;
  dim3 grid((unsigned)gx, 1, (unsigned)p.ksplit);
  size_t tab_bytes = (size_t)BPIX * (p.ntaps + 1) * sizeof(unsigned);
  q.use_tab = (p.cin_ld < 64 && tab_bytes <= 24 * 1024) ? 1 : 0;
  q.xcd_remap = 1;
  if (!q.use_tab) tab_bytes = 0;
  // two K stages in flight except for the medium fp32 tiles, whose 96-register budget has no room for
  // the second prefetch set (it would spill into scratch inside the K loop)
  constexpr int D32 = (nt_min_blocks(BCO, BPIX, WCO, WPIX, 32) == 3 && BCO >= 64) ? 1 : 2;   // wide bf16 stages: 16 prefetch registers per set
  constexpr int D16 = 2;
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_nt_kernel<%d, %d, %d, %d, %d, %d, %s, %d>", BCO, BPIX, WCO, WPIX, tbk,
           tbk == 32 ? D32 : D16, q.use_tab ? "true" : "false", MODE);
  if (tbk == 32) {
    if constexpr (MODE != NT_PLAIN) return ITG_ERR_ARG;      // the fused / blocked-accumulation forms exist for fp32 operands only
    else {
    if (q.use_tab) hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 32, D32, true, MODE>), grid, dim3(256), tab_bytes, s, q);
    else hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 32, D32, false, MODE>), grid, dim3(256), tab_bytes, s, q);
    }
  } else {
    if (q.use_tab) hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 16, D16, true, MODE>), grid, dim3(256), tab_bytes, s, q);
    else hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 16, D16, false, MODE>), grid, dim3(256), tab_bytes, s, q);
  }
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// the tile shapes of plan_nt, for one MODE
template <int MODE>
int launch_nt_shape(int bco, int bpix, const ConvP& p, int k, hipStream_t s) {
  if (bco == 16)
    return bpix == 256 ? launch_nt<16, 256, 16, 64, MODE>(p, k, s) : bpix == 128 ? launch_nt<16, 128, 16, 32, MODE>(p, k, s)
                                                                                 : launch_nt<16, 64, 16, 16, MODE>(p, k, s);
  if (bco == 32)
    return bpix == 256 ? launch_nt<32, 256, 32, 64, MODE>(p, k, s) : bpix == 128 ? launch_nt<32, 128, 32, 32, MODE>(p, k, s)
                                                                                 : launch_nt<32, 64, 32, 16, MODE>(p, k, s);
  if (bco == 64)
    return bpix == 256 ? launch_nt<64, 256, 64, 64, MODE>(p, k, s) : bpix == 128 ? launch_nt<64, 128, 64, 32, MODE>(p, k, s)
                                                                                 : launch_nt<64, 64, 32, 32, MODE>(p, k, s);
  if (bco == 112) return bpix == 128 ? launch_nt<112, 128, 112, 32, MODE>(p, k, s) : launch_nt<112, 64, 112, 16, MODE>(p, k, s);
  return bpix == 128 ? launch_nt<128, 128, 64, 64, MODE>(p, k, s)
         : bpix == 96 ? launch_nt<128, 96, 64, 48, MODE>(p, k, s) : launch_nt<128, 64, 64, 32, MODE>(p, k, s);
}

// NT_W64: tile shapes with <= 8 fragments per wave (fp64 accumulators take two registers per element)
inline int w64_bpix(int bco, int bpix) {
  if (bco >= 112) return 64;
  if (bco == 64) return bpix > 128 ? 128 : bpix;
  return bpix;
}
template <int MODE>
inline int launch_nt_shape_w64(int bco, int bpix, const ConvP& p, int k, hipStream_t s) {
  bpix = w64_bpix(bco, bpix);
  if (bco == 16)
    return bpix == 256 ? launch_nt<16, 256, 16, 64, MODE>(p, k, s) : bpix == 128 ? launch_nt<16, 128, 16, 32, MODE>(p, k, s)
                                                                                 : launch_nt<16, 64, 16, 16, MODE>(p, k, s);
  if (bco == 32)
    return bpix == 256 ? launch_nt<32, 256, 32, 64, MODE>(p, k, s) : bpix == 128 ? launch_nt<32, 128, 32, 32, MODE>(p, k, s)
                                                                                 : launch_nt<32, 64, 32, 16, MODE>(p, k, s);
  if (bco == 64) return bpix == 128 ? launch_nt<64, 128, 64, 32, MODE>(p, k, s) : launch_nt<64, 64, 32, 32, MODE>(p, k, s);
  if (bco == 112) return launch_nt<112, 64, 112, 16, MODE>(p, k, s);
  return launch_nt<128, 64, 64, 32, MODE>(p, k, s);
}

}  // namespace itgk
