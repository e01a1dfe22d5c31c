// Implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 for gfx950 (MI355X): forward and input gradient.
//
//   conv_nt_kernel : C[co][pixel] = sum_k W[co][k] * P[pixel][k], k = (tap, ci) with ci innermost; both operands
//                    K-contiguous, staged global -> registers -> LDS (double buffered), fragments read with
//                    ds_read_b128 under a K permutation (lane group g holds k = 4g..4g+3, MFMA step s consumes
//                    component s of every group).
// The pixel operand is gathered in merged-image coordinates from a patch-grid NHWC tensor, so the LocalPadder halo
// (reference models/layers.py:145-173) is a neighbour-patch read and the outer replicate / zero padding (layers.py:82)
// a clamp / predicate; nothing is materialised.
#include "conv_common.h"

namespace itgk {

thread_local char g_last_launch[96] = "";

// DEPTH = number of K stages whose global loads are in flight while one stage is computed.
// TBK = K elements per stage: 16 -> fp32 operands on v_mfma_f32_16x16x4_f32; 32 -> operands rounded to
// bf16 when they are staged into LDS (tensors stay fp32 in HBM) and contracted by ONE
// v_mfma_f32_16x16x32_bf16 per fragment pair and stage, fp32 accumulation (BASELINE config 3's path).
// Either way a tile row occupies 16 dwords of a 20-dword LDS row and lane group g reads dwords 4g..4g+3.
// Workgroups per CU the register budget is pinned to: 3 (168 VGPRs) for the wide tiles, 5 (96) for the
// medium fp32 tiles, 4 (128) for the medium bf16 tiles (their stage holds twice the prefetch registers).
constexpr int nt_min_blocks(int bco, int bpix, int wco, int wpix, int tbk) {
  return ((wco / 16) * (wpix / 16) <= 8 && (bco + bpix) <= 192) ? 4 : 3;
}

template <int BCO, int BPIX, int WCO, int WPIX, int TBK, int DEPTH, bool TAB>
__global__ __launch_bounds__(256, nt_min_blocks(BCO, BPIX, WCO, WPIX, TBK)) void conv_nt_kernel(const ConvP p) {
  // per-class geometry (class 0 for ordinary launches); all wave-uniform scalars
  const int cls = blockIdx.y;
  const int cMT = p.cMT[cls], cMU = p.cMU[cls], cM = p.cM[cls];
  const int cioy = p.cioy[cls], ciox = p.ciox[cls], cooy = p.cooy[cls], coox = p.coox[cls];
  const float* const cw = p.w + p.cwoff[cls];
  float* const cpartial = p.partial + p.cpoff[cls];
  // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2): hand every XCD one contiguous run of
  // tile ids instead, so that an L2 serves neighbouring pixel tiles (shared halo rows, all output-channel tiles of a
  // pixel tile) and not a 1-in-8 sample of the whole image.  Bijective for any grid size; speed only.
  int bx = blockIdx.x;
  if (p.xcd_remap) {
    const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = bx & 7;
    bx = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bx >> 3);
  }
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
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in.p, 0, p.in_bytes, 0x00020000);
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
  auto load_tiles = [&](int kk, f32x4 (&rp_)[PL], f32x4 (&rw_v)[WL]) {
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
  auto store_tiles = [&](int buf, const f32x4 (&rp_)[PL], const f32x4 (&rw_v)[WL]) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      if (BPIX % RPP != 0 && lrow + i * RPP >= BPIX) continue;
      float* dst = Ps + (buf * BPIX + lrow + i * RPP) * LDT + swz;
      if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pack_bf16x4(rp_[i]);
      else *reinterpret_cast<f32x4*>(dst) = rp_[i];
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
    load_tiles(kk0, rp[0], rw[0]);
    store_tiles(0, rp[0], rw[0]);
    __syncthreads();
    for (int kk = kk0; kk < kk1; ++kk) {
      const int buf = (kk - kk0) & 1;
      if (kk + 1 < kk1) load_tiles(kk + 1, rp[0], rw[0]);
      compute(buf);
      if (kk + 1 < kk1) store_tiles(buf ^ 1, rp[0], rw[0]);
      __syncthreads();
    }
  } else {
    // two stages in flight: register set A holds stage kk+2 while set B (stage kk+1) drains into LDS
    load_tiles(kk0, rp[0], rw[0]);
    if (kk0 + 1 < kk1) load_tiles(kk0 + 1, rp[1], rw[1]);
    store_tiles(0, rp[0], rw[0]);
    __syncthreads();
    for (int kk = kk0; kk < kk1; kk += 2) {
      if (kk + 2 < kk1) load_tiles(kk + 2, rp[0], rw[0]);
      compute(0);
      if (kk + 1 < kk1) store_tiles(1, rp[1], rw[1]);
      __syncthreads();
      if (kk + 1 >= kk1) break;
      if (kk + 3 < kk1) load_tiles(kk + 3, rp[1], rw[1]);
      compute(1);
      if (kk + 2 < kk1) store_tiles(0, rp[0], rw[0]);
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds 4 consecutive output channels of one pixel per fragment
  const int cq = (lane >> 4) * 4;
  if (p.scale && p.ksplit <= 1) {          // 1/sigma of an unscaled panel (split-K: applied by the second stage)
    const float osc = *p.scale;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] *= osc;
  }
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
    return;
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
    const int roff = p.res.p ? grid_off(p.res, n, oy, ox) : 0;
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
      float* dst = p.out.p + off + co;
      if (border) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, v[e]);
      } else {
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  }
  if (p.stats) {      // workgroup-uniform
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
        atomicAdd(&p.stats[co], ls[t]);
        atomicAdd(&p.stats[p.out.ld + co], ls[BCO + t]);
      }
    }
  }
}

// zero the 1-pixel frame of the merged image (targets of the fold-mode atomics)
__global__ void zero_border_kernel(GridT g) {
  int per = 2 * g.W + 2 * (g.H - 2 > 0 ? g.H - 2 : 0);
  int q4 = g.ld >> 2;
  int64_t total = (int64_t)g.n * per * q4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % q4);
    int64_t r = i / q4;
    int b = (int)(r % per);
    int n = (int)(r / per);
    int Y, X;
    if (b < g.W) { Y = 0; X = b; }
    else if (b < 2 * g.W) { Y = g.H - 1; X = b - g.W; }
    else { int k = b - 2 * g.W; Y = 1 + (k >> 1); X = (k & 1) ? g.W - 1 : 0; }
    if (g.H == 1 && b >= g.W) continue;
    *reinterpret_cast<f32x4*>(g.p + grid_off(g, n, Y, X) + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// split-K second stage: out = act(sum_z partial[z] + bias [+ residual]) with the same output mapping
__global__ void splitk_epilogue_kernel(ConvP p) {
  const int q4 = p.out.ld >> 2;
  int64_t total = (int64_t)p.M * q4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c4 = (int)(i % q4);
    int m = (int)(i / q4);
    int co = c4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    {
      const float* q = p.partial + (size_t)m * p.co_rows + co;
      const size_t zs = (size_t)p.M * p.co_rows;
      f32x4 v1 = v, v2 = v, v3 = v;                       // four slab loads in flight, fixed summation order
      int z = 0;
      for (; z + 4 <= p.ksplit; z += 4) {
        v += *reinterpret_cast<const f32x4*>(q + (size_t)z * zs);
        v1 += *reinterpret_cast<const f32x4*>(q + (size_t)(z + 1) * zs);
        v2 += *reinterpret_cast<const f32x4*>(q + (size_t)(z + 2) * zs);
        v3 += *reinterpret_cast<const f32x4*>(q + (size_t)(z + 3) * zs);
      }
      for (; z < p.ksplit; ++z) v += *reinterpret_cast<const f32x4*>(q + (size_t)z * zs);
      v = (v + v1) + (v2 + v3);
    }
    if (p.scale) v *= *p.scale;
    int n, t, u;
    decode_m(m, p.MT, p.MU, n, t, u);
    int oy = t * p.osy + p.ooy, ox = u * p.osx + p.oox;
    bool border = false;
    if (p.out_mode == 1) {
      int ty = min(max(oy, 0), p.out.H - 1), tx = min(max(ox, 0), p.out.W - 1);
      border = (ty == 0) | (ty == p.out.H - 1) | (tx == 0) | (tx == p.out.W - 1);
      oy = ty; ox = tx;
    }
    const int off = grid_off(p.out, n, oy, ox);
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (co + e < p.out.c) v[e] += p.bias[co + e];
    }
    if (p.res.p) {
      f32x4 r = *reinterpret_cast<const f32x4*>(p.res.p + grid_off(p.res, n, oy, ox) + co);
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
    float* dst = p.out.p + off + co;
    if (border) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(dst + e, v[e]);
    } else {
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  }
}

int launch_zero_border(const GridT& gx, hipStream_t s) {
  int64_t tot = (int64_t)gx.n * (2 * gx.W + 2 * gx.H) * (gx.ld >> 2);
  int blocks = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
  hipLaunchKernelGGL(zero_border_kernel, dim3(blocks), dim3(256), 0, s, gx);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// Tile / split-K plan.  The chip has 256 CUs; every workgroup is 4 waves (one per SIMD), so a CU's
// time is (#workgroups it runs) x (work of one), and equal-sized workgroups quantise badly when
// their count is a small non-multiple of 256.  Pick the pixel-tile width that minimises
// ceil(blocks / 256) * tile work, then split K when the grid still under-fills the chip.
NtPlan plan_nt(int64_t M_total, int co_rows, int Kpad, int ncls, int prec) {
  NtPlan pl;
  const int64_t M = M_total / ncls;      // per-class pixel count (classes are launched as one grid)
  pl.tbk = prec == ITG_PREC_BF16 ? 32 : 16;
  if (co_rows <= 16) pl.bco = 16;
  else if (co_rows <= 32) pl.bco = 32;
  else if (co_rows <= 64) pl.bco = 64;
  else {
    // channel counts of this model are multiples of 13 (104, 208, 416): 112-row tiles waste 7 % of the
    // MFMA rows where 128-row tiles waste 19 %
    const int pad128 = (co_rows + 127) / 128 * 128, pad112 = (co_rows + 111) / 112 * 112;
    pl.bco = pad112 < pad128 ? 112 : 128;
  }
  const int nco = (co_rows + pl.bco - 1) / pl.bco;
  const int nk = (Kpad + pl.tbk - 1) / pl.tbk;
  // Joint choice of the pixel-tile width and the K split.  Efficiency model per candidate:
  //   quantisation  (B*ks/256) / ceil(B*ks/256)      equal-sized workgroups on 256 CUs
  //   fill          < 2 workgroups per CU leaves the MFMA pipe idle between phases
  //   split cost    the second stage's slab round trip ~ ks * 100 / K of the kernel's own time
  //   tile penalty  narrower tiles re-read the weight panel more often and carry more issue overhead
  static const double split_cost = (double)env_int("ITG_SPLIT_COST", 200);
  pl.bpix = 128; pl.ksplit = 1;
  double best_eff = 0.0;
  const int cands_big[4] = {256, 128, 96, 64};
  const int cand_ks[13] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16};
  static const int fill_env = env_int("ITG_FILL_MIN", 0);      // workgroups per CU below which a launch counts as under-filled
  const double fill_min = (fill_env ? fill_env : (prec == ITG_PREC_BF16 ? 200 : 300)) / 100.0;   // bf16 stages are short: 2 (config 3 +1 %)
  static const int allow96 = env_int("ITG_NT_96", 1);
  static const double pen96 = env_int("ITG_PEN96", 104) / 100.0;
  for (int ci = 0; ci < 4; ++ci) {
    int bp = cands_big[ci];
    if (pl.bco >= 112 && bp == 256) continue;              // 128x256 / 112x256 are not instantiated
    if (bp == 96 && (pl.bco != 128 || !allow96 || pl.tbk != 16)) continue;   // 128x96 (fp32): 3 workgroups per CU exactly on M = 73728
    int64_t blocks = ((M + bp - 1) / bp) * nco * ncls;
    double pen = bp >= 256 ? 1.0 : (bp == 128 ? (pl.bco >= 112 ? 1.0 : 1.04) : bp == 96 ? pen96 : (pl.bco >= 112 ? 1.08 : 1.12));
    for (int i = 0; i < 13; ++i) {
      int ks = cand_ks[i];
      if (ks > 1 && nk * pl.tbk / ks < 256) break;
      double b = (double)blocks * ks / 256.0;
      double eff = b / (double)((int64_t)(b + 0.999999));
      if (b < fill_min) eff *= b / fill_min;
      if (ks > 1) eff /= 1.0 + ks * split_cost / (double)Kpad;
      eff /= pen;
      if (eff > best_eff * 1.02) { best_eff = eff; pl.bpix = bp; pl.ksplit = ks; }
    }
  }
  pl.kchunks = (nk + pl.ksplit - 1) / pl.ksplit;
  pl.ksplit = (nk + pl.kchunks - 1) / pl.kchunks;
  pl.ws_floats = pl.ksplit > 1 ? (int64_t)pl.ksplit * M * co_rows * ncls : 0;
  return pl;
}

template <int BCO, int BPIX, int WCO, int WPIX>
int launch_nt(const ConvP& p, int tbk, hipStream_t s) {
  ConvP q = p;
  q.nco_tiles = (p.co_rows + BCO - 1) / BCO;
  int64_t npix = ((int64_t)p.M + BPIX - 1) / BPIX;
  int64_t blocks = npix * q.nco_tiles;
  if (blocks <= 0 || blocks > 0x7fffffff) return ITG_ERR_ARG;
  dim3 grid((unsigned)blocks, (unsigned)(p.ncls > 1 ? p.ncls : 1), (unsigned)p.ksplit);
  size_t tab_bytes = (size_t)BPIX * (p.ntaps + 1) * sizeof(unsigned);
  q.use_tab = (p.cin_ld < 64 && tab_bytes <= 24 * 1024) ? 1 : 0;
  static const int xcd = env_int("ITG_NT_XCD", 1);
  q.xcd_remap = xcd;
  if (!q.use_tab) tab_bytes = 0;
  // two K stages in flight except for the medium fp32 tiles, whose 96-register budget has no room for
  // the second prefetch set (it would spill into scratch inside the K loop)
  constexpr int D32 = (nt_min_blocks(BCO, BPIX, WCO, WPIX, 32) == 3 && BCO >= 64) ? 1 : 2;   // wide bf16 stages: 16 prefetch registers per set
  constexpr int D16 = 2;
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_nt_kernel<%d, %d, %d, %d, %d, %d, %s>", BCO, BPIX, WCO, WPIX, tbk,
           tbk == 32 ? D32 : D16, q.use_tab ? "true" : "false");
  if (tbk == 32) {
    if (q.use_tab) hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 32, D32, true>), grid, dim3(256), tab_bytes, s, q);
    else hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 32, D32, false>), grid, dim3(256), 0, s, q);
  } else {
    if (q.use_tab) hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 16, D16, true>), grid, dim3(256), tab_bytes, s, q);
    else hipLaunchKernelGGL((conv_nt_kernel<BCO, BPIX, WCO, WPIX, 16, D16, false>), grid, dim3(256), 0, s, q);
  }
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

// the statistics pass as its own launch over the finished output (paths whose epilogue does not take them)
int stats_after(const ConvP& p, double* stats, hipStream_t s) {
  itg_tensor t = {p.out.p, p.out.n, p.out.gh, p.out.gw, p.out.ph, p.out.pw, p.out.c, p.out.ld};
  return itg_bn_stats(&t, stats, s);
}

int dispatch_nt(ConvP p, float* workspace, int64_t workspace_floats, hipStream_t s) {
  // ITG_STATS_PATHS: bit 0 halo-tile kernel, bit 1 implicit-GEMM epilogue take the consumer BatchNorm's statistics
  // themselves; a cleared bit - and always the split-K second stage, where fusing them measured slower - runs the
  // separate statistics launch over the finished output instead.  (Measured neutral on the step: 764.9 vs 764.3 crops/s;
  // it removes 8 of the 13 statistics launches of a generator forward.)
  static const int stats_paths = env_int("ITG_STATS_PATHS", 3);
  double* const want_stats = p.stats;
  {
    int rc_v = ITG_OK;
    if (try_conv_valu(p, s, &rc_v)) return rc_v;
  }
  {
    int rc_tile = ITG_OK;
    ConvP pt = p;
    if (!(stats_paths & 1)) pt.stats = nullptr;
    if (try_conv_tile(pt, s, &rc_tile)) {
      if (rc_tile == ITG_OK && want_stats && !pt.stats) return stats_after(p, want_stats, s);
      return rc_tile;
    }
  }
  const int ncls_ = p.ncls > 1 ? p.ncls : 1;
  if (ncls_ == 1) {
    p.cMT[0] = p.MT; p.cMU[0] = p.MU; p.cM[0] = p.M;
    p.cioy[0] = p.ioy; p.ciox[0] = p.iox; p.cooy[0] = p.ooy; p.coox[0] = p.oox;
    p.cwoff[0] = 0;
  }
  NtPlan pl = plan_nt((int64_t)p.M * ncls_, p.co_rows, p.Kpad, ncls_, p.prec);
  if (pl.ws_floats > workspace_floats || (pl.ws_floats && !workspace)) return ITG_ERR_WORKSPACE;
  p.ksplit = pl.ksplit; p.kchunks = pl.kchunks; p.partial = workspace;
  if (pl.ksplit > 1 || !(stats_paths & 2)) p.stats = nullptr;
  for (int c = 0; c < 4; ++c) p.cpoff[c] = (unsigned)((size_t)c * pl.ksplit * p.M * p.co_rows);
  {
    int64_t ib = (int64_t)p.in.n * p.in.gh * p.in.gw * p.in.ph * p.in.pw * p.in.ld * 4;
    int64_t wb = (int64_t)p.co_rows * p.Kpad * 4;
    if (ib >= 0xFFFF0000LL || wb >= 0xFFFF0000LL) return ITG_ERR_ARG;   // 32-bit buffer offsets
    p.in_bytes = (unsigned)ib; p.w_bytes = (unsigned)wb;
  }
  const int k = pl.tbk;
  static const int plan_debug = env_int("ITG_PLAN_DEBUG", 0);
  if (plan_debug)
    fprintf(stderr, "[nt] M=%d x%d co_rows=%d Kpad=%d -> bco=%d bpix=%d ksplit=%d kchunks=%d\n", p.M, ncls_, p.co_rows, p.Kpad,
            pl.bco, pl.bpix, pl.ksplit, pl.kchunks);
  int rc;
  if (pl.bco == 16) {
    rc = pl.bpix == 256 ? launch_nt<16, 256, 16, 64>(p, k, s) : pl.bpix == 128 ? launch_nt<16, 128, 16, 32>(p, k, s)
                                                                                : launch_nt<16, 64, 16, 16>(p, k, s);
  } else if (pl.bco == 32) {
    rc = pl.bpix == 256 ? launch_nt<32, 256, 32, 64>(p, k, s) : pl.bpix == 128 ? launch_nt<32, 128, 32, 32>(p, k, s)
                                                                                : launch_nt<32, 64, 32, 16>(p, k, s);
  } else if (pl.bco == 64) {
    rc = pl.bpix == 256 ? launch_nt<64, 256, 64, 64>(p, k, s) : pl.bpix == 128 ? launch_nt<64, 128, 64, 32>(p, k, s)
                                                                                : launch_nt<64, 64, 32, 32>(p, k, s);
  } else if (pl.bco == 112) {
    rc = pl.bpix == 128 ? launch_nt<112, 128, 112, 32>(p, k, s) : launch_nt<112, 64, 112, 16>(p, k, s);
  } else {
    rc = pl.bpix == 128 ? launch_nt<128, 128, 64, 64>(p, k, s)
         : pl.bpix == 96 ? launch_nt<128, 96, 64, 48>(p, k, s) : launch_nt<128, 64, 64, 32>(p, k, s);
  }
  if (rc) return rc;
  if (pl.ksplit == 1) return (want_stats && !p.stats) ? stats_after(p, want_stats, s) : ITG_OK;
  const int ncls = p.ncls > 1 ? p.ncls : 1;
  for (int c = 0; c < ncls; ++c) {
    ConvP q = p;
    if (p.ncls > 1) {
      q.MT = p.cMT[c]; q.MU = p.cMU[c]; q.M = p.cM[c];
      q.ioy = p.cioy[c]; q.iox = p.ciox[c]; q.ooy = p.cooy[c]; q.oox = p.coox[c];
      q.partial = p.partial + p.cpoff[c];
      if (q.M <= 0) continue;
    }
    int64_t total = (int64_t)q.M * (q.out.ld >> 2);
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, q);
    ITG_CHECK_LAUNCH();
  }
  return (want_stats && !p.stats) ? stats_after(p, want_stats, s) : ITG_OK;
}


}  // namespace itgk
