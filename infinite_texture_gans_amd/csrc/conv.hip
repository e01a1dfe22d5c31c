// Implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 for gfx950 (MI355X).
//
//   conv_nt_kernel : forward and input-gradient.  C[co][pixel] = sum_k W[co][k] * P[pixel][k],
//                    k = (tap, ci) with ci innermost; both operands K-contiguous, staged
//                    global -> registers -> LDS (double buffered), fragments read with
//                    ds_read_b128 under a K permutation (lane group g holds k = 4g..4g+3,
//                    MFMA step s consumes component s of every group).
//   conv_tn_kernel : weight-gradient.  C[(tap,ci)][co] = sum_pixel X[pixel+tap][ci] * dY[pixel][co],
//                    split over pixel ranges (grid.z) into fp32 slabs, reduced by
//                    wgrad_reduce_kernel into the OIHW gradient.
//
// The pixel operand is gathered in merged-image coordinates from a patch-grid NHWC tensor, so
// the LocalPadder halo (reference models/layers.py:145-173) is a neighbour-patch read and the
// outer replicate / zero padding (layers.py:82) a clamp / predicate; nothing is materialised.
#include <cstdio>
#include <cstdlib>
#include "itg_common.h"

extern "C" int itg_bn_stats(const itg_tensor* x, double* sums, void* stream);

namespace {

constexpr int BK = 16;    // K elements per pipeline stage
constexpr int LDK = 20;   // LDS row pitch (floats): BK + 4 keeps rows 16-B aligned

// name of the GEMM kernel instantiation launched by this thread's last conv call, exactly as a profiler prints it
thread_local char g_last_launch[96] = "";

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

struct ConvP {
  GridT in, out, res;
  const float* w;
  const float* bias;
  const float* scale;           // one device float multiplied into the contraction (1/sigma), or null
  int ntaps, kw, cin_ld, Kpad;
  int MT, MU, M;
  int isy, ioy, isx, iox;
  int osy, ooy, osx, oox;
  int pad_mode, out_mode, act;
  float slope;
  int res_mode;                 // 0: out += res;  ITG_ACT_*: out *= act'(res), res = the activation's OUTPUT (fused act backward)
  float res_slope;
  int co_rows, nco_tiles;
  unsigned in_bytes, w_bytes;   // buffer-resource extents of the pixel operand / packed weights
  int use_tab;                  // per-row tap-offset table in LDS (narrow layers)
  int xcd_remap;                // deal contiguous runs of tiles to each XCD (its L2 then sees 1/8 of the pixel tiles)
  double* stats;                // fwd only, or null: [2][out.ld] per-channel sum / sum of squares of the stored output
  int prec;                     // ITG_PREC_F32 | ITG_PREC_BF16
  float* partial;      // split-K slabs [ksplit][M][co_rows] (ksplit > 1)
  int ksplit, kchunks; // K chunks (of BK) per split
  // stride-2 input-gradient: the 4 output-parity classes run as ONE grid (blockIdx.y = class)
  int ncls;
  int cMT[4], cMU[4], cM[4], cioy[4], ciox[4], cooy[4], coox[4];
  unsigned cwoff[4];   // float offset of the class's packed sub-kernel
  unsigned cpoff[4];   // float offset of the class's split-K slabs
};

// derivative of an activation expressed through its OUTPUT o (as itg_act_bwd does)
__device__ __forceinline__ f32x4 act_deriv(f32x4 o, int act, float slope) {
  f32x4 d;
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = act == ITG_ACT_LRELU ? (o[e] > 0.f ? 1.f : slope) : (act == ITG_ACT_TANH ? 1.f - o[e] * o[e] : 1.f);
  return d;
}

__device__ __forceinline__ void decode_m(int m, int MT, int MU, int& n, int& t, int& u) {
  int per = MT * MU;
  n = m / per;
  int r = m - n * per;
  t = r / MU;
  u = r - t * MU;
}

// DEPTH = number of K stages whose global loads are in flight while one stage is computed.
// TBK = K elements per stage: 16 -> fp32 operands on v_mfma_f32_16x16x4_f32; 32 -> operands rounded to
// bf16 when they are staged into LDS (tensors stay fp32 in HBM) and contracted by ONE
// v_mfma_f32_16x16x32_bf16 per fragment pair and stage, fp32 accumulation (BASELINE config 3's path).
// Either way a tile row occupies 16 dwords of a 20-dword LDS row and lane group g reads dwords 4g..4g+3.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint2 pack_bf16x4(f32x4 v) {
  bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  return __builtin_bit_cast(uint2, h);
}

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

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__device__ __forceinline__ int round_up_d(int x, int m) { return (x + m - 1) / m * m; }
// ------------------------------------------------------------------------------- small-channel 3x3 (LDS halo tile)
// Stride-1 3x3 convolutions with <= 32 input and <= 32 output channels (the generator's last blocks
// and `final`, forward and input-gradient).  The implicit-GEMM kernel above re-gathers every input
// pixel 9 times through L2; here a workgroup stages one (8+2) x (32+2) pixel halo tile ONCE into LDS
// (coalesced NHWC rows, raw buffer loads with hardware zero-fill / clamped replicate coordinates), keeps
// the whole filter bank in LDS, and every wave runs its 64 pixels x 9 taps on MFMA from there.
constexpr int TT_H = 8, TT_W = 32;
constexpr int TT_PIX = (TT_H + 2) * (TT_W + 2);

// Epilogue of the persistent tile kernels.  Bias and 1/sigma are loaded ONCE per workgroup and the residual tile is
// fetched BEFORE the next tile's prefetch is issued: an epilogue that loads anything would wait vmcnt(0) and with it
// drain the prefetch that is meant to stay in flight across the tile boundary.
__device__ __forceinline__ f32x4 store_out(const ConvP& p, int n, int oy, int ox, int co, f32x4 v, float osc, f32x4 biasv,
                                           bool has_res, f32x4 r) {
  bool border = false;
  if (p.out_mode == 1) {
    int ty = min(max(oy, 0), p.out.H - 1), tx = min(max(ox, 0), p.out.W - 1);
    border = (ty == 0) | (ty == p.out.H - 1) | (tx == 0) | (tx == p.out.W - 1);
    oy = ty; ox = tx;
  }
  const int off = grid_off(p.out, n, oy, ox);
  v = v * osc + biasv;
  if (has_res) {
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
  return v;
}

// Persistent workgroups: the filter bank is staged into LDS once per workgroup, then the workgroup walks
// over tiles; the global loads of the NEXT tile are issued into registers before the current tile's MFMA
// phase and drained into the (single) LDS tile buffer after it.
// LDS: Wl[nch][16*FI][20] (K chunk q of the packed panel, 16 k per row) | koff[nch][4] | Xt[TT_PIX][cpt].
// K index k = 16 q + 4 g + e maps to (tap, c) = divmod(k, cin_ld); cin_ld % 4 == 0 keeps the four e of a lane
// in one tap, so lane group g of chunk q reads 16 B at pixel * cpt + koff[q][g].
// NLD = b128 loads per thread and tile = ceil(340 * (cin_ld / 4) / 256): 6 up to cin_ld 16, 11 up to 32
// STATS: also accumulate the consumer BatchNorm's statistics (p.stats) - its own instantiation, so that the input-gradient
// and plain forward launches keep the register budget they were tuned to
template <int FI, int NLD, bool STATS>
__global__ __launch_bounds__(256, (NLD <= 6 ? 4 : 3)) void conv_tile_kernel(const ConvP p, int tiles_x, int tiles_y, int ntiles, int cpt, int nch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_rows = 16 * FI;
  float* Wl = lds;
  int* koff = reinterpret_cast<int*>(lds + nch * co_rows * 20);
  float* biasl = lds + nch * co_rows * 20 + ((nch * 4 + 3) & ~3);       // [32] bias per output row (zero past out.c)
  double* lstat = reinterpret_cast<double*>(biasl + 32);                // [2][32] BatchNorm sums of this workgroup (p.stats)
  float* Xt = biasl + 32 + 128;
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
    for (int i = 0; i < NLD; ++i)
      if (e_r[i] >= 0) *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = rt[i];
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
        const float* rp = p.res.p + grid_off(p.res, n, oy, ox) + g * 4;
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
          r = *reinterpret_cast<const f32x4*>(p.res.p + grid_off(p.res, n, oy, ox) + co);
        }
        const f32x4 v = store_out(p, n, t * p.osy + p.ooy, u * p.osx + p.oox, co, acc[i][f], osc,
                                  *reinterpret_cast<const f32x4*>(biasl + co), has_res, r);
        if constexpr (STATS) { ts1[i] += v; ts2[i] += v * v; }
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
    if (tid < 32 && tid < p.out.ld) {
      atomicAdd(&p.stats[tid], lstat[tid]);
      atomicAdd(&p.stats[p.out.ld + tid], lstat[32 + tid]);
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
    if (has_res) r = *reinterpret_cast<const f32x4*>(p.res.p + grid_off(p.res, n, ry, rx) + o * 4);
    store_out(p, n, oy, ox, o * 4, acc[o], osc, bv, has_res, r);
  }
}

// eligibility + launch of the vector-ALU kernel; returns 1 when it handled the call
int try_conv_valu(const ConvP& p, hipStream_t s, int* rc) {
  static const int enable = env_int("ITG_CONV_VALU", 1);
  if (!enable || p.ncls > 1 || p.ntaps != 9 || p.kw != 3 || p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1) return 0;
  if (p.prec != ITG_PREC_F32 || p.stats || p.co_rows != 16) return 0;
  const int ci4 = p.cin_ld >> 2, co4 = p.out.ld >> 2;
  // (4 input groups, 1 output group) = the forward of `final`: 44 us against 68 us on the halo-tile MFMA kernel; its
  // input gradient (1, 4) measured slower here (61 vs 41 us) and stays on the tile kernel
  static const int dgrad_too = env_int("ITG_CONV_VALU_DGRAD", 0);
  if (!((ci4 == 4 && co4 == 1) || (dgrad_too && ci4 == 1 && (co4 == 4 || co4 == 1)))) return 0;
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
  if (ci4 == 4) hipLaunchKernelGGL((conv_valu_kernel<4, 1>), dim3((unsigned)ntiles), dim3(256), lds, s, q, tiles_x, tiles_y, cpt);
  else if (co4 == 4) hipLaunchKernelGGL((conv_valu_kernel<1, 4>), dim3((unsigned)ntiles), dim3(256), lds, s, q, tiles_x, tiles_y, cpt);
  else hipLaunchKernelGGL((conv_valu_kernel<1, 1>), dim3((unsigned)ntiles), dim3(256), lds, s, q, tiles_x, tiles_y, cpt);
  *rc = hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
  return 1;
}

// eligibility + launch of the halo-tile kernel; returns 1 when it handled the call
int try_conv_tile(ConvP& p, hipStream_t s, int* rc) {
  static const int enable = env_int("ITG_CONV_TILE", 1);
  if (!enable || p.ncls > 1 || p.ntaps != 9 || p.kw != 3 || p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1)
    return 0;
  if (p.prec != ITG_PREC_F32 || p.cin_ld > 32 || p.co_rows > 32) return 0;
  if ((int64_t)p.MT * p.MU < 64 * 64) return 0;          // tiny images: the gather kernel with split-K wins
  const int FI = p.co_rows / 16;
  if (FI == 2) p.stats = nullptr;      // two row tiles + statistics spill (17-24 VGPRs): the caller runs the separate pass
  ConvP q = p;
  int64_t ib = (int64_t)p.in.n * p.in.gh * p.in.gw * p.in.ph * p.in.pw * p.in.ld * 4;
  if (ib >= 0xFFFF0000LL) return 0;
  q.in_bytes = (unsigned)ib;
  q.scale = p.scale;
  const int cpt = (p.cin_ld % 8 == 4) ? p.cin_ld : p.cin_ld + 4;     // conflict-free b128 fragment reads
  const int nch = (9 * p.cin_ld + 15) / 16;
  const int tiles_x = (p.MU + TT_W - 1) / TT_W, tiles_y = (p.MT + TT_H - 1) / TT_H;
  const int64_t ntiles = (int64_t)p.in.n * tiles_x * tiles_y;
  const size_t lds = ((size_t)nch * 16 * FI * 20 + ((nch * 4 + 3) & ~3) + 32 + 128 + (size_t)TT_PIX * cpt) * sizeof(float);
  const int nld = (TT_PIX * (p.cin_ld >> 2) + 255) / 256;
  if (ntiles > 0x7fffffff || lds > 64 * 1024 || nld > 11) return 0;
  const bool st = p.stats != nullptr;
  const void* kern;
  if (FI == 1 && nld <= 6) kern = st ? (const void*)&conv_tile_kernel<1, 6, true> : (const void*)&conv_tile_kernel<1, 6, false>;
  else if (FI == 1) kern = st ? (const void*)&conv_tile_kernel<1, 11, true> : (const void*)&conv_tile_kernel<1, 11, false>;
  else if (nld <= 6) kern = st ? (const void*)&conv_tile_kernel<2, 6, true> : (const void*)&conv_tile_kernel<2, 6, false>;
  else kern = st ? (const void*)&conv_tile_kernel<2, 11, true> : (const void*)&conv_tile_kernel<2, 11, false>;
  static const void* attr_set[8] = {nullptr};
  {
    bool seen = false;
    int slot = 0;
    for (; slot < 8 && attr_set[slot]; ++slot) seen = seen || attr_set[slot] == kern;
    if (!seen && slot < 8) {
      (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr_set[slot] = kern;
    }
  }
  // persistent grid = what is resident at once (register budget: 4 workgroups per CU with 6 loads, 3 with 11)
  int per_cu = (int)((160 * 1024) / lds);
  static const int tile_cu = env_int("ITG_TILE_CU", 0);       // tuning override of the persistent workgroups per CU
  const int reg_cu = tile_cu > 0 ? tile_cu : (nld <= 6 ? 4 : 3);
  if (per_cu > reg_cu) per_cu = reg_cu;
  if (per_cu < 1) per_cu = 1;
  const int64_t want = 256 * (int64_t)per_cu;
  const unsigned blocks = (unsigned)(ntiles < want ? ntiles : want);
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_tile_kernel<%d, %d, %s>", FI, nld <= 6 ? 6 : 11, st ? "true" : "false");
  int a_tx = tiles_x, a_ty = tiles_y, a_nt = (int)ntiles, a_cpt = cpt, a_nch = nch;
  void* args[] = {(void*)&q, &a_tx, &a_ty, &a_nt, &a_cpt, &a_nch};
  (void)hipLaunchKernel(kern, dim3(blocks), dim3(256), args, lds, s);
  *rc = hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
  return 1;
}


// Tile / split-K plan.  The chip has 256 CUs; every workgroup is 4 waves (one per SIMD), so a CU's
// time is (#workgroups it runs) x (work of one), and equal-sized workgroups quantise badly when
// their count is a small non-multiple of 256.  Pick the pixel-tile width that minimises
// ceil(blocks / 256) * tile work, then split K when the grid still under-fills the chip.
struct NtPlan { int bco, bpix, tbk, ksplit, kchunks; int64_t ws_floats; };

NtPlan plan_nt(int64_t M_total, int co_rows, int Kpad, int ncls = 1, int prec = ITG_PREC_F32) {
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


// ------------------------------------------------------------------------------- packing
// fwd: out[co][k], k = (ky*kw+kx)*ci_ld + ci, rows co >= co zero, k >= K zero
__global__ void pack_fwd_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                int co, int ci, int ci_ld, int kh, int kw, int co_pad, int Kpad) {
  int64_t total = (int64_t)co_pad * Kpad;
  float sc = scale ? *scale : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(i % Kpad);
    int o = (int)(i / Kpad);
    int tap = k / ci_ld, c = k - tap * ci_ld;
    float v = 0.f;
    if (o < co && tap < kh * kw && c < ci) {
      int y = tap / kw, x = tap - y * kw;
      v = w[(((size_t)o * ci + c) * kh + y) * kw + x] * sc;
    }
    out[i] = v;
  }
}

// dgrad stride 1: out[ci][k], k = (ky'*kw+kx')*co_ld + co, W[co][ci][kh-1-ky'][kw-1-kx']
// dgrad stride 2: class (ry,rx) major; taps (jy',jx') of the (kh/2 x kw/2) sub-kernel,
//                 ky = ay + 2*(kh/2-1-jy'), ay = (ry+pad)&1 (same for x)
__global__ void pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                  int co, int ci, int co_ld, int kh, int kw, int stride, int pad, int ci_pad,
                                  int Kpad) {
  int ncls = stride * stride;
  int skh = kh / stride, skw = kw / stride;
  int64_t total = (int64_t)ncls * ci_pad * Kpad;
  float sc = scale ? *scale : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(i % Kpad);
    int64_t r = i / Kpad;
    int c_in = (int)(r % ci_pad);
    int cls = (int)(r / ci_pad);
    int ry = cls / stride, rx = cls - ry * stride;
    int tap = k / co_ld, o = k - tap * co_ld;
    float v = 0.f;
    if (c_in < ci && o < co && tap < skh * skw) {
      int jy = tap / skw, jx = tap - jy * skw;
      int ay = (ry + pad) % stride, ax = (rx + pad) % stride;
      int y = ay + stride * (skh - 1 - jy), x = ax + stride * (skw - 1 - jx);
      v = w[(((size_t)o * ci + c_in) * kh + y) * kw + x] * sc;
    }
    out[i] = v;
  }
}


// every packed panel of a model in one launch.  The job table lives in DEVICE memory (it is static for a
// model: built once, no per-launch upload, capturable in a hipGraph): row j = 10 x int64
// {w_oihw, out, co, ci, ld, kh, kw, stride, dgrad, start}; job j owns elements [start_j, start_{j+1}).
constexpr int PACK_ROW = 10;

__global__ void pack_multi_kernel(const long long* __restrict__ table, int n, long long total) {
  __shared__ long long T[ITG_PACK_MAX_JOBS * PACK_ROW];
  for (int i = threadIdx.x; i < n * PACK_ROW; i += blockDim.x) T[i] = table[i];
  __syncthreads();
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                       // last job with start <= i
      int mid = (lo + hi + 1) >> 1;
      if (T[mid * PACK_ROW + 9] <= i) lo = mid; else hi = mid - 1;
    }
    const long long* b = T + lo * PACK_ROW;
    const float* w = reinterpret_cast<const float*>(b[0]);
    float* out = reinterpret_cast<float*>(b[1]);
    const int co = (int)b[2], ci = (int)b[3], ld = (int)b[4], kh = (int)b[5], kw = (int)b[6], stride = (int)b[7];
    const long long e = i - b[9];
    float v = 0.f;
    if (!b[8]) {
      const int Kpad = round_up_d(kh * kw * ld, BK);
      int k = (int)(e % Kpad), o = (int)(e / Kpad);
      int tap = k / ld, c = k - tap * ld;
      if (o < co && tap < kh * kw && c < ci) {
        int y = tap / kw, x = tap - y * kw;
        v = w[(((size_t)o * ci + c) * kh + y) * kw + x];
      }
    } else {
      const int skh = kh / stride, skw = kw / stride;
      const int Kpad = round_up_d(skh * skw * ld, BK);
      const int ci_pad = round_up_d(ci, 16);
      int k = (int)(e % Kpad);
      long long r = e / Kpad;
      int c_in = (int)(r % ci_pad), cls = (int)(r / ci_pad);
      int ry = cls / stride, rx = cls - ry * stride;
      int tap = k / ld, o = k - tap * ld;
      if (c_in < ci && o < co && tap < skh * skw) {
        int jy = tap / skw, jx = tap - jy * skw;
        int ay = (ry + 1) % stride, ax = (rx + 1) % stride;     // pad = 1 for stride-2 convs (ABI)
        int y = ay + stride * (skh - 1 - jy), x = ax + stride * (skw - 1 - jx);
        v = w[(((size_t)o * ci + c_in) * kh + y) * kw + x];
      }
    }
    out[e] = v;
  }
}

// ------------------------------------------------------------------------------- wgrad (TN)
struct WgP {
  GridT x, dy;
  float* slab;       // [splits][co_pad][Kpad]
  float* dbslab;     // [splits][co_pad] bias-gradient partials, or null
  int ntaps, kw, cin_ld, Kpad, Ktot;
  int MT, MU, M;     // output-pixel domain of the conv
  int stride, pad, pad_h, pad_mode;
  int co_rows, ncol_tiles, nco_tiles;
  int chunks_per_split, nchunks;
  unsigned x_bytes, dy_bytes;
};

constexpr int BKP = 16;  // pixels per pipeline stage (fp32 operands; 32 with bf16 operands)

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

template <int BCOL, int BCO, int WCOL, int WCO, bool BF, int DEPTH>
__global__ __launch_bounds__(256, tn_min_blocks(BCOL, WCOL, WCO, BF, DEPTH)) void conv_tn_kernel(WgP p, int otp) {
  constexpr int FI = WCOL / 16, FJ = WCO / 16;
  constexpr int WAVES_COL = BCOL / WCOL;
  static_assert(WAVES_COL * (BCO / WCO) == 4, "4 waves per workgroup");
  constexpr int KP = BF ? 32 : BKP;                  // pixels per stage
  constexpr int LDX = BCOL + 16, LDY = BCO + 16;     // fp32 tiles: row pitch in floats
  constexpr int LHX = BCOL + 8, LHY = BCO + 8;       // bf16 tiles: row pitch in halfwords (8-B aligned rows)
  constexpr int XG = BCOL / 4, YG = BCO / 4;         // float4 groups per pixel row
  constexpr int XL = (KP * XG + 255) / 256, YL = (KP * YG + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * BKP * (LDX + LDY)];
  static_assert(2 * 32 * (LHX + LHY) * 2 <= 2 * BKP * (LDX + LDY) * 4, "bf16 tiles fit the fp32 allocation");
  float* Xs = smem;
  float* Ys = smem + 2 * BKP * LDX;
  unsigned short* Xh = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Yh = Xh + 2 * KP * LHX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col_tile = blockIdx.x % p.ncol_tiles;
  const int co_tile = blockIdx.x / p.ncol_tiles;
  const int col0 = col_tile * BCOL, co0 = co_tile * BCO;
  const int wcol0 = (wave % WAVES_COL) * WCOL, wco0 = (wave / WAVES_COL) * WCO;
  const int split = blockIdx.z;
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
  const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x.p, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy.p, 0, p.dy_bytes, 0x00020000);
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
  auto produce = [&](int buf) {
    const bool live = pn < p.x.n;
#pragma unroll
    for (int i = 0; i < PE; ++i) {
      if (!pact[i]) continue;
      unsigned o;
      if (pdy[i]) {
        o = live ? (unsigned)grid_off(p.dy, pn, pt_, pu) * 4u : p.dy_bytes;
      } else {
        int iy = pt_ * p.stride - p.pad_h + pky[i], ix = pu * p.stride - p.pad + pkx[i];
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
  auto load_tiles = [&](int slot, f32x4 (&rx)[XL], f32x4 (&ry)[YL]) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      unsigned o = xok[i] ? otab[(slot * KP + xr[i]) * OTP + xj[i]] + xcb[i] : p.x_bytes;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxr, o, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
      unsigned o = yok[i] ? otab[(slot * KP + yr[i]) * OTP + nt] + ycb[i] : p.dy_bytes;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ryr, o, 0, 0));
    }
  };
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
  const bool do_db = p.dbslab != nullptr && col_tile == 0;
  auto store_tiles = [&](int buf, const f32x4 (&rx)[XL], const f32x4 (&ry)[YL]) {
#pragma unroll
    for (int i = 0; i < XL; ++i)
      if (xr[i] < KP) {
        if constexpr (BF) *reinterpret_cast<uint2*>(Xh + (buf * KP + xr[i]) * LHX + xcol[i]) = pack_bf16x4(rx[i]);
        else *reinterpret_cast<f32x4*>(Xs + (buf * KP + xr[i]) * LDX + xcol[i]) = rx[i];
      }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
      if (yr[i] < KP) {
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
      load_tiles(0, rxs[0], rys[0]);
      store_tiles(0, rxs[0], rys[0]);
      __syncthreads();
      for (int kk = 0; kk < nk; ++kk) {
        const int buf = kk & 1;
        if (kk + 1 < nk) load_tiles(buf ^ 1, rxs[0], rys[0]);   // table of stage kk + 1: written one barrier ago
        if (kk + 2 < nk) produce(buf);                           // stage kk + 2 -> the slot stage kk's loads have finished with
        compute(buf);
        if (kk + 1 < nk) store_tiles(buf ^ 1, rxs[0], rys[0]);
        __syncthreads();
      }
    } else {
      // two stages in flight: register set A holds stage kk + 2 while set B (stage kk + 1) drains into LDS; the offset
      // table runs three stages ahead in a ring of three slots (slot of stage s = s % 3)
      produce(0);
      if (nk > 1) produce(1);
      if (nk > 2) produce(2);
      __syncthreads();
      load_tiles(0, rxs[0], rys[0]);
      if (nk > 1) load_tiles(1, rxs[DEPTH - 1], rys[DEPTH - 1]);
      store_tiles(0, rxs[0], rys[0]);
      __syncthreads();
      int s0 = 0;                                                // kk % 3
      for (int kk = 0; kk < nk; kk += 2) {
        const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        if (kk + 2 < nk) load_tiles(s2, rxs[0], rys[0]);
        if (kk + 3 < nk) produce(s0);                            // stage kk + 3
        compute(0);
        if (kk + 1 < nk) store_tiles(1, rxs[DEPTH - 1], rys[DEPTH - 1]);
        __syncthreads();
        if (kk + 1 >= nk) break;
        if (kk + 3 < nk) load_tiles(s0, rxs[DEPTH - 1], rys[DEPTH - 1]);
        if (kk + 4 < nk) produce(s1);                            // stage kk + 4
        compute(1);
        if (kk + 2 < nk) store_tiles(0, rxs[0], rys[0]);
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
      p.dbslab[(size_t)split * p.co_rows + co0 + tid] = sdb;
    }
  }
  // D[row = column index (4 consecutive per lane)][col = co]
  float* slab = p.slab + (size_t)split * p.co_rows * p.Kpad;
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
__global__ __launch_bounds__(256, 2) void wgrad_tile_kernel(const WgP p, int tiles_x, int tiles_y, int ntiles, int cpt) {
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
    for (int i = 0; i < NLD; ++i)
      if (e_r[i] >= 0) *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = rt[i];
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
__global__ __launch_bounds__(256, 2) void wgrad_thin_kernel(const WgP p, int tiles_x, int tiles_y, int ntiles, int gpp) {
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
  auto load_tile = [&](int tile, f32x4 (&rt)[NLD], f32x4& ry) {
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
  auto store_tile = [&](const f32x4 (&rt)[NLD], const f32x4& ry) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (e_r[i] >= 0) *reinterpret_cast<f32x4*>(Xt + e_lds[i]) = rt[i];
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
  if (tile < ntiles) load_tile(tile, rtA, ryA);
  if (tile + step < ntiles) load_tile(tile + step, rtB, ryB);
  for (; tile < ntiles; tile += 2 * step) {
    store_tile(rtA, ryA);
    __syncthreads();
    if (tile + 2 * step < ntiles) load_tile(tile + 2 * step, rtA, ryA);
    contract();
    __syncthreads();
    if (tile + step >= ntiles) break;
    store_tile(rtB, ryB);
    __syncthreads();
    if (tile + 3 * step < ntiles) load_tile(tile + 3 * step, rtB, ryB);
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

struct TileWgPlan { int ok, mf, nld, cpt, tiles_x, tiles_y, blocks, thin, gpp; int64_t ntiles; size_t lds; };

TileWgPlan plan_wgrad_tile(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g) {
  TileWgPlan t;
  t.ok = 0;
  static const int enable = env_int("ITG_WGRAD_TILE", 1);
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  if (!enable || g->kh != 3 || g->kw != 3 || g->stride != 1 || g->pad != 1 || ph != 1) return t;
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
  static const int thin_en = env_int("ITG_WGRAD_THIN", 1);
  t.thin = thin_en && dy->ld == 4 && x->ld <= 16;         // <= 36 (tap, c4) groups: three passes of 12 or 16
  t.gpp = 16;
  if (t.thin) { t.cpt = 16; t.gpp = x->ld == 16 ? 12 : 16; }   // 16 channels: three taps per pass, bank-conflict free at pitch 16
  t.tiles_x = (W + TT_W - 1) / TT_W; t.tiles_y = (H + TT_H - 1) / TT_H;
  t.ntiles = (int64_t)dy->n * t.tiles_x * t.tiles_y;
  size_t fl = (size_t)TT_PIX * t.cpt + (size_t)TT_H * TT_W * 16;
  size_t red = (size_t)9 * 32 * 16 + 4 + 256 * 4;               // reduction buffer + bias partials reuse the tiles' space
  if (red > fl) fl = red;
  t.lds = fl * sizeof(float);
  if (t.lds > 64 * 1024 || t.ntiles > 0x7fffffff) return t;
  // one persistent workgroup per CU: alone the kernel is 13 % faster with two, but it runs beside the input-gradient chain
  // of the same backward pass and two would crowd that out of LDS (step: 780 vs 774 crops/s)
  static const int wtile_cu = env_int("ITG_WTILE_CU", 1);
  static const int wthin_cu = env_int("ITG_WTHIN_CU", 1);
  int per_cu = (int)((160 * 1024) / t.lds);
  if (per_cu > (t.thin ? wthin_cu : wtile_cu)) per_cu = t.thin ? wthin_cu : wtile_cu;
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
                     (int)t.ntiles, t.cpt);
}

void launch_wgrad_thin(const WgP& p, const TileWgPlan& t, hipStream_t s) {
  snprintf(g_last_launch, sizeof(g_last_launch), "wgrad_thin_kernel<3, 6>");
  hipLaunchKernelGGL((wgrad_thin_kernel<3, 6>), dim3((unsigned)t.blocks), dim3(256), t.lds, s, p, t.tiles_x, t.tiles_y,
                     (int)t.ntiles, t.gpp);
}

// dW[co][ci][ky][kx] (+)= sum_z slab[z][co][(ky*kw+kx)*ci_ld + ci]
// One workgroup per (o, 64-channel chunk): slab reads are coalesced along ci, the (ci, tap) tile is
// transposed through LDS so that the OIHW store is one contiguous run of 64*taps floats.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           float* __restrict__ db, const float* __restrict__ dbslab,
                                                           int dbsplits, int splits, int co, int ci, int ci_ld, int kh,
                                                           int kw, int co_rows, int Kpad, int accumulate) {
  __shared__ float tile[64 * 49];
  const int taps = kh * kw;
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
    tile[c * taps + t] = s;
  }
  __syncthreads();
  float* dst = dw + ((size_t)o * ci + c0) * taps;
  for (int idx = threadIdx.x; idx < cn * taps; idx += 256) dst[idx] = (accumulate & ITG_ACC_DW) ? dst[idx] + tile[idx] : tile[idx];
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
  dim3 grid((unsigned)(p.ncol_tiles * p.nco_tiles), 1, (unsigned)splits);
  // offset-table pitch: the taps one column tile can touch (+ the dY slot)
  int taps_tile = (BCOL + p.cin_ld - 1) / p.cin_ld + 1;
  if (taps_tile > p.ntaps) taps_tile = p.ntaps;
  const int otp = taps_tile + 1;
  // long per-workgroup pixel loops run best with the occupancy of the single-prefetch variant (4 waves per SIMD), short
  // ones with two stages in flight (measured on D's 256->512 layer vs its 64->128 / 128->256 layers)
  static const int depth_env = env_int("ITG_TN_DEPTH", 0);
  const int depth = depth_env ? depth_env : (p.chunks_per_split >= 128 ? 1 : 2);
  const int kp = prec == ITG_PREC_BF16 ? 32 : BKP;
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_tn_kernel<%d, %d, %d, %d, %s, %d>", BCOL, BCO, WCOL, WCO,
           prec == ITG_PREC_BF16 ? "true" : "false", prec == ITG_PREC_BF16 ? 1 : depth);
  if (prec == ITG_PREC_BF16)
    hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, true, 1>), grid, dim3(256), (size_t)2 * kp * otp * 4, s, p, otp);
  else if (depth == 2)
    hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, false, 2>), grid, dim3(256), (size_t)3 * kp * otp * 4, s, p, otp);
  else
    hipLaunchKernelGGL((conv_tn_kernel<BCOL, BCO, WCOL, WCO, false, 1>), grid, dim3(256), (size_t)2 * kp * otp * 4, s, p, otp);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

static const int RED_GROUP = env_int("ITG_RED_GROUP", 16) < 2 ? 2 : env_int("ITG_RED_GROUP", 16);   // slabs summed per thread in either reduce stage (>= 2: it is a divisor)
struct TnPlan { int bcol, bco, splits, chunks_per_split, nchunks, co_rows, Kpad, ngroups; int64_t slab_floats, ws_floats; };

TnPlan plan_tn(int64_t M, int co_ld, int Ktot, int prec = ITG_PREC_F32) {
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
  static const int want_env = env_int("ITG_TN_BLOCKS", 0);
  const int want_blocks = want_env ? want_env : (prec == ITG_PREC_BF16 ? 512 : 768);
  int want = (want_blocks + tiles - 1) / tiles;
  int max_splits = (t.nchunks + 7) / 8;             // at least 8 chunks per split
  int splits = want < max_splits ? want : max_splits;
  if (splits < 1) splits = 1;
  t.chunks_per_split = (t.nchunks + splits - 1) / splits;
  t.splits = (t.nchunks + t.chunks_per_split - 1) / t.chunks_per_split;
  t.slab_floats = (int64_t)t.splits * t.co_rows * t.Kpad;
  t.ngroups = t.splits > RED_GROUP ? (t.splits + RED_GROUP - 1) / RED_GROUP : 0;
  t.ws_floats = t.slab_floats + (int64_t)t.ngroups * t.co_rows * t.Kpad;
  return t;
}


// ------------------------------------------------------------------------------- single-output-channel convs
// The discriminator's logit layer (512 -> 1, 4x4, stride 1) has ONE output channel: the implicit GEMM
// above would spend 15 of its 16 MFMA rows on zeros (and the weight-gradient 15 of 16 columns).  Here the
// 16 filter taps take the place of the 16 rows:
//   forward   P[i][t] = x[i] . w[t]  for every INPUT pixel i (a 1x1 conv with 16 "channels", K = cin), then
//             y[o] = b + sum_t P[o + t][t]                              (tap_gather_fwd_kernel)
//   wgrad     Q[i][t] = dy[i - t]  (tap_scatter_dy_kernel), dW[t][c] = sum_i Q[i][t] x[i][c] (1x1 wgrad),
//             transposed into the OIHW gradient; the bias gradient is the column of the tap that sees
//             every dy pixel exactly once (ky = pad_h, kx = pad).
__global__ void tap_gather_fwd_kernel(const float* __restrict__ P, int H, int W, GridT out, const float* __restrict__ bias,
                                      const float* __restrict__ scale, int kh, int kw, int pad_h, int pad_w, int act,
                                      float slope) {
  const int64_t total = (int64_t)out.n * out.H * out.W;
  const int ntaps = kh * kw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int ox = (int)(i % out.W);
    int64_t r = i / out.W;
    int oy = (int)(r % out.H);
    int n = (int)(r / out.H);
    float v = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      int iy = oy - pad_h + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        int ix = ox - pad_w + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        v += P[(((size_t)n * H + iy) * W + ix) * ntaps + ky * kw + kx];
      }
    }
    if (scale) v *= *scale;
    if (bias) v += bias[0];
    v = act_apply(v, act, slope);
    *reinterpret_cast<f32x4*>(out.p + grid_off(out, n, oy, ox)) = f32x4{v, 0.f, 0.f, 0.f};
  }
}

__global__ void tap_scatter_dy_kernel(GridT dy, float* __restrict__ Q, int H, int W, int kh, int kw, int pad_h, int pad_w) {
  const int ntaps = kh * kw;
  const int64_t total = (int64_t)dy.n * H * W * ntaps;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int t = (int)(i % ntaps);
    int64_t r = i / ntaps;
    int ix = (int)(r % W); r /= W;
    int iy = (int)(r % H);
    int n = (int)(r / H);
    int ky = t / kw, kx = t - ky * kw;
    int oy = iy + pad_h - ky, ox = ix + pad_w - kx;
    float v = 0.f;
    if ((unsigned)oy < (unsigned)dy.H && (unsigned)ox < (unsigned)dy.W) v = dy.p[grid_off(dy, n, oy, ox)];
    Q[i] = v;
  }
}

// dw[c][t] (+)= tmp[t][c]; db[0] (+)= dbtmp[tdb]
__global__ void tap_wgrad_finish_kernel(const float* __restrict__ tmp, const float* __restrict__ dbtmp, float* __restrict__ dw,
                                        float* __restrict__ db, int ci, int ntaps, int tdb, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ci * ntaps) {
    int c = i / ntaps, t = i - c * ntaps;
    float v = tmp[(size_t)t * ci + c];
    dw[i] = (accumulate & ITG_ACC_DW) ? dw[i] + v : v;
  }
  if (i == 0 && db) db[0] = (accumulate & ITG_ACC_DB) ? db[0] + dbtmp[tdb] : dbtmp[tdb];
}


// ------------------------------------------------------------------------------- few-input-channel stride-2 convs
// Input gradient of the discriminator's first layer (3 -> 64, 4x4, stride 2): the parity-class implicit
// GEMM would fill 3 of its 16 MFMA rows.  Instead Q[o][(class, c, tap')] = dy[o] . w[., c, tap] for every
// OUTPUT pixel o (a 1x1 conv with 4*c*4 <= 64 rows, K = cout), then each input pixel gathers its 4 taps.
// compact panel row r = (cls * 4 + tap') * 4 + c  <-  dgrad panel row (cls * ci_pad + c), columns tap' * co_ld ..
// (c padded to 4: the gather then reads one 16-byte vector per tap)
__global__ void thin_dgrad_panel_kernel(const float* __restrict__ wd, float* __restrict__ out, int cin, int ci_pad, int co_ld,
                                        int rows_pad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_pad * co_ld) return;
  int o = i % co_ld, r = i / co_ld;
  float v = 0.f;
  if (r < 64) {
    int c = r & 3, tap = (r >> 2) & 3, cls = r >> 4;
    if (c < cin) v = wd[((size_t)(cls * ci_pad + c) * 4 + tap) * co_ld + o];
  }
  out[i] = v;
}

__global__ void thin_dgrad_gather_kernel(const float* __restrict__ Q, int Ho, int Wo, int qld, GridT dx, GridT act_out,
                                         int act, float slope) {
  const int64_t total = (int64_t)dx.n * dx.H * dx.W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int ix = (int)(i % dx.W);
    int64_t r = i / dx.W;
    int iy = (int)(r % dx.H);
    int n = (int)(r / dx.H);
    const int ry = iy & 1, rx = ix & 1, t = iy >> 1, u = ix >> 1;
    const int by = (ry + 1 - ((ry + 1) & 1)) >> 1, bx = (rx + 1 - ((rx + 1) & 1)) >> 1;
    const int cls = ry * 2 + rx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) {
      int oy = t + by - 1 + jy;
      if ((unsigned)oy >= (unsigned)Ho) continue;
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) {
        int ox = u + bx - 1 + jx;
        if ((unsigned)ox >= (unsigned)Wo) continue;
        v += *reinterpret_cast<const f32x4*>(Q + (((size_t)n * Ho + oy) * Wo + ox) * qld + (cls * 4 + jy * 2 + jx) * 4);
      }
    }
    const int off = grid_off(dx, n, iy, ix);
    if (act_out.p) v *= act_deriv(*reinterpret_cast<const f32x4*>(act_out.p + grid_off(act_out, n, iy, ix)), act, slope);
    *reinterpret_cast<f32x4*>(dx.p + off) = v;
  }
}

inline bool thin_in_conv(const itg_tensor* dy, const itg_tensor* dx, const itg_conv_geom* g) {
  static const int enable = env_int("ITG_THIN_CONV", 1);
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  return enable && dx->c <= 4 && dx->ld == 4 && g->kh == 4 && g->kw == 4 && g->stride == 2 && g->pad == 1 && ph == 1 &&
         g->pad_mode == ITG_PAD_ZERO && (dy->ld % 16) == 0;
}

inline bool thin_out_conv(const itg_tensor* in, const itg_tensor* out, const itg_conv_geom* g) {
  static const int enable = env_int("ITG_THIN_CONV", 1);
  const int ph = g->pad_h >= 0 ? g->pad_h : g->pad;
  return enable && out->c == 1 && g->kh * g->kw == 16 && g->stride == 1 && g->pad_mode == ITG_PAD_ZERO &&
         (in->ld % 16) == 0 && ph < g->kh && g->pad < g->kw && 2 * ph <= g->kh - 1 && 2 * g->pad <= g->kw - 1;
}

// TnPlan of the halo-tile weight gradient: one slab per persistent workgroup, same reduction stages
TnPlan tn_plan_for_tiles(const TileWgPlan& tw, int co_ld, int Ktot) {
  TnPlan t;
  t.bcol = -1; t.bco = 16;
  t.co_rows = round_up(co_ld, 16);
  t.Kpad = round_up(Ktot, 16);
  t.splits = tw.blocks; t.chunks_per_split = 0; t.nchunks = 0;
  t.slab_floats = (int64_t)t.splits * t.co_rows * t.Kpad;
  t.ngroups = t.splits > RED_GROUP ? (t.splits + RED_GROUP - 1) / RED_GROUP : 0;
  t.ws_floats = t.slab_floats + (int64_t)t.ngroups * t.co_rows * t.Kpad;
  return t;
}

int conv_out_dim(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }
// vertical padding may differ from the horizontal one (row-sharded patch grids carry their halo rows
// explicitly and pad only the columns): pad_h < 0 means "same as pad"
inline int pad_v(const itg_conv_geom* g) { return g->pad_h >= 0 ? g->pad_h : g->pad; }
inline int prec_of(const itg_conv_geom* g) { return g->precision == ITG_PREC_BF16 ? ITG_PREC_BF16 : ITG_PREC_F32; }

}  // namespace

// =============================================================================== C ABI
extern "C" {

int itg_version(void) { return 100; }

const char* itg_last_conv_kernel(void) { return g_last_launch; }

int64_t itg_pack_fwd_size(int co, int ci_ld, int kh, int kw) {
  return (int64_t)round_up(co, 16) * round_up(kh * kw * ci_ld, BK);
}

int64_t itg_pack_dgrad_size(int ci, int co_ld, int kh, int kw, int stride) {
  int taps = (kh / stride) * (kw / stride);
  return (int64_t)stride * stride * round_up(ci, 16) * round_up(taps * co_ld, BK);
}

int itg_pack_fwd(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, int kh, int kw,
                 void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || ci_ld < ci || (ci_ld & 3)) return ITG_ERR_ARG;
  int co_pad = round_up(co, 16), Kpad = round_up(kh * kw * ci_ld, BK);
  int64_t total = (int64_t)co_pad * Kpad;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci, ci_ld,
                     kh, kw, co_pad, Kpad);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_pack_dgrad(const float* w, const float* scale, float* out, int co, int ci, int co_ld, int kh, int kw,
                   int stride, void* stream) {
  if (!w || !out || co <= 0 || ci <= 0 || co_ld < co || (co_ld & 3)) return ITG_ERR_ARG;
  if (stride != 1 && stride != 2) return ITG_ERR_ARG;
  if (stride == 2 && ((kh & 1) || (kw & 1))) return ITG_ERR_ARG;
  // pad enters only through parity (ry+pad)&1 for stride 2; the ABI fixes pad = 1 for stride-2 convs
  int pad = 1;
  int ci_pad = round_up(ci, 16);
  int taps = (kh / stride) * (kw / stride);
  int Kpad = round_up(taps * co_ld, BK);
  int64_t total = (int64_t)stride * stride * ci_pad * Kpad;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, scale, out, co, ci,
                     co_ld, kh, kw, stride, pad, ci_pad, Kpad);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int itg_pack_multi(const int64_t* table_dev, int n, int64_t total, void* stream) {
  if (!table_dev || n <= 0 || n > ITG_PACK_MAX_JOBS || total <= 0) return ITG_ERR_ARG;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)table_dev, n,
                     (long long)total);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

int64_t itg_conv2d_fwd_workspace(const itg_tensor* in, const itg_tensor* out, const itg_conv_geom* g) {
  if (!in || !out || !g) return 0;
  if (thin_out_conv(in, out, g)) {
    int64_t Min = grid_pixels(in);
    return Min * 16 + plan_nt(Min, 16, round_up(in->ld, BK), 1, prec_of(g)).ws_floats;
  }
  return plan_nt(grid_pixels(out), round_up(out->c, 16), round_up(g->kh * g->kw * in->ld, BK), 1, prec_of(g)).ws_floats;
}

int64_t itg_conv2d_dgrad_workspace(const itg_tensor* dy, const itg_tensor* dx, const itg_conv_geom* g) {
  if (!dy || !dx || !g) return 0;
  if (thin_in_conv(dy, dx, g)) {
    const int rows = 64;
    const int64_t Mo = grid_pixels(dy);
    return Mo * rows + (int64_t)rows * dy->ld + plan_nt(Mo, rows, round_up(dy->ld, BK), 1, prec_of(g)).ws_floats;
  }
  int co_rows = round_up(dx->c, 16);
  int64_t H = (int64_t)dx->gh * dx->ph, W = (int64_t)dx->gw * dx->pw;
  if (g->stride == 1) {
    int eh = (g->pad_mode == ITG_PAD_REPLICATE) ? 2 * pad_v(g) : 0, ew = (g->pad_mode == ITG_PAD_REPLICATE) ? 2 * g->pad : 0;
    return plan_nt((int64_t)dx->n * (H + eh) * (W + ew), co_rows, round_up(g->kh * g->kw * dy->ld, BK), 1, prec_of(g)).ws_floats;
  }
  int Kpad = round_up((g->kh / 2) * (g->kw / 2) * dy->ld, BK);
  int64_t Mmax = (int64_t)dx->n * ((H + 1) / 2) * ((W + 1) / 2);
  return plan_nt(Mmax * 4, co_rows, Kpad, 4, prec_of(g)).ws_floats;
}

int itg_conv2d_fwd(const itg_tensor* in, const float* w_packed, const float* bias, const float* out_scale,
                   const itg_tensor* residual, const itg_tensor* out, const itg_conv_geom* g, int act, float slope,
                   float* workspace, int64_t workspace_floats, void* stream) {
  int rc;
  if ((rc = check_tensor(in)) || (rc = check_tensor(out))) return rc;
  if (!w_packed || !g || g->kh <= 0 || g->kw <= 0 || g->stride <= 0 || g->pad < 0) return ITG_ERR_ARG;
  if (thin_out_conv(in, out, g) && !(residual && residual->ptr)) {
    if (g->out_stats) return ITG_ERR_ARG;       // single-output-channel layers have no BatchNorm consumer on this path
    // taps-as-rows path (see tap_gather_fwd_kernel): a 1x1 conv into P[pixel][16], then the tap gather
    const int H = in->gh * in->ph, W = in->gw * in->pw;
    if (in->n != out->n || conv_out_dim(H, g->kh, 1, pad_v(g)) != out->gh * out->ph ||
        conv_out_dim(W, g->kw, 1, g->pad) != out->gw * out->pw) return ITG_ERR_ARG;
    const int64_t Min = grid_pixels(in), pf = Min * 16;
    if (!workspace || workspace_floats < pf || pf >= ((int64_t)1 << 31)) return ITG_ERR_WORKSPACE;
    itg_tensor P = {workspace, in->n, 1, 1, H, W, 16, 16};
    itg_conv_geom g1 = {1, 1, 1, 0, ITG_PAD_ZERO, 0, prec_of(g)};
    // row 0 of the packed filter is (tap, ci)-ordered with ci_ld a multiple of 16: read as 16 rows of ci_ld
    if ((rc = itg_conv2d_fwd(in, w_packed, nullptr, nullptr, nullptr, &P, &g1, ITG_ACT_NONE, 0.f, workspace + pf,
                             workspace_floats - pf, stream))) return rc;
    GridT og = make_grid(out);
    int64_t total = (int64_t)og.n * og.H * og.W;
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(tap_gather_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, H, W,
                       og, bias, out_scale, g->kh, g->kw, pad_v(g), g->pad, act, slope);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  ConvP p;
  p.ncls = 1;
  p.prec = prec_of(g);
  p.in = make_grid(in);
  p.out = make_grid(out);
  p.res = null_grid();
  if (residual && residual->ptr) {
    if ((rc = check_tensor(residual))) return rc;
    if (!same_shape(residual, out)) return ITG_ERR_ARG;
    p.res = make_grid(residual);
  }
  if (in->n != out->n) return ITG_ERR_ARG;
  int Ho = conv_out_dim(p.in.H, g->kh, g->stride, pad_v(g)), Wo = conv_out_dim(p.in.W, g->kw, g->stride, g->pad);
  if (Ho != p.out.H || Wo != p.out.W) return ITG_ERR_ARG;
  if (g->pad_mode == ITG_PAD_REPLICATE && g->stride != 1) return ITG_ERR_ARG;
  p.w = w_packed; p.bias = bias; p.scale = out_scale; p.res_mode = 0; p.res_slope = 0.f;
  p.stats = g->out_stats;
  if (p.stats && out->ld > 512) return ITG_ERR_ARG;
  p.ntaps = g->kh * g->kw; p.kw = g->kw; p.cin_ld = in->ld;
  p.Kpad = round_up(p.ntaps * in->ld, BK);
  p.MT = Ho; p.MU = Wo;
  int64_t M = (int64_t)in->n * Ho * Wo;
  if (M >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
  p.M = (int)M;
  p.isy = p.isx = g->stride; p.ioy = -pad_v(g); p.iox = -g->pad;
  p.osy = p.osx = 1; p.ooy = p.oox = 0;
  p.pad_mode = g->pad_mode; p.out_mode = 0; p.act = act; p.slope = slope;
  p.co_rows = round_up(out->c, 16);
  return dispatch_nt(p, workspace, workspace_floats, (hipStream_t)stream);
}

int itg_conv2d_dgrad(const itg_tensor* dy, const float* w_packed_dgrad, const float* out_scale, const itg_tensor* dx,
                     const itg_tensor* act_out, int act, float slope, const itg_conv_geom* g, float* workspace,
                     int64_t workspace_floats, void* stream) {
  int rc;
  if ((rc = check_tensor(dy)) || (rc = check_tensor(dx))) return rc;
  if (!w_packed_dgrad || !g) return ITG_ERR_ARG;
  if (dy->n != dx->n) return ITG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (thin_in_conv(dy, dx, g)) {
    const int Ho = dy->gh * dy->ph, Wo = dy->gw * dy->pw, H = dx->gh * dx->ph, W = dx->gw * dx->pw;
    if (conv_out_dim(H, 4, 2, 1) != Ho || conv_out_dim(W, 4, 2, 1) != Wo) return ITG_ERR_ARG;
    const int rows = 64;     // 4 parity classes x 4 taps x 4 (padded) input channels
    const int64_t Mo = grid_pixels(dy), qf = Mo * rows, pf = (int64_t)rows * dy->ld;
    if (!workspace || workspace_floats < qf + pf || qf >= ((int64_t)1 << 31)) return ITG_ERR_WORKSPACE;
    float* Q = workspace;
    float* panel = workspace + qf;
    {
      int tot = rows * dy->ld;
      hipLaunchKernelGGL(thin_dgrad_panel_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, w_packed_dgrad, panel, dx->c,
                         round_up(dx->c, 16), dy->ld, rows);
      ITG_CHECK_LAUNCH();
    }
    itg_tensor Qt = {Q, dy->n, 1, 1, Ho, Wo, rows, rows};
    itg_conv_geom g1 = {1, 1, 1, 0, ITG_PAD_ZERO, 0, prec_of(g)};
    if ((rc = itg_conv2d_fwd(dy, panel, nullptr, out_scale, nullptr, &Qt, &g1, ITG_ACT_NONE, 0.f, workspace + qf + pf,
                             workspace_floats - qf - pf, stream))) return rc;
    GridT ao = null_grid();
    if (act_out && act_out->ptr && act != ITG_ACT_NONE) {
      if ((rc = check_tensor(act_out))) return rc;
      if (!same_shape(act_out, dx)) return ITG_ERR_ARG;
      ao = make_grid(act_out);
    }
    GridT gx = make_grid(dx);
    int64_t total = (int64_t)gx.n * gx.H * gx.W;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(thin_dgrad_gather_kernel, dim3(blocks), dim3(256), 0, s, (const float*)Q, Ho, Wo, rows, gx, ao, act,
                       slope);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  ConvP p;
  p.stats = nullptr;
  p.ncls = 1;
  p.prec = prec_of(g);
  p.in = make_grid(dy);
  p.out = make_grid(dx);
  p.res = null_grid();
  p.res_mode = 0; p.res_slope = 0.f;
  if (act_out && act_out->ptr && act != ITG_ACT_NONE) {
    // dx is the gradient w.r.t. act(.)'s output `act_out`: hand back the gradient w.r.t. its input instead
    if ((rc = check_tensor(act_out))) return rc;
    if (!same_shape(act_out, dx)) return ITG_ERR_ARG;
    p.res = make_grid(act_out); p.res_mode = act; p.res_slope = slope;
  }
  const int padh = pad_v(g);
  int Ho = conv_out_dim(p.out.H, g->kh, g->stride, padh), Wo = conv_out_dim(p.out.W, g->kw, g->stride, g->pad);
  if (Ho != p.in.H || Wo != p.in.W) return ITG_ERR_ARG;
  p.bias = nullptr; p.scale = out_scale; p.act = ITG_ACT_NONE; p.slope = 0.f;
  p.cin_ld = dy->ld;
  p.co_rows = round_up(dx->c, 16);
  p.pad_mode = ITG_PAD_ZERO;
  if (g->stride == 1) {
    p.w = w_packed_dgrad;
    p.ntaps = g->kh * g->kw; p.kw = g->kw;
    p.Kpad = round_up(p.ntaps * dy->ld, BK);
    p.isy = p.isx = 1; p.osy = p.osx = 1;
    if (g->pad_mode == ITG_PAD_REPLICATE && (g->pad > 0 || padh > 0)) {
      // padded domain, gradients of the replicated frame fold onto the edge pixels
      p.MT = p.out.H + 2 * padh; p.MU = p.out.W + 2 * g->pad;
      p.ioy = -(g->kh - 1); p.iox = -(g->kw - 1);
      p.ooy = -padh; p.oox = -g->pad;
      p.out_mode = 1;
      GridT gx = p.out;
      int64_t tot = (int64_t)gx.n * (2 * gx.W + 2 * gx.H) * (gx.ld >> 2);
      int blocks = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
      hipLaunchKernelGGL(zero_border_kernel, dim3(blocks), dim3(256), 0, s, gx);
      ITG_CHECK_LAUNCH();
    } else {
      p.MT = p.out.H; p.MU = p.out.W;
      p.ioy = -(g->kh - 1 - padh); p.iox = -(g->kw - 1 - g->pad);
      p.ooy = p.oox = 0;
      p.out_mode = 0;
    }
    int64_t M = (int64_t)dx->n * p.MT * p.MU;
    if (M >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
    p.M = (int)M;
    return dispatch_nt(p, workspace, workspace_floats, s);
  }
  if (g->stride != 2 || g->pad != 1 || padh != 1 || (g->kh & 1) || (g->kw & 1) || g->pad_mode != ITG_PAD_ZERO)
    return ITG_ERR_ARG;
  int skh = g->kh / 2, skw = g->kw / 2;
  p.ntaps = skh * skw; p.kw = skw;
  p.Kpad = round_up(p.ntaps * dy->ld, BK);
  p.isy = p.isx = 1; p.osy = p.osx = 2; p.out_mode = 0;
  int ci_pad = round_up(dx->c, 16);
  p.ncls = 4;
  int Mmax = 0;
  for (int ry = 0; ry < 2; ++ry)
    for (int rx = 0; rx < 2; ++rx) {
      const int c = ry * 2 + rx;
      int ay = (ry + g->pad) & 1, ax = (rx + g->pad) & 1;
      int by = (ry + g->pad - ay) / 2, bx = (rx + g->pad - ax) / 2;
      p.cMT[c] = (p.out.H - ry + 1) / 2; p.cMU[c] = (p.out.W - rx + 1) / 2;
      int64_t M = (int64_t)dx->n * p.cMT[c] * p.cMU[c];
      if (M >= ((int64_t)1 << 29)) return ITG_ERR_ARG;
      p.cM[c] = (int)M;
      if (p.cM[c] > Mmax) Mmax = p.cM[c];
      p.cioy[c] = by - (skh - 1); p.ciox[c] = bx - (skw - 1);
      p.cooy[c] = ry; p.coox[c] = rx;
      p.cwoff[c] = (unsigned)((size_t)c * ci_pad * p.Kpad);
    }
  if (Mmax <= 0) return ITG_OK;
  p.w = w_packed_dgrad;
  p.M = Mmax; p.MT = p.cMT[0]; p.MU = p.cMU[0];
  p.ioy = p.cioy[0]; p.iox = p.ciox[0]; p.ooy = 0; p.oox = 0;
  return dispatch_nt(p, workspace, workspace_floats, s);
}

int64_t itg_conv2d_wgrad_workspace(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g) {
  if (!x || !dy || !g) return 0;
  if (thin_out_conv(x, dy, g)) {
    int64_t Min = grid_pixels(x);
    TnPlan t = plan_tn(Min, 16, x->ld, prec_of(g));
    return Min * 16 + 16 * (int64_t)x->c + 16 + t.ws_floats + (int64_t)t.splits * t.co_rows;
  }
  int64_t M = grid_pixels(dy);
  TileWgPlan tw = plan_wgrad_tile(x, dy, g);
  TnPlan t = tw.ok ? tn_plan_for_tiles(tw, dy->ld, g->kh * g->kw * x->ld) : plan_tn(M, dy->ld, g->kh * g->kw * x->ld, prec_of(g));
  return t.ws_floats + (int64_t)t.splits * t.co_rows;
}

int itg_conv2d_wgrad(const itg_tensor* x, const itg_tensor* dy, float* dw, float* db, const itg_conv_geom* g,
                     int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
  int rc;
  if ((rc = check_tensor(x)) || (rc = check_tensor(dy))) return rc;
  if (!dw || !g || !workspace) return ITG_ERR_ARG;
  if (x->n != dy->n) return ITG_ERR_ARG;
  if (accumulate & ~(ITG_ACC_DW | ITG_ACC_DB)) return ITG_ERR_ARG;     // a bit set, not a boolean (ABI note)
  hipStream_t s = (hipStream_t)stream;
  if (thin_out_conv(x, dy, g)) {
    const int H = x->gh * x->ph, W = x->gw * x->pw;
    if (conv_out_dim(H, g->kh, 1, pad_v(g)) != dy->gh * dy->ph || conv_out_dim(W, g->kw, 1, g->pad) != dy->gw * dy->pw)
      return ITG_ERR_ARG;
    const int64_t Min = grid_pixels(x), qf = Min * 16, tf = 16 * (int64_t)x->c + 16;
    if (workspace_floats < qf + tf || qf >= ((int64_t)1 << 31)) return ITG_ERR_WORKSPACE;
    float* Q = workspace;
    float* tmp = workspace + qf;            // [16][ci] then [16] bias partial
    int blocks = (int)((qf + 255) / 256 < 8192 ? (qf + 255) / 256 : 8192);
    hipLaunchKernelGGL(tap_scatter_dy_kernel, dim3(blocks), dim3(256), 0, s, make_grid(dy), Q, H, W, g->kh, g->kw, pad_v(g),
                       g->pad);
    ITG_CHECK_LAUNCH();
    itg_tensor Qt = {Q, x->n, 1, 1, H, W, 16, 16};
    itg_conv_geom g1 = {1, 1, 1, 0, ITG_PAD_ZERO, 0, prec_of(g)};
    if ((rc = itg_conv2d_wgrad(x, &Qt, tmp, tmp + 16 * (int64_t)x->c, &g1, 0, workspace + qf + tf,
                               workspace_floats - qf - tf, stream))) return rc;
    int n = x->c * 16;
    hipLaunchKernelGGL(tap_wgrad_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)tmp,
                       (const float*)(tmp + 16 * (int64_t)x->c), dw, db, x->c, 16, pad_v(g) * g->kw + g->pad, accumulate);
    ITG_CHECK_LAUNCH();
    return ITG_OK;
  }
  WgP p;
  p.x = make_grid(x);
  p.dy = make_grid(dy);
  int Ho = conv_out_dim(p.x.H, g->kh, g->stride, pad_v(g)), Wo = conv_out_dim(p.x.W, g->kw, g->stride, g->pad);
  if (Ho != p.dy.H || Wo != p.dy.W) return ITG_ERR_ARG;
  if (g->pad_mode == ITG_PAD_REPLICATE && g->stride != 1) return ITG_ERR_ARG;
  int64_t M = (int64_t)x->n * Ho * Wo;
  if (M >= ((int64_t)1 << 31)) return ITG_ERR_ARG;
  p.ntaps = g->kh * g->kw; p.kw = g->kw; p.cin_ld = x->ld;
  p.Ktot = p.ntaps * x->ld;
  const int prec = prec_of(g);
  const TileWgPlan tw = plan_wgrad_tile(x, dy, g);
  TnPlan t = tw.ok ? tn_plan_for_tiles(tw, dy->ld, p.Ktot) : plan_tn(M, dy->ld, p.Ktot, prec);
  if (t.ws_floats + (int64_t)t.splits * t.co_rows > workspace_floats) return ITG_ERR_WORKSPACE;
  p.Kpad = t.Kpad; p.co_rows = t.co_rows;
  p.slab = workspace;
  p.dbslab = db ? workspace + t.ws_floats : nullptr;       // [splits][co_rows] after the slabs
  p.MT = Ho; p.MU = Wo; p.M = (int)M;
  p.stride = g->stride; p.pad = g->pad; p.pad_h = pad_v(g); p.pad_mode = g->pad_mode;
  p.chunks_per_split = t.chunks_per_split; p.nchunks = t.nchunks;
  {
    int64_t xb = grid_pixels(x) * x->ld * 4, yb = grid_pixels(dy) * dy->ld * 4;
    if (xb >= 0xFFFF0000LL || yb >= 0xFFFF0000LL) return ITG_ERR_ARG;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
  }
  static const int plan_debug = env_int("ITG_PLAN_DEBUG", 0);
  if (plan_debug)
    fprintf(stderr, "[tn] M=%lld co_rows=%d Kpad=%d -> bcol=%d bco=%d splits=%d ngroups=%d tile=%d\n", (long long)M, t.co_rows,
            t.Kpad, t.bcol, t.bco, t.splits, t.ngroups, tw.ok);
  if (tw.ok) {
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
  if (rc) return rc;
  const float* red_src = workspace;
  int red_n = t.splits;
  if (t.ngroups > 0) {
    int64_t e4 = (int64_t)t.co_rows * t.Kpad / 4;
    float* stage = workspace + t.slab_floats;
    int64_t tot4 = e4 * t.ngroups;
    int b2 = (int)((tot4 + 255) / 256 < 8192 ? (tot4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(slab_group_reduce_kernel, dim3(b2), dim3(256), 0, s, (const f32x4*)workspace, (f32x4*)stage, e4,
                       t.splits, RED_GROUP, t.ngroups);
    ITG_CHECK_LAUNCH();
    red_src = stage; red_n = t.ngroups;
  }
  if (g->kh * g->kw > 49) return ITG_ERR_ARG;
  int blocks = dy->c * ((x->c + 63) / 64);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, red_src, dw, db, (const float*)p.dbslab,
                     t.splits, red_n, dy->c, x->c, x->ld, g->kh, g->kw, t.co_rows, t.Kpad, accumulate);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
}

}  // extern "C"
