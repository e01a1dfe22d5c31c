// Strip kernels of the narrow stride-1 3x3 layers at high resolution (the generator's last blocks: 13 / 26 channels on
// 64^2 ... 128^2 patches; forward and input gradient) for gfx950.
//
// The halo-tile kernels (conv_tile.hip) stage an (8+2) x (32+2) pixel tile in LDS per 144 ... 576 MFMAs and pay two workgroup
// barriers, ~6 gather-address computations per thread and an epilogue of per-fragment patch-grid arithmetic for it: 3 - 5.6
// VALU instructions per MFMA, MFMA pipe 40 - 46 % busy.  Here NOTHING of the activation tensor goes through LDS and the main
// loop has no barrier:
//   * a WAVE owns a strip of 32 pixel columns (inside one patch: patch widths are multiples of 32) and walks down a segment
//     of rows.  Lane (nl = lane & 15, g = lane >> 4) loads the 4 channels 4g .. 4g+3 of pixel nl of each 16-pixel half - one
//     b128 buffer load per half, row and 16-channel chunk, 1 KB contiguous per load for 16-float pixels - which IS the B
//     operand layout of v_mfma_f32_16x16x4_f32 under the K permutation of conv_nt_kernel (lane group g holds k = 4g .. 4g+3);
//     one more load per row fetches the two pixels left and right of the strip into lanes 0 / 15 (the LocalPadder halo of
//     reference models/layers.py:145-173 is a neighbour-patch address, the replicate frame a clamped one);
//   * the horizontal taps are LANE SHIFTS of that row: DPP row_shr:1 / row_shl:1 with the neighbour half's edge pixel
//     rotated in (row_ror), 6 DPP moves per register and row instead of two more LDS reads per tap;
//   * the vertical taps are three ACCUMULATOR sets: input row Y feeds output rows Y + 1, Y, Y - 1 (ky = 0, 1, 2), row Y - 1
//     is complete after it and leaves through the epilogue (1 / sigma, bias, residual or activation derivative, activation,
//     ) as 1 KB contiguous stores; rows Y + 1 .. Y + 2 are in flight in two more register sets;
//   * the filter bank sits in LDS once per workgroup (read-only afterwards): 9 x chunks x row tiles A fragments per row;
//   * row addresses are scalar (the row's byte offset is the buffer load's soffset), lane offsets are fixed per strip.
// Input gradients of replicate-padded layers (itg_conv2d_dgrad's padded-extent + atomic fold, conv.hip) are folded IN
// REGISTER instead: the frame's gradient is a linear function of dy, so output row 0 takes the ky = 2 products of dy row 0 as
// well (row -1's only term), column 0 adds dy column 0 to its kx = 2 operand, and likewise at the far edges - no atomics, no
// zeroed frame, plain stores.
#include "conv_common.h"

namespace itgk {

struct StripP {
  int nstrips;      // 32-pixel strips per image row
  int nseg, L;      // row segments per image, rows per segment
  int units;        // n * nseg * nstrips
  int lpw;          // log2 of the patch width (a power of two >= 32); patch heights: powers of two, or any height on 1 x 1 grids
  int fold_v, fold_h;   // input gradient of a replicate-padded layer: frame rows / columns folded in register
  int rep_v;        // the rows above / below the image are the clamped edge rows (replicate padding with implicit halo rows)
  int rsh;          // output row t reads input rows t + ky + rsh - 1: 0 with one implicit padding row (the standard layer), 1 for a
                    // row-sharded band whose input carries its halo rows (pad_h = 0), -1 for that band's input gradient
  int H, Hin, W;    // output rows, input rows, width of the merged image
  int res_half;     // residual has half the patch extent (read through a nearest x2 upsample)
};

// pixel index of the first pixel of merged row Y of image n (column 0 of patch column 0)
__device__ __forceinline__ int strip_rowpix(const GridT& g, int lpw, int n, int Y) {
  int R = 0, y = Y;
  if (g.gh > 1) { const int lph = 31 - __clz(g.ph); R = Y >> lph; y = Y & (g.ph - 1); }      // (power-of-two patch height)
  return (((n * g.gh + R) * g.gw) * g.ph + y) << lpw;
}
// ... plus this for merged column X
__device__ __forceinline__ int strip_colpix(const GridT& g, int lpw, int X) {
  return (X >> lpw) * (g.ph << lpw) + (X & (g.pw - 1));
}

template <int CTRL>
__device__ __forceinline__ f32x4 dpp4(f32x4 old, f32x4 src) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {       // (through scalars: __builtin_bit_cast of a vector ELEMENT reads element 0 for every e, hipcc 7.2)
    const float o = old[e], v = src[e];
    r[e] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(o), __float_as_int(v), CTRL, 0xf, 0xf, false));
  }
  return r;
}
// every lane has a source (rotations): no old operand, hence no copy in front of the DPP move
template <int CTRL>
__device__ __forceinline__ f32x4 dppmov4(f32x4 src) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = src[e];
    r[e] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false));
  }
  return r;
}
// byte offset of a store that must be dropped: above every admissible tensor size (try_conv_strip: < 0xFFFF0000 bytes) and 16
// bytes below the 32-bit wrap, so the hardware range check discards it whatever the row
constexpr unsigned STRIP_DROP = 0xFFFFFFF0u;
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111, DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR15 = 0x12F;

// KC = 16-channel chunks of an input pixel (1: cin_ld <= 16, 2: cin_ld <= 32), FI = 16-row tiles of output channels
template <int KC, int FI>
__global__ __launch_bounds__(256, (KC * FI == 1 ? 3 : 2)) void conv_strip_kernel(const ConvP p, const StripP sp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // filter bank: [9 taps][KC][FI] blocks of 16 rows x 16 k, unpadded, the four 16-byte k groups of a row XOR-swizzled with bit 3
  // of the row (conv_nt_kernel's layout: every 16-lane group of a ds_read_b128 fragment read hits 64 distinct banks; the
  // padded pitch-20 rows of the first version were 2-way conflicted - SQ_LDS_BANK_CONFLICT 47 % of the LDS cycles)
  float* Wl = lds;
  float* biasl = lds + 9 * KC * FI * 256;                 // [32]
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index as an SGPR: everything a wave decides (its unit, rows, frame cases) is then provably uniform - scalar
  // branches and scalar row offsets instead of exec-mask regions and a waterfall loop around every buffer access (the first
  // version spent 134 VALU + 86 SALU instructions per row on those)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, g = lane >> 4;
  {
    // filter bank -> LDS: every 16-byte k group of the packed panel (k = tap * cin_ld + ci, four channels of one tap: cin_ld is
    // a multiple of 4) is one b128 load and one b128 LDS store; all of a thread's loads are in flight before its first store
    // (the first version's scalar loop serialised nine global-load latencies: 3.8 us before the first MFMA)
    constexpr int NV = 9 * KC * FI * 64, PER = (NV + 255) / 256;
    f32x4 wv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = tid + j * 256;
      const int kg = e & 3, row = (e >> 2) & 15, q = e >> 6;            // q = (tap * KC + c) * FI + i
      const int i = q % FI, c = (q / FI) % KC, t = q / (FI * KC);
      const int ci = 16 * c + 4 * kg, co = 16 * i + row;
      wv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < NV && ci < p.cin_ld && co < p.out.c) wv[j] = *reinterpret_cast<const f32x4*>(p.w + (size_t)co * p.Kpad + t * p.cin_ld + ci);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = tid + j * 256;
      const int kg = e & 3, row = (e >> 2) & 15, q = e >> 6;
      if (e < NV) *reinterpret_cast<f32x4*>(Wl + q * 256 + row * 16 + ((kg ^ (((row >> 3) & 1) << 1)) << 2)) = wv[j];
    }
  }
  if (tid < 32) biasl[tid] = (p.bias && tid < p.out.c) ? p.bias[tid] : 0.f;
  __syncthreads();
  // (the nine A fragments of the 16 -> 16 layer as 36 permanent registers - no LDS read inside the MFMA bursts - was tried:
  // 25 - 46 spilled registers inside the 168-register budget, 64 -> 72 us; the switch stays for a larger budget)
  constexpr bool AREG = false;
  const float osc = p.scale ? *p.scale : 1.f;
  const int afrag = nl * 16 + ((g ^ (((nl >> 3) & 1) << 1)) << 2);      // this lane's A fragment inside a 16 x 16 block
  f32x4 areg[AREG ? 9 : 1];
  if constexpr (AREG) {
#pragma unroll
    for (int t = 0; t < 9; ++t) areg[t] = *reinterpret_cast<const f32x4*>(Wl + t * 256 + afrag);
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in.p, 0, p.in_bytes, 0x00020000);
  const unsigned out_bytes = (unsigned)((size_t)p.out.n * p.out.gh * p.out.gw * p.out.ph * p.out.pw * p.out.ld * 4);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out.p, 0, out_bytes, 0x00020000);
  const bool has_res = p.res.p != nullptr;
  const unsigned res_bytes = has_res ? (unsigned)((size_t)p.res.n * p.res.gh * p.res.gw * p.res.ph * p.res.pw * p.res.ld * 4) : 0u;
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? p.res.p : p.out.p), 0, res_bytes, 0x00020000);
  const bool replicate = p.pad_mode == ITG_PAD_REPLICATE;
  const int H = sp.H, Hin = sp.Hin, W = sp.W, lpw = sp.lpw;
  const unsigned ild4 = (unsigned)p.in.ld * 4u, old4 = (unsigned)p.out.ld * 4u, rld4 = (unsigned)p.res.ld * 4u;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int unit = blockIdx.x * 4 + wave; unit < sp.units; unit += gridDim.x * 4) {
    const int xs = unit % sp.nstrips;
    const int seg = (unit / sp.nstrips) % sp.nseg;
    const int n = unit / (sp.nstrips * sp.nseg);
    const int X0 = xs * 32, y0 = seg * sp.L, y1 = min(H, y0 + sp.L);
    // ---- lane byte offsets of this strip (the row's offset is added as the scalar operand of every access)
    const int cpix = strip_colpix(p.in, lpw, X0);
    const unsigned vc0 = (unsigned)(cpix + nl) * ild4 + (unsigned)g * 16u, vc1 = vc0 + 16u * ild4;
    unsigned ved = p.in_bytes;                            // lanes 0 / 15: the pixel left / right of the strip
    {
      int XE = nl == 0 ? X0 - 1 : X0 + 32;
      bool ok = nl == 0 || nl == 15;
      if (replicate) XE = min(max(XE, 0), W - 1); else ok = ok && (unsigned)XE < (unsigned)W;
      if (ok) ved = (unsigned)strip_colpix(p.in, lpw, XE) * ild4 + (unsigned)g * 16u;
    }
    unsigned vo[FI][2], vr[FI][2];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int co = 16 * i + 4 * g;
        vo[i][f] = co < p.out.ld ? (unsigned)(strip_colpix(p.out, lpw, X0) + 16 * f + nl) * old4 + (unsigned)co * 4u : STRIP_DROP;
        vr[i][f] = res_bytes;
        if (has_res && co < p.res.ld) {
          const int rp = sp.res_half ? strip_colpix(p.res, lpw - 1, X0 >> 1) + ((16 * f + nl) >> 1) : strip_colpix(p.res, lpw, X0) + 16 * f + nl;
          vr[i][f] = (unsigned)rp * rld4 + (unsigned)co * 4u;
        }
      }
    const bool foldl = sp.fold_h && X0 == 0, foldr = sp.fold_h && X0 + 32 == W;

    f32x4 c0[3][KC], c1[3][KC], ed[3][KC];                // three rows in flight: row Yv lives in set (Yv - y0 + 1) % 3
    auto load_row = [&](int Yv, f32x4 (&a0)[KC], f32x4 (&a1)[KC], f32x4 (&ae)[KC]) {
      int Yc = Yv + sp.rsh;                               // the input row behind virtual row Yv
      bool ok = Yv <= y1;
      if (sp.rep_v) Yc = min(max(Yc, 0), Hin - 1); else ok = ok && (unsigned)Yc < (unsigned)Hin;
      if (!ok) return;                                    // wave-uniform: a row outside a zero-padded image is never used
      const unsigned rb = (unsigned)strip_rowpix(p.in, lpw, n, Yc) * ild4;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        a0[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vc0 + 64u * c, rb, 0));
        a1[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vc1 + 64u * c, rb, 0));
        ae[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ved + 64u * c, rb, 0));
      }
    };
    f32x4 accA[FI][2], accB[FI][2], accC[FI][2];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int f = 0; f < 2; ++f) { accA[i][f] = zero4; accB[i][f] = zero4; accC[i][f] = zero4; }

    // products of one filter row ky with the operands of one input row: acc[i][f] += W(ky, kx) B_kx[f]
    auto mfma_ky = [&](f32x4 (&acc)[FI][2], int ky, const f32x4 (&bl)[2][KC], const f32x4 (&bc0)[KC], const f32x4 (&bc1)[KC],
                       const f32x4 (&br)[2][KC]) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          f32x4 a[FI];
#pragma unroll
          for (int i = 0; i < FI; ++i)
            a[i] = AREG ? areg[AREG ? ky * 3 + kx : 0] : *reinterpret_cast<const f32x4*>(Wl + (((ky * 3 + kx) * KC + c) * FI + i) * 256 + afrag);
          const f32x4 b0 = kx == 0 ? bl[0][c] : (kx == 1 ? bc0[c] : br[0][c]);
          const f32x4 b1 = kx == 0 ? bl[1][c] : (kx == 1 ? bc1[c] : br[1][c]);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < FI; ++i) {
              acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b0[s], acc[i][0], 0, 0, 0);
              acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b1[s], acc[i][1], 0, 0, 0);
            }
        }
    };
    // an interior row feeds all three output rows: per (kx, chunk) the three filter rows' A fragments are fetched together and
    // their 24 x FI MFMAs run on six independent accumulators with the B operands shared (one exposed LDS latency per group,
    // not per 8 MFMAs)
    auto mfma_row3 = [&](f32x4 (&A0)[FI][2], f32x4 (&A1)[FI][2], f32x4 (&A2)[FI][2], const f32x4 (&bl)[2][KC], const f32x4 (&bc0)[KC],
                         const f32x4 (&bc1)[KC], const f32x4 (&br)[2][KC]) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          f32x4 a[3][FI];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int i = 0; i < FI; ++i)
              a[ky][i] = AREG ? areg[AREG ? ky * 3 + kx : 0] : *reinterpret_cast<const f32x4*>(Wl + (((ky * 3 + kx) * KC + c) * FI + i) * 256 + afrag);
          const f32x4 b0 = kx == 0 ? bl[0][c] : (kx == 1 ? bc0[c] : br[0][c]);
          const f32x4 b1 = kx == 0 ? bl[1][c] : (kx == 1 ? bc1[c] : br[1][c]);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < FI; ++i) {
              A0[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][i][s], b0[s], A0[i][0], 0, 0, 0);
              A0[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][i][s], b1[s], A0[i][1], 0, 0, 0);
              A1[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][i][s], b0[s], A1[i][0], 0, 0, 0);
              A1[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][i][s], b1[s], A1[i][1], 0, 0, 0);
              A2[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2][i][s], b0[s], A2[i][0], 0, 0, 0);
              A2[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2][i][s], b1[s], A2[i][1], 0, 0, 0);
            }
        }
    };
    // finished output row t: epilogue + store
    auto finish = [&](f32x4 (&acc)[FI][2], int t, const f32x4 (&rv)[FI][2]) {
      const unsigned ob = (unsigned)strip_rowpix(p.out, lpw, n, t) * old4;
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(biasl + 16 * i + 4 * g);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          f32x4 v = acc[i][f] * osc + bv;
          // wave-uniform mode tests around whole vectors (per-element act_apply / act_deriv calls compiled into a scalar
          // branch chain per ELEMENT: ~1 200 instructions of epilogue per row against 72 MFMAs)
          if (has_res) {
            const f32x4 r = rv[i][f];
            if (p.res_mode == 0) v += r;
            else if (p.res_mode == ITG_ACT_LRELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= r[e] > 0.f ? 1.f : p.res_slope;
            } else if (p.res_mode == ITG_ACT_TANH) v *= 1.f - r * r;
          }
          if (p.act == ITG_ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
          } else if (p.act == ITG_ACT_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
          }
          // The row offset goes into the VECTOR offset here, not the scalar operand as for the loads: a buffer_store_dwordx4 whose
          // data registers a VALU instruction overwrites in the next cycle needs a wait state, and LLVM's hazard recognizer only
          // inserts it for stores WITHOUT an SGPR offset (the GFX9 rule).  On gfx950 the SGPR-offset form is hit as well: with the
          // accumulator zeroed right behind the store (`buffer_store_dwordx4 v[20:23], ..., s78 offen; v_mov_b32 v23, 0`) the
          // generator's 128^2-patch layers stored zeros in ~6 % of their pixels once every wave ran several units.
          // A lane whose channel group lies past out.ld drops its store through the range check: its offset stays the sentinel
          // (adding the row offset to it would wrap back INTO a tensor above 2 GiB and zero 16 bytes of another pixel).
          const unsigned so = vo[i][f] == STRIP_DROP ? STRIP_DROP : vo[i][f] + ob;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rout, so, 0, 0);
          acc[i][f] = zero4;
        }
      }
    };
    // one input row: A0 takes ky = 0 (output row Yv + 1), A1 ky = 1 (row Yv), A2 ky = 2 (row Yv - 1, complete afterwards)
    auto step = [&](int Yv, f32x4 (&s0)[KC], f32x4 (&s1)[KC], f32x4 (&se)[KC], f32x4 (&A0)[FI][2], f32x4 (&A1)[FI][2],
                    f32x4 (&A2)[FI][2]) {
      const int t2 = Yv - 1;
      const bool done2 = t2 >= y0 && t2 < y1;
      f32x4 rv[FI][2];
      if (has_res && done2) {
        const unsigned rb = (unsigned)(sp.res_half ? strip_rowpix(p.res, lpw - 1, n, t2 >> 1) : strip_rowpix(p.res, lpw, n, t2)) * rld4;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int f = 0; f < 2; ++f) rv[i][f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, vr[i][f], rb, 0));
      }
      const bool rowok = sp.rep_v || (unsigned)(Yv + sp.rsh) < (unsigned)Hin;
      if (rowok) {
        f32x4 bl[2][KC], br[2][KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          bl[0][c] = dpp4<DPP_ROW_SHR1>(se[c], s0[c]);                                   // lane 0: the pixel left of the strip
          bl[1][c] = dpp4<DPP_ROW_SHR1>(dppmov4<DPP_ROW_ROR1>(s0[c]), s1[c]);        // lane 0: pixel 15 of the left half
          br[1][c] = dpp4<DPP_ROW_SHL1>(se[c], s1[c]);                                   // lane 15: the pixel right of the strip
          br[0][c] = dpp4<DPP_ROW_SHL1>(dppmov4<DPP_ROW_ROR15>(s1[c]), s0[c]);       // lane 15: pixel 0 of the right half
          if (foldl) {         // column -1's gradient (its only term: kx = 2 on dy column 0) lands on column 0
#pragma unroll
            for (int e = 0; e < 4; ++e) br[0][c][e] += nl == 0 ? s0[c][e] : 0.f;
          }
          if (foldr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bl[1][c][e] += nl == 15 ? s1[c][e] : 0.f;
          }
        }
        const int t0 = Yv + 1, t1 = Yv;
        const bool d0 = t0 >= y0 && t0 < y1, d1 = t1 >= y0 && t1 < y1;
        if (d0 && d1 && done2) {
          mfma_row3(A0, A1, A2, bl, s0, s1, br);
        } else {                                                                         // first / last rows of a segment
          if (d0) mfma_ky(A0, 0, bl, s0, s1, br);
          else if (sp.fold_v && Yv == H - 1 && d1) mfma_ky(A1, 0, bl, s0, s1, br);         // row H's gradient folds onto row H - 1
          if (d1) mfma_ky(A1, 1, bl, s0, s1, br);
          if (done2) mfma_ky(A2, 2, bl, s0, s1, br);
          else if (sp.fold_v && Yv == 0 && d1) mfma_ky(A1, 2, bl, s0, s1, br);             // row -1's gradient folds onto row 0
        }
      }
      if (done2) finish(A2, t2, rv);
      load_row(Yv + 3, s0, s1, se);
    };

    load_row(y0 - 1, c0[0], c1[0], ed[0]);
    load_row(y0, c0[1], c1[1], ed[1]);
    load_row(y0 + 1, c0[2], c1[2], ed[2]);
    for (int Yv = y0 - 1; Yv <= y1; Yv += 3) {
      step(Yv, c0[0], c1[0], ed[0], accA, accC, accB);          // targets: rows Yv + 1 (A), Yv (C), Yv - 1 (B)
      if (Yv + 1 > y1) break;
      step(Yv + 1, c0[1], c1[1], ed[1], accB, accA, accC);
      if (Yv + 2 > y1) break;
      step(Yv + 2, c0[2], c1[2], ed[2], accC, accB, accA);
    }
  }
}

static inline int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// eligibility + launch; returns 1 when it handled the call
int try_conv_strip(const ConvP& p, hipStream_t s, int* rc) {
  const int enable = kernel_on(KM_STRIP);
  if (!enable || p.ncls > 1 || p.ucls || p.ntaps != 9 || p.kw != 3 || p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1) return 0;
  if (p.prec != ITG_PREC_F32 || p.cin_ld > 32 || p.cin_ld < 8 || p.co_rows > 32) return 0;      // (4-float pixels: a chunk would be 3/4 padding - the halo-tile kernel packs four taps into one)
  const GridT& gi = p.in;
  const GridT& go = p.out;
  if (gi.n != go.n || gi.gh != go.gh || gi.gw != go.gw || gi.pw != go.pw) return 0;
  const int lpw = ilog2_exact(gi.pw);
  if (lpw < 5) return 0;                                      // 32-pixel strips stay inside a patch
  if (gi.gh > 1 && (ilog2_exact(gi.ph) < 1 || gi.ph != go.ph)) return 0;      // (a half-size residual needs ph >= 2)
  int fold_v = 0, fold_h = 0, rsh = 0, rep_v = 0;
  if (gi.ph == go.ph) {
    if (p.out_mode == 0) {
      if (p.ioy != -1 || p.iox != -1 || p.ooy != 0 || p.oox != 0 || p.MT != gi.H || p.MU != gi.W) return 0;
      rep_v = p.pad_mode == ITG_PAD_REPLICATE;
    } else {
      // the padded-extent input gradient of a replicate-padded layer (itg_conv2d_dgrad): same result, folded in register
      if (p.ioy != -2 || p.iox != -2 || p.ooy != -1 || p.oox != -1 || p.MT != go.H + 2 || p.MU != go.W + 2 || p.pad_mode != ITG_PAD_ZERO) return 0;
      fold_v = fold_h = 1;
    }
  } else if (gi.gh == 1 && gi.ph == go.ph + 2) {
    // a row-sharded band whose input carries its halo rows (itg_conv_geom.pad_h = 0): every input row exists
    if (p.out_mode != 0 || p.ioy != 0 || p.iox != -1 || p.ooy != 0 || p.oox != 0 || p.MT != go.H || p.MU != go.W) return 0;
    rsh = 1;
  } else if (gi.gh == 1 && gi.ph + 2 == go.ph) {
    // ... and its input gradient: dx has the halo rows (their gradients travel back to the neighbours), dy rows outside the band
    // are zero; the left / right frame of a replicate-padded layer is folded as above
    if (p.ioy != -2 || p.ooy != 0 || p.MT != go.H || p.pad_mode != ITG_PAD_ZERO) return 0;
    if (p.out_mode == 0) { if (p.iox != -1 || p.oox != 0 || p.MU != go.W) return 0; }
    else { if (p.iox != -2 || p.oox != -1 || p.MU != go.W + 2) return 0; fold_h = 1; }
    rsh = -1;
  } else {
    return 0;
  }
  if ((int64_t)go.H * go.W < 64 * 64) return 0;
  int res_half = 0;
  if (p.res.p) {
    const GridT& gr = p.res;
    if (gr.n != go.n || gr.gh != go.gh || gr.gw != go.gw || gr.ld != go.ld) return 0;
    if (p.res_ups) { if (2 * gr.ph != go.ph || 2 * gr.pw != go.pw || (gr.gh > 1 && ilog2_exact(gr.ph) < 0)) return 0; res_half = 1; }
    else if (gr.ph != go.ph || gr.pw != go.pw) return 0;
  }
  const int64_t ib = (int64_t)gi.n * gi.gh * gi.gw * gi.ph * gi.pw * gi.ld * 4, ob = (int64_t)go.n * go.gh * go.gw * go.ph * go.pw * go.ld * 4;
  if (ib >= 0xFFFF0000LL || ob >= 0xFFFF0000LL) return 0;
  const int KC = p.cin_ld <= 16 ? 1 : 2, FI = p.co_rows / 16;
  if (p.stats) return 0;      // the consumer BatchNorm's statistics: the caller's separate pass (dispatch_nt; in-kernel sums + 768 workgroups' closing
                              // fp64 atomics measured 0.3 - 1.6 % slower per step than bn_stats at 4.9 TB/s)
  ConvP q = p;
  q.in_bytes = (unsigned)ib;
  StripP sp;
  sp.nstrips = go.W / 32; sp.lpw = lpw; sp.fold_v = fold_v; sp.fold_h = fold_h; sp.rep_v = rep_v; sp.rsh = rsh;
  sp.H = go.H; sp.Hin = gi.H; sp.W = go.W; sp.res_half = res_half;
  // rows per segment: every wave slot of the chip (1 024 SIMDs x resident waves) gets one unit if the image allows it; a
  // segment re-loads two halo rows, so short segments cost loads (not MFMAs)
  const int occ = KC * FI == 1 ? 3 : 2;
  const int64_t slots = 1024LL * occ;
  int bestL = go.H;
  double best = 1e30;
  for (int L = 4; L <= go.H && L <= 128; ++L) {
    const int nseg = (go.H + L - 1) / L;
    const int64_t units = (int64_t)gi.n * nseg * sp.nstrips;
    const int64_t rounds = (units + slots - 1) / slots;
    const double cost = (double)rounds * (L + 0.35 * 2 + 1.0);      // rows of MFMA work per slot + halo loads + prologue
    if (cost < best) { best = cost; bestL = L; }
  }
  sp.L = bestL;
  sp.nseg = (go.H + sp.L - 1) / sp.L;
  const int64_t units = (int64_t)gi.n * sp.nseg * sp.nstrips;
  if (units > 0x7fffffff) return 0;
  sp.units = (int)units;
  size_t lds = ((size_t)9 * KC * FI * 256 + 32) * sizeof(float);
  // every CU gets the SAME number of workgroups: the kernel is MFMA-bound with all waves resident from the start, so a CU that
  // was handed four workgroups while its neighbour got two finishes a third later than the even deal (the dispatcher fills by
  // resources, not evenly: 2.5 resident waves per SIMD measured where 3 were launched).  LDS is what caps a CU at `occ`.
  constexpr int even = 1;
  if (even) { const size_t cap = (size_t)(160 * 1024) / (occ + 1) + 512; if (lds < cap) lds = cap; }
  int64_t blocks = (units + 3) / 4;
  const int64_t maxb = 256LL * occ;
  if (blocks > maxb) blocks = maxb;
  const void* kern = nullptr;
#define ITG_STRIP_PICK(KC_, FI_) kern = (const void*)&conv_strip_kernel<KC_, FI_>
  if (KC == 1 && FI == 1) { ITG_STRIP_PICK(1, 1); }
  else if (KC == 1) { ITG_STRIP_PICK(1, 2); }
  else if (FI == 1) { ITG_STRIP_PICK(2, 1); }
  else { ITG_STRIP_PICK(2, 2); }
#undef ITG_STRIP_PICK
  snprintf(g_last_launch, sizeof(g_last_launch), "conv_strip_kernel<%d, %d>", KC, FI);
  if (lds > 48 * 1024) {
    static const void* attr_set[8] = {nullptr};
    bool seen = false;
    int slot = 0;
    for (; slot < 8 && attr_set[slot]; ++slot) seen = seen || attr_set[slot] == kern;
    if (!seen && slot < 8) {
      if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) { (void)hipGetLastError(); return 0; }
      attr_set[slot] = kern;
    }
  }
  void* args[] = {(void*)&q, (void*)&sp};
  (void)hipLaunchKernel(kern, dim3((unsigned)blocks), dim3(256), args, lds, s);
  *rc = hipGetLastError() == hipSuccess ? ITG_OK : ITG_ERR_LAUNCH;
  return 1;
}

}  // namespace itgk
