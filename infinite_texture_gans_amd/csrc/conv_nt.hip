// Implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 for gfx950 (MI355X): forward and input gradient.
//
//   conv_nt_kernel : C[co][pixel] = sum_k W[co][k] * P[pixel][k], k = (tap, ci) with ci innermost; both operands
//                    K-contiguous, staged global -> registers -> LDS (double buffered), fragments read with
//                    ds_read_b128 under a K permutation (lane group g holds k = 4g..4g+3, MFMA step s consumes
//                    component s of every group).
// The pixel operand is gathered in merged-image coordinates from a patch-grid NHWC tensor, so the LocalPadder halo
// (reference models/layers.py:145-173) is a neighbour-patch read and the outer replicate / zero padding (layers.py:82)
// a clamp / predicate; nothing is materialised.
#include "conv_nt_kernel.h"

namespace itgk {

thread_local char g_last_launch[96] = "";

// DEPTH = number of K stages whose global loads are in flight while one stage is computed.
// TBK = K elements per stage: 16 -> fp32 operands on v_mfma_f32_16x16x4_f32; 32 -> operands rounded to
// bf16 when they are staged into LDS (tensors stay fp32 in HBM) and contracted by ONE
// v_mfma_f32_16x16x32_bf16 per fragment pair and stage, fp32 accumulation (BASELINE config 3's path).
// Either way a tile row occupies 16 dwords of a 20-dword LDS row and lane group g reads dwords 4g..4g+3.
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
// (blockIdx.y = parity class of a multi-class launch: one second-stage launch for all of them; the kernel argument stays
// read-only - a modified copy of the struct would live in scratch memory)
// one output element group (4 channels of pixel m of class cls): slab sum in a fixed order, epilogue, store; returns the values
__device__ __forceinline__ f32x4 splitk_finish(const ConvP& p, const float* __restrict__ partial, int M, int MT, int MU, int ooy, int oox,
                                               int m, int co, int* out_off = nullptr) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  {
    const float* q = partial + (size_t)m * p.co_rows + co;
    const size_t zs = (size_t)M * p.co_rows;
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
  decode_m(m, MT, MU, n, t, u);
  int oy = t * p.osy + ooy, ox = u * p.osx + oox;
  bool border = false;
  if (p.out_mode == 1) {
    int ty = min(max(oy, 0), p.out.H - 1), tx = min(max(ox, 0), p.out.W - 1);
    border = (ty == 0) | (ty == p.out.H - 1) | (tx == 0) | (tx == p.out.W - 1);
    oy = ty; ox = tx;
  }
  const int off = grid_off(p.out, n, oy, ox);
  if (out_off) *out_off = off;
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (co + e < p.out.c) v[e] += p.bias[co + e];
  }
  if (p.res.p) {
    f32x4 r = *reinterpret_cast<const f32x4*>(p.res.p + grid_off(p.res, n, oy >> p.res_ups, ox >> p.res_ups) + co);
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

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvP p) {
  const int q4 = p.out.ld >> 2;
  const int cls = blockIdx.y;
  const int M = p.cM[cls], MT = p.cMT[cls], MU = p.cMU[cls], ooy = p.cooy[cls], oox = p.coox[cls];
  const float* const partial = p.partial + p.cpoff[cls];
  int64_t total = (int64_t)M * q4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    splitk_finish(p, partial, M, MT, MU, ooy, oox, (int)(i / q4), (int)(i % q4) * 4);
}

// ... and the consumer BatchNorm's statistics of the values as stored (p.stats).  Round 4: a workgroup OWNS 64 channels (16
// lanes x 4) of a range of pixels - blockIdx.x = (channel group, pixel range) - so the per-channel sums of a workgroup meet in
// LDS by plain stores and a fixed-order sum, and a channel receives one global fp64 atomic per PIXEL RANGE (tens) instead of one
// per workgroup of a channel-agnostic sweep (round 3: ~58 workgroups of 8 sequential elements per thread with ds_add_f64
// scatter, 20 us per launch on the generator's 4 x 4 ... 16 x 16 layers); one pixel group per thread keeps the slab loads wide.
// BNB (round 6, itg_bn_bwd_fuse): the launch is an INPUT gradient and the sums are the backward sums of the BatchNorm whose
// output the conv read: s1 += dy', s2 += dy' * xhat with dy' = v * act'(a x + b), xhat = (x - mean) * rstd, x read at the pixel the
// value lands on (a replicated frame's gradient at the border pixel it folds onto - the sums are linear in dx) - the arithmetic of
// bn_bwd_reduce_kernel (norm.hip), whose launch the caller then skips.
template <bool BNB>
__global__ __launch_bounds__(256) void splitk_epilogue_stats_kernel(const ConvP p, int ngroups, int nranges) {
  __shared__ double lst[8][256];
  const int q4 = p.out.ld >> 2;
  const int cls = blockIdx.y;
  const int M = p.cM[cls], MT = p.cMT[cls], MU = p.cMU[cls], ooy = p.cooy[cls], oox = p.coox[cls];
  const float* const partial = p.partial + p.cpoff[cls];
  const int grp = blockIdx.x % ngroups, rng = blockIdx.x / ngroups;
  const int lane16 = threadIdx.x & 15, prow = threadIdx.x >> 4;
  const int c4 = grp * 16 + lane16;
  const int per = (M + nranges - 1) / nranges;
  const int m0 = rng * per, m1 = min(M, m0 + per);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c4 < q4) {
    if constexpr (BNB) {
      const int ld = p.out.ld;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.bnb_ab + c4 * 4), b = *reinterpret_cast<const f32x4*>(p.bnb_ab + ld + c4 * 4);
      const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bnb_mr + c4 * 4), rs = *reinterpret_cast<const f32x4*>(p.bnb_mr + ld + c4 * 4);
      for (int m = m0 + prow; m < m1; m += 16) {
        int off;
        const f32x4 v = splitk_finish(p, partial, M, MT, MU, ooy, oox, m, c4 * 4, &off);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.bnb_x + off + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pre = fmaf(xv[e], a[e], b[e]);
          float d1 = 1.f;
          if (p.bnb_act == ITG_ACT_LRELU) d1 = pre > 0.f ? 1.f : p.bnb_slope;
          else if (p.bnb_act == ITG_ACT_TANH) { const float t = tanhf(pre); d1 = 1.f - t * t; }
          const float ge = v[e] * d1;
          const float xh = (xv[e] - mu[e]) * rs[e];
          s1[e] += (double)ge;
          s2[e] += (double)ge * (double)xh;
        }
      }
    } else {
      for (int m = m0 + prow; m < m1; m += 16) {
        const f32x4 v = splitk_finish(p, partial, M, MT, MU, ooy, oox, m, c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const double d = v[e]; s1[e] += d; s2[e] += d * d; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { lst[e][threadIdx.x] = s1[e]; lst[4 + e][threadIdx.x] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 128) {                 // (which, lane16, e): the 16 pixel rows of the workgroup in a fixed order
    const int which = threadIdx.x >> 6, l = (threadIdx.x >> 2) & 15, e = threadIdx.x & 3;
    const int ch = (grp * 16 + l) * 4 + e;
    if (ch < p.out.ld) {
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += lst[which * 4 + e][r * 16 + l];
      atomicAdd(&(BNB ? p.bnb_sums : p.stats)[which * p.out.ld + ch], t);
    }
  }
}

// the frames of several tensors in one launch (itg_zero_frames)
struct FrameJobs { GridT g[ITG_ZERO_FRAMES_MAX]; long long start[ITG_ZERO_FRAMES_MAX + 1]; int n; };
__global__ void zero_frames_kernel(const FrameJobs fj) {
  const long long total = fj.start[fj.n];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int j = 0;
    while (j + 1 < fj.n && i >= fj.start[j + 1]) ++j;
    const GridT& g = fj.g[j];
    const long long w = i - fj.start[j];
    const int per = 2 * g.W + 2 * (g.H - 2 > 0 ? g.H - 2 : 0);
    const int q4 = g.ld >> 2;
    const int c4 = (int)(w % q4);
    const long long r = w / q4;
    const int b = (int)(r % per), n = (int)(r / per);
    int Y, X;
    if (b < g.W) { Y = 0; X = b; }
    else if (b < 2 * g.W) { Y = g.H - 1; X = b - g.W; }
    else { int k = b - 2 * g.W; Y = 1 + (k >> 1); X = (k & 1) ? g.W - 1 : 0; }
    if (g.H == 1 && b >= g.W) continue;
    *reinterpret_cast<f32x4*>(g.p + grid_off(g, n, Y, X) + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

int launch_zero_frames(const itg_tensor* t, int n, hipStream_t s) {
  if (!t || n <= 0 || n > ITG_ZERO_FRAMES_MAX) return ITG_ERR_ARG;
  FrameJobs fj;
  fj.n = n;
  long long tot = 0;
  for (int i = 0; i < n; ++i) {
    int rc = check_tensor(&t[i]);
    if (rc) return rc;
    fj.g[i] = make_grid(&t[i]);
    fj.start[i] = tot;
    const GridT& g = fj.g[i];
    tot += (long long)g.n * (2 * g.W + 2 * (g.H - 2 > 0 ? g.H - 2 : 0)) * (g.ld >> 2);
  }
  fj.start[n] = tot;
  if (tot <= 0) return ITG_OK;
  const int blocks = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
  hipLaunchKernelGGL(zero_frames_kernel, dim3(blocks), dim3(256), 0, s, fj);
  ITG_CHECK_LAUNCH();
  return ITG_OK;
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
  constexpr double split_cost = 200.0;
  pl.bpix = 128; pl.ksplit = 1;
  double best_eff = 0.0;
  const int cands_big[4] = {256, 128, 96, 64};
  const int cand_ks[13] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16};
  // workgroups per CU below which a launch counts as under-filled (bf16 stages are short: 2, config 3 +1 %)
  const double fill_min = prec == ITG_PREC_BF16 ? 2.0 : 3.0;
  constexpr double pen96 = 1.04;
  for (int ci = 0; ci < 4; ++ci) {
    int bp = cands_big[ci];
    if (pl.bco >= 112 && bp == 256) continue;              // 128x256 / 112x256 are not instantiated
    if (bp == 96 && (pl.bco != 128 || pl.tbk != 16)) continue;   // 128x96 (fp32): 3 workgroups per CU exactly on M = 73728
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
  // Round 6 (measured with tools/probes/r6_plans.sh): un-split 32 x 64 tiles over the full K for layers with
  // more than 64 filter rows.  On the generator's 3 x 3 layers (K = 936 ... 3 744) they LOSE to the split plan - 95 vs 59 us on
  // 416 -> 416 at 4 x 4 patches, even at 208 -> 208 - because one workgroup per CU runs its K loop latency-bound and 96 operand
  // rows per 2 048 outputs ask the L2 for ~14 TB/s; where the K loop is SHORT (<= 512: the 1 x 1 shortcuts, which cannot split)
  // the wide tile leaves 36 / 72 workgroups on 256 CUs and the narrow one wins: 25.8 -> 16.5 us and 18.6 -> 13.1 us forward,
  // 15.3 -> 11.9 and 12.2 -> 10.0 us input gradient.
  if (co_rows > 64 && Kpad <= 512 && ncls <= 4) {        // (ncls > 4: the Winograd GEMMs' uniform classes keep their plan)
    const int64_t b112 = ((M + 63) / 64) * nco * ncls, b32 = ((M + 63) / 64) * ((co_rows + 31) / 32) * ncls;
    if (b112 < 256 && b32 > b112) { pl.bco = 32; pl.bpix = 64; pl.ksplit = 1; }
  }
  pl.kchunks = (nk + pl.ksplit - 1) / pl.ksplit;
  pl.ksplit = (nk + pl.kchunks - 1) / pl.kchunks;
  pl.ws_floats = pl.ksplit > 1 ? (int64_t)pl.ksplit * M * co_rows * ncls : 0;
  return pl;
}

// the statistics pass as its own launch over the finished output (paths whose epilogue does not take them)
int stats_after(const ConvP& p, double* stats, hipStream_t s) {
  itg_tensor t = {p.out.p, p.out.n, p.out.gh, p.out.gw, p.out.ph, p.out.pw, p.out.c, p.out.ld};
  return itg_bn_stats(&t, stats, s);
}

int dispatch_nt(ConvP p, float* workspace, int64_t workspace_floats, hipStream_t s, int* bnb_taken) {
  // ITG_STATS_PATHS: bit 0 halo-tile kernels, bit 1 implicit-GEMM epilogue, bit 2 split-K second stage take the consumer
  // BatchNorm's statistics themselves; a cleared bit runs the separate statistics launch over the finished output instead.
  // Default 5 since round 4: the implicit-GEMM epilogue's flush is one fp64 atomic per channel and WORKGROUP onto the same
  // 2 * ld doubles, thousands of workgroups deep on the generator's 32 x 32 ... 64 x 64 layers, and those serialise at
  // ~13 ns each (norm.hip) - with bn_stats at 4.9 TB/s the separate pass is cheaper: 1 095 (7) / 1 097 (6) / 1 101 (5) /
  // 1 101 (4) crops/s in one call.  The persistent halo-tile kernels (<= 1 024 workgroups) and the split-K second stage
  // (channel-owning workgroups) keep theirs.
  static const int stats_paths = env_int("ITG_STATS_PATHS", 5);
  double* const want_stats = p.stats;
  {
    int rc_v = ITG_OK;
    if (try_conv_valu(p, s, &rc_v)) return rc_v;
    if (try_conv_s2k4(p, s, &rc_v)) return rc_v;
    if (p.ncls == 4 && try_conv_up2_tile(p, s, &rc_v)) return rc_v;
    {
      ConvP ps = p;
      ps.stats = nullptr;                                     // strip kernels: statistics by the separate pass (conv_strip.hip)
      if (try_conv_strip(ps, s, &rc_v)) return (rc_v == ITG_OK && want_stats) ? stats_after(p, want_stats, s) : rc_v;
    }
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
  NtPlan pl = plan_nt((int64_t)p.M * (p.ucls ? p.ucls : ncls_), p.co_rows, p.Kpad, p.ucls ? p.ucls : ncls_, p.prec);
  if (p.ucls) {                      // uniform classes: thousands of workgroups already, and the slab layout has no class stride
    pl.ksplit = 1; pl.kchunks = (p.Kpad + pl.tbk - 1) / pl.tbk; pl.ws_floats = 0;
    // 49 classes x (tiles / bpix) x (co / 128) workgroups: the 64-pixel tile quantises best on 256 CUs x 4 resident
    // workgroups (forward GEMM of the generated batch 206 -> 186 us, the real batch's 74 -> 70; 96 / 128 are slower)
    if (pl.bco == 128) pl.bpix = 64;
  }
  if (pl.ws_floats > workspace_floats || (pl.ws_floats && !workspace)) return ITG_ERR_WORKSPACE;
  p.ksplit = pl.ksplit; p.kchunks = pl.kchunks; p.partial = workspace;
  const bool second_stage = pl.ksplit > 1;      // (an in-launch combine - agent-scope release / ticket / acquire - measured slower: DESIGN section 3)
  const bool stats_in_stage2 = second_stage && p.stats && (stats_paths & 4) && p.out_mode == 0 && p.out.ld <= 512;
  // ... or, for an input gradient, the backward sums of the BatchNorm in front of the conv (itg_bn_bwd_fuse; also with the frame fold)
  const bool bnb_in_stage2 = second_stage && p.bnb_x && p.bnb_sums && !p.stats && p.out.ld <= 512 && p.act == ITG_ACT_NONE && !p.bias;
  if (second_stage || !(stats_paths & 2)) p.stats = nullptr;
  for (int c = 0; c < 4; ++c) p.cpoff[c] = (unsigned)((size_t)c * pl.ksplit * p.M * p.co_rows);
  {
    int64_t ib = (int64_t)p.in.n * p.in.gh * p.in.gw * p.in.ph * p.in.pw * p.in.ld * 4;
    int64_t wb = (int64_t)p.co_rows * p.Kpad * 4;
    if (ib >= 0xFFFF0000LL || wb >= 0xFFFF0000LL) return ITG_ERR_ARG;   // 32-bit buffer offsets
    p.in_bytes = (unsigned)ib; p.w_bytes = (unsigned)wb;
  }
  const int k = pl.tbk;
  static const int plan_debug = env_int("ITG_DEBUG", 0) & 1;
  if (plan_debug)
    fprintf(stderr, "[nt] M=%d x%d co_rows=%d Kpad=%d -> bco=%d bpix=%d ksplit=%d kchunks=%d\n", p.M, ncls_, p.co_rows, p.Kpad,
            pl.bco, pl.bpix, pl.ksplit, pl.kchunks);
  // uniform-class launches (the Winograd GEMMs) with fp32 operands accumulate in blocks of 16 (conv_nt_w64.hip), the block sums
  // in fp64 (NT_W64) or in a second fp32 accumulator (NT_W32), as the launch asks (ConvP.u_acc): the FORWARD GEMMs do - their
  // rounding decides which LeakyReLU inputs change sign, and every flip moves all upstream gradients by ~1e-3 -
  // F(4 x 4, 4 x 4) in fp64, F(4 x 4, 2 x 2) (a milder output transform) in fp32; the input-gradient GEMMs do not (their 6.6e-6
  // against 1.6e-6 enters the gradients linearly: invisible next to the flips; 23 us per launch saved).
  // ITG_WINO_ACC64: 1 (default) as asked, 0 = one chain everywhere, 2 = fp64 everywhere, 3 / 4 = fp64 / fp32 block sums for every launch that asks
  static const int w64 = env_int("ITG_WINO_ACC64", 1);
  int accm = (p.ucls && k == 16) ? p.u_acc : 0;
  if (p.ucls && k == 16) accm = w64 == 0 ? 0 : w64 == 2 ? 2 : (w64 == 3 && accm) ? 2 : (w64 == 4 && accm) ? 1 : accm;
  int rc = accm == 2 ? launch_nt_w64(pl.bco, pl.bpix, p, k, s) : accm == 1 ? launch_nt_w32(pl.bco, pl.bpix, p, k, s)
                                                                          : launch_nt_shape<NT_PLAIN>(pl.bco, pl.bpix, p, k, s);
  if (rc) return rc;
  if (!second_stage) return (want_stats && !p.stats) ? stats_after(p, want_stats, s) : ITG_OK;
  {
    const int ncls = p.ncls > 1 ? p.ncls : 1;
    int mmax = 0;
    for (int c = 0; c < ncls; ++c) mmax = p.cM[c] > mmax ? p.cM[c] : mmax;
    int64_t total = (int64_t)mmax * (p.out.ld >> 2);
    int per = 8192 / ncls;
    int blocks = (int)((total + 255) / 256 < per ? (total + 255) / 256 : per);
    if (blocks > 0 && (stats_in_stage2 || bnb_in_stage2)) {
      // (channel group of 64, pixel range) workgroups: enough ranges for ~2 workgroups per CU, at least 16 pixels each
      const int q4 = p.out.ld >> 2;
      const int ngroups = (q4 + 15) / 16;
      int nranges = (512 / ncls + ngroups - 1) / ngroups;
      if (nranges > (mmax + 15) / 16) nranges = (mmax + 15) / 16;
      if (nranges < 1) nranges = 1;
      p.stats = want_stats;
      if (bnb_in_stage2) {
        hipLaunchKernelGGL(splitk_epilogue_stats_kernel<true>, dim3(ngroups * nranges, ncls), dim3(256), 0, s, p, ngroups, nranges);
        if (bnb_taken) *bnb_taken = 1;
      } else {
        hipLaunchKernelGGL(splitk_epilogue_stats_kernel<false>, dim3(ngroups * nranges, ncls), dim3(256), 0, s, p, ngroups, nranges);
      }
      ITG_CHECK_LAUNCH();
    } else if (blocks > 0) {
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks, ncls), dim3(256), 0, s, p);
      ITG_CHECK_LAUNCH();
    }
  }
  return (want_stats && !p.stats) ? stats_after(p, want_stats, s) : ITG_OK;
}


}  // namespace itgk
