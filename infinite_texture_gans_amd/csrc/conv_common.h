// Shared declarations of the convolution translation units (conv_nt.hip: implicit-GEMM forward / input gradient,
// conv_tile.hip: halo-tile kernels of the narrow 3x3 layers, conv_wgrad.hip: weight gradients, conv.hip: packing,
// taps-as-rows paths and the C ABI).  Internal: not part of include/itg.h.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "itg_common.h"

extern "C" int itg_bn_stats(const itg_tensor* x, double* sums, void* stream);

namespace itgk {

constexpr int BK = 16;    // K elements per pipeline stage
constexpr int LDK = 20;   // LDS row pitch (floats): BK + 4 keeps rows 16-B aligned

// name of the GEMM kernel instantiation launched by this thread's last conv call, exactly as a profiler prints it
extern thread_local char g_last_launch[96];

// ITG_KERNEL_MASK: the specialised kernels, one bit each (default all on; a cleared bit sends the layer to the generic kernel -
// how tools compare paths): 0 conv_s2k4, 1 conv_cin1, 2 conv_valu, 3 wgrad_thin, 4 conv_tile, 5 wgrad_tile, 6 thin (taps-as-rows)
// convs, 7 tiled attention, 8 flat weight-gradient producer, 9 conv_strip, 10 conv_up2_tile, 11 wgrad_up2_tile
enum { KM_S2K4 = 0, KM_CIN1, KM_VALU, KM_WGRAD_THIN, KM_CONV_TILE, KM_WGRAD_TILE, KM_THIN_CONV, KM_ATT_TILED, KM_TN_FLAT, KM_STRIP,
       KM_UP2_TILE, KM_UP2_WTILE };
inline int env_int(const char* name, int dflt);
inline bool kernel_on(int bit) {
  static const int mask = env_int("ITG_KERNEL_MASK", 0xFFF);
  return (mask >> bit) & 1;
}
inline int env_int(const char* name, int dflt) {
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
  int res_ups;                  // 1: the residual tensor has half the output's patch extent and is read through a nearest x2 upsample
  int co_rows, nco_tiles;
  unsigned in_bytes, w_bytes;   // buffer-resource extents of the pixel operand / packed weights
  int use_tab;                  // per-row tap-offset table in LDS (narrow layers)
  int xcd_remap;                // deal contiguous runs of tiles to each XCD (its L2 then sees 1/8 of the pixel tiles)
  double* stats;                // fwd only, or null: [2][out.ld] per-channel sum / sum of squares of the stored output
  int prec;                     // ITG_PREC_F32 | ITG_PREC_BF16
  float* partial;      // split-K slabs [ksplit][M][co_rows] (ksplit > 1)
  int ksplit, kchunks; // K chunks (of BK) per split
  // uniform classes (ucls > 0; the 49 GEMMs of a Winograd-domain convolution, conv_wino.hip): ucls independent problems of
  // the geometry in slot 0 in ONE grid, class c on the input at p.in.p + c * u_in, the panel at p.w + c * u_w, the output at
  // p.out.p + c * u_out (floats); no split-K.  The class is the SLOWEST index of the tile id (classes share nothing).
  int ucls;
  unsigned u_in, u_w, u_out;
  int u_acc;           // accumulation the uniform-class launch asks for: 0 one fp32 chain, 1 blocks of 16 summed in a second
                       // fp32 accumulator (NT_W32), 2 blocks summed in fp64 (NT_W64); ITG_WINO_ACC64 overrides (conv_nt.hip)
  // stride-2 input-gradient: the 4 output-parity classes run as ONE grid (blockIdx.y = class)
  int ncls;
  int cMT[4], cMU[4], cM[4], cioy[4], ciox[4], cooy[4], coox[4];
  unsigned cwoff[4];   // float offset of the class's packed sub-kernel
  unsigned cpoff[4];   // float offset of the class's split-K slabs
  // input gradients only (itg_bn_bwd_fuse, round 6): the split-K second stage also accumulates the backward sums of the BatchNorm
  // whose output the conv read - bnb_x = that BatchNorm's input (p.out's shape), sums into bnb_sums[2][out.ld]
  const float* bnb_x = nullptr;
  const float* bnb_ab = nullptr;
  const float* bnb_mr = nullptr;
  double* bnb_sums = nullptr;
  int bnb_act = 0;
  float bnb_slope = 0.f;
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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint2 pack_bf16x4(f32x4 v) {
  bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  return __builtin_bit_cast(uint2, h);
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__device__ __forceinline__ int round_up_d(int x, int m) { return (x + m - 1) / m * m; }

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

// ---- plans (host)
struct NtPlan { int bco, bpix, tbk, ksplit, kchunks; int64_t ws_floats; };
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
  // itg_conv_geom.up2 (conv_tn_kernel only): x is the half-size source tensor and blockIdx.y = output-parity class (ry, rx):
  // pixel (t, u) of the SOURCE domain pairs dY(2t + ry, 2u + rx) with the 2 x 2 taps at (t + ry - 1, u + rx - 1); slab and
  // bias partial of (split, class) sit at index split * 4 + class
  int up2;
  // uniform classes (conv_tn_kernel only; conv_wino.hip's weight gradient): blockIdx.y = class, whose operands sit u_x / u_dy
  // floats behind x / dy (x_bytes / dy_bytes are the bytes of ONE class) and whose slab is slab[split][class]
  int ucls;
  unsigned u_x, u_dy;
};

constexpr int BKP = 16;  // pixels per pipeline stage (fp32 operands; 32 with bf16 operands)
struct TileWgPlan { int ok, mf, nld, cpt, tiles_x, tiles_y, blocks, thin, gpp, coef_off; int64_t ntiles; size_t lds; };
struct TnPlan { int bcol, bco, splits, chunks_per_split, nchunks, co_rows, Kpad, ngroups; int64_t slab_floats, ws_floats; };

inline int red_group() { return 16; }       // slabs summed per thread in either weight-gradient reduce stage (64: neutral)

// conv_tile.hip
int try_conv_valu(const ConvP& p, hipStream_t s, int* rc);
int try_conv_tile(ConvP& p, hipStream_t s, int* rc);
int try_conv_s2k4(const ConvP& p, hipStream_t s, int* rc);
int try_conv_up2_tile(const ConvP& p, hipStream_t s, int* rc);
// conv_strip.hip
int try_conv_strip(const ConvP& p, hipStream_t s, int* rc);
// conv_wino.hip: Winograd F(4 x 4, R x R) for wide stride-1 layers (R = 4: the discriminator's 256 -> 512 layer; R = 3: the
// generator's wide blocks)
int64_t wino_workspace_floats(const itg_tensor* in, const itg_tensor* out, int R, int fold);
int wino_conv(const itg_tensor* in, const float* u_panel, const float* bias, const float* out_scale, const itg_tensor* res, int res_ups,
              int res_mode, float res_slope, const itg_tensor* out, int R, int pad, int pad_mode, int fold, int act, float slope,
              int prec, float* workspace, int64_t workspace_floats, hipStream_t s, bool input_gradient = false);
int64_t wino_s2_workspace_floats(const itg_tensor* in, const itg_tensor* out);
int wino_conv_s2(const itg_tensor* in, const float* u_panel, const float* bias, const float* out_scale, const itg_tensor* out, int act,
                 float slope, int prec, float* workspace, int64_t workspace_floats, hipStream_t s);
int64_t wino_s2_dgrad_workspace_floats(const itg_tensor* dy, const itg_tensor* dx);
int wino_conv_s2_dgrad(const itg_tensor* dy, const float* ut_panel, const float* out_scale, const itg_tensor* dx, const itg_tensor* act_out,
                       int act, float slope, int prec, float* workspace, int64_t workspace_floats, hipStream_t s);
// conv_nt.hip
NtPlan plan_nt(int64_t M_total, int co_rows, int Kpad, int ncls = 1, int prec = ITG_PREC_F32);
int dispatch_nt(ConvP p, float* workspace, int64_t workspace_floats, hipStream_t s, int* bnb_taken = nullptr);
int launch_zero_border(const GridT& g, hipStream_t s);
int launch_zero_frames(const itg_tensor* t, int n, hipStream_t s);
// conv_nt_w64.hip
int launch_nt_w64(int bco, int bpix, const ConvP& p, int k, hipStream_t s);
int launch_nt_w32(int bco, int bpix, const ConvP& p, int k, hipStream_t s);
// conv_wgrad.hip
TileWgPlan plan_wgrad_tile(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g);
TnPlan plan_tn(int64_t M, int co_ld, int Ktot, int prec = ITG_PREC_F32, int ncls = 1);
TnPlan tn_plan_for_tiles(const TileWgPlan& tw, int co_ld, int Ktot);
TileWgPlan plan_wgrad_up2_tile(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g);
TnPlan tn_plan_for_up2_tiles(const TileWgPlan& tw, int co_ld, int cin_ld);
int run_wgrad(WgP& p, const TnPlan& t, const TileWgPlan& tw, int prec, const itg_tensor* x, const itg_tensor* dy,
              const itg_conv_geom* g, float* dw, float* db, int accumulate, float* workspace, hipStream_t s);
int run_wgrad_slabs(WgP& p, const TnPlan& t, const TileWgPlan& tw, int prec, hipStream_t s);
int launch_wgrad_reduce(const float* slab, int splits, const float* dbslab, int dbsplits, float* dw, float* db, int co, int ci,
                        int ci_ld, int kh, int kw, int co_rows, int Kpad, int accumulate, hipStream_t s);
// conv_wino.hip: the Winograd weight gradient of the same layers -> ONE slab in the generic layout [co_rows][16 * ci_ld] plus
// bias partials, finished by the generic reduce stage (single or multi-layer)
struct WinoWgPlan { int R, Ty, Tx, Rr, co_rows, Kpad; int64_t tiles, v_off, m_off, u_off, slab_off, db_off, ws_floats; TnPlan tn; };
WinoWgPlan plan_wino_wgrad(const itg_tensor* x, const itg_tensor* dy, int R);
int wino_wgrad_slabs(const itg_tensor* x, const itg_tensor* dy, int pad, int pad_mode, int prec, const WinoWgPlan& w, float* workspace,
                     bool want_db, hipStream_t s, const float* v_fwd = nullptr);
int launch_reduce_multi(const itg_wgrad_job* jobs, int n, hipStream_t s);

}  // namespace itgk
