/* itg.h - C ABI of libitg_hip.so, the MI355X (gfx950) kernels behind the
 * patch-by-patch texture-GAN train step / patch-grid inference.
 *
 * The reference (ai4netzero/Infinite_Texture_GANs) has no native layer: its hot
 * path is torch ops called from Python.  Each entry point
 * below names the reference call site(s) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller, fp32 unless noted,
 *     16-byte aligned; nothing is allocated inside the library (workspaces are
 *     passed in); all work is enqueued asynchronously on `stream` (a
 *     hipStream_t passed as void*); no global state; re-entrant per stream.
 *   - return value: 0 on success, negative itg_status on a rejected call (bad
 *     shape, misaligned pointer, launch failure).  Nothing throws.
 *   - activations are "patch-grid NHWC" tensors (itg_tensor): n images, each a
 *     gh x gw grid of ph x pw patches, channels innermost, `ld` floats per pixel
 *     (ld >= c, ld % 4 == 0, channels c..ld-1 are ZERO).  Memory order is
 *     [n][gr][gc][y][x][ld]; a plain NHWC image is gh = gw = 1.  The reference's
 *     patch batch (N*gh*gw, C, P, P) with p = n*gh*gw + r*gw + c (utils.py:604-607)
 *     is exactly this order, so "merged image" coordinates (Y, X) address patch
 *     (Y / ph, X / pw) at (Y % ph, X % pw) with no copy.
 *   - weights stay in the reference's OIHW layout at the boundary (state_dict
 *     compatible); packed forms are produced by itg_pack_* into caller buffers.
 */
#ifndef ITG_H
#define ITG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef enum {
  ITG_OK = 0,
  ITG_ERR_ARG = -1,      /* inconsistent shapes / unsupported geometry */
  ITG_ERR_ALIGN = -2,    /* pointer or ld not 16-byte / 4-float aligned */
  ITG_ERR_LAUNCH = -3,   /* hip launch error */
  ITG_ERR_WORKSPACE = -4 /* workspace too small */
} itg_status;

typedef struct {
  void* ptr;
  int32_t n, gh, gw, ph, pw, c, ld;
} itg_tensor;

enum { ITG_PAD_ZERO = 0, ITG_PAD_REPLICATE = 1 };
enum { ITG_ACT_NONE = 0, ITG_ACT_LRELU = 1, ITG_ACT_TANH = 2 };

/* Round 6: the generator's backward chain is latency-bound: conv input gradient (+ split-K second stage) -> itg_bn_bwd_reduce ->
 * itg_bn_bwd_apply -> next input gradient (reference models/layers.py:301-322 run backwards).  When the input gradient of the conv
 * that consumed y = act(BatchNorm(x)) runs with a split-K second stage, that stage already sweeps dx once: it then also
 * accumulates the BatchNorm's backward sums  sums[c] += dy'  and  sums[ld + c] += dy' * xhat  with  dy' = dx * act'(a x + b),
 * xhat = (x - mean) * rstd  - exactly what itg_bn_bwd_reduce would compute from (x, dx) in a launch of its own (a replicate-padded
 * layer's frame gradients count with the border pixel they fold onto: the sums are linear in dx).  `taken` tells the caller
 * whether it happened; if not (another kernel family ran the layer, ld > 512) the caller launches itg_bn_bwd_reduce as before. */
typedef struct itg_bn_bwd_fuse {
  const itg_tensor* x;       /* the BatchNorm's input: dx's shape */
  const float* ab;           /* 2 * ld floats: y = act(a x + b), as itg_bn_finalize(_apply) left them */
  const float* mean_rstd;    /* 2 * ld floats */
  int32_t act;               /* ITG_ACT_* of the BatchNorm's fused activation */
  float slope;
  double* sums;              /* 2 * ld doubles, zeroed by the caller, accumulated into */
  int32_t taken;             /* OUT, written on the host before the call returns: 1 = the sums are (being) accumulated on `stream` */
  int32_t reserved;          /* 0 */
} itg_bn_bwd_fuse;

typedef struct {
  int32_t kh, kw, stride, pad;
  int32_t pad_mode; /* ITG_PAD_*: how reads outside the merged image resolve */
  int32_t pad_h;    /* vertical padding when it differs from `pad` (row-sharded grids: 0); < 0 = same as pad */
  int32_t precision; /* ITG_PREC_*: MFMA operand type of the contraction (tensors are fp32 in memory either way) */
  int32_t up2;       /* 0 | 1: the conv runs on the nearest x2 upsample of the tensor it is handed (`in` / `dx` / `x` have HALF
                      * the conv input's patch extent) and the upsample is folded into the filter: every output-parity class
                      * (Y % 2, X % 2) sees only 2 x 2 distinct source pixels, so conv3x3(up2x(x)) is four 2 x 2 convolutions of x
                      * with phase-summed weights - 4/9 of the multiply-adds, a quarter of the input bytes, nothing materialised
                      * (reference models/generators.py:52 + layers.py:301-311: nn.Upsample in front of every block's first
                      * conv2d_lp).  3 x 3, stride 1, pad 1 only; panels from itg_pack_up2_fwd / itg_pack_up2_dgrad; the input
                      * gradient is the 4 x 4 stride-2 convolution of dy with the same phase sums and lands on the half-size
                      * tensor (the upsample's backward included).                                                           */
  double* out_stats; /* itg_conv2d_fwd only, or NULL: 2 * out.ld doubles (sum | sum of squares per channel over every
                      * output pixel), ACCUMULATED into by the conv's epilogue - the BatchNorm statistics of the
                      * layer that consumes this output (nn.BatchNorm2d after every generator conv, reference
                      * models/layers.py:279-280,301-322) without a second pass over the tensor (out.ld <= 512) */
  struct itg_bn_bwd_fuse* bn_bwd; /* itg_conv2d_dgrad only, or NULL (every other entry point requires NULL): the BatchNorm whose OUTPUT
                               * this conv read - see itg_bn_bwd_fuse below.  (Rounds 3-4 kept an input transform in this slot:
                               * BatchNorm-apply inside the conv's tile loader, measured slower and removed.) */
  int32_t flags;     /* ITG_GEOM_*; zero-initialise the struct */
  int32_t reserved;  /* 0 */
  const float* wino_v; /* itg_conv2d_wgrad / itg_conv2d_wgrad_slabs of an ITG_GEOM_WINO layer only, or NULL: the transformed
                      * input V = B^T d B of the SAME x, as itg_conv2d_fwd left it in the first 49 (36) * tiles * x.ld floats of
                      * its workspace - the weight gradient then skips its own input transform (the caller keeps the forward's
                      * workspace alive until the backward pass; round 4) */
} itg_conv_geom;
/* itg_conv2d_dgrad with ITG_PAD_REPLICATE folds the gradients of the replicated frame onto the 1-pixel border of dx
 * with atomics and zeroes that border first (one small launch per call).  ITG_GEOM_FRAME_ZEROED: the caller has zeroed
 * it already - itg_zero_frames does so for up to ITG_ZERO_FRAMES_MAX tensors in ONE launch (the step engine keeps the
 * dx buffers of the generator's convs and clears all their frames at the start of the backward pass). */
#define ITG_GEOM_FRAME_ZEROED 1
/* ITG_GEOM_WINO: a wide stride-1 pad-1 conv (ld multiples of 16) runs as Winograd F(4 x 4, R x R):
 *   kh = kw = 4, zero padding: 49 multiplications per 4 x 4 output tile instead of 256 (the discriminator's 256 -> 512 layer,
 *     reference models/discriminators.py:196-206); panels itg_pack_wino_fwd / itg_pack_wino_dgrad;
 *   kh = kw = 3, zero or replicate padding, any patch grid: 36 instead of 144 (the generator's wide blocks, reference
 *     models/layers.py:25-34,301-311); panels itg_pack_wino3_fwd / itg_pack_wino3_dgrad.
 * w_packed must be that panel (itg_conv2d_fwd: the fwd one, itg_conv2d_dgrad: the dgrad one); the *_workspace queries size
 * the transformed-domain buffers; residual (also half-extent), bias, out_scale, activation, act_out and out_stats (as a pass
 * of its own over the output) are honoured; itg_conv2d_wgrad / _wgrad_slabs take the transformed route as well (fp32 operands).
 * fp32 error of R = 4: ~5e-6 rel-L2 against 3e-7 of the direct form (tools/gen_winograd.py). */
#define ITG_GEOM_WINO 2
int64_t itg_pack_wino_size(int rows, int k_ld);    /* 49 * round_up(rows, 16) * round_up(k_ld, 16) floats */
int64_t itg_pack_wino3_size(int rows, int k_ld);   /* 36 * ... */
/* fwd panel U[xi][co_pad][ci_ld'] = G g G^T of g = w[co][ci]; dgrad panel U'[xi][ci_pad][co_ld'] of the flipped filter */
int itg_pack_wino_fwd(const float* w_oihw, const float* scale, float* out, int co, int ci, int ci_ld, void* stream);
int itg_pack_wino_dgrad(const float* w_oihw, const float* scale, float* out, int co, int ci, int co_ld, void* stream);
/* F(4 x 4, 2 x 2) panel of a 4 x 4 STRIDE-2 pad-1 layer (the discriminator's 64 -> 128 / 128 -> 256 layers, reference
 * models/discriminators.py:190-195) for itg_conv2d_fwd with ITG_GEOM_WINO: U[25][co_pad][4 * ci_ld], the four parity classes
 * of the filter concatenated along K.  Forward only: the layer's itg_conv2d_dgrad / _wgrad take the ordinary panels / no flag. */
int64_t itg_pack_wino_s2_size(int co, int ci_ld);
int itg_pack_wino_s2_fwd(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, void* stream);
/* ... and the transposed panel UT[25][4 * ci_ld (padded to 16)][co_ld] for itg_conv2d_dgrad with ITG_GEOM_WINO on that layer: the
 * input gradient as the adjoint of the forward pipeline (dM = A dY A^T, 25 GEMMs, gathered B dV B^T)                        */
int64_t itg_pack_wino_s2_dgrad_size(int ci_ld, int co_ld);
int itg_pack_wino_s2_dgrad(const float* w, const float* scale, float* out, int co, int ci, int ci_ld, int co_ld, void* stream);
int itg_pack_wino3_fwd(const float* w_oihw, const float* scale, float* out, int co, int ci, int ci_ld, void* stream);
int itg_pack_wino3_dgrad(const float* w_oihw, const float* scale, float* out, int co, int ci, int co_ld, void* stream);
#define ITG_ZERO_FRAMES_MAX 32
int itg_zero_frames(const itg_tensor* tensors, int n, void* stream);

/* ITG_PREC_F32: v_mfma_f32_16x16x4_f32, the reference's arithmetic (BASELINE configs 1, 2, 4, 5).
 * ITG_PREC_BF16: operands rounded to bf16 while staged into LDS, v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation (BASELINE config 3's "bf16 MFMA path"); parity tolerance 1e-2 rel-L2 per conv. */
enum { ITG_PREC_F32 = 0, ITG_PREC_BF16 = 1 };

int itg_version(void);

/* Name of the implicit-GEMM kernel instantiation the calling thread's last itg_conv2d_* call launched,
 * spelled as rocprofv3 prints it (bench.py labels its HIP-event timings with it).                    */
const char* itg_last_conv_kernel(void);

/* ---- weight packing ------------------------------------------------------------
 * OIHW master weights (nn.Conv2d.weight, reference models/layers.py:178-200) ->
 * K-contiguous packed forms.  `scale` points to one device float multiplied into
 * every element (1/sigma for spectral norm, layers.py:180), or NULL.
 *   fwd   : [co_pad][kh][kw][ci_ld]           co_pad = round_up(co, 16)
 *   dgrad : stride 1: [ci_pad][kh][kw][co_ld] taps flipped;
 *           stride 2: 4 parity classes x [ci_pad][kh/2][kw/2][co_ld]
 * Sizes in floats are returned by the *_size helpers (host side, no launch).     */
int64_t itg_pack_fwd_size(int co, int ci_ld, int kh, int kw);
int64_t itg_pack_dgrad_size(int ci, int co_ld, int kh, int kw, int stride);
int itg_pack_fwd(const float* w_oihw, const float* scale, float* out, int co, int ci, int ci_ld,
                 int kh, int kw, void* stream);
int itg_pack_dgrad(const float* w_oihw, const float* scale, float* out, int co, int ci, int co_ld,
                   int kh, int kw, int stride, void* stream);
/* Panels of a 3x3 conv behind a nearest x2 upsample (itg_conv_geom.up2), w_oihw = the layer's [co][ci][3][3] weight:
 *   fwd   : 4 output-parity classes (ry, rx) x [co_pad][2][2][ci_ld]; tap (jy, jx) of class (ry, rx) reads source pixel
 *           (y + ry - 1 + jy, x + rx - 1 + jx) and holds  sum_{i in S(ry,jy)} sum_{j in S(rx,jx)} w[i][j],
 *           S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2};
 *   dgrad : [ci_pad][4][4][co_ld], the 4x4 stride-2 pad-1 kernel whose transpose the folded forward is:
 *           tap t holds the sum over T(t), T(0) = {2}, T(1) = {1,2}, T(2) = {0,1}, T(3) = {0} per axis.            */
int64_t itg_pack_up2_fwd_size(int co, int ci_ld);
int64_t itg_pack_up2_dgrad_size(int ci, int co_ld);
int itg_pack_up2_fwd(const float* w_oihw, const float* scale, float* out, int co, int ci, int ci_ld, void* stream);
int itg_pack_up2_dgrad(const float* w_oihw, const float* scale, float* out, int co, int ci, int co_ld, void* stream);

/* Every panel of a model in ONE launch (the train step repacks a model once per optimizer step:
 * 2 launches per iteration instead of ~65).  The job table is static for a model and lives in DEVICE
 * memory (no per-launch upload, capturable in a hipGraph): n <= ITG_PACK_MAX_JOBS rows of 10 x int64
 *   { w_oihw (pointer), out (pointer), co, ci, ld, kh, kw, stride, dgrad, start }
 * ld = ci_ld for a forward panel (dgrad = 0, itg_pack_fwd layout) or co_ld for a dgrad panel (dgrad = 1,
 * itg_pack_dgrad layout); dgrad = 2 / 3: the itg_pack_up2_fwd / itg_pack_up2_dgrad panels (kh = kw = 3); 4 / 5: the
 * itg_pack_wino_fwd / itg_pack_wino_dgrad panels (kh = kw = 4); 6 / 7: the itg_pack_wino3_* panels (kh = kw = 3); row j owns flat elements [start_j, start_j + size_j) of the launch, `total` is
 * the sum of the panel sizes (itg_pack_*_size).  No scale here: the spectral-norm 1/sigma of such
 * panels is applied through `out_scale` below. */
#define ITG_PACK_MAX_JOBS 48
int itg_pack_multi(const int64_t* table_dev, int n, int64_t total, void* stream);

/* ---- convolution (implicit GEMM on v_mfma_f32_16x16x4_f32) ----------------------
 * Replaces nn.Conv2d forward/backward at reference models/layers.py:25-34 (3x3 under
 * local padding: the 1-px halo of LocalPadder, layers.py:145-173, is read straight
 * from the neighbouring patches / clamped at the image border, never materialised),
 * layers.py:196-200 (1x1 shortcut) and layers.py:190-194 (4x4 discriminator convs).
 *   out = act( conv(in, w) + bias [+ residual] )
 * in/out are patch-grid tensors; the conv runs in merged-image coordinates.
 * act: ITG_ACT_* with slope for LRELU (slope 0 = ReLU).  bias may be NULL (length
 * out.ld, zero-padded); residual (same layout as out) may have ptr == NULL.
 * out_scale: NULL or one device float multiplied into the contraction before the bias
 * (1/sigma of a spectrally normalised layer whose panel was packed unscaled).       */
/* workspace (floats) for the split-K path taken when the grid would under-fill the 256 CUs;
 * 0 when the call does not split.  The same sizes are re-derived inside the launch.            */
int64_t itg_conv2d_fwd_workspace(const itg_tensor* in, const itg_tensor* out, const itg_conv_geom* g);
int64_t itg_conv2d_dgrad_workspace(const itg_tensor* dy, const itg_tensor* dx, const itg_conv_geom* g);
int itg_conv2d_fwd(const itg_tensor* in, const float* w_packed, const float* bias, const float* out_scale,
                   const itg_tensor* residual, const itg_tensor* out, const itg_conv_geom* g,
                   int act, float slope, float* workspace, int64_t workspace_floats, void* stream);

/* dX of the same conv: `dy` has the conv's output shape, `dx` its input shape.
 * With ITG_PAD_REPLICATE the gradient of the replicated border folds back onto the
 * edge pixels (the autograd of F.pad(..., 'replicate') at layers.py:82).           */
/* act_out (may be NULL / ptr NULL) + act + slope: the conv's input was `act_out = act(u)` (the fused
 * activation of the producing layer); dx is then multiplied by act'(u), expressed through act_out as in
 * itg_act_bwd, i.e. the producing layer's activation backward is fused into this epilogue.          */
int itg_conv2d_dgrad(const itg_tensor* dy, const float* w_packed_dgrad, const float* out_scale,
                     const itg_tensor* dx, const itg_tensor* act_out, int act, float slope,
                     const itg_conv_geom* g, float* workspace, int64_t workspace_floats, void* stream);

/* dW (OIHW) and optional db (length co).  accumulate: ITG_ACC_DW adds into dw instead of overwriting it, ITG_ACC_DB
 * likewise for db (gradient sinks: the flat .grad buffer of the step engine; a spectrally normalised layer sinks its
 * bias but takes dW into a temporary for itg_spectral_norm_bwd).
 * workspace: itg_conv2d_wgrad_workspace() floats (split-K slabs + fp64 bias scratch). */
#define ITG_ACC_DW 1
#define ITG_ACC_DB 2
#define ITG_WS_ZEROED 4   /* itg_spectral_norm_bwd: the caller hands over an accumulator that is already zero */
int64_t itg_conv2d_wgrad_workspace(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g);
int itg_conv2d_wgrad(const itg_tensor* x, const itg_tensor* dy, float* dw_oihw, float* db,
                     const itg_conv_geom* g, int accumulate, float* workspace, int64_t workspace_floats,
                     void* stream);

/* Deferred form of the weight gradient (the train step's backward pass): itg_conv2d_wgrad_slabs runs only the contraction
 * of ONE layer (split over pixel ranges into fp32 slabs in `workspace`, which must stay untouched until the reduce) and
 * describes the rest of the work in *job; itg_wgrad_reduce_multi then finishes up to ITG_WGRAD_MAX_JOBS layers in ONE
 * launch: fixed-order sum of every layer's slabs, OIHW transposition, bias gradients - instead of two or three small
 * launches per layer (79 per train step before).  The caller fills job->dw / job->db / job->accumulate (ITG_ACC_* as in
 * itg_conv2d_wgrad) and, for a spectrally normalised layer, job->w_orig and job->dot (one zeroed double): dw is then a
 * temporary that receives G = dL/dW and <G, w_orig> is accumulated into *dot for itg_spectral_norm_bwd_multi.
 * Two jobs of one launch must not share dw / db.  Returns ITG_ERR_ARG for layers that cannot be deferred (the
 * single-output-channel taps-as-rows path): use itg_conv2d_wgrad for those.                                         */
#define ITG_WGRAD_MAX_JOBS 24
typedef struct {
  const float* slab;    /* [splits][co_rows][Kpad]                      (filled by itg_conv2d_wgrad_slabs) */
  const float* dbslab;  /* [dbsplits][co_rows] bias partials, or NULL   (filled) */
  float* dw;            /* OIHW target                                   (caller) */
  float* db;            /* bias-gradient target or NULL                  (caller) */
  const float* w_orig;  /* spectral norm: OIHW weight_orig, else NULL    (caller) */
  double* dot;          /* spectral norm: zeroed accumulator of <G, w_orig> (caller) */
  int32_t splits, dbsplits, co, ci, ci_ld, kh, kw, co_rows, Kpad;      /* (filled) */
  int32_t accumulate;   /* ITG_ACC_DW | ITG_ACC_DB                       (caller) */
  float* stage;         /* [ngroups][co_rows][Kpad] scratch of the group stage inside `workspace`, or NULL (filled) */
  int32_t ngroups, group;   /* layers with many slabs: slabs are first summed in groups of `group` (filled) */
} itg_wgrad_job;
int itg_conv2d_wgrad_slabs(const itg_tensor* x, const itg_tensor* dy, const itg_conv_geom* g, float* workspace,
                           int64_t workspace_floats, itg_wgrad_job* job, void* stream);
int itg_wgrad_reduce_multi(const itg_wgrad_job* jobs, int n, void* stream);

/* ---- LocalPadder as a standalone operator -----------------------------------------
 * reference models/layers.py:145-173 + utils.py:577-613,658-742 (training branch /
 * first sub-image): x (n*gh*gw patches of p x p) -> y (patches of (p+2) x (p+2)),
 * NCHW fp32 exactly as the reference module takes and returns them.  bwd is the
 * scatter-add of halo gradients (autograd of merge + F.pad + crop).
 * `merged` != 0: x is the already merged, pre-padded latent (n, c, gh*p+2, gw*p+2),
 * only cropped (generator `start`, layers.py:152-155).                              */
int itg_local_pad_fwd(const float* x, float* y, int n, int c, int gh, int gw, int p, int pad_mode,
                      int merged, void* stream);
int itg_local_pad_bwd(const float* dy, float* dx, int n, int c, int gh, int gw, int p, int pad_mode,
                      int merged, void* stream);
/* same operator on patch-grid NHWC tensors (out.ph == in.ph + 2) */
int itg_local_pad_nhwc_fwd(const itg_tensor* x, const itg_tensor* y, int pad_mode, void* stream);

/* eval-mode streaming variant (layers.py:84-143): left column / top row carried from
 * earlier sub-images; left/top may be NULL (then outer padding on that side).
 * left: (n, gh*p, ld) column; top: (n, gw*p+2, ld) row incl. its two corner pixels.
 * bottom: same shape as top - the halo row of the patch row BELOW when the patch grid is sharded
 * by rows across GPUs (the row arrives over RCCL); NULL = outer padding.                  */
int itg_local_pad_stream_fwd(const itg_tensor* x, const float* left, const float* top, const float* bottom,
                             const itg_tensor* y, int pad_mode, void* stream);

/* ---- halo rows of a row-sharded band (training; reference models/layers.py:145-173 LocalPadder.forward, with the patch
 * grid sharded by patch rows over ranks: SURVEY 8e / BASELINE config 4) -------------------------------------------------
 * ext = (n, 1, 1, H + 2, W, ld): a rank's band in image layout with ONE HALO ROW above and below every image.  Rows
 * 1 .. H are written by the producer (itg_bn_finalize_apply / itg_bn_apply with y.ph == x.ph + 2 write exactly those rows;
 * itg_band_interior_copy for producers that cannot), rows 0 / H + 1 by itg_band_halo_fill; the conv then reads ext with
 * itg_conv_geom.pad_h = 0, and its input gradient has the same layout: itg_band_halo_grad folds the halo rows' gradients
 * (their own, at a replicate border; the neighbour's, arrived over RCCL) onto rows 1 / H, and itg_bn_bwd_* with
 * dy.ph == x.ph + 2 read rows 1 .. H of it.  No concatenated copy of the band exists in either direction.
 * mode_*: 0 = the (n, W, ld) buffer `top` / `bottom` (`above` / `below`), 1 = replicate border, 2 = zero border.        */
int itg_band_halo_fill(const itg_tensor* ext, const float* top, const float* bottom, int mode_top, int mode_bottom, void* stream);
int itg_band_halo_grad(const itg_tensor* gext, const float* above, const float* below, int mode_top, int mode_bottom, void* stream);
/* rows r0 / r1 of every image -> compact (n, W, ld) buffers (what the halo exchange sends); either output may be NULL */
int itg_band_rows_get(const itg_tensor* ext, int r0, int r1, float* out0, float* out1, void* stream);
/* to_ext != 0: rows 1 .. H of ext = band; else band = rows 1 .. H of ext (band: (n, 1, 1, H, W, ld))                      */
int itg_band_interior_copy(const itg_tensor* band, const itg_tensor* ext, int to_ext, void* stream);

/* ---- stream placement probe ---------------------------------------------------------------
 * One wave that occupies `stream`'s hardware queue for `microseconds` (<= 100000) and exits.  No reference counterpart
 * (the reference runs one CUDA stream, train.py:122-171); the step engine overlaps D(real) with the generator forward
 * and the weight gradients with the input-gradient chain on HIP streams, and HIP multiplexes streams onto at most
 * GPU_MAX_HW_QUEUES (4) hardware queues in creation order: two streams that share a queue run in order, and more than
 * four busy queues cost a third of the step.  engine.Trainer launches this on candidate streams in pairs and keeps
 * streams that ran concurrently.                                                           */
int itg_stream_spin(int microseconds, void* stream);

/* ---- layout conversion at the NCHW boundary -------------------------------------------- */
int itg_nchw_to_grid(const float* src, const itg_tensor* dst, int merged_src, void* stream);
int itg_grid_to_nchw(const itg_tensor* src, float* dst, int merged_dst, void* stream);

/* ---- BatchNorm2d (reference models/layers.py:279-280, generators.py:78; torch
 * semantics: biased batch variance, eps, momentum running stats with unbiased var) ------
 * stats: per-channel (sum, sumsq) in fp64 over all pixels of x -> sums[2*ld] (must be
 *        zeroed by the caller; several ranks may all-reduce it before finalize).
 * finalize: mean/rstd -> alpha = gamma*rstd, beta' = beta - mean*alpha (fp32, length
 *        ld each, written to ab[0..ld) and ab[ld..2ld)), saves mean/rstd, updates
 *        running stats (count = total pixels incl. other ranks; count_scale multiplies
 *        the count used for the unbiased factor only, 4 when stats were taken before
 *        a nearest x2 upsample).  gamma/beta NULL = affine-free (SSM's inner BN).
 * apply: y = act(alpha*x + beta') with optional nearest x2 upsample (y.ph == 2*x.ph).
 * bwd_reduce / bwd_apply: the autograd of apply∘finalize∘stats w.r.t. x, gamma, beta,
 *        act derivative taken on the recomputed pre-activation.                          */
int itg_bn_stats(const itg_tensor* x, double* sums, void* stream);
int itg_bn_finalize(const double* sums, double count, double count_scale, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* mean_rstd, float* ab,
                    int c, int ld, int training, void* stream);
/* itg_bn_finalize + itg_bn_apply of a training-mode BatchNorm in one launch (the statistics sums are
 * complete, e.g. after the cross-rank all-reduce): writes mean_rstd / ab for the backward, updates the
 * running statistics and num_batches_tracked exactly as itg_bn_finalize does.                       */
int itg_bn_finalize_apply(const itg_tensor* x, const double* sums, double count, double count_scale,
                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, float* mean_rstd, float* ab,
                          const itg_tensor* y, int act, float slope, void* stream);
int itg_bn_apply(const itg_tensor* x, const float* ab, const itg_tensor* y, int act, float slope,
                 void* stream);
int itg_bn_bwd_reduce(const itg_tensor* x, const itg_tensor* dy, const float* ab, const float* mean_rstd,
                      int act, float slope, double* sums, void* stream);
/* sums: (possibly all-reduced) global backward sums used for dx; sums_local: this rank's own sums that
 * become dgamma/dbeta (NULL = sums).  accumulate != 0 adds into dgamma/dbeta (gradient sinks).      */
int itg_bn_bwd_apply(const itg_tensor* x, const itg_tensor* dy, const float* ab, const float* mean_rstd,
                     const double* sums_local, const double* sums, double count, int act, float slope,
                     const itg_tensor* dx, float* dgamma, float* dbeta, int accumulate, void* stream);
/* The same with a second gradient of the BatchNorm's input added into dx (`addend`: x's shape, NULL = none, not dx
 * itself): the block input of reference models/layers.py:301-322 feeds bn1 AND the residual shortcut, and autograd's sum
 * of the two gradients is otherwise a launch of its own. */
int itg_bn_bwd_apply_add(const itg_tensor* x, const itg_tensor* dy, const float* ab, const float* mean_rstd,
                         const double* sums_local, const double* sums, double count, int act, float slope,
                         const itg_tensor* dx, float* dgamma, float* dbeta, int accumulate, const itg_tensor* addend,
                         void* stream);

/* ---- SSM modulation (reference models/layers.py:228-234): out = (1+gamma)*xhat + beta,
 * gamma/beta the two halves of `emb` (channels [0,c) and [c,2c)), then optional act ---- */
int itg_ssm_modulate_fwd(const itg_tensor* x, const float* mean_rstd, const itg_tensor* emb,
                         const itg_tensor* y, int act, float slope, void* stream);
int itg_ssm_modulate_bwd(const itg_tensor* x, const float* mean_rstd, const itg_tensor* emb,
                         const itg_tensor* dy, int act, float slope, const itg_tensor* dxhat,
                         const itg_tensor* demb, void* stream);

/* ---- pointwise ------------------------------------------------------------------------
 * act fwd/bwd: nn.LeakyReLU / nn.ReLU / nn.Tanh (layers.py:289-292, generators.py:121,
 * discriminators.py:188); bwd takes the forward OUTPUT.  upsample: nn.Upsample(x2,
 * nearest) (generators.py:52) and its 2x2-sum backward.  axpby: out = a*x + b*y.       */
int itg_act_fwd(const itg_tensor* x, const itg_tensor* y, int act, float slope, void* stream);
int itg_act_bwd(const itg_tensor* out, const itg_tensor* dout, const itg_tensor* dx, int act, float slope,
                void* stream);
int itg_upsample2x_fwd(const itg_tensor* x, const itg_tensor* y, void* stream);
int itg_upsample2x_bwd(const itg_tensor* dy, const itg_tensor* dx, void* stream);
int itg_add(const itg_tensor* a, const itg_tensor* b, const itg_tensor* out, void* stream);
/* flat helpers: out = a*(a_dev?*a_dev:1)*x + b*y over n floats (attention gate gamma*o + x at
 * layers.py:258, EMA of buffers at train.py:176-180); dot: *out(fp64, zeroed by the call) = <x,y> */
int itg_axpby(const float* x, const float* y, float* out, float a, const float* a_dev, float b, int64_t n,
              void* stream);
int itg_dot(const float* x, const float* y, int64_t n, double* out, void* stream);
int itg_colsum(const itg_tensor* x, float* out /*c*/, double* acc /*ld scratch*/, void* stream); /* bias grad */

/* ---- attention (reference models/layers.py:246-258), per patch ---------------------------
 * beta_save: itg_attention_scratch_floats() floats; fwd stores the softmax in the first
 * NB * HW * (HW/4) of them, bwd uses the rest as scratch (dS, or the per-chunk dK / dV shares). */
int64_t itg_attention_scratch_floats(const itg_tensor* theta, const itg_tensor* phi_pooled, const itg_tensor* g_pooled);
int itg_attention_fwd(const itg_tensor* theta, const itg_tensor* phi_pooled, const itg_tensor* g_pooled,
                      const itg_tensor* o_mid, float* beta_save, void* stream);
int itg_attention_bwd(const itg_tensor* theta, const itg_tensor* phi_pooled, const itg_tensor* g_pooled,
                      float* beta_save, const itg_tensor* d_o_mid, const itg_tensor* d_theta,
                      const itg_tensor* d_phi_pooled, const itg_tensor* d_g_pooled, void* stream);
int itg_maxpool2_fwd(const itg_tensor* x, const itg_tensor* y, void* stream);
int itg_maxpool2_bwd(const itg_tensor* x, const itg_tensor* y, const itg_tensor* dy, const itg_tensor* dx,
                     void* stream);

/* ---- losses (train.py:81,131-132,148-149,164-165): mean BCE-with-logits against a
 * constant target; grad = (sigmoid(x)-t)/count * upstream.  Hinge is a build-side extra
 * (the reference never reads --loss, utils.py:85): mode 0 = D real relu(1-x),
 * 1 = D fake relu(1+x), 2 = G  -x.  loss_out: one device float (must be zeroed).         */
int itg_bce_logits_fwd(const float* logits, int64_t count, float target, float* loss_out, void* stream);
int itg_bce_logits_bwd(const float* logits, int64_t count, float target, const float* upstream,
                       float* dlogits, void* stream);
/* The same heads on the logit map in the patch-grid layout the discriminator's last conv writes (c == 1): the mean loss and
 * d loss / d logit (upstream 1, same layout; padding channels zeroed) in ONE launch.  kind 0 = BCE against `target`,
 * 1..3 = hinge mode 0..2.  workspace_zeroed: itg_logit_loss_grid_workspace() doubles that are ZERO at launch (left dirty). */
int64_t itg_logit_loss_grid_workspace(void);
int itg_logit_loss_grid(const itg_tensor* logits, int kind, float target, float* loss_out, itg_tensor* dlogits,
                        void* workspace_zeroed, void* stream);
int itg_hinge_fwd(const float* logits, int64_t count, int mode, float* loss_out, void* stream);
int itg_hinge_bwd(const float* logits, int64_t count, int mode, const float* upstream, float* dlogits,
                  void* stream);

/* ---- spectral norm (torch.nn.utils.spectral_norm as used at layers.py:190-194): one
 * power iteration in place on u (rows) and v (cols), sigma = u^T W v, inv_sigma out;
 * workspace: 8*cols + rows floats ----------------------------------------------------------- */
int itg_spectral_norm_power_iter(const float* w, float* u, float* v, int rows, int cols, int do_iter,
                                 float eps, float* sigma_out, float* inv_sigma_out, float* workspace,
                                 void* stream);
/* the same iteration for n <= 8 layers in 4 launches; host arrays of n device pointers / sizes,
 * workspace[l]: 8*cols[l] + rows[l] floats, inv_sigma_out[l]: one device float                     */
int itg_spectral_norm_power_iter_multi(int n, const float* const* w, float* const* u, float* const* v,
                                       const int* rows, const int* cols, int do_iter, float eps,
                                       float* const* inv_sigma_out, float* const* workspace, void* stream);
/* dW_orig = (G - <G, W/sigma> u v^T) / sigma.  accumulate: ITG_ACC_DW adds into d_w_orig; ITG_WS_ZEROED = the fp64
 * accumulator in workspace is zero already (no memset launch).  */
int itg_spectral_norm_bwd(const float* g_w, const float* w_orig, const float* u, const float* v,
                          const float* inv_sigma, int rows, int cols, float* d_w_orig, int accumulate,
                          float* workspace, void* stream);
/* the second half of itg_spectral_norm_bwd for up to ITG_WGRAD_MAX_JOBS layers in one launch: the dots <G, W> come from
 * itg_wgrad_reduce_multi (job->dot).  accumulate: ITG_ACC_DW adds into d_w_orig.                                   */
typedef struct {
  const float* g_w; const float* u; const float* v; const float* inv_sigma; const double* dot; float* d_w_orig;
  int32_t rows, cols, accumulate, reserved;
} itg_sn_job;
int itg_spectral_norm_bwd_multi(const itg_sn_job* jobs, int n, void* stream);

/* ---- optimiser (train.py:57-58,153,169,176-180): Adam over a flat parameter buffer with
 * optional fused EMA (ema = decay*ema + (1-decay)*p); step >= 1, read from *step_dev (device int32)
 * when given so that a captured hipGraph replays with the right bias correction ---------------- */
int itg_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, int64_t count, float lr,
                      float beta1, float beta2, float eps, int step, const int32_t* step_dev, float ema_decay,
                      void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
