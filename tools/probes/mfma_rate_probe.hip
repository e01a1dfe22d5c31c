// What issue patterns of v_mfma_f32_16x16x4_f32 reach the 32-cycle rate?  hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Kernels: NACC independent accumulators, MODE 0 = one (a, b) register pair for all, 1 = a and b rotate over 4 registers each (as a
// K loop's fragment components do), 2 = MODE 1 + one ds_read_b128 per 8 MFMAs with the wait right behind it, 3 = MODE 1 + a
// wave-uniform scalar branch every 8 MFMAs.  Grid = 256 CUs x waves-per-SIMD x 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int flag) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a = {1.f + threadIdx.x, 2.f, 3.f, 4.f}, b = {0.5f, 0.25f, 0.125f, 1.f};
  const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {      // 8 groups of NACC x 4 MFMAs... each group = 8 MFMAs when NACC == 2
      if (MODE == 2) { a = lp[(it * 8 + g) & 63]; }
      if (MODE == 3) { if (flag == it * 8 + g) { b = b * 2.f; } }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE == 0 ? a[0] : a[s], MODE == 0 ? b[0] : b[(s + i) & 3], acc[i], 0, 0, 0);
    }
  }
  f32x4 r = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) r += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}

static int g_mfmas_per_wave = 0, g_reps = 1;
template <int NACC, int MODE>
void run(const char* name, float* out) {
  for (int wps = 1; wps <= 4; ++wps) {
    const int blocks = 256 * wps, iters = g_mfmas_per_wave ? g_mfmas_per_wave / (32 * NACC) : 2000 / NACC * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, -1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < g_reps; ++r) hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, -1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= g_reps;
    const double mfmas_per_simd = (double)wps * iters * 8 * 4 * NACC;
    const double tf = mfmas_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d: %8.1f us  %6.1f TF  (%.1f ns per MFMA and SIMD)\n", name, wps, ms * 1e3, tf, ms * 1e6 / mfmas_per_simd);
  }
}

int main(int argc, char** argv) {
  if (argc > 1) g_mfmas_per_wave = atoi(argv[1]);
  if (argc > 2) g_reps = atoi(argv[2]);
  float* out; (void)hipMalloc(&out, 256 * 4 * 256 * 4);
  run<2, 0>("2 acc, fixed operands", out);
  run<2, 1>("2 acc, rotating operands", out);
  run<6, 1>("6 acc, rotating operands", out);
  run<12, 1>("12 acc, rotating operands", out);
  run<2, 2>("2 acc + ds_read_b128 per 8", out);
  run<6, 2>("6 acc + ds_read_b128 per 24", out);
  run<6, 3>("6 acc + scalar branch per 24", out);
  return 0;
}
