// Round 6 probe: does the shape of a wave's 16-byte stores matter for HBM write throughput on gfx950?
//   pattern A: lane (p = lane & 15, g = lane >> 4) stores 16 B at row p * 256 B + (16 i + 4 g) * 4 B, i = 0..3 - the conv epilogues' shape:
//              one store instruction = 16 rows x 64 contiguous bytes, a pixel's 256 B completed by four instructions
//   pattern B: lane stores 16 B at row (lane >> 4) * 256 B + (lane & 15) * 16 B - one instruction = 4 rows x 256 contiguous bytes
// Both write the same 75.5 MB (295 k rows of 64 floats).  build: hipcc --offload-arch=gfx950 -O3 store_pattern_probe.hip -o /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void pat_a(float* out, int rows) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int p = lane & 15, g = lane >> 4;
  for (int r0 = wave * 16; r0 < rows; r0 += nw * 16) {
    f32x4 v = {1.f * r0, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(out + (size_t)(r0 + p) * 64 + 16 * i + 4 * g) = v;
  }
}
__global__ void pat_b(float* out, int rows) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int r0 = wave * 16; r0 < rows; r0 += nw * 16) {
    f32x4 v = {1.f * r0, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(out + (size_t)(r0 + 4 * i + (lane >> 4)) * 64 + (lane & 15) * 4) = v;
  }
}
int main() {
  const int rows = 8 * 192 * 192;
  float* d; hipMalloc(&d, (size_t)rows * 64 * 4 * 9);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pat = 0; pat < 2; ++pat)
    for (int blocks : {512, 1024, 2048, 4096}) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) {                       // 8 different 75 MB buffers: past the Infinity Cache
          float* o = d + (size_t)(k + 1) * rows * 64;
          if (pat == 0) hipLaunchKernelGGL(pat_a, dim3(blocks), dim3(256), 0, 0, o, rows);
          else hipLaunchKernelGGL(pat_b, dim3(blocks), dim3(256), 0, 0, o, rows);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("pattern %c, %4d blocks: %.1f us per 75.5 MB = %.2f TB/s\n", pat ? 'B' : 'A', blocks, best / 8 * 1e3, rows * 256.0 / (best / 8 * 1e-3) / 1e12);
    }
  return 0;
}
