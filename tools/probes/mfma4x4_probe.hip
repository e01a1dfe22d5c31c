// Probe of the operand / result lane maps of v_mfma_f32_4x4x1_16b_f32 on gfx950 (not documented in the guides):
// 16 independent 4x4 outer products per instruction.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

int main() {
  float ha[64], hb[64], hd[256];
  // a[l] = 100 + l, b[l] = 1000 + l: product a[x] * b[y] identifies (x, y) uniquely
  for (int l = 0; l < 64; ++l) { ha[l] = 100.f + l; hb[l] = 1000.f + l; }
  float *da, *db, *dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      // find (x, y) with (100 + x) * (1000 + y) == value
      int fx = -1, fy = -1;
      for (int x = 0; x < 64 && fx < 0; ++x)
        for (int y = 0; y < 64; ++y)
          if ((100.f + x) * (1000.f + y) == hd[l * 4 + r]) { fx = x; fy = y; break; }
      printf("  reg%d = a[%2d]*b[%2d]", r, fx, fy);
    }
    printf("\n");
  }
  return 0;
}
