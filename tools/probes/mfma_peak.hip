// Calibration probe (not part of the library): sustained v_mfma_f32_16x16x32_bf16 rate with operands in registers,
// random vs zero data, 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void __launch_bounds__(256) k(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x + 256 * i) % 1024]; b[i] = in[(threadIdx.x + 256 * i + 517) % 1024]; }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int n = 1024 * 8;
  std::vector<unsigned short> h(n);
  bf16x8* d; float* o;
  hipMalloc(&d, n * 2); hipMalloc(&o, 4096 * 256 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < n; ++i) {                 // bf16 bit patterns: random values in [-1,1) or zeros
      float f = mode ? (rand() / (float)RAND_MAX * 2.f - 1.f) : 0.f;
      unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
      const int grid = 256 * blocks_per_cu, iters = 20000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, o, 2000);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 5.0 * grid * 4.0 * iters * 16 * 16384.0;
      printf("%s data, %d wave(s)/SIMD: %.1f TFLOP/s  (%.3f ms per launch)\n", mode ? "random" : "zero  ", blocks_per_cu,
             flop / (ms * 1e-3) / 1e12, ms / 5);
    }
  }
  return 0;
}
