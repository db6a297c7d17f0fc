// Sustained MFMA rate of the device, registers only (no memory traffic): what a kernel can reach at best at the clocks the
// part actually holds under matrix load.  hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(256) void f32_kernel(float* out, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void bf16_kernel(float* out, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.0f + j); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;
}
template <typename K>
static double run(K kern, int blocks, int iters, double flop_per_mfma, int chains) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters / 8);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  const double flops = (double)blocks * 4 * iters * 4 * chains * flop_per_mfma;
  printf("  %d blocks x 4 waves, %d chains, %.1f ms: %.1f TFLOP/s\n", blocks, chains, ms, flops / ms * 1e-9);
  return flops / ms * 1e-9;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  const int cus = p.multiProcessorCount;
  printf("v_mfma_f32_32x32x2_f32 (4096 flop, nominal 64 cycles):\n");
  for (int rep = 0; rep < 2; ++rep) {
    run(f32_kernel<1>, cus * 1, 40000, 4096.0, 1);
    run(f32_kernel<2>, cus * 1, 40000, 4096.0, 2);
    run(f32_kernel<4>, cus * 1, 20000, 4096.0, 4);
    run(f32_kernel<2>, cus * 2, 20000, 4096.0, 2);
    run(f32_kernel<4>, cus * 2, 60000, 4096.0, 4);     // ~0.5 s: sustained clocks
  }
  printf("v_mfma_f32_32x32x16_bf16 (32768 flop, nominal 32 cycles):\n");
  for (int rep = 0; rep < 2; ++rep) {
    run(bf16_kernel<1>, cus * 1, 40000, 32768.0, 1);
    run(bf16_kernel<2>, cus * 1, 40000, 32768.0, 2);
    run(bf16_kernel<4>, cus * 2, 100000, 32768.0, 4);
  }
  return 0;
}
