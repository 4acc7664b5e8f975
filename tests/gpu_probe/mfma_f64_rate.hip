// Sustained rate of v_mfma_f64_16x16x4_f64 on every CU (operands in registers, 16 independent accumulators per wave,
// 1 or 2 waves per SIMD): the ceiling the FP64 GEMM tile kernel is priced against.
// hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters) {
  double4_t acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *d;
  hipMalloc(&d, sizeof(double) * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wgs : {256, 512, 1024}) {
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)wgs * 4 * iters * 16 * 2048.0;
    printf("%4d workgroups of 4 waves: %.3f ms, %.1f TFLOP/s (f64 16x16x4)\n", wgs, ms, flops / ms * 1e-9);
  }
  return 0;
}
