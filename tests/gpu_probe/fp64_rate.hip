// FP64 issue rates on gfx950, register-only loops on every CU, with the shader clock read during the loop:
//   (a) v_fma_f64 (vector unit), 1/2/4/8 waves per SIMD
//   (b) v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 (matrix unit)
// Explains the ceiling the FP64 GEMM tile kernel is priced against (profiles/r02_fp64_rate.txt).
// hipcc --offload-arch=gfx950 -O3 fp64_rate.hip -o fp64_rate && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fma(double *out, unsigned long long *clk, int iters) {
  double acc[32];
  for (int i = 0; i < 32; i++) acc[i] = i * 1e-3;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-12;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 32; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

__global__ __launch_bounds__(256) void k_mfma16(double *out, unsigned long long *clk, int iters) {
  double4_t acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

__global__ __launch_bounds__(256) void k_mfma4(double *out, unsigned long long *clk, int iters) {
  double acc[32];
  for (int i = 0; i < 32; i++) acc[i] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 32; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

// the same instruction with DIFFERENT operand registers from one instruction to the next (8 a-operands x 8 b-operands, 64
// accumulators), as a GEMM tile issues them
__global__ __launch_bounds__(256) void k_mfma4_var(double *out, unsigned long long *clk, int iters) {
  double acc[64], a[8], b[8];
  for (int i = 0; i < 64; i++) acc[i] = 0.0;
  for (int i = 0; i < 8; i++) {
    a[i] = threadIdx.x * 1e-3 + i;
    b[i] = 1.0 + threadIdx.x * 1e-4 * (i + 1);
  }
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i * 8 + j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i * 8 + j], 0, 0, 0);
    // keep the operands live and changing so that nothing is hoisted
#pragma unroll
    for (int i = 0; i < 8; i++) {
      a[i] += 1e-9;
      b[i] -= 1e-9;
    }
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 64; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

__global__ __launch_bounds__(256) void k_mfma16_var(double *out, unsigned long long *clk, int iters) {
  double4_t acc[16];
  double a[4], b[4];
  for (int i = 0; i < 16; i++) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < 4; i++) {
    a[i] = threadIdx.x * 1e-3 + i;
    b[i] = 1.0 + threadIdx.x * 1e-4 * (i + 1);
  }
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a[i] += 1e-9;
      b[i] -= 1e-9;
    }
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

// Round 3: the two 16x16x4 loops above do NOT measure the instruction -- hipcc keeps their accumulators in VGPRs across the
// loop and copies all 128 registers into AGPRs and back around the 16 MFMAs of every iteration (256 v_accvgpr moves per
// 16 MFMAs in the ISA), which is where their 36 / 31.5 TFLOP/s came from.  These loops issue the instruction through inline
// assembly with the accumulators pinned in VGPRs: 16 independent accumulators (a dependent MFMA is 16 instructions = more
// than the 16 passes of the pipeline away), the same operand pair everywhere or 4 x 4 distinct operand registers.
__global__ __launch_bounds__(256) void k_mfma16_asm(double *out, unsigned long long *clk, int iters) {
  double4_t acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}
__global__ __launch_bounds__(256) void k_mfma16_asm_var(double *out, unsigned long long *clk, int iters) {
  double4_t acc[16];
  double a[4], b[4];
  for (int i = 0; i < 16; i++) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < 4; i++) {
    a[i] = threadIdx.x * 1e-3 + i;
    b[i] = 1.0 + threadIdx.x * 1e-4 * (i + 1);
  }
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[4 * i + j]) : "v"(a[i]), "v"(b[j]));
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

template <typename K>
static void run(const char *name, K kern, double flops_per_wave_iter, int iters) {
  double *d;
  unsigned long long *clk;
  hipMalloc(&d, sizeof(double) * 256 * 8192);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs : {256, 512, 1024, 2048}) {
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, clk, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, clk, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)wgs * 4 * iters * flops_per_wave_iter;
    // wall_clock64 ticks at 100 MHz; clock64 at the shader clock
    printf("%-28s %4d workgroups of 4 waves: %8.3f ms  %6.1f TFLOP/s   shader clock during the loop %.0f MHz\n", name, wgs, ms,
           flops / ms * 1e-9, (double)h[0] / (double)h[1] * 100.0);
  }
  hipFree(d);
  hipFree(clk);
}

int main() {
  run("v_fma_f64 x32", k_fma, 32 * 64 * 2.0, 20000);
  run("v_mfma_f64_16x16x4 x16", k_mfma16, 16 * 2048.0, 20000);
  run("v_mfma_f64_4x4x4_4b x32", k_mfma4, 32 * 4 * 4 * 4 * 4 * 2.0, 20000);
  run("4x4x4_4b, 8x8 operands", k_mfma4_var, 64 * 512.0, 10000);
  run("16x16x4, 4x4 operands", k_mfma16_var, 16 * 2048.0, 10000);
  run("16x16x4 asm, pinned acc", k_mfma16_asm, 16 * 2048.0, 10000);
  run("16x16x4 asm, 4x4 operands", k_mfma16_asm_var, 16 * 2048.0, 10000);
  return 0;
}
