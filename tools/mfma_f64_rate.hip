// Micro-benchmark (not part of the product): issue rate of v_mfma_f64_16x16x4_f64 and of
// v_fma_f64 on gfx950, to decide which unit a contraction should use.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int t = 0; t < NACC; t++) acc[t] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  double s = 0;
  for (int t = 0; t < NACC; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int t = 0; t < NACC; t++) acc[t] = t;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) acc[t] = fma(acc[t], a, b);
  }
  double s = 0;
  for (int t = 0; t < NACC; t++) s += acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double *out;
  CK(hipMalloc(&out, 8 * 256 * 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wavesPerSimd = 1; wavesPerSimd <= 2; wavesPerSimd++) {
    const int blocks = 256 * wavesPerSimd;   // 4 waves per block: one per SIMD
    float ms;
#define RUN(K, NACC, FLOPS_PER_INST, NAME) \
    hipLaunchKernelGGL((K<NACC>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 2.0); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((K<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0); CK(hipEventRecord(e1)); \
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-10s acc=%d waves/SIMD=%d : %8.3f ms  %7.2f TFLOP/s  %6.1f clk/inst/wave at 2.4 GHz\n", NAME, NACC, wavesPerSimd, ms, \
           (double)blocks * 4 * iters * NACC * FLOPS_PER_INST / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * wavesPerSimd));
    RUN(k_mfma, 1, 2048.0, "mfma_f64")
    RUN(k_mfma, 4, 2048.0, "mfma_f64")
    RUN(k_fma, 8, 128.0, "v_fma_f64")
  }
  return 0;
}
