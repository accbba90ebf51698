// Tuning harness (not part of the product): HBM read rate of a pure streaming kernel for
// each cache-policy modifier combination of global_load_dwordx4 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

#define LOADS(MOD) \
  asm volatile("global_load_dwordx4 %0, %4, off " MOD "\n\t" \
               "global_load_dwordx4 %1, %5, off " MOD "\n\t" \
               "global_load_dwordx4 %2, %6, off " MOD "\n\t" \
               "global_load_dwordx4 %3, %7, off " MOD "\n\t" \
               "s_waitcnt vmcnt(0)" \
               : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(x + p), "v"(x + p + stride), "v"(x + p + 2 * stride), "v"(x + p + 3 * stride) : "memory")

template <int POL>
__global__ __launch_bounds__(256) void k_read(const v2d *__restrict__ x, int64_t n2, double *out) {
  double acc = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p + 3 * stride < n2; p += 4 * stride) {
    v2d a, b, c, d;
    if (POL == 0) LOADS("");
    else if (POL == 1) LOADS("nt");
    else if (POL == 2) LOADS("sc0");
    else if (POL == 3) LOADS("sc1");
    else if (POL == 4) LOADS("sc0 sc1");
    else if (POL == 5) LOADS("sc0 nt");
    else if (POL == 6) LOADS("sc1 nt");
    else LOADS("sc0 sc1 nt");
    acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

int main() {
  const int64_t n = (int64_t)1 << 30;      // 8 GiB of doubles
  double *x, *out;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(x, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  for (int g : {256, 512, 1024, 2048}) {
    for (int pol = 0; pol < 8; pol++) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        switch (pol) {
#define C(P) case P: hipLaunchKernelGGL((k_read<P>), dim3(g), dim3(256), 0, 0, (const v2d *)x, n / 2, out); break;
          C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("grid %5d  %-12s %8.3f ms  %7.1f GB/s\n", g, names[pol], best, n * 8.0 / best / 1e6);
    }
  }
  return 0;
}
