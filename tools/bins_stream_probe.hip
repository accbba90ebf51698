// Streaming-rate probe for the two passes of the binned SpMV (ksg_bins.hip): which access shape
// reaches what rate.  Synthetic data: N products (8 B) with a 2-byte row number each, cut into bins
// of BIN entries.  Build: hipcc -O3 --offload-arch=gfx950 tools/bins_stream_probe.hip -o /tmp/bins_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include <algorithm>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// R1: one wave per bin, one entry per lane and load, U loads in flight, sums in LDS
template <int U, bool ADD, bool ROWS>
__global__ __launch_bounds__(64) void r1(const uint16_t *__restrict__ rl, const double *__restrict__ xe, double *__restrict__ y, int64_t bin, int H) {
  extern __shared__ double acc[];
  const int lane = threadIdx.x;
  for (int i = lane; i < H; i += 64) acc[i] = 0.0;
  const int64_t p0 = (int64_t)blockIdx.x * bin, p1 = p0 + bin;
  double t = 0.0;
  for (int64_t p = p0 + lane; p + 64 * (U - 1) < p1; p += 64 * U) {
    uint16_t r[U]; double v[U];
#pragma unroll
    for (int j = 0; j < U; j++) { r[j] = ROWS ? __builtin_nontemporal_load(rl + p + 64 * j) : (uint16_t)(lane + j); v[j] = __builtin_nontemporal_load(xe + p + 64 * j); }
#pragma unroll
    for (int j = 0; j < U; j++) { if (ADD) atomicAdd(&acc[r[j] & (H - 1)], v[j]); else t += v[j] * (double)r[j]; }
  }
  if (!ADD) acc[lane] = t;
  for (int i = lane; i < H; i += 64) y[(int64_t)blockIdx.x * H + i] = acc[i];
}

// R4: one wave per bin, four consecutive entries per lane: one 8-byte load of rows, two 16-byte loads of products
template <int U>
__global__ __launch_bounds__(64) void r4(const uint16_t *__restrict__ rl, const double *__restrict__ xe, double *__restrict__ y, int64_t bin, int H) {
  extern __shared__ double acc[];
  const int lane = threadIdx.x;
  for (int i = lane; i < H; i += 64) acc[i] = 0.0;
  const int64_t p0 = (int64_t)blockIdx.x * bin, p1 = p0 + bin;
  for (int64_t p = p0 + 4 * lane; p + 256 * (U - 1) < p1; p += 256 * U) {
    u32x2 r[U]; f64x2 a[U], b[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      r[j] = __builtin_nontemporal_load((const u32x2 *)(rl + p + 256 * j));
      a[j] = __builtin_nontemporal_load((const f64x2 *)(xe + p + 256 * j));
      b[j] = __builtin_nontemporal_load((const f64x2 *)(xe + p + 256 * j + 2));
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      atomicAdd(&acc[(r[j].x & 0xffff) & (H - 1)], a[j].x); atomicAdd(&acc[(r[j].x >> 16) & (H - 1)], a[j].y);
      atomicAdd(&acc[(r[j].y & 0xffff) & (H - 1)], b[j].x); atomicAdd(&acc[(r[j].y >> 16) & (H - 1)], b[j].y);
    }
  }
  for (int i = lane; i < H; i += 64) y[(int64_t)blockIdx.x * H + i] = acc[i];
}

// R256: a 256-thread workgroup, every wave its own bin and its own LDS slice
template <int U>
__global__ __launch_bounds__(256) void r256(const uint16_t *__restrict__ rl, const double *__restrict__ xe, double *__restrict__ y, int64_t bin, int H) {
  extern __shared__ double accs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double *acc = accs + (size_t)w * H;
  for (int i = lane; i < H; i += 64) acc[i] = 0.0;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  const int64_t p0 = b * bin, p1 = p0 + bin;
  for (int64_t p = p0 + lane; p + 64 * (U - 1) < p1; p += 64 * U) {
    uint16_t r[U]; double v[U];
#pragma unroll
    for (int j = 0; j < U; j++) { r[j] = __builtin_nontemporal_load(rl + p + 64 * j); v[j] = __builtin_nontemporal_load(xe + p + 64 * j); }
#pragma unroll
    for (int j = 0; j < U; j++) atomicAdd(&acc[r[j] & (H - 1)], v[j]);
  }
  for (int i = lane; i < H; i += 64) y[b * H + i] = acc[i];
}

// E1: expand-like: a workgroup streams a task of TG groups: 2-byte column in, LDS lookup, 8-byte product out
template <int U, int T, bool LDS>
__global__ __launch_bounds__(T) void e1(const uint16_t *__restrict__ c16, double *__restrict__ xe, const double *__restrict__ x, int64_t tg, int W) {
  extern __shared__ double xs[];
  if (LDS) { for (int i = threadIdx.x; i < W; i += T) xs[i] = x[(int64_t)(blockIdx.x % 64) * W + i]; __syncthreads(); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t g0 = (int64_t)blockIdx.x * tg, g1 = g0 + tg;
  for (int64_t g = g0 + wave * U; g + U <= g1; g += (T / 64) * U) {
    uint16_t u[U];
#pragma unroll
    for (int j = 0; j < U; j++) u[j] = __builtin_nontemporal_load(c16 + (g + j) * 64 + lane);
#pragma unroll
    for (int j = 0; j < U; j++) xe[(g + j) * 64 + lane] = LDS ? xs[u[j] & (W - 1)] : (double)u[j];
  }
}

// E4: the same with four consecutive entries per lane (8-byte load of columns, two 16-byte stores)
template <int U, int T>
__global__ __launch_bounds__(T) void e4(const uint16_t *__restrict__ c16, double *__restrict__ xe, const double *__restrict__ x, int64_t tg, int W) {
  extern __shared__ double xs[];
  for (int i = threadIdx.x; i < W; i += T) xs[i] = x[(int64_t)(blockIdx.x % 64) * W + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t g0 = (int64_t)blockIdx.x * tg, g1 = g0 + tg;      // groups of 64; a wave step takes 4 groups
  for (int64_t g = g0 + wave * 4 * U; g + 4 * U <= g1; g += (T / 64) * 4 * U) {
    u32x2 u[U];
#pragma unroll
    for (int j = 0; j < U; j++) u[j] = __builtin_nontemporal_load((const u32x2 *)(c16 + (g + 4 * j) * 64 + 4 * lane));
#pragma unroll
    for (int j = 0; j < U; j++) {
      f64x2 a, b;
      a.x = xs[(u[j].x & 0xffff) & (W - 1)]; a.y = xs[(u[j].x >> 16) & (W - 1)];
      b.x = xs[(u[j].y & 0xffff) & (W - 1)]; b.y = xs[(u[j].y >> 16) & (W - 1)];
      *(f64x2 *)(xe + (g + 4 * j) * 64 + 4 * lane) = a;
      *(f64x2 *)(xe + (g + 4 * j) * 64 + 4 * lane + 2) = b;
    }
  }
}

// ES: expand-like with scattered runs: the groups of a task go to destinations dst[g] (one offset per run of RL groups, read
// like the run table of the real kernel: a dependent load before the store)
template <int U, int T, bool NT = false>
__global__ __launch_bounds__(T) void es(const uint16_t *__restrict__ c16, double *__restrict__ xe, const double *__restrict__ x, int64_t tg, int W,
                                        const int64_t *__restrict__ dst, int rl, int live) {
  extern __shared__ double xs[];
  for (int i = threadIdx.x; i < W; i += T) xs[i] = x[(int64_t)(blockIdx.x % 64) * W + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t g0 = (int64_t)blockIdx.x * tg, g1 = g0 + tg;
  for (int64_t g = g0 + wave * U; g + U <= g1; g += (T / 64) * U) {
    uint16_t u[U]; int64_t d[U];
#pragma unroll
    for (int j = 0; j < U; j++) u[j] = __builtin_nontemporal_load(c16 + (g + j) * 64 + lane);
#pragma unroll
    for (int j = 0; j < U; j++) d[j] = dst[(g + j) / rl + (u[j] >> 16)];
#pragma unroll
    for (int j = 0; j < U; j++) if (lane < live) {
      if (NT) __builtin_nontemporal_store(xs[u[j] & (W - 1)], &xe[d[j] + ((g + j) % rl) * 64 + lane]);
      else xe[d[j] + ((g + j) % rl) * 64 + lane] = xs[u[j] & (W - 1)];
    }
  }
}

// RS: reduce-like with the products read run by run from scattered places: run k (rl entries) starts at src[k]; the rows
// stream in order.  What the second pass would do if the first wrote its products in its own order.
template <int U>
__global__ __launch_bounds__(64) void rs(const uint16_t *__restrict__ rl, const double *__restrict__ xe, double *__restrict__ y, int64_t bin, int H,
                                         const int64_t *__restrict__ src, int rlen) {
  extern __shared__ double acc[];
  const int lane = threadIdx.x;
  for (int i = lane; i < H; i += 64) acc[i] = 0.0;
  const int64_t p0 = (int64_t)blockIdx.x * bin, p1 = p0 + bin;
  for (int64_t p = p0 + lane; p + 64 * (U - 1) < p1; p += 64 * U) {
    uint16_t r[U]; int64_t s[U]; double v[U];
#pragma unroll
    for (int j = 0; j < U; j++) { r[j] = __builtin_nontemporal_load(rl + p + 64 * j); s[j] = src[(p + 64 * j) / rlen]; }
#pragma unroll
    for (int j = 0; j < U; j++) v[j] = __builtin_nontemporal_load(xe + s[j] + (p + 64 * j) % rlen);
#pragma unroll
    for (int j = 0; j < U; j++) atomicAdd(&acc[r[j] & (H - 1)], v[j]);
  }
  for (int i = lane; i < H; i += 64) y[(int64_t)blockIdx.x * H + i] = acc[i];
}

template <class F>
static void timeit(const char *name, int64_t n, double bytes_per_entry, F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  CK(hipGetLastError());
  printf("%-58s %8.3f ms  %7.1f G entries/s  %6.2f TB/s\n", name, ms, n / ms / 1e6, n * bytes_per_entry / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int64_t N = (int64_t)256 << 20;
  const int64_t bin = argc > 1 ? atoll(argv[1]) : 32768;
  const int H = 2048;
  uint16_t *rl; double *xe, *y, *x;
  CK(hipMalloc(&rl, N * 2)); CK(hipMalloc(&xe, N * 8)); CK(hipMalloc(&y, (N / bin) * H * 8)); CK(hipMalloc(&x, 64 * 16384 * 8));
  std::vector<uint16_t> h((size_t)N);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < N; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[(size_t)i] = (uint16_t)(s >> 20); }
  CK(hipMemcpy(rl, h.data(), N * 2, hipMemcpyHostToDevice));
  CK(hipMemset(xe, 0, N * 8)); CK(hipMemset(x, 0, 64 * 16384 * 8));
  const int nb = (int)(N / bin);
  printf("N = %lld entries, bins of %lld (%d bins), H = %d\n", (long long)N, (long long)bin, nb, H);
  timeit("R1 U=8  wave/bin, add", N, 10, [&] { hipLaunchKernelGGL((r1<8, true, true>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H); });
  timeit("R1 U=16 wave/bin, add", N, 10, [&] { hipLaunchKernelGGL((r1<16, true, true>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H); });
  timeit("R1 U=16 wave/bin, no add", N, 10, [&] { hipLaunchKernelGGL((r1<16, false, true>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H); });
  timeit("R1 U=16 wave/bin, no add, no row stream", N, 8, [&] { hipLaunchKernelGGL((r1<16, false, false>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H); });
  timeit("R1 U=16 wave/bin, add, no row stream", N, 8, [&] { hipLaunchKernelGGL((r1<16, true, false>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H); });
  timeit("R1 U=16 wave/bin, add, LDS 8 KiB (20 waves/CU)", N, 10, [&] { hipLaunchKernelGGL((r1<16, true, true>), dim3(nb), dim3(64), 1024 * 8, 0, rl, xe, y, bin, 1024); });
  timeit("R1 U=16 wave/bin, add, LDS 4 KiB (32 waves/CU)", N, 10, [&] { hipLaunchKernelGGL((r1<16, true, true>), dim3(nb), dim3(64), 512 * 8, 0, rl, xe, y, bin, 512); });
  timeit("R4 U=4  wave/bin, 4 entries per lane, add", N, 10, [&] { hipLaunchKernelGGL((r4<4>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H); });
  timeit("R4 U=8  wave/bin, 4 entries per lane, add", N, 10, [&] { hipLaunchKernelGGL((r4<8>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H); });
  timeit("R256 U=16 4 waves per workgroup, a bin each, add", N, 10, [&] { hipLaunchKernelGGL((r256<16>), dim3(nb / 4), dim3(256), 4 * H * 8, 0, rl, xe, y, bin, H); });
  const int64_t tg = 4096;
  const int nt = (int)(N / 64 / tg);
  CK(hipFuncSetAttribute((const void *)e1<16, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void *)e4<4, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  timeit("E1 U=16 1024 threads, W=8192, LDS lookup", N, 10, [&] { hipLaunchKernelGGL((e1<16, 1024, true>), dim3(nt), dim3(1024), 8192 * 8, 0, rl, xe, x, tg, 8192); });
  timeit("E1 U=16 1024 threads, W=16384, LDS lookup", N, 10, [&] { hipLaunchKernelGGL((e1<16, 1024, true>), dim3(nt), dim3(1024), 16384 * 8, 0, rl, xe, x, tg, 16384); });
  timeit("E1 U=16 1024 threads, no lookup", N, 10, [&] { hipLaunchKernelGGL((e1<16, 1024, false>), dim3(nt), dim3(1024), 0, 0, rl, xe, x, tg, 8192); });
  timeit("E1 U=8  256 threads, W=4096, LDS lookup", N, 10, [&] { hipLaunchKernelGGL((e1<8, 256, true>), dim3(nt), dim3(256), 4096 * 8, 0, rl, xe, x, tg, 4096); });
  timeit("E1 U=16 512 threads, W=8192, LDS lookup", N, 10, [&] { hipLaunchKernelGGL((e1<16, 512, true>), dim3(nt), dim3(512), 8192 * 8, 0, rl, xe, x, tg, 8192); });
  timeit("E4 U=4  1024 threads, W=8192, 4 entries per lane", N, 10, [&] { hipLaunchKernelGGL((e4<4, 1024>), dim3(nt), dim3(1024), 8192 * 8, 0, rl, xe, x, tg, 8192); });
  timeit("E4 U=4  1024 threads, W=16384, 4 entries per lane", N, 10, [&] { hipLaunchKernelGGL((e4<4, 1024>), dim3(nt), dim3(1024), 16384 * 8, 0, rl, xe, x, tg, 16384); });
  // row patterns of the reduce pass: sorted rows within runs of 64 (what a cell's entries look like), rows repeated, and a
  // share of the lanes adding to one address (what fill entries that all name row 0 do)
  {
    uint16_t *rl2; CK(hipMalloc(&rl2, N * 2));
    std::vector<uint16_t> h2((size_t)N);
    for (int pat = 0; pat < 7; pat++) {
      for (int64_t i = 0; i < N; i += 64) {
        uint16_t t[64];
        for (int l = 0; l < 64; l++) t[l] = h[(size_t)(i + l)] & 2047;
        if (pat == 0) std::sort(t, t + 64);
        if (pat == 1) { for (int l = 0; l < 64; l++) t[l] = h[(size_t)(i + l / 4)] & 2047; std::sort(t, t + 64); }
        if (pat == 2) { for (int l = 52; l < 64; l++) t[l] = 0; }
        if (pat == 3) { for (int l = 52; l < 64; l++) t[l] = (uint16_t)(l - 52); }
        if (pat == 4) { for (int l = 0; l < 64; l++) t[l] = h[(size_t)(i + l / 16)] & 2047; std::sort(t, t + 64); }
        if (pat == 5) { for (int l = 0; l < 64; l++) t[l] = h[(size_t)i] & 2047; }
        if (pat == 6) { if ((i / 64) % 5 == 0) for (int l = 0; l < 64; l++) t[l] = h[(size_t)i] & 2047; else std::sort(t, t + 64); }
        for (int l = 0; l < 64; l++) h2[(size_t)(i + l)] = t[l];
      }
      CK(hipMemcpy(rl2, h2.data(), N * 2, hipMemcpyHostToDevice));
      const char *names[7] = {"R1 U=16 add, rows sorted within groups of 64", "R1 U=16 add, every row four times in a row, sorted",
                              "R1 U=16 add, 12 of 64 lanes add to row 0", "R1 U=16 add, 12 of 64 lanes add to rows 0..11",
                              "R1 U=16 add, every row 16 times in a row", "R1 U=16 add, all 64 lanes of a group the same row",
                              "R1 U=16 add, one group in five all the same row"};
      timeit(names[pat], N, 10, [&] { hipLaunchKernelGGL((r1<16, true, true>), dim3(nb), dim3(64), H * 8, 0, rl2, xe, y, bin, H); });
    }
    CK(hipFree(rl2));
  }
  // the second pass reading scattered runs
  for (int rlen : {8, 16, 32, 64, 256}) for (int shift : {0, 3}) {
    const int64_t nrun = N / rlen;
    std::vector<int64_t> perm((size_t)nrun);
    for (int64_t k = 0; k < nrun; k++) perm[(size_t)k] = k;
    uint64_t z = 7654321;
    for (int64_t k = nrun - 1; k > 0; k--) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; std::swap(perm[(size_t)k], perm[(size_t)(z % (uint64_t)(k + 1))]); }
    for (int64_t k = 0; k < nrun; k++) perm[(size_t)k] = perm[(size_t)k] * rlen + (perm[(size_t)k] == nrun - 1 ? 0 : shift);
    int64_t *srcd; CK(hipMalloc(&srcd, nrun * 8)); CK(hipMemcpy(srcd, perm.data(), nrun * 8, hipMemcpyHostToDevice));
    char name[128];
    snprintf(name, sizeof name, "RS U=16 wave/bin, add, products read in scattered runs of %3d%s", rlen, shift ? ", off by 24 bytes" : "");
    timeit(name, N, 10, [&] { hipLaunchKernelGGL((rs<16>), dim3(nb), dim3(64), H * 8, 0, rl, xe, y, bin, H, srcd, rlen); });
    CK(hipFree(srcd));
  }
  // scattered runs: the groups of 64 entries in runs of rl groups; run k goes to position perm[k] * rl * 64 (+ shift)
  for (int rlen : {1, 2}) for (int shift : {0, 3, 8, 16, 100, 101}) {
    const int live = shift >= 100 ? (shift == 100 ? 60 : 56) : 64;
    if (shift >= 100 && rlen > 1) continue;
    const int64_t nrun = N / 64 / rlen;
    std::vector<int64_t> perm((size_t)nrun);
    for (int64_t k = 0; k < nrun; k++) perm[(size_t)k] = k;
    uint64_t z = 1234567;
    for (int64_t k = nrun - 1; k > 0; k--) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; std::swap(perm[(size_t)k], perm[(size_t)(z % (uint64_t)(k + 1))]); }
    // the last run must not run over the end with the shift: keep its slot for itself
    for (int64_t k = 0; k < nrun; k++) perm[(size_t)k] = perm[(size_t)k] * rlen * 64 + (perm[(size_t)k] == nrun - 1 || shift >= 100 ? 0 : shift);
    int64_t *dst; CK(hipMalloc(&dst, nrun * 8)); CK(hipMemcpy(dst, perm.data(), nrun * 8, hipMemcpyHostToDevice));
    char name[128];
    snprintf(name, sizeof name, "ES U=8 512 thr W=8192, runs of %4d entries, %s", rlen * 64, shift == 0 ? "aligned to 512 B" : shift == 3 ? "off by 24 bytes" : shift == 8 ? "off by 64 bytes" : shift == 16 ? "off by 128 bytes" : shift == 100 ? "aligned, 60 of 64 written" : "aligned, 56 of 64 written");
    CK(hipFuncSetAttribute((const void *)es<8, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    timeit(name, N, 10, [&] { hipLaunchKernelGGL((es<8, 512>), dim3(nt), dim3(512), 8192 * 8, 0, rl, xe, x, tg, 8192, dst, rlen, live); });
    CK(hipFuncSetAttribute((const void *)es<8, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    strcat(name, " NT");
    timeit(name, N, 10, [&] { hipLaunchKernelGGL((es<8, 512, true>), dim3(nt), dim3(512), 8192 * 8, 0, rl, xe, x, tg, 8192, dst, rlen, live); });
    CK(hipFree(dst));
  }
  return 0;
}
