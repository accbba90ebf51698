// What does the one store of the Gram-Schmidt update cost, and does writing in phases common to
// the whole device avoid it?  y' = beta*y + X q over K columns of n rows (column-major, the
// layout of the basis), every thread owning row pairs and walking all K columns:
//   A  nothing stored (the ceiling of this access pattern)
//   B  every result stored as soon as it is formed (plain or non-temporal store)
//   C  cooperative launch: every workgroup keeps the results of a chunk of rows in registers,
//      the grid synchronises, then all workgroups store their chunks together
//   hipcc -O3 --offload-arch=gfx950 tools/write_phase.hip -o tools/bin/write_phase
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct qarg { double v[64]; };

__device__ __forceinline__ double2 ld_nt(const double2 *p) { double2 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); return r; }
__device__ __forceinline__ void fence2(double2 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }

// sum over the K columns for one row pair p, 16 columns requested together
__device__ __forceinline__ double2 row_sum(const double2 *__restrict__ X2, int64_t ld2, int K, const qarg &q, int64_t p) {
  double2 a = make_double2(0.0, 0.0);
  int c = 0;
  for (; c + 16 <= K; c += 16) {
    double2 xv[16];
#pragma unroll
    for (int u = 0; u < 16; u++) xv[u] = ld_nt(X2 + (int64_t)(c + u) * ld2 + p);
#pragma unroll
    for (int u = 0; u < 16; u++) fence2(xv[u]);
#pragma unroll
    for (int u = 0; u < 16; u++) { a.x = fma(xv[u].x, q.v[c + u], a.x); a.y = fma(xv[u].y, q.v[c + u], a.y); }
  }
  for (; c < K; c++) { const double2 x = ld_nt(X2 + (int64_t)c * ld2 + p); a.x = fma(x.x, q.v[c], a.x); a.y = fma(x.y, q.v[c], a.y); }
  return a;
}

template <int MODE>   // 0: no store, 1: plain store, 2: non-temporal store
__global__ __launch_bounds__(256) void k_direct(const double *__restrict__ X, int64_t ld, int K, qarg q, double beta,
                                                double *__restrict__ y, int64_t n, double *sink) {
  const double2 *X2 = reinterpret_cast<const double2 *>(X);
  double2 *y2 = reinterpret_cast<double2 *>(y);
  const int64_t npair = n >> 1, stride = (int64_t)gridDim.x * 256;
  double keep = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npair; p += stride) {
    const double2 yv = y2[p];
    double2 a = row_sum(X2, ld >> 1, K, q, p);
    a.x = fma(beta, yv.x, a.x); a.y = fma(beta, yv.y, a.y);
    if (MODE == 1) y2[p] = a;
    else if (MODE == 2) { __builtin_nontemporal_store(a.x, &y2[p].x); __builtin_nontemporal_store(a.y, &y2[p].y); }
    else keep += a.x + a.y;
  }
  if (MODE == 0 && keep == 1.2345e-300) sink[0] = keep;
}

// R row pairs per thread and round kept in registers; all workgroups store after a grid barrier
template <int R>
__global__ __launch_bounds__(256) void k_phased(const double *__restrict__ X, int64_t ld, int K, qarg q, double beta,
                                                double *__restrict__ y, int64_t n) {
  cg::grid_group grid = cg::this_grid();
  const double2 *X2 = reinterpret_cast<const double2 *>(X);
  double2 *y2 = reinterpret_cast<double2 *>(y);
  const int64_t npair = n >> 1;
  const int64_t per_round = (int64_t)gridDim.x * 256 * R;
  const int64_t rounds = (npair + per_round - 1) / per_round;          // the same for every thread of the grid
  for (int64_t r = 0; r < rounds; r++) {
    double2 res[R];
    const int64_t base = r * per_round + (int64_t)blockIdx.x * 256 * R + threadIdx.x;
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int64_t p = base + (int64_t)i * 256;
      res[i] = make_double2(0.0, 0.0);
      if (p < npair) {
        const double2 yv = y2[p];
        double2 a = row_sum(X2, ld >> 1, K, q, p);
        res[i].x = fma(beta, yv.x, a.x); res[i].y = fma(beta, yv.y, a.y);
      }
    }
    grid.sync();
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int64_t p = base + (int64_t)i * 256;
      if (p < npair) y2[p] = res[i];
    }
  }
}

// D: no barrier: every workgroup computes the results of its chunk of rows (R row pairs per thread, in
// registers) and stores them in one burst when it is done
template <int R>
__global__ __launch_bounds__(256) void k_burst(const double *__restrict__ X, int64_t ld, int K, qarg q, double beta,
                                               double *__restrict__ y, int64_t n) {
  const double2 *X2 = reinterpret_cast<const double2 *>(X);
  double2 *y2 = reinterpret_cast<double2 *>(y);
  const int64_t npair = n >> 1;
  const int64_t base = (int64_t)blockIdx.x * 256 * R + threadIdx.x;
  double2 res[R];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int64_t p = base + (int64_t)i * 256;
    res[i] = make_double2(0.0, 0.0);
    if (p < npair) {
      const double2 yv = y2[p];
      double2 a = row_sum(X2, ld >> 1, K, q, p);
      res[i].x = fma(beta, yv.x, a.x); res[i].y = fma(beta, yv.y, a.y);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int64_t p = base + (int64_t)i * 256;
    if (p < npair) y2[p] = res[i];
  }
}

// E: as D, but the rows are split over several launches of one round of workgroups each, so that the
// bursts of a launch all fall at its end: the launch boundary is the barrier
template <int R>
__global__ __launch_bounds__(256) void k_burst_part(const double *__restrict__ X, int64_t ld, int K, qarg q, double beta,
                                                    double *__restrict__ y, int64_t n, int64_t pair0) {
  const double2 *X2 = reinterpret_cast<const double2 *>(X);
  double2 *y2 = reinterpret_cast<double2 *>(y);
  const int64_t npair = n >> 1;
  const int64_t base = pair0 + (int64_t)blockIdx.x * 256 * R + threadIdx.x;
  double2 res[R];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int64_t p = base + (int64_t)i * 256;
    res[i] = make_double2(0.0, 0.0);
    if (p < npair) {
      const double2 yv = y2[p];
      double2 a = row_sum(X2, ld >> 1, K, q, p);
      res[i].x = fma(beta, yv.x, a.x); res[i].y = fma(beta, yv.y, a.y);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int64_t p = base + (int64_t)i * 256;
    if (p < npair) y2[p] = res[i];
  }
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 16777216;
  const int K = argc > 2 ? atoi(argv[2]) : 48;
  const int64_t ld = n;
  double *X, *y, *sink;
  CHECK(hipMalloc(&X, (size_t)ld * K * sizeof(double)));
  CHECK(hipMalloc(&y, (size_t)n * sizeof(double)));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(X, 0, (size_t)ld * K * sizeof(double)));
  CHECK(hipMemset(y, 0, (size_t)n * sizeof(double)));
  qarg q; for (int c = 0; c < 64; c++) q.v[c] = 1.0 / (c + 1);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes_rd = 8.0 * n * (K + 1), bytes_wr = 8.0 * n;
  auto time = [&](const char *name, double bytes, auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
      CHECK(hipEventRecord(e0, 0)); launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    printf("%-46s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6); fflush(stdout);
  };
  printf("n = %lld rows, K = %d columns: %.2f GB read, %.3f GB written per launch\n", (long long)n, K, bytes_rd / 1e9, bytes_wr / 1e9);
  for (int grid : {1024, 2048, 4096}) {
    char nm[96];
    snprintf(nm, sizeof nm, "A no store, grid %d", grid);
    time(nm, bytes_rd, [&] { hipLaunchKernelGGL((k_direct<0>), dim3(grid), dim3(256), 0, 0, X, ld, K, q, 1.0, y, n, sink); });
    snprintf(nm, sizeof nm, "B plain store, grid %d", grid);
    time(nm, bytes_rd + bytes_wr, [&] { hipLaunchKernelGGL((k_direct<1>), dim3(grid), dim3(256), 0, 0, X, ld, K, q, 1.0, y, n, sink); });
    snprintf(nm, sizeof nm, "B non-temporal store, grid %d", grid);
    time(nm, bytes_rd + bytes_wr, [&] { hipLaunchKernelGGL((k_direct<2>), dim3(grid), dim3(256), 0, 0, X, ld, K, q, 1.0, y, n, sink); });
  }
  {
    const int64_t npair = n >> 1;
    char nm[96];
#define BURST(R) snprintf(nm, sizeof nm, "D burst of %d KiB per workgroup, no barrier", 256 * R * 16 / 1024); \
    time(nm, bytes_rd + bytes_wr, [&] { hipLaunchKernelGGL((k_burst<R>), dim3((unsigned)((npair + 256 * R - 1) / (256 * R))), dim3(256), 0, 0, X, ld, K, q, 1.0, y, n); });
    BURST(4) BURST(8) BURST(16) BURST(32) BURST(48)
#undef BURST
  }
  int dev = 0, ncu = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  {
    const int64_t npair = n >> 1;
    char nm[128];
#define PART(R) { int per_cu = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_burst_part<R>, 256, 0)); \
    for (int use = per_cu; use >= 1; use = (use > 2 ? use - 1 : use - 1)) { \
      const int64_t G = (int64_t)ncu * use, per = G * 256 * R; const int L = (int)((npair + per - 1) / per); \
      snprintf(nm, sizeof nm, "E %d launches of %d workgroups/CU (max %d), %d KiB bursts", L, use, per_cu, 256 * R * 16 / 1024); \
      time(nm, bytes_rd + bytes_wr, [&] { for (int l = 0; l < L; l++) { const int64_t left = npair - (int64_t)l * per; \
        const unsigned g = (unsigned)((left < per ? left : per) + 256 * R - 1) / (256 * R); \
        hipLaunchKernelGGL((k_burst_part<R>), dim3(g), dim3(256), 0, 0, X, ld, K, q, 1.0, y, n, (int64_t)l * per); } }); } }
    PART(2) PART(4) PART(8) PART(16)
#undef PART
  }
  auto coop = [&](auto kern, int R, const char *label) {
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0));
    for (int want : {1, 2, 4, 8}) {
      if (want > per_cu) break;
      const int grid = ncu * want;
      double beta = 1.0; int Kk = K; int64_t nn = n, ldd = ld;
      void *args[] = {(void *)&X, (void *)&ldd, (void *)&Kk, (void *)&q, (void *)&beta, (void *)&y, (void *)&nn};
      char nm[96];
      snprintf(nm, sizeof nm, "C phased, %s, %d workgroups per CU (max %d)", label, want, per_cu);
      time(nm, bytes_rd + bytes_wr, [&] { CHECK(hipLaunchCooperativeKernel((const void *)kern, dim3(grid), dim3(256), args, 0, 0)); });
    }
  };
  coop(k_phased<4>, 4, "4 row pairs per thread");
  coop(k_phased<8>, 8, "8 row pairs per thread");
  coop(k_phased<16>, 16, "16 row pairs per thread");
  return 0;
}
