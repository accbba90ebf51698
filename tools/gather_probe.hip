// Gather-rate probe for the power-law SpMV: how many 8-byte x entries per second can the
// memory system deliver to a kernel that streams 4-byte indices (coalesced, non-temporal) and
// gathers x[idx], for tables of several sizes (L1 / L2 / Infinity Cache / HBM resident), index
// streams of several shapes, and three ways of issuing the gather.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/bin/gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void fence_i(int &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void fence_d(double &v) { asm volatile("" : "+v"(v)); }

// MODE 0: plain load, 1: non-temporal load, 2: table in LDS (first `ldsn` entries; others from global)
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ idx, int64_t n, const double *__restrict__ x,
                                                double *__restrict__ out, int ldsn) {
  extern __shared__ double tab[];
  if (MODE == 2) {
    for (int i = threadIdx.x; i < ldsn; i += 256) tab[i] = x[i];
    __syncthreads();
  }
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    int ci[U];
    double xv[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t p = base + (int64_t)u * 256; ci[u] = p < n ? __builtin_nontemporal_load(idx + p) : 0; }
#pragma unroll
    for (int u = 0; u < U; u++) fence_i(ci[u]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (MODE == 0) xv[u] = x[ci[u]];
      else if (MODE == 1) xv[u] = __builtin_nontemporal_load(x + ci[u]);
      else xv[u] = ci[u] < ldsn ? tab[ci[u]] : x[ci[u]];
    }
#pragma unroll
    for (int u = 0; u < U; u++) fence_d(xv[u]);
#pragma unroll
    for (int u = 0; u < U; u++) acc += xv[u];
  }
  if (acc == 1.2345e-300) out[0] = acc;      // keep the loads alive
}

// The index stream cut into 8 parts, part g consumed only by workgroups with blockIdx % 8 == g
// (the workgroups of one XCD under round-robin dispatch): with the table laid out so that part
// g's indices fall into its own eighth of the table, every XCD's L2 caches a different eighth.
template <int U>
__global__ __launch_bounds__(256) void k_gather_parts(const int *__restrict__ idx, int64_t npart, const double *__restrict__ x,
                                                      double *__restrict__ out) {
  const int g = blockIdx.x & 7;
  const int64_t b = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const int *ip = idx + (int64_t)g * npart;
  double acc = 0.0;
  const int64_t stride = nb * 256 * U;
  for (int64_t base = b * 256 * U + threadIdx.x; base < npart; base += stride) {
    int ci[U];
    double xv[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t p = base + (int64_t)u * 256; ci[u] = p < npart ? __builtin_nontemporal_load(ip + p) : 0; }
#pragma unroll
    for (int u = 0; u < U; u++) fence_i(ci[u]);
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = x[ci[u]];
#pragma unroll
    for (int u = 0; u < U; u++) fence_d(xv[u]);
#pragma unroll
    for (int u = 0; u < U; u++) acc += xv[u];
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : (int64_t)1 << 26;          // gathers per launch
  int *d_idx; double *d_x, *d_out;
  const int64_t maxtab = (int64_t)1 << 26;                                  // 512 MB of x
  CHECK(hipMalloc(&d_idx, n * sizeof(int)));
  CHECK(hipMalloc(&d_x, maxtab * sizeof(double)));
  CHECK(hipMalloc(&d_out, 64));
  CHECK(hipMemset(d_x, 0, maxtab * sizeof(double)));
  std::vector<int> h((size_t)n);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int64_t sizes[] = {4096, 262144, 1048576, 8388608, 33554432, 67108864};
  printf("gathers per launch: %lld; rates in G gathers/s (kernel also streams 4 B of index per gather)\n", (long long)n);
  printf("%-34s %10s %10s %10s %10s\n", "table / index stream", "plain U8", "nt U8", "plain U16", "lds8k+U8");
  for (int shape = 0; shape < 3; shape++) {
    for (int64_t tab : sizes) {
      // shape 0: uniform random; 1: sorted in runs of 2048 (what sorting a tile's entries by column gives);
      // 2: power-law mix like an R-MAT column stream (0.76/0.24 per bit), the table ordered by popularity
      int bits = 0; while (((int64_t)1 << bits) < tab) bits++;
      std::vector<int> rank;
      if (shape == 2) {
        // position of every index when the table is ordered by popularity (few one-bits first):
        // the packed layout the operator gives its most referenced columns
        rank.resize((size_t)tab);
        std::vector<int64_t> start(bits + 2, 0);
        for (int64_t v = 0; v < tab; v++) start[__builtin_popcountll(v) + 1]++;
        for (int b = 0; b <= bits; b++) start[b + 1] += start[b];
        for (int64_t v = 0; v < tab; v++) rank[(size_t)v] = (int)start[__builtin_popcountll(v)]++;
      }
      for (int64_t i = 0; i < n; i++) {
        if (shape == 2) {
          // 0.24 per bit: 6 bits of randomness per index bit (15/64 = 0.234)
          int v = 0;
          uint64_t r = 0;
          for (int b = 0; b < bits; b++) { if (b % 10 == 0) r = rnd(); v = (v << 1) | ((r & 63) < 15 ? 1 : 0); r >>= 6; }
          h[(size_t)i] = rank[(size_t)v];
        } else h[(size_t)i] = (int)(rnd() % (uint64_t)tab);
      }
      if (shape == 1) for (int64_t i = 0; i + 2048 <= n; i += 2048) std::sort(h.begin() + i, h.begin() + i + 2048);
      CHECK(hipMemcpy(d_idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
      double rate[4] = {0, 0, 0, 0};
      for (int v = 0; v < 4; v++) {
        for (int rep = 0; rep < 3; rep++) {
          CHECK(hipEventRecord(e0, 0));
          const int grid = 2048;
          if (v == 0) hipLaunchKernelGGL((k_gather<0, 8>), dim3(grid), dim3(256), 0, 0, d_idx, n, d_x, d_out, 0);
          if (v == 1) hipLaunchKernelGGL((k_gather<1, 8>), dim3(grid), dim3(256), 0, 0, d_idx, n, d_x, d_out, 0);
          if (v == 2) hipLaunchKernelGGL((k_gather<0, 16>), dim3(grid), dim3(256), 0, 0, d_idx, n, d_x, d_out, 0);
          if (v == 3) hipLaunchKernelGGL((k_gather<2, 8>), dim3(512), dim3(256), 65536, 0, d_idx, n, d_x, d_out, 8192);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) rate[v] = rate[v] > n / (ms * 1e6) ? rate[v] : n / (ms * 1e6);
        }
      }
      char name[64];
      snprintf(name, sizeof name, "%s %7.2f MB", shape == 0 ? "uniform" : shape == 1 ? "sorted/2048" : "powerlaw", tab * 8 / 1048576.0);
      printf("%-34s %10.1f %10.1f %10.1f %10.1f\n", name, rate[0], rate[1], rate[2], rate[3]);
      fflush(stdout);
    }
  }
  // power-law stream, table ordered by popularity, then dealt to 8 groups by rank (rank r -> group r % 8,
  // slot r / 8 of the group's eighth of the table): all workgroups on everything, against XCD g on group g only
  printf("%-34s %10s %10s\n", "powerlaw dealt to 8 groups", "all XCDs", "XCD=group");
  for (int64_t tab : {(int64_t)8388608, (int64_t)67108864}) {
    int bits = 0; while (((int64_t)1 << bits) < tab) bits++;
    std::vector<int> rank((size_t)tab);
    std::vector<int64_t> start(bits + 2, 0);
    for (int64_t v = 0; v < tab; v++) start[__builtin_popcountll(v) + 1]++;
    for (int b = 0; b <= bits; b++) start[b + 1] += start[b];
    for (int64_t v = 0; v < tab; v++) rank[(size_t)v] = (int)start[__builtin_popcountll(v)]++;
    const int64_t npart = n / 8;
    std::vector<int64_t> fill(8, 0);
    std::vector<int> hp((size_t)n, 0);
    int64_t placed = 0;
    while (placed < 8 * npart) {
      int v = 0; uint64_t r = 0;
      for (int b = 0; b < bits; b++) { if (b % 10 == 0) r = rnd(); v = (v << 1) | ((r & 63) < 15 ? 1 : 0); r >>= 6; }
      const int rk = rank[(size_t)v], g = rk & 7;
      if (fill[g] >= npart) continue;
      hp[(size_t)(g * npart + fill[g]++)] = (int)(g * (tab / 8) + (rk >> 3));
      placed++;
    }
    CHECK(hipMemcpy(d_idx, hp.data(), n * sizeof(int), hipMemcpyHostToDevice));
    double rate[2] = {0, 0};
    for (int v = 0; v < 2; v++)
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        if (v == 0) hipLaunchKernelGGL((k_gather<0, 8>), dim3(2048), dim3(256), 0, 0, d_idx, 8 * npart, d_x, d_out, 0);
        else hipLaunchKernelGGL((k_gather_parts<8>), dim3(2048), dim3(256), 0, 0, d_idx, npart, d_x, d_out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) rate[v] = rate[v] > 8 * npart / (ms * 1e6) ? rate[v] : 8 * npart / (ms * 1e6);
      }
    char name[64];
    snprintf(name, sizeof name, "powerlaw %7.2f MB", tab * 8 / 1048576.0);
    printf("%-34s %10.1f %10.1f\n", name, rate[0], rate[1]);
    fflush(stdout);
  }
  return 0;
}
