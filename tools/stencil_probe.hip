// What a 7-point stencil product costs beyond its streams: y[i] = sum_k v_k x[i + o_k] over 2^27 rows with the
// offsets of the 512^3 grid (+-1, +-512, +-512^2), one 8-byte word of "codes" read per row as the row-code
// kernel does, with 1, 2 or 4 consecutive rows per lane (8-, 16- and 32-byte accesses) and 3, 5 or 7 points.
//   hipcc -O3 --offload-arch=gfx950 tools/stencil_probe.hip -o tools/bin/stencil_probe && tools/bin/stencil_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); return 1; } } while (0)

typedef double __attribute__((ext_vector_type(2), aligned(8))) d2u;    // two doubles, 8-byte aligned
typedef double __attribute__((ext_vector_type(4), aligned(8))) d4u;

// R rows per lane, NP points (3: +-1; 5: +-1, +-p1; 7: +-1, +-p1, +-p2).  Rows [lo, hi) only: the borders are left out.
template <int R, int NP>
__global__ __launch_bounds__(256) void k_stencil(const uint64_t *__restrict__ codes, const double *__restrict__ x, double *__restrict__ y,
                                                 int64_t lo, int64_t hi, int64_t p1, int64_t p2) {
  const int64_t stride = (int64_t)gridDim.x * 256 * R;
  for (int64_t r0 = lo + ((int64_t)blockIdx.x * 256 + threadIdx.x) * R; r0 + R <= hi; r0 += stride) {
    uint64_t c[R];
    double xm[R + 2], a[R], b[R], cc[R], d[R];
    if (R == 1) {
      c[0] = __builtin_nontemporal_load(codes + r0);
      xm[0] = x[r0 - 1]; xm[1] = x[r0]; xm[2] = x[r0 + 1];
      if (NP >= 5) { a[0] = x[r0 - p1]; b[0] = x[r0 + p1]; }
      if (NP >= 7) { cc[0] = x[r0 - p2]; d[0] = x[r0 + p2]; }
    } else if (R == 2) {
      const d2u cw = *reinterpret_cast<const d2u *>(codes + r0);
      c[0] = __double_as_longlong(cw.x); c[1] = __double_as_longlong(cw.y);
      const d2u l = *reinterpret_cast<const d2u *>(x + r0 - 1), h = *reinterpret_cast<const d2u *>(x + r0 + 1);
      xm[0] = l.x; xm[1] = l.y; xm[2] = h.x; xm[3] = h.y;
      if (NP >= 5) { const d2u u = *reinterpret_cast<const d2u *>(x + r0 - p1), v = *reinterpret_cast<const d2u *>(x + r0 + p1); a[0] = u.x; a[1] = u.y; b[0] = v.x; b[1] = v.y; }
      if (NP >= 7) { const d2u u = *reinterpret_cast<const d2u *>(x + r0 - p2), v = *reinterpret_cast<const d2u *>(x + r0 + p2); cc[0] = u.x; cc[1] = u.y; d[0] = v.x; d[1] = v.y; }
    } else {
      const d4u cw = *reinterpret_cast<const d4u *>(codes + r0);
      c[0] = __double_as_longlong(cw.x); c[1] = __double_as_longlong(cw.y); c[2] = __double_as_longlong(cw.z); c[3] = __double_as_longlong(cw.w);
      const d4u m = *reinterpret_cast<const d4u *>(x + r0);
      xm[0] = x[r0 - 1]; xm[1] = m.x; xm[2] = m.y; xm[3] = m.z; xm[4] = m.w; xm[5] = x[r0 + 4];
      if (NP >= 5) { const d4u u = *reinterpret_cast<const d4u *>(x + r0 - p1), v = *reinterpret_cast<const d4u *>(x + r0 + p1);
                     a[0] = u.x; a[1] = u.y; a[2] = u.z; a[3] = u.w; b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w; }
      if (NP >= 7) { const d4u u = *reinterpret_cast<const d4u *>(x + r0 - p2), v = *reinterpret_cast<const d4u *>(x + r0 + p2);
                     cc[0] = u.x; cc[1] = u.y; cc[2] = u.z; cc[3] = u.w; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    }
    double s[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
      const double w = (double)(c[i] & 7);
      s[i] = 6.0 * xm[i + 1] - xm[i] - xm[i + 2] + w;
      if (NP >= 5) s[i] -= a[i] + b[i];
      if (NP >= 7) s[i] -= cc[i] + d[i];
    }
    if (R == 1) __builtin_nontemporal_store(s[0], y + r0);
    else if (R == 2) { d2u o; o.x = s[0]; o.y = s[1]; __builtin_nontemporal_store(o, reinterpret_cast<d2u *>(y + r0)); }
    else { d4u o; o.x = s[0]; o.y = s[1]; o.z = s[2]; o.w = s[3]; __builtin_nontemporal_store(o, reinterpret_cast<d4u *>(y + r0)); }
  }
}

template <int R, int NP>
static int run(const uint64_t *codes, const double *x, double *y, int64_t n, int64_t p1, int64_t p2, int grid) {
  const int64_t lo = p2, hi = n - p2;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_stencil<R, NP>), dim3(grid), dim3(256), 0, 0, codes, x, y, lo, hi, p1, p2);
  CHECK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stencil<R, NP>), dim3(grid), dim3(256), 0, 0, codes, x, y, lo, hi, p1, p2);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<double> h(4);
  CHECK(hipMemcpy(h.data(), y + lo + 1000, 32, hipMemcpyDeviceToHost));
  printf("%d rows per lane, %d points, grid %6d: %.4f ms  %.0f GB/s of 24 B per row   y = %.6g %.6g\n", R, NP, grid, ms, 24.0 * (hi - lo) / ms / 1e6, h[0], h[3]);
  return 0;
}

__global__ void k_fill(uint64_t *codes, double *x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { codes[i] = (uint64_t)(i * 2654435761u); x[i] = (double)(i % 1000) * 1e-3; }
}

int main() {
  const int64_t n = 1ll << 27, p1 = 512, p2 = 512 * 512;
  uint64_t *codes; double *x, *y;
  CHECK(hipMalloc(&codes, n * 8)); CHECK(hipMalloc(&x, n * 8)); CHECK(hipMalloc(&y, n * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, codes, x, n);
  CHECK(hipDeviceSynchronize());
  for (int grid : {16384, 4096, 2048}) {
    if (run<1, 3>(codes, x, y, n, p1, p2, grid)) return 1;
    if (run<1, 5>(codes, x, y, n, p1, p2, grid)) return 1;
    if (run<1, 7>(codes, x, y, n, p1, p2, grid)) return 1;
    if (run<2, 3>(codes, x, y, n, p1, p2, grid)) return 1;
    if (run<2, 7>(codes, x, y, n, p1, p2, grid)) return 1;
    if (run<4, 3>(codes, x, y, n, p1, p2, grid)) return 1;
    if (run<4, 7>(codes, x, y, n, p1, p2, grid)) return 1;
  }
  return 0;
}
