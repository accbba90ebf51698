// Tuning harness (not part of the product): streaming-kernel variants for the
// tall-skinny GEMV^T / GEMV at the benchmark shape, to find what the HBM system
// sustains for these access patterns.  hipcc --offload-arch=gfx950 -O3 tools/tune_stream.hip -o tune_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <bool NT> __device__ __forceinline__ double2 ld2(const double2 *p) {
  if (NT) { double2 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); return r; }
  return *p;
}

// pure read: sum of everything
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ x, int64_t n2, double *out) {
  double a = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; p + (U - 1) * stride < n2; p += U * stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ld2<NT>(x + p + u * stride);
#pragma unroll
    for (int u = 0; u < U; u++) a += v[u].x + v[u].y;
  }
  for (; p < n2; p += stride) { double2 v = x[p]; a += v.x + v.y; }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, a);
}

template <int C, bool NT, int BLK>
__global__ __launch_bounds__(BLK) void k_dot(const double *__restrict__ X, int64_t ld, int k, const double *__restrict__ y,
                                             int64_t n, double *__restrict__ partial, int nb) {
  const int c0 = blockIdx.y * C;
  const int kc = min(C, k - c0);
  const double *col[C];
#pragma unroll
  for (int c = 0; c < C; c++) col[c] = X + (int64_t)(c0 + (c < kc ? c : kc - 1)) * ld;
  double acc[C];
#pragma unroll
  for (int c = 0; c < C; c++) acc[c] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const int64_t npair = n >> 1;
  const double2 *y2 = (const double2 *)y;
  for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < npair; p += stride) {
    const double2 yv = y2[p];
    double2 xv[C];
#pragma unroll
    for (int c = 0; c < C; c++) xv[c] = ld2<NT>((const double2 *)col[c] + p);
#pragma unroll
    for (int c = 0; c < C; c++) { acc[c] = fma(xv[c].x, yv.x, acc[c]); acc[c] = fma(xv[c].y, yv.y, acc[c]); }
  }
  __shared__ double sm[BLK / 64][C];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; c++) { const double v = wave_sum(acc[c]); if (lane == 0) sm[w][c] = v; }
  __syncthreads();
  if ((int)threadIdx.x < kc) {
    double s = 0;
    for (int ww = 0; ww < BLK / 64; ww++) s += sm[ww][threadIdx.x];
    partial[(int64_t)(c0 + threadIdx.x) * nb + blockIdx.x] = s;
  }
}

struct qarg { double v[128]; };
template <int U, bool NT, int BLK>
__global__ __launch_bounds__(BLK) void k_mv(const double *__restrict__ X, int64_t ld, int k, qarg q, double *__restrict__ y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const int64_t npair = n >> 1;
  double2 *y2 = (double2 *)y;
  const int64_t ld2v = ld >> 1;
  for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < npair; p += stride) {
    double2 a = y2[p];
    const double2 *xp = (const double2 *)X + p;
    int j = 0;
    for (; j + U <= k; j += U) {
      double2 xv[U];
#pragma unroll
      for (int u = 0; u < U; u++) xv[u] = ld2<NT>(xp + (int64_t)(j + u) * ld2v);
#pragma unroll
      for (int u = 0; u < U; u++) { a.x = fma(xv[u].x, q.v[j + u], a.x); a.y = fma(xv[u].y, q.v[j + u], a.y); }
    }
    for (; j < k; j++) { const double2 xv = xp[(int64_t)j * ld2v]; a.x = fma(xv.x, q.v[j], a.x); a.y = fma(xv.y, q.v[j], a.y); }
    y2[p] = a;
  }
}

// dot with the columns split over the waves of one block: wave w owns columns
// [w*CW, w*CW+CW) and every wave of the block walks the same rows, so y comes from
// HBM once and the other waves hit it in cache.
template <int CW, bool NT>
__global__ void k_dotw(const double *__restrict__ X, int64_t ld, int k, const double *__restrict__ y,
                       int64_t n, double *__restrict__ partial, int nb) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = w * CW;
  const int kc = min(CW, k - c0);
  const double *col[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) col[c] = X + (int64_t)(c0 + (c < kc ? c : kc - 1)) * ld;
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  const int64_t npair = n >> 1;
  const double2 *y2 = (const double2 *)y;
  for (int64_t p = (int64_t)blockIdx.x * 64 + lane; p < npair; p += (int64_t)gridDim.x * 64) {
    const double2 yv = y2[p];
    double2 xv[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) xv[c] = ld2<NT>((const double2 *)col[c] + p);
#pragma unroll
    for (int c = 0; c < CW; c++) { acc[c] = fma(xv[c].x, yv.x, acc[c]); acc[c] = fma(xv[c].y, yv.y, acc[c]); }
  }
#pragma unroll
  for (int c = 0; c < CW; c++) { const double v = wave_sum(acc[c]); if (lane == 0 && c < kc) partial[(int64_t)(c0 + c) * nb + blockIdx.x] = v; }
}

// fused y' = beta*y + X q ; c' = [X y']^T y' (k dots + self dot).  4 waves, wave w owns CW columns.
__device__ double *g_yout;
template <int CW, bool NT, bool PF, int DIAG = 0>
__global__ __launch_bounds__(256) void k_mvdot(const double *__restrict__ X, int64_t ld, int k, qarg q, double beta,
                                               double *__restrict__ y, int64_t n, double *__restrict__ partial, int nb) {
  __shared__ double2 sm[2][4][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = w * CW;
  const int kc = max(0, min(CW, k - c0));
  const double2 *col[CW];
  double qc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const int cc = c0 + (c < kc ? c : (kc > 0 ? kc - 1 : -c0));
    col[c] = (const double2 *)(X + (int64_t)cc * ld);
    qc[c] = c < kc ? q.v[c0 + c] : 0.0;
  }
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  double self = 0.0;
  const int64_t npair = n >> 1;
  double2 *y2 = (double2 *)y;
  const int64_t step = (int64_t)gridDim.x * 64;
  int buf = 0;
  int64_t p0 = (int64_t)blockIdx.x * 64;
  double2 xn[CW]; double2 yn = make_double2(0, 0);
  if (PF && p0 < npair) {
    const int64_t p = min(p0 + lane, npair - 1);
#pragma unroll
    for (int c = 0; c < CW; c++) xn[c] = ld2<NT>(col[c] + p);
    yn = y2[p];
  }
  for (; p0 < npair; p0 += step, buf ^= 1) {
    const int64_t p = p0 + lane;
    const bool live = p < npair;
    const int64_t pc = live ? p : npair - 1;
    double2 xv[CW]; double2 yv;
    if (PF) {
#pragma unroll
      for (int c = 0; c < CW; c++) xv[c] = xn[c];
      yv = yn;
      if (p0 + step < npair) {
        const int64_t pn = min(p + step, npair - 1);
#pragma unroll
        for (int c = 0; c < CW; c++) xn[c] = ld2<NT>(col[c] + pn);
        yn = y2[pn];
      }
    } else {
#pragma unroll
      for (int c = 0; c < CW; c++) xv[c] = ld2<NT>(col[c] + pc);
      yv = (DIAG == 4 || DIAG == 6) ? ld2<true>(y2 + pc) : y2[pc];
    }
    double2 a = make_double2(0, 0);
#pragma unroll
    for (int c = 0; c < CW; c++) { a.x = fma(xv[c].x, qc[c], a.x); a.y = fma(xv[c].y, qc[c], a.y); }
    double2 s0, s1, s2, s3;
    if (DIAG == 2) { s0 = a; s1 = a; s2 = a; s3 = a; }
    else {
    sm[buf][w][lane] = a;
    __syncthreads();
    s0 = sm[buf][0][lane]; s1 = sm[buf][1][lane]; s2 = sm[buf][2][lane]; s3 = sm[buf][3][lane];
    }
    double2 r;
    r.x = beta * yv.x + ((s0.x + s1.x) + (s2.x + s3.x));
    r.y = beta * yv.y + ((s0.y + s1.y) + (s2.y + s3.y));
    if (!live) { r.x = 0; r.y = 0; }
    if (w == 0) { if (live && DIAG != 1 && !(DIAG == 7 && ((p0 / step) & 3)) && !(DIAG == 8 && ((p0 / step) & 15))) { if (DIAG == 5) { ((double2 *)g_yout)[p] = r; } else if (DIAG == 9) { ((double2 *)g_yout)[p & 0xFFFF] = r; } else if (DIAG == 10) { typedef double v2d __attribute__((ext_vector_type(2))); v2d rv = {r.x, r.y}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(&y2[p]), "v"(rv) : "memory"); } else if (DIAG == 3 || DIAG == 6) { __builtin_nontemporal_store(r.x, &y2[p].x); __builtin_nontemporal_store(r.y, &y2[p].y); } else y2[p] = r; } self = fma(r.x, r.x, self); self = fma(r.y, r.y, self); }
#pragma unroll
    for (int c = 0; c < CW; c++) { acc[c] = fma(xv[c].x, r.x, acc[c]); acc[c] = fma(xv[c].y, r.y, acc[c]); }
  }
#pragma unroll
  for (int c = 0; c < CW; c++) { const double v = wave_sum(acc[c]); if (lane == 0 && c < kc) partial[(int64_t)(c0 + c) * nb + blockIdx.x] = v; }
  self = wave_sum(self);
  if (w == 0 && lane == 0) partial[(int64_t)k * nb + blockIdx.x] = self;
}

// fused kernel with a dedicated writer wave: waves 0..3 only load, wave 4 owns the y store
template <int CW, bool NT>
__global__ __launch_bounds__(320) void k_mvdot5(const double *__restrict__ X, int64_t ld, int k, qarg q, double beta,
                                                double *__restrict__ y, int64_t n, double *__restrict__ partial, int nb) {
  __shared__ double2 sm[2][4][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t npair = n >> 1;
  double2 *y2 = (double2 *)y;
  const int64_t step = (int64_t)gridDim.x * 64;
  int buf = 0;
  if (w < 4) {
    const int c0 = w * CW;
    const int kc = max(0, min(CW, k - c0));
    const double2 *col[CW];
    double qc[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) {
      const int cc = c0 + (c < kc ? c : (kc > 0 ? kc - 1 : -c0));
      col[c] = (const double2 *)(X + (int64_t)cc * ld);
      qc[c] = c < kc ? q.v[c0 + c] : 0.0;
    }
    double acc[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) acc[c] = 0.0;
    for (int64_t p0 = (int64_t)blockIdx.x * 64; p0 < npair; p0 += step, buf ^= 1) {
      const int64_t p = p0 + lane;
      const bool live = p < npair;
      const int64_t pc = live ? p : npair - 1;
      double2 xv[CW];
#pragma unroll
      for (int c = 0; c < CW; c++) xv[c] = ld2<NT>(col[c] + pc);
      const double2 yv = y2[pc];
      double2 a = make_double2(0, 0);
#pragma unroll
      for (int c = 0; c < CW; c++) { a.x = fma(xv[c].x, qc[c], a.x); a.y = fma(xv[c].y, qc[c], a.y); }
      sm[buf][w][lane] = a;
      __syncthreads();
      const double2 s0 = sm[buf][0][lane], s1 = sm[buf][1][lane], s2 = sm[buf][2][lane], s3 = sm[buf][3][lane];
      double2 r;
      r.x = beta * yv.x + ((s0.x + s1.x) + (s2.x + s3.x));
      r.y = beta * yv.y + ((s0.y + s1.y) + (s2.y + s3.y));
      if (!live) { r.x = 0; r.y = 0; }
#pragma unroll
      for (int c = 0; c < CW; c++) { acc[c] = fma(xv[c].x, r.x, acc[c]); acc[c] = fma(xv[c].y, r.y, acc[c]); }
    }
#pragma unroll
    for (int c = 0; c < CW; c++) { const double v = wave_sum(acc[c]); if (lane == 0 && c < kc) partial[(int64_t)(c0 + c) * nb + blockIdx.x] = v; }
  } else {
    double self = 0.0;
    for (int64_t p0 = (int64_t)blockIdx.x * 64; p0 < npair; p0 += step, buf ^= 1) {
      const int64_t p = p0 + lane;
      const bool live = p < npair;
      const double2 yv = y2[live ? p : npair - 1];
      __syncthreads();
      const double2 s0 = sm[buf][0][lane], s1 = sm[buf][1][lane], s2 = sm[buf][2][lane], s3 = sm[buf][3][lane];
      double2 r;
      r.x = beta * yv.x + ((s0.x + s1.x) + (s2.x + s3.x));
      r.y = beta * yv.y + ((s0.y + s1.y) + (s2.y + s3.y));
      if (live) { y2[p] = r; self = fma(r.x, r.x, self); self = fma(r.y, r.y, self); }
    }
    self = wave_sum(self);
    if (lane == 0) partial[(int64_t)k * nb + blockIdx.x] = self;
  }
}

// fused kernel, contiguous row range per block, y' staged in LDS and written T KiB at a time
template <int CW, int T, bool STORE>
__global__ __launch_bounds__(256) void k_mvdot_b(const double *__restrict__ X, int64_t ld, int k, qarg q, double beta,
                                                 double *__restrict__ y, int64_t n, double *__restrict__ partial, int nb) {
  __shared__ double2 sm[2][4][64];
  __shared__ double2 ybuf[T][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = w * CW;
  const int kc = max(0, min(CW, k - c0));
  const double2 *col[CW];
  double qc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const int cc = c0 + (c < kc ? c : (kc > 0 ? kc - 1 : -c0));
    col[c] = (const double2 *)(X + (int64_t)cc * ld);
    qc[c] = c < kc ? q.v[c0 + c] : 0.0;
  }
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  double self = 0.0;
  const int64_t npair = n >> 1;
  double2 *y2 = (double2 *)y;
  const int64_t chunk = (((npair + gridDim.x - 1) / gridDim.x) + 63) / 64 * 64;
  const int64_t pbeg = (int64_t)blockIdx.x * chunk;
  const int64_t pend = min(npair, pbeg + chunk);
  int buf = 0, it = 0;
  for (int64_t p0 = pbeg; p0 < pend; p0 += 64, buf ^= 1, it++) {
    const int64_t p = p0 + lane;
    const bool live = p < pend;
    const int64_t pc = live ? p : pend - 1;
    double2 xv[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) xv[c] = ld2<true>(col[c] + pc);
    double2 yv = y2[pc];
#pragma unroll
    for (int c = 0; c < CW; c++) asm volatile("" : "+v"(xv[c].x), "+v"(xv[c].y));
    double2 a = make_double2(0, 0);
#pragma unroll
    for (int c = 0; c < CW; c++) { a.x = fma(xv[c].x, qc[c], a.x); a.y = fma(xv[c].y, qc[c], a.y); }
    sm[buf][w][lane] = a;
    __syncthreads();
    const double2 s0 = sm[buf][0][lane], s1 = sm[buf][1][lane], s2 = sm[buf][2][lane], s3 = sm[buf][3][lane];
    double2 r;
    r.x = beta * yv.x + ((s0.x + s1.x) + (s2.x + s3.x));
    r.y = beta * yv.y + ((s0.y + s1.y) + (s2.y + s3.y));
    if (!live) { r.x = 0; r.y = 0; }
    if (w == 0) {
      if (T == 1) { if (live && STORE) y2[p] = r; }
      else ybuf[it % T][lane] = r;
      self = fma(r.x, r.x, self); self = fma(r.y, r.y, self);
    }
#pragma unroll
    for (int c = 0; c < CW; c++) { acc[c] = fma(xv[c].x, r.x, acc[c]); acc[c] = fma(xv[c].y, r.y, acc[c]); }
    if (T > 1 && ((it % T) == T - 1 || p0 + 64 >= pend)) {
      __syncthreads();
      const int cnt = (it % T) + 1;
      const int64_t base = p0 - (int64_t)(cnt - 1) * 64;
      if (STORE)
        for (int e = threadIdx.x; e < cnt * 64; e += 256)
          if (base + e < pend) y2[base + e] = ybuf[e >> 6][e & 63];
    }
  }
#pragma unroll
  for (int c = 0; c < CW; c++) { const double v = wave_sum(acc[c]); if (lane == 0 && c < kc) partial[(int64_t)(c0 + c) * nb + blockIdx.x] = v; }
  self = wave_sum(self);
  if (w == 0 && lane == 0) partial[(int64_t)k * nb + blockIdx.x] = self;
}

// fused kernel, one big workgroup per CU: 4 column groups x RG row groups, y' staged in LDS
// and written in bursts of T*RG KiB
template <int CW, int RG, int T>
__global__ __launch_bounds__(256 * RG) void k_mvdot_c(const double *__restrict__ X, int64_t ld, int k, qarg q, double beta,
                                                      double *__restrict__ y, int64_t n, double *__restrict__ partial, int nb) {
  extern __shared__ double2 dyn[];
  double2 (*sm)[RG][4][64] = (double2 (*)[RG][4][64])dyn;                 // [2][RG][4][64]
  double2 *ybuf = dyn + 2 * RG * 4 * 64;                                   // [T][RG*64]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = w & 3, rg = w >> 2;
  const int c0 = cg * CW;
  const int kc = max(0, min(CW, k - c0));
  const double2 *col[CW];
  double qc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const int cc = c0 + (c < kc ? c : (kc > 0 ? kc - 1 : -c0));
    col[c] = (const double2 *)(X + (int64_t)cc * ld);
    qc[c] = c < kc ? q.v[c0 + c] : 0.0;
  }
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  double self = 0.0;
  const int64_t npair = n >> 1;
  double2 *y2 = (double2 *)y;
  constexpr int RW = RG * 64;
  const int64_t chunk = (((npair + gridDim.x - 1) / gridDim.x) + RW - 1) / RW * RW;
  const int64_t pbeg = (int64_t)blockIdx.x * chunk;
  const int64_t pend = min(npair, pbeg + chunk);
  int buf = 0, it = 0;
  for (int64_t p0 = pbeg; p0 < pend; p0 += RW, buf ^= 1, it++) {
    const int64_t p = p0 + rg * 64 + lane;
    const bool live = p < pend;
    const int64_t pc = live ? p : pend - 1;
    double2 xv[CW];
#pragma unroll
    for (int c = 0; c < CW; c++) xv[c] = ld2<true>(col[c] + pc);
    double2 yv = y2[pc];
#pragma unroll
    for (int c = 0; c < CW; c++) asm volatile("" : "+v"(xv[c].x), "+v"(xv[c].y));
    double2 a = make_double2(0, 0);
#pragma unroll
    for (int c = 0; c < CW; c++) { a.x = fma(xv[c].x, qc[c], a.x); a.y = fma(xv[c].y, qc[c], a.y); }
    sm[buf][rg][cg][lane] = a;
    __syncthreads();
    const double2 s0 = sm[buf][rg][0][lane], s1 = sm[buf][rg][1][lane], s2 = sm[buf][rg][2][lane], s3 = sm[buf][rg][3][lane];
    double2 r;
    r.x = beta * yv.x + ((s0.x + s1.x) + (s2.x + s3.x));
    r.y = beta * yv.y + ((s0.y + s1.y) + (s2.y + s3.y));
    if (!live) { r.x = 0; r.y = 0; }
    if (cg == 0) {
      ybuf[(it % T) * RW + rg * 64 + lane] = r;
      self = fma(r.x, r.x, self); self = fma(r.y, r.y, self);
    }
#pragma unroll
    for (int c = 0; c < CW; c++) { acc[c] = fma(xv[c].x, r.x, acc[c]); acc[c] = fma(xv[c].y, r.y, acc[c]); }
    if ((it % T) == T - 1 || p0 + RW >= pend) {
      __syncthreads();
      const int cnt = (it % T) + 1;
      const int64_t base = p0 - (int64_t)(cnt - 1) * RW;
      for (int e = threadIdx.x; e < cnt * RW; e += 256 * RG)
        if (base + e < pend) y2[base + e] = ybuf[e];
    }
  }
  // (partials: one slot per (block,row group))
#pragma unroll
  for (int c = 0; c < CW; c++) { const double v = wave_sum(acc[c]); if (lane == 0 && c < kc) partial[(int64_t)(c0 + c) * nb + blockIdx.x * RG + rg] = v; }
  self = wave_sum(self);
  if (cg == 0 && lane == 0) partial[(int64_t)k * nb + blockIdx.x * RG + rg] = self;
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int64_t n = (argc > 1) ? atoll(argv[1]) : (int64_t)4096 * 4096;
  const int m = 66;
  const int64_t ld = n + ((argc > 2) ? atoll(argv[2]) : 0);
  double *X, *part, *out;
  CK(hipMalloc(&X, sizeof(double) * ld * m));
  CK(hipMalloc(&part, sizeof(double) * 128 * 65536));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(X, 0, sizeof(double) * ld * m));
  // non-trivial data
  {
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int64_t off = 0; off + (int64_t)h.size() <= ld * m; off += h.size() * 7) CK(hipMemcpy(X + off, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  }
  const int reps = 5;
  double *y = X + 65 * ld;
  printf("n=%lld ld=%lld\n", (long long)n, (long long)ld);
  {
    float ms = timeit([&] { CK(hipMemcpyAsync(X + 33 * ld, X, 8 * ld * 32, hipMemcpyDeviceToDevice, 0)); }, reps);
    printf("memcpy D2D 32 cols            : %8.3f ms %8.1f GB/s (r+w)\n", ms, 2.0 * 8 * ld * 32 / ms / 1e6);
  }
  const int gfull[] = {512, 1024, 2048, 4096, 8192, 16384}; const int gshort[] = {2048}; const bool brief = argc > 3;
  std::vector<int> grids(brief ? gshort : gfull, brief ? gshort + 1 : gfull + 6);
  for (int g : grids) {
    float ms = timeit([&] { hipLaunchKernelGGL((k_read<false, 4>), dim3(g), dim3(256), 0, 0, (const double2 *)X, ld * 64 / 2, out); }, reps);
    float ms2 = timeit([&] { hipLaunchKernelGGL((k_read<true, 4>), dim3(g), dim3(256), 0, 0, (const double2 *)X, ld * 64 / 2, out); }, reps);
    float ms3 = timeit([&] { hipLaunchKernelGGL((k_read<false, 8>), dim3(g), dim3(256), 0, 0, (const double2 *)X, ld * 64 / 2, out); }, reps);
    printf("read 64 cols grid %5d       : U4 %8.1f  U4nt %8.1f  U8 %8.1f GB/s\n", g, 8.0 * ld * 64 / ms / 1e6, 8.0 * ld * 64 / ms2 / 1e6, 8.0 * ld * 64 / ms3 / 1e6);
  }
  const int k = 65;
  const double bytes = 8.0 * n * (k + 1);
#define DOT(C, NT, BLK, G) { int nb = G; int nch = (k + C - 1) / C; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_dot<C, NT, BLK>), dim3(nb, nch), dim3(BLK), 0, 0, X, ld, k, y, n, part, nb); }, reps); \
    printf("dot  C=%2d nt=%d blk=%3d grid=%5d : %8.3f ms %8.1f GB/s\n", C, (int)NT, BLK, nb, ms, bytes / ms / 1e6); }
  for (int g : grids) { DOT(16, false, 256, g) }
  for (int g : grids) { DOT(16, true, 256, g) }
  if (!brief) for (int g : {1024, 2048, 4096}) { DOT(8, false, 256, g) DOT(32, false, 256, g) DOT(16, false, 512, g) DOT(13, false, 256, g) DOT(22, false, 256, g) }
  qarg q;
  for (int j = 0; j < 128; j++) q.v[j] = 1e-3;
  const int km = 64;
  const double mbytes = 8.0 * n * (km + 2);
#define MV(U, NT, BLK, G) { float ms = timeit([&] { hipLaunchKernelGGL((k_mv<U, NT, BLK>), dim3(G), dim3(BLK), 0, 0, X, ld, km, q, y, n); }, reps); \
    printf("mv   U=%2d nt=%d blk=%3d grid=%5d : %8.3f ms %8.1f GB/s\n", U, (int)NT, BLK, G, ms, mbytes / ms / 1e6); }
  for (int g : grids) { MV(8, false, 256, g) }
  for (int g : {2048}) { MV(8, true, 256, g) MV(16, false, 256, g) MV(4, false, 256, g) MV(8, false, 512, g) MV(32, false, 256, g) }
  if (!brief) for (int g : {1024, 2048, 4096}) { DOT(13, true, 256, g) }
#define DOTW(CW, NT, NW, G) { int nb = G; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_dotw<CW, NT>), dim3(nb), dim3(64 * NW), 0, 0, X, ld, k, y, n, part, nb); }, reps); \
    printf("dotw CW=%2d nt=%d nw=%2d grid=%5d : %8.3f ms %8.1f GB/s\n", CW, (int)NT, NW, nb, ms, bytes / ms / 1e6); }
  for (int g : {1024, 2048, 4096, 8192}) { DOTW(13, true, 5, g) DOTW(13, false, 5, g) }
  for (int g : {2048, 4096}) { DOTW(11, true, 6, g) DOTW(7, true, 10, g) }
#define DOTWK(CW, NW, G, KK) { int nb = G; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_dotw<CW, true>), dim3(nb), dim3(64 * NW), 0, 0, X, ld, KK, y, n, part, nb); }, reps); \
    printf("dotw k=%2d CW=%2d nw=%2d grid=%5d : %8.3f ms %8.1f GB/s\n", KK, CW, NW, nb, ms, 8.0 * n * (KK + 1) / ms / 1e6); }
#define MVDK(CW, PF, G, KK) { float ms = timeit([&] { hipLaunchKernelGGL((k_mvdot<CW, true, PF>), dim3(G), dim3(256), 0, 0, X, ld, KK, q, 1.0, y, n, part, G); }, reps); \
    printf("mvdot k=%2d CW=%2d pf=%d grid=%5d : %8.3f ms %8.1f GB/s\n", KK, CW, (int)PF, G, ms, 8.0 * n * (KK + 2) / ms / 1e6); }
  if (argc > 4) {
#define MVDD(D, G) { float ms = timeit([&] { hipLaunchKernelGGL((k_mvdot<16, true, false, D>), dim3(G), dim3(256), 0, 0, X, ld, 64, q, 1.0, y, n, part, G); }, reps); \
    printf("mvdot diag=%d grid=%5d : %8.3f ms %8.1f GB/s\n", D, G, ms, 8.0 * n * (64 + 2) / ms / 1e6); }
    { double *yo; CK(hipMalloc(&yo, 8 * n)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_yout), &yo, sizeof(yo))); }
    for (int g : {512, 1024, 2048}) { MVDD(0, g) MVDD(1, g) }
#define MVC(RG, TT, G) { size_t lds = sizeof(double2) * (2 * RG * 4 * 64 + (size_t)TT * RG * 64); \
    CK(hipFuncSetAttribute((const void *)k_mvdot_c<16, RG, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_mvdot_c<16, RG, TT>), dim3(G), dim3(256 * RG), lds, 0, X, ld, 64, q, 1.0, y, n, part, G * RG); }, reps); \
    printf("mvdot_c RG=%d T=%3d lds=%6zu grid=%5d : %8.3f ms %8.1f GB/s\n", RG, TT, lds, G, ms, 8.0 * n * (64 + 2) / ms / 1e6); }
    for (int g : {256, 512}) { MVC(4, 1, g) MVC(4, 8, g) MVC(4, 24, g) MVC(2, 8, g) MVC(2, 32, g) MVC(2, 64, g) }
    MVC(1, 8, 1024) MVC(1, 32, 1024) MVC(3, 32, 256) MVC(3, 40, 256)
#define MVB(TT, ST, G) { float ms = timeit([&] { hipLaunchKernelGGL((k_mvdot_b<16, TT, ST>), dim3(G), dim3(256), 0, 0, X, ld, 64, q, 1.0, y, n, part, G); }, reps); \
    printf("mvdot_b T=%2d store=%d grid=%5d : %8.3f ms %8.1f GB/s\n", TT, (int)ST, G, ms, 8.0 * n * (64 + 2) / ms / 1e6); }
    for (int g : {512, 1024, 2048}) { MVB(1, false, g) MVB(1, true, g) MVB(8, true, g) MVB(32, true, g) }
    for (int g : {512, 768, 1024, 1536, 2048}) { float ms = timeit([&] { hipLaunchKernelGGL((k_mvdot5<16, true>), dim3(g), dim3(320), 0, 0, X, ld, 64, q, 1.0, y, n, part, g); }, reps);
      printf("mvdot5 grid=%5d : %8.3f ms %8.1f GB/s\n", g, ms, 8.0 * n * (64 + 2) / ms / 1e6); }
    for (int g : {256, 384, 512, 768, 1024, 1280, 1536, 1792, 2048, 3072}) { DOTWK(13, 5, g, 65) }
    for (int g : {512, 768, 1024, 1280, 1536, 1792, 2048, 3072}) { DOTWK(13, 4, g, 49) }
    for (int g : {512, 768, 1024, 1280, 1536, 2048, 2560, 3072}) { DOTWK(12, 3, g, 34) }
    for (int g : {512, 768, 1024, 1280, 1536, 2048, 2560, 3072}) { DOTWK(16, 3, g, 34) }
    for (int g : {512, 768, 1024, 1280, 1536, 2048, 3072}) { MVDK(16, false, g, 64) }
    for (int g : {512, 768, 1024, 1280, 1536, 2048, 3072}) { MVDK(16, true, g, 64) }
    for (int g : {512, 768, 1024, 1280, 1536, 2048, 3072}) { MVDK(12, false, g, 48) }
    for (int g : {512, 768, 1024, 1280, 1536, 2048, 3072}) { MVDK(9, false, g, 33) }
    for (int g : {512, 768, 1024, 1280, 1536, 2048, 3072}) { MVDK(9, true, g, 33) }
    return 0;
  }
  {
    // fused kernel: correctness against k_mv followed by k_dot, then timing
    const int kf = 64;
    std::vector<double> hq(kf);
    for (int j = 0; j < kf; j++) hq[j] = 1e-3 * ((j % 7) - 3);
    qarg qf; for (int j = 0; j < 128; j++) qf.v[j] = j < kf ? hq[j] : 0.0;
    double *ysave, *yref;
    CK(hipMalloc(&ysave, 8 * n)); CK(hipMalloc(&yref, 8 * n));
    CK(hipMemcpy(ysave, y, 8 * n, hipMemcpyDeviceToDevice));
    const int nb = 2048;
    auto reduce = [&](int cnt, std::vector<double> &out) {
      std::vector<double> hp((size_t)cnt * nb);
      CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
      out.assign(cnt, 0.0);
      for (int c = 0; c < cnt; c++) for (int b = 0; b < nb; b++) out[c] += hp[(size_t)c * nb + b];
    };
    hipLaunchKernelGGL((k_mv<8, false, 256>), dim3(nb), dim3(256), 0, 0, X, ld, kf, qf, y, n);
    CK(hipMemcpy(yref, y, 8 * n, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL((k_dot<16, false, 256>), dim3(nb, 5), dim3(256), 0, 0, X, ld, kf + 1, y, n, part, nb);
    CK(hipDeviceSynchronize());
    std::vector<double> ref, got;
    reduce(kf + 1, ref);   // column kf+? : y is column 65, not 64 -> self dot compared separately below
    auto check = [&](const char *name) {
      CK(hipDeviceSynchronize());
      reduce(kf + 1, got);
      std::vector<double> hy(1 << 16), hr(1 << 16);
      CK(hipMemcpy(hy.data(), y + n - (1 << 16), 8 << 16, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), yref + n - (1 << 16), 8 << 16, hipMemcpyDeviceToHost));
      double dy = 0, dc = 0, ss = 0;
      for (size_t i = 0; i < hy.size(); i++) dy = fmax(dy, fabs(hy[i] - hr[i]));
      for (int c = 0; c < kf; c++) dc = fmax(dc, fabs(got[c] - ref[c]) / (fabs(ref[c]) + 1e-300));
      printf("check %s: max|y-yref| %.3e  max rel dot err %.3e  self %.15e\n", name, dy, dc, got[kf]);
    };
#define MVD(CW, NT, PF, G) { CK(hipMemcpy(y, ysave, 8 * n, hipMemcpyDeviceToDevice)); \
    hipLaunchKernelGGL((k_mvdot<CW, NT, PF>), dim3(nb), dim3(256), 0, 0, X, ld, kf, qf, 1.0, y, n, part, nb); \
    check("mvdot CW=" #CW " nt=" #NT " pf=" #PF); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_mvdot<CW, NT, PF>), dim3(G), dim3(256), 0, 0, X, ld, kf, qf, 1.0, y, n, part, G); }, reps); \
    printf("mvdot CW=%2d nt=%d pf=%d grid=%5d : %8.3f ms %8.1f GB/s (8n(k+2))\n", CW, (int)NT, (int)PF, G, ms, 8.0 * n * (kf + 2) / ms / 1e6); }
    {
      // reference self dot
      hipLaunchKernelGGL((k_dot<16, false, 256>), dim3(nb, 1), dim3(256), 0, 0, yref, ld, 1, yref, n, part, nb);
      CK(hipDeviceSynchronize());
      std::vector<double> sref; reduce(1, sref);
      printf("reference self dot %.15e\n", sref[0]);
    }
    MVD(16, true, false, 2048) MVD(16, true, true, 2048) MVD(16, false, false, 2048)
    MVD(16, true, false, 1024) MVD(16, true, false, 4096) MVD(16, true, true, 1024) MVD(16, true, true, 4096) MVD(16, true, false, 8192)
  }
  return 0;
}
