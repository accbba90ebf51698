/* BV kernels and their C ABI: the op table of the reference's BVSVEC back-end
 * (src/sys/classes/bv/impls/svec/svec.c, CPU kernels in
 * src/sys/classes/bv/interface/bvblas.c) as hand-written gfx950 kernels.
 *
 * All kernels are HBM-bandwidth bound.  The basis is column-major with a padded
 * leading dimension, so a wave reads 64 consecutive row pairs of one column as
 * one 1 KiB coalesced request (16 B per lane); tall-skinny products keep the
 * per-column accumulators in registers and stream every column exactly once. */
#include <mutex>

#include "ksg_internal.h"

/* ------------------------------------------------------------------------ */
/* wave / block reductions                                                    */
/* ------------------------------------------------------------------------ */
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

/* Sum (or max) of the nb per-workgroup partials of column blockIdx.x, always in
 * the same order: results do not depend on scheduling. */
template <int OP>
__global__ __launch_bounds__(KSG_BLOCK) void k_reduce_partials(const double *__restrict__ partial,
                                                                int nb, double *__restrict__ out) {
  __shared__ double sm[KSG_BLOCK / 64];
  const double *p = partial + (int64_t)blockIdx.x * nb;
  double v = 0.0;
  for (int b = threadIdx.x; b < nb; b += KSG_BLOCK) v = OP ? fmax(v, p[b]) : v + p[b];
  v = OP ? wave_max(v) : wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = sm[0];
#pragma unroll
    for (int w = 1; w < KSG_BLOCK / 64; w++) s = OP ? fmax(s, sm[w]) : s + sm[w];
    out[blockIdx.x] = s;
  }
}

int ksg_launch_reduce_sum(ksg_ctx c, const double *partial, int nb, int ncols, double *d_out) {
  hipLaunchKernelGGL((k_reduce_partials<0>), dim3(ncols), dim3(KSG_BLOCK), 0, c->stream, partial, nb, d_out);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* ------------------------------------------------------------------------ */
/* streaming loads                                                            */
/* ------------------------------------------------------------------------ */
/* Basis columns are read once per kernel and are far larger than the caches:
 * loading them with the non-temporal hint keeps them from displacing the lines
 * that are reused (y, partial sums) and measurably raises the sustained HBM
 * read rate. */
__device__ __forceinline__ double2 ld_stream(const double2 *p) {
  double2 r;
  r.x = __builtin_nontemporal_load(&p->x);
  r.y = __builtin_nontemporal_load(&p->y);
  return r;
}
__device__ __forceinline__ double ld_stream(const double *p) { return __builtin_nontemporal_load(p); }
/* Scheduling fence for a batch of independent loads: placed on every loaded value
 * after the batch, it keeps the compiler from interleaving each load with its use
 * (which would leave one request in flight per lane instead of the whole batch). */
__device__ __forceinline__ void ld_fence(double2 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void ld_fence(double &v) { asm volatile("" : "+v"(v)); }

template <int VEC> struct ksg_pack;
template <> struct ksg_pack<2> { typedef double2 type; };
template <> struct ksg_pack<1> { typedef double type; };
__device__ __forceinline__ double dot_acc(double2 x, double2 y, double a) { return fma(x.y, y.y, fma(x.x, y.x, a)); }
__device__ __forceinline__ double dot_acc(double x, double y, double a) { return fma(x, y, a); }
__device__ __forceinline__ void axpy_acc(double2 x, double q, double2 &a) { a.x = fma(x.x, q, a.x); a.y = fma(x.y, q, a.y); }
__device__ __forceinline__ void axpy_acc(double x, double q, double &a) { a = fma(x, q, a); }

/* ------------------------------------------------------------------------ */
/* dotvec: q = X^T y  (tall-skinny GEMV^T)                                    */
/* ------------------------------------------------------------------------ */
/* A workgroup is G x R waves: wave (g, r) owns the CW columns of column group g
 * and the r-th of the R blocks of 64 row pairs the workgroup walks per step, so
 * the G waves that share rows read y from HBM once (the others hit in cache) and
 * every basis element is loaded exactly once, 16 B per lane, CW independent
 * loads in flight per lane.  The grid is small on purpose (a few waves per SIMD,
 * all resident at once): read streams this wide sustain more bandwidth with
 * fewer, longer-lived waves.  partial[col * nb + row block] holds one value per
 * wave; k_reduce_partials adds them in a fixed order.  VEC = 2 needs 16-byte
 * aligned columns; VEC = 1 is the fallback for unaligned vectors. */
template <int CW, int NC, int VEC>
__device__ __forceinline__ void dotw_rows(const typename ksg_pack<VEC>::type *const (&col)[CW],
                                          const typename ksg_pack<VEC>::type *__restrict__ y, int64_t p,
                                          int64_t npack, int64_t stride, double (&acc)[CW]) {
  typedef typename ksg_pack<VEC>::type T;
  for (; p < npack; p += stride) {
    const T yv = y[p];
    T xv[NC > 0 ? NC : 1];
#pragma unroll
    for (int c = 0; c < NC; c++) xv[c] = ld_stream(col[c] + p);
#pragma unroll
    for (int c = 0; c < NC; c++) ld_fence(xv[c]);
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = dot_acc(xv[c], yv, acc[c]);
  }
}

/* The k columns are dealt to NG column groups as evenly as possible: the first
 * NG-d groups take CW columns, the last d take CW-1 (k = NG*CW - d, 0 <= d < NG), so
 * a wave runs one of two fully unrolled loop bodies and never issues a load it
 * does not need. */
__device__ __forceinline__ void group_columns(int gi, int k, int CW, int NG, int &c0, int &kc) {
  const int d = NG * CW - k;
  const int nfull = NG - d;
  c0 = gi * CW - max(0, gi - nfull);
  kc = gi < nfull ? CW : CW - 1;
  if (gi >= NG) kc = 0;
}

template <int CW, int VEC>
__global__ void k_dotw(const double *__restrict__ X, int64_t ld, int k, const double *__restrict__ y, int64_t n,
                       double *__restrict__ partial, int nb, int G, int NG) {
  typedef typename ksg_pack<VEC>::type T;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int R = (blockDim.x >> 6) / G;
  const int g = w % G, r = w / G;
  int c0, kc;                                   /* wave-uniform */
  group_columns(blockIdx.y * G + g, k, CW, NG, c0, kc);
  if (kc <= 0) return;
  const T *col[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) col[c] = reinterpret_cast<const T *>(X + (int64_t)(c0 + (c < kc ? c : 0)) * ld);
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  const int64_t rb = (int64_t)blockIdx.x * R + r;
  const int64_t stride = (int64_t)gridDim.x * R * 64;
  const int64_t npack = (VEC == 2) ? (n >> 1) : n;
  if (kc == CW) dotw_rows<CW, CW, VEC>(col, reinterpret_cast<const T *>(y), rb * 64 + lane, npack, stride, acc);
  else          dotw_rows<CW, CW - 1, VEC>(col, reinterpret_cast<const T *>(y), rb * 64 + lane, npack, stride, acc);
  if (VEC == 2 && (n & 1) && rb == 0 && lane == 0) {
    const double yv = y[n - 1];
#pragma unroll
    for (int c = 0; c < CW; c++)
      if (c < kc) acc[c] = fma(X[(int64_t)(c0 + c) * ld + n - 1], yv, acc[c]);
  }
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const double v = wave_sum(acc[c]);
    if (lane == 0 && c < kc) partial[(int64_t)(c0 + c) * nb + rb] = v;
  }
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

template <int VEC>
static void launch_dotw(int CW, dim3 grid, dim3 block, hipStream_t st, const double *X, int64_t ld, int k,
                        const double *y, int64_t n, double *partial, int nb, int G, int NG) {
#define KSG_DOTW_CASE(C) case C: hipLaunchKernelGGL((k_dotw<C, VEC>), grid, block, 0, st, X, ld, k, y, n, partial, nb, G, NG); break;
  switch (CW) {
    KSG_DOTW_CASE(1) KSG_DOTW_CASE(2) KSG_DOTW_CASE(3) KSG_DOTW_CASE(4) KSG_DOTW_CASE(5) KSG_DOTW_CASE(6)
    KSG_DOTW_CASE(7) KSG_DOTW_CASE(8) KSG_DOTW_CASE(9) KSG_DOTW_CASE(10) KSG_DOTW_CASE(11) KSG_DOTW_CASE(12)
    KSG_DOTW_CASE(13) KSG_DOTW_CASE(14) KSG_DOTW_CASE(15)
    default: hipLaunchKernelGGL((k_dotw<16, VEC>), grid, block, 0, st, X, ld, k, y, n, partial, nb, G, NG); break;
  }
#undef KSG_DOTW_CASE
}

/* Launch geometry of the reduction: column groups of at most 16 columns, evenly filled, at
 * most 16 groups per workgroup; nb partial sums per column. */
struct ksg_dot_geom { bool vec2; int CW, G, R, ny, ngroups, nb; unsigned gx; };

static ksg_dot_geom dot_geometry(ksg_ctx c, const double *X, int64_t ld, int k, const double *y, int64_t n) {
  ksg_dot_geom g;
  g.vec2 = aligned16(X) && aligned16(y) && (ld % 2 == 0);
  g.ngroups = (k + 15) / 16;
  g.CW = (k + g.ngroups - 1) / g.ngroups;
  g.G = g.ngroups < 16 ? g.ngroups : 16;
  g.R = g.G == 1 ? 4 : (g.G == 2 ? 2 : 1);
  g.ny = (g.ngroups + g.G - 1) / g.G;
  const int64_t nwork = ((g.vec2 ? (n + 1) / 2 : n) + 63) / 64;          /* blocks of 64 lanes */
  int64_t gx = (nwork + g.R - 1) / g.R;
  if (gx > c->dot_grid) gx = c->dot_grid;
  g.gx = (unsigned)gx;
  g.nb = (int)gx * g.R;
  return g;
}

/* d_out[i] = sum_r X[r + i*ld] * y[r], i < k (local rows only); `partial` holds k*g.nb doubles,
 * every one of which the first kernel writes (nothing has to be cleared beforehand). */
static int launch_dotvec_geom(ksg_ctx c, const ksg_dot_geom &g, const double *X, int64_t ld, int k, const double *y,
                              int64_t n, double *partial, double *d_out) {
  if (g.vec2) launch_dotw<2>(g.CW, dim3(g.gx, g.ny), dim3(64 * g.G * g.R), c->stream, X, ld, k, y, n, partial, g.nb, g.G, g.ngroups);
  else        launch_dotw<1>(g.CW, dim3(g.gx, g.ny), dim3(64 * g.G * g.R), c->stream, X, ld, k, y, n, partial, g.nb, g.G, g.ngroups);
  hipLaunchKernelGGL((k_reduce_partials<0>), dim3(k), dim3(KSG_BLOCK), 0, c->stream, partial, g.nb, d_out);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

int ksg_launch_dotvec(ksg_ctx c, const double *X, int64_t ld, int k, const double *y, int64_t n,
                      double *d_out) {
  if (k <= 0) return KSG_OK;
  if (n <= 0) { KSG_HIP(hipMemsetAsync(d_out, 0, (size_t)k * sizeof(double), c->stream)); return KSG_OK; }
  const ksg_dot_geom g = dot_geometry(c, X, ld, k, y, n);
  KSG_CHK(ksg_ensure_partial(c, (size_t)k * g.nb));
  return launch_dotvec_geom(c, g, X, ld, k, y, n, c->d_partial, d_out);
}

/* ------------------------------------------------------------------------ */
/* multvec: y = beta*y + X q  (tall-skinny GEMV), q pre-scaled by alpha,       */
/* optionally fused with the next Gram-Schmidt reduction                      */
/* ------------------------------------------------------------------------ */
struct ksg_qarg { double v[KSG_QMAX]; };

/* NW waves per workgroup; wave w owns columns [w*CW, w*CW+CW) of the k <= NW*CW
 * active ones and all waves walk the same 64 row pairs per step.  Each wave forms
 * its partial row sums from registers, the partials meet in LDS (double
 * buffered: one barrier per step) and every wave rebuilds the updated row
 * y' = beta*y + X q from them in the same fixed order; wave 0 stores it.  With
 * `dots` set the wave then multiplies the columns it still holds in registers by
 * y': when the kernel ends, partial[] holds the pieces of X^T y' (k values) and of
 * y'^T y' (slot k) -- the coefficients the next classical Gram-Schmidt pass would
 * otherwise re-read the whole basis for.  With `store` clear y' is not written
 * (only its products are wanted for now).  TWO applies two updates in one pass,
 * y'' = beta2*(beta*y + X q) + X q2, each rounded exactly as two separate
 * launches would round it.  Requires 16-byte aligned columns. */
template <int CW, int NC, int NW, bool TWO>
__device__ __forceinline__ void mvd_rows(const double2 *const (&col)[CW], const double (&qc)[CW],
                                         const double (&qd)[TWO ? CW : 1], double beta, double beta2,
                                         double post, double2 *__restrict__ y2, int64_t npair, int w, int lane, int dots, int store,
                                         double2 (&sm)[2][NW][64], double2 (&sm2)[TWO ? 2 : 1][TWO ? NW : 1][TWO ? 64 : 1],
                                         double (&acc)[CW], double &self) {
  const int64_t step = (int64_t)gridDim.x * 64;
  int buf = 0;
  int sw = 0;                                       /* the wave that stores this step's rows */
  /* the trip count is the same for every thread of the workgroup: the barrier inside is safe */
  for (int64_t p0 = (int64_t)blockIdx.x * 64; p0 < npair; p0 += step, buf ^= 1, sw = (sw + 1) & (NW - 1)) {
    const int64_t p = p0 + lane;
    const bool live = p < npair;
    const int64_t pc = live ? p : npair - 1;
    double2 xv[NC > 0 ? NC : 1];
#pragma unroll
    for (int c = 0; c < NC; c++) xv[c] = ld_stream(col[c] + pc);
    double2 yv = make_double2(0.0, 0.0);
    if (beta != 0.0) yv = y2[pc];
#pragma unroll
    for (int c = 0; c < NC; c++) ld_fence(xv[c]);
    double2 a = make_double2(0.0, 0.0);
#pragma unroll
    for (int c = 0; c < NC; c++) axpy_acc(xv[c], qc[c], a);
    sm[buf][w][lane] = a;
    if (TWO) {
      double2 a2 = make_double2(0.0, 0.0);
#pragma unroll
      for (int c = 0; c < NC; c++) axpy_acc(xv[c], qd[c], a2);
      sm2[buf][w][lane] = a2;
    }
    __syncthreads();
    double2 s = sm[buf][0][lane];
#pragma unroll
    for (int ww = 1; ww < NW; ww++) { const double2 t = sm[buf][ww][lane]; s.x += t.x; s.y += t.y; }
    double2 r;
    r.x = fma(beta, yv.x, s.x);
    r.y = fma(beta, yv.y, s.y);
    if (TWO) {
      double2 s2 = sm2[buf][0][lane];
#pragma unroll
      for (int ww = 1; ww < NW; ww++) { const double2 t = sm2[buf][ww][lane]; s2.x += t.x; s2.y += t.y; }
      r.x = fma(beta2, r.x, s2.x);
      r.y = fma(beta2, r.y, s2.y);
    }
    if (post != 1.0) { r.x *= post; r.y *= post; }
    if (!live) r = make_double2(0.0, 0.0);
    if (w == sw && live && store) y2[p] = r;
    if (dots) {
      if (w == 0) self = dot_acc(r, r, self);
#pragma unroll
      for (int c = 0; c < NC; c++) acc[c] = dot_acc(xv[c], r, acc[c]);
    }
  }
}

template <int CW, int NW, bool TWO>
__global__ __launch_bounds__(64 * NW) void k_multvec_dot(const double *__restrict__ X, int64_t ld, int k, ksg_qarg q,
                                                          double beta, ksg_qarg q2, double beta2, double post,
                                                          double *__restrict__ y, int64_t n,
                                                          double *__restrict__ partial, int nb, int dots, int store) {
  __shared__ double2 sm[2][NW][64];
  __shared__ double2 sm2[TWO ? 2 : 1][TWO ? NW : 1][TWO ? 64 : 1];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int c0, kc;                                   /* wave-uniform; kc may be 0 */
  group_columns(w, k, CW, NW, c0, kc);
  const double2 *col[CW];
  double qc[CW], qd[TWO ? CW : 1];
#pragma unroll
  for (int c = 0; c < CW; c++) {
    col[c] = reinterpret_cast<const double2 *>(X + (int64_t)(c < kc ? c0 + c : 0) * ld);
    qc[c] = c < kc ? q.v[c0 + c] : 0.0;
    if (TWO) qd[c] = c < kc ? q2.v[c0 + c] : 0.0;
  }
  double acc[CW];
#pragma unroll
  for (int c = 0; c < CW; c++) acc[c] = 0.0;
  double self = 0.0;
  double2 *y2 = reinterpret_cast<double2 *>(y);
  if (kc == CW) mvd_rows<CW, CW, NW, TWO>(col, qc, qd, beta, beta2, post, y2, n >> 1, w, lane, dots, store, sm, sm2, acc, self);
  else          mvd_rows<CW, CW - 1, NW, TWO>(col, qc, qd, beta, beta2, post, y2, n >> 1, w, lane, dots, store, sm, sm2, acc, self);
  if ((n & 1) && blockIdx.x == 0) {               /* last row of an odd-length vector */
    const int64_t rl = n - 1;
    double t = (beta != 0.0) ? beta * y[rl] : 0.0;
    for (int j = 0; j < k; j++) t = fma(X[(int64_t)j * ld + rl], q.v[j], t);
    if (TWO) {
      t *= beta2;
      for (int j = 0; j < k; j++) t = fma(X[(int64_t)j * ld + rl], q2.v[j], t);
    }
    if (post != 1.0) t *= post;
    __syncthreads();                              /* every thread has read y[rl] */
    if (threadIdx.x == 0) { if (store) y[rl] = t; self = fma(t, t, self); }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < CW; c++)
        if (c < kc) acc[c] = fma(X[(int64_t)(c0 + c) * ld + rl], t, acc[c]);
    }
  }
  if (dots) {
#pragma unroll
    for (int c = 0; c < CW; c++) {
      const double v = wave_sum(acc[c]);
      if (lane == 0 && c < kc) partial[(int64_t)(c0 + c) * nb + blockIdx.x] = v;
    }
    if (w == 0) {
      self = wave_sum(self);
      if (lane == 0) partial[(int64_t)k * nb + blockIdx.x] = self;
    }
  }
}

/* fallback for vectors that are not 16-byte aligned */
__global__ __launch_bounds__(KSG_BLOCK) void k_multvec_scalar(const double *__restrict__ X, int64_t ld, int k, ksg_qarg q,
                                                               double beta, double *__restrict__ y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride) {
    double a = (beta != 0.0) ? beta * y[p] : 0.0;
    int j = 0;
    for (; j + 8 <= k; j += 8) {
      double xv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) xv[u] = ld_stream(X + (int64_t)(j + u) * ld + p);
#pragma unroll
      for (int u = 0; u < 8; u++) a = fma(xv[u], q.v[j + u], a);
    }
    for (; j < k; j++) a = fma(X[(int64_t)j * ld + p], q.v[j], a);
    y[p] = a;
  }
}

template <int NW, bool TWO>
static void launch_mvd(int CW, int nb, hipStream_t st, const double *X, int64_t ld, int k, const ksg_qarg &q, double beta,
                       const ksg_qarg &q2, double beta2, double post, double *y, int64_t n, double *partial, int dots, int store) {
#define KSG_MVD_CASE(C) case C: hipLaunchKernelGGL((k_multvec_dot<C, NW, TWO>), dim3(nb), dim3(64 * NW), 0, st, X, ld, k, q, beta, q2, beta2, post, y, n, partial, nb, dots, store); break;
  switch (CW) {
    KSG_MVD_CASE(1) KSG_MVD_CASE(2) KSG_MVD_CASE(3) KSG_MVD_CASE(4) KSG_MVD_CASE(5) KSG_MVD_CASE(6)
    KSG_MVD_CASE(7) KSG_MVD_CASE(8) KSG_MVD_CASE(9) KSG_MVD_CASE(10) KSG_MVD_CASE(11) KSG_MVD_CASE(12)
    KSG_MVD_CASE(13) KSG_MVD_CASE(14) KSG_MVD_CASE(15)
    default: hipLaunchKernelGGL((k_multvec_dot<16, NW, TWO>), dim3(nb), dim3(64 * NW), 0, st, X, ld, k, q, beta, q2, beta2, post, y, n, partial, nb, dots, store); break;
  }
#undef KSG_MVD_CASE
}

static inline bool mv_vec2(const double *X, int64_t ld, const double *y) { return aligned16(X) && aligned16(y) && (ld % 2 == 0); }
static inline int mv_grid(ksg_ctx c, int64_t n) {
  int64_t nb = ((n + 1) / 2 + 63) / 64;
  return (int)(nb > c->mv_grid ? c->mv_grid : nb);
}

/* One launch of the fused kernel on k <= 64 aligned columns with pre-scaled coefficients:
 * y' = b1*y + X q1, then (q2 != NULL) y'' = b2*y' + X q2; dots -> d_dots (k+1 values). */
static int launch_fused(ksg_ctx c, const double *X, int64_t ld, int k, const ksg_qarg &q1, double b1,
                        const ksg_qarg *q2, double b2, double *y, int64_t n, double *d_dots, int store,
                        double post = 1.0) {
  const int nb = mv_grid(c, n);
  if (d_dots) KSG_CHK(ksg_ensure_partial(c, (size_t)(k + 1) * nb));
  if (q2) launch_mvd<4, true>((k + 3) / 4, nb, c->stream, X, ld, k, q1, b1, *q2, b2, post, y, n, c->d_partial, d_dots ? 1 : 0, store);
  else    launch_mvd<4, false>((k + 3) / 4, nb, c->stream, X, ld, k, q1, b1, q1, 1.0, post, y, n, c->d_partial, d_dots ? 1 : 0, store);
  if (d_dots) hipLaunchKernelGGL((k_reduce_partials<0>), dim3(k + 1), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nb, d_dots);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* y = beta*y + alpha * X(:,0:k) q.  With d_dots != NULL (and k <= KSG_QMAX, aligned
 * vectors) the launch also leaves X^T y' and y'^T y' (k+1 values) in d_dots and
 * *fused is set; otherwise *fused is cleared. */
static int launch_multvec(ksg_ctx c, const double *X, int64_t ld, int k, const double *q_host,
                          double alpha, double beta, double *y, int64_t n, double *d_dots, int *fused) {
  if (fused) *fused = 0;
  if (n <= 0) return KSG_OK;
  const bool vec2 = mv_vec2(X, ld, y);
  const bool dots = d_dots && vec2 && k <= KSG_QMAX;
  const int nb = vec2 ? mv_grid(c, n) : 0;
  if (dots) KSG_CHK(ksg_ensure_partial(c, (size_t)(k + 1) * nb));
  int done = 0;
  do {   /* at most KSG_QMAX coefficients travel with one launch */
    const int kk = (k - done < KSG_QMAX) ? k - done : KSG_QMAX;
    ksg_qarg q;
    for (int j = 0; j < kk; j++) q.v[j] = alpha * q_host[done + j];
    const double b = done ? 1.0 : beta;
    const double *Xc = X + (int64_t)done * ld;
    if (vec2) {
      if (kk <= 64) launch_mvd<4, false>((kk + 3) / 4, nb, c->stream, Xc, ld, kk, q, b, q, 1.0, 1.0, y, n, c->d_partial, dots ? 1 : 0, 1);
      else          launch_mvd<8, false>((kk + 7) / 8, nb, c->stream, Xc, ld, kk, q, b, q, 1.0, 1.0, y, n, c->d_partial, dots ? 1 : 0, 1);
    } else {
      hipLaunchKernelGGL(k_multvec_scalar, dim3(ksg_grid_for(n)), dim3(KSG_BLOCK), 0, c->stream, Xc, ld, kk, q, b, y, n);
    }
    done += kk;
  } while (done < k);
  if (dots) {
    hipLaunchKernelGGL((k_reduce_partials<0>), dim3(k + 1), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nb, d_dots);
    if (fused) *fused = 1;
  }
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* ------------------------------------------------------------------------ */
/* multinplace: V(:,s:e) = V(:,l:k) * B,  B = K x ne                           */
/* ------------------------------------------------------------------------ */
/* One thread owns R rows and all ne <= E output columns of them: it streams its
 * K input values once, multiplies by the coefficient panel held in LDS (read as
 * wave-uniform broadcasts) and only then stores, so the update is in place with
 * no scratch copy of the basis.  Bq is the K x E panel, row-major, zero padded.  Vin and Vout may
 * be the same columns: neither is declared restrict. */
#define KSG_MIP_LDS 4096   /* doubles of the panel resident in LDS at a time */
/* 1: in the two-rows-per-thread body the coefficients come straight from the device copy of the
 * panel with wave-uniform (scalar) loads and enter the multiply-adds as scalar operands; the
 * LDS copy serves the remainder loops.  With LDS broadcasts every pair of multiply-adds waits
 * for a read of the shared LDS pipe (64x32: 2.75 ms against 2.56 ms). */
#ifndef KSG_MIP_SCALAR
#define KSG_MIP_SCALAR 1
#endif
/* columns of the input requested together by a thread (64x32: 3.12 ms with 4, 2.72 ms with 8 or 16) */
#ifndef KSG_MIP_BATCH
#define KSG_MIP_BATCH 16
#endif

template <int E, int R>
__global__ __launch_bounds__(KSG_BLOCK) void k_multinplace(const double *Vin, int64_t ld, int K,
                                                            const double *__restrict__ Bq,
                                                            double *Vout, int64_t ldo, int ne,
                                                            int64_t n, double beta) {
  __shared__ double Bs[KSG_MIP_LDS];
  constexpr int KT = KSG_MIP_LDS / E;            /* panel rows per LDS fill */
  const int64_t ngroups = (n + R - 1) / R;
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  /* every thread of the block must take part in the LDS fills: loop bound is per block */
  for (int64_t g0 = (int64_t)blockIdx.x * KSG_BLOCK; g0 < ngroups; g0 += stride) {
    const int64_t g = g0 + threadIdx.x;
    const int64_t r0 = g * R;
    const bool live = g < ngroups;
    double acc[R][E];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int e = 0; e < E; e++) acc[r][e] = 0.0;
    for (int kt = 0; kt < K; kt += KT) {
      const int kn = min(KT, K - kt);
      __syncthreads();
      for (int t = threadIdx.x; t < kn * E; t += KSG_BLOCK) Bs[t] = Bq[(int64_t)kt * E + t];
      __syncthreads();
      if (live) {
        if (R == 2 && r0 + 1 < n) {
          const double2 *xp = reinterpret_cast<const double2 *>(Vin + (int64_t)kt * ld + r0);
          const int64_t ld2 = ld >> 1;
          int i = 0;
          /* U columns requested together: with 2 waves per SIMD (the accumulators take the
           * registers) the bytes in flight per wave decide the rate */
          constexpr int U = KSG_MIP_BATCH;
          for (; i + U <= kn; i += U) {
            double2 xv[U];
#pragma unroll
            for (int u = 0; u < U; u++) xv[u] = xp[(int64_t)(i + u) * ld2];
#pragma unroll
            for (int u = 0; u < U; u++) ld_fence(xv[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
              for (int e = 0; e < E; e++) {
                const double b = KSG_MIP_SCALAR ? Bq[(int64_t)(kt + i + u) * E + e] : Bs[(i + u) * E + e];
                acc[0][e] = fma(xv[u].x, b, acc[0][e]);
                acc[R - 1][e] = fma(xv[u].y, b, acc[R - 1][e]);
              }
            }
          }
          for (; i < kn; i++) {
            const double2 xv = xp[(int64_t)i * ld2];
#pragma unroll
            for (int e = 0; e < E; e++) {
              const double b = Bs[i * E + e];
              acc[0][e] = fma(xv.x, b, acc[0][e]);
              acc[R - 1][e] = fma(xv.y, b, acc[R - 1][e]);
            }
          }
        } else {
          const double *xp = Vin + (int64_t)kt * ld + r0;
          constexpr int U = KSG_MIP_BATCH;
          int i = 0;
          for (; i + U <= kn; i += U) {
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; u++) xv[u] = xp[(int64_t)(i + u) * ld];
#pragma unroll
            for (int u = 0; u < U; u++) ld_fence(xv[u]);
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
              for (int e = 0; e < E; e++) acc[0][e] = fma(xv[u], Bs[(i + u) * E + e], acc[0][e]);
          }
          for (; i < kn; i++) {
            const double xv = xp[(int64_t)i * ld];
#pragma unroll
            for (int e = 0; e < E; e++) acc[0][e] = fma(xv, Bs[i * E + e], acc[0][e]);
          }
        }
      }
    }
    if (live) {
      if (R == 2 && r0 + 1 < n) {
#pragma unroll
        for (int e = 0; e < E; e++)
          if (e < ne) {
            double2 *o = reinterpret_cast<double2 *>(Vout + (int64_t)e * ldo + r0);
            double2 v = make_double2(acc[0][e], acc[R - 1][e]);
            if (beta != 0.0) { const double2 old = *o; v.x = fma(beta, old.x, v.x); v.y = fma(beta, old.y, v.y); }
            *o = v;
          }
      } else {
#pragma unroll
        for (int e = 0; e < E; e++)
          if (e < ne) {
            double *o = Vout + (int64_t)e * ldo + r0;
            *o = (beta != 0.0) ? fma(beta, *o, acc[0][e]) : acc[0][e];
          }
      }
    }
  }
}

template <int E, int R>
static void launch_mip(ksg_ctx c, const double *Vin, int64_t ld, int K, const double *Bq, double *Vout,
                       int64_t ldo, int ne, int64_t n, double beta) {
  const int nb = ksg_grid_for((n + R - 1) / R);
  hipLaunchKernelGGL((k_multinplace<E, R>), dim3(nb), dim3(KSG_BLOCK), 0, c->stream, Vin, ld, K, Bq, Vout, ldo, ne, n, beta);
}

/* out(:,0:ne) = Vin(:,0:K) * B, B given on the device as K x E row-major panel. */
static int mip_panel(ksg_ctx c, const double *Vin, int64_t ld, int K, const double *Bq, int E, double *Vout,
                     int64_t ldo, int ne, int64_t n, double beta = 0.0) {
  const bool two = aligned16(Vin) && aligned16(Vout) && (ld % 2 == 0) && (ldo % 2 == 0);
  switch (E) {
    case 8:  if (two) launch_mip<8, 2>(c, Vin, ld, K, Bq, Vout, ldo, ne, n, beta);  else launch_mip<8, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n, beta);  break;
    case 16: if (two) launch_mip<16, 2>(c, Vin, ld, K, Bq, Vout, ldo, ne, n, beta); else launch_mip<16, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n, beta); break;
    case 32: if (two) launch_mip<32, 2>(c, Vin, ld, K, Bq, Vout, ldo, ne, n, beta); else launch_mip<32, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n, beta); break;
    default: launch_mip<64, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n, beta); break;
  }
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

static inline int mip_width(int ne) { return ne <= 8 ? 8 : ne <= 16 ? 16 : ne <= 32 ? 32 : 64; }

/* The same product on the matrix cores, for K <= 64 input columns, at most 64 output columns
 * and 16-byte aligned panels.  A wave owns 64 rows.  It first requests all K input values of
 * those rows (up to sixteen 32-byte reads per lane, all in flight together) and only then
 * multiplies and stores, so the product can be formed in place whatever the output width, with
 * no scratch copy.  One v_mfma_f64_16x16x4 instruction forms, for 16 output columns and 4 input
 * columns, a 16 (output column) x 16 (row) tile of the transposed result: operand A is a block
 * of the coefficient panel (read from LDS), operand B four input columns of 16 rows.  Which 16
 * rows a tile stands for is free: tiles 2u and 2u + 1 take rows 32u + 2j and 32u + 2j + 1 of the
 * wave's 64 (j = lane & 15), so that a 16-byte request per lane covers 256 contiguous bytes of
 * each of four columns.  Measured at 16.7 M rows, 64 input columns: 2.86 / 3.51 / 4.12 ms for
 * 32 / 48 / 64 output columns, against 2.62 / 5.57 / 6.87 ms of the vector kernel above (its
 * two wider cases through a scratch panel): this kernel serves results wider than 32 columns.
 * Two waves per SIMD hide each other's loads; a single wave that requests its next tile before
 * it multiplies (twice the registers) measured 4.2 ms at 64 x 32, and 32-row tiles, whose two
 * accumulator chains leave the matrix core waiting for its own results, 4.0 ms.
 * Bq is the K x E row-major panel, E a multiple of 16, zero padded. */
typedef double ksg_d4 __attribute__((ext_vector_type(4)));
#define KSG_MIPM_KMAX 64
/* The input values of one 64-row tile: xv[ks][t] = Vin(row of tile t, column 4 ks + kq).  NKS: slices
 * of 4 input columns the code is unrolled for (slices past K hold zeros and are not read); FULL:
 * all rows of the tile exist. */
template <int NKS, bool FULL>
__device__ __forceinline__ void mipm_load(double (&xv)[NKS][4], const double *Vin, int64_t ld, int K,
                                          int64_t n, int64_t r, int kq) {
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) {
    const int col = 4 * ks + kq;
    const bool cin = col < K;                          /* columns past K enter as zeros */
    const double *p = Vin + (int64_t)(cin ? col : K - 1) * ld + r;
    if (4 * ks < K) {                                  /* wave-uniform */
      if (FULL) {
        const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 32);
        xv[ks][0] = cin ? a.x : 0.0; xv[ks][1] = cin ? a.y : 0.0;
        xv[ks][2] = cin ? b.x : 0.0; xv[ks][3] = cin ? b.y : 0.0;
      } else {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int dr = 16 * (t & ~1) + (t & 1);
          xv[ks][t] = (cin && r + dr < n) ? p[dr] : 0.0;
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) xv[ks][t] = 0.0;
    }
  }
}

/* Products and stores of one tile whose input values are in xv. */
template <int NKS, bool FULL>
__device__ __forceinline__ void mipm_multiply(const double (&xv)[NKS][4], const double *Bs, int E, int nct,
                                              double *Vout, int64_t ldo, int ne, int64_t n,
                                              double beta, int64_t r, int j, int kq) {
#pragma unroll 1
  for (int ct = 0; ct < nct; ct++) {
    ksg_d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (ksg_d4){0.0, 0.0, 0.0, 0.0};
    const double *bp = Bs + kq * E + 16 * ct + j;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      const double a = bp[4 * ks * E];
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xv[ks][t], acc[t], 0, 0, 0);
    }
    /* element v of tile t: output column 16 ct + kq + 4 v, row r + 16 (t & ~1) + (t & 1) */
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int oc = 16 * ct + kq + 4 * v;
      if (oc < ne) {
        double *o = Vout + (int64_t)oc * ldo + r;
        if (FULL) {
#pragma unroll
          for (int t = 0; t < 4; t += 2) {
            double2 w = make_double2(acc[t][v], acc[t + 1][v]);
            if (beta != 0.0) {
              const double2 old = *reinterpret_cast<const double2 *>(o + 16 * t);
              w.x = fma(beta, old.x, w.x); w.y = fma(beta, old.y, w.y);
            }
            *reinterpret_cast<double2 *>(o + 16 * t) = w;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const int dr = 16 * (t & ~1) + (t & 1);
            if (r + dr < n) o[dr] = (beta != 0.0) ? fma(beta, o[dr], acc[t][v]) : acc[t][v];
          }
        }
      }
    }
  }
}

template <int NKS>
__global__ __launch_bounds__(256, 2) void k_mip_mfma(const double *Vin, int64_t ld, int K,
                                                      const double *__restrict__ Bq, int E,
                                                      double *Vout, int64_t ldo, int ne,
                                                      int64_t n, double beta) {
  __shared__ double Bs[KSG_MIPM_KMAX * 64];
  for (int t = threadIdx.x; t < 4 * NKS * E; t += 256) Bs[t] = t < K * E ? Bq[t] : 0.0;
  __syncthreads();                                     /* the only barrier: waves run on their own from here */
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int nct = E >> 4;
  const int64_t ntiles = (n + 63) >> 6, nfull = n >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  double xv[NKS][4];
  for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += nwaves) {
    /* a lane's rows: r, r + 1 (tiles 0, 1) and r + 32, r + 33 (tiles 2, 3) of the wave's 64, so that
     * the 16 lanes of a column group read 256 contiguous bytes per request */
    const int64_t r = (tile << 6) + 2 * j;
    if (tile < nfull) {
      mipm_load<NKS, true>(xv, Vin, ld, K, n, r, kq);
      mipm_multiply<NKS, true>(xv, Bs, E, nct, Vout, ldo, ne, n, beta, r, j, kq);
    } else {
      mipm_load<NKS, false>(xv, Vin, ld, K, n, r, kq);
      mipm_multiply<NKS, false>(xv, Bs, E, nct, Vout, ldo, ne, n, beta, r, j, kq);
    }
  }
}

static int g_mip_grid = 0;   /* workgroups of k_mip_mfma (KSG_MIP_GRID) */
static bool mip_mfma_enabled() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("KSG_MIP_VALU");
    on = (e && e[0] == '1') ? 0 : 1;
    const char *g = getenv("KSG_MIP_GRID");
    g_mip_grid = g ? atoi(g) : 0;
    if (g_mip_grid < 1) g_mip_grid = 1024;
  }
  return on == 1;
}
static inline bool mip_mfma_fits(const double *Vin, int64_t ld, int K, const double *Vout, int64_t ldo) {
  return K >= 1 && K <= KSG_MIPM_KMAX && aligned16(Vin) && aligned16(Vout) && (ld % 2 == 0) && (ldo % 2 == 0) &&
         mip_mfma_enabled();
}
/* out(:,0:ne) = beta*out + Vin(:,0:K) * B on the matrix cores; ne <= 64, Bq the K x E panel, E = 16*ceil(ne/16) */
static int mip_panel_mfma(ksg_ctx c, const double *Vin, int64_t ld, int K, const double *Bq, int E, double *Vout,
                          int64_t ldo, int ne, int64_t n, double beta) {
  const int64_t nwg = ((n + 63) / 64 + 3) / 4;
  const dim3 grid((unsigned)(nwg > g_mip_grid ? g_mip_grid : nwg)), block(256);
#define KSG_MIPM_CASE(NKS) \
  hipLaunchKernelGGL((k_mip_mfma<NKS>), grid, block, 0, c->stream, Vin, ld, K, Bq, E, Vout, ldo, ne, n, beta); break;
  switch ((K + 15) / 16) {
    case 1:  KSG_MIPM_CASE(4)
    case 2:  KSG_MIPM_CASE(8)
    case 3:  KSG_MIPM_CASE(12)
    default: KSG_MIPM_CASE(16)
  }
#undef KSG_MIPM_CASE
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* ------------------------------------------------------------------------ */
/* element-wise kernels over a block of columns                               */
/* ------------------------------------------------------------------------ */
__global__ __launch_bounds__(KSG_BLOCK) void k_scale_cols(double *__restrict__ X, int64_t ld, int64_t n, double alpha) {
  double *x = X + (int64_t)blockIdx.y * ld;
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  const int64_t npair = n >> 1;
  double2 *x2 = reinterpret_cast<double2 *>(x);
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < npair; p += stride) {
    double2 v = x2[p]; v.x *= alpha; v.y *= alpha; x2[p] = v;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] *= alpha;
}

__global__ __launch_bounds__(KSG_BLOCK) void k_set_cols(double *__restrict__ X, int64_t ld, int64_t n, double alpha) {
  double *x = X + (int64_t)blockIdx.y * ld;
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride) x[p] = alpha;
}

/* y = beta*y + alpha*x over a block of columns (BVAXPY_BLAS_Private, bvblas.c:179); X may be Y */
__global__ __launch_bounds__(KSG_BLOCK) void k_axpby_cols(double *Y, int64_t ldy, const double *X,
                                                           int64_t ldx, int64_t n, double alpha, double beta) {
  double *y = Y + (int64_t)blockIdx.y * ldy;
  const double *x = X + (int64_t)blockIdx.y * ldx;
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride)
    y[p] = (beta == 0.0 ? 0.0 : beta * y[p]) + alpha * x[p];
}

/* per-column sum of squares (MODE 0) or of absolute values (MODE 1) */
template <int MODE>
__global__ __launch_bounds__(KSG_BLOCK) void k_colsum(const double *__restrict__ X, int64_t ld, int64_t n,
                                                       double *__restrict__ partial, int nb) {
  const double *x = X + (int64_t)blockIdx.y * ld;
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  const int64_t npair = n >> 1;
  const double2 *x2 = reinterpret_cast<const double2 *>(x);
  double a = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < npair; p += stride) {
    const double2 v = x2[p];
    if (MODE == 0) { a = fma(v.x, v.x, a); a = fma(v.y, v.y, a); }
    else a += fabs(v.x) + fabs(v.y);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double v = x[n - 1];
    a += MODE == 0 ? v * v : fabs(v);
  }
  __shared__ double sm[KSG_BLOCK / 64];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = sm[0];
#pragma unroll
    for (int w = 1; w < KSG_BLOCK / 64; w++) s += sm[w];
    partial[(int64_t)blockIdx.y * nb + blockIdx.x] = s;
  }
}

/* max over rows of the row sums of absolute values of `ncols` columns */
__global__ __launch_bounds__(KSG_BLOCK) void k_rowabs_max(const double *__restrict__ X, int64_t ld, int ncols,
                                                           int64_t n, double *__restrict__ partial) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  double m = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride) {
    double s = 0.0;
    for (int j = 0; j < ncols; j++) s += fabs(X[(int64_t)j * ld + p]);
    m = fmax(m, s);
  }
  __shared__ double sm[KSG_BLOCK / 64];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = sm[0];
#pragma unroll
    for (int w = 1; w < KSG_BLOCK / 64; w++) s = fmax(s, sm[w]);
    partial[blockIdx.x] = s;
  }
}

/* ------------------------------------------------------------------------ */
/* Gram block: M = Y^T X  (tall-skinny GEMM^T)                                 */
/* ------------------------------------------------------------------------ */
/* A wave owns an 8 x 8 tile of the result (8 columns of Y, 8 of X) and walks 64 row
 * pairs per step with all 16 loads in flight; the WY x WX waves of a workgroup walk the
 * same rows, so within a workgroup every column comes from HBM once and the other waves
 * find it in cache.  Columns past the end of a ragged tile re-read the tile's first
 * column (a cache hit) and their results are dropped.  partial[(j*ky + i)*nb + row block]. */
#define KSG_GT 8
__global__ __launch_bounds__(256) void k_gram(const double *__restrict__ Y, int64_t ldy, int ky,
                                              const double *__restrict__ X, int64_t ldx, int kx, int64_t n,
                                              double *__restrict__ partial, int nb, int WY, int ngy, int ngx, int ntb) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int WX = (blockDim.x >> 6) / WY;
  const int tiles_y = (ngy + WY - 1) / WY;
  /* Workgroups are dealt to the 8 XCDs round robin, each XCD with its own L2.  All the
   * workgroups that walk the same rows (one per block of tiles) are given ids that land on
   * one XCD next to each other, so they run together and share those rows through its L2
   * instead of each fetching them from HBM. */
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tb = q % ntb;                          /* which block of tiles */
  const int rb = (q / ntb) * 8 + xcd;              /* which block of rows  */
  const int gy = (tb % tiles_y) * WY + w % WY;
  const int gx = (tb / tiles_y) * WX + w / WY;
  if (gy >= ngy || gx >= ngx) return;             /* no barriers in this kernel */
  const int i0 = gy * KSG_GT, j0 = gx * KSG_GT;
  const int ni = min(KSG_GT, ky - i0), nj = min(KSG_GT, kx - j0);
  const double2 *yc[KSG_GT], *xc[KSG_GT];
#pragma unroll
  for (int t = 0; t < KSG_GT; t++) {
    yc[t] = reinterpret_cast<const double2 *>(Y + (int64_t)(i0 + (t < ni ? t : 0)) * ldy);
    xc[t] = reinterpret_cast<const double2 *>(X + (int64_t)(j0 + (t < nj ? t : 0)) * ldx);
  }
  double acc[KSG_GT][KSG_GT];
#pragma unroll
  for (int a = 0; a < KSG_GT; a++)
#pragma unroll
    for (int b = 0; b < KSG_GT; b++) acc[a][b] = 0.0;
  const int64_t npair = n >> 1;
  const int64_t stride = (int64_t)nb * 64;
  int64_t p = (int64_t)rb * 64 + lane;
  /* one wave per SIMD: the loads of the next step are issued before the 128 FMAs of this
   * one, so the matrix pipe never waits for memory */
  double2 yn[KSG_GT], xn[KSG_GT];
  if (p < npair) {
#pragma unroll
    for (int t = 0; t < KSG_GT; t++) { yn[t] = yc[t][p]; xn[t] = xc[t][p]; }
  }
  for (; p < npair; p += stride) {
    double2 yv[KSG_GT], xv[KSG_GT];
#pragma unroll
    for (int t = 0; t < KSG_GT; t++) { yv[t] = yn[t]; xv[t] = xn[t]; }
    const int64_t pn = p + stride;
    if (pn < npair) {
#pragma unroll
      for (int t = 0; t < KSG_GT; t++) { yn[t] = yc[t][pn]; xn[t] = xc[t][pn]; }
    }
#pragma unroll
    for (int t = 0; t < KSG_GT; t++) { ld_fence(yv[t]); ld_fence(xv[t]); }
#pragma unroll
    for (int a = 0; a < KSG_GT; a++)
#pragma unroll
      for (int b = 0; b < KSG_GT; b++) acc[a][b] = dot_acc(yv[a], xv[b], acc[a][b]);
  }
  if ((n & 1) && rb == 0 && lane == 0) {
#pragma unroll
    for (int a = 0; a < KSG_GT; a++)
#pragma unroll
      for (int b = 0; b < KSG_GT; b++)
        acc[a][b] = fma(Y[(int64_t)(i0 + (a < ni ? a : 0)) * ldy + n - 1], X[(int64_t)(j0 + (b < nj ? b : 0)) * ldx + n - 1], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < KSG_GT; a++)
#pragma unroll
    for (int b = 0; b < KSG_GT; b++) {
      const double v = wave_sum(acc[a][b]);
      if (lane == 0 && a < ni && b < nj) partial[((int64_t)(j0 + b) * ky + i0 + a) * nb + rb] = v;
    }
}

/* The same block product on the matrix cores.  Accumulators of v_mfma_f64_16x16x4_f64 are
 * spread over the wave (4 results per lane and 16 x 16 tile), so one wave holds a whole
 * 64 x 64 block of results and every element of Y and X is fetched once per wave and row --
 * the register-tiled kernel above fetches each column once per 8 x 8 tile and is bound by the
 * L2 -> L1 path at 64 columns.  A lane supplies one entry of A = Y^T (row i = lane & 15 of the
 * tile, k = lane >> 4) and one of B = X (column j = lane & 15, k = lane >> 4) per instruction;
 * since the order of the rows in the sum is free, k stands for the four consecutive rows
 * r0 + 4k .. r0 + 4k + 3 of a group of 16, one per instruction, so that a lane reads 32
 * contiguous bytes of a column and the four lanes of a column one whole cache line.  D holds
 * row (lane >> 4) + 4 v, column lane & 15 in element v.
 * sym: X and Y are the same panel (M is symmetric): blocks and tiles below the diagonal are
 * left out and filled in by the caller. */

/* M(i,j) = M(j,i) for i > j, M k x k column-major */
__global__ __launch_bounds__(KSG_BLOCK) void k_mirror_upper(double *__restrict__ M, int k) {
  const int e = blockIdx.x * KSG_BLOCK + threadIdx.x;
  if (e >= k * k) return;
  const int i = e % k, j = e / k;
  if (i > j) M[(int64_t)j * k + i] = M[(int64_t)i * k + j];
}

template <int T>
__global__ __launch_bounds__(256, 2) void k_gram_mfma(const double *__restrict__ Y, int64_t ldy, int ky,
                                                   const double *__restrict__ X, int64_t ldx, int kx, int64_t n,
                                                   double *__restrict__ partial, int nrb, int nby, int nbx, int sym) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nob = nby * nbx;
  const int ob = gw % nob, rb = gw / nob;
  if (rb >= nrb) return;                               /* no barriers in this kernel */
  const int by = ob % nby, bx = ob / nby;
  if (sym && bx < by) return;
  const bool diag = sym && bx == by;
  const int c = lane & 15, k = lane >> 4;
  const int i0 = by * (16 * T), j0 = bx * (16 * T);
  const double *yc[T], *xc[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
    yc[t] = Y + (int64_t)min(i0 + 16 * t + c, ky - 1) * ldy + 4 * k;
    xc[t] = X + (int64_t)min(j0 + 16 * t + c, kx - 1) * ldx + 4 * k;
  }
  ksg_d4 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; a++)
#pragma unroll
    for (int b = 0; b < T; b++) acc[a][b] = (ksg_d4){0.0, 0.0, 0.0, 0.0};
  const int64_t ngroups = n >> 4;                      /* whole groups of 16 rows */
  double yn[T][4], xn[T][4];
  int64_t g = rb;
  auto load = [&](int64_t gg) {
    const int64_t r0 = gg << 4;
#pragma unroll
    for (int t = 0; t < T; t++) {
      const double2 a0 = *reinterpret_cast<const double2 *>(yc[t] + r0), a1 = *reinterpret_cast<const double2 *>(yc[t] + r0 + 2);
      yn[t][0] = a0.x; yn[t][1] = a0.y; yn[t][2] = a1.x; yn[t][3] = a1.y;
    }
    if (!diag) {
#pragma unroll
      for (int t = 0; t < T; t++) {
        const double2 b0 = *reinterpret_cast<const double2 *>(xc[t] + r0), b1 = *reinterpret_cast<const double2 *>(xc[t] + r0 + 2);
        xn[t][0] = b0.x; xn[t][1] = b0.y; xn[t][2] = b1.x; xn[t][3] = b1.y;
      }
    }
  };
  auto multiply = [&](double (&yv)[T][4], double (&xv)[T][4]) {
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int a = 0; a < T; a++)
#pragma unroll
        for (int b = 0; b < T; b++) {
          if (diag && b < a) continue;
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(yv[a][s], diag ? yv[b][s] : xv[b][s], acc[a][b], 0, 0, 0);
        }
  };
  /* two or more waves per SIMD hide each other's loads (a register-level prefetch costs the
   * second wave and measured slower) */
  for (; g < ngroups; g += nrb) {
    load(g);
    multiply(yn, xn);
  }
  if ((n & 15) && rb == 0) {
    /* the last, partial group: rows past the end count as zeros */
    const int64_t r0 = ngroups << 4;
    double yv[T][4], xv[T][4];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const bool in = r0 + 4 * k + s < n;
        yv[t][s] = in ? yc[t][r0 + s] : 0.0;
        xv[t][s] = in ? xc[t][r0 + s] : 0.0;
      }
    multiply(yv, xv);
  }
#pragma unroll
  for (int a = 0; a < T; a++)
#pragma unroll
    for (int b = 0; b < T; b++) {
      if (diag && b < a) continue;
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int i = i0 + 16 * a + k + 4 * v, j = j0 + 16 * b + c;
        if (i < ky && j < kx) partial[((int64_t)j * ky + i) * nrb + rb] = acc[a][b][v];
      }
    }
}

/* d_out[j*ky + i] = Y(:,i)^T X(:,j) over the local rows; needs 16-byte aligned columns */
static int launch_gram(ksg_ctx c, const double *Y, int64_t ldy, int ky, const double *X, int64_t ldx, int kx,
                       int64_t n, double *d_out) {
  const int gram_sym = (Y == X && ldy == ldx && ky == kx) ? 1 : 0;
  if ((ky > 16 || kx > 16 || (gram_sym && ky > KSG_GT)) && ldy % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)Y & 31) == 0 && ((uintptr_t)X & 31) == 0 && !getenv("KSG_GRAM_VALU")) {
    /* more than one register tile: the matrix-core kernel */
    const int kmax = ky > kx ? ky : kx;
    const int T = kmax <= 16 ? 1 : (kmax <= 32 ? 2 : 4);
    const int nby = (ky + 16 * T - 1) / (16 * T), nbx = (kx + 16 * T - 1) / (16 * T);
    const int sym = gram_sym;
    const int64_t groups = n >> 4;
    int nrb = 4096 / (nby * nbx);                  /* 4 waves per SIMD in flight or queued */
    if (nrb < 8) nrb = 8;
    if ((int64_t)nrb > groups) nrb = groups > 0 ? (int)groups : 1;
    const int nwaves = nrb * nby * nbx;
    KSG_CHK(ksg_ensure_partial(c, (size_t)kx * ky * nrb));
    const dim3 grid((unsigned)((nwaves + 3) / 4)), block(256);
    if (T == 1) hipLaunchKernelGGL((k_gram_mfma<1>), grid, block, 0, c->stream, Y, ldy, ky, X, ldx, kx, n, c->d_partial, nrb, nby, nbx, sym);
    else if (T == 2) hipLaunchKernelGGL((k_gram_mfma<2>), grid, block, 0, c->stream, Y, ldy, ky, X, ldx, kx, n, c->d_partial, nrb, nby, nbx, sym);
    else hipLaunchKernelGGL((k_gram_mfma<4>), grid, block, 0, c->stream, Y, ldy, ky, X, ldx, kx, n, c->d_partial, nrb, nby, nbx, sym);
    hipLaunchKernelGGL((k_reduce_partials<0>), dim3(kx * ky), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nrb, d_out);
    KSG_HIP(hipGetLastError());
    if (sym) {
      /* results below the diagonal were left out: mirror them on the device */
      hipLaunchKernelGGL(k_mirror_upper, dim3((unsigned)((kx * ky + KSG_BLOCK - 1) / KSG_BLOCK)), dim3(KSG_BLOCK), 0, c->stream, d_out, ky);
      KSG_HIP(hipGetLastError());
    }
    return KSG_OK;
  }
  const int ngy = (ky + KSG_GT - 1) / KSG_GT, ngx = (kx + KSG_GT - 1) / KSG_GT;
  int WY = ngy < 4 ? ngy : 4;
  int WX = 4 / WY; if (WX > ngx) WX = ngx;
  const int tiles_y = (ngy + WY - 1) / WY, tiles_x = (ngx + WX - 1) / WX;
  int64_t nbl = ((n + 1) / 2 + 63) / 64;
  const int nb = (int)(nbl > 256 ? 256 : (nbl + 7) / 8 * 8);      /* row blocks: a multiple of the 8 XCDs */
  const int ntb = tiles_y * tiles_x;
  KSG_CHK(ksg_ensure_partial(c, (size_t)kx * ky * nb));
  hipLaunchKernelGGL(k_gram, dim3(nb * ntb), dim3(64 * WY * WX), 0, c->stream, Y, ldy, ky, X, ldx, kx, n,
                     c->d_partial, nb, WY, ngy, ngx, ntb);
  hipLaunchKernelGGL((k_reduce_partials<0>), dim3(kx * ky), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nb, d_out);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* d_out[j] = sum_r X[r + j*ld]^2, j < ncols */
int ksg_launch_sumsq(ksg_ctx c, const double *X, int64_t ld, int ncols, int64_t n, double *d_out) {
  if (ncols <= 0) return KSG_OK;
  if (n <= 0) { KSG_HIP(hipMemsetAsync(d_out, 0, (size_t)ncols * sizeof(double), c->stream)); return KSG_OK; }
  const int nb = ksg_grid_for((n + 1) / 2);
  KSG_CHK(ksg_ensure_partial(c, (size_t)ncols * nb));
  hipLaunchKernelGGL((k_colsum<0>), dim3(nb, ncols), dim3(KSG_BLOCK), 0, c->stream, X, ld, n, c->d_partial, nb);
  hipLaunchKernelGGL((k_reduce_partials<0>), dim3(ncols), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nb, d_out);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* ------------------------------------------------------------------------ */
/* C ABI                                                                      */
/* ------------------------------------------------------------------------ */
#define BV_CHECK(bv, name) do { if (!(bv)) KSG_FAIL(KSG_ERR_ARG, name ": NULL BV"); } while (0)

static inline double *bv_col(ksg_bv bv, int j) { return bv->d + (int64_t)(bv->nc + j) * bv->ld; }

static int check_active(ksg_bv bv, const char *name) {
  if (bv->l < -bv->nc || bv->k > bv->m || bv->l > bv->k)
    KSG_FAIL(KSG_ERR_ARG, "%s: active columns [%d,%d) outside [-%d,%d]", name, bv->l, bv->k, bv->nc, bv->m);
  return KSG_OK;
}

extern "C" int ksg_bv_create(ksg_ctx ctx, int64_t n, int64_t N, int m, ksg_bv *out) {
  if (!ctx || !out || n < 0 || N < n || m < 1) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_create: bad arguments (n=%lld N=%lld m=%d)", (long long)n, (long long)N, m);
  KSG_HIP(hipSetDevice(ctx->device));
  ksg_bv bv = new (std::nothrow) ksg_bv_s();
  if (!bv) KSG_FAIL(KSG_ERR_MEM, "ksg_bv_create: out of host memory");
  bv->ctx = ctx; bv->n = n; bv->N = N; bv->m = m; bv->nc = 0; bv->l = 0; bv->k = m;
  /* column stride: the row count rounded up to 32 doubles; from 8 M rows on to 2^20 doubles (8 MiB), so
   * that every column starts at the same offset of the memory system's interleaving -- measured at 16.0 M
   * rows: 100.8 ms per restart against 102.5 ms (strides rounded to 2^12 or 2^16 doubles: 104.1, 102.9) */
  int64_t align = n >= ((int64_t)1 << 23) ? ((int64_t)1 << 20) : 32;
  { const char *pe = getenv("KSG_BV_ALIGN"); if (pe && atoll(pe) >= 32) align = atoll(pe) / 32 * 32; }
  bv->ld = (n + align - 1) / align * align;
  if (bv->ld == 0) bv->ld = 32;
  const size_t bytes = (size_t)bv->ld * m * sizeof(double);
  hipError_t e = hipMalloc(&bv->d, bytes);
  if (e != hipSuccess) { delete bv; KSG_FAIL(KSG_ERR_MEM, "ksg_bv_create: %zu bytes: %s", bytes, hipGetErrorString(e)); }
  KSG_HIP(hipMemsetAsync(bv->d, 0, bytes, ctx->stream));
  *out = bv;
  return KSG_OK;
}

extern "C" int ksg_bv_destroy(ksg_bv bv) {
  if (!bv) return KSG_OK;
  hipSetDevice(bv->ctx->device);
  ksg_flush_pending(bv->ctx);
  hipStreamSynchronize(bv->ctx->stream);
  hipFree(bv->d);
  bv->ctx->gen++;
  delete bv;
  return KSG_OK;
}

extern "C" int ksg_bv_resize(ksg_bv bv, int m, int copy) {
  BV_CHECK(bv, "ksg_bv_resize");
  if (m < 1) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_resize: m=%d", m);
  if (m == bv->m) return KSG_OK;
  ksg_ctx c = bv->ctx;
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(ksg_flush_pending(c));
  double *nd = NULL;
  const size_t bytes = (size_t)bv->ld * (bv->nc + m) * sizeof(double);
  hipError_t e = hipMalloc(&nd, bytes);
  if (e != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "ksg_bv_resize: %zu bytes: %s", bytes, hipGetErrorString(e));
  KSG_HIP(hipMemsetAsync(nd, 0, bytes, c->stream));
  if (copy) {
    const int mc = (m < bv->m ? m : bv->m) + bv->nc;
    KSG_HIP(hipMemcpyAsync(nd, bv->d, (size_t)bv->ld * mc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  KSG_HIP(hipStreamSynchronize(c->stream));
  KSG_HIP(hipFree(bv->d));
  c->gen++;
  bv->d = nd;
  bv->m = m;
  if (bv->k > m) bv->k = m;
  if (bv->l > bv->k) bv->l = bv->k;
  return KSG_OK;
}

/* The first nc of the nc+m columns held become constraints (BVInsertConstraints,
 * bvfunc.c:433-439; BVSetNumConstraints, bvbasic.c:276-294): columns are then counted from
 * the first regular one, constraints have the indices -nc..-1.  No data moves. */
extern "C" int ksg_bv_set_num_constraints(ksg_bv bv, int nc) {
  BV_CHECK(bv, "ksg_bv_set_num_constraints");
  const int total = bv->nc + bv->m;
  if (nc < 0 || nc >= total) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_set_num_constraints: nc=%d with %d columns in all", nc, total);
  if (nc == bv->nc) return KSG_OK;
  KSG_HIP(hipSetDevice(bv->ctx->device));
  KSG_CHK(ksg_flush_pending(bv->ctx));
  bv->ctx->gen++;
  bv->nc = nc; bv->m = total - nc;
  bv->l = 0; bv->k = bv->m;
  return KSG_OK;
}

extern "C" int ksg_bv_set_active(ksg_bv bv, int l, int k) {
  BV_CHECK(bv, "ksg_bv_set_active");
  if (l < -bv->nc || k > bv->m || l > k) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_set_active: [%d,%d) invalid for m=%d", l, k, bv->m);
  bv->l = l; bv->k = k;
  return KSG_OK;
}

extern "C" int ksg_bv_sizes(ksg_bv bv, int64_t *n, int64_t *N, int *m, int64_t *ld) {
  BV_CHECK(bv, "ksg_bv_sizes");
  if (n) *n = bv->n;
  if (N) *N = bv->N;
  if (m) *m = bv->m;
  if (ld) *ld = bv->ld;
  return KSG_OK;
}

extern "C" double *ksg_bv_column(ksg_bv bv, int j) {
  if (!bv || j < -bv->nc || j >= bv->m) { ksg_set_error("ksg_bv_column: column %d out of range", j); return NULL; }
  return bv_col(bv, j);
}

/* The reduction a fused multvec left behind is reused when nothing has been written
 * through the library since and the request is exactly the one it anticipated: the
 * same active columns followed by the updated column itself. */
static bool spec_matches(ksg_ctx c, ksg_bv X, const double *y, int k) {
  const ksg_spec &sp = c->spec;
  return sp.valid && sp.gen == c->gen && sp.X == bv_col(X, X->l) && sp.ld == X->ld && sp.n == X->n &&
         sp.k + 1 == k && sp.y == y && y == sp.X + (int64_t)sp.k * sp.ld;
}

static int pending_products(ksg_ctx c);

static bool pending_matches(ksg_ctx c, ksg_bv X, const double *y, int k) {
  const ksg_pending &pd = c->pend;
  return pd.active && c->fuse_dots && pd.X == bv_col(X, X->l) && pd.ld == X->ld && pd.n == X->n &&
         pd.k + 1 == k && pd.y == y && y == pd.X + (int64_t)pd.k * pd.ld;
}

static int dotvec_impl(ksg_bv X, const double *y, double *q_host, int global) {
  BV_CHECK(X, "ksg_bv_dotvec");
  if (!y || !q_host) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_dotvec: NULL vector");
  KSG_CHK(check_active(X, "ksg_bv_dotvec"));
  ksg_ctx c = X->ctx;
  const int k = X->k - X->l;
  if (k == 0) return KSG_OK;
  if (k > KSG_RED_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_bv_dotvec: %d active columns", k);
  double *d_res = c->d_red;
  bool hit = spec_matches(c, X, y, k);
  if (!hit && pending_matches(c, X, y, k)) {
    /* a refinement pass on a column whose update is still pending: one pass over the basis
     * forms the updated column in registers and its products, which is what is asked for */
    KSG_CHK(pending_products(c));
    hit = true;
  }
  if (hit) {
    KSG_HIP(hipSetDevice(c->device));
    d_res = c->d_spec;                      /* no reduction kernel: the update kernel formed the products */
    if (c->nranks > 1) {                    /* the reduction over ranks works in place: keep the local values */
      KSG_HIP(hipMemcpyAsync(c->d_red, c->d_spec, (size_t)k * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      d_res = c->d_red;
    }
    c->stats.calls[KSG_EV_DOTVEC_FUSED] += 1;
    c->stats.flops[KSG_EV_DOTVEC_FUSED] += 2.0 * (double)X->n * k;
  } else {
    KSG_CHK(ksg_op_begin(c, KSG_EV_DOTVEC));
    KSG_CHK(ksg_launch_dotvec(c, bv_col(X, X->l), X->ld, k, y, X->n, c->d_red));
    KSG_CHK(ksg_op_end(c, KSG_EV_DOTVEC, 8.0 * (double)X->n * (k + 1), 2.0 * (double)X->n * k));
  }
  if (global) return ksg_comm_allreduce(c, d_res, q_host, k, 0);
  KSG_HIP(hipMemcpyAsync(c->h_red, d_res, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  memcpy(q_host, c->h_red, (size_t)k * sizeof(double));
  return KSG_OK;
}

extern "C" int ksg_bv_dotvec(ksg_bv X, const double *y, double *q) { return dotvec_impl(X, y, q, 1); }
extern "C" int ksg_bv_dotvec_local(ksg_bv X, const double *y, double *q) { return dotvec_impl(X, y, q, 0); }

/* ---- the reduction for callers that cannot allocate ----------------------------------------
 * The per-workgroup partial sums live in a fixed device array instead of the context's
 * workspace (which ksg_ensure_partial grows with hipMalloc).  The array's name is only a
 * host-side handle: its device address is looked up once per device (ksg_scratch_bind, from
 * ksg_ctx_create with that device current) and kept; kernels and copies get that address. */
#define KSG_SCRATCH_DOUBLES 131072            /* 1 MiB per device, held until the process ends */
#define KSG_SCRATCH_MAXDEV  64
static __device__ __align__(256) double ksg_scratch[KSG_SCRATCH_DOUBLES];
static double *g_scratch_ptr[KSG_SCRATCH_MAXDEV];
static std::mutex g_scratch_mutex;

int ksg_scratch_bind(ksg_ctx c) {
  c->d_scratch = NULL;
  if (c->device < 0 || c->device >= KSG_SCRATCH_MAXDEV) return KSG_OK;      /* the entry point reports it */
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[c->device]) {
    double *p = NULL;
    KSG_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(ksg_scratch)));
    g_scratch_ptr[c->device] = p;
  }
  c->d_scratch = g_scratch_ptr[c->device];
  return KSG_OK;
}

extern "C" int ksg_bv_dotvec_noalloc(ksg_bv X, const double *y, double *q_dev) {
  BV_CHECK(X, "ksg_bv_dotvec_noalloc");
  if (!y || !q_dev) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_dotvec_noalloc: NULL vector");
  KSG_CHK(check_active(X, "ksg_bv_dotvec_noalloc"));
  ksg_ctx c = X->ctx;
  const int k = X->k - X->l;
  if (k == 0) return KSG_OK;
  if (!c->d_scratch) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_bv_dotvec_noalloc: no scratch array bound on device %d", c->device);
  KSG_HIP(hipSetDevice(c->device));
  const double *Xb = bv_col(X, X->l);
  ksg_dot_geom g;
  if (X->n > 0) {
    g = dot_geometry(c, Xb, X->ld, k, y, X->n);
    if ((int64_t)k * g.nb > KSG_SCRATCH_DOUBLES)
      KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_bv_dotvec_noalloc: %d columns x %d partial sums exceed the fixed scratch of %d doubles "
               "(at most %d columns at the default KSG_OPT_DOT_GRID)", k, g.nb, KSG_SCRATCH_DOUBLES, KSG_SCRATCH_DOUBLES / 1024);
  }
  if (c->pend.active) {
    /* a deferred Gram-Schmidt update is applied first, without the event pair that timing
     * would create for it */
    const int timing = c->timing;
    c->timing = 0;
    const int rc = ksg_flush_pending(c);
    c->timing = timing;
    if (rc) return rc;
  }
  if (X->n <= 0) { KSG_HIP(hipMemsetAsync(q_dev, 0, (size_t)k * sizeof(double), c->stream)); }
  else KSG_CHK(launch_dotvec_geom(c, g, Xb, X->ld, k, y, X->n, c->d_scratch, q_dev));
  c->stats.calls[KSG_EV_DOTVEC] += 1;
  c->stats.bytes[KSG_EV_DOTVEC] += 8.0 * (double)X->n * (k + 1);
  c->stats.flops[KSG_EV_DOTVEC] += 2.0 * (double)X->n * k;
  return KSG_OK;
}

/* ---- posted reductions ------------------------------------------------------------------
 * Split-phase reductions (BVDotVecBegin/End, VecDotBegin/End, ... of the reference,
 * bvglobal.c:190-310, which ride on PETSc's merged MPI_Allreduce): every post launches its
 * reduction kernel into the next free slot of one device buffer and returns at once; the
 * close does one all-reduce over all slots and one copy to the host.  A step of the delayed
 * Arnoldi or of a one-sided bidiagonalisation thus costs one synchronisation. */
extern "C" int ksg_red_open(ksg_ctx c) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "ksg_red_open: NULL context");
  if (c->batch_open) KSG_FAIL(KSG_ERR_ARG, "ksg_red_open: a batch of reductions is already open");
  c->batch_open = 1;
  c->batch_count = 0;
  return KSG_OK;
}

static int red_slot(ksg_ctx c, int count, const char *name, double **d_out, int *offset) {
  if (!c->batch_open) KSG_FAIL(KSG_ERR_ARG, "%s: no batch of reductions is open", name);
  if (c->batch_count + count > KSG_RED_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "%s: more than %d values in one batch", name, KSG_RED_MAX);
  *d_out = c->d_batch + c->batch_count;
  if (offset) *offset = c->batch_count;
  c->batch_count += count;
  return KSG_OK;
}

extern "C" int ksg_red_post_dotvec(ksg_bv X, const double *y, int *offset) {
  BV_CHECK(X, "ksg_red_post_dotvec");
  if (!y) KSG_FAIL(KSG_ERR_ARG, "ksg_red_post_dotvec: NULL vector");
  KSG_CHK(check_active(X, "ksg_red_post_dotvec"));
  ksg_ctx c = X->ctx;
  const int k = X->k - X->l;
  double *d_out;
  KSG_CHK(red_slot(c, k, "ksg_red_post_dotvec", &d_out, offset));
  if (k == 0) return KSG_OK;
  KSG_CHK(ksg_op_begin(c, KSG_EV_DOTVEC));
  if (X->n) { KSG_CHK(ksg_launch_dotvec(c, bv_col(X, X->l), X->ld, k, y, X->n, d_out)); }
  else KSG_HIP(hipMemsetAsync(d_out, 0, (size_t)k * sizeof(double), c->stream));
  return ksg_op_end(c, KSG_EV_DOTVEC, 8.0 * (double)X->n * (k + 1), 2.0 * (double)X->n * k);
}

extern "C" int ksg_red_post_dot(ksg_ctx c, const double *x, const double *y, int64_t n, int *offset) {
  if (!c || n < 0 || (n && (!x || !y))) KSG_FAIL(KSG_ERR_ARG, "ksg_red_post_dot: bad arguments");
  double *d_out;
  KSG_CHK(red_slot(c, 1, "ksg_red_post_dot", &d_out, offset));
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  if (n) { KSG_CHK(ksg_launch_dotvec(c, x, 2, 1, y, n, d_out)); }
  else KSG_HIP(hipMemsetAsync(d_out, 0, sizeof(double), c->stream));
  return ksg_op_end(c, KSG_EV_VECOPS, (x == y ? 8.0 : 16.0) * (double)n, 2.0 * (double)n);
}

extern "C" int ksg_red_close(ksg_ctx c, double *out_host, int count) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "ksg_red_close: NULL context");
  if (!c->batch_open) KSG_FAIL(KSG_ERR_ARG, "ksg_red_close: no batch of reductions is open");
  if (count < 0) { c->batch_open = 0; return KSG_OK; }      /* abandon: the values posted are dropped */
  if (count != c->batch_count) KSG_FAIL(KSG_ERR_ARG, "ksg_red_close: %d values asked for, %d posted", count, c->batch_count);
  c->batch_open = 0;
  if (!count) return KSG_OK;
  if (!out_host) KSG_FAIL(KSG_ERR_ARG, "ksg_red_close: NULL output");
  KSG_HIP(hipSetDevice(c->device));
  return ksg_comm_allreduce(c, c->d_batch, out_host, count, 0);
}

/* Applies the deferred update on its own.  The values written are bit for bit the ones
 * whose products ksg_spec holds: same kernel, same column groups, same order of sums. */
static int apply_pending(ksg_ctx c, const ksg_pending &pd, double post) {
  KSG_CHK(ksg_op_begin_ex(c, KSG_EV_MULTVEC, KSG_OPB_NOFLUSH | KSG_OPB_NOBUMP));
  ksg_qarg q1, q2;
  for (int j = 0; j < pd.k; j++) { q1.v[j] = pd.q[j]; q2.v[j] = pd.q2[j]; }
  KSG_CHK(launch_fused(c, pd.X, pd.ld, pd.k, q1, pd.beta, pd.stage == 2 ? &q2 : NULL, pd.beta2, pd.y, pd.n, NULL, 1, post));
  KSG_CHK(ksg_op_end(c, KSG_EV_MULTVEC, 8.0 * (double)pd.n * (pd.k + 2), (pd.stage == 2 ? 4.0 : 2.0) * (double)pd.n * pd.k));
  return KSG_OK;
}

int ksg_flush_pending(ksg_ctx c) {
  if (!c->pend.active) return KSG_OK;
  const ksg_pending pd = c->pend;
  c->pend.active = false;
  return apply_pending(c, pd, 1.0);
}

/* Products of the pending column (as it will be once updated) with the active columns and
 * itself -> d_spec.  One update pending: the column is formed in registers only and stays
 * pending; two pending: both are applied and stored. */
static int pending_products(ksg_ctx c) {
  const ksg_pending pd = c->pend;
  KSG_CHK(ksg_op_begin_ex(c, KSG_EV_MULTVEC, KSG_OPB_NOFLUSH | KSG_OPB_NOBUMP));
  ksg_qarg q1, q2;
  for (int j = 0; j < pd.k; j++) { q1.v[j] = pd.q[j]; q2.v[j] = pd.q2[j]; }
  if (pd.stage == 1) {
    KSG_CHK(launch_fused(c, pd.X, pd.ld, pd.k, q1, pd.beta, NULL, 1.0, pd.y, pd.n, c->d_spec, 0));
    KSG_CHK(ksg_op_end(c, KSG_EV_MULTVEC, 8.0 * (double)pd.n * (pd.k + 1), 4.0 * (double)pd.n * pd.k));
  } else {
    c->pend.active = false;
    KSG_CHK(launch_fused(c, pd.X, pd.ld, pd.k, q1, pd.beta, &q2, pd.beta2, pd.y, pd.n, c->d_spec, 1));
    KSG_CHK(ksg_op_end(c, KSG_EV_MULTVEC, 8.0 * (double)pd.n * (pd.k + 2), 6.0 * (double)pd.n * pd.k));
  }
  ksg_spec &sp = c->spec;
  sp.valid = true; sp.gen = c->gen; sp.X = pd.X; sp.ld = pd.ld; sp.n = pd.n; sp.k = pd.k; sp.y = pd.y;
  return KSG_OK;
}

static void set_spec(ksg_ctx c, ksg_bv X, int k, const double *y) {
  ksg_spec &sp = c->spec;
  sp.valid = true; sp.gen = c->gen; sp.X = bv_col(X, X->l); sp.ld = X->ld; sp.n = X->n; sp.k = k; sp.y = y;
}

extern "C" int ksg_bv_multvec(ksg_bv X, double alpha, double beta, double *y, const double *q_host) {
  BV_CHECK(X, "ksg_bv_multvec");
  if (!y) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_multvec: NULL vector");
  KSG_CHK(check_active(X, "ksg_bv_multvec"));
  ksg_ctx c = X->ctx;
  const int k = X->k - X->l;
  const double *Xb = bv_col(X, X->l);
  const int64_t n = X->n;
  if (k > 0 && !q_host) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_multvec: NULL coefficients");
  /* y is the column right after the active ones: the update step of Gram-Schmidt on that
   * column (bvorthog.c:262-268); if a refinement pass follows it will ask for the column's
   * products with the same columns and with itself, which the update kernel can form while
   * the columns are in registers */
  const bool gs_column = c->fuse_dots && k > 0 && n > 0 && X->k < X->m && y == bv_col(X, X->k);
  const bool one_launch = gs_column && k <= 64 && mv_vec2(Xb, X->ld, y);

  if (c->lazy_update && one_launch) {
    /* Nothing is launched yet.  What the caller does next decides how the basis is streamed:
     *  - it asks for the updated column's products (a refinement pass): one read-only pass
     *    forms them, the column stays pending;
     *  - it updates the column again: both updates wait together;
     *  - it scales the column (the normalisation, bvorthog.c:561-568): updates and scaling
     *    go in one pass with one store;
     *  - anything else that touches device data: the pending updates are applied first. */
    ksg_pending &pd = c->pend;
    KSG_HIP(hipSetDevice(c->device));
    if (pd.active && pd.stage == 1 && pd.X == Xb && pd.y == y && pd.k == k && pd.ld == X->ld && pd.n == n) {
      for (int j = 0; j < k; j++) pd.q2[j] = alpha * q_host[j];
      pd.beta2 = beta;
      pd.stage = 2;
    } else {
      KSG_CHK(ksg_flush_pending(c));
      for (int j = 0; j < k; j++) pd.q[j] = alpha * q_host[j];
      pd.X = Xb; pd.y = y; pd.ld = X->ld; pd.n = n; pd.k = k; pd.beta = beta; pd.stage = 1; pd.active = true;
    }
    c->gen++;                                     /* y has changed as far as any reader can tell */
    return KSG_OK;
  }

  KSG_CHK(ksg_op_begin(c, KSG_EV_MULTVEC));
  int fused = 0;
  double bytes = 8.0 * (double)n * (k + 2);
  if (k == 0) {
    if (beta != 1.0 && n > 0) {
      if (beta == 0.0) hipLaunchKernelGGL(k_set_cols, dim3(ksg_grid_for(n), 1), dim3(KSG_BLOCK), 0, c->stream, y, X->ld, n, 0.0);
      else hipLaunchKernelGGL(k_axpby_cols, dim3(ksg_grid_for(n), 1), dim3(KSG_BLOCK), 0, c->stream, y, X->ld, y, X->ld, n, 0.0, beta);
    }
  } else {
    KSG_CHK(launch_multvec(c, Xb, X->ld, k, q_host, alpha, beta, y, n, gs_column ? c->d_spec : NULL, &fused));
  }
  KSG_CHK(ksg_op_end(c, KSG_EV_MULTVEC, bytes, (fused ? 4.0 : 2.0) * (double)n * k));
  if (fused) set_spec(c, X, k, y);
  return KSG_OK;
}

/* Column storage was modified by something other than this library (a kernel of the
 * caller writing through a pointer obtained from ksg_bv_column, a host copy, ...). */
extern "C" int ksg_bv_touch(ksg_bv bv) {
  BV_CHECK(bv, "ksg_bv_touch");
  KSG_CHK(ksg_flush_pending(bv->ctx));
  bv->ctx->gen++;
  return KSG_OK;
}

/* pinned staging for small host matrices */
static int ensure_hq(ksg_ctx c, size_t doubles) {
  if (c->hq_cap >= doubles) return KSG_OK;
  if (c->h_q) KSG_HIP(hipHostFree(c->h_q));
  c->h_q = NULL; c->hq_cap = 0;
  KSG_HIP(hipHostMalloc(&c->h_q, (doubles + 1024) * sizeof(double), hipHostMallocDefault));
  c->hq_cap = doubles + 1024;
  return KSG_OK;
}

/* out(:, 0:ne) = beta*out + Vin(:,0:K) * B where B[i][c] is produced by `coef(i,c)`.
 * In place (output columns among the input columns, beta = 0): one launch holds up to 64
 * output columns per row in registers; wider results go through a scratch copy.  Out of
 * place: panels of 32 output columns (the fastest width), the input streamed once per
 * panel. */
template <class F>
static int mip_general(ksg_ctx c, const double *Vin, int64_t ld, int K, int ne, F coef, double *Vout,
                       int64_t ldo, int64_t n, bool inplace_overlap, double beta = 0.0) {
  if (ne <= 0 || n <= 0) return KSG_OK;
  if (ne > 32 && mip_mfma_fits(Vin, ld, K, Vout, ldo) && (ne <= 64 || !inplace_overlap)) {
    /* more than 32 output columns: the matrix-core kernel, in panels of up to 64 columns; a single
     * panel may overwrite its own input columns */
    const int npan = (ne + 63) / 64;
    const int E = npan == 1 ? (ne + 15) / 16 * 16 : 64;
    KSG_CHK(ksg_ensure_q(c, (size_t)K * E * npan));
    KSG_HIP(hipStreamSynchronize(c->stream));               /* the staging buffer may still be in use */
    KSG_CHK(ensure_hq(c, (size_t)K * E * npan));
    for (int pnl = 0; pnl < npan; pnl++) {
      const int w = (ne - pnl * 64 < 64) ? ne - pnl * 64 : 64;
      double *h = c->h_q + (size_t)pnl * K * E;
      for (int i = 0; i < K; i++)
        for (int e = 0; e < E; e++) h[(size_t)i * E + e] = e < w ? coef(i, pnl * 64 + e) : 0.0;
    }
    KSG_HIP(hipMemcpyAsync(c->d_q, c->h_q, (size_t)K * E * npan * sizeof(double), hipMemcpyHostToDevice, c->stream));
    for (int pnl = 0; pnl < npan; pnl++) {
      const int w = (ne - pnl * 64 < 64) ? ne - pnl * 64 : 64;
      KSG_CHK(mip_panel_mfma(c, Vin, ld, K, c->d_q + (size_t)pnl * K * E, E, Vout + (int64_t)pnl * 64 * ldo, ldo, w, n, beta));
    }
    return KSG_OK;
  }
  if (inplace_overlap && ne > 32 && ne <= 64 && beta == 0.0) {
    /* 33..64 output columns in place where the matrix-core kernel does not apply (more than 64
     * input columns, unaligned panels): the kernel that holds 64 outputs per row in registers
     * is slow (one row per thread, 8-byte loads), so the result is formed as two panels of
     * at most 32 columns with the fast kernel: the second panel goes to scratch first (the
     * first panel's columns are among its inputs), then the first panel is written in place,
     * then the scratch panel is copied behind it. */
    const int n2 = ne - 32;
    const int E2 = mip_width(n2);
    KSG_CHK(ksg_ensure_work(c, (size_t)ld * n2));
    KSG_CHK(ksg_ensure_q(c, (size_t)K * (32 + E2)));
    KSG_HIP(hipStreamSynchronize(c->stream));
    KSG_CHK(ensure_hq(c, (size_t)K * (32 + E2)));
    double *h1 = c->h_q, *h2 = c->h_q + (size_t)K * 32;
    for (int i = 0; i < K; i++) {
      for (int e = 0; e < 32; e++) h1[(size_t)i * 32 + e] = coef(i, e);
      for (int e = 0; e < E2; e++) h2[(size_t)i * E2 + e] = e < n2 ? coef(i, 32 + e) : 0.0;
    }
    KSG_HIP(hipMemcpyAsync(c->d_q, c->h_q, (size_t)K * (32 + E2) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    KSG_CHK(mip_panel(c, Vin, ld, K, c->d_q + (size_t)K * 32, E2, c->d_work, ld, n2, n));
    KSG_CHK(mip_panel(c, Vin, ld, K, c->d_q, 32, Vout, ldo, 32, n));
    KSG_HIP(hipMemcpy2DAsync(Vout + (int64_t)32 * ldo, (size_t)ldo * sizeof(double), c->d_work, (size_t)ld * sizeof(double),
                             (size_t)n * sizeof(double), (size_t)n2, hipMemcpyDeviceToDevice, c->stream));
    return KSG_OK;
  }
  const int PW = inplace_overlap ? 64 : 32;             /* panel width */
  const int npan = (ne + PW - 1) / PW;
  const int E = npan == 1 ? mip_width(ne) : PW;          /* padded width of a panel in d_q */
  double *dst = Vout;
  int64_t ldd = ldo;
  const bool scratch = inplace_overlap && npan > 1;
  if (scratch) {
    KSG_CHK(ksg_ensure_work(c, (size_t)ld * ne));
    dst = c->d_work; ldd = ld;
  }
  KSG_CHK(ksg_ensure_q(c, (size_t)K * E * npan));
  KSG_HIP(hipStreamSynchronize(c->stream));               /* the staging buffer may still be in use */
  KSG_CHK(ensure_hq(c, (size_t)K * E * npan));
  for (int pnl = 0; pnl < npan; pnl++) {
    const int w = (ne - pnl * PW < PW) ? ne - pnl * PW : PW;
    double *h = c->h_q + (size_t)pnl * K * E;
    for (int i = 0; i < K; i++)
      for (int e = 0; e < E; e++) h[(size_t)i * E + e] = e < w ? coef(i, pnl * PW + e) : 0.0;
  }
  KSG_HIP(hipMemcpyAsync(c->d_q, c->h_q, (size_t)K * E * npan * sizeof(double), hipMemcpyHostToDevice, c->stream));
  for (int pnl = 0; pnl < npan; pnl++) {
    const int w = (ne - pnl * PW < PW) ? ne - pnl * PW : PW;
    KSG_CHK(mip_panel(c, Vin, ld, K, c->d_q + (size_t)pnl * K * E, E, dst + (int64_t)pnl * PW * ldd, ldd, w, n,
                      scratch ? 0.0 : beta));
  }
  if (scratch)
    KSG_HIP(hipMemcpy2DAsync(Vout, (size_t)ldo * sizeof(double), dst, (size_t)ldd * sizeof(double),
                             (size_t)n * sizeof(double), (size_t)ne, hipMemcpyDeviceToDevice, c->stream));
  return KSG_OK;
}

extern "C" int ksg_bv_multinplace(ksg_bv V, const double *Q, int ldq, int s, int e, int trans) {
  BV_CHECK(V, "ksg_bv_multinplace");
  KSG_CHK(check_active(V, "ksg_bv_multinplace"));
  const int l = V->l, k = V->k;
  if (s >= e) return KSG_OK;
  if (!Q || s < l || e > (trans ? V->m : k) || e > V->m || ldq < 1)
    KSG_FAIL(KSG_ERR_ARG, "ksg_bv_multinplace: bad arguments (s=%d e=%d l=%d k=%d ldq=%d)", s, e, l, k, ldq);
  if (l < 0) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_bv_multinplace: constraints cannot be updated");
  ksg_ctx c = V->ctx;
  const int K = k - l, ne = e - s;
  KSG_CHK(ksg_op_begin(c, KSG_EV_MULTINPLACE));
  int rc;
  if (!trans) {
    auto coef = [=](int i, int cc) { return Q[(size_t)(l + i) + (size_t)(s + cc) * ldq]; };
    rc = mip_general(c, bv_col(V, l), V->ld, K, ne, coef, bv_col(V, s), V->ld, V->n, true);
  } else {
    auto coef = [=](int i, int cc) { return Q[(size_t)(s + cc) + (size_t)(l + i) * ldq]; };
    rc = mip_general(c, bv_col(V, l), V->ld, K, ne, coef, bv_col(V, s), V->ld, V->n, true);
  }
  if (rc) return rc;
  KSG_CHK(ksg_op_end(c, KSG_EV_MULTINPLACE, 8.0 * (double)V->n * (K + ne), 2.0 * (double)V->n * K * ne));
  return KSG_OK;
}

extern "C" int ksg_bv_scale(ksg_bv bv, int j, double alpha) {
  BV_CHECK(bv, "ksg_bv_scale");
  ksg_ctx c = bv->ctx;
  double *base; int ncols;
  if (j < 0) { KSG_CHK(check_active(bv, "ksg_bv_scale")); base = bv_col(bv, bv->l); ncols = bv->k - bv->l; }
  else { if (j >= bv->m) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_scale: column %d out of range", j); base = bv_col(bv, j); ncols = 1; }
  if (ncols == 0 || bv->n == 0 || alpha == 1.0) return KSG_OK;
  if (c->pend.active && j >= 0 && alpha != 0.0 && c->pend.y == base) {
    /* the column's pending update(s) and this scaling in one pass, one store */
    const ksg_pending pd = c->pend;
    c->pend.active = false;
    KSG_HIP(hipSetDevice(c->device));
    c->gen++;
    KSG_CHK(apply_pending(c, pd, alpha));
    c->stats.calls[KSG_EV_SCALE] += 1;
    c->stats.flops[KSG_EV_SCALE] += (double)bv->n;
    return KSG_OK;
  }
  KSG_CHK(ksg_op_begin(c, KSG_EV_SCALE));
  dim3 grid(ksg_grid_for((bv->n + 1) / 2), ncols);
  if (alpha == 0.0) hipLaunchKernelGGL(k_set_cols, dim3(ksg_grid_for(bv->n), ncols), dim3(KSG_BLOCK), 0, c->stream, base, bv->ld, bv->n, 0.0);
  else hipLaunchKernelGGL(k_scale_cols, grid, dim3(KSG_BLOCK), 0, c->stream, base, bv->ld, bv->n, alpha);
  KSG_HIP(hipGetLastError());
  KSG_CHK(ksg_op_end(c, KSG_EV_SCALE, (alpha == 0.0 ? 8.0 : 16.0) * (double)bv->n * ncols, (double)bv->n * ncols));
  return KSG_OK;
}

static int norm_impl(ksg_bv bv, int j, int type, double *val, int global) {
  BV_CHECK(bv, "ksg_bv_norm");
  if (!val) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_norm: NULL result");
  ksg_ctx c = bv->ctx;
  const double *base; int ncols;
  if (j < 0) { KSG_CHK(check_active(bv, "ksg_bv_norm")); base = bv_col(bv, bv->l); ncols = bv->k - bv->l; }
  else { if (j >= bv->m) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_norm: column %d out of range", j); base = bv_col(bv, j); ncols = 1; }
  if (ncols > KSG_RED_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_bv_norm: %d columns", ncols);
  const int64_t n = bv->n;
  std::vector<double> h((size_t)(ncols > 0 ? ncols : 1), 0.0);
  KSG_CHK(ksg_op_begin(c, KSG_EV_NORM));
  int count = ncols, op = 0;
  if (type == KSG_NORM_2 || type == KSG_NORM_FROBENIUS) {
    KSG_CHK(ksg_launch_sumsq(c, base, bv->ld, ncols, n, c->d_red));
  } else if (type == KSG_NORM_1) {
    if (ncols && n > 0) {
      const int nb = ksg_grid_for((n + 1) / 2);
      KSG_CHK(ksg_ensure_partial(c, (size_t)ncols * nb));
      hipLaunchKernelGGL((k_colsum<1>), dim3(nb, ncols), dim3(KSG_BLOCK), 0, c->stream, base, bv->ld, n, c->d_partial, nb);
      hipLaunchKernelGGL((k_reduce_partials<0>), dim3(ncols), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nb, c->d_red);
    } else if (ncols) KSG_HIP(hipMemsetAsync(c->d_red, 0, (size_t)ncols * sizeof(double), c->stream));
  } else if (type == KSG_NORM_INFINITY) {
    count = 1; op = 1;
    if (ncols && n > 0) {
      const int nb = ksg_grid_for(n);
      KSG_CHK(ksg_ensure_partial(c, (size_t)nb));
      hipLaunchKernelGGL(k_rowabs_max, dim3(nb), dim3(KSG_BLOCK), 0, c->stream, base, bv->ld, ncols, n, c->d_partial);
      hipLaunchKernelGGL((k_reduce_partials<1>), dim3(1), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nb, c->d_red);
    } else KSG_HIP(hipMemsetAsync(c->d_red, 0, sizeof(double), c->stream));
  } else KSG_FAIL(KSG_ERR_ARG, "ksg_bv_norm: unknown norm type %d", type);
  KSG_HIP(hipGetLastError());
  KSG_CHK(ksg_op_end(c, KSG_EV_NORM, 8.0 * (double)n * ncols, 2.0 * (double)n * ncols));
  if (count == 0) { *val = 0.0; return KSG_OK; }
  if (global) { KSG_CHK(ksg_comm_allreduce(c, c->d_red, h.data(), count, op)); }
  else {
    KSG_HIP(hipMemcpyAsync(c->h_red, c->d_red, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    KSG_HIP(hipStreamSynchronize(c->stream));
    memcpy(h.data(), c->h_red, (size_t)count * sizeof(double));
  }
  if (type == KSG_NORM_2 || type == KSG_NORM_FROBENIUS) {
    double s = 0.0; for (int i = 0; i < ncols; i++) s += h[i];
    *val = sqrt(s);
  } else if (type == KSG_NORM_1) {
    double m = 0.0; for (int i = 0; i < ncols; i++) if (h[i] > m) m = h[i];
    *val = m;
  } else *val = h[0];
  return KSG_OK;
}

extern "C" int ksg_bv_norm(ksg_bv bv, int j, int type, double *val) { return norm_impl(bv, j, type, val, 1); }
extern "C" int ksg_bv_norm_local(ksg_bv bv, int j, int type, double *val) { return norm_impl(bv, j, type, val, 0); }

extern "C" int ksg_bv_copy(ksg_bv V, ksg_bv W) {
  BV_CHECK(V, "ksg_bv_copy"); BV_CHECK(W, "ksg_bv_copy");
  KSG_CHK(check_active(V, "ksg_bv_copy")); KSG_CHK(check_active(W, "ksg_bv_copy"));
  if (V->n != W->n || V->k - V->l != W->k - W->l) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_copy: size mismatch");
  ksg_ctx c = V->ctx;
  const int ncols = V->k - V->l;
  if (!ncols || !V->n) return KSG_OK;
  KSG_CHK(ksg_op_begin(c, KSG_EV_COPY));
  KSG_HIP(hipMemcpy2DAsync(bv_col(W, W->l), (size_t)W->ld * sizeof(double), bv_col(V, V->l), (size_t)V->ld * sizeof(double),
                           (size_t)V->n * sizeof(double), (size_t)ncols, hipMemcpyDeviceToDevice, c->stream));
  KSG_CHK(ksg_op_end(c, KSG_EV_COPY, 16.0 * (double)V->n * ncols, 0.0));
  return KSG_OK;
}

extern "C" int ksg_bv_copy_column(ksg_bv V, int j, int i) {
  BV_CHECK(V, "ksg_bv_copy_column");
  if (j < -V->nc || j >= V->m || i < -V->nc || i >= V->m) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_copy_column: %d -> %d out of range", j, i);
  if (i == j || !V->n) return KSG_OK;
  ksg_ctx c = V->ctx;
  KSG_CHK(ksg_op_begin(c, KSG_EV_COPY));
  KSG_HIP(hipMemcpyAsync(bv_col(V, i), bv_col(V, j), (size_t)V->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  KSG_CHK(ksg_op_end(c, KSG_EV_COPY, 16.0 * (double)V->n, 0.0));
  return KSG_OK;
}

extern "C" int ksg_bv_mult(ksg_bv Y, double alpha, double beta, ksg_bv X, const double *Q, int ldq) {
  BV_CHECK(Y, "ksg_bv_mult"); BV_CHECK(X, "ksg_bv_mult");
  KSG_CHK(check_active(Y, "ksg_bv_mult")); KSG_CHK(check_active(X, "ksg_bv_mult"));
  if (X->n != Y->n) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_mult: local sizes differ");
  if (X == Y) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_mult: X and Y must be different objects");
  ksg_ctx c = Y->ctx;
  const int ky = Y->k - Y->l, kx = X->k - X->l;
  if (!ky || !Y->n) return KSG_OK;
  KSG_CHK(ksg_op_begin(c, KSG_EV_MULT));
  if (!Q) {
    if (kx != ky) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_mult: column counts differ");
    hipLaunchKernelGGL(k_axpby_cols, dim3(ksg_grid_for(Y->n), ky), dim3(KSG_BLOCK), 0, c->stream,
                       bv_col(Y, Y->l), Y->ld, bv_col(X, X->l), X->ld, Y->n, alpha, beta);
    KSG_HIP(hipGetLastError());
    KSG_CHK(ksg_op_end(c, KSG_EV_MULT, 24.0 * (double)Y->n * ky, 3.0 * (double)Y->n * ky));
    return KSG_OK;
  }
  if (ldq < 1) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_mult: ldq=%d", ldq);
  /* Y(:,ly:ky) = beta*Y + X(:,lx:kx) * (alpha*Q(lx:kx, ly:ky)): the restart kernel, out of place
   * (BVMult_Svec svec.c:27 -> BVMult_BLAS_Private bvblas.c:34) */
  const int xl = X->l, yl = Y->l;
  auto coef = [=](int i, int cc) { return alpha * Q[(size_t)(xl + i) + (size_t)(yl + cc) * ldq]; };
  if (kx == 0) {
    if (beta == 0.0)
      hipLaunchKernelGGL(k_set_cols, dim3(ksg_grid_for(Y->n), ky), dim3(KSG_BLOCK), 0, c->stream, bv_col(Y, Y->l), Y->ld, Y->n, 0.0);
    else if (beta != 1.0)
      hipLaunchKernelGGL(k_scale_cols, dim3(ksg_grid_for((Y->n + 1) / 2), ky), dim3(KSG_BLOCK), 0, c->stream, bv_col(Y, Y->l), Y->ld, Y->n, beta);
  } else {
    int rc = mip_general(c, bv_col(X, X->l), X->ld, kx, ky, coef, bv_col(Y, Y->l), Y->ld, Y->n, false, beta);
    if (rc) return rc;
  }
  const int npan = (ky + 31) / 32;
  KSG_CHK(ksg_op_end(c, KSG_EV_MULT, 8.0 * (double)Y->n * ((double)npan * kx + (beta != 0.0 ? 2.0 : 1.0) * ky),
                     2.0 * (double)Y->n * ky * kx));
  return KSG_OK;
}

extern "C" int ksg_bv_dot(ksg_bv X, ksg_bv Y, double *M, int ldm) {
  BV_CHECK(X, "ksg_bv_dot"); BV_CHECK(Y, "ksg_bv_dot");
  KSG_CHK(check_active(X, "ksg_bv_dot")); KSG_CHK(check_active(Y, "ksg_bv_dot"));
  if (X->n != Y->n || !M || ldm < 1) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_dot: bad arguments");
  ksg_ctx c = X->ctx;
  const int kx = X->k - X->l, ky = Y->k - Y->l;
  if (!kx || !ky) return KSG_OK;
  if ((size_t)kx * ky > KSG_RED_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_bv_dot: %d x %d result", ky, kx);
  KSG_CHK(ksg_op_begin(c, KSG_EV_DOT));
  /* M(ly+i, lx+j) = Y(:,ly+i)^T X(:,lx+j) (BVDot_Svec svec.c:102 -> BVDot_BLAS_Private bvblas.c:202):
   * register-tiled block kernel, one reduction over ranks for the whole matrix */
  const double *Yb = bv_col(Y, Y->l), *Xb = bv_col(X, X->l);
  if (X->n > 0 && aligned16(Yb) && aligned16(Xb) && Y->ld % 2 == 0 && X->ld % 2 == 0) {
    KSG_CHK(launch_gram(c, Yb, Y->ld, ky, Xb, X->ld, kx, X->n, c->d_red));
  } else {
    for (int j = 0; j < kx; j++)
      KSG_CHK(ksg_launch_dotvec(c, Yb, Y->ld, ky, Xb + (int64_t)j * X->ld, X->n, c->d_red + (size_t)j * ky));
  }
  KSG_CHK(ksg_op_end(c, KSG_EV_DOT, 8.0 * (double)X->n * (kx + ky), 2.0 * (double)X->n * kx * ky));
  std::vector<double> h((size_t)kx * ky);
  KSG_CHK(ksg_comm_allreduce(c, c->d_red, h.data(), kx * ky, 0));
  for (int j = 0; j < kx; j++)
    for (int i = 0; i < ky; i++) M[(size_t)(Y->l + i) + (size_t)(X->l + j) * ldm] = h[(size_t)j * ky + i];
  return KSG_OK;
}
