/* BV kernels and their C ABI: the op table of the reference's BVSVEC back-end
 * (src/sys/classes/bv/impls/svec/svec.c, CPU kernels in
 * src/sys/classes/bv/interface/bvblas.c) as hand-written gfx950 kernels.
 *
 * All kernels are HBM-bandwidth bound.  The basis is column-major with a padded
 * leading dimension, so a wave reads 64 consecutive row pairs of one column as
 * one 1 KiB coalesced request (16 B per lane); tall-skinny products keep the
 * per-column accumulators in registers and stream every column exactly once. */
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

/* ------------------------------------------------------------------------ */
/* dotvec: q = X^T y  (tall-skinny GEMV^T)                                    */
/* ------------------------------------------------------------------------ */
/* grid.x = row blocks (grid-stride), grid.y = chunks of C columns.  Each thread
 * keeps C accumulators; y is read once per chunk, every column once.  The last
 * chunk may hold fewer than C columns: its spare slots re-read the chunk's last
 * column (cache hits) and are dropped at the end, which keeps the loop free of
 * per-load branches.  VEC = 2 reads row pairs as double2 (pointers 16-byte
 * aligned), VEC = 1 is the fallback for unaligned vectors. */
template <int C, int VEC>
__global__ __launch_bounds__(KSG_BLOCK) void k_dotvec(const double *__restrict__ X, int64_t ld, int k,
                                                       const double *__restrict__ y, int64_t n,
                                                       double *__restrict__ partial, int nb) {
  const int c0 = blockIdx.y * C;
  const int kc = min(C, k - c0);
  const double *col[C];
#pragma unroll
  for (int c = 0; c < C; c++) col[c] = X + (int64_t)(c0 + (c < kc ? c : kc - 1)) * ld;
  double acc[C];
#pragma unroll
  for (int c = 0; c < C; c++) acc[c] = 0.0;

  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x;
  if (VEC == 2) {
    const int64_t npair = n >> 1;
    const double2 *y2 = reinterpret_cast<const double2 *>(y);
    for (; p < npair; p += stride) {
      const double2 yv = y2[p];
      double2 xv[C];
#pragma unroll
      for (int c = 0; c < C; c++) xv[c] = reinterpret_cast<const double2 *>(col[c])[p];
#pragma unroll
      for (int c = 0; c < C; c++) {
        acc[c] = fma(xv[c].x, yv.x, acc[c]);
        acc[c] = fma(xv[c].y, yv.y, acc[c]);
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const double yv = y[n - 1];
#pragma unroll
      for (int c = 0; c < C; c++) acc[c] = fma(col[c][n - 1], yv, acc[c]);
    }
  } else {
    for (; p < n; p += stride) {
      const double yv = y[p];
      double xv[C];
#pragma unroll
      for (int c = 0; c < C; c++) xv[c] = col[c][p];
#pragma unroll
      for (int c = 0; c < C; c++) acc[c] = fma(xv[c], yv, acc[c]);
    }
  }

  __shared__ double sm[KSG_BLOCK / 64][C];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; c++) {
    const double v = wave_sum(acc[c]);
    if (lane == 0) sm[w][c] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < kc) {
    double s = sm[0][threadIdx.x];
#pragma unroll
    for (int ww = 1; ww < KSG_BLOCK / 64; ww++) s += sm[ww][threadIdx.x];
    partial[(int64_t)(c0 + threadIdx.x) * nb + blockIdx.x] = s;
  }
}

template <int C>
static void launch_dotvec_c(ksg_ctx c, const double *X, int64_t ld, int k, const double *y, int64_t n,
                            int nb, int nchunks, bool vec2) {
  dim3 grid(nb, nchunks);
  if (vec2)
    hipLaunchKernelGGL((k_dotvec<C, 2>), grid, dim3(KSG_BLOCK), 0, c->stream, X, ld, k, y, n, c->d_partial, nb);
  else
    hipLaunchKernelGGL((k_dotvec<C, 1>), grid, dim3(KSG_BLOCK), 0, c->stream, X, ld, k, y, n, c->d_partial, nb);
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

/* d_out[i] = sum_r X[r + i*ld] * y[r], i < k (local rows only). */
int ksg_launch_dotvec(ksg_ctx c, const double *X, int64_t ld, int k, const double *y, int64_t n,
                      double *d_out) {
  if (k <= 0) return KSG_OK;
  if (n <= 0) { KSG_HIP(hipMemsetAsync(d_out, 0, (size_t)k * sizeof(double), c->stream)); return KSG_OK; }
  const bool vec2 = aligned16(X) && aligned16(y) && (ld % 2 == 0);
  const int nb = ksg_grid_for(vec2 ? (n + 1) / 2 : n);
  KSG_CHK(ksg_ensure_partial(c, (size_t)k * nb));
  const int nchunks16 = (k + 15) / 16;
  const int per = (k + nchunks16 - 1) / nchunks16;      /* columns per chunk, <= 16 */
  if (per <= 4)       launch_dotvec_c<4>(c, X, ld, k, y, n, nb, (k + 3) / 4, vec2);
  else if (per <= 8)  launch_dotvec_c<8>(c, X, ld, k, y, n, nb, (k + 7) / 8, vec2);
  else if (per <= 12) launch_dotvec_c<12>(c, X, ld, k, y, n, nb, (k + 11) / 12, vec2);
  else                launch_dotvec_c<16>(c, X, ld, k, y, n, nb, (k + 15) / 16, vec2);
  hipLaunchKernelGGL((k_reduce_partials<0>), dim3(k), dim3(KSG_BLOCK), 0, c->stream, c->d_partial, nb, d_out);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* ------------------------------------------------------------------------ */
/* multvec: y = beta*y + X q  (tall-skinny GEMV), q pre-scaled by alpha        */
/* ------------------------------------------------------------------------ */
struct ksg_qarg { double v[KSG_QMAX]; };

template <int VEC>
__global__ __launch_bounds__(KSG_BLOCK) void k_multvec(const double *__restrict__ X, int64_t ld, int k,
                                                        ksg_qarg q, double beta, double *__restrict__ y,
                                                        int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x;
  if (VEC == 2) {
    const int64_t npair = n >> 1;
    double2 *y2 = reinterpret_cast<double2 *>(y);
    for (; p < npair; p += stride) {
      double2 a = make_double2(0.0, 0.0);
      if (beta != 0.0) { const double2 yv = y2[p]; a.x = beta * yv.x; a.y = beta * yv.y; }
      const double2 *xp = reinterpret_cast<const double2 *>(X) + p;
      const int64_t ld2 = ld >> 1;
      int j = 0;
      for (; j + 8 <= k; j += 8) {
        double2 xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = xp[(int64_t)(j + u) * ld2];
#pragma unroll
        for (int u = 0; u < 8; u++) { a.x = fma(xv[u].x, q.v[j + u], a.x); a.y = fma(xv[u].y, q.v[j + u], a.y); }
      }
      for (; j < k; j++) {
        const double2 xv = xp[(int64_t)j * ld2];
        a.x = fma(xv.x, q.v[j], a.x); a.y = fma(xv.y, q.v[j], a.y);
      }
      y2[p] = a;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const int64_t r = n - 1;
      double a = (beta != 0.0) ? beta * y[r] : 0.0;
      for (int j = 0; j < k; j++) a = fma(X[(int64_t)j * ld + r], q.v[j], a);
      y[r] = a;
    }
  } else {
    for (; p < n; p += stride) {
      double a = (beta != 0.0) ? beta * y[p] : 0.0;
      int j = 0;
      for (; j + 8 <= k; j += 8) {
        double xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = X[(int64_t)(j + u) * ld + p];
#pragma unroll
        for (int u = 0; u < 8; u++) a = fma(xv[u], q.v[j + u], a);
      }
      for (; j < k; j++) a = fma(X[(int64_t)j * ld + p], q.v[j], a);
      y[p] = a;
    }
  }
}

static int launch_multvec(ksg_ctx c, const double *X, int64_t ld, int k, const double *q_host,
                          double alpha, double beta, double *y, int64_t n) {
  if (n <= 0) return KSG_OK;
  const bool vec2 = aligned16(X) && aligned16(y) && (ld % 2 == 0);
  const int nb = ksg_grid_for(vec2 ? (n + 1) / 2 : n);
  int done = 0;
  do {   /* at most KSG_QMAX coefficients travel with one launch */
    const int kk = (k - done < KSG_QMAX) ? k - done : KSG_QMAX;
    ksg_qarg q;
    for (int j = 0; j < kk; j++) q.v[j] = alpha * q_host[done + j];
    const double b = done ? 1.0 : beta;
    const double *Xc = X + (int64_t)done * ld;
    if (vec2) hipLaunchKernelGGL((k_multvec<2>), dim3(nb), dim3(KSG_BLOCK), 0, c->stream, Xc, ld, kk, q, b, y, n);
    else      hipLaunchKernelGGL((k_multvec<1>), dim3(nb), dim3(KSG_BLOCK), 0, c->stream, Xc, ld, kk, q, b, y, n);
    done += kk;
  } while (done < k);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* ------------------------------------------------------------------------ */
/* multinplace: V(:,s:e) = V(:,l:k) * B,  B = K x ne                           */
/* ------------------------------------------------------------------------ */
/* One thread owns R rows and all ne <= E output columns of them: it streams its
 * K input values once, multiplies by the coefficient panel held in LDS (read as
 * wave-uniform broadcasts) and only then stores, so the update is in place with
 * no scratch copy of the basis.  Bq is the K x E panel, row-major, zero padded. */
#define KSG_MIP_LDS 4096   /* doubles of the panel resident in LDS at a time */

template <int E, int R>
__global__ __launch_bounds__(KSG_BLOCK) void k_multinplace(const double *__restrict__ Vin, int64_t ld, int K,
                                                            const double *__restrict__ Bq,
                                                            double *__restrict__ Vout, int64_t ldo, int ne,
                                                            int64_t n) {
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
          for (; i + 4 <= kn; i += 4) {
            double2 xv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) xv[u] = xp[(int64_t)(i + u) * ld2];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
              for (int e = 0; e < E; e++) {
                const double b = Bs[(i + u) * E + e];
                acc[0][e] = fma(xv[u].x, b, acc[0][e]);
                acc[R - 1][e] = fma(xv[u].y, b, acc[R - 1][e]);
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
          for (int i = 0; i < kn; i++) {
            const double xv = Vin[(int64_t)(kt + i) * ld + r0];
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
          if (e < ne) *reinterpret_cast<double2 *>(Vout + (int64_t)e * ldo + r0) = make_double2(acc[0][e], acc[R - 1][e]);
      } else {
#pragma unroll
        for (int e = 0; e < E; e++)
          if (e < ne) Vout[(int64_t)e * ldo + r0] = acc[0][e];
      }
    }
  }
}

template <int E, int R>
static void launch_mip(ksg_ctx c, const double *Vin, int64_t ld, int K, const double *Bq, double *Vout,
                       int64_t ldo, int ne, int64_t n) {
  const int nb = ksg_grid_for((n + R - 1) / R);
  hipLaunchKernelGGL((k_multinplace<E, R>), dim3(nb), dim3(KSG_BLOCK), 0, c->stream, Vin, ld, K, Bq, Vout, ldo, ne, n);
}

/* out(:,0:ne) = Vin(:,0:K) * B, B given on the device as K x E row-major panel. */
static int mip_panel(ksg_ctx c, const double *Vin, int64_t ld, int K, const double *Bq, int E, double *Vout,
                     int64_t ldo, int ne, int64_t n) {
  const bool two = aligned16(Vin) && aligned16(Vout) && (ld % 2 == 0) && (ldo % 2 == 0);
  switch (E) {
    case 8:  if (two) launch_mip<8, 2>(c, Vin, ld, K, Bq, Vout, ldo, ne, n);  else launch_mip<8, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n);  break;
    case 16: if (two) launch_mip<16, 2>(c, Vin, ld, K, Bq, Vout, ldo, ne, n); else launch_mip<16, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n); break;
    case 32: if (two) launch_mip<32, 2>(c, Vin, ld, K, Bq, Vout, ldo, ne, n); else launch_mip<32, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n); break;
    default: launch_mip<64, 1>(c, Vin, ld, K, Bq, Vout, ldo, ne, n); break;
  }
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

static inline int mip_width(int ne) { return ne <= 8 ? 8 : ne <= 16 ? 16 : ne <= 32 ? 32 : 64; }

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

/* y = beta*y + alpha*x over a block of columns (BVAXPY_BLAS_Private, bvblas.c:179) */
__global__ __launch_bounds__(KSG_BLOCK) void k_axpby_cols(double *__restrict__ Y, int64_t ldy, const double *__restrict__ X,
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
  bv->ld = ((n + 31) / 32) * 32;
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
  hipStreamSynchronize(bv->ctx->stream);
  hipFree(bv->d);
  delete bv;
  return KSG_OK;
}

extern "C" int ksg_bv_resize(ksg_bv bv, int m, int copy) {
  BV_CHECK(bv, "ksg_bv_resize");
  if (m < 1) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_resize: m=%d", m);
  if (m == bv->m) return KSG_OK;
  ksg_ctx c = bv->ctx;
  KSG_HIP(hipSetDevice(c->device));
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
  bv->d = nd;
  bv->m = m;
  if (bv->k > m) bv->k = m;
  if (bv->l > bv->k) bv->l = bv->k;
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

static int dotvec_impl(ksg_bv X, const double *y, double *q_host, int global) {
  BV_CHECK(X, "ksg_bv_dotvec");
  if (!y || !q_host) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_dotvec: NULL vector");
  KSG_CHK(check_active(X, "ksg_bv_dotvec"));
  ksg_ctx c = X->ctx;
  const int k = X->k - X->l;
  if (k == 0) return KSG_OK;
  if (k > KSG_RED_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_bv_dotvec: %d active columns", k);
  KSG_CHK(ksg_op_begin(c, KSG_EV_DOTVEC));
  KSG_CHK(ksg_launch_dotvec(c, bv_col(X, X->l), X->ld, k, y, X->n, c->d_red));
  KSG_CHK(ksg_op_end(c, KSG_EV_DOTVEC, 8.0 * (double)X->n * (k + 1), 2.0 * (double)X->n * k));
  if (global) return ksg_comm_allreduce(c, c->d_red, q_host, k, 0);
  KSG_HIP(hipMemcpyAsync(c->h_red, c->d_red, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  memcpy(q_host, c->h_red, (size_t)k * sizeof(double));
  return KSG_OK;
}

extern "C" int ksg_bv_dotvec(ksg_bv X, const double *y, double *q) { return dotvec_impl(X, y, q, 1); }
extern "C" int ksg_bv_dotvec_local(ksg_bv X, const double *y, double *q) { return dotvec_impl(X, y, q, 0); }

extern "C" int ksg_bv_multvec(ksg_bv X, double alpha, double beta, double *y, const double *q_host) {
  BV_CHECK(X, "ksg_bv_multvec");
  if (!y) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_multvec: NULL vector");
  KSG_CHK(check_active(X, "ksg_bv_multvec"));
  ksg_ctx c = X->ctx;
  const int k = X->k - X->l;
  KSG_CHK(ksg_op_begin(c, KSG_EV_MULTVEC));
  if (k == 0) {
    if (beta != 1.0 && X->n > 0) {
      if (beta == 0.0) hipLaunchKernelGGL(k_set_cols, dim3(ksg_grid_for(X->n), 1), dim3(KSG_BLOCK), 0, c->stream, y, X->ld, X->n, 0.0);
      else hipLaunchKernelGGL(k_axpby_cols, dim3(ksg_grid_for(X->n), 1), dim3(KSG_BLOCK), 0, c->stream, y, X->ld, y, X->ld, X->n, 0.0, beta);
    }
  } else {
    if (!q_host) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_multvec: NULL coefficients");
    KSG_CHK(launch_multvec(c, bv_col(X, X->l), X->ld, k, q_host, alpha, beta, y, X->n));
  }
  KSG_CHK(ksg_op_end(c, KSG_EV_MULTVEC, 8.0 * (double)X->n * (k + 2), 2.0 * (double)X->n * k));
  return KSG_OK;
}

/* pinned staging for small host matrices */
static double *g_hq = NULL;
static size_t g_hq_cap = 0;
static int ensure_hq(size_t doubles) {
  if (g_hq_cap >= doubles) return KSG_OK;
  if (g_hq) KSG_HIP(hipHostFree(g_hq));
  g_hq = NULL; g_hq_cap = 0;
  KSG_HIP(hipHostMalloc(&g_hq, (doubles + 1024) * sizeof(double), hipHostMallocDefault));
  g_hq_cap = doubles + 1024;
  return KSG_OK;
}

/* out(:, 0:ne) = Vin(:,0:K) * B where B[i][c] is produced by `coef(i,c)`; handles
 * any ne by panels of 64 columns through a scratch copy when it cannot be done
 * in place. */
template <class F>
static int mip_general(ksg_ctx c, const double *Vin, int64_t ld, int K, int ne, F coef, double *Vout,
                       int64_t ldo, int64_t n, bool inplace_overlap) {
  if (ne <= 0 || n <= 0) return KSG_OK;
  if (ne <= 64) {
    const int E = mip_width(ne);
    KSG_CHK(ksg_ensure_q(c, (size_t)K * E));
    KSG_HIP(hipStreamSynchronize(c->stream));
    KSG_CHK(ensure_hq((size_t)K * E));
    for (int i = 0; i < K; i++)
      for (int e = 0; e < E; e++) g_hq[(size_t)i * E + e] = e < ne ? coef(i, e) : 0.0;
    KSG_HIP(hipMemcpyAsync(c->d_q, g_hq, (size_t)K * E * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return mip_panel(c, Vin, ld, K, c->d_q, E, Vout, ldo, ne, n);
  }
  /* wide result: panels of 64 columns; via scratch if the output overlaps the input */
  const int npan = (ne + 63) / 64;
  double *dst = Vout;
  int64_t ldd = ldo;
  if (inplace_overlap) {
    KSG_CHK(ksg_ensure_work(c, (size_t)ld * ne));
    dst = c->d_work; ldd = ld;
  }
  KSG_CHK(ksg_ensure_q(c, (size_t)K * 64 * npan));
  KSG_HIP(hipStreamSynchronize(c->stream));
  KSG_CHK(ensure_hq((size_t)K * 64 * npan));
  for (int pnl = 0; pnl < npan; pnl++) {
    const int w = (ne - pnl * 64 < 64) ? ne - pnl * 64 : 64;
    double *h = g_hq + (size_t)pnl * K * 64;
    for (int i = 0; i < K; i++)
      for (int e = 0; e < 64; e++) h[(size_t)i * 64 + e] = e < w ? coef(i, pnl * 64 + e) : 0.0;
  }
  KSG_HIP(hipMemcpyAsync(c->d_q, g_hq, (size_t)K * 64 * npan * sizeof(double), hipMemcpyHostToDevice, c->stream));
  for (int pnl = 0; pnl < npan; pnl++) {
    const int w = (ne - pnl * 64 < 64) ? ne - pnl * 64 : 64;
    KSG_CHK(mip_panel(c, Vin, ld, K, c->d_q + (size_t)pnl * K * 64, 64, dst + (int64_t)pnl * 64 * ldd, ldd, w, n));
  }
  if (inplace_overlap)
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
  /* column by column through the GEMV kernel: Y(:,ly+j) = beta*Y(:,ly+j) + alpha*X*Q(lx:kx, ly+j) */
  for (int j = 0; j < ky; j++)
    KSG_CHK(launch_multvec(c, bv_col(X, X->l), X->ld, kx, Q + (size_t)(Y->l + j) * ldq + X->l, alpha, beta,
                           bv_col(Y, Y->l + j), Y->n));
  KSG_CHK(ksg_op_end(c, KSG_EV_MULT, 8.0 * (double)Y->n * ky * (kx + 2), 2.0 * (double)Y->n * ky * kx));
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
  /* M(ly+i, lx+j) = Y(:,ly+i)^T X(:,lx+j): one GEMV^T per column of X, one reduction for all */
  for (int j = 0; j < kx; j++)
    KSG_CHK(ksg_launch_dotvec(c, bv_col(Y, Y->l), Y->ld, ky, bv_col(X, X->l + j), X->n, c->d_red + (size_t)j * ky));
  KSG_CHK(ksg_op_end(c, KSG_EV_DOT, 8.0 * (double)X->n * kx * (ky + 1), 2.0 * (double)X->n * kx * ky));
  std::vector<double> h((size_t)kx * ky);
  KSG_CHK(ksg_comm_allreduce(c, c->d_red, h.data(), kx * ky, 0));
  for (int j = 0; j < kx; j++)
    for (int i = 0; i < ky; i++) M[(size_t)(Y->l + i) + (size_t)(X->l + j) * ldm] = h[(size_t)j * ky + i];
  return KSG_OK;
}
