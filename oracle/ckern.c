/* TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product (slepc_amd/).
 *
 * Compiled, row-parallel CPU restatement of the vector kernels of the reference's
 * Krylov-Schur path, used (a) as the checker's fast back end and (b) as the CPU baseline
 * that bench.py times on the host cores.  Rows are split into one contiguous block per
 * thread, the way the reference splits them over MPI ranks; a reduction adds the threads'
 * partial results in thread order, the role of MPI_Allreduce in
 * src/sys/classes/bv/interface/bvblas.c:252-260,303-306.
 *
 *   ck_spmv        y = A x, CSR ("AIJ"): the loop of PETSc's MatMult_SeqAIJ that
 *                  STApply_Shift reaches (src/sys/classes/st/impls/shift/shift.c:38);
 *                  PETSc is not in the reference tree, the loop is the textbook one:
 *                  every row's products added in storage order.
 *   ck_dotvec      q = X^T y    BVDotVec_BLAS_Private, bvblas.c:242-266 (gemv 'C')
 *   ck_multvec     y = beta y + alpha X q   BVMultVec_BLAS_Private, bvblas.c:65-76 (gemv 'N')
 *   ck_multinplace V(:,s:e) = V Q(:,s:e) in blocks of 64 rows through a 64 x (e-s) scratch,
 *                  BVMultInPlace_BLAS_Private, bvblas.c:83-119
 *   ck_scale       BVScale_BLAS_Private, bvblas.c:271-285
 *   ck_sumsq       local part of BVNorm_LAPACK_Private (Frobenius), bvblas.c:290-306
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CK_CHUNK 512            /* rows handled per pass over the columns: y stays in L1 */
#define CK_BS    64             /* BLOCKSIZE of bvblas.c:27 */

static void row_block(int64_t n, int t, int nt, int64_t *r0, int64_t *r1) {
  /* PetscSplitOwnership: the first n % nt blocks get one row more */
  const int64_t base = n / nt, extra = n % nt;
  *r0 = t * base + (t < extra ? t : extra);
  *r1 = *r0 + base + (t < extra ? 1 : 0);
}

int ck_max_threads(void) { return omp_get_max_threads(); }

#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")      /* products and sums rounded separately, as a plain C loop does */
void ck_spmv(int nthreads, int64_t n, const int64_t *rowptr, const int32_t *colidx, const double *val,
             const double *x, double *y) {
#pragma omp parallel num_threads(nthreads)
  {
    int64_t r0, r1;
    row_block(n, omp_get_thread_num(), omp_get_num_threads(), &r0, &r1);
    for (int64_t i = r0; i < r1; i++) {
      double s = 0.0;
      for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++) s += val[p] * x[colidx[p]];
      y[i] = s;
    }
  }
}
#pragma GCC pop_options

/* work: nthreads * k doubles */
void ck_dotvec(int nthreads, int64_t n, int k, const double *X, int64_t ld, const double *y, double *q, double *work) {
  int used = 1;
#pragma omp parallel num_threads(nthreads)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    int64_t r0, r1;
    row_block(n, t, nt, &r0, &r1);
    double *acc = work + (size_t)t * k;
    for (int c = 0; c < k; c++) acc[c] = 0.0;
    for (int64_t b = r0; b < r1; b += CK_CHUNK) {
      const int64_t e = b + CK_CHUNK < r1 ? b + CK_CHUNK : r1;
      for (int c = 0; c < k; c++) {
        const double *col = X + (size_t)c * ld;
        /* eight running sums, so that the compiler may keep them in vector registers */
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t i = b;
        for (; i + 8 <= e; i += 8)
          for (int u = 0; u < 8; u++) s[u] += col[i + u] * y[i + u];
        for (; i < e; i++) s[0] += col[i] * y[i];
        acc[c] += ((s[0] + s[4]) + (s[1] + s[5])) + ((s[2] + s[6]) + (s[3] + s[7]));
      }
    }
#pragma omp single
    used = nt;
  }
  for (int c = 0; c < k; c++) {
    double s = 0.0;
    for (int t = 0; t < used; t++) s += work[(size_t)t * k + c];
    q[c] = s;
  }
}

void ck_multvec(int nthreads, int64_t n, int k, double alpha, double beta, const double *X, int64_t ld, const double *q, double *y) {
#pragma omp parallel num_threads(nthreads)
  {
    int64_t r0, r1;
    row_block(n, omp_get_thread_num(), omp_get_num_threads(), &r0, &r1);
    double tmp[CK_CHUNK];
    for (int64_t b = r0; b < r1; b += CK_CHUNK) {
      const int64_t e = b + CK_CHUNK < r1 ? b + CK_CHUNK : r1;
      const int m = (int)(e - b);
      for (int i = 0; i < m; i++) tmp[i] = 0.0;
      for (int c = 0; c < k; c++) {
        const double *col = X + (size_t)c * ld + b;
        const double qc = q[c];
        for (int i = 0; i < m; i++) tmp[i] += col[i] * qc;
      }
      if (beta == 0.0) for (int i = 0; i < m; i++) y[b + i] = alpha * tmp[i];
      else             for (int i = 0; i < m; i++) y[b + i] = beta * y[b + i] + alpha * tmp[i];
    }
  }
}

/* V(:, s:e) = V(:, 0:k) * B(0:k, s:e); B column-major with leading dimension ldb */
void ck_multinplace(int nthreads, int64_t n, int k, double *V, int64_t ld, const double *B, int ldb, int s, int e) {
  const int w = e - s;
  if (w <= 0 || n <= 0) return;
#pragma omp parallel num_threads(nthreads)
  {
    int64_t r0, r1;
    row_block(n, omp_get_thread_num(), omp_get_num_threads(), &r0, &r1);
    double *scr = (double *)malloc((size_t)CK_BS * w * sizeof(double));
    for (int64_t b = r0; b < r1; b += CK_BS) {
      const int m = (int)((b + CK_BS < r1 ? b + CK_BS : r1) - b);
      for (int j = 0; j < w; j++) {
        double *out = scr + (size_t)j * CK_BS;
        const double *bj = B + (size_t)(s + j) * ldb;
        for (int i = 0; i < m; i++) out[i] = 0.0;
        for (int c = 0; c < k; c++) {
          const double *col = V + (size_t)c * ld + b;
          const double bc = bj[c];
          for (int i = 0; i < m; i++) out[i] += col[i] * bc;
        }
      }
      for (int j = 0; j < w; j++) memcpy(V + (size_t)(s + j) * ld + b, scr + (size_t)j * CK_BS, (size_t)m * sizeof(double));
    }
    free(scr);
  }
}

void ck_scale(int nthreads, int64_t n, double alpha, double *x) {
#pragma omp parallel num_threads(nthreads)
  {
    int64_t r0, r1;
    row_block(n, omp_get_thread_num(), omp_get_num_threads(), &r0, &r1);
    if (alpha == 0.0) memset(x + r0, 0, (size_t)(r1 - r0) * sizeof(double));
    else for (int64_t i = r0; i < r1; i++) x[i] *= alpha;
  }
}

void ck_copy(int nthreads, int64_t n, const double *x, double *y) {
#pragma omp parallel num_threads(nthreads)
  {
    int64_t r0, r1;
    row_block(n, omp_get_thread_num(), omp_get_num_threads(), &r0, &r1);
    memcpy(y + r0, x + r0, (size_t)(r1 - r0) * sizeof(double));
  }
}

/* work: nthreads doubles */
double ck_sumsq(int nthreads, int64_t n, const double *x, double *work) {
  int used = 1;
#pragma omp parallel num_threads(nthreads)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    int64_t r0, r1;
    row_block(n, t, nt, &r0, &r1);
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = r0;
    for (; i + 8 <= r1; i += 8)
      for (int u = 0; u < 8; u++) s[u] += x[i + u] * x[i + u];
    for (; i < r1; i++) s[0] += x[i] * x[i];
    work[t] = ((s[0] + s[4]) + (s[1] + s[5])) + ((s[2] + s[6]) + (s[3] + s[7]));
#pragma omp single
    used = nt;
  }
  double s = 0.0;
  for (int t = 0; t < used; t++) s += work[t];
  return s;
}

/* a[i] = b[i] + scalar * c[i]: the STREAM triad, 24 bytes per element, to state what the host memory gives */
void ck_triad(int nthreads, int64_t n, double *a, const double *b, const double *c, double scalar) {
#pragma omp parallel num_threads(nthreads)
  {
    int64_t r0, r1;
    row_block(n, omp_get_thread_num(), omp_get_num_threads(), &r0, &r1);
    for (int64_t i = r0; i < r1; i++) a[i] = b[i] + scalar * c[i];
  }
}
