/* Eigenvalues of largest real part of the transition matrix of a Markov chain (a random
 * walk on a triangular grid with m points per side), a nonsymmetric problem: the program
 * of the reference's tutorial src/eps/examples/tutorials/ex5.c written against the host
 * interface of this package.  The start vector has ones in its first two entries, as
 * there.
 *
 *   ex5 [-m <points per side>] [-eps_nev k] [-eps_ncv k] [-eps_tol t] ...
 */
#include "common.h"

int main(int argc, char **argv) {
  const long m = opt_int(argc, argv, "-m", 15);
  const int64_t N = (int64_t)m * (m + 1) / 2;
  ksg_ctx ctx;
  Mat A;
  EPS eps;
  Vec v0;

  printf("\nMarkov Model, N=%lld (m=%ld)\n\n", (long long)N, m);
  CHKG(ksg_ctx_create((int)opt_int(argc, argv, "-device", 0), &ctx));

  /* Transition probabilities: from grid point (i,j), 1 <= j <= m-i+1, the walk moves "down"
   * (to (i,j+1) or (i+1,j)) with probability pd each and "up" (to (i,j-1) or (i-1,j)) with
   * probability pu each; on the two straight edges the downward moves count double.  The
   * matrix holds the probability of arriving at the row's point from the column's point. */
  const double cst = 0.5 / (double)(m - 1);
  int64_t *cnt = calloc((size_t)(N + 1), sizeof(*cnt));
  int64_t cap = 4 * N;
  int64_t *er = malloc((size_t)cap * sizeof(*er)), *ec = malloc((size_t)cap * sizeof(*ec));
  double *ev = malloc((size_t)cap * sizeof(*ev));
  if (!cnt || !er || !ec || !ev) { fprintf(stderr, "out of memory\n"); return 1; }
  int64_t ne = 0, ix = 0;
  for (long i = 1; i <= m; i++) {
    const long jmax = m - i + 1;
    for (long j = 1; j <= jmax; j++) {
      ix++;                                          /* 1-based number of point (i,j) */
      if (j != jmax) {
        const double pd = cst * (double)(i + j - 1);
        er[ne] = ix - 1; ec[ne] = ix;            ev[ne++] = (i == 1) ? 2.0 * pd : pd;   /* from (i,j+1) */
        er[ne] = ix - 1; ec[ne] = ix + jmax - 1; ev[ne++] = (j == 1) ? 2.0 * pd : pd;   /* from (i+1,j) */
      }
      const double pu = 0.5 - cst * (double)(i + j - 3);
      if (j > 1) { er[ne] = ix - 1; ec[ne] = ix - 2;        ev[ne++] = pu; }            /* from (i,j-1) */
      if (i > 1) { er[ne] = ix - 1; ec[ne] = ix - jmax - 2; ev[ne++] = pu; }            /* from (i-1,j) */
    }
  }
  /* to CSR with ascending columns: count, prefix, scatter, then sort each short row */
  for (int64_t e = 0; e < ne; e++) cnt[er[e] + 1]++;
  for (int64_t r = 0; r < N; r++) cnt[r + 1] += cnt[r];
  int64_t *rowptr = malloc((size_t)(N + 1) * sizeof(*rowptr));
  int64_t *fill = malloc((size_t)N * sizeof(*fill));
  int32_t *colidx = malloc((size_t)ne * sizeof(*colidx));
  double *val = malloc((size_t)ne * sizeof(*val));
  memcpy(rowptr, cnt, (size_t)(N + 1) * sizeof(*rowptr));
  memcpy(fill, cnt, (size_t)N * sizeof(*fill));
  for (int64_t e = 0; e < ne; e++) { const int64_t q = fill[er[e]]++; colidx[q] = (int32_t)ec[e]; val[q] = ev[e]; }
  for (int64_t r = 0; r < N; r++)
    for (int64_t a = rowptr[r] + 1; a < rowptr[r + 1]; a++) {
      const int32_t c = colidx[a]; const double v = val[a];
      int64_t b = a - 1;
      while (b >= rowptr[r] && colidx[b] > c) { colidx[b + 1] = colidx[b]; val[b + 1] = val[b]; b--; }
      colidx[b + 1] = c; val[b + 1] = v;
    }
  CHK(MatCreateAIJHIP(ctx, N, N, 0, rowptr, colidx, val, &A));
  free(cnt); free(er); free(ec); free(ev); free(rowptr); free(fill); free(colidx); free(val);

  CHK(EPSCreate(ctx, &eps));
  CHK(EPSSetOperators(eps, A, NULL));
  CHK(EPSSetProblemType(eps, EPS_NHEP));
  eps_set_from_options(eps, argc, argv);

  /* initial vector: ones in the first two positions */
  CHK(MatCreateVecs(A, &v0, NULL));
  double *h = calloc((size_t)N, sizeof(*h));
  h[0] = 1.0; if (N > 1) h[1] = 1.0;
  CHK(VecSetValuesHost(v0, h));
  free(h);
  CHK(EPSSetInitialSpace(eps, 1, &v0));

  CHK(EPSSolve(eps));
  eps_report(eps);

  CHK(EPSDestroy(&eps));
  CHK(VecDestroy(&v0));
  CHK(MatDestroy(&A));
  CHKG(ksg_ctx_destroy(ctx));
  return 0;
}
