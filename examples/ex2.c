/* Standard symmetric eigenproblem of the 2-D Laplacian on an n x m grid, solved with the
 * Krylov-Schur solver on one MI355X: the program of the reference's tutorial
 * src/eps/examples/tutorials/ex2.c written against the host interface of this package
 * (include/kseps.h).  The matrix has 4 on the diagonal and -1 for each of the four grid
 * neighbours of a point; rows are numbered row by row of the grid.
 *
 *   ex2 [-n <points in x>] [-m <points in y>] [-eps_nev k] [-eps_ncv k] [-eps_tol t] ...
 */
#include "common.h"

int main(int argc, char **argv) {
  const long n = opt_int(argc, argv, "-n", 10);
  const long m = opt_int(argc, argv, "-m", n);
  const int64_t N = (int64_t)n * m;
  ksg_ctx ctx;
  Mat A;
  EPS eps;

  printf("\n2-D Laplacian Eigenproblem, N=%lld (%ldx%ld grid)\n\n", (long long)N, n, m);
  CHKG(ksg_ctx_create((int)opt_int(argc, argv, "-device", 0), &ctx));

  /* CSR of the operator, columns ascending within a row */
  int64_t *rowptr = malloc((size_t)(N + 1) * sizeof(*rowptr));
  int32_t *colidx = malloc((size_t)5 * N * sizeof(*colidx));
  double *val = malloc((size_t)5 * N * sizeof(*val));
  if (!rowptr || !colidx || !val) { fprintf(stderr, "out of memory\n"); return 1; }
  int64_t p = 0;
  for (int64_t row = 0; row < N; row++) {
    const int64_t gi = row / n, gj = row % n;       /* grid coordinates of this unknown */
    rowptr[row] = p;
    if (gi > 0)     { colidx[p] = (int32_t)(row - n); val[p++] = -1.0; }
    if (gj > 0)     { colidx[p] = (int32_t)(row - 1); val[p++] = -1.0; }
    colidx[p] = (int32_t)row; val[p++] = 4.0;
    if (gj < n - 1) { colidx[p] = (int32_t)(row + 1); val[p++] = -1.0; }
    if (gi < m - 1) { colidx[p] = (int32_t)(row + n); val[p++] = -1.0; }
  }
  rowptr[N] = p;
  CHK(MatCreateAIJHIP(ctx, N, N, 0, rowptr, colidx, val, &A));
  free(rowptr); free(colidx); free(val);

  CHK(EPSCreate(ctx, &eps));
  CHK(EPSSetOperators(eps, A, NULL));
  CHK(EPSSetProblemType(eps, EPS_HEP));
  eps_set_from_options(eps, argc, argv);
  CHK(EPSSolve(eps));
  eps_report(eps);

  CHK(EPSDestroy(&eps));
  CHK(MatDestroy(&A));
  CHKG(ksg_ctx_destroy(ctx));
  return 0;
}
