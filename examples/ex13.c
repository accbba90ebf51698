/* Generalized symmetric eigenproblem A x = lambda B x, A the 2-D Laplacian and B diagonal,
 * possibly singular: the program of the reference's tutorial
 * src/eps/examples/tutorials/ex13.c written against the host interface of this package.
 * The eigenvalues closest to 0 are computed with shift-and-invert, which never solves a
 * system with B.
 *
 *   ex13 [-n nx] [-m ny] [-nulldim k] [-eps_nev k] [-eps_ncv k] [-st_ksp_type cg] ...
 */
#include "common.h"

int main(int argc, char **argv) {
  ksg_ctx ctx;
  Mat A, B;
  EPS eps;
  ST st;
  const PetscInt n = (PetscInt)opt_int(argc, argv, "-n", 10), m = (PetscInt)opt_int(argc, argv, "-m", n);
  const PetscInt nulldim = (PetscInt)opt_int(argc, argv, "-nulldim", 0);
  const PetscInt64 N = (PetscInt64)n * m;

  printf("\nGeneralized Symmetric Eigenproblem, N=%lld (%dx%d grid), null(B)=%d\n\n", (long long)N, (int)n, (int)m, (int)nulldim);
  CHKG(ksg_ctx_create((int)opt_int(argc, argv, "-device", 0), &ctx));

  /* A: 5-point Laplacian (ex13.c:67-74); B: 4 on the diagonal except the first nulldim rows */
  int64_t *rp = (int64_t *)malloc((size_t)(N + 1) * sizeof(*rp)), *rb = (int64_t *)malloc((size_t)(N + 1) * sizeof(*rb));
  int32_t *ci = (int32_t *)malloc((size_t)(5 * N) * sizeof(*ci)), *cb = (int32_t *)malloc((size_t)N * sizeof(*cb));
  double *va = (double *)malloc((size_t)(5 * N) * sizeof(*va)), *vb = (double *)malloc((size_t)N * sizeof(*vb));
  int64_t q = 0, qb = 0;
  rp[0] = rb[0] = 0;
  for (PetscInt64 II = 0; II < N; II++) {
    const PetscInt64 i = II / n, j = II - i * n;
    if (i > 0) { ci[q] = (int32_t)(II - n); va[q++] = -1.0; }
    if (j > 0) { ci[q] = (int32_t)(II - 1); va[q++] = -1.0; }
    ci[q] = (int32_t)II; va[q++] = 4.0;
    if (j < n - 1) { ci[q] = (int32_t)(II + 1); va[q++] = -1.0; }
    if (i < m - 1) { ci[q] = (int32_t)(II + n); va[q++] = -1.0; }
    rp[II + 1] = q;
    if (II >= nulldim) { cb[qb] = (int32_t)II; vb[qb++] = 4.0; }
    rb[II + 1] = qb;
  }
  CHK(MatCreateAIJHIP(ctx, N, N, 0, rp, ci, va, &A));
  CHK(MatCreateAIJHIP(ctx, N, N, 0, rb, cb, vb, &B));
  free(rp); free(rb); free(ci); free(cb); free(va); free(vb);

  CHK(EPSCreate(ctx, &eps));
  CHK(EPSSetOperators(eps, A, B));
  CHK(EPSSetProblemType(eps, EPS_GHEP));
  /* select the portion of the spectrum and the transformation (ex13.c:100-112) */
  CHK(EPSSetTarget(eps, 0.0));
  CHK(EPSSetWhichEigenpairs(eps, EPS_TARGET_MAGNITUDE));
  CHK(EPSGetST(eps, &st));
  CHK(STSetType(st, "sinvert"));
  eps_set_from_options(eps, argc, argv);

  CHK(EPSSolve(eps));
  eps_report(eps);

  CHK(EPSDestroy(&eps));
  CHK(MatDestroy(&A));
  CHK(MatDestroy(&B));
  CHKG(ksg_ctx_destroy(ctx));
  return 0;
}
