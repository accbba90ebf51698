/* Estimates the 2-norm condition number of a Grcar matrix from its largest and smallest
 * singular values: the program of the reference's tutorial src/svd/examples/tutorials/ex8.c
 * written against the host interface of this package.
 *
 *   ex8 [-n <dimension>] [-svd_type cross|trlanczos] [-svd_trlanczos_oneside] [-svd_tol t] ...
 */
#include "common.h"

int main(int argc, char **argv) {
  const long N = opt_int(argc, argv, "-n", 30);
  ksg_ctx ctx;
  Mat A;
  SVD svd;
  PetscInt nconv1, nconv2;
  PetscReal sigma_1 = 0.0, sigma_n = 0.0;

  printf("\nEstimate the condition number of a Grcar matrix, n=%ld\n\n", N);
  CHKG(ksg_ctx_create((int)opt_int(argc, argv, "-device", 0), &ctx));

  /* -1 on the subdiagonal, 1 on the diagonal and on three superdiagonals */
  int64_t *rowptr = malloc((size_t)(N + 1) * sizeof(*rowptr));
  int32_t *colidx = malloc((size_t)(5 * N) * sizeof(*colidx));
  double *val = malloc((size_t)(5 * N) * sizeof(*val));
  if (!rowptr || !colidx || !val) { fprintf(stderr, "out of memory\n"); return 1; }
  int64_t ne = 0;
  rowptr[0] = 0;
  for (long i = 0; i < N; i++) {
    for (long c = i - 1; c <= i + 3; c++)
      if (c >= 0 && c < N) { colidx[ne] = (int32_t)c; val[ne++] = (c == i - 1) ? -1.0 : 1.0; }
    rowptr[i + 1] = ne;
  }
  CHK(MatCreateAIJHIP(ctx, N, N, 0, rowptr, colidx, val, &A));
  free(rowptr); free(colidx); free(val);

  CHK(SVDCreate(ctx, &svd));
  CHK(SVDSetOperator(svd, A));
  svd_set_from_options(svd, argc, argv);
  CHK(SVDSetDimensions(svd, 1, 0, 0));

  /* first one end of the spectrum, then the other */
  CHK(SVDSetWhichSingularTriplets(svd, SVD_LARGEST));
  CHK(SVDSolve(svd));
  CHK(SVDGetConverged(svd, &nconv1));
  if (nconv1 > 0) CHK(SVDGetSingularTriplet(svd, 0, &sigma_1, NULL, NULL));
  else printf(" Unable to compute large singular value!\n\n");

  CHK(SVDSetWhichSingularTriplets(svd, SVD_SMALLEST));
  CHK(SVDSolve(svd));
  CHK(SVDGetConverged(svd, &nconv2));
  if (nconv2 > 0) CHK(SVDGetSingularTriplet(svd, 0, &sigma_n, NULL, NULL));
  else printf(" Unable to compute small singular value!\n\n");

  if (nconv1 > 0 && nconv2 > 0) {
    printf(" Computed singular values: sigma_1=%.4f, sigma_n=%.4f\n", sigma_1, sigma_n);
    printf(" Estimated condition number: sigma_1/sigma_n=%.4f\n\n", sigma_1 / sigma_n);
  }
  CHK(SVDDestroy(&svd));
  CHK(MatDestroy(&A));
  CHKG(ksg_ctx_destroy(ctx));
  return 0;
}
