/* Singular values of a matrix read from a file in PETSc binary format: the program of the
 * reference's tutorial src/svd/examples/tutorials/ex14.c written against the host interface
 * of this package.
 *
 *   ex14 -file <matrix.petsc> [-svd_nsv k] [-svd_ncv k] [-svd_type cross|trlanczos] [-terse] ...
 */
#include "common.h"

int main(int argc, char **argv) {
  ksg_ctx ctx;
  Mat A;
  SVD svd;
  PetscInt its, nsv, maxit, nconv;
  PetscReal tol, sigma, err;
  const char *type;
  const int fi = opt_find(argc, argv, "-file");

  printf("\nSingular value problem stored in file.\n\n");
  if (!fi || fi + 1 >= argc) { fprintf(stderr, "Must indicate a file name with the -file option\n"); return 1; }
  CHKG(ksg_ctx_create((int)opt_int(argc, argv, "-device", 0), &ctx));

  printf(" Reading REAL matrix from a binary file...\n");
  CHK(MatLoadAIJHIP(ctx, argv[fi + 1], 0, 1, &A));

  CHK(SVDCreate(ctx, &svd));
  CHK(SVDSetOperator(svd, A));
  svd_set_from_options(svd, argc, argv);

  CHK(SVDSolve(svd));
  CHK(SVDGetIterationNumber(svd, &its));
  printf(" Number of iterations of the method: %d\n", (int)its);
  CHK(SVDGetType(svd, &type));
  printf(" Solution method: %s\n\n", type);
  CHK(SVDGetDimensions(svd, &nsv, NULL, NULL));
  printf(" Number of requested singular values: %d\n", (int)nsv);
  CHK(SVDGetTolerances(svd, &tol, &maxit));
  printf(" Stopping condition: tol=%.4g, maxit=%d\n", tol, (int)maxit);

  /* SVDErrorView(svd,SVD_ERROR_RELATIVE,NULL) */
  CHK(SVDGetConverged(svd, &nconv));
  if (nconv < nsv) { printf(" Problem: less than %d singular values converged\n\n", (int)nsv); }
  else {
    int ok = 1;
    for (PetscInt i = 0; i < nsv && ok; i++) { CHK(SVDComputeError(svd, i, SVD_ERROR_RELATIVE, &err)); if (err > 5.0 * tol) ok = 0; }
    if (!ok) printf(" Problem: some of the first %d relative errors are higher than the tolerance\n\n", (int)nsv);
    else {
      printf(" All requested singular values computed up to the required tolerance:");
      for (PetscInt i = 0; i < nsv; i++) {
        CHK(SVDGetSingularTriplet(svd, i, &sigma, NULL, NULL));
        if (i % 8 == 0) printf("\n     ");
        printf("%.5f", sigma);
        if (i + 1 < nsv) printf(", ");
      }
      printf("\n\n");
    }
  }
  CHK(SVDDestroy(&svd));
  CHK(MatDestroy(&A));
  CHKG(ksg_ctx_destroy(ctx));
  return 0;
}
