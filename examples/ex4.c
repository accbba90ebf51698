/* Standard eigenproblem with the matrix read from a file in PETSc binary format: the
 * program of the reference's tutorial src/eps/examples/tutorials/ex4.c written against
 * the host interface of this package.  The matrix need not be symmetric, so the problem
 * is solved as a non-Hermitian one unless -eps_hermitian is given.
 *
 *   ex4 -file <matrix.petsc> [-eps_nev k] [-eps_ncv k] [-eps_tol t] [-terse] ...
 */
#include "common.h"

int main(int argc, char **argv) {
  ksg_ctx ctx;
  Mat A;
  EPS eps;
  PetscInt its;
  const int fi = opt_find(argc, argv, "-file");

  printf("\nEigenproblem stored in file.\n\n");
  if (!fi || fi + 1 >= argc) { fprintf(stderr, "Must indicate a file name with the -file option\n"); return 1; }
  CHKG(ksg_ctx_create((int)opt_int(argc, argv, "-device", 0), &ctx));

  printf(" Reading REAL matrix from a binary file...\n");
  CHK(MatLoadAIJHIP(ctx, argv[fi + 1], 0, 1, &A));

  CHK(EPSCreate(ctx, &eps));
  CHK(EPSSetOperators(eps, A, NULL));
  if (opt_find(argc, argv, "-eps_hermitian")) CHK(EPSSetProblemType(eps, EPS_HEP));
  eps_set_from_options(eps, argc, argv);

  CHK(EPSSolve(eps));
  CHK(EPSGetIterationNumber(eps, &its));
  printf(" Number of iterations of the method: %d\n", (int)its);
  eps_report_ex(eps, 1);

  CHK(EPSDestroy(&eps));
  CHK(MatDestroy(&A));
  CHKG(ksg_ctx_destroy(ctx));
  return 0;
}
