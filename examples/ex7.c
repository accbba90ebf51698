/* Generalized eigenproblem A x = lambda B x with both matrices read from files in PETSc binary
 * format: the program of the reference's tutorial src/eps/examples/tutorials/ex7.c written
 * against the host interface of this package.  With the default transformation the solver
 * works with B^-1 A; the systems with B are solved by the ST's linear solver.
 *
 *   ex7 -f1 <A.petsc> [-f2 <B.petsc>] [-eps_nev k] [-st_type sinvert -eps_target t] [-st_ksp_type gmres] ...
 */
#include "common.h"

int main(int argc, char **argv) {
  ksg_ctx ctx;
  Mat A, B = NULL;
  EPS eps;
  ST st;
  PetscInt its;
  PetscInt64 lits;
  const int f1 = opt_find(argc, argv, "-f1"), f2 = opt_find(argc, argv, "-f2");

  printf("\nGeneralized eigenproblem stored in file.\n\n");
  if (!f1 || f1 + 1 >= argc) { fprintf(stderr, "Must indicate a file name for matrix A with the -f1 option\n"); return 1; }
  CHKG(ksg_ctx_create((int)opt_int(argc, argv, "-device", 0), &ctx));

  printf(" Reading REAL matrices from binary files...\n");
  CHK(MatLoadAIJHIP(ctx, argv[f1 + 1], 0, 1, &A));
  if (f2 && f2 + 1 < argc) CHK(MatLoadAIJHIP(ctx, argv[f2 + 1], 0, 1, &B));
  else printf(" Matrix B was not provided, setting B=I\n\n");

  CHK(EPSCreate(ctx, &eps));
  CHK(EPSSetOperators(eps, A, B));
  eps_set_from_options(eps, argc, argv);

  CHK(EPSSolve(eps));
  CHK(EPSGetIterationNumber(eps, &its));
  printf(" Number of iterations of the method: %d\n", (int)its);
  CHK(EPSGetST(eps, &st));
  CHK(STGetKSPIterations(st, &lits, NULL));
  printf(" Number of linear iterations of the method: %lld\n", (long long)lits);
  eps_report_ex(eps, 1);

  CHK(EPSDestroy(&eps));
  CHK(MatDestroy(&A));
  CHK(MatDestroy(&B));
  CHKG(ksg_ctx_destroy(ctx));
  return 0;
}
