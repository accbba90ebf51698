/* Helpers shared by the example programs: a minimal command line in the style of the
 * reference's options (-n 72 -eps_nev 4 -eps_ncv 20 -eps_tol 1e-8 -eps_max_it 100
 * -eps_largest_real -eps_type arnoldi ...), error checking, and the report the reference's examples print
 * (EPSErrorView with the default ASCII format, e.g.
 * src/eps/examples/tutorials/output/ex2_1.out). */
#ifndef EXAMPLES_COMMON_H
#define EXAMPLES_COMMON_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "kseps.h"

#define CHK(call) do { PetscErrorCode ierr_ = (call); if (ierr_) { \
    fprintf(stderr, "%s:%d: error %d: %s\n", __FILE__, __LINE__, (int)ierr_, KSGetLastError()); exit(1); } } while (0)
#define CHKG(call) do { int rc_ = (call); if (rc_) { \
    fprintf(stderr, "%s:%d: libksgpu error %d: %s\n", __FILE__, __LINE__, rc_, ksg_last_error()); exit(1); } } while (0)

static int opt_find(int argc, char **argv, const char *name) {
  for (int i = 1; i < argc; i++) if (!strcmp(argv[i], name)) return i;
  return 0;
}
static long opt_int(int argc, char **argv, const char *name, long dflt) {
  int i = opt_find(argc, argv, name);
  return (i && i + 1 < argc) ? atol(argv[i + 1]) : dflt;
}
static double opt_real(int argc, char **argv, const char *name, double dflt) {
  int i = opt_find(argc, argv, name);
  return (i && i + 1 < argc) ? atof(argv[i + 1]) : dflt;
}

/* -eps_nev -eps_ncv -eps_mpd -eps_tol -eps_max_it -eps_largest_real ... (EPSSetFromOptions) */
static void eps_set_from_options(EPS eps, int argc, char **argv) {
  PetscInt nev, ncv, mpd;
  int it = opt_find(argc, argv, "-eps_type");
  if (it && it + 1 < argc) CHK(EPSSetType(eps, argv[it + 1]));       /* krylovschur | arnoldi | lanczos */
  it = opt_find(argc, argv, "-eps_lanczos_reorthog");                  /* local | full | selective | periodic | partial | delayed */
  if (it && it + 1 < argc) {
    static const char *names[] = { "local", "full", "selective", "periodic", "partial", "delayed" };
    int r = -1;
    for (int q = 0; q < 6; q++) if (!strcmp(argv[it + 1], names[q])) r = q;
    if (r < 0) { fprintf(stderr, "unknown reorthogonalization %s\n", argv[it + 1]); exit(1); }
    CHK(EPSLanczosSetReorthog(eps, (EPSLanczosReorthogType)r));
  }
  if (opt_find(argc, argv, "-eps_arnoldi_delayed")) CHK(EPSArnoldiSetDelayed(eps, PETSC_TRUE));
  CHK(EPSGetDimensions(eps, &nev, &ncv, &mpd));
  nev = (PetscInt)opt_int(argc, argv, "-eps_nev", nev);
  CHK(EPSSetDimensions(eps, nev, (PetscInt)opt_int(argc, argv, "-eps_ncv", 0), (PetscInt)opt_int(argc, argv, "-eps_mpd", 0)));
  CHK(EPSSetTolerances(eps, opt_real(argc, argv, "-eps_tol", 0.0), (PetscInt)opt_int(argc, argv, "-eps_max_it", 0)));
  if (opt_find(argc, argv, "-eps_largest_magnitude")) CHK(EPSSetWhichEigenpairs(eps, EPS_LARGEST_MAGNITUDE));
  if (opt_find(argc, argv, "-eps_smallest_magnitude")) CHK(EPSSetWhichEigenpairs(eps, EPS_SMALLEST_MAGNITUDE));
  if (opt_find(argc, argv, "-eps_largest_real")) CHK(EPSSetWhichEigenpairs(eps, EPS_LARGEST_REAL));
  if (opt_find(argc, argv, "-eps_smallest_real")) CHK(EPSSetWhichEigenpairs(eps, EPS_SMALLEST_REAL));
  if (opt_find(argc, argv, "-eps_target_magnitude")) CHK(EPSSetWhichEigenpairs(eps, EPS_TARGET_MAGNITUDE));
  if (opt_find(argc, argv, "-eps_target_real")) CHK(EPSSetWhichEigenpairs(eps, EPS_TARGET_REAL));
  if (opt_find(argc, argv, "-eps_target")) CHK(EPSSetTarget(eps, opt_real(argc, argv, "-eps_target", 0.0)));
  /* -st_type shift|sinvert  -st_shift s  -st_ksp_type preonly|cg|gmres  -st_ksp_rtol t  -st_ksp_gmres_restart m (STSetFromOptions) */
  ST st;
  CHK(EPSGetST(eps, &st));
  it = opt_find(argc, argv, "-st_type");
  if (it && it + 1 < argc) CHK(STSetType(st, argv[it + 1]));
  if (opt_find(argc, argv, "-st_shift")) CHK(STSetShift(st, opt_real(argc, argv, "-st_shift", 0.0)));
  it = opt_find(argc, argv, "-st_ksp_type");
  if (it && it + 1 < argc) CHK(STSetKSPType(st, argv[it + 1]));
  CHK(STSetKSPTolerances(st, opt_real(argc, argv, "-st_ksp_rtol", 0.0), (PetscInt)opt_int(argc, argv, "-st_ksp_max_it", 0),
                         (PetscInt)opt_int(argc, argv, "-st_ksp_gmres_restart", 0)));
}

/* -svd_type -svd_nsv -svd_ncv -svd_mpd -svd_tol -svd_max_it -svd_largest -svd_smallest
 * -svd_trlanczos_oneside (SVDSetFromOptions) */
static void svd_set_from_options(SVD svd, int argc, char **argv) {
  PetscInt nsv, ncv, mpd;
  int it = opt_find(argc, argv, "-svd_type");
  if (it && it + 1 < argc) CHK(SVDSetType(svd, argv[it + 1]));       /* cross | trlanczos | lanczos | cyclic */
  CHK(SVDGetDimensions(svd, &nsv, &ncv, &mpd));
  nsv = (PetscInt)opt_int(argc, argv, "-svd_nsv", nsv);
  CHK(SVDSetDimensions(svd, nsv, (PetscInt)opt_int(argc, argv, "-svd_ncv", 0), (PetscInt)opt_int(argc, argv, "-svd_mpd", 0)));
  CHK(SVDSetTolerances(svd, opt_real(argc, argv, "-svd_tol", 0.0), (PetscInt)opt_int(argc, argv, "-svd_max_it", 0)));
  if (opt_find(argc, argv, "-svd_largest")) CHK(SVDSetWhichSingularTriplets(svd, SVD_LARGEST));
  if (opt_find(argc, argv, "-svd_smallest")) CHK(SVDSetWhichSingularTriplets(svd, SVD_SMALLEST));
  if (opt_find(argc, argv, "-svd_trlanczos_oneside")) CHK(SVDTRLanczosSetOneSide(svd, PETSC_TRUE));
  if (opt_find(argc, argv, "-svd_lanczos_oneside")) CHK(SVDLanczosSetOneSide(svd, PETSC_TRUE));
}

/* the report of EPSErrorView(eps,EPS_ERROR_RELATIVE,NULL), after the solver lines the
 * tutorials print; `stopping` adds the line with the tolerance and the restart cap */
static void eps_report_ex(EPS eps, int stopping) {
  PetscInt nev, ncv, mpd, nconv, its, maxit;
  PetscReal tol, err;
  CHK(EPSGetDimensions(eps, &nev, &ncv, &mpd));
  CHK(EPSGetConverged(eps, &nconv));
  CHK(EPSGetIterationNumber(eps, &its));
  CHK(EPSGetTolerances(eps, &tol, &maxit));
  const char *type;
  CHK(EPSGetType(eps, &type));
  printf(" Solution method: %s\n\n", type);
  printf(" Number of requested eigenvalues: %d\n", (int)nev);
  if (stopping) printf(" Stopping condition: tol=%.4g, maxit=%d\n", tol, (int)maxit);
  if (nconv < nev) {
    printf(" Problem: less than %d eigenvalues converged\n\n", (int)nev);
    return;
  }
  for (PetscInt i = 0; i < nev; i++) {
    CHK(EPSComputeError(eps, i, EPS_ERROR_RELATIVE, &err));
    if (err > 5.0 * tol) {
      printf(" Problem: some of the first %d relative errors are higher than the tolerance\n\n", (int)nev);
      return;
    }
  }
  printf(" All requested eigenvalues computed up to the required tolerance:");
  for (PetscInt i = 0; i < nev; i++) {
    PetscScalar re, im;
    CHK(EPSGetEigenvalue(eps, i, &re, &im));
    if (i % 8 == 0) printf("\n     ");
    if (im != 0.0) printf("%.5f%+.5fi", re, im);
    else printf("%.5f", re);
    if (i + 1 < nev) printf(", ");
  }
  printf("\n\n");
  if (getenv("KS_EXAMPLE_VERBOSE")) fprintf(stderr, "restarts %d\n", (int)its);
}
static void eps_report(EPS eps) { eps_report_ex(eps, 0); }
#endif
