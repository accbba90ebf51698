/* SVD: the singular value solvers that run on the kernels of the eigensolver -- host
 * control flow only, every vector operation goes through the BV/Vec/Mat op tables.
 *   "cross"     (the reference's default) the eigenproblem of A^T A through EPS:
 *               src/svd/impls/cross/cross.c
 *   "trlanczos" thick-restart Lanczos bidiagonalisation, two-sided or one-sided:
 *               src/svd/impls/trlanczos/trlanczos.c, src/svd/impls/lanczos/gklanczos.c:67-107
 * Interface after src/svd/interface/{svdbasic,svdopts,svdsetup,svdsolve,svddefault}.c.  The
 * transpose is always built explicitly (MatTranspose); the solver works with the taller of
 * A and A^T, as SVDSetUp arranges it (svdsetup.c:139-158). */
#include "ksimpl.h"

struct _p_SVD {
  void *ctx;
  Mat OP, A, AT;            /* operator as given; the tall one and its transpose (one of them is OP) */
  BV U, V;
  DS ds;
  EPS eps;                  /* cross, cyclic */
  Mat cross_mat; Vec cross_w;
  Mat cyclic_mat; Vec cx1, cx2, cy1, cy2;   /* cyclic: [0 A; A^T 0] and views of the two halves of its vectors */
  int type;                 /* 0 cross, 1 trlanczos, 2 lanczos, 3 cyclic */
  PetscBool oneside, leftbasis;
  PetscInt nsv, ncv, mpd, max_it;
  PetscReal tol;
  SVDWhich which;
  SVDConv conv;
  Vec *IS; PetscInt nini;
  PetscReal *sigma, *errest;
  PetscInt *perm;
  PetscInt nconv, its, nalloc;
  SVDConvergedReason reason;
  int state;                /* 0 initial, 1 setup, 2 solved, 3 left vectors available */
};

#define SVDCheckSolved(svd) do { if ((svd)->state < 2) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Must call SVDSolve() first: Parameter #1"); } while (0)

static const char *svd_type_names[] = { "cross", "trlanczos", "lanczos", "cyclic" };

PetscErrorCode SVDCreate(void *ctx, SVD *outsvd) {
  PetscErrorCode ierr;
  if (!outsvd) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  SVD svd = (SVD)calloc(1, sizeof(*svd));
  if (!svd) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  svd->ctx = ctx;
  svd->nsv = 1;
  svd->tol = 0.0;                 /* PETSC_DEFAULT -> 1e-8 at setup */
  svd->which = SVD_LARGEST;
  svd->conv = SVD_CONV_REL;
  ierr = BVCreate(ctx, &svd->V);CHKERRQ(ierr);
  ierr = BVCreate(ctx, &svd->U);CHKERRQ(ierr);
  ierr = DSCreate(&svd->ds);CHKERRQ(ierr);
  *outsvd = svd;
  return 0;
}

PetscErrorCode SVDSetOperator(SVD svd, Mat A) {
  if (!svd || !A) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  svd->OP = A;
  svd->state = 0;
  return 0;
}

PetscErrorCode SVDSetType(SVD svd, const char *type) {
  if (!svd || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  int t;
  if (!strcmp(type, "cross")) t = 0;
  else if (!strcmp(type, "trlanczos")) t = 1;
  else if (!strcmp(type, "lanczos")) t = 2;
  else if (!strcmp(type, "cyclic")) t = 3;
  else SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown SVD type given: %s", type);
  if (t != svd->type) { svd->type = t; svd->state = 0; }
  return 0;
}
PetscErrorCode SVDGetType(SVD svd, const char **type) {
  if (!svd || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  *type = svd_type_names[svd->type];
  return 0;
}

/* svdopts.c (SVDSetDimensions): 0 stands for PETSC_DEFAULT */
PetscErrorCode SVDSetDimensions(SVD svd, PetscInt nsv, PetscInt ncv, PetscInt mpd) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (nsv < 1) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of nsv. Must be > 0");
  if (ncv < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of ncv. Must be > 0");
  if (mpd < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of mpd. Must be > 0");
  svd->nsv = nsv; svd->ncv = ncv; svd->mpd = mpd;
  svd->state = 0;
  return 0;
}
PetscErrorCode SVDGetDimensions(SVD svd, PetscInt *nsv, PetscInt *ncv, PetscInt *mpd) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (nsv) *nsv = svd->nsv;
  if (ncv) *ncv = svd->ncv;
  if (mpd) *mpd = svd->mpd;
  return 0;
}
PetscErrorCode SVDSetTolerances(SVD svd, PetscReal tol, PetscInt maxits) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (tol < 0.0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of tol. Must be > 0");
  if (maxits < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of maxits. Must be > 0");
  svd->tol = tol; svd->max_it = maxits;
  svd->state = 0;
  return 0;
}
PetscErrorCode SVDGetTolerances(SVD svd, PetscReal *tol, PetscInt *maxits) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (tol) *tol = svd->tol;
  if (maxits) *maxits = svd->max_it;
  return 0;
}
PetscErrorCode SVDSetWhichSingularTriplets(SVD svd, SVDWhich which) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (which != SVD_LARGEST && which != SVD_SMALLEST) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid 'which' parameter");
  if (svd->which != which) { svd->state = 0; svd->which = which; }
  return 0;
}
PetscErrorCode SVDSetConvergenceTest(SVD svd, SVDConv conv) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (conv != SVD_CONV_ABS && conv != SVD_CONV_REL) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid 'conv' value");
  svd->conv = conv;
  return 0;
}
PetscErrorCode SVDTRLanczosSetOneSide(SVD svd, PetscBool oneside) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  svd->oneside = oneside ? PETSC_TRUE : PETSC_FALSE;
  return 0;
}
PetscErrorCode SVDLanczosSetOneSide(SVD svd, PetscBool oneside) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (svd->oneside != (oneside ? PETSC_TRUE : PETSC_FALSE)) { svd->oneside = oneside ? PETSC_TRUE : PETSC_FALSE; svd->state = 0; }
  return 0;
}
PetscErrorCode SVDGetBV(SVD svd, BV *V, BV *U) {
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (V) *V = svd->V;
  if (U) *U = svd->U;
  return 0;
}

static PetscErrorCode SVDFreeInitialSpace(SVD svd) {
  PetscErrorCode ierr;
  for (PetscInt i = 0; i < svd->nini && svd->IS; i++) { ierr = VecDestroy(&svd->IS[i]);CHKERRQ(ierr); }
  free(svd->IS); svd->IS = NULL; svd->nini = 0;
  return 0;
}

/* svdsetup.c:222-232; vectors live in the column space of the operator as given */
PetscErrorCode SVDSetInitialSpace(SVD svd, PetscInt n, Vec *is) {
  PetscErrorCode ierr;
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (n < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument n cannot be negative");
  ierr = SVDFreeInitialSpace(svd);CHKERRQ(ierr);
  if (n > 0) {
    svd->IS = (Vec *)calloc((size_t)n, sizeof(Vec));
    if (!svd->IS) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    for (PetscInt i = 0; i < n; i++) {
      ierr = VecDuplicate(is[i], &svd->IS[i]);CHKERRQ(ierr);
      ierr = VecCopy(is[i], svd->IS[i]);CHKERRQ(ierr);
    }
    svd->nini = n;
    svd->state = 0;
  }
  return 0;
}

/* SVDMatMult (include/slepc/private/svdimpl.h): y = A x or y = A^T x of the tall operator */
static PetscErrorCode SVDMatMult(SVD svd, PetscBool trans, Vec x, Vec y) {
  PetscErrorCode ierr;
  ierr = MatMult(trans ? svd->AT : svd->A, x, y);CHKERRQ(ierr);
  return 0;
}

/* ---- cross: the shell matrix A^T A (cross.c:36-64) ------------------------------------------- */
static PetscErrorCode MatMult_Cross(Mat B, Vec x, Vec y) {
  PetscErrorCode ierr;
  SVD svd;
  ierr = MatShellGetContext(B, (void **)&svd);CHKERRQ(ierr);
  ierr = SVDMatMult(svd, PETSC_FALSE, x, svd->cross_w);CHKERRQ(ierr);
  ierr = SVDMatMult(svd, PETSC_TRUE, svd->cross_w, y);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode MatCreateVecs_Cross(Mat B, Vec *right, Vec *left) {
  PetscErrorCode ierr;
  SVD svd;
  ierr = MatShellGetContext(B, (void **)&svd);CHKERRQ(ierr);
  if (right) { ierr = MatCreateVecs(svd->A, right, NULL);CHKERRQ(ierr); }
  if (left) { ierr = MatCreateVecs(svd->A, left, NULL);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode SVDCyclicGetEPS(SVD svd, EPS *eps) { return SVDCrossGetEPS(svd, eps); }

PetscErrorCode SVDCrossGetEPS(SVD svd, EPS *eps) {
  PetscErrorCode ierr;
  if (!svd || !eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!svd->eps) { ierr = EPSCreate(svd->ctx, &svd->eps);CHKERRQ(ierr); }
  *eps = svd->eps;
  return 0;
}

/* svdsetup.c:295-315 */
static PetscErrorCode SVDSetDimensions_Default(SVD svd) {
  PetscInt64 N;
  PetscErrorCode ierr;
  ierr = MatGetSize(svd->A, NULL, &N);CHKERRQ(ierr);
  if (svd->ncv) {
    if (svd->ncv < svd->nsv) SETERRQ(1, "The value of ncv must be at least nsv");
  } else if (svd->mpd) {
    svd->ncv = (PetscInt)PetscMin(N, (PetscInt64)svd->nsv + svd->mpd);
  } else {
    if (svd->nsv < 500) svd->ncv = (PetscInt)PetscMin(N, (PetscInt64)PetscMax(2 * svd->nsv, 10));
    else { svd->mpd = 500; svd->ncv = (PetscInt)PetscMin(N, (PetscInt64)svd->nsv + svd->mpd); }
  }
  if (!svd->mpd) svd->mpd = svd->ncv;
  return 0;
}

/* svdsetup.c:337-385 */
static PetscErrorCode SVDAllocateSolution(SVD svd, PetscInt extra) {
  PetscErrorCode ierr;
  Vec t;
  const PetscInt requested = svd->ncv + extra;
  if (requested != svd->nalloc || !svd->sigma) {
    free(svd->sigma); free(svd->perm); free(svd->errest);
    svd->sigma = (PetscReal *)calloc((size_t)requested, sizeof(PetscReal));
    svd->errest = (PetscReal *)calloc((size_t)requested, sizeof(PetscReal));
    svd->perm = (PetscInt *)calloc((size_t)requested, sizeof(PetscInt));
    if (!svd->sigma || !svd->errest || !svd->perm) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    svd->nalloc = requested;
  }
  if (!svd->V->t) {
    if (!svd->V->type_name[0]) { ierr = BVSetType(svd->V, "hip");CHKERRQ(ierr); }
    ierr = MatCreateVecs(svd->A, &t, NULL);CHKERRQ(ierr);
    ierr = BVSetSizesFromVec(svd->V, t, requested);CHKERRQ(ierr);
    ierr = VecDestroy(&t);CHKERRQ(ierr);
  } else { ierr = BVResize(svd->V, requested, PETSC_FALSE);CHKERRQ(ierr); }
  if (svd->leftbasis) {
    if (!svd->U->t) {
      if (!svd->U->type_name[0]) { ierr = BVSetType(svd->U, svd->V->type_name);CHKERRQ(ierr); }
      ierr = MatCreateVecs(svd->A, NULL, &t);CHKERRQ(ierr);
      ierr = BVSetSizesFromVec(svd->U, t, requested);CHKERRQ(ierr);
      ierr = VecDestroy(&t);CHKERRQ(ierr);
    } else { ierr = BVResize(svd->U, requested, PETSC_FALSE);CHKERRQ(ierr); }
  }
  return 0;
}

/* cross.c:118-172 */
static PetscErrorCode SVDSetUp_Cross(SVD svd) {
  PetscErrorCode ierr;
  PetscInt64 n, N;
  EPS eps;
  if (!svd->cross_mat) {
    ierr = MatGetLocalSize(svd->A, NULL, &n);CHKERRQ(ierr);
    ierr = MatGetSize(svd->A, NULL, &N);CHKERRQ(ierr);
    ierr = MatCreateShell(svd->ctx, n, N, svd->A->col0, svd, MatMult_Cross, MatCreateVecs_Cross, &svd->cross_mat);CHKERRQ(ierr);
    ierr = MatCreateVecs(svd->A, NULL, &svd->cross_w);CHKERRQ(ierr);
  }
  ierr = SVDCrossGetEPS(svd, &eps);CHKERRQ(ierr);
  ierr = EPSSetOperators(eps, svd->cross_mat, NULL);CHKERRQ(ierr);
  ierr = EPSSetProblemType(eps, EPS_HEP);CHKERRQ(ierr);
  ierr = EPSSetWhichEigenpairs(eps, svd->which == SVD_LARGEST ? EPS_LARGEST_REAL : EPS_SMALLEST_REAL);CHKERRQ(ierr);
  ierr = EPSSetDimensions(eps, svd->nsv, svd->ncv, svd->mpd);CHKERRQ(ierr);
  ierr = EPSSetTolerances(eps, svd->tol == 0.0 ? 1e-8 / 10.0 : svd->tol, svd->max_it);CHKERRQ(ierr);
  ierr = EPSSetConvergenceTest(eps, svd->conv == SVD_CONV_ABS ? EPS_CONV_ABS : EPS_CONV_REL);CHKERRQ(ierr);
  if (svd->V->type_name[0]) {                       /* the basis type chosen for the SVD also serves its EPS */
    BV ev;
    ierr = EPSGetBV(eps, &ev);CHKERRQ(ierr);
    if (!ev->type_name[0]) { ierr = BVSetType(ev, svd->V->type_name);CHKERRQ(ierr); }
  }
  /* transfer the initial space (right vectors of the tall operator) */
  if (svd->nini > 0 && svd->IS[0]->n == svd->A->cn) { ierr = EPSSetInitialSpace(eps, svd->nini, svd->IS);CHKERRQ(ierr); }
  ierr = EPSSetUp(eps);CHKERRQ(ierr);
  ierr = EPSGetDimensions(eps, NULL, &svd->ncv, &svd->mpd);CHKERRQ(ierr);
  ierr = EPSGetTolerances(eps, NULL, &svd->max_it);CHKERRQ(ierr);
  if (svd->tol == 0.0) svd->tol = 1e-8;
  svd->leftbasis = PETSC_FALSE;
  ierr = SVDAllocateSolution(svd, 0);CHKERRQ(ierr);
  return 0;
}

/* ---- cyclic: the shell matrix [0 A; A^T 0] (cyclic.c:45-79) ------------------------------- */
/* The two halves of a vector of the shell are addressed through views placed on its storage
 * (device memory for the "hip" vectors: nothing is copied). */
static PetscErrorCode MatMult_Cyclic(Mat B, Vec x, Vec y) {
  PetscErrorCode ierr;
  SVD svd;
  ierr = MatShellGetContext(B, (void **)&svd);CHKERRQ(ierr);
  const PetscInt64 m = svd->A->n;
  ierr = VecPlaceArray(svd->cx1, x->array);CHKERRQ(ierr);
  ierr = VecPlaceArray(svd->cx2, x->array + m);CHKERRQ(ierr);
  ierr = VecPlaceArray(svd->cy1, y->array);CHKERRQ(ierr);
  ierr = VecPlaceArray(svd->cy2, y->array + m);CHKERRQ(ierr);
  ierr = SVDMatMult(svd, PETSC_FALSE, svd->cx2, svd->cy1);
  if (!ierr) ierr = SVDMatMult(svd, PETSC_TRUE, svd->cx1, svd->cy2);
  VecResetArray(svd->cx1); VecResetArray(svd->cx2); VecResetArray(svd->cy1); VecResetArray(svd->cy2);
  CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode MatCreateVecs_Cyclic(Mat B, Vec *right, Vec *left) {
  PetscErrorCode ierr;
  SVD svd;
  ierr = MatShellGetContext(B, (void **)&svd);CHKERRQ(ierr);
  const Vec t = svd->cx1;
  if (right) { ierr = VecCreateWithOps(t->ops, t->ctx, B->n, B->N, B->row0, PETSC_TRUE, right);CHKERRQ(ierr); }
  if (left) { ierr = VecCreateWithOps(t->ops, t->ctx, B->n, B->N, B->row0, PETSC_TRUE, left);CHKERRQ(ierr); }
  return 0;
}

/* cyclic.c:81-226, shell variant */
static PetscErrorCode SVDSetUp_Cyclic(SVD svd) {
  PetscErrorCode ierr;
  PetscInt64 M, N, m, n;
  EPS eps;
  Vec tl, tr;
  ierr = MatGetSize(svd->A, &M, &N);CHKERRQ(ierr);
  ierr = MatGetLocalSize(svd->A, &m, &n);CHKERRQ(ierr);
  if (svd->cyclic_mat) {           /* the shape may have changed */
    ierr = MatDestroy(&svd->cyclic_mat);CHKERRQ(ierr);
    ierr = VecDestroy(&svd->cx1);CHKERRQ(ierr); ierr = VecDestroy(&svd->cx2);CHKERRQ(ierr);
    ierr = VecDestroy(&svd->cy1);CHKERRQ(ierr); ierr = VecDestroy(&svd->cy2);CHKERRQ(ierr);
  }
  ierr = MatCreateVecs(svd->A, &tr, &tl);CHKERRQ(ierr);
  ierr = VecCreateWithOps(tl->ops, tl->ctx, m, M, tl->row0, PETSC_FALSE, &svd->cx1);CHKERRQ(ierr);
  ierr = VecCreateWithOps(tr->ops, tr->ctx, n, N, tr->row0, PETSC_FALSE, &svd->cx2);CHKERRQ(ierr);
  ierr = VecCreateWithOps(tl->ops, tl->ctx, m, M, tl->row0, PETSC_FALSE, &svd->cy1);CHKERRQ(ierr);
  ierr = VecCreateWithOps(tr->ops, tr->ctx, n, N, tr->row0, PETSC_FALSE, &svd->cy2);CHKERRQ(ierr);
  ierr = MatCreateShell(svd->ctx, m + n, M + N, tl->row0 + tr->row0, svd, MatMult_Cyclic, MatCreateVecs_Cyclic, &svd->cyclic_mat);CHKERRQ(ierr);
  ierr = VecDestroy(&tl);CHKERRQ(ierr);
  ierr = VecDestroy(&tr);CHKERRQ(ierr);
  ierr = SVDCyclicGetEPS(svd, &eps);CHKERRQ(ierr);
  ierr = EPSSetOperators(eps, svd->cyclic_mat, NULL);CHKERRQ(ierr);
  ierr = EPSSetProblemType(eps, EPS_HEP);CHKERRQ(ierr);
  if (svd->which == SVD_LARGEST) { ierr = EPSSetWhichEigenpairs(eps, EPS_LARGEST_REAL);CHKERRQ(ierr); }
  else { ierr = EPSSetEigenvalueComparison(eps, SlepcCompareSmallestPosReal, NULL);CHKERRQ(ierr); }
  ierr = EPSSetDimensions(eps, svd->nsv, svd->ncv, svd->mpd);CHKERRQ(ierr);
  ierr = EPSSetTolerances(eps, svd->tol == 0.0 ? 1e-8 / 10.0 : svd->tol, svd->max_it);CHKERRQ(ierr);
  ierr = EPSSetConvergenceTest(eps, svd->conv == SVD_CONV_ABS ? EPS_CONV_ABS : EPS_CONV_REL);CHKERRQ(ierr);
  if (svd->V->type_name[0]) {
    BV ev;
    ierr = EPSGetBV(eps, &ev);CHKERRQ(ierr);
    if (!ev->type_name[0]) { ierr = BVSetType(ev, svd->V->type_name);CHKERRQ(ierr); }
  }
  /* a right initial vector v0 of the tall operator becomes [0; v0] (cyclic.c:183-213) */
  if (svd->nini > 0 && svd->IS[0]->n == n) {
    Vec x;
    ierr = MatCreateVecs(svd->cyclic_mat, &x, NULL);CHKERRQ(ierr);
    ierr = VecSet(x, 0.0);CHKERRQ(ierr);
    ierr = VecPlaceArray(svd->cx2, x->array + m);CHKERRQ(ierr);
    ierr = VecCopy(svd->IS[0], svd->cx2);
    VecResetArray(svd->cx2);
    if (!ierr) ierr = EPSSetInitialSpace(eps, 1, &x);
    VecDestroy(&x);
    CHKERRQ(ierr);
  }
  ierr = EPSSetUp(eps);CHKERRQ(ierr);
  ierr = EPSGetDimensions(eps, NULL, &svd->ncv, &svd->mpd);CHKERRQ(ierr);
  ierr = EPSGetTolerances(eps, NULL, &svd->max_it);CHKERRQ(ierr);
  if (svd->tol == 0.0) svd->tol = 1e-8;
  svd->leftbasis = PETSC_TRUE;
  /* the bases hold every eigenvector the EPS may return; the reported ncv is capped as the reference does */
  ierr = SVDAllocateSolution(svd, 0);CHKERRQ(ierr);
  svd->ncv = (PetscInt)PetscMin((PetscInt64)svd->ncv, PetscMin(M, N));
  return 0;
}

/* cyclic.c:228-267.  A unit eigenvector of the shell holds u/sqrt(2) and v/sqrt(2): both
 * halves are scaled by sqrt(2) (the reference revision at hand divides instead, which leaves
 * singular vectors of norm 1/2). */
static PetscErrorCode SVDSolve_Cyclic(SVD svd) {
  PetscErrorCode ierr;
  PetscInt i, j, nconv;
  PetscScalar sigma;
  EPSConvergedReason reason;
  Vec x;
  const PetscInt64 m = svd->A->n;
  ierr = EPSSolve(svd->eps);CHKERRQ(ierr);
  ierr = EPSGetConverged(svd->eps, &nconv);CHKERRQ(ierr);
  ierr = EPSGetIterationNumber(svd->eps, &svd->its);CHKERRQ(ierr);
  ierr = EPSGetConvergedReason(svd->eps, &reason);CHKERRQ(ierr);
  svd->reason = (SVDConvergedReason)reason;
  ierr = MatCreateVecs(svd->cyclic_mat, &x, NULL);CHKERRQ(ierr);
  for (i = 0, j = 0; i < nconv && j < svd->nalloc; i++) {
    ierr = EPSGetEigenpair(svd->eps, i, &sigma, NULL, x, NULL);
    if (!ierr && sigma > 0.0) {
      svd->sigma[j] = sigma;
      ierr = VecPlaceArray(svd->cx1, x->array);
      if (!ierr) ierr = VecPlaceArray(svd->cx2, x->array + m);
      if (!ierr) ierr = BVInsertVec(svd->U, j, svd->cx1);
      if (!ierr) ierr = BVScaleColumn(svd->U, j, sqrt(2.0));
      if (!ierr) ierr = BVInsertVec(svd->V, j, svd->cx2);
      if (!ierr) ierr = BVScaleColumn(svd->V, j, sqrt(2.0));
      VecResetArray(svd->cx1); VecResetArray(svd->cx2);
      j++;
    }
    if (ierr) { VecDestroy(&x); CHKERRQ(ierr); }
  }
  svd->nconv = j;
  ierr = VecDestroy(&x);CHKERRQ(ierr);
  return 0;
}

/* trlanczos.c:59-75; SVDSetUp_Lanczos (gklanczos.c:46-64) sets the same things up */
static PetscErrorCode SVDSetUp_TRLanczos(SVD svd) {
  PetscErrorCode ierr;
  PetscInt64 N;
  ierr = MatGetSize(svd->A, NULL, &N);CHKERRQ(ierr);
  ierr = SVDSetDimensions_Default(svd);CHKERRQ(ierr);
  if (svd->ncv > svd->nsv + svd->mpd) SETERRQ(1, "The value of ncv must not be larger than nev+mpd");
  if (!svd->max_it) svd->max_it = (PetscInt)PetscMax(N / svd->ncv, (PetscInt64)100);
  svd->leftbasis = (svd->type == 2 && svd->oneside) ? PETSC_FALSE : PETSC_TRUE;
  ierr = SVDAllocateSolution(svd, 1);CHKERRQ(ierr);
  ierr = DSSetType(svd->ds, "svd");CHKERRQ(ierr);
  ierr = DSSetCompact(svd->ds, PETSC_TRUE);CHKERRQ(ierr);
  ierr = DSAllocate(svd->ds, svd->ncv);CHKERRQ(ierr);
  return 0;
}

/* svdsetup.c:98-205 */
PetscErrorCode SVDSetUp(SVD svd) {
  PetscErrorCode ierr;
  PetscInt64 M, N;
  PetscInt k;
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (svd->state) return 0;
  svd->reason = SVD_CONVERGED_ITERATING;
  if (!svd->OP) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "SVDSetOperator must be called first");
  /* build the transpose; the solver works with the taller of the two */
  if (svd->A && svd->A != svd->OP) { ierr = MatDestroy(&svd->A);CHKERRQ(ierr); }
  if (svd->AT && svd->AT != svd->OP) { ierr = MatDestroy(&svd->AT);CHKERRQ(ierr); }
  svd->A = svd->AT = NULL;
  ierr = MatGetSize(svd->OP, &M, &N);CHKERRQ(ierr);
  if (M >= N) { svd->A = svd->OP; ierr = MatTranspose(svd->OP, &svd->AT);CHKERRQ(ierr); }
  else { ierr = MatTranspose(svd->OP, &svd->A);CHKERRQ(ierr); svd->AT = svd->OP; }
  if (svd->ncv > PetscMin(M, N)) svd->ncv = (PetscInt)PetscMin(M, N);
  if (svd->nsv > PetscMin(M, N)) svd->nsv = (PetscInt)PetscMin(M, N);
  if (svd->ncv && svd->nsv > svd->ncv) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "nsv bigger than ncv");
  if (svd->type == 1 || svd->type == 2) { ierr = SVDSetUp_TRLanczos(svd);CHKERRQ(ierr); }
  else if (svd->type == 3) { ierr = SVDSetUp_Cyclic(svd);CHKERRQ(ierr); }
  else { ierr = SVDSetUp_Cross(svd);CHKERRQ(ierr); }
  if (svd->tol == 0.0) svd->tol = 1e-8;
  ierr = DSSetSlepcSC(svd->ds, svd->which == SVD_LARGEST ? SlepcCompareLargestReal : SlepcCompareSmallestReal, NULL);CHKERRQ(ierr);
  /* initial vectors: right vectors of the operator as given; with a wide operator they would
   * be left vectors of the tall one and are ignored (svdsetup.c:160-164,183-189) */
  k = 0;
  if ((svd->type == 1 || svd->type == 2) && svd->nini > 0 && M >= N) {
    k = svd->nini;
    if (k > svd->ncv) SETERRQ(1, "The number of initial vectors is larger than ncv");
    ierr = BVInsertVecs(svd->V, 0, &k, svd->IS, PETSC_TRUE);CHKERRQ(ierr);
  }
  ierr = SVDFreeInitialSpace(svd);CHKERRQ(ierr);
  svd->nini = k;                   /* number of start columns already in V */
  svd->state = 1;
  return 0;
}

static PetscErrorCode SVDConverged(SVD svd, PetscReal sigma, PetscReal res, PetscReal *errest) {
  *errest = svd->conv == SVD_CONV_ABS ? res : res / sigma;      /* svddefault.c:29-45 */
  return 0;
}

/* y_jy = A x_jx (or A^T x_jx) between columns of two bases */
static PetscErrorCode SVDMatMultColumns(SVD svd, PetscBool trans, BV X, PetscInt jx, BV Y, PetscInt jy) {
  PetscErrorCode ierr;
  Vec x, y;
  ierr = BVGetColumn(X, jx, &x);CHKERRQ(ierr);
  ierr = BVGetColumn(Y, jy, &y);CHKERRQ(ierr);
  ierr = SVDMatMult(svd, trans, x, y);
  if (!ierr) ierr = BVRestoreColumn(X, jx, &x);
  if (!ierr) ierr = BVRestoreColumn(Y, jy, &y);
  CHKERRQ(ierr);
  return 0;
}

/* Golub-Kahan-Lanczos bidiagonalisation with both bases orthogonalised, columns k..n
 * (gklanczos.c:67-107): alpha_i = ||u_i||, beta_i = ||v_{i+1}|| before normalisation; v_n is
 * left unnormalised for the caller */
static PetscErrorCode SVDTwoSideLanczos(SVD svd, PetscReal *alpha, PetscReal *beta, BV V, BV U, PetscInt k, PetscInt n) {
  PetscErrorCode ierr;
  for (PetscInt i = k; i < n; i++) {
    ierr = SVDMatMultColumns(svd, PETSC_FALSE, V, i, U, i);CHKERRQ(ierr);              /* u_i = A v_i */
    ierr = BVOrthogonalizeColumn(U, i, NULL, &alpha[i], NULL);CHKERRQ(ierr);
    ierr = BVScaleColumn(U, i, 1.0 / alpha[i]);CHKERRQ(ierr);
    ierr = SVDMatMultColumns(svd, PETSC_TRUE, U, i, V, i + 1);CHKERRQ(ierr);           /* v_{i+1} = A^T u_i */
    ierr = BVOrthogonalizeColumn(V, i + 1, NULL, &beta[i], NULL);CHKERRQ(ierr);
    if (i + 1 < n) { ierr = BVScaleColumn(V, i + 1, 1.0 / beta[i]);CHKERRQ(ierr); }
  }
  return 0;
}

/* trlanczos.c:134-173: h holds the coefficients of the first pass (i+1 of them when the
 * refinement is "if needed": the last one is the squared norm before the pass) */
static PetscErrorCode SVDOrthogonalizeCGS(BV V, PetscInt i, PetscScalar *h, PetscReal a, BVOrthogRefineType refine, PetscReal eta, PetscReal *norm) {
  PetscErrorCode ierr;
  PetscReal sum, onorm;
  PetscScalar dot;
  PetscInt j;
  switch (refine) {
    case BV_ORTHOG_REFINE_NEVER:
      ierr = BVNormColumn(V, i, NORM_2, norm);CHKERRQ(ierr);
      break;
    case BV_ORTHOG_REFINE_ALWAYS:
      ierr = BVSetActiveColumns(V, 0, i);CHKERRQ(ierr);
      ierr = BVDotColumn(V, i, h);CHKERRQ(ierr);
      ierr = BVMultColumn(V, -1.0, 1.0, i, h);CHKERRQ(ierr);
      ierr = BVNormColumn(V, i, NORM_2, norm);CHKERRQ(ierr);
      break;
    case BV_ORTHOG_REFINE_IFNEEDED:
      dot = h[i];
      onorm = sqrt(dot) / a;
      sum = 0.0;
      for (j = 0; j < i; j++) sum += h[j] * h[j];
      *norm = dot / (a * a) - sum;
      if (*norm > 0.0) *norm = sqrt(*norm);
      else { ierr = BVNormColumn(V, i, NORM_2, norm);CHKERRQ(ierr); }
      if (*norm < eta * onorm) {
        ierr = BVSetActiveColumns(V, 0, i);CHKERRQ(ierr);
        ierr = BVDotColumn(V, i, h);CHKERRQ(ierr);
        ierr = BVMultColumn(V, -1.0, 1.0, i, h);CHKERRQ(ierr);
        ierr = BVNormColumn(V, i, NORM_2, norm);CHKERRQ(ierr);
      }
      break;
  }
  return 0;
}

/* trlanczos.c:175-270: only the right basis is orthogonalised; the norm of the previous left
 * vector is obtained together with the Gram-Schmidt coefficients (one reduction round per
 * step), and the left vectors are orthonormalised once at the end */
static PetscErrorCode SVDOneSideTRLanczosCGS(SVD svd, PetscReal *alpha, PetscReal *beta, BV V, BV U, PetscInt nconv, PetscInt l, PetscInt n, PetscScalar *work) {
  PetscErrorCode ierr;
  PetscReal a, b;
  PetscInt i, j, k = nconv + l;
  Vec ui, ui1, vi;
  const BVOrthogRefineType refine = V->orthog_ref;
  const PetscReal eta = V->orthog_eta;
  ierr = SVDMatMultColumns(svd, PETSC_FALSE, V, k, U, k);CHKERRQ(ierr);
  if (l > 0) {
    for (i = 0; i < l; i++) work[i] = beta[i + nconv];
    ierr = BVMultColumn(U, -1.0, 1.0, k, work);CHKERRQ(ierr);
  }
  for (i = k + 1; i <= n; i++) {
    ierr = SVDMatMultColumns(svd, PETSC_TRUE, U, i - 1, V, i);CHKERRQ(ierr);
    /* the norm of the left vector and the coefficients of the right one in one group
     * (trlanczos.c:201-217) */
    ierr = BVNormColumnBegin(U, i - 1, NORM_2, &a);CHKERRQ(ierr);
    if (refine == BV_ORTHOG_REFINE_IFNEEDED) {
      ierr = BVSetActiveColumns(V, 0, i + 1);CHKERRQ(ierr);
      ierr = BVGetColumn(V, i, &vi);CHKERRQ(ierr);
      ierr = BVDotVecBegin(V, vi, work);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V, i, &vi);CHKERRQ(ierr);
    } else {
      ierr = BVSetActiveColumns(V, 0, i);CHKERRQ(ierr);
      ierr = BVDotColumnBegin(V, i, work);CHKERRQ(ierr);
    }
    ierr = BVNormColumnEnd(U, i - 1, NORM_2, &a);CHKERRQ(ierr);
    if (refine == BV_ORTHOG_REFINE_IFNEEDED) {
      ierr = BVDotVecEnd(V, NULL, work);CHKERRQ(ierr);
      ierr = BVSetActiveColumns(V, 0, i);CHKERRQ(ierr);
    } else {
      ierr = BVDotColumnEnd(V, i, work);CHKERRQ(ierr);
    }
    ierr = BVScaleColumn(U, i - 1, 1.0 / a);CHKERRQ(ierr);
    for (j = 0; j < i; j++) work[j] = work[j] / a;
    ierr = BVMultColumn(V, -1.0, 1.0 / a, i, work);CHKERRQ(ierr);
    ierr = SVDOrthogonalizeCGS(V, i, work, a, refine, eta, &b);CHKERRQ(ierr);
    if (i < n) {
      ierr = BVScaleColumn(V, i, 1.0 / b);CHKERRQ(ierr);
      if (fabs(b) < 10 * PETSC_MACHINE_EPSILON) SETERRQ(1, "Recurrence generated a zero vector; use a two-sided variant");
      ierr = SVDMatMultColumns(svd, PETSC_FALSE, V, i, U, i);CHKERRQ(ierr);
      ierr = BVGetColumn(U, i, &ui);CHKERRQ(ierr);
      ierr = BVGetColumn(U, i - 1, &ui1);CHKERRQ(ierr);
      ierr = VecAXPY(ui, -b, ui1);CHKERRQ(ierr);
      ierr = BVRestoreColumn(U, i, &ui);CHKERRQ(ierr);
      ierr = BVRestoreColumn(U, i - 1, &ui1);CHKERRQ(ierr);
    }
    alpha[i - 1] = a;
    beta[i - 1] = b;
  }
  ierr = BVSetActiveColumns(V, nconv, n);CHKERRQ(ierr);
  return 0;
}

/* BVMultInPlace / BVMultInPlaceTranspose with the leading rows x cols part of a DS matrix */
static PetscErrorCode UpdateBasisDS(DS ds, DSMatType mat, BV X, PetscBool trans, PetscInt s, PetscInt e) {
  PetscErrorCode ierr;
  PetscInt rows, cols;
  ierr = DSGetMat(ds, mat, NULL, &rows, &cols);CHKERRQ(ierr);
  PetscScalar *Q = (PetscScalar *)malloc((size_t)PetscMax(rows * cols, 1) * sizeof(PetscScalar));
  if (!Q) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  ierr = DSGetMat(ds, mat, Q, &rows, &cols);
  if (!ierr) ierr = trans ? BVMultInPlaceTranspose(X, Q, rows, s, e) : BVMultInPlace(X, Q, rows, s, e);
  free(Q);
  CHKERRQ(ierr);
  return 0;
}

/* trlanczos.c:272-385 */
static PetscErrorCode SVDSolve_TRLanczos(SVD svd) {
  PetscErrorCode ierr;
  PetscReal *alpha, *beta, lastbeta, resnorm;
  PetscScalar *Q, *swork = NULL, *w;
  PetscInt i, k, l, nv, ld;
  PetscBool conv;
  ierr = DSGetLeadingDimension(svd->ds, &ld);CHKERRQ(ierr);
  w = (PetscScalar *)malloc((size_t)ld * sizeof(*w));
  swork = (PetscScalar *)malloc((size_t)(svd->ncv + 2) * sizeof(*swork));
  if (!w || !swork) { free(w); free(swork); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
#define TCHK(e) do { ierr = (e); if (ierr) { free(w); free(swork); CHKERRQ(ierr); } } while (0)
  /* normalize start vector */
  if (!svd->nini) {
    TCHK(BVSetRandomColumn(svd->V, 0));
    TCHK(BVOrthonormalizeColumn(svd->V, 0, PETSC_TRUE, NULL, NULL));
  }
  l = 0;
  while (svd->reason == SVD_CONVERGED_ITERATING) {
    svd->its++;
    /* inner loop */
    nv = PetscMin(svd->nconv + svd->mpd, svd->ncv);
    TCHK(BVSetActiveColumns(svd->V, svd->nconv, nv));
    TCHK(BVSetActiveColumns(svd->U, svd->nconv, nv));
    TCHK(DSGetArrayReal(svd->ds, DS_MAT_T, &alpha));
    beta = alpha + ld;
    if (svd->oneside) {
      if (svd->V->orthog_type == BV_ORTHOG_MGS) { free(w); free(swork); SETERRQ(PETSC_ERR_SUP, "The one-sided variant is available with classical Gram-Schmidt only"); }
      TCHK(SVDOneSideTRLanczosCGS(svd, alpha, beta, svd->V, svd->U, svd->nconv, l, nv, swork));
    } else {
      TCHK(SVDTwoSideLanczos(svd, alpha, beta, svd->V, svd->U, svd->nconv + l, nv));
    }
    lastbeta = beta[nv - 1];
    TCHK(BVScaleColumn(svd->V, nv, 1.0 / lastbeta));

    /* compute SVD of general matrix */
    TCHK(DSSetDimensions(svd->ds, nv, nv, svd->nconv, svd->nconv + l));
    TCHK(DSSetState(svd->ds, l == 0 ? DS_STATE_INTERMEDIATE : DS_STATE_RAW));
    TCHK(DSSolve(svd->ds, w, NULL));
    TCHK(DSSort(svd->ds, w, NULL, NULL, NULL, NULL));

    /* compute error estimates */
    k = 0;
    conv = PETSC_TRUE;
    TCHK(DSGetArray(svd->ds, DS_MAT_U, &Q));
    for (i = svd->nconv; i < nv; i++) {
      svd->sigma[i] = w[i];
      beta[i] = Q[nv - 1 + i * ld] * lastbeta;
      resnorm = fabs(beta[i]);
      TCHK(SVDConverged(svd, svd->sigma[i], resnorm, &svd->errest[i]));
      if (conv) {
        if (svd->errest[i] < svd->tol) k++;
        else conv = PETSC_FALSE;
      }
    }

    /* check convergence and update l (SVDStoppingBasic, svddefault.c:79-93) */
    if (svd->nconv + k >= svd->nsv) svd->reason = SVD_CONVERGED_TOL;
    else if (svd->its >= svd->max_it) svd->reason = SVD_DIVERGED_ITS;
    if (svd->reason != SVD_CONVERGED_ITERATING) l = 0;
    else l = PetscMax((nv - svd->nconv - k) / 2, 0);

    /* compute converged singular vectors and restart vectors */
    TCHK(UpdateBasisDS(svd->ds, DS_MAT_VT, svd->V, PETSC_TRUE, svd->nconv, svd->nconv + k + l));
    TCHK(UpdateBasisDS(svd->ds, DS_MAT_U, svd->U, PETSC_FALSE, svd->nconv, svd->nconv + k + l));

    /* copy the last vector to be the next initial vector */
    if (svd->reason == SVD_CONVERGED_ITERATING) TCHK(BVCopyColumn(svd->V, nv, svd->nconv + k + l));
    svd->nconv += k;
  }
  /* orthonormalize U columns in one side method */
  if (svd->oneside) for (i = 0; i < svd->nconv; i++) TCHK(BVOrthonormalizeColumn(svd->U, i, PETSC_FALSE, NULL, NULL));
#undef TCHK
  free(w); free(swork);
  return 0;
}

/* gklanczos.c:109-169: the left vectors are not kept, only the current one (u) and the next
 * (u1); the norm of u travels with the first Gram-Schmidt pass of the new right vector */
static PetscErrorCode SVDOneSideLanczos(SVD svd, PetscReal *alpha, PetscReal *beta, BV V, Vec u, Vec u1, PetscInt k, PetscInt n, PetscScalar *work) {
  PetscErrorCode ierr;
  PetscInt i, bvl, bvk;
  PetscReal a, b;
  Vec z, tmp;
  ierr = BVGetActiveColumns(V, &bvl, &bvk);CHKERRQ(ierr);
  ierr = BVGetColumn(V, k, &z);CHKERRQ(ierr);
  ierr = SVDMatMult(svd, PETSC_FALSE, z, u);CHKERRQ(ierr);
  ierr = BVRestoreColumn(V, k, &z);CHKERRQ(ierr);
  for (i = k + 1; i <= n; i++) {
    ierr = BVGetColumn(V, i, &z);CHKERRQ(ierr);
    ierr = SVDMatMult(svd, PETSC_TRUE, u, z);CHKERRQ(ierr);
    ierr = BVRestoreColumn(V, i, &z);CHKERRQ(ierr);
    ierr = VecNormBegin(u, NORM_2, &a);CHKERRQ(ierr);
    ierr = BVSetActiveColumns(V, 0, i);CHKERRQ(ierr);
    ierr = BVDotColumnBegin(V, i, work);CHKERRQ(ierr);
    ierr = VecNormEnd(u, NORM_2, &a);CHKERRQ(ierr);
    ierr = BVDotColumnEnd(V, i, work);CHKERRQ(ierr);
    ierr = VecScale(u, 1.0 / a);CHKERRQ(ierr);
    ierr = BVMultColumn(V, -1.0 / a, 1.0 / a, i, work);CHKERRQ(ierr);
    /* second pass: h = V^T z, z = z - V h */
    ierr = BVDotColumn(V, i, work);CHKERRQ(ierr);
    ierr = BVMultColumn(V, -1.0, 1.0, i, work);CHKERRQ(ierr);
    ierr = BVNormColumn(V, i, NORM_2, &b);CHKERRQ(ierr);
    if (i < n) {
      if (fabs(b) < 10 * PETSC_MACHINE_EPSILON) SETERRQ(1, "Recurrence generated a zero vector; use a two-sided variant");
      ierr = BVScaleColumn(V, i, 1.0 / b);CHKERRQ(ierr);
      ierr = BVGetColumn(V, i, &z);CHKERRQ(ierr);
      ierr = SVDMatMult(svd, PETSC_FALSE, z, u1);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V, i, &z);CHKERRQ(ierr);
      ierr = VecAXPY(u1, -b, u);CHKERRQ(ierr);
      tmp = u; u = u1; u1 = tmp;
    }
    alpha[i - 1] = a;
    beta[i - 1] = b;
  }
  ierr = BVSetActiveColumns(V, bvl, bvk);CHKERRQ(ierr);
  return 0;
}

/* gklanczos.c:171-275: explicit restart with the first unconverged right singular vector */
static PetscErrorCode SVDSolve_Lanczos(SVD svd) {
  PetscErrorCode ierr;
  PetscReal *alpha, *beta, lastbeta, norm, resnorm;
  PetscScalar *swork = NULL, *w = NULL, *Q, *PT;
  PetscInt i, k, j, nv, ld;
  Vec u = NULL, u1 = NULL;
  PetscBool conv;
  ierr = DSGetLeadingDimension(svd->ds, &ld);CHKERRQ(ierr);
  w = (PetscScalar *)malloc((size_t)ld * sizeof(*w));
  swork = (PetscScalar *)malloc((size_t)(svd->ncv + 2) * sizeof(*swork));
  if (!w || !swork) { free(w); free(swork); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
#define LCHK2(e) do { ierr = (e); if (ierr) { free(w); free(swork); VecDestroy(&u); VecDestroy(&u1); CHKERRQ(ierr); } } while (0)
  if (svd->oneside) {
    LCHK2(MatCreateVecs(svd->A, NULL, &u));
    LCHK2(MatCreateVecs(svd->A, NULL, &u1));
  }
  /* normalize start vector */
  if (!svd->nini) {
    LCHK2(BVSetRandomColumn(svd->V, 0));
    LCHK2(BVNormColumn(svd->V, 0, NORM_2, &norm));
    LCHK2(BVScaleColumn(svd->V, 0, 1.0 / norm));
  }
  while (svd->reason == SVD_CONVERGED_ITERATING) {
    svd->its++;
    nv = PetscMin(svd->nconv + svd->mpd, svd->ncv);
    LCHK2(BVSetActiveColumns(svd->V, svd->nconv, nv));
    LCHK2(DSGetArrayReal(svd->ds, DS_MAT_T, &alpha));
    beta = alpha + ld;
    if (svd->oneside) LCHK2(SVDOneSideLanczos(svd, alpha, beta, svd->V, u, u1, svd->nconv, nv, swork));
    else {
      LCHK2(BVSetActiveColumns(svd->U, svd->nconv, nv));
      LCHK2(SVDTwoSideLanczos(svd, alpha, beta, svd->V, svd->U, svd->nconv, nv));
    }
    lastbeta = beta[nv - 1];
    /* SVD of the bidiagonal matrix */
    LCHK2(DSSetDimensions(svd->ds, nv, nv, svd->nconv, 0));
    LCHK2(DSSetState(svd->ds, DS_STATE_INTERMEDIATE));
    LCHK2(DSSolve(svd->ds, w, NULL));
    LCHK2(DSSort(svd->ds, w, NULL, NULL, NULL, NULL));
    /* error estimates */
    k = 0;
    conv = PETSC_TRUE;
    LCHK2(DSGetArray(svd->ds, DS_MAT_U, &Q));
    for (i = svd->nconv; i < nv; i++) {
      svd->sigma[i] = w[i];
      resnorm = fabs(Q[nv - 1 + i * ld]) * lastbeta;
      LCHK2(SVDConverged(svd, svd->sigma[i], resnorm, &svd->errest[i]));
      if (conv) {
        if (svd->errest[i] < svd->tol) k++;
        else conv = PETSC_FALSE;
      }
    }
    if (svd->nconv + k >= svd->nsv) svd->reason = SVD_CONVERGED_TOL;
    else if (svd->its >= svd->max_it) svd->reason = SVD_DIVERGED_ITS;
    /* restart vector: the first unconverged right singular vector, kept in column nv */
    if (svd->reason == SVD_CONVERGED_ITERATING) {
      LCHK2(DSGetArray(svd->ds, DS_MAT_VT, &PT));
      for (j = svd->nconv; j < nv; j++) swork[j - svd->nconv] = PT[k + svd->nconv + j * ld];
      LCHK2(BVMultColumn(svd->V, 1.0, 0.0, nv, swork));
    }
    /* converged singular vectors */
    LCHK2(UpdateBasisDS(svd->ds, DS_MAT_VT, svd->V, PETSC_TRUE, svd->nconv, svd->nconv + k));
    if (!svd->oneside) LCHK2(UpdateBasisDS(svd->ds, DS_MAT_U, svd->U, PETSC_FALSE, svd->nconv, svd->nconv + k));
    if (svd->reason == SVD_CONVERGED_ITERATING) LCHK2(BVCopyColumn(svd->V, nv, svd->nconv + k));
    svd->nconv += k;
  }
#undef LCHK2
  free(w); free(swork);
  ierr = VecDestroy(&u);CHKERRQ(ierr);
  ierr = VecDestroy(&u1);CHKERRQ(ierr);
  return 0;
}

/* cross.c:174-200 */
static PetscErrorCode SVDSolve_Cross(SVD svd) {
  PetscErrorCode ierr;
  PetscInt i;
  PetscScalar sigma;
  EPSConvergedReason reason;
  Vec v;
  ierr = EPSSolve(svd->eps);CHKERRQ(ierr);
  ierr = EPSGetConverged(svd->eps, &svd->nconv);CHKERRQ(ierr);
  ierr = EPSGetIterationNumber(svd->eps, &svd->its);CHKERRQ(ierr);
  ierr = EPSGetConvergedReason(svd->eps, &reason);CHKERRQ(ierr);
  svd->reason = (SVDConvergedReason)reason;
  if (svd->nconv > svd->ncv) svd->nconv = svd->ncv;
  for (i = 0; i < svd->nconv; i++) {
    ierr = BVGetColumn(svd->V, i, &v);CHKERRQ(ierr);
    ierr = EPSGetEigenpair(svd->eps, i, &sigma, NULL, v, NULL);CHKERRQ(ierr);
    ierr = BVRestoreColumn(svd->V, i, &v);CHKERRQ(ierr);
    if (sigma < 0.0) SETERRQ(1, "Negative eigenvalue computed by EPS");
    svd->sigma[i] = sqrt(sigma);
  }
  return 0;
}

static int svd_sigma_cmp_perm(const PetscReal *sigma, PetscInt a, PetscInt b) { return (sigma[a] > sigma[b]) - (sigma[a] < sigma[b]); }

/* svdsolve.c:80-128 */
PetscErrorCode SVDSolve(SVD svd) {
  PetscErrorCode ierr;
  PetscInt i, j;
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  if (svd->state >= 2) return 0;
  ierr = SVDSetUp(svd);CHKERRQ(ierr);
  svd->its = 0;
  svd->nconv = 0;
  for (i = 0; i < svd->ncv; i++) { svd->sigma[i] = 0.0; svd->errest[i] = 0.0; svd->perm[i] = i; }
  if (svd->type == 1) { ierr = SVDSolve_TRLanczos(svd);CHKERRQ(ierr); }
  else if (svd->type == 2) { ierr = SVDSolve_Lanczos(svd);CHKERRQ(ierr); }
  else if (svd->type == 3) { ierr = SVDSolve_Cyclic(svd);CHKERRQ(ierr); }
  else { ierr = SVDSolve_Cross(svd);CHKERRQ(ierr); }
  svd->state = svd->leftbasis ? 3 : 2;
  /* sort singular triplets: ascending permutation (insertion sort, stable), reversed for "largest" */
  for (i = 0; i < svd->nconv; i++) svd->perm[i] = i;
  for (i = 1; i < svd->nconv; i++) {
    const PetscInt p = svd->perm[i];
    for (j = i - 1; j >= 0 && svd_sigma_cmp_perm(svd->sigma, svd->perm[j], p) > 0; j--) svd->perm[j + 1] = svd->perm[j];
    svd->perm[j + 1] = p;
  }
  if (svd->which == SVD_LARGEST)
    for (i = 0; i < svd->nconv / 2; i++) { const PetscInt t = svd->perm[i]; svd->perm[i] = svd->perm[svd->nconv - 1 - i]; svd->perm[svd->nconv - 1 - i] = t; }
  svd->nini = 0;
  return 0;
}

PetscErrorCode SVDGetIterationNumber(SVD svd, PetscInt *its) { if (!svd || !its) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *its = svd->its; return 0; }
PetscErrorCode SVDGetConverged(SVD svd, PetscInt *nconv) { if (!svd || !nconv) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); SVDCheckSolved(svd); *nconv = svd->nconv; return 0; }
PetscErrorCode SVDGetConvergedReason(SVD svd, SVDConvergedReason *reason) { if (!svd || !reason) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); SVDCheckSolved(svd); *reason = svd->reason; return 0; }

/* svdsolve.c:26-57: u_j = A v_j, orthonormalised */
static PetscErrorCode SVDComputeVectors(SVD svd) {
  PetscErrorCode ierr;
  Vec tl, uj, vj;
  PetscInt j;
  SVDCheckSolved(svd);
  if (svd->state == 2) {
    if (!svd->U->t) {
      if (!svd->U->type_name[0]) { ierr = BVSetType(svd->U, svd->V->type_name);CHKERRQ(ierr); }
      ierr = MatCreateVecs(svd->A, NULL, &tl);CHKERRQ(ierr);
      ierr = BVSetSizesFromVec(svd->U, tl, svd->ncv);CHKERRQ(ierr);
      ierr = VecDestroy(&tl);CHKERRQ(ierr);
    } else if (svd->U->m < svd->ncv) { ierr = BVResize(svd->U, svd->ncv, PETSC_FALSE);CHKERRQ(ierr); }
    for (j = 0; j < svd->nconv; j++) {
      ierr = BVGetColumn(svd->V, j, &vj);CHKERRQ(ierr);
      ierr = BVGetColumn(svd->U, j, &uj);CHKERRQ(ierr);
      ierr = SVDMatMult(svd, PETSC_FALSE, vj, uj);CHKERRQ(ierr);
      ierr = BVRestoreColumn(svd->V, j, &vj);CHKERRQ(ierr);
      ierr = BVRestoreColumn(svd->U, j, &uj);CHKERRQ(ierr);
      ierr = BVOrthonormalizeColumn(svd->U, j, PETSC_FALSE, NULL, NULL);CHKERRQ(ierr);
    }
  }
  svd->state = 3;
  return 0;
}

/* svdsolve.c:256-280 */
PetscErrorCode SVDGetSingularTriplet(SVD svd, PetscInt i, PetscReal *sigma, Vec u, Vec v) {
  PetscErrorCode ierr;
  PetscInt64 M, N;
  Vec w;
  if (!svd) SETERRQ(PETSC_ERR_ARG_NULL, "Null SVD");
  SVDCheckSolved(svd);
  if (i < 0 || i >= svd->nconv) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument 2 out of range");
  if (sigma) *sigma = svd->sigma[svd->perm[i]];
  ierr = MatGetSize(svd->OP, &M, &N);CHKERRQ(ierr);
  if (M < N) { w = u; u = v; v = w; }
  if (u) {
    ierr = SVDComputeVectors(svd);CHKERRQ(ierr);
    ierr = BVCopyVec(svd->U, svd->perm[i], u);CHKERRQ(ierr);
  }
  if (v) { ierr = BVCopyVec(svd->V, svd->perm[i], v);CHKERRQ(ierr); }
  return 0;
}

/* svdsolve.c:286-379: sqrt(||A v - sigma u||^2 + ||A^T u - sigma v||^2), relative to sigma on request */
PetscErrorCode SVDComputeError(SVD svd, PetscInt i, SVDErrorType type, PetscReal *error) {
  PetscErrorCode ierr;
  Vec u = NULL, v = NULL, x = NULL, y = NULL;
  PetscReal sigma, norm1, norm2;
  PetscInt64 M, N;
  if (!svd || !error) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  SVDCheckSolved(svd);
#define ECHK(e) do { ierr = (e); if (ierr) { VecDestroy(&u); VecDestroy(&v); VecDestroy(&x); VecDestroy(&y); CHKERRQ(ierr); } } while (0)
  ECHK(MatCreateVecs(svd->OP, &v, &u));
  ECHK(SVDGetSingularTriplet(svd, i, &sigma, u, v));
  ECHK(VecDuplicate(u, &x));
  ECHK(MatMult(svd->OP, v, x));
  ECHK(VecAXPY(x, -sigma, u));
  ECHK(VecNorm(x, NORM_2, &norm1));
  ECHK(VecDuplicate(v, &y));
  ECHK(MatGetSize(svd->OP, &M, &N));
  ECHK(MatMult(M < N ? svd->A : svd->AT, u, y));
  ECHK(VecAXPY(y, -sigma, v));
  ECHK(VecNorm(y, NORM_2, &norm2));
#undef ECHK
  ierr = VecDestroy(&u);CHKERRQ(ierr);
  ierr = VecDestroy(&v);CHKERRQ(ierr);
  ierr = VecDestroy(&x);CHKERRQ(ierr);
  ierr = VecDestroy(&y);CHKERRQ(ierr);
  *error = sqrt(norm1 * norm1 + norm2 * norm2);
  switch (type) {
    case SVD_ERROR_ABSOLUTE: break;
    case SVD_ERROR_RELATIVE: *error /= sigma; break;
    default: SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid error type");
  }
  return 0;
}

PetscErrorCode SVDDestroy(SVD *psvd) {
  PetscErrorCode ierr;
  if (!psvd || !*psvd) return 0;
  SVD svd = *psvd;
  ierr = SVDFreeInitialSpace(svd);CHKERRQ(ierr);
  ierr = EPSDestroy(&svd->eps);CHKERRQ(ierr);
  ierr = MatDestroy(&svd->cross_mat);CHKERRQ(ierr);
  ierr = VecDestroy(&svd->cross_w);CHKERRQ(ierr);
  ierr = MatDestroy(&svd->cyclic_mat);CHKERRQ(ierr);
  ierr = VecDestroy(&svd->cx1);CHKERRQ(ierr);
  ierr = VecDestroy(&svd->cx2);CHKERRQ(ierr);
  ierr = VecDestroy(&svd->cy1);CHKERRQ(ierr);
  ierr = VecDestroy(&svd->cy2);CHKERRQ(ierr);
  if (svd->A && svd->A != svd->OP) { ierr = MatDestroy(&svd->A);CHKERRQ(ierr); }
  if (svd->AT && svd->AT != svd->OP) { ierr = MatDestroy(&svd->AT);CHKERRQ(ierr); }
  ierr = BVDestroy(&svd->V);CHKERRQ(ierr);
  ierr = BVDestroy(&svd->U);CHKERRQ(ierr);
  ierr = DSDestroy(&svd->ds);CHKERRQ(ierr);
  free(svd->sigma); free(svd->errest); free(svd->perm);
  free(svd);
  *psvd = NULL;
  return 0;
}
