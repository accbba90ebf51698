/* EPS: the eigensolver object and its Krylov-Schur drivers -- host control flow
 * only; every vector operation goes through the BV/Vec/Mat op tables and so runs
 * on the device.  Follows src/eps/interface/{epsbasic,epsopts,epssetup,epssolve,
 * epsdefault}.c, src/eps/impls/krylov/epskrylov.c and
 * src/eps/impls/krylov/krylovschur/{krylovschur,ks-symm}.c of the reference. */
#include "dense.h"
#include "ksimpl.h"

#define EPSCheckSolved(eps) do { if ((eps)->state < 2) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first: Parameter #1"); } while (0)

/* ---- creation and options (epsbasic.c:60-80, epsopts.c) ----------------------------------- */
PetscErrorCode EPSCreate(void *ctx, EPS *outeps) {
  PetscErrorCode ierr;
  if (!outeps) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  EPS eps = (EPS)calloc(1, sizeof(*eps));
  if (!eps) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  eps->ctx = ctx;
  eps->nev = 1;
  eps->tol = 0.0;                 /* PETSC_DEFAULT -> 1e-8 at setup (epssetup.c:184) */
  eps->lock = PETSC_TRUE;
  eps->conv = EPS_CONV_REL;
  eps->purify = PETSC_TRUE;       /* epsbasic.c:80 */
  eps->problem_type = (EPSProblemType)0;
  ierr = BVCreate(ctx, &eps->V);CHKERRQ(ierr);
  ierr = DSCreate(&eps->ds);CHKERRQ(ierr);
  *outeps = eps;
  return 0;
}

PetscErrorCode EPSSetOperators(EPS eps, Mat A, Mat B) {
  if (!eps || !A) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (B && (A->n != B->n || A->N != B->N)) SETERRQ(PETSC_ERR_ARG_INCOMP, "Incompatible matrix sizes");
  eps->A = A;
  eps->B = B;
  eps->state = 0;
  return 0;
}

/* the eigensolver's ST, created on first use (epsbasic.c, EPSGetST) */
PetscErrorCode EPSGetST(EPS eps, ST *st) {
  PetscErrorCode ierr;
  if (!eps || !st) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!eps->st) { ierr = STCreate(eps->ctx, &eps->st);CHKERRQ(ierr); }
  *st = eps->st;
  return 0;
}

PetscErrorCode EPSSetTarget(EPS eps, PetscScalar target) {
  PetscErrorCode ierr;
  ST st;
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  eps->target = target;
  ierr = EPSGetST(eps, &st);CHKERRQ(ierr);
  ierr = STSetDefaultShift(st, target);CHKERRQ(ierr);
  eps->state = 0;
  return 0;
}
PetscErrorCode EPSGetTarget(EPS eps, PetscScalar *target) { if (!eps || !target) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *target = eps->target; return 0; }
PetscErrorCode EPSSetPurify(EPS eps, PetscBool purify) { if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS"); eps->purify = purify ? PETSC_TRUE : PETSC_FALSE; eps->state = 0; return 0; }

/* epssetup.c (EPSSetDeflationSpace): copies are kept until the setup turns them into constraints */
PetscErrorCode EPSSetDeflationSpace(EPS eps, PetscInt n, Vec *v) {
  PetscErrorCode ierr;
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (n < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument n out of range");
  if (eps->defl) for (PetscInt i = 0; i < -eps->nds; i++) { ierr = VecDestroy(&eps->defl[i]);CHKERRQ(ierr); }
  free(eps->defl); eps->defl = NULL;
  if (eps->nds > 0) { ierr = BVSetNumConstraints(eps->V, 0);CHKERRQ(ierr); }
  eps->nds = 0;
  if (n > 0) {
    if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
    eps->defl = (Vec *)calloc((size_t)n, sizeof(Vec));
    if (!eps->defl) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    for (PetscInt i = 0; i < n; i++) {
      ierr = VecDuplicate(v[i], &eps->defl[i]);CHKERRQ(ierr);
      ierr = VecCopy(v[i], eps->defl[i]);CHKERRQ(ierr);
    }
    eps->nds = -n;           /* pending, as the reference marks them */
  }
  eps->state = 0;
  return 0;
}

PetscErrorCode EPSSetProblemType(EPS eps, EPSProblemType type) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  switch (type) {
    case EPS_HEP: eps->ishermitian = PETSC_TRUE; eps->isgeneralized = PETSC_FALSE; eps->ispositive = PETSC_FALSE; break;
    case EPS_NHEP: eps->ishermitian = PETSC_FALSE; eps->isgeneralized = PETSC_FALSE; eps->ispositive = PETSC_FALSE; break;
    case EPS_GHEP: eps->ishermitian = PETSC_TRUE; eps->isgeneralized = PETSC_TRUE; eps->ispositive = PETSC_TRUE; break;
    case EPS_GNHEP: eps->ishermitian = PETSC_FALSE; eps->isgeneralized = PETSC_TRUE; eps->ispositive = PETSC_FALSE; break;
    case EPS_PGNHEP: eps->ishermitian = PETSC_FALSE; eps->isgeneralized = PETSC_TRUE; eps->ispositive = PETSC_TRUE; break;
    default: SETERRQ(PETSC_ERR_ARG_WRONG, "Unknown eigenvalue problem type");
  }
  eps->problem_type = type;
  eps->state = 0;
  return 0;
}

/* epsbasic.c (EPSSetType): "krylovschur" (default), "arnoldi" or "lanczos" */
PetscErrorCode EPSSetType(EPS eps, const char *type) {
  if (!eps || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  int t;
  if (!strcmp(type, "krylovschur")) t = 0;
  else if (!strcmp(type, "arnoldi")) t = 1;
  else if (!strcmp(type, "lanczos")) t = 2;
  else SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown EPS type given: %s", type);
  if (t != eps->type) { eps->type = t; eps->state = 0; }
  return 0;
}

PetscErrorCode EPSGetType(EPS eps, const char **type) {
  if (!eps || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  *type = eps->type == 1 ? "arnoldi" : eps->type == 2 ? "lanczos" : "krylovschur";
  return 0;
}

PetscErrorCode EPSSetDimensions(EPS eps, PetscInt nev, PetscInt ncv, PetscInt mpd) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (nev < 1) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of nev. Must be > 0");
  eps->nev = nev;
  if (ncv < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of ncv. Must be > 0");
  eps->ncv = ncv;      /* 0 = PETSC_DEFAULT */
  if (mpd < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of mpd. Must be > 0");
  eps->mpd = mpd;
  eps->state = 0;
  return 0;
}

PetscErrorCode EPSSetTolerances(EPS eps, PetscReal tol, PetscInt maxits) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (tol < 0.0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of tol. Must be > 0");
  if (maxits < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of maxits. Must be > 0");
  eps->tol = tol;            /* 0 = PETSC_DEFAULT */
  eps->max_it = maxits;      /* 0 = PETSC_DEFAULT */
  return 0;
}

PetscErrorCode EPSSetWhichEigenpairs(EPS eps, EPSWhich which) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (which < EPS_LARGEST_MAGNITUDE || which > EPS_TARGET_REAL) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid 'which' value");
  eps->which = which;
  eps->state = 0;
  return 0;
}

PetscErrorCode EPSSetEigenvalueComparison(EPS eps, SlepcCompareFn cmp, void *ctx) {
  if (!eps || !cmp) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (ctx) SETERRQ(PETSC_ERR_SUP, "A context for the comparison routine is not available");
  eps->usercmp = cmp;
  eps->which = EPS_WHICH_USER;
  eps->state = 0;
  return 0;
}

PetscErrorCode EPSSetInitialSpace(EPS eps, PetscInt n, Vec *is) {
  PetscErrorCode ierr;
  PetscInt i;
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (n < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument n cannot be negative");
  if (eps->IS) for (i = 0; i < eps->nini; i++) { ierr = VecDestroy(&eps->IS[i]);CHKERRQ(ierr); }
  free(eps->IS); eps->IS = NULL; eps->nini = 0;
  if (n > 0) {
    eps->IS = (Vec *)calloc((size_t)n, sizeof(Vec));
    if (!eps->IS) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    for (i = 0; i < n; i++) {       /* SlepcBasisReference_Private keeps references; we keep copies */
      ierr = VecDuplicate(is[i], &eps->IS[i]);CHKERRQ(ierr);
      ierr = VecCopy(is[i], eps->IS[i]);CHKERRQ(ierr);
    }
    eps->nini = n;
    eps->state = 0;
  }
  return 0;
}

PetscErrorCode EPSKrylovSchurSetRestart(EPS eps, PetscReal keep) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (keep == 0.0) eps->keep = 0.5;
  else {
    if (keep < 0.1 || keep > 0.9) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "The keep argument must be in the range [0.1,0.9]");
    eps->keep = keep;
  }
  return 0;
}
PetscErrorCode EPSSetConvergenceTest(EPS eps, EPSConv conv) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (conv != EPS_CONV_ABS && conv != EPS_CONV_REL) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid 'conv' value");
  eps->conv = conv;
  return 0;
}
/* lanczos.c:740-790 */
PetscErrorCode EPSLanczosSetReorthog(EPS eps, EPSLanczosReorthogType reorthog) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (reorthog < EPS_LANCZOS_REORTHOG_LOCAL || reorthog > EPS_LANCZOS_REORTHOG_DELAYED) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid reorthogonalization type");
  if (eps->reorthog != reorthog) { eps->reorthog = reorthog; eps->state = 0; }
  return 0;
}
/* arnoldi.c:168-175 */
PetscErrorCode EPSArnoldiSetDelayed(EPS eps, PetscBool delayed) { if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS"); eps->delayed = delayed ? PETSC_TRUE : PETSC_FALSE; return 0; }
PetscErrorCode EPSKrylovSchurSetLocking(EPS eps, PetscBool lock) { if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS"); eps->lock = lock; return 0; }
PetscErrorCode EPSMonitorSet(EPS eps, EPSMonitorFn monitor, void *ctx) { if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS"); eps->monitor = monitor; eps->monitorctx = ctx; return 0; }
PetscErrorCode EPSGetBV(EPS eps, BV *bv) { if (!eps || !bv) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *bv = eps->V; return 0; }
PetscErrorCode EPSGetDS(EPS eps, DS *ds) { if (!eps || !ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *ds = eps->ds; return 0; }

/* The solver orders the eigenvalues of the transformed operator by what they are in the original
 * problem: both values go through STBackTransform first (SlepcSCCompare with sc->map =
 * SlepcMap_ST, slepcsc.c:51-81,162-171; epssetup.c:239-248). */
static PetscErrorCode EPSCompareMapped(PetscScalar ar, PetscScalar ai, PetscScalar br, PetscScalar bi, PetscInt *res, void *ctx) {
  PetscErrorCode ierr;
  EPS eps = (EPS)ctx;
  ierr = STBackTransform(eps->st, 1, &ar, &ai);CHKERRQ(ierr);
  ierr = STBackTransform(eps->st, 1, &br, &bi);CHKERRQ(ierr);
  return eps->cmp0(ar, ai, br, bi, res, eps->cmp0ctx);
}

/* ---- setup (epssetup.c:100-288,466-555; krylovschur.c:73-167) ----------------------------------- */
static PetscErrorCode EPSSetDimensions_Default(EPS eps, PetscInt nev, PetscInt *ncv, PetscInt *mpd) {
  const PetscInt n = (PetscInt)PetscMin(eps->n, (PetscInt64)2147483647);
  if (*ncv) {
    if (*ncv < nev + 1 && !(*ncv == nev && *ncv == n)) SETERRQ(1, "The value of ncv must be at least nev+1");
  } else if (*mpd) *ncv = PetscMin(n, nev + (*mpd));
  else {
    if (nev < 500) *ncv = PetscMin(n, PetscMax(2 * nev, nev + 15));
    else { *mpd = 500; *ncv = PetscMin(n, nev + (*mpd)); }
  }
  if (!*mpd) *mpd = *ncv;
  return 0;
}

PetscErrorCode EPSSetUp(EPS eps) {
  PetscErrorCode ierr;
  PetscInt oldm = 0, k;
  Vec t;
  static const SlepcCompareFn cmps[] = { NULL, SlepcCompareLargestMagnitude, SlepcCompareSmallestMagnitude, SlepcCompareLargestReal,
                                         SlepcCompareSmallestReal, SlepcCompareLargestImaginary, SlepcCompareSmallestImaginary,
                                         SlepcCompareTargetMagnitude, SlepcCompareTargetReal };
  ST st;
  const char *sttype;
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (eps->state) return 0;
  if (!eps->A) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "EPSSetOperators must be called first");
  ierr = EPSGetST(eps, &st);CHKERRQ(ierr);
  ierr = STSetOperators(st, eps->A, eps->B);CHKERRQ(ierr);
  /* default problem type, consistency (epssetup.c:137-149) */
  if (!eps->problem_type) { ierr = EPSSetProblemType(eps, eps->B ? EPS_GNHEP : EPS_NHEP);CHKERRQ(ierr); }
  else if (!eps->B && eps->isgeneralized) { ierr = EPSSetProblemType(eps, eps->ishermitian ? EPS_HEP : EPS_NHEP);CHKERRQ(ierr); }
  else if (eps->B && !eps->isgeneralized) SETERRQ(PETSC_ERR_ARG_INCOMP, "Inconsistent EPS state");
  ierr = STGetType(st, &sttype);CHKERRQ(ierr);
  {
    PetscInt64 ncols;
    ierr = MatGetSize(eps->A, &eps->n, &ncols);CHKERRQ(ierr);
    if (ncols != eps->n) SETERRQ(PETSC_ERR_ARG_WRONG, "A is a non-square matrix");        /* epssetup.c:324-326 */
  }
  ierr = MatGetLocalSize(eps->A, &eps->nloc, NULL);CHKERRQ(ierr);
  if (eps->nev > eps->n) eps->nev = (PetscInt)eps->n;
  if (eps->ncv > eps->n) eps->ncv = (PetscInt)eps->n;

  /* EPSSetUp_KrylovSchur */
  ierr = EPSSetDimensions_Default(eps, eps->nev, &eps->ncv, &eps->mpd);CHKERRQ(ierr);
  if (eps->ncv > eps->nev + eps->mpd) SETERRQ(1, "The value of ncv must not be larger than nev+mpd");
  if (!eps->max_it) eps->max_it = (PetscInt)PetscMax((PetscInt64)100, 2 * eps->n / eps->ncv);
  if (!eps->which) eps->which = strcmp(sttype, "sinvert") ? EPS_LARGEST_MAGNITUDE : EPS_TARGET_MAGNITUDE;   /* epsdefault.c:243-253 */
  if (!eps->lock && eps->mpd < eps->ncv) SETERRQ(PETSC_ERR_SUP, "Should not use mpd parameter in non-locking variant");
  if (eps->ishermitian && (eps->which == EPS_LARGEST_IMAGINARY || eps->which == EPS_SMALLEST_IMAGINARY)) SETERRQ(1, "Wrong value of eps->which");
  if (!eps->keep) eps->keep = 0.5;
  if (eps->tol == 0.0) eps->tol = 1e-8;                                                   /* SLEPC_DEFAULT_TOL */
  eps->cmp0 = eps->which == EPS_WHICH_USER ? eps->usercmp : cmps[eps->which];
  eps->cmp0ctx = (eps->which == EPS_TARGET_MAGNITUDE || eps->which == EPS_TARGET_REAL) ? (void *)&eps->target : NULL;
  /* purification only where it makes sense (epssetup.c:169-177) */
  if (eps->purify && (!eps->isgeneralized || (!eps->ishermitian && !eps->ispositive))) eps->purify = PETSC_FALSE;
  ierr = STSetUp(st);CHKERRQ(ierr);
  if (STIsTrivial_Private(st)) { eps->cmp = eps->cmp0; eps->cmpctx = eps->cmp0ctx; }
  else { eps->cmp = EPSCompareMapped; eps->cmpctx = eps; }

  /* EPSAllocateSolution(eps,1) */
  ierr = BVGetSizes(eps->V, NULL, NULL, &oldm);CHKERRQ(ierr);
  const PetscInt requested = eps->ncv + 1;
  if (requested != oldm || !eps->eigr) {
    free(eps->eigr); free(eps->eigi); free(eps->errest); free(eps->perm);
    eps->eigr = (PetscScalar *)calloc((size_t)requested, sizeof(PetscScalar));
    eps->eigi = (PetscScalar *)calloc((size_t)requested, sizeof(PetscScalar));
    eps->errest = (PetscReal *)calloc((size_t)requested, sizeof(PetscReal));
    eps->perm = (PetscInt *)calloc((size_t)requested, sizeof(PetscInt));
    if (!eps->eigr || !eps->eigi || !eps->errest || !eps->perm) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  }
  if (!eps->V->t) {
    if (!eps->V->type_name[0]) { ierr = BVSetType(eps->V, "hip");CHKERRQ(ierr); }        /* default type (BVSVEC there) */
    ierr = MatCreateVecs(eps->A, &t, NULL);CHKERRQ(ierr);
    ierr = BVSetSizesFromVec(eps->V, t, requested);CHKERRQ(ierr);
    ierr = VecDestroy(&t);CHKERRQ(ierr);
  } else {
    if (eps->nds > 0) { ierr = BVSetNumConstraints(eps->V, 0);CHKERRQ(ierr); eps->nds = 0; }
    ierr = BVResize(eps->V, requested, PETSC_FALSE);CHKERRQ(ierr);
  }
  /* EPS_SetInnerProduct (epsimpl.h:146-161): B-inner product where it makes the operator self-adjoint */
  if (eps->ispositive || (eps->isgeneralized && eps->ishermitian)) {
    Mat Bf;
    ierr = STGetBilinearForm(st, &Bf);CHKERRQ(ierr);
    ierr = BVSetMatrix(eps->V, Bf, eps->ispositive ? PETSC_FALSE : PETSC_TRUE);CHKERRQ(ierr);
  } else { ierr = BVSetMatrix(eps->V, NULL, PETSC_FALSE);CHKERRQ(ierr); }

  /* dispatch solve method: DS type (EPSSetUp_KrylovSchur krylovschur.c:141-148;
     EPSSetUp_Arnoldi arnoldi.c:58-64: always the non-symmetric projected problem, extra row) */
  if (eps->type == 1) {
    ierr = DSSetType(eps->ds, "nhep");CHKERRQ(ierr);
    ierr = DSSetCompact(eps->ds, PETSC_FALSE);CHKERRQ(ierr);
    ierr = DSSetExtraRow(eps->ds, PETSC_TRUE);CHKERRQ(ierr);
  } else if (eps->type == 2) {                     /* EPSSetUp_Lanczos lanczos.c:47-108 */
    if (!eps->ishermitian) SETERRQ(PETSC_ERR_SUP, "Requested method is only available for Hermitian problems");
    if (eps->reorthog != EPS_LANCZOS_REORTHOG_FULL) eps->V->orthog_type = BV_ORTHOG_MGS;     /* lanczos.c:77-81 */
    if (eps->reorthog == EPS_LANCZOS_REORTHOG_SELECTIVE) {                                   /* lanczos.c:82-91 */
      if (eps->AV && eps->AV->m != eps->V->m) { ierr = BVDestroy(&eps->AV);CHKERRQ(ierr); }
      if (!eps->AV) {
        ierr = BVCreate(eps->ctx, &eps->AV);CHKERRQ(ierr);
        ierr = BVSetType(eps->AV, eps->V->type_name);CHKERRQ(ierr);
        ierr = BVSetSizesFromVec(eps->AV, eps->V->t, eps->V->m);CHKERRQ(ierr);
      }
      ierr = BVSetOrthogonalization(eps->AV, BV_ORTHOG_MGS, eps->V->orthog_ref, eps->V->orthog_eta, eps->V->orthog_block);CHKERRQ(ierr);
    }
    if (eps->reorthog == EPS_LANCZOS_REORTHOG_LOCAL && eps->nwork < 1) {                     /* EPSSetWorkVecs(eps,1) */
      ierr = VecDuplicate(eps->V->t, &eps->work[0]);CHKERRQ(ierr);
      eps->nwork = 1;
    }
    ierr = DSSetType(eps->ds, "hep");CHKERRQ(ierr);
    ierr = DSSetCompact(eps->ds, PETSC_TRUE);CHKERRQ(ierr);
    ierr = DSSetExtraRow(eps->ds, PETSC_FALSE);CHKERRQ(ierr);
  } else if (eps->ishermitian) {
    ierr = DSSetType(eps->ds, "hep");CHKERRQ(ierr);
    ierr = DSSetCompact(eps->ds, PETSC_TRUE);CHKERRQ(ierr);
    ierr = DSSetExtraRow(eps->ds, PETSC_TRUE);CHKERRQ(ierr);
  } else {
    ierr = DSSetType(eps->ds, "nhep");CHKERRQ(ierr);
    ierr = DSSetCompact(eps->ds, PETSC_FALSE);CHKERRQ(ierr);
    ierr = DSSetExtraRow(eps->ds, PETSC_FALSE);CHKERRQ(ierr);
  }
  ierr = DSAllocate(eps->ds, eps->ncv + 1);CHKERRQ(ierr);
  ierr = DSSetSlepcSC(eps->ds, eps->cmp, eps->cmpctx);CHKERRQ(ierr);

  /* process deflation and initial vectors (epssetup.c:263-283) */
  if (eps->nds < 0) {
    k = -eps->nds;
    ierr = BVInsertConstraints(eps->V, &k, eps->defl);CHKERRQ(ierr);
    for (PetscInt i = 0; i < -eps->nds; i++) { ierr = VecDestroy(&eps->defl[i]);CHKERRQ(ierr); }
    free(eps->defl); eps->defl = NULL;
    eps->nds = k;
  }
  if (eps->nini > 0) {
    k = eps->nini;
    ierr = BVInsertVecs(eps->V, 0, &k, eps->IS, PETSC_TRUE);CHKERRQ(ierr);
    for (PetscInt i = 0; i < eps->nini; i++) { ierr = VecDestroy(&eps->IS[i]);CHKERRQ(ierr); }
    free(eps->IS); eps->IS = NULL;
    eps->nini = k;      /* number of usable initial vectors now stored in V */
  }
  eps->state = 1;
  return 0;
}

/* ---- building blocks ------------------------------------------------------------------------------ */
/* the operator of the solver: STApply of the eigensolver's ST (stsolve.c:44-72) */
static PetscErrorCode EPSApplyOperator(EPS eps, Vec x, Vec y) {
  PetscErrorCode ierr;
  ierr = STApply(eps->st, x, y);CHKERRQ(ierr);
  eps->matvecs++;
  return 0;
}

/* epssolve.c:675-710 */
static PetscErrorCode EPSGetStartVector(EPS eps, PetscInt i, PetscBool *breakdown) {
  PetscErrorCode ierr;
  PetscReal norm;
  PetscBool lindep;
  if (i > 0 || eps->nini == 0) { ierr = BVSetRandomColumn(eps->V, i);CHKERRQ(ierr); }
  ierr = BVOrthogonalizeColumn(eps->V, i, NULL, &norm, &lindep);CHKERRQ(ierr);
  if (breakdown) *breakdown = lindep;
  else if (lindep || norm == 0.0) {
    if (i == 0) SETERRQ(1, "Initial vector is zero or belongs to the deflation space");
    else SETERRQ(1, "Unable to generate more start vectors");
  }
  ierr = BVScaleColumn(eps->V, i, 1.0 / norm);CHKERRQ(ierr);
  return 0;
}

/* epskrylov.c:347-376 */
static PetscErrorCode EPSFullLanczos(EPS eps, PetscReal *alpha, PetscReal *beta, PetscInt k, PetscInt *M, PetscBool *breakdown) {
  PetscErrorCode ierr;
  PetscScalar *a;
  PetscInt j, nc = eps->V->nc, ld, m = *M;
  Vec vj, vj1;
  ierr = BVSetActiveColumns(eps->V, 0, m);CHKERRQ(ierr);
  for (j = k; j < m; j++) {
    ierr = BVGetColumn(eps->V, j, &vj);CHKERRQ(ierr);
    ierr = BVGetColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
    ierr = EPSApplyOperator(eps, vj, vj1);CHKERRQ(ierr);
    ierr = BVRestoreColumn(eps->V, j, &vj);CHKERRQ(ierr);
    ierr = BVRestoreColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
    ierr = BVOrthonormalizeColumn(eps->V, j + 1, PETSC_FALSE, beta + j, breakdown);CHKERRQ(ierr);
    if (*breakdown) { *M = j + 1; break; }
  }
  /* the diagonal of T sits in the BV buffer (beta is already in its place) */
  ierr = BVGetBuffer(eps->V, &a, &ld);CHKERRQ(ierr);
  for (j = k; j < *M; j++) alpha[j] = a[nc + j + (size_t)(j + 1) * ld];
  return 0;
}

/* epskrylov.c:39-74 */
static PetscErrorCode EPSBasicArnoldi(EPS eps, PetscScalar *H, PetscInt ldh, PetscInt k, PetscInt *M, PetscReal *beta, PetscBool *breakdown) {
  PetscErrorCode ierr;
  PetscScalar *a;
  PetscInt j, nc = eps->V->nc, ld, m = *M;
  Vec vj, vj1;
  ierr = BVSetActiveColumns(eps->V, 0, m);CHKERRQ(ierr);
  for (j = k; j < m; j++) {
    ierr = BVGetColumn(eps->V, j, &vj);CHKERRQ(ierr);
    ierr = BVGetColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
    ierr = EPSApplyOperator(eps, vj, vj1);CHKERRQ(ierr);
    ierr = BVRestoreColumn(eps->V, j, &vj);CHKERRQ(ierr);
    ierr = BVRestoreColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
    ierr = BVOrthonormalizeColumn(eps->V, j + 1, PETSC_FALSE, beta, breakdown);CHKERRQ(ierr);
    if (*breakdown) { *M = j + 1; break; }
  }
  /* extract the Hessenberg matrix from the BV buffer */
  ierr = BVGetBuffer(eps->V, &a, &ld);CHKERRQ(ierr);
  for (j = k; j < *M; j++) memcpy(H + (size_t)j * ldh, a + nc + (size_t)(j + 1) * ld, (size_t)(j + 2) * sizeof(PetscScalar));
  return 0;
}

/* epskrylov.c:82-186: Arnoldi with the reorthogonalisation of column j and the
 * normalisation of column j-1 postponed to step j+1, so that all reductions of a step can
 * travel together */
static PetscErrorCode EPSDelayedArnoldi(EPS eps, PetscScalar *H, PetscInt ldh, PetscInt k, PetscInt *M, PetscReal *beta, PetscBool *breakdown) {
  PetscErrorCode ierr;
  PetscInt i, j, m = *M;
  Vec u = NULL, t = NULL, vj, vj1;
  PetscScalar *lhh, dot = 0.0;
  PetscReal norm1 = 0.0, norm2 = 1.0;
  lhh = (PetscScalar *)malloc((size_t)(m + 1) * sizeof(*lhh));
  if (!lhh) SETERRQ(PETSC_ERR_MEM, "Out of memory");
#define DCHK(e) do { ierr = (e); if (ierr) { free(lhh); VecDestroy(&u); VecDestroy(&t); CHKERRQ(ierr); } } while (0)
  DCHK(VecDuplicate(eps->V->t, &u));
  DCHK(VecDuplicate(eps->V->t, &t));
  DCHK(BVSetActiveColumns(eps->V, 0, m));
  for (j = k; j < m; j++) {
    DCHK(BVGetColumn(eps->V, j, &vj));
    DCHK(BVGetColumn(eps->V, j + 1, &vj1));
    DCHK(EPSApplyOperator(eps, vj, vj1));
    DCHK(BVRestoreColumn(eps->V, j, &vj));
    DCHK(BVRestoreColumn(eps->V, j + 1, &vj1));

    /* every reduction of the step in one group (epskrylov.c:112-135) */
    DCHK(BVDotColumnBegin(eps->V, j + 1, H + (size_t)ldh * j));
    if (j > k) {
      DCHK(BVDotColumnBegin(eps->V, j, lhh));
      DCHK(BVGetColumn(eps->V, j, &vj));
      DCHK(VecDotBegin(vj, vj, &dot));
    }
    if (j > k + 1) DCHK(BVNormVecBegin(eps->V, u, NORM_2, &norm2));
    DCHK(BVDotColumnEnd(eps->V, j + 1, H + (size_t)ldh * j));
    if (j > k) {
      DCHK(BVDotColumnEnd(eps->V, j, lhh));
      DCHK(VecDotEnd(vj, vj, &dot));
      DCHK(BVRestoreColumn(eps->V, j, &vj));
    }
    if (j > k + 1) DCHK(BVNormVecEnd(eps->V, u, NORM_2, &norm2));

    if (j > k) {
      norm1 = sqrt(dot);
      for (i = 0; i < j; i++) H[(size_t)ldh * j + i] = H[(size_t)ldh * j + i] / norm1;
      H[(size_t)ldh * j + j] = H[(size_t)ldh * j + j] / dot;
      DCHK(BVCopyVec(eps->V, j, t));
      DCHK(BVScaleColumn(eps->V, j, 1.0 / norm1));
      DCHK(BVScaleColumn(eps->V, j + 1, 1.0 / norm1));
    }
    DCHK(BVMultColumn(eps->V, -1.0, 1.0, j + 1, H + (size_t)ldh * j));
    if (j > k) {
      DCHK(BVSetActiveColumns(eps->V, 0, j));
      DCHK(BVMultVec(eps->V, -1.0, 1.0, t, lhh));
      DCHK(BVSetActiveColumns(eps->V, 0, m));
      for (i = 0; i < j; i++) H[(size_t)ldh * (j - 1) + i] += lhh[i];
    }
    if (j > k + 1) {
      DCHK(BVInsertVec(eps->V, j - 1, u));
      DCHK(BVScaleColumn(eps->V, j - 1, 1.0 / norm2));
      H[(size_t)ldh * (j - 2) + j - 1] = norm2;
    }
    if (j < m - 1) DCHK(VecCopy(t, u));
  }
  DCHK(BVNormVec(eps->V, t, NORM_2, &norm2));
  DCHK(VecScale(t, 1.0 / norm2));
  DCHK(BVInsertVec(eps->V, m - 1, t));
  H[(size_t)ldh * (m - 2) + m - 1] = norm2;
  DCHK(BVDotColumn(eps->V, m, lhh));
  DCHK(BVMultColumn(eps->V, -1.0, 1.0, m, lhh));
  for (i = 0; i < m; i++) H[(size_t)ldh * (m - 1) + i] += lhh[i];
  DCHK(BVNormColumn(eps->V, m, NORM_2, beta));
  DCHK(BVScaleColumn(eps->V, m, 1.0 / *beta));
#undef DCHK
  *breakdown = PETSC_FALSE;
  free(lhh);
  ierr = VecDestroy(&u);CHKERRQ(ierr);
  ierr = VecDestroy(&t);CHKERRQ(ierr);
  return 0;
}

/* epskrylov.c:192-233: delayed normalisation only, no reorthogonalisation */
static PetscErrorCode EPSDelayedArnoldi1(EPS eps, PetscScalar *H, PetscInt ldh, PetscInt k, PetscInt *M, PetscReal *beta, PetscBool *breakdown) {
  PetscErrorCode ierr;
  PetscInt i, j, m = *M;
  PetscScalar dot;
  PetscReal norm;
  Vec vj, vj1;
  ierr = BVSetActiveColumns(eps->V, 0, m);CHKERRQ(ierr);
  for (j = k; j < m; j++) {
    ierr = BVGetColumn(eps->V, j, &vj);CHKERRQ(ierr);
    ierr = BVGetColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
    ierr = EPSApplyOperator(eps, vj, vj1);CHKERRQ(ierr);
    ierr = BVRestoreColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
    ierr = BVRestoreColumn(eps->V, j, &vj);CHKERRQ(ierr);
    ierr = BVDotColumnBegin(eps->V, j + 1, H + (size_t)ldh * j);CHKERRQ(ierr);
    if (j > k) {
      ierr = BVGetColumn(eps->V, j, &vj);CHKERRQ(ierr);
      ierr = VecDotBegin(vj, vj, &dot);CHKERRQ(ierr);
    }
    ierr = BVDotColumnEnd(eps->V, j + 1, H + (size_t)ldh * j);CHKERRQ(ierr);
    if (j > k) {
      ierr = VecDotEnd(vj, vj, &dot);CHKERRQ(ierr);
      ierr = BVRestoreColumn(eps->V, j, &vj);CHKERRQ(ierr);
      norm = sqrt(dot);
      ierr = BVScaleColumn(eps->V, j, 1.0 / norm);CHKERRQ(ierr);
      H[(size_t)ldh * (j - 1) + j] = norm;
      for (i = 0; i < j; i++) H[(size_t)ldh * j + i] = H[(size_t)ldh * j + i] / norm;
      H[(size_t)ldh * j + j] = H[(size_t)ldh * j + j] / dot;
      ierr = BVScaleColumn(eps->V, j + 1, 1.0 / norm);CHKERRQ(ierr);
      *beta = norm;
    }
    ierr = BVMultColumn(eps->V, -1.0, 1.0, j + 1, H + (size_t)ldh * j);CHKERRQ(ierr);
  }
  *breakdown = PETSC_FALSE;
  return 0;
}

/* epskrylov.c:252-327 (trivial region, estimated residuals) with the relative
 * convergence test EPSConvergedRelative (epsdefault.c:258-266) */
static PetscErrorCode EPSKrylovConvergence(EPS eps, PetscInt kini, PetscInt nits, PetscReal beta, PetscReal corrf, PetscInt *kout) {
  PetscErrorCode ierr;
  PetscInt k, newk, marker = -1;
  PetscReal resnorm;
  for (k = kini; k < kini + nits; k++) {
    PetscScalar re = eps->eigr[k], im = eps->eigi[k];
    if (STIsShift_Private(eps->st)) { ierr = STBackTransform(eps->st, 1, &re, &im);CHKERRQ(ierr); }    /* epskrylov.c:285-287 */
    newk = k;
    ierr = DSVectors(eps->ds, DS_MAT_X, &newk, &resnorm);CHKERRQ(ierr);
    resnorm *= beta * corrf;
    /* EPSConvergedRelative / EPSConvergedAbsolute (epsdefault.c:258-280) */
    eps->errest[k] = eps->conv == EPS_CONV_ABS ? resnorm : resnorm / SlepcAbsEigenvalue(re, im);
    if (marker == -1 && eps->errest[k] >= eps->tol) marker = k;
    if (newk == k + 1) { eps->errest[k + 1] = eps->errest[k]; k++; }
    if (marker != -1) break;
  }
  if (marker != -1) k = marker;
  *kout = k;
  return 0;
}

/* epsdefault.c:324-338 */
static void EPSStoppingBasic(EPS eps, PetscInt its, PetscInt max_it, PetscInt nconv, PetscInt nev, EPSConvergedReason *reason) {
  *reason = EPS_CONVERGED_ITERATING;
  if (nconv >= nev) *reason = EPS_CONVERGED_TOL;
  else if (its >= max_it) *reason = EPS_DIVERGED_ITS;
}

/* V(:,s:e) = V * Q(:,s:e) with Q taken from the DS (DSGetMat + BVMultInPlace + MatDestroy) */
static PetscErrorCode UpdateBasis(EPS eps, DSMatType m, PetscInt s, PetscInt e) {
  PetscErrorCode ierr;
  PetscInt rows, cols;
  ierr = DSGetMat(eps->ds, m, NULL, &rows, &cols);CHKERRQ(ierr);
  PetscScalar *U = (PetscScalar *)malloc((size_t)PetscMax(rows * cols, 1) * sizeof(PetscScalar));
  if (!U) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  ierr = DSGetMat(eps->ds, m, U, &rows, &cols);
  if (!ierr) ierr = BVMultInPlace(eps->V, U, rows, s, e);
  free(U);
  CHKERRQ(ierr);
  return 0;
}

/* ---- ks-symm.c:30-111 ----------------------------------------------------------------------------- */
static PetscErrorCode EPSSolve_KrylovSchur_Symm(EPS eps) {
  PetscErrorCode ierr;
  PetscInt k, l, ld, nv, nconv;
  PetscReal *a, *b, beta;
  PetscBool breakdown;
  ierr = DSGetLeadingDimension(eps->ds, &ld);CHKERRQ(ierr);
  /* get the starting Lanczos vector */
  ierr = EPSGetStartVector(eps, 0, NULL);CHKERRQ(ierr);
  l = 0;
  /* restart loop */
  while (eps->reason == EPS_CONVERGED_ITERATING) {
    eps->its++;
    /* compute an nv-step Lanczos factorization */
    nv = PetscMin(eps->nconv + eps->mpd, eps->ncv);
    ierr = DSGetArrayReal(eps->ds, DS_MAT_T, &a);CHKERRQ(ierr);
    b = a + ld;
    ierr = EPSFullLanczos(eps, a, b, eps->nconv + l, &nv, &breakdown);CHKERRQ(ierr);
    beta = b[nv - 1];
    ierr = DSSetDimensions(eps->ds, nv, 0, eps->nconv, eps->nconv + l);CHKERRQ(ierr);
    ierr = DSSetState(eps->ds, l == 0 ? DS_STATE_INTERMEDIATE : DS_STATE_RAW);CHKERRQ(ierr);
    ierr = BVSetActiveColumns(eps->V, eps->nconv, nv);CHKERRQ(ierr);
    /* solve projected problem */
    ierr = DSSolve(eps->ds, eps->eigr, NULL);CHKERRQ(ierr);
    ierr = DSSort(eps->ds, eps->eigr, NULL, NULL, NULL, NULL);CHKERRQ(ierr);
    ierr = DSUpdateExtraRow(eps->ds);CHKERRQ(ierr);
    /* check convergence */
    ierr = EPSKrylovConvergence(eps, eps->nconv, nv - eps->nconv, beta, 1.0, &k);CHKERRQ(ierr);
    EPSStoppingBasic(eps, eps->its, eps->max_it, k, eps->nev, &eps->reason);
    nconv = k;
    /* update l */
    if (eps->reason != EPS_CONVERGED_ITERATING || breakdown) l = 0;
    else l = PetscMax(1, (PetscInt)((nv - k) * eps->keep));
    if (!eps->lock && l > 0) { l += k; k = 0; }     /* non-locking variant: reset no. of converged pairs */
    if (eps->reason == EPS_CONVERGED_ITERATING) {
      if (breakdown) {
        /* start a new Lanczos factorization */
        if (k < eps->nev) {
          ierr = EPSGetStartVector(eps, k, &breakdown);CHKERRQ(ierr);
          if (breakdown) eps->reason = EPS_DIVERGED_BREAKDOWN;
        }
      } else {
        /* prepare the Rayleigh quotient for restart */
        ierr = DSTruncate(eps->ds, k + l);CHKERRQ(ierr);
      }
    }
    /* update the corresponding vectors V(:,idx) = V*Q(:,idx) */
    ierr = UpdateBasis(eps, DS_MAT_Q, eps->nconv, k + l);CHKERRQ(ierr);
    /* normalize u and append it to V */
    if (eps->reason == EPS_CONVERGED_ITERATING && !breakdown) { ierr = BVCopyColumn(eps->V, nv, k + l);CHKERRQ(ierr); }
    eps->nconv = k;
    if (eps->monitor) { ierr = eps->monitor(eps, eps->its, nconv, eps->eigr, eps->eigi, eps->errest, nv, eps->monitorctx);CHKERRQ(ierr); }
  }
  return 0;
}

/* ---- krylovschur.c:169-288 (Ritz extraction) ---------------------------------------------------------- */
static PetscErrorCode EPSSolve_KrylovSchur_Default(EPS eps) {
  PetscErrorCode ierr;
  PetscInt i, k, l, nv, ld, nconv;
  PetscScalar *S, *Q;
  PetscReal beta;
  PetscBool breakdown;
  ierr = DSGetLeadingDimension(eps->ds, &ld);CHKERRQ(ierr);
  /* get the starting Arnoldi vector */
  ierr = EPSGetStartVector(eps, 0, NULL);CHKERRQ(ierr);
  l = 0;
  while (eps->reason == EPS_CONVERGED_ITERATING) {
    eps->its++;
    /* compute an nv-step Arnoldi factorization */
    nv = PetscMin(eps->nconv + eps->mpd, eps->ncv);
    ierr = DSGetArray(eps->ds, DS_MAT_A, &S);CHKERRQ(ierr);
    ierr = EPSBasicArnoldi(eps, S, ld, eps->nconv + l, &nv, &beta, &breakdown);CHKERRQ(ierr);
    ierr = DSSetDimensions(eps->ds, nv, 0, eps->nconv, eps->nconv + l);CHKERRQ(ierr);
    ierr = DSSetState(eps->ds, l == 0 ? DS_STATE_INTERMEDIATE : DS_STATE_RAW);CHKERRQ(ierr);
    ierr = BVSetActiveColumns(eps->V, eps->nconv, nv);CHKERRQ(ierr);
    /* solve projected problem */
    ierr = DSSolve(eps->ds, eps->eigr, eps->eigi);CHKERRQ(ierr);
    ierr = DSSort(eps->ds, eps->eigr, eps->eigi, NULL, NULL, NULL);CHKERRQ(ierr);
    /* check convergence */
    ierr = EPSKrylovConvergence(eps, eps->nconv, nv - eps->nconv, beta, 1.0, &k);CHKERRQ(ierr);
    EPSStoppingBasic(eps, eps->its, eps->max_it, k, eps->nev, &eps->reason);
    nconv = k;
    /* update l */
    if (eps->reason != EPS_CONVERGED_ITERATING || breakdown) l = 0;
    else {
      l = PetscMax(1, (PetscInt)((nv - k) * eps->keep));
      if (S[k + l + (size_t)(k + l - 1) * ld] != 0.0) {      /* do not split a 2x2 block */
        if (k + l < nv - 1) l = l + 1;
        else l = l - 1;
      }
    }
    if (!eps->lock && l > 0) { l += k; k = 0; }
    if (eps->reason == EPS_CONVERGED_ITERATING) {
      if (breakdown) {
        if (k < eps->nev) {
          ierr = EPSGetStartVector(eps, k, &breakdown);CHKERRQ(ierr);
          if (breakdown) eps->reason = EPS_DIVERGED_BREAKDOWN;
        }
      } else {
        /* prepare the Rayleigh quotient for restart */
        ierr = DSGetArray(eps->ds, DS_MAT_Q, &Q);CHKERRQ(ierr);
        for (i = k; i < k + l; i++) S[k + l + (size_t)i * ld] = Q[nv - 1 + (size_t)i * ld] * beta;
      }
    }
    /* update the corresponding vectors V(:,idx) = V*Q(:,idx) */
    ierr = UpdateBasis(eps, DS_MAT_Q, eps->nconv, k + l);CHKERRQ(ierr);
    if (eps->reason == EPS_CONVERGED_ITERATING && !breakdown) { ierr = BVCopyColumn(eps->V, nv, k + l);CHKERRQ(ierr); }
    eps->nconv = k;
    if (eps->monitor) { ierr = eps->monitor(eps, eps->its, nconv, eps->eigr, eps->eigi, eps->errest, nv, eps->monitorctx);CHKERRQ(ierr); }
  }
  /* truncate the Schur decomposition and change the state to raw so that
     DSVectors() computes eigenvectors from scratch */
  ierr = DSSetDimensions(eps->ds, eps->nconv, 0, 0, 0);CHKERRQ(ierr);
  ierr = DSSetState(eps->ds, DS_STATE_RAW);CHKERRQ(ierr);
  return 0;
}

/* ---- epsdefault.c:40-74,136-205 -------------------------------------------------------------------------- */
/* arnoldi.c:78-154: explicitly restarted Arnoldi with locking (Ritz extraction, the
 * orthogonalisation of every step completed before the next one starts) */
static PetscErrorCode EPSSolve_Arnoldi(EPS eps) {
  PetscErrorCode ierr;
  PetscInt k, nv, ld;
  PetscScalar *H;
  PetscReal beta;
  PetscBool breakdown;
  ierr = DSGetLeadingDimension(eps->ds, &ld);CHKERRQ(ierr);
  ierr = EPSGetStartVector(eps, 0, NULL);CHKERRQ(ierr);
  while (eps->reason == EPS_CONVERGED_ITERATING) {
    eps->its++;
    /* compute an nv-step Arnoldi factorization; the locked Schur vectors stay in front */
    nv = PetscMin(eps->nconv + eps->mpd, eps->ncv);
    ierr = DSGetArray(eps->ds, DS_MAT_A, &H);CHKERRQ(ierr);
    if (!eps->delayed) { ierr = EPSBasicArnoldi(eps, H, ld, eps->nconv, &nv, &beta, &breakdown);CHKERRQ(ierr); }
    else if (eps->V->orthog_ref == BV_ORTHOG_REFINE_NEVER) { ierr = EPSDelayedArnoldi1(eps, H, ld, eps->nconv, &nv, &beta, &breakdown);CHKERRQ(ierr); }
    else { ierr = EPSDelayedArnoldi(eps, H, ld, eps->nconv, &nv, &beta, &breakdown);CHKERRQ(ierr); }
    ierr = DSSetDimensions(eps->ds, nv, 0, eps->nconv, 0);CHKERRQ(ierr);
    ierr = DSSetState(eps->ds, DS_STATE_INTERMEDIATE);CHKERRQ(ierr);
    ierr = BVSetActiveColumns(eps->V, eps->nconv, nv);CHKERRQ(ierr);
    /* solve projected problem */
    ierr = DSSolve(eps->ds, eps->eigr, eps->eigi);CHKERRQ(ierr);
    ierr = DSSort(eps->ds, eps->eigr, eps->eigi, NULL, NULL, NULL);CHKERRQ(ierr);
    ierr = DSUpdateExtraRow(eps->ds);CHKERRQ(ierr);
    /* check convergence, then restart with the first unconverged Schur vector */
    ierr = EPSKrylovConvergence(eps, eps->nconv, nv - eps->nconv, beta, 1.0, &k);CHKERRQ(ierr);
    ierr = UpdateBasis(eps, DS_MAT_Q, eps->nconv, PetscMin(k + 1, nv));CHKERRQ(ierr);
    EPSStoppingBasic(eps, eps->its, eps->max_it, k, eps->nev, &eps->reason);
    if (eps->reason == EPS_CONVERGED_ITERATING && breakdown) {
      ierr = EPSGetStartVector(eps, k, &breakdown);CHKERRQ(ierr);
      if (breakdown) eps->reason = EPS_DIVERGED_BREAKDOWN;
    }
    eps->nconv = k;
    if (eps->monitor) { ierr = eps->monitor(eps, eps->its, eps->nconv, eps->eigr, eps->eigi, eps->errest, nv, eps->monitorctx);CHKERRQ(ierr); }
  }
  ierr = DSSetDimensions(eps->ds, eps->nconv, 0, 0, 0);CHKERRQ(ierr);
  ierr = DSSetState(eps->ds, DS_STATE_RAW);CHKERRQ(ierr);
  return 0;
}

/* ---- Lanczos recurrences with cheaper reorthogonalisation (lanczos.c:110-541) ----------------- */
/* one Lanczos step with local reorthogonalisation: column j+1 = OP v_j made orthogonal to the
 * locked columns and to v_{j-1}, v_j */
static PetscErrorCode LanczosLocalStep(EPS eps, PetscInt j, PetscInt k, PetscBool *which, PetscScalar *hwork, PetscReal *alpha, PetscReal *norm, PetscBool *breakdown) {
  PetscErrorCode ierr;
  Vec vj, vj1;
  ierr = BVGetColumn(eps->V, j, &vj);CHKERRQ(ierr);
  ierr = BVGetColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
  ierr = EPSApplyOperator(eps, vj, vj1);CHKERRQ(ierr);
  ierr = BVRestoreColumn(eps->V, j, &vj);CHKERRQ(ierr);
  ierr = BVRestoreColumn(eps->V, j + 1, &vj1);CHKERRQ(ierr);
  which[j] = PETSC_TRUE;
  if (j - 2 >= k) which[j - 2] = PETSC_FALSE;
  ierr = BVOrthogonalizeSomeColumn(eps->V, j + 1, which, hwork, norm, breakdown);CHKERRQ(ierr);
  *alpha = hwork[j];
  return 0;
}

/* lanczos.c:110-148 */
static PetscErrorCode EPSLocalLanczos(EPS eps, PetscReal *alpha, PetscReal *beta, PetscInt k, PetscInt *M, PetscBool *breakdown, PetscBool *which, PetscScalar *hwork) {
  PetscErrorCode ierr;
  PetscInt i, j, m = *M;
  for (i = 0; i <= m; i++) which[i] = i < k ? PETSC_TRUE : PETSC_FALSE;
  ierr = BVSetActiveColumns(eps->V, 0, m);CHKERRQ(ierr);
  for (j = k; j < m; j++) {
    ierr = LanczosLocalStep(eps, j, k, which, hwork, &alpha[j], &beta[j], breakdown);CHKERRQ(ierr);
    if (*breakdown) { *M = j + 1; break; }
    ierr = BVScaleColumn(eps->V, j + 1, 1.0 / beta[j]);CHKERRQ(ierr);
  }
  return 0;
}

/* lanczos.c:213-288: local reorthogonalisation plus orthogonalisation against the nearly
 * converged Ritz vectors, which are kept in eps->AV */
static PetscErrorCode EPSSelectiveLanczos(EPS eps, PetscReal *alpha, PetscReal *beta, PetscInt k, PetscInt *M, PetscBool *breakdown, PetscReal anorm, PetscBool *which, PetscScalar *hwork) {
  PetscErrorCode ierr = 0;
  PetscInt i, j, m = *M, n, nritz = 0, nritzo;
  Vec vj1, av;
  PetscReal *d, *e, norm;
  PetscScalar *Y;
  const PetscReal sqrteps = sqrt(PETSC_MACHINE_EPSILON);
  d = (PetscReal *)calloc((size_t)2 * (m + 1), sizeof(*d));
  Y = (PetscScalar *)calloc((size_t)m * m + 1, sizeof(*Y));
  if (!d || !Y) { free(d); free(Y); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  e = d + m + 1;
#define SCHK(call) do { ierr = (call); if (ierr) { free(d); free(Y); CHKERRQ(ierr); } } while (0)
  for (i = 0; i <= m; i++) which[i] = i < k ? PETSC_TRUE : PETSC_FALSE;
  for (j = k; j < m; j++) {
    SCHK(BVSetActiveColumns(eps->V, 0, m));
    SCHK(LanczosLocalStep(eps, j, k, which, hwork, &alpha[j], &norm, breakdown));
    beta[j] = norm;
    if (*breakdown) { *M = j + 1; break; }
    /* eigenvalues and eigenvectors Y of the tridiagonal block (DenseTridiagonal, :150-211) */
    n = j - k + 1;
    for (i = 0; i < n; i++) { d[i] = alpha[i + k]; e[i] = beta[i + k]; }
    memset(Y, 0, (size_t)n * n * sizeof(*Y));
    for (i = 0; i < n; i++) Y[i + (size_t)i * n] = 1.0;
    if (ksd_steqr(n, d, e, Y, n, n)) { free(d); free(Y); SETERRQ(PETSC_ERR_LIB, "Error in the tridiagonal eigensolver"); }
    for (i = 0; i < n; i++) if (fabs(d[i]) > anorm) anorm = fabs(d[i]);          /* estimate ||A|| */
    /* nearly converged Ritz vectors */
    nritzo = 0;
    for (i = 0; i < n; i++) if (norm * fabs(Y[(size_t)i * n + n - 1]) < sqrteps * anorm) nritzo++;
    if (nritzo > nritz) {
      nritz = 0;
      for (i = 0; i < n; i++) {
        if (norm * fabs(Y[(size_t)i * n + n - 1]) < sqrteps * anorm) {
          SCHK(BVSetActiveColumns(eps->V, k, k + n));
          SCHK(BVGetColumn(eps->AV, nritz, &av));
          SCHK(BVMultVec(eps->V, 1.0, 0.0, av, Y + (size_t)i * n));
          SCHK(BVRestoreColumn(eps->AV, nritz, &av));
          nritz++;
        }
      }
    }
    if (nritz > 0) {
      SCHK(BVGetColumn(eps->V, j + 1, &vj1));
      SCHK(BVSetActiveColumns(eps->AV, 0, nritz));
      SCHK(BVOrthogonalizeVec(eps->AV, vj1, hwork, &norm, breakdown));
      SCHK(BVRestoreColumn(eps->V, j + 1, &vj1));
      if (*breakdown) { *M = j + 1; break; }
    }
    SCHK(BVScaleColumn(eps->V, j + 1, 1.0 / norm));
  }
#undef SCHK
  free(d); free(Y);
  return 0;
}

/* lanczos.c:290-318: recurrence for the estimated loss of orthogonality; beta is the
 * reference's beta-1 (beta[i] here is the norm of step i-1) */
static void update_omega(PetscReal *omega, PetscReal *omega_old, PetscInt j, const PetscReal *alpha, const PetscReal *beta, PetscReal eps1, PetscReal anorm) {
  PetscInt k;
  const PetscReal T = eps1 * anorm, binv = 1.0 / beta[j + 1];
  omega_old[0] = beta[1] * omega[1] + (alpha[0] - alpha[j]) * omega[0] - beta[j] * omega_old[0];
  omega_old[0] = omega_old[0] > 0 ? binv * (omega_old[0] + T) : binv * (omega_old[0] - T);
  for (k = 1; k < j - 1; k++) {
    omega_old[k] = beta[k + 1] * omega[k + 1] + (alpha[k] - alpha[j]) * omega[k] + beta[k] * omega[k - 1] - beta[j] * omega_old[k];
    omega_old[k] = omega_old[k] > 0 ? binv * (omega_old[k] + T) : binv * (omega_old[k] - T);
  }
  omega_old[j - 1] = binv * T;
  for (k = 0; k < j; k++) { omega[k] = omega_old[k]; omega_old[k] = omega[k]; }
  omega[j] = eps1;
}

/* lanczos.c:320-357: the columns to reorthogonalise against, as intervals around the entries
 * of mu that exceed delta, extended while they exceed eta */
static void compute_int(PetscBool *which, const PetscReal *mu, PetscInt j, PetscReal delta, PetscReal eta) {
  PetscInt i, k, maxpos = 0;
  PetscReal max = 0.0;
  PetscBool found = PETSC_FALSE;
  for (i = 0; i < j; i++) {
    if (fabs(mu[i]) >= delta) { which[i] = PETSC_TRUE; found = PETSC_TRUE; }
    else which[i] = PETSC_FALSE;
    if (fabs(mu[i]) > max) { maxpos = i; max = fabs(mu[i]); }
  }
  if (!found) which[maxpos] = PETSC_TRUE;
  for (i = 0; i < j; i++) {
    if (which[i]) {
      for (k = i; k >= 0; k--) { if (fabs(mu[k]) < eta || which[k]) break; else which[k] = PETSC_TRUE; }
      for (k = i + 1; k < j; k++) { if (fabs(mu[k]) < eta || which[k]) break; else which[k] = PETSC_TRUE; }
    }
  }
}

/* lanczos.c:359-468: periodic or partial reorthogonalisation */
static PetscErrorCode EPSPartialLanczos(EPS eps, PetscReal *alpha, PetscReal *beta, PetscInt k, PetscInt *M, PetscBool *breakdown, PetscReal anorm, PetscBool *which, PetscScalar *hwork) {
  PetscErrorCode ierr = 0;
  PetscInt i, j, m = *M;
  Vec vj, vj1;
  PetscReal norm = 0.0, *omega, *omega_old, eps1, delta, eta;
  PetscBool *which2, reorth = PETSC_FALSE, force_reorth = PETSC_FALSE, fro = PETSC_FALSE, estimate_anorm = PETSC_FALSE;
  omega = (PetscReal *)calloc((size_t)2 * (m + 2), sizeof(*omega));
  which2 = (PetscBool *)calloc((size_t)m + 2, sizeof(*which2));
  if (!omega || !which2) { free(omega); free(which2); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  omega_old = omega + m + 2;
#define PCHK(call) do { ierr = (call); if (ierr) { free(omega); free(which2); CHKERRQ(ierr); } } while (0)
  eps1 = sqrt((PetscReal)eps->n) * PETSC_MACHINE_EPSILON / 2;
  delta = sqrt(PETSC_MACHINE_EPSILON) / sqrt((PetscReal)eps->ncv);
  eta = pow(PETSC_MACHINE_EPSILON, 3.0 / 4.0) / sqrt((PetscReal)eps->ncv);
  if (anorm < 0.0) { anorm = 1.0; estimate_anorm = PETSC_TRUE; }
  for (i = 0; i <= m; i++) which[i] = i < k ? PETSC_TRUE : PETSC_FALSE;
  PCHK(BVSetActiveColumns(eps->V, 0, m));
  for (j = k; j < m; j++) {
    if (fro) {
      /* Lanczos step with full reorthogonalisation */
      PCHK(BVGetColumn(eps->V, j, &vj));
      PCHK(BVGetColumn(eps->V, j + 1, &vj1));
      PCHK(EPSApplyOperator(eps, vj, vj1));
      PCHK(BVRestoreColumn(eps->V, j, &vj));
      PCHK(BVRestoreColumn(eps->V, j + 1, &vj1));
      PCHK(BVOrthogonalizeColumn(eps->V, j + 1, hwork, &norm, breakdown));
      alpha[j] = hwork[j];
    } else {
      PCHK(LanczosLocalStep(eps, j, k, which, hwork, &alpha[j], &norm, breakdown));
      beta[j] = norm;
      if (estimate_anorm) {
        if (j > k) anorm = PetscMax(anorm, fabs(alpha[j]) + norm + beta[j - 1]);
        else anorm = PetscMax(anorm, fabs(alpha[j]) + norm);
      }
      /* is reorthogonalisation needed? */
      reorth = PETSC_FALSE;
      if (j > k) {
        update_omega(omega, omega_old, j, alpha, beta - 1, eps1, anorm);
        for (i = 0; i < j - k; i++) if (fabs(omega[i]) > delta) reorth = PETSC_TRUE;
      }
      if (reorth || force_reorth) {
        for (i = 0; i < k; i++) which2[i] = PETSC_FALSE;
        for (i = k; i <= j; i++) which2[i] = PETSC_TRUE;
        if (eps->reorthog == EPS_LANCZOS_REORTHOG_PERIODIC) {
          force_reorth = force_reorth ? PETSC_FALSE : PETSC_TRUE;
          for (i = 0; i < j - k; i++) omega[i] = eps1;
        } else {
          if (force_reorth) force_reorth = PETSC_FALSE;
          else {
            force_reorth = PETSC_TRUE;
            compute_int(which2 + k, omega, j - k, delta, eta);
            for (i = 0; i < j - k; i++) if (which2[i + k]) omega[i] = eps1;
          }
        }
        PCHK(BVOrthogonalizeSomeColumn(eps->V, j + 1, which2, hwork, &norm, breakdown));
      }
    }
    if (*breakdown || norm < eps->n * anorm * PETSC_MACHINE_EPSILON) { *M = j + 1; break; }
    if (!fro && norm * delta < anorm * eps1) fro = PETSC_TRUE;      /* switch to full reorthogonalisation */
    beta[j] = norm;
    PCHK(BVScaleColumn(eps->V, j + 1, 1.0 / norm));
  }
#undef PCHK
  free(omega); free(which2);
  return 0;
}

/* lanczos.c:470-541 */
static PetscErrorCode EPSBasicLanczos(EPS eps, PetscReal *alpha, PetscReal *beta, PetscInt k, PetscInt *m, PetscBool *breakdown, PetscReal anorm) {
  PetscErrorCode ierr = 0;
  PetscInt i, n = *m;
  PetscReal betam;
  PetscScalar *T;
  if (eps->reorthog == EPS_LANCZOS_REORTHOG_FULL) return EPSFullLanczos(eps, alpha, beta, k, m, breakdown);
  if (eps->reorthog == EPS_LANCZOS_REORTHOG_DELAYED) {
    T = (PetscScalar *)calloc((size_t)n * n + 1, sizeof(*T));
    if (!T) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    if (eps->V->orthog_ref == BV_ORTHOG_REFINE_NEVER) ierr = EPSDelayedArnoldi1(eps, T, n, k, m, &betam, breakdown);
    else ierr = EPSDelayedArnoldi(eps, T, n, k, m, &betam, breakdown);
    if (!ierr) {
      for (i = k; i < n - 1; i++) { alpha[i] = T[(size_t)n * i + i]; beta[i] = T[(size_t)n * i + i + 1]; }
      alpha[n - 1] = T[(size_t)n * (n - 1) + n - 1];
      beta[n - 1] = betam;
    }
    free(T);
    CHKERRQ(ierr);
    return 0;
  }
  PetscBool *which = (PetscBool *)calloc((size_t)n + 2, sizeof(*which));
  PetscScalar *hwork = (PetscScalar *)calloc((size_t)n + 2, sizeof(*hwork));
  if (!which || !hwork) { free(which); free(hwork); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  switch (eps->reorthog) {
    case EPS_LANCZOS_REORTHOG_LOCAL: ierr = EPSLocalLanczos(eps, alpha, beta, k, m, breakdown, which, hwork); break;
    case EPS_LANCZOS_REORTHOG_SELECTIVE: ierr = EPSSelectiveLanczos(eps, alpha, beta, k, m, breakdown, anorm, which, hwork); break;
    default: ierr = EPSPartialLanczos(eps, alpha, beta, k, m, breakdown, anorm, which, hwork); break;
  }
  free(which); free(hwork);
  CHKERRQ(ierr);
  return 0;
}

/* lanczos.c:543-726: explicit restart with the wanted unconverged Ritz vector, converged Ritz
 * vectors locked in front; with local reorthogonalisation repeated and spurious Ritz values
 * are purged */
static PetscErrorCode EPSSolve_Lanczos(EPS eps) {
  PetscErrorCode ierr;
  PetscInt nconv = 0, i, j, k, l, x, n, restart, ncv = eps->ncv, r, ld;
  PetscScalar *Y, *ritz, stmp;
  PetscReal *d, *e, *bnd, anorm = -1.0, beta, rtmp, resnorm, norm;
  PetscInt *perm;
  PetscBool breakdown;
  char *conv, ctmp;
  Vec vi, vj, w;
  const PetscBool local = eps->reorthog == EPS_LANCZOS_REORTHOG_LOCAL ? PETSC_TRUE : PETSC_FALSE;
  ierr = DSGetLeadingDimension(eps->ds, &ld);CHKERRQ(ierr);
  ritz = (PetscScalar *)calloc((size_t)ncv + 1, sizeof(*ritz));
  bnd = (PetscReal *)calloc((size_t)ncv + 1, sizeof(*bnd));
  perm = (PetscInt *)calloc((size_t)ncv + 1, sizeof(*perm));
  conv = (char *)calloc((size_t)ncv + 1, 1);
  if (!ritz || !bnd || !perm || !conv) { free(ritz); free(bnd); free(perm); free(conv); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
#define LCHK(call) do { ierr = (call); if (ierr) { free(ritz); free(bnd); free(perm); free(conv); return ierr; } } while (0)
  LCHK(EPSGetStartVector(eps, 0, NULL));
  while (eps->reason == EPS_CONVERGED_ITERATING) {
    eps->its++;
    /* compute an ncv-step Lanczos factorization */
    n = PetscMin(nconv + eps->mpd, ncv);
    LCHK(DSGetArrayReal(eps->ds, DS_MAT_T, &d));
    e = d + ld;
    LCHK(EPSBasicLanczos(eps, d, e, nconv, &n, &breakdown, anorm));
    beta = e[n - 1];
    LCHK(DSSetDimensions(eps->ds, n, 0, nconv, 0));
    LCHK(DSSetState(eps->ds, DS_STATE_INTERMEDIATE));
    LCHK(BVSetActiveColumns(eps->V, nconv, n));
    /* solve projected problem */
    LCHK(DSSolve(eps->ds, ritz, NULL));
    LCHK(DSSort(eps->ds, ritz, NULL, NULL, NULL, NULL));
    /* estimate ||A||, then residual estimates beta*abs(Y(m,:)) + eps*||A|| */
    for (i = nconv; i < n; i++) anorm = PetscMax(anorm, fabs(ritz[i]));
    LCHK(DSGetArray(eps->ds, DS_MAT_Q, &Y));
    for (i = nconv; i < n; i++) {
      resnorm = beta * fabs(Y[n - 1 + (size_t)i * ld]) + PETSC_MACHINE_EPSILON * anorm;
      bnd[i] = eps->conv == EPS_CONV_ABS ? resnorm : resnorm / fabs(ritz[i]);
      conv[i] = bnd[i] < eps->tol ? 'C' : 'N';
    }
    /* purge repeated ritz values */
    if (local) for (i = nconv + 1; i < n; i++) if (conv[i] == 'C' && fabs((ritz[i] - ritz[i - 1]) / ritz[i]) < eps->tol) conv[i] = 'R';
    /* restart vector: the wanted one among the unconverged Ritz vectors */
    if (!breakdown) {
      restart = nconv;
      while (restart < n && conv[restart] != 'N') restart++;
      if (restart >= n) breakdown = PETSC_TRUE;
      else {
        for (i = restart + 1; i < n; i++) {
          if (conv[i] == 'N') {
            LCHK(eps->cmp(ritz[restart], 0.0, ritz[i], 0.0, &r, eps->cmpctx));
            if (r > 0) restart = i;
          }
        }
        LCHK(BVMultColumn(eps->V, 1.0, 0.0, n, Y + (size_t)restart * ld + nconv));
      }
    }
    /* count and put converged eigenvalues first */
    for (i = nconv; i < n; i++) perm[i] = i;
    for (k = nconv; k < n; k++) {
      if (conv[perm[k]] != 'C') {
        j = k + 1;
        while (j < n && conv[perm[j]] != 'C') j++;
        if (j >= n) break;
        l = perm[k]; perm[k] = perm[j]; perm[j] = l;
      }
    }
    /* sort eigenvectors according to the permutation */
    for (i = nconv; i < k; i++) {
      x = perm[i];
      if (x != i) {
        j = i + 1;
        while (perm[j] != i) j++;
        stmp = ritz[x]; ritz[x] = ritz[i]; ritz[i] = stmp;
        rtmp = bnd[x]; bnd[x] = bnd[i]; bnd[i] = rtmp;
        ctmp = conv[x]; conv[x] = conv[i]; conv[i] = ctmp;
        perm[j] = x; perm[i] = i;
        for (l = 0; l < n; l++) { stmp = Y[l + (size_t)x * ld]; Y[l + (size_t)x * ld] = Y[l + (size_t)i * ld]; Y[l + (size_t)i * ld] = stmp; }
      }
    }
    /* compute converged eigenvectors */
    LCHK(UpdateBasis(eps, DS_MAT_Q, nconv, k));
    /* purge spurious ritz values: the true residual decides */
    if (local) {
      w = eps->work[0];
      for (i = nconv; i < k; i++) {
        LCHK(BVGetColumn(eps->V, i, &vi));
        LCHK(VecNorm(vi, NORM_2, &norm));
        LCHK(VecScale(vi, 1.0 / norm));
        LCHK(EPSApplyOperator(eps, vi, w));
        LCHK(VecAXPY(w, -ritz[i], vi));
        LCHK(BVRestoreColumn(eps->V, i, &vi));
        LCHK(VecNorm(w, NORM_2, &norm));
        bnd[i] = eps->conv == EPS_CONV_ABS ? norm : norm / fabs(ritz[i]);
        if (bnd[i] >= eps->tol) conv[i] = 'S';
      }
      for (i = nconv; i < k; i++) {
        if (conv[i] != 'C') {
          j = i + 1;
          while (j < k && conv[j] != 'C') j++;
          if (j >= k) break;
          stmp = ritz[j]; ritz[j] = ritz[i]; ritz[i] = stmp;
          rtmp = bnd[j]; bnd[j] = bnd[i]; bnd[i] = rtmp;
          ctmp = conv[j]; conv[j] = conv[i]; conv[i] = ctmp;
          /* swap eigenvectors i and j through the work vector */
          LCHK(BVGetColumn(eps->V, i, &vi));
          LCHK(BVGetColumn(eps->V, j, &vj));
          LCHK(VecCopy(vi, w));
          LCHK(VecCopy(vj, vi));
          LCHK(VecCopy(w, vj));
          LCHK(BVRestoreColumn(eps->V, i, &vi));
          LCHK(BVRestoreColumn(eps->V, j, &vj));
        }
      }
      k = i;
    }
    /* store ritz values and estimated errors */
    for (i = nconv; i < n; i++) { eps->eigr[i] = ritz[i]; eps->errest[i] = bnd[i]; }
    nconv = k;
    if (eps->monitor) LCHK(eps->monitor(eps, eps->its, nconv, eps->eigr, eps->eigi, eps->errest, n, eps->monitorctx));
    EPSStoppingBasic(eps, eps->its, eps->max_it, nconv, eps->nev, &eps->reason);
    if (eps->reason == EPS_CONVERGED_ITERATING) {       /* copy restart vector */
      LCHK(BVCopyColumn(eps->V, n, nconv));
      if (local && !breakdown) LCHK(BVOrthonormalizeColumn(eps->V, nconv, PETSC_FALSE, NULL, &breakdown));   /* reorthonormalise the restart vector */
      if (breakdown) {                                  /* use a random vector for restarting */
        LCHK(EPSGetStartVector(eps, nconv, &breakdown));
        if (breakdown) eps->reason = EPS_DIVERGED_BREAKDOWN;
      }
    }
  }
  eps->nconv = nconv;
#undef LCHK
  free(ritz); free(bnd); free(perm); free(conv);
  return 0;
}

/* epsimpl.h:166-183: each of the first k basis vectors through the operator once more */
static PetscErrorCode EPS_Purify(EPS eps, PetscInt k) {
  PetscErrorCode ierr;
  Vec v = NULL, z;
  ierr = VecDuplicate(eps->V->t, &v);CHKERRQ(ierr);
  for (PetscInt i = 0; i < k; i++) {
    ierr = BVCopyVec(eps->V, i, v);
    if (!ierr) ierr = BVGetColumn(eps->V, i, &z);
    if (!ierr) {
      ierr = EPSApplyOperator(eps, v, z);
      { const PetscErrorCode ierr2 = BVRestoreColumn(eps->V, i, &z); if (!ierr) ierr = ierr2; }
    }
    if (ierr) break;
  }
  { const PetscErrorCode ierr2 = VecDestroy(&v); if (!ierr) ierr = ierr2; }
  CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode EPSComputeVectors(EPS eps) {
  PetscErrorCode ierr;
  PetscInt n;
  EPSCheckSolved(eps);
  if (eps->state == 3) return 0;
  eps->state = 3;
  if (eps->ishermitian && eps->type != 1) {
    /* EPSComputeVectors_Hermitian (epsdefault.c:40-74): the Lanczos vectors are the eigenvectors;
       with purification each one goes through the operator once more and is normalised again */
    if (eps->purify) {
      ierr = EPS_Purify(eps, eps->nconv);CHKERRQ(ierr);
      for (PetscInt i = 0; i < eps->nconv; i++) {
        PetscReal norm;
        ierr = BVNormColumn(eps->V, i, NORM_2, &norm);CHKERRQ(ierr);
        ierr = BVScaleColumn(eps->V, i, 1.0 / norm);CHKERRQ(ierr);
      }
    }
    return 0;
  }
  ierr = DSGetDimensions(eps->ds, &n, NULL, NULL, NULL, NULL);CHKERRQ(ierr);
  if (!n) return 0;
  ierr = DSVectors(eps->ds, DS_MAT_X, NULL, NULL);CHKERRQ(ierr);        /* right eigenvectors of T */
  ierr = BVSetActiveColumns(eps->V, 0, n);CHKERRQ(ierr);
  ierr = UpdateBasis(eps, DS_MAT_X, 0, n);CHKERRQ(ierr);                /* V = V * Z */
  if (eps->purify) {                                                    /* epsdefault.c:168-207 */
    Vec v, v1;
    PetscReal norm, normi;
    ierr = EPS_Purify(eps, eps->nconv);CHKERRQ(ierr);
    for (PetscInt i = 0; i < n; i++) {
      if (eps->eigi[i] != 0.0 && i + 1 < n) {
        ierr = BVGetColumn(eps->V, i, &v);CHKERRQ(ierr);
        ierr = BVGetColumn(eps->V, i + 1, &v1);CHKERRQ(ierr);
        ierr = VecNorm(v, NORM_2, &norm);CHKERRQ(ierr);
        ierr = VecNorm(v1, NORM_2, &normi);CHKERRQ(ierr);
        const PetscScalar tmp = 1.0 / SlepcAbsEigenvalue(norm, normi);
        ierr = VecScale(v, tmp);CHKERRQ(ierr);
        ierr = VecScale(v1, tmp);CHKERRQ(ierr);
        ierr = BVRestoreColumn(eps->V, i, &v);CHKERRQ(ierr);
        ierr = BVRestoreColumn(eps->V, i + 1, &v1);CHKERRQ(ierr);
        i++;
      } else {
        ierr = BVGetColumn(eps->V, i, &v);CHKERRQ(ierr);
        ierr = VecNorm(v, NORM_2, &norm);CHKERRQ(ierr);
        if (norm != 0.0) { ierr = VecScale(v, 1.0 / norm);CHKERRQ(ierr); }
        ierr = BVRestoreColumn(eps->V, i, &v);CHKERRQ(ierr);
      }
    }
  }
  return 0;
}

/* ---- epssolve.c:63-146 ---------------------------------------------------------------------------------------- */
PetscErrorCode EPSSolve(EPS eps) {
  PetscErrorCode ierr;
  PetscInt i;
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (eps->state >= 2) return 0;
  ierr = EPSSetUp(eps);CHKERRQ(ierr);
  eps->nconv = 0;
  eps->its = 0;
  eps->reason = EPS_CONVERGED_ITERATING;
  for (i = 0; i < eps->ncv; i++) { eps->eigr[i] = 0.0; eps->eigi[i] = 0.0; eps->errest[i] = 0.0; eps->perm[i] = i; }
  if (eps->type == 1) { ierr = EPSSolve_Arnoldi(eps);CHKERRQ(ierr); }
  else if (eps->type == 2) { ierr = EPSSolve_Lanczos(eps);CHKERRQ(ierr); }
  else if (eps->ishermitian) { ierr = EPSSolve_KrylovSchur_Symm(eps);CHKERRQ(ierr); }
  else { ierr = EPSSolve_KrylovSchur_Default(eps);CHKERRQ(ierr); }
  eps->state = 2;
  if (!eps->reason) SETERRQ(PETSC_ERR_PLIB, "Internal error, solver returned without setting converged reason");
  /* map the eigenvalues back to the original problem (EPSBackTransform_Default, epsdefault.c:27-34) */
  ierr = STBackTransform(eps->st, eps->nconv, eps->eigr, eps->eigi);CHKERRQ(ierr);
  /* reorder conjugate eigenvalues (positive imaginary first) */
  for (i = 0; i < eps->nconv - 1; i++) {
    if (eps->eigi[i] != 0) {
      if (eps->eigi[i] < 0) {
        eps->eigi[i] = -eps->eigi[i];
        eps->eigi[i + 1] = -eps->eigi[i + 1];
        /* the next correction only works with eigenvectors */
        ierr = EPSComputeVectors(eps);CHKERRQ(ierr);
        ierr = BVScaleColumn(eps->V, i + 1, -1.0);CHKERRQ(ierr);
      }
      i++;
    }
  }
  /* sort eigenvalues according to eps->which */
  for (i = 0; i < eps->ncv; i++) eps->perm[i] = i;
  ierr = SlepcSortEigenvalues(eps->cmp0, eps->cmp0ctx, eps->nconv, eps->eigr, eps->eigi, eps->perm);CHKERRQ(ierr);
  /* remove deflation and initial subspaces (epssolve.c:139-144) */
  if (eps->nds > 0) { ierr = BVSetNumConstraints(eps->V, 0);CHKERRQ(ierr); eps->nds = 0; }
  eps->nini = 0;
  return 0;
}

/* ---- results (epssolve.c:148-462,580-649) ------------------------------------------------------------------------- */
PetscErrorCode EPSGetIterationNumber(EPS eps, PetscInt *its) { if (!eps || !its) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *its = eps->its; return 0; }
PetscErrorCode EPSGetConverged(EPS eps, PetscInt *nconv) { if (!eps || !nconv) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); EPSCheckSolved(eps); *nconv = eps->nconv; return 0; }
PetscErrorCode EPSGetConvergedReason(EPS eps, EPSConvergedReason *reason) { if (!eps || !reason) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); EPSCheckSolved(eps); *reason = eps->reason; return 0; }
PetscErrorCode EPSGetDimensions(EPS eps, PetscInt *nev, PetscInt *ncv, PetscInt *mpd) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (nev) *nev = eps->nev;
  if (ncv) *ncv = eps->ncv;
  if (mpd) *mpd = eps->mpd;
  return 0;
}
PetscErrorCode EPSGetTolerances(EPS eps, PetscReal *tol, PetscInt *maxits) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (tol) *tol = eps->tol;
  if (maxits) *maxits = eps->max_it;
  return 0;
}

PetscErrorCode EPSGetEigenvalue(EPS eps, PetscInt i, PetscScalar *eigr, PetscScalar *eigi) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  EPSCheckSolved(eps);
  if (i < 0 || i >= eps->nconv) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument 2 out of range");
  const PetscInt k = eps->perm[i];
  if (eigr) *eigr = eps->eigr[k];
  if (eigi) *eigi = eps->eigi[k];
  return 0;
}

PetscErrorCode EPSGetEigenvector(EPS eps, PetscInt i, Vec Vr, Vec Vi) {
  PetscErrorCode ierr;
  if (!eps || !Vr) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  EPSCheckSolved(eps);
  if (i < 0 || i >= eps->nconv) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument 2 out of range");
  ierr = EPSComputeVectors(eps);CHKERRQ(ierr);
  const PetscInt k = eps->perm[i];
  if (eps->eigi[k] > 0) {            /* first value of conjugate pair */
    ierr = BVCopyVec(eps->V, k, Vr);CHKERRQ(ierr);
    if (Vi) { ierr = BVCopyVec(eps->V, k + 1, Vi);CHKERRQ(ierr); }
  } else if (eps->eigi[k] < 0) {     /* second value of conjugate pair */
    ierr = BVCopyVec(eps->V, k - 1, Vr);CHKERRQ(ierr);
    if (Vi) { ierr = BVCopyVec(eps->V, k, Vi);CHKERRQ(ierr); ierr = VecScale(Vi, -1.0);CHKERRQ(ierr); }
  } else {                           /* real eigenvalue */
    ierr = BVCopyVec(eps->V, k, Vr);CHKERRQ(ierr);
    if (Vi) { ierr = VecSet(Vi, 0.0);CHKERRQ(ierr); }
  }
  return 0;
}

PetscErrorCode EPSGetEigenpair(EPS eps, PetscInt i, PetscScalar *eigr, PetscScalar *eigi, Vec Vr, Vec Vi) {
  PetscErrorCode ierr;
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  EPSCheckSolved(eps);
  if (i < 0 || i >= eps->nconv) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument 2 out of range");
  ierr = EPSGetEigenvalue(eps, i, eigr, eigi);CHKERRQ(ierr);
  if (Vr) { ierr = EPSGetEigenvector(eps, i, Vr, Vi);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode EPSGetErrorEstimate(EPS eps, PetscInt i, PetscReal *errest) {
  if (!eps || !errest) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  EPSCheckSolved(eps);
  if (i < 0 || i >= eps->nconv) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument 2 out of range");
  *errest = eps->errest[eps->perm[i]];
  return 0;
}

/* epssolve.c:506-556 */
static PetscErrorCode EPSComputeResidualNorm_Private(EPS eps, PetscScalar kr, PetscScalar ki, Vec xr, Vec xi, Vec *z, PetscReal *norm) {
  PetscErrorCode ierr;
  Vec u = z[0], v = z[1], w = z[2];
  PetscReal ni, nr;
  if (eps->A->ops.residualnorm && !eps->isgeneralized) {
    /* the operator forms the same residuals in one pass each, storing nothing */
    if (ki == 0 || fabs(ki) < fabs(kr * PETSC_MACHINE_EPSILON)) {
      ierr = eps->A->ops.residualnorm(eps->A, fabs(kr) > PETSC_MACHINE_EPSILON ? kr : 0.0, 0.0, xr, NULL, u, norm);CHKERRQ(ierr);
    } else {
      const PetscBool sh = SlepcAbsEigenvalue(kr, ki) > PETSC_MACHINE_EPSILON;
      ierr = eps->A->ops.residualnorm(eps->A, sh ? kr : 0.0, sh ? ki : 0.0, xr, xi, u, &nr);CHKERRQ(ierr);
      ierr = eps->A->ops.residualnorm(eps->A, sh ? kr : 0.0, sh ? -ki : 0.0, xi, xr, u, &ni);CHKERRQ(ierr);
      *norm = SlepcAbsEigenvalue(nr, ni);
    }
    return 0;
  }
  if (ki == 0 || fabs(ki) < fabs(kr * PETSC_MACHINE_EPSILON)) {
    ierr = MatMult(eps->A, xr, u);CHKERRQ(ierr);                       /* u=A*x */
    if (fabs(kr) > PETSC_MACHINE_EPSILON) {
      if (eps->isgeneralized) { ierr = MatMult(eps->B, xr, w);CHKERRQ(ierr); }
      else { ierr = VecCopy(xr, w);CHKERRQ(ierr); }                    /* w=B*x */
      ierr = VecAXPY(u, -kr, w);CHKERRQ(ierr);                         /* u=A*x-k*B*x */
    }
    ierr = VecNorm(u, NORM_2, norm);CHKERRQ(ierr);
  } else {
    ierr = MatMult(eps->A, xr, u);CHKERRQ(ierr);                       /* u=A*xr */
    if (SlepcAbsEigenvalue(kr, ki) > PETSC_MACHINE_EPSILON) {
      if (eps->isgeneralized) { ierr = MatMult(eps->B, xr, v);CHKERRQ(ierr); }
      else { ierr = VecCopy(xr, v);CHKERRQ(ierr); }                    /* v=B*xr */
      ierr = VecAXPY(u, -kr, v);CHKERRQ(ierr);                         /* u=A*xr-kr*B*xr */
      if (eps->isgeneralized) { ierr = MatMult(eps->B, xi, w);CHKERRQ(ierr); }
      else { ierr = VecCopy(xi, w);CHKERRQ(ierr); }                    /* w=B*xi */
      ierr = VecAXPY(u, ki, w);CHKERRQ(ierr);                          /* u=A*xr-kr*B*xr+ki*B*xi */
    }
    ierr = VecNorm(u, NORM_2, &nr);CHKERRQ(ierr);
    ierr = MatMult(eps->A, xi, u);CHKERRQ(ierr);                       /* u=A*xi */
    if (SlepcAbsEigenvalue(kr, ki) > PETSC_MACHINE_EPSILON) {
      ierr = VecAXPY(u, -kr, w);CHKERRQ(ierr);                         /* u=A*xi-kr*B*xi */
      ierr = VecAXPY(u, -ki, v);CHKERRQ(ierr);                         /* u=A*xi-kr*B*xi-ki*B*xr */
    }
    ierr = VecNorm(u, NORM_2, &ni);CHKERRQ(ierr);
    *norm = SlepcAbsEigenvalue(nr, ni);
  }
  return 0;
}

/* epssolve.c:580-649 */
PetscErrorCode EPSComputeError(EPS eps, PetscInt i, EPSErrorType type, PetscReal *error) {
  PetscErrorCode ierr;
  PetscScalar kr, ki;
  Vec t, w[3];
  if (!eps || !error) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  EPSCheckSolved(eps);
  if (eps->nwork < 5) {          /* EPSSetWorkVecs(eps,5) */
    ierr = BVGetColumn(eps->V, 0, &t);CHKERRQ(ierr);
    for (PetscInt j = eps->nwork; j < 5; j++) { ierr = VecDuplicate(t, &eps->work[j]);CHKERRQ(ierr); }
    ierr = BVRestoreColumn(eps->V, 0, &t);CHKERRQ(ierr);
    eps->nwork = 5;
  }
  Vec xr = eps->work[0], xi = eps->work[3];
  w[0] = eps->work[1]; w[1] = eps->work[4]; w[2] = eps->work[2];
  ierr = EPSGetEigenpair(eps, i, &kr, &ki, xr, xi);CHKERRQ(ierr);
  ierr = EPSComputeResidualNorm_Private(eps, kr, ki, xr, xi, w, error);CHKERRQ(ierr);
  switch (type) {
    case EPS_ERROR_ABSOLUTE: break;
    case EPS_ERROR_RELATIVE: *error /= SlepcAbsEigenvalue(kr, ki); break;
    default: SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid error type");
  }
  return 0;
}

PetscErrorCode EPSGetCounters(EPS eps, PetscInt64 *matvecs, PetscInt64 *gs_passes) {
  if (!eps) SETERRQ(PETSC_ERR_ARG_NULL, "Null EPS");
  if (matvecs) *matvecs = eps->matvecs;
  if (gs_passes) *gs_passes = ks_gs_passes;
  return 0;
}

PetscErrorCode EPSDestroy(EPS *peps) {
  PetscErrorCode ierr;
  if (!peps || !*peps) return 0;
  EPS eps = *peps;
  for (PetscInt j = 0; j < eps->nwork; j++) { ierr = VecDestroy(&eps->work[j]);CHKERRQ(ierr); }
  for (PetscInt j = 0; j < eps->nini && eps->IS; j++) { ierr = VecDestroy(&eps->IS[j]);CHKERRQ(ierr); }
  free(eps->IS);
  if (eps->defl) for (PetscInt j = 0; j < -eps->nds; j++) { ierr = VecDestroy(&eps->defl[j]);CHKERRQ(ierr); }
  free(eps->defl);
  ierr = BVDestroy(&eps->V);CHKERRQ(ierr);
  ierr = BVDestroy(&eps->AV);CHKERRQ(ierr);
  ierr = STDestroy(&eps->st);CHKERRQ(ierr);
  ierr = DSDestroy(&eps->ds);CHKERRQ(ierr);
  free(eps->eigr); free(eps->eigi); free(eps->errest); free(eps->perm);
  free(eps);
  *peps = NULL;
  return 0;
}
