/* ST: the spectral transformation between the eigensolver and the operators, types "shift"
 * (src/sys/classes/st/impls/shift/shift.c) and "sinvert" (impls/sinvert/sinvert.c), with the
 * linear solves the reference delegates to PETSc's KSP (src/sys/classes/st/interface/stsles.c).
 *
 *   shift   : standard  y = (A - sigma I) x          generalized  y = B^-1 (A - sigma B) x
 *   sinvert : standard  y = (A - sigma I)^-1 x       generalized  y = (A - sigma B)^-1 B x
 *
 * A - sigma B is formed once as an operator of its own (STMatMAXPY_Private, stsolve.c:314-394):
 * a merged CSR matrix when both are device CSR operators, a shell otherwise.  The systems are
 * solved by
 *   "preonly" : LU of the matrix, dense, on the host -- the role of the reference's default
 *               (KSPPREONLY + PCLU, a host factorisation there too); one process, small N;
 *   "cg"      : conjugate gradients on the device vectors (symmetric positive definite);
 *   "gmres"   : restarted GMRES with modified Gram-Schmidt on the device vectors.
 * Every vector operation of the iterative solvers is a device kernel through the Vec/Mat op
 * tables; only the recurrence coefficients live on the host. */
#include "dense.h"
#include "ksimpl.h"

#define ST_LU_MAX 4096

struct _p_ST {
  void *ctx;
  int type;                        /* 0 shift, 1 sinvert */
  Mat A[2]; PetscInt nmat;
  PetscScalar sigma, defsigma;
  PetscBool sigma_set;
  Mat T[2]; PetscBool ownT[2];     /* shift: T[0] = A - sigma B, T[1] = B; sinvert: T[0] = B, T[1] = A - sigma B */
  Mat P;                           /* matrix of the linear systems (one of T), NULL if none */
  Vec w;
  int state;                       /* 0 initial, 1 set up */
  /* linear solver */
  int ksp_type;                    /* 0 chosen at setup, 1 preonly, 2 cg, 3 gmres */
  int ksp_used;
  PetscReal ksp_rtol; PetscInt ksp_maxit, ksp_restart;
  double *lu; int *piv; PetscInt lun; double *hb;
  Vec *kv; PetscInt nkv;
  PetscInt64 linits, nsolves;
};

/* ---- A + alpha B as an operator ------------------------------------------------------------- */
typedef struct { Mat A, B; PetscScalar alpha; Vec w; } AXPYShell;

static PetscErrorCode MatMult_AXPYShell(Mat M, Vec x, Vec y) {
  PetscErrorCode ierr;
  AXPYShell *s;
  ierr = MatShellGetContext(M, (void **)&s);CHKERRQ(ierr);
  ierr = MatMult(s->A, x, y);CHKERRQ(ierr);
  if (s->B) {
    if (!s->w) { ierr = VecDuplicate(y, &s->w);CHKERRQ(ierr); }
    ierr = MatMult(s->B, x, s->w);CHKERRQ(ierr);
    ierr = VecAXPY(y, s->alpha, s->w);CHKERRQ(ierr);
  } else { ierr = VecAXPY(y, s->alpha, x);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode MatCreateVecs_AXPYShell(Mat M, Vec *r, Vec *l) {
  PetscErrorCode ierr;
  AXPYShell *s;
  ierr = MatShellGetContext(M, (void **)&s);CHKERRQ(ierr);
  return MatCreateVecs(s->A, r, l);
}

/* T = A + alpha B (B == NULL: the identity).  Device CSR operators are merged entry by entry on
 * the host and uploaded as a new operator with the rows of A; anything else becomes a shell. */
static PetscErrorCode MatCreateAXPY(void *ctx, Mat A, PetscScalar alpha, Mat B, Mat *T, AXPYShell **shell) {
  PetscErrorCode ierr;
  ksg_mat ha = (ksg_mat)MatGetDeviceHandle_Private(A), hb = B ? (ksg_mat)MatGetDeviceHandle_Private(B) : NULL;
  *shell = NULL;
  if (ha && (!B || hb)) {
    int64_t nnza = 0, nnzb = 0;
    const int64_t n = A->n;
    CHKKSG(ksg_mat_sizes(ha, NULL, NULL, NULL, &nnza));
    if (hb) CHKKSG(ksg_mat_sizes(hb, NULL, NULL, NULL, &nnzb));
    else nnzb = n;
    int64_t *rpa = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t)), *rpb = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
    int32_t *cia = (int32_t *)malloc((size_t)PetscMax(nnza, 1) * sizeof(int32_t)), *cib = (int32_t *)malloc((size_t)PetscMax(nnzb, 1) * sizeof(int32_t));
    double *va = (double *)malloc((size_t)PetscMax(nnza, 1) * sizeof(double)), *vb = (double *)malloc((size_t)PetscMax(nnzb, 1) * sizeof(double));
    int64_t *rp = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
    int32_t *ci = (int32_t *)malloc((size_t)PetscMax(nnza + nnzb, 1) * sizeof(int32_t));
    double *v = (double *)malloc((size_t)PetscMax(nnza + nnzb, 1) * sizeof(double));
    ierr = 0;
    if (!rpa || !rpb || !cia || !cib || !va || !vb || !rp || !ci || !v) ierr = KSError(__func__, __LINE__, PETSC_ERR_MEM, "Out of memory");
    if (!ierr && ksg_mat_get_csr(ha, rpa, cia, va)) ierr = KSError(__func__, __LINE__, PETSC_ERR_LIB, "libksgpu: %s", ksg_last_error());
    if (!ierr) {
      if (hb) { if (ksg_mat_get_csr(hb, rpb, cib, vb)) ierr = KSError(__func__, __LINE__, PETSC_ERR_LIB, "libksgpu: %s", ksg_last_error()); }
      else for (int64_t i = 0; i <= n; i++) { rpb[i] = i; if (i < n) { cib[i] = (int32_t)(A->row0 + i); vb[i] = 1.0; } }
    }
    if (!ierr) {
      /* rows merged in ascending column order (both inputs are kept that way by the library) */
      int64_t q = 0;
      rp[0] = 0;
      for (int64_t i = 0; i < n && !ierr; i++) {
        int64_t a = rpa[i], ae = rpa[i + 1], b = rpb[i], be = rpb[i + 1];
        for (int64_t t = a + 1; t < ae; t++) if (cia[t] < cia[t - 1]) ierr = KSError(__func__, __LINE__, PETSC_ERR_SUP, "Rows must be sorted by column to form A - sigma B");
        for (int64_t t = b + 1; t < be; t++) if (cib[t] < cib[t - 1]) ierr = KSError(__func__, __LINE__, PETSC_ERR_SUP, "Rows must be sorted by column to form A - sigma B");
        while (a < ae || b < be) {
          if (b >= be || (a < ae && cia[a] < cib[b])) { ci[q] = cia[a]; v[q++] = va[a++]; }
          else if (a >= ae || cib[b] < cia[a]) { ci[q] = cib[b]; v[q++] = alpha * vb[b++]; }
          else { ci[q] = cia[a]; v[q++] = va[a++] + alpha * vb[b++]; }
        }
        rp[i + 1] = q;
      }
    }
    if (!ierr) ierr = MatCreateAIJHIPRect((ksg_ctx)A->ctx, A->n, A->N, A->row0, A->cn, A->cN, A->col0, rp, ci, v, T);
    free(rpa); free(rpb); free(cia); free(cib); free(va); free(vb); free(rp); free(ci); free(v);
    CHKERRQ(ierr);
    return 0;
  }
  AXPYShell *s = (AXPYShell *)calloc(1, sizeof(*s));
  if (!s) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  s->A = A; s->B = B; s->alpha = alpha;
  ierr = MatCreateShell(ctx, A->n, A->N, A->row0, s, MatMult_AXPYShell, MatCreateVecs_AXPYShell, T);
  if (ierr) { free(s); CHKERRQ(ierr); }
  *shell = s;
  return 0;
}

/* ---- object ------------------------------------------------------------------------------------ */
PetscErrorCode STCreate(void *ctx, ST *newst) {
  if (!newst) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ST st = (ST)calloc(1, sizeof(*st));
  if (!st) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  st->ctx = ctx;
  st->ksp_rtol = 1e-12;          /* tighter than the eigensolver's tolerance, as an exact solve would be */
  st->ksp_maxit = 10000;
  st->ksp_restart = 30;
  *newst = st;
  return 0;
}

static PetscErrorCode STReset(ST st) {
  PetscErrorCode ierr;
  for (int k = 0; k < 2; k++) {
    if (st->ownT[k] && st->T[k]) {
      AXPYShell *s = NULL;
      if (!MatGetDeviceHandle_Private(st->T[k])) { ierr = MatShellGetContext(st->T[k], (void **)&s);CHKERRQ(ierr); }
      ierr = MatDestroy(&st->T[k]);CHKERRQ(ierr);
      if (s) { ierr = VecDestroy(&s->w);CHKERRQ(ierr); free(s); }
    }
    st->T[k] = NULL; st->ownT[k] = PETSC_FALSE;
  }
  st->P = NULL;
  ierr = VecDestroy(&st->w);CHKERRQ(ierr);
  free(st->lu); st->lu = NULL; free(st->piv); st->piv = NULL; free(st->hb); st->hb = NULL; st->lun = 0;
  for (PetscInt i = 0; i < st->nkv; i++) { ierr = VecDestroy(&st->kv[i]);CHKERRQ(ierr); }
  free(st->kv); st->kv = NULL; st->nkv = 0;
  st->state = 0;
  return 0;
}

PetscErrorCode STDestroy(ST *st) {
  PetscErrorCode ierr;
  if (!st || !*st) return 0;
  ierr = STReset(*st);CHKERRQ(ierr);
  free(*st);
  *st = NULL;
  return 0;
}

PetscErrorCode STSetType(ST st, const char *type) {
  PetscErrorCode ierr;
  if (!st || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  int t;
  if (!strcmp(type, "shift")) t = 0;
  else if (!strcmp(type, "sinvert")) t = 1;
  else SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown ST type given: %s", type);
  if (t != st->type) { ierr = STReset(st);CHKERRQ(ierr); st->type = t; }
  return 0;
}
PetscErrorCode STGetType(ST st, const char **type) { if (!st || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *type = st->type ? "sinvert" : "shift"; return 0; }

/* stfunc.c (STSetOperators): A, and B for a generalized problem (NULL otherwise) */
PetscErrorCode STSetOperators(ST st, Mat A, Mat B) {
  PetscErrorCode ierr;
  if (!st || !A) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (A->n != A->cn || A->N != A->cN) SETERRQ(PETSC_ERR_ARG_WRONG, "A is a non-square matrix");
  if (B && (B->n != A->n || B->N != A->N || B->cn != A->cn)) SETERRQ(PETSC_ERR_ARG_INCOMP, "A and B must have the same sizes");
  if (st->A[0] != A || st->A[1] != B) { ierr = STReset(st);CHKERRQ(ierr); }
  st->A[0] = A; st->A[1] = B; st->nmat = B ? 2 : 1;
  return 0;
}
PetscErrorCode STGetNumMatrices(ST st, PetscInt *n) { if (!st || !n) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *n = st->nmat; return 0; }

PetscErrorCode STSetShift(ST st, PetscScalar shift) {
  PetscErrorCode ierr;
  if (!st) SETERRQ(PETSC_ERR_ARG_NULL, "Null ST");
  if (st->state && st->sigma != shift) { ierr = STReset(st);CHKERRQ(ierr); }      /* rebuilt at the next setup */
  st->sigma = shift;
  st->sigma_set = PETSC_TRUE;
  return 0;
}
PetscErrorCode STGetShift(ST st, PetscScalar *shift) { if (!st || !shift) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *shift = st->sigma; return 0; }
/* the value sinvert takes when no shift was given: the target of the eigensolver (stfunc.c) */
PetscErrorCode STSetDefaultShift(ST st, PetscScalar defaultshift) {
  PetscErrorCode ierr;
  if (!st) SETERRQ(PETSC_ERR_ARG_NULL, "Null ST");
  if (st->state && st->type == 1 && !st->sigma_set && st->defsigma != defaultshift) { ierr = STReset(st);CHKERRQ(ierr); }   /* the operator was built for the old value */
  st->defsigma = defaultshift;
  return 0;
}

PetscErrorCode STSetKSPType(ST st, const char *type) {
  PetscErrorCode ierr;
  if (!st || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  int t;
  if (!strcmp(type, "preonly")) t = 1;
  else if (!strcmp(type, "cg")) t = 2;
  else if (!strcmp(type, "gmres")) t = 3;
  else SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown linear solver: %s", type);
  if (t != st->ksp_type) { ierr = STReset(st);CHKERRQ(ierr); st->ksp_type = t; }
  return 0;
}
PetscErrorCode STGetKSPType(ST st, const char **type) {
  static const char *names[] = { "none", "preonly", "cg", "gmres" };
  if (!st || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  *type = names[st->state ? st->ksp_used : st->ksp_type];
  return 0;
}
PetscErrorCode STSetKSPTolerances(ST st, PetscReal rtol, PetscInt maxit, PetscInt restart) {
  if (!st) SETERRQ(PETSC_ERR_ARG_NULL, "Null ST");
  if (rtol < 0.0 || maxit < 0 || restart < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Negative value");
  if (rtol) st->ksp_rtol = rtol;
  if (maxit) st->ksp_maxit = maxit;
  if (restart) st->ksp_restart = restart;
  return 0;
}
/* linear iterations so far (KSPGetTotalIterations; one per solve for "preonly") and solves */
PetscErrorCode STGetKSPIterations(ST st, PetscInt64 *its, PetscInt64 *solves) {
  if (!st) SETERRQ(PETSC_ERR_ARG_NULL, "Null ST");
  if (its) *its = st->linits;
  if (solves) *solves = st->nsolves;
  return 0;
}

/* stfunc.c:140-164: the matrix of the inner product that makes the operator self-adjoint */
PetscErrorCode STGetBilinearForm(ST st, Mat *B) { if (!st || !B) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *B = st->nmat > 1 ? st->A[1] : NULL; return 0; }

/* ---- setup (shift.c:84-140, sinvert.c:101-160) -------------------------------------------------- */
PetscErrorCode STSetUp(ST st) {
  PetscErrorCode ierr;
  AXPYShell *sh;
  if (!st) SETERRQ(PETSC_ERR_ARG_NULL, "Null ST");
  if (st->state) return 0;
  if (!st->A[0]) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "STSetOperators must be called first");
  if (st->type == 1 && !st->sigma_set) st->sigma = st->defsigma;      /* sinvert.c:112 */
  const int kt = st->type == 0 ? 0 : 1;                                /* where A - sigma B goes */
  const int kb = 1 - kt;
  st->T[kb] = st->A[1]; st->ownT[kb] = PETSC_FALSE;
  if (st->sigma == 0.0) { st->T[kt] = st->A[0]; st->ownT[kt] = PETSC_FALSE; }
  else {
    ierr = MatCreateAXPY(st->ctx, st->A[0], -st->sigma, st->A[1], &st->T[kt], &sh);CHKERRQ(ierr);
    st->ownT[kt] = PETSC_TRUE;
  }
  st->P = st->T[1];                   /* shift: B (none for a standard problem); sinvert: A - sigma B */
  if (st->nmat > 1) { ierr = MatCreateVecs(st->A[0], &st->w, NULL);CHKERRQ(ierr); }
  st->ksp_used = 0;
  if (st->P) {
    ksg_mat hp = (ksg_mat)MatGetDeviceHandle_Private(st->P);
    const PetscBool lu_ok = hp && st->P->n == st->P->N && st->P->N <= ST_LU_MAX;
    st->ksp_used = st->ksp_type ? st->ksp_type : (lu_ok ? 1 : 3);
    if (st->ksp_used == 1) {
      if (!lu_ok) SETERRQ(PETSC_ERR_SUP, "The direct solver needs a device CSR matrix on one process with at most %d rows; choose cg or gmres", ST_LU_MAX);
      const PetscInt n = (PetscInt)st->P->N;
      int64_t nnz = 0;
      CHKKSG(ksg_mat_sizes(hp, NULL, NULL, NULL, &nnz));
      int64_t *rp = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
      int32_t *ci = (int32_t *)malloc((size_t)PetscMax(nnz, 1) * sizeof(int32_t));
      double *v = (double *)malloc((size_t)PetscMax(nnz, 1) * sizeof(double));
      st->lu = (double *)calloc((size_t)n * n, sizeof(double));
      st->piv = (int *)malloc((size_t)n * sizeof(int));
      st->hb = (double *)malloc((size_t)n * sizeof(double));
      ierr = 0;
      if (!rp || !ci || !v || !st->lu || !st->piv || !st->hb) ierr = KSError(__func__, __LINE__, PETSC_ERR_MEM, "Out of memory");
      if (!ierr && ksg_mat_get_csr(hp, rp, ci, v)) ierr = KSError(__func__, __LINE__, PETSC_ERR_LIB, "libksgpu: %s", ksg_last_error());
      if (!ierr) {
        for (PetscInt i = 0; i < n; i++) for (int64_t p = rp[i]; p < rp[i + 1]; p++) st->lu[i + (size_t)ci[p] * n] += v[p];
        const int info = ksd_getrf(n, st->lu, n, st->piv);
        if (info) ierr = KSError(__func__, __LINE__, PETSC_ERR_MAT_LU_ZRPVT, "Zero pivot in LU factorization, row %d: the matrix of the linear systems is singular", info - 1);
      }
      free(rp); free(ci); free(v);
      CHKERRQ(ierr);
      st->lun = n;
    } else {
      st->nkv = st->ksp_used == 2 ? 3 : st->ksp_restart + 2;
      st->kv = (Vec *)calloc((size_t)st->nkv, sizeof(Vec));
      if (!st->kv) SETERRQ(PETSC_ERR_MEM, "Out of memory");
      for (PetscInt i = 0; i < st->nkv; i++) { ierr = MatCreateVecs(st->A[0], &st->kv[i], NULL);CHKERRQ(ierr); }
    }
  }
  st->state = 1;
  return 0;
}

/* ---- linear solves (stsles.c: STMatSolve -> KSPSolve) ------------------------------------------ */
static PetscErrorCode KSPSolve_CG(ST st, Vec b, Vec x) {
  PetscErrorCode ierr;
  Vec r = st->kv[0], p = st->kv[1], q = st->kv[2];
  PetscScalar rr, rrnew, pq;
  PetscReal bnorm;
  ierr = VecNorm(b, NORM_2, &bnorm);CHKERRQ(ierr);
  ierr = VecSet(x, 0.0);CHKERRQ(ierr);
  if (bnorm == 0.0) return 0;
  ierr = VecCopy(b, r);CHKERRQ(ierr);
  ierr = VecCopy(b, p);CHKERRQ(ierr);
  rr = bnorm * bnorm;
  for (PetscInt it = 0; it < st->ksp_maxit; it++) {
    ierr = MatMult(st->P, p, q);CHKERRQ(ierr);
    ierr = VecDot(p, q, &pq);CHKERRQ(ierr);
    if (pq <= 0.0) SETERRQ(PETSC_ERR_NOT_CONVERGED, "Conjugate gradients: the matrix is not positive definite; choose gmres");
    const PetscScalar a = rr / pq;
    ierr = VecAXPY(x, a, p);CHKERRQ(ierr);
    ierr = VecAXPY(r, -a, q);CHKERRQ(ierr);
    ierr = VecDot(r, r, &rrnew);CHKERRQ(ierr);
    st->linits++;
    if (sqrt(rrnew) <= st->ksp_rtol * bnorm) return 0;
    ierr = VecScale(p, rrnew / rr);CHKERRQ(ierr);          /* p = r + beta p */
    ierr = VecAXPY(p, 1.0, r);CHKERRQ(ierr);
    rr = rrnew;
  }
  SETERRQ(PETSC_ERR_NOT_CONVERGED, "Conjugate gradients did not reach the tolerance in %d iterations", st->ksp_maxit);
}

static PetscErrorCode KSPSolve_GMRES(ST st, Vec b, Vec x) {
  PetscErrorCode ierr;
  const PetscInt m = st->ksp_restart;
  Vec *V = st->kv, w = st->kv[m + 1];
  PetscReal bnorm, beta;
  PetscScalar *H = (PetscScalar *)calloc((size_t)(m + 1) * m, sizeof(*H)), *cs = (PetscScalar *)calloc((size_t)3 * (m + 1), sizeof(*cs));
  if (!H || !cs) { free(H); free(cs); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  PetscScalar *sn = cs + m + 1, *g = sn + m + 1;
#define GCHK(e) do { ierr = (e); if (ierr) { free(H); free(cs); CHKERRQ(ierr); } } while (0)
  GCHK(VecNorm(b, NORM_2, &bnorm));
  GCHK(VecSet(x, 0.0));
  if (bnorm == 0.0) { free(H); free(cs); return 0; }
  PetscInt total = 0;
  PetscBool first = PETSC_TRUE, done = PETSC_FALSE;
  while (!done) {
    /* residual of the current iterate */
    if (first) { GCHK(VecCopy(b, V[0])); beta = bnorm; first = PETSC_FALSE; }
    else {
      GCHK(MatMult(st->P, x, w));
      GCHK(VecCopy(b, V[0]));
      GCHK(VecAXPY(V[0], -1.0, w));
      GCHK(VecNorm(V[0], NORM_2, &beta));
      if (beta <= st->ksp_rtol * bnorm) break;
    }
    GCHK(VecScale(V[0], 1.0 / beta));
    memset(g, 0, (size_t)(m + 1) * sizeof(*g));
    g[0] = beta;
    PetscInt j;
    for (j = 0; j < m && !done; j++) {
      PetscScalar *h = H + (size_t)j * (m + 1);
      PetscReal hn;
      GCHK(MatMult(st->P, V[j], V[j + 1]));
      for (PetscInt i = 0; i <= j; i++) {              /* modified Gram-Schmidt */
        GCHK(VecDot(V[j + 1], V[i], &h[i]));
        GCHK(VecAXPY(V[j + 1], -h[i], V[i]));
      }
      GCHK(VecNorm(V[j + 1], NORM_2, &hn));
      h[j + 1] = hn;
      if (hn != 0.0) GCHK(VecScale(V[j + 1], 1.0 / hn));
      for (PetscInt i = 0; i < j; i++) {               /* previous rotations */
        const PetscScalar t = cs[i] * h[i] + sn[i] * h[i + 1];
        h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
        h[i] = t;
      }
      const PetscReal rho = hypot(h[j], h[j + 1]);
      cs[j] = h[j] / rho; sn[j] = h[j + 1] / rho;
      h[j] = rho; h[j + 1] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      st->linits++; total++;
      if (fabs(g[j + 1]) <= st->ksp_rtol * bnorm || hn == 0.0) done = PETSC_TRUE;
      if (total >= st->ksp_maxit && !done) { free(H); free(cs); SETERRQ(PETSC_ERR_NOT_CONVERGED, "GMRES did not reach the tolerance in %d iterations", st->ksp_maxit); }
    }
    /* x += V y with the triangular system H y = g */
    for (PetscInt i = j - 1; i >= 0; i--) {
      PetscScalar s = g[i];
      for (PetscInt l = i + 1; l < j; l++) s -= H[i + (size_t)l * (m + 1)] * g[l];
      g[i] = s / H[i + (size_t)i * (m + 1)];
    }
    for (PetscInt i = 0; i < j; i++) GCHK(VecAXPY(x, g[i], V[i]));
  }
#undef GCHK
  free(H); free(cs);
  return 0;
}

PetscErrorCode STMatSolve(ST st, Vec b, Vec x) {
  PetscErrorCode ierr;
  if (!st || !b || !x) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!st->state || !st->P) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "The transformation has no linear systems to solve, or STSetUp was not called");
  if (b == x) SETERRQ(PETSC_ERR_ARG_WRONG, "b and x must be different vectors");
  st->nsolves++;
  switch (st->ksp_used) {
    case 1:
      ierr = VecGetValuesHost(b, st->hb);CHKERRQ(ierr);
      ksd_getrs(st->lun, st->lu, st->lun, st->piv, st->hb);
      ierr = VecSetValuesHost(x, st->hb);CHKERRQ(ierr);
      st->linits++;
      break;
    case 2: ierr = KSPSolve_CG(st, b, x);CHKERRQ(ierr); break;
    default: ierr = KSPSolve_GMRES(st, b, x);CHKERRQ(ierr); break;
  }
  return 0;
}

/* ---- the operator (stsolve.c:44-72; shift.c:27-41; sinvert.c:26-40) --------------------------- */
PetscErrorCode STApply(ST st, Vec x, Vec y) {
  PetscErrorCode ierr;
  if (!st || !x || !y) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (x == y) SETERRQ(PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  if (!st->state) { ierr = STSetUp(st);CHKERRQ(ierr); }
  if (st->nmat > 1) {
    ierr = MatMult(st->T[0], x, st->w);CHKERRQ(ierr);
    ierr = STMatSolve(st, st->w, y);CHKERRQ(ierr);
  } else if (st->type == 0) { ierr = MatMult(st->T[0], x, y);CHKERRQ(ierr); }
  else { ierr = STMatSolve(st, x, y);CHKERRQ(ierr); }
  return 0;
}

/* shift.c:60-70, sinvert.c:59-82 */
PetscErrorCode STBackTransform(ST st, PetscInt n, PetscScalar *eigr, PetscScalar *eigi) {
  if (!st || (n && (!eigr || !eigi))) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (st->type == 0) for (PetscInt j = 0; j < n; j++) eigr[j] += st->sigma;
  else
    for (PetscInt j = 0; j < n; j++) {
      if (eigi[j] == 0) eigr[j] = 1.0 / eigr[j] + st->sigma;
      else {
        const PetscScalar t = eigr[j] * eigr[j] + eigi[j] * eigi[j];
        eigr[j] = eigr[j] / t + st->sigma;
        eigi[j] = -eigi[j] / t;
      }
    }
  return 0;
}

PetscBool STIsTrivial_Private(ST st) { return (!st || (st->type == 0 && st->sigma == 0.0 && st->nmat < 2)) ? PETSC_TRUE : PETSC_FALSE; }
PetscBool STIsShift_Private(ST st) { return (!st || st->type == 0) ? PETSC_TRUE : PETSC_FALSE; }
