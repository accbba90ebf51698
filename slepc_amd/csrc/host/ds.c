/* DS: the dense projected eigenproblem of order <= ncv+1 that Krylov-Schur solves
 * on the host once per restart.  Two types, as the reference configures them in
 * EPSSetUp_KrylovSchur (krylovschur.c:135-148):
 *   "hep"  - symmetric, compact storage (arrow + tridiagonal in T), extra row
 *            (src/sys/classes/ds/impls/hep/dshep.c)
 *   "nhep" - non-symmetric, real Schur form (src/sys/classes/ds/impls/nhep/dsnhep.c)
 *   "svd"  - singular values of an arrow + upper bidiagonal matrix, compact storage, as the
 *            thick-restart Lanczos bidiagonalisation sets it up (trlanczos.c:71-73;
 *            src/sys/classes/ds/impls/svd/dssvd.c)
 * Interface functions follow src/sys/classes/ds/interface/dsops.c, dsbasic.c,
 * dspriv.c.  The LAPACK calls of the reference are replaced by dense.c. */
#include "dense.h"
#include "ksimpl.h"

#define DSCheckAlloc(ds) do { if (!(ds)) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS"); \
  if (!(ds)->ld) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Must call DSAllocate() first"); } while (0)
#define DSCheckSolved(ds) do { if ((ds)->state < DS_STATE_CONDENSED) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Must call DSSolve() first"); } while (0)

PetscErrorCode DSCreate(DS *ds) {
  if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  DS d = (DS)calloc(1, sizeof(*d));
  if (!d) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  d->type = -1;
  *ds = d;
  return 0;
}

PetscErrorCode DSSetType(DS ds, const char *type) {
  if (!ds || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!strcmp(type, "hep")) ds->type = 0;
  else if (!strcmp(type, "nhep")) ds->type = 1;
  else if (!strcmp(type, "svd")) ds->type = 2;
  else SETERRQ(PETSC_ERR_ARG_WRONG, "Unable to find requested DS type %s", type);
  return 0;
}
PetscErrorCode DSSetCompact(DS ds, PetscBool comp) { if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS"); ds->compact = comp; return 0; }
PetscErrorCode DSSetExtraRow(DS ds, PetscBool ext) {
  if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS");
  if (ds->n > 0 && ds->n == ds->ld) SETERRQ(PETSC_ERR_ORDER, "Cannot set extra row after setting n=ld");
  ds->extrarow = ext;
  return 0;
}

PetscErrorCode DSAllocate(DS ds, PetscInt ld) {
  if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS");
  if (ds->type < 0) SETERRQ(PETSC_ERR_ORDER, "Must call DSSetType() first");
  if (ld < 1) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Leading dimension should be at least one");
  for (int i = 0; i < DS_NUM_MAT; i++) { free(ds->mat[i]); ds->mat[i] = NULL; }
  free(ds->rmat); free(ds->perm);
  ds->ld = ld;
  ds->mat[DS_MAT_A] = (PetscScalar *)calloc((size_t)ld * ld, sizeof(PetscScalar));
  ds->mat[DS_MAT_Q] = (PetscScalar *)calloc((size_t)ld * ld, sizeof(PetscScalar));
  ds->mat[DS_MAT_X] = (PetscScalar *)calloc((size_t)ld * ld, sizeof(PetscScalar));
  if (ds->type == 2) {          /* DSAllocate_SVD dssvd.c:24-37 */
    ds->mat[DS_MAT_U] = (PetscScalar *)calloc((size_t)ld * ld, sizeof(PetscScalar));
    ds->mat[DS_MAT_VT] = (PetscScalar *)calloc((size_t)ld * ld, sizeof(PetscScalar));
    if (!ds->mat[DS_MAT_U] || !ds->mat[DS_MAT_VT]) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  }
  ds->rmat = (PetscReal *)calloc((size_t)3 * ld, sizeof(PetscReal));
  ds->perm = (PetscInt *)calloc((size_t)ld, sizeof(PetscInt));
  if (!ds->mat[DS_MAT_A] || !ds->mat[DS_MAT_Q] || !ds->mat[DS_MAT_X] || !ds->rmat || !ds->perm) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  return 0;
}

PetscErrorCode DSSetSlepcSC(DS ds, SlepcCompareFn cmp, void *ctx) { if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS"); ds->cmp = cmp; ds->cmpctx = ctx; return 0; }

/* dsops.c:139-177 */
PetscErrorCode DSSetDimensions(DS ds, PetscInt n, PetscInt m, PetscInt l, PetscInt k) {
  DSCheckAlloc(ds);
  if (n < 0 || n > ds->ld) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of n. Must be between 0 and ld");
  if (ds->extrarow && n + 1 > ds->ld) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "A value of n equal to ld leaves no room for extra row");
  ds->n = n;
  ds->t = ds->n;
  if (m) {
    if (m < 0 || m > ds->ld) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of m. Must be between 0 and ld");
    ds->m = m;
  }
  if (l < 0 || l > ds->n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of l. Must be between 0 and n");
  ds->l = l;
  if (k < 0 || k > ds->n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of k. Must be between 0 and n");
  ds->k = k;
  return 0;
}
PetscErrorCode DSGetDimensions(DS ds, PetscInt *n, PetscInt *m, PetscInt *l, PetscInt *k, PetscInt *t) {
  DSCheckAlloc(ds);
  if (n) *n = ds->n;
  if (m) *m = ds->m;
  if (l) *l = ds->l;
  if (k) *k = ds->k;
  if (t) *t = ds->t;
  return 0;
}
PetscErrorCode DSGetLeadingDimension(DS ds, PetscInt *ld) { if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS"); *ld = ds->ld; return 0; }
PetscErrorCode DSSetState(DS ds, DSStateType state) {
  if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS");
  if (state < DS_STATE_RAW || state > DS_STATE_TRUNCATED) SETERRQ(PETSC_ERR_ARG_WRONG, "Wrong state");
  ds->state = state;
  return 0;
}
PetscErrorCode DSGetState(DS ds, DSStateType *state) { if (!ds) SETERRQ(PETSC_ERR_ARG_NULL, "Null DS"); *state = ds->state; return 0; }
PetscErrorCode DSGetArray(DS ds, DSMatType m, PetscScalar **a) {
  DSCheckAlloc(ds);
  if (m >= DS_NUM_MAT || m == DS_MAT_T) SETERRQ(PETSC_ERR_ARG_WRONG, "Invalid matrix");
  *a = ds->mat[m];
  return 0;
}
PetscErrorCode DSGetArrayReal(DS ds, DSMatType m, PetscReal **a) {
  DSCheckAlloc(ds);
  if (m != DS_MAT_T) SETERRQ(PETSC_ERR_ARG_WRONG, "Invalid matrix");
  *a = ds->rmat;
  return 0;
}

/* ---- HEP (dshep.c) ---------------------------------------------------------------------- */
/* dshep.c:254-301: Rutishauser reduction of the arrowhead block to tridiagonal */
static void ArrowTridiag(PetscInt n, PetscReal *d, PetscReal *e, PetscScalar *Q, PetscInt ld) {
  PetscInt i, j;
  PetscReal c, s, p, off, temp;
  if (n <= 2) return;
  for (j = 0; j < n - 2; j++) {
    /* eliminate entry e(j) by a rotation in the planes (j,j+1) */
    temp = e[j + 1];
    ksd_lartg(temp, e[j], &c, &s, &e[j + 1]);
    s = -s;
    /* apply rotation to diagonal elements */
    temp = d[j + 1];
    e[j] = c * s * (temp - d[j]);
    d[j + 1] = s * s * d[j] + c * c * temp;
    d[j] = c * c * d[j] + s * s * temp;
    /* apply rotation to Q */
    ksd_rot(j + 2, Q + (size_t)j * ld, Q + (size_t)(j + 1) * ld, c, s);
    /* chase newly introduced off-diagonal entry to the top left corner */
    for (i = j - 1; i >= 0; i--) {
      off = -s * e[i];
      e[i] = c * e[i];
      temp = e[i + 1];
      ksd_lartg(temp, off, &c, &s, &e[i + 1]);
      s = -s;
      temp = (d[i] - d[i + 1]) * s - 2.0 * c * e[i];
      p = s * temp;
      d[i + 1] += p;
      d[i] -= p;
      e[i] = -e[i] - c * temp;
      ksd_rot(j + 2, Q + (size_t)i * ld, Q + (size_t)(i + 1) * ld, c, s);
    }
  }
}

/* dshep.c:64-100 */
static void DSSwitchFormat_HEP(DS ds, PetscBool tocompact) {
  PetscReal *T = ds->rmat;
  PetscScalar *A = ds->mat[DS_MAT_A];
  PetscInt i, n = ds->n, k = ds->k, ld = ds->ld;
  if (tocompact) {
    memset(T, 0, (size_t)3 * ld * sizeof(PetscReal));
    for (i = 0; i < k; i++) { T[i] = A[i + i * ld]; T[i + ld] = A[k + i * ld]; }
    for (i = k; i < n - 1; i++) { T[i] = A[i + i * ld]; T[i + ld] = A[i + 1 + i * ld]; }
    T[n - 1] = A[n - 1 + (n - 1) * ld];
    if (ds->extrarow) T[n - 1 + ld] = A[n + (n - 1) * ld];
  } else {
    memset(A, 0, (size_t)ld * ld * sizeof(PetscScalar));
    for (i = 0; i < k; i++) { A[i + i * ld] = T[i]; A[k + i * ld] = T[i + ld]; A[i + k * ld] = T[i + ld]; }
    A[k + k * ld] = T[k];
    for (i = k + 1; i < n; i++) { A[i + i * ld] = T[i]; A[i - 1 + i * ld] = T[i - 1 + ld]; A[i + (i - 1) * ld] = T[i - 1 + ld]; }
    if (ds->extrarow) A[n + (n - 1) * ld] = T[n - 1 + ld];
  }
}

/* dshep.c:420-471 with DSIntermediate_HEP (dshep.c:306-359).  The non-compact
 * storage is handled by switching to compact form and back, which requires the
 * dense matrix to have the arrow-plus-tridiagonal pattern of dshep.c:39-62. */
static PetscErrorCode DSSolve_HEP_QR(DS ds, PetscScalar *wr, PetscScalar *wi) {
  PetscInt i, n = ds->n, l = ds->l, ld = ds->ld;
  const PetscInt n1 = ds->k - l + 1, n2 = n - ds->k - 1, n3 = n1 + n2, off = l + l * ld;
  PetscScalar *Q = ds->mat[DS_MAT_Q], *A = ds->mat[DS_MAT_A];
  PetscReal *d = ds->rmat, *e = ds->rmat + ld;
  const PetscBool compact = ds->compact;
  if (!compact) DSSwitchFormat_HEP(ds, PETSC_TRUE);
  memset(Q, 0, (size_t)ld * ld * sizeof(PetscScalar));
  for (i = 0; i < n; i++) Q[i + i * ld] = 1.0;
  if (ds->state < DS_STATE_INTERMEDIATE) ArrowTridiag(n1, d + l, e + l, Q + off, ld);
  for (i = 0; i < l; i++) wr[i] = d[i];
  int info = ksd_steqr(n3, d + l, e + l, Q + off, ld, n3);
  if (info) SETERRQ(PETSC_ERR_LIB, "Error in tridiagonal QL iteration %d", info);
  for (i = l; i < n; i++) wr[i] = d[i];
  if (compact) memset(e, 0, (size_t)(n > 1 ? n - 1 : 0) * sizeof(PetscReal));
  else {
    for (i = l; i < n; i++) memset(A + l + i * ld, 0, (size_t)(n - l) * sizeof(PetscScalar));
    for (i = l; i < n; i++) A[i + i * ld] = d[i];
  }
  if (wi) for (i = l; i < n; i++) wi[i] = 0.0;
  return 0;
}

/* dspriv.c:288-309 */
static void DSPermuteColumns(DS ds, PetscInt l, PetscInt n, PetscScalar *Q, PetscInt *perm) {
  PetscInt i, j, k, p, ld = ds->ld;
  for (i = l; i < n; i++) {
    p = perm[i];
    if (p != i) {
      j = i + 1;
      while (perm[j] != i) j++;
      perm[j] = p; perm[i] = i;
      for (k = 0; k < n; k++) { PetscScalar t = Q[k + p * ld]; Q[k + p * ld] = Q[k + i * ld]; Q[k + i * ld] = t; }
    }
  }
}

/* dshep.c:361-387 with DSSortEigenvaluesReal_Private (dspriv.c:242-264) */
static PetscErrorCode DSSort_HEP(DS ds, PetscScalar *wr, PetscScalar *wi) {
  PetscErrorCode ierr;
  PetscInt n = ds->t, l = ds->l, i, j, result, tmp, *perm = ds->perm, ld = ds->ld;
  PetscReal *d = ds->rmat;
  PetscScalar *A = ds->mat[DS_MAT_A];
  for (i = l + 1; i < n; i++) {
    const PetscScalar re = d[perm[i]];
    j = i - 1;
    ierr = ds->cmp(re, 0.0, d[perm[j]], 0.0, &result, ds->cmpctx);CHKERRQ(ierr);
    while (result < 0 && j >= l) {
      tmp = perm[j]; perm[j] = perm[j + 1]; perm[j + 1] = tmp; j--;
      if (j >= l) { ierr = ds->cmp(re, 0.0, d[perm[j]], 0.0, &result, ds->cmpctx);CHKERRQ(ierr); }
    }
  }
  n = ds->n;
  for (i = l; i < n; i++) wr[i] = d[perm[i]];
  DSPermuteColumns(ds, l, n, ds->mat[DS_MAT_Q], perm);
  for (i = l; i < n; i++) d[i] = wr[i];
  if (!ds->compact) for (i = l; i < n; i++) A[i + i * ld] = wr[i];
  return 0;
}

/* dshep.c:389-418 */
static PetscErrorCode DSUpdateExtraRow_HEP(DS ds) {
  PetscInt i, j, n = ds->n, ld = ds->ld;
  PetscScalar *A = ds->mat[DS_MAT_A], *Q = ds->mat[DS_MAT_Q];
  PetscReal *e = ds->rmat + ld;
  if (ds->compact) {
    const PetscReal beta = e[n - 1];      /* in compact, all entries are zero except the last one */
    for (i = 0; i < n; i++) e[i] = beta * Q[n - 1 + i * ld];
  } else {
    PetscScalar *x = (PetscScalar *)malloc((size_t)2 * ld * sizeof(PetscScalar));
    if (!x) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    PetscScalar *y = x + ld;
    for (i = 0; i < n; i++) x[i] = A[n + i * ld];
    for (j = 0; j < n; j++) { PetscScalar s = 0.0; for (i = 0; i < n; i++) s += Q[i + j * ld] * x[i]; y[j] = s; }
    for (i = 0; i < n; i++) A[n + i * ld] = y[i];
    free(x);
  }
  ds->k = n;
  return 0;
}

/* dshep.c:170-205 */
static PetscErrorCode DSVectors_HEP(DS ds, DSMatType mat, PetscInt *j, PetscReal *rnorm) {
  PetscScalar *Q = ds->mat[DS_MAT_Q], *X = ds->mat[mat];
  PetscInt ld = ds->ld, i;
  if (mat != DS_MAT_X) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid mat parameter");
  if (j) {
    if (ds->state >= DS_STATE_CONDENSED) memcpy(X + (size_t)(*j) * ld, Q + (size_t)(*j) * ld, (size_t)ld * sizeof(PetscScalar));
    else { memset(X + (size_t)(*j) * ld, 0, (size_t)ld * sizeof(PetscScalar)); X[(*j) + (*j) * ld] = 1.0; }
  } else {
    if (ds->state >= DS_STATE_CONDENSED) memcpy(X, Q, (size_t)ld * ld * sizeof(PetscScalar));
    else { memset(X, 0, (size_t)ld * ld * sizeof(PetscScalar)); for (i = 0; i < ds->n; i++) X[i + i * ld] = 1.0; }
  }
  if (rnorm && j) *rnorm = fabs(Q[ds->n - 1 + (*j) * ld]);
  return 0;
}

/* dshep.c:689-704 */
static PetscErrorCode DSTruncate_HEP(DS ds, PetscInt n) {
  PetscInt i, ld = ds->ld, l = ds->l;
  PetscScalar *A = ds->mat[DS_MAT_A];
  if (ds->state == DS_STATE_CONDENSED) ds->t = ds->n;
  if (!ds->compact && ds->extrarow && ds->k == ds->n) for (i = l; i < n; i++) A[n + i * ld] = A[ds->n + i * ld];
  if (ds->extrarow) ds->k = n;
  else ds->k = 0;
  ds->n = n;
  return 0;
}

/* ---- NHEP (dsnhep.c) ------------------------------------------------------------------------ */
static void SchurEigenvalue(const PetscScalar *A, PetscInt ld, PetscInt n, PetscInt *j, PetscScalar *wr, PetscScalar *wi) {
  const PetscInt jj = *j;
  if (jj == n - 1 || A[jj + 1 + jj * ld] == 0.0) { wr[jj] = A[jj + jj * ld]; wi[jj] = 0.0; }
  else {
    wr[jj] = A[jj + jj * ld]; wr[jj + 1] = A[jj + jj * ld];
    wi[jj] = sqrt(fabs(A[jj + 1 + jj * ld])) * sqrt(fabs(A[jj + (jj + 1) * ld]));
    wi[jj + 1] = -wi[jj];
    (*j)++;
  }
}

/* dsnhep.c:469-543 */
static PetscErrorCode DSSolve_NHEP(DS ds, PetscScalar *wr, PetscScalar *wi) {
  PetscInt i, j, n = ds->n, ld = ds->ld;
  PetscScalar *A = ds->mat[DS_MAT_A], *Q = ds->mat[DS_MAT_Q];
  if (!wi) SETERRQ(PETSC_ERR_ARG_NULL, "Null pointer: wi");
  memset(Q, 0, (size_t)ld * ld * sizeof(PetscScalar));
  for (i = 0; i < n; i++) Q[i + i * ld] = 1.0;
  if (n == 1) { wr[0] = A[0]; wi[0] = 0.0; return 0; }
  if (ds->state < DS_STATE_INTERMEDIATE) {
    if (ksd_gehrd_orghr(n, ds->l, A, ld, Q, ld)) SETERRQ(PETSC_ERR_MEM, "Out of memory in Hessenberg reduction");
  }
  int info = ksd_hseqr(n, ds->l, A, ld, wr, wi, Q, ld);
  for (j = 0; j < ds->l; j++) SchurEigenvalue(A, ld, n, &j, wr, wi);
  if (info) SETERRQ(PETSC_ERR_LIB, "Error in Hessenberg QR iteration %d", info);
  return 0;
}

/* dsnhep.c:347-432: selection sort of the diagonal blocks with orthogonal swaps */
static PetscErrorCode DSSort_NHEP_Total(DS ds, PetscScalar *wr, PetscScalar *wi) {
  PetscErrorCode ierr;
  PetscScalar re, im;
  PetscInt i, j, pos, result, n = ds->n, ld = ds->ld;
  PetscScalar *T = ds->mat[DS_MAT_A], *Q = ds->mat[DS_MAT_Q];
  for (i = ds->l; i < n - 1; i++) {
    re = wr[i]; im = wi[i];
    pos = 0;
    j = i + 1;
    if (im != 0) j = i + 2;
    for (; j < n; j++) {
      ierr = ds->cmp(re, im, wr[j], wi[j], &result, ds->cmpctx);CHKERRQ(ierr);
      if (result > 0) { re = wr[j]; im = wi[j]; pos = j; }
      if (wi[j] != 0) j++;
    }
    if (pos) {
      if (ksd_trexc(n, T, ld, Q, ld, n, pos, i)) SETERRQ(PETSC_ERR_LIB, "Error swapping diagonal blocks of the Schur form");
      for (j = i; j < n; j++) SchurEigenvalue(T, ld, n, &j, wr, wi);
    }
    if (wi[i] != 0) i++;
  }
  return 0;
}

/* dsnhep.c:123-192 */
static PetscErrorCode DSVectors_NHEP_Eigen_Some(DS ds, PetscInt *k, PetscReal *rnorm) {
  PetscInt i, j, n = ds->n, ld = ds->ld;
  PetscScalar *A = ds->mat[DS_MAT_A], *Q = ds->mat[DS_MAT_Q], *X = ds->mat[DS_MAT_X];
  PetscScalar *Y = X + (size_t)(*k) * ld;
  PetscBool iscomplex = PETSC_FALSE;
  PetscReal norm, tmp;
  if ((*k) < n - 1 && A[(*k) + 1 + (*k) * ld] != 0.0) iscomplex = PETSC_TRUE;
  PetscScalar *work = (PetscScalar *)malloc((size_t)2 * ld * sizeof(PetscScalar));
  if (!work) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  ksd_trevc_right(n, A, ld, *k, work, work + ld);
  if (ds->state >= DS_STATE_CONDENSED) {
    /* accumulate and normalize eigenvectors */
    for (i = 0; i < n; i++) { PetscScalar s = 0.0; for (j = 0; j < n; j++) s += Q[i + j * ld] * work[j]; Y[i] = s; }
    if (iscomplex) for (i = 0; i < n; i++) { PetscScalar s = 0.0; for (j = 0; j < n; j++) s += Q[i + j * ld] * work[ld + j]; Y[ld + i] = s; }
    norm = 0.0; for (i = 0; i < n; i++) norm = hypot(norm, Y[i]);
    if (iscomplex) { tmp = 0.0; for (i = 0; i < n; i++) tmp = hypot(tmp, Y[ld + i]); norm = SlepcAbsEigenvalue(norm, tmp); }
    tmp = 1.0 / norm;
    for (i = 0; i < n; i++) Y[i] *= tmp;
    if (iscomplex) for (i = 0; i < n; i++) Y[ld + i] *= tmp;
  } else {
    memcpy(Y, work, (size_t)n * sizeof(PetscScalar));
    if (iscomplex) memcpy(Y + ld, work + ld, (size_t)n * sizeof(PetscScalar));
  }
  free(work);
  if (rnorm) {
    if (iscomplex) *rnorm = SlepcAbsEigenvalue(Y[n - 1], Y[n - 1 + ld]);
    else *rnorm = fabs(Y[n - 1]);
  }
  if (iscomplex) (*k)++;
  return 0;
}

/* dsnhep.c:194-254 */
static PetscErrorCode DSVectors_NHEP_Eigen_All(DS ds) {
  PetscErrorCode ierr;
  PetscInt i, n = ds->n, ld = ds->ld;
  PetscScalar *A = ds->mat[DS_MAT_A];
  const DSStateType st = ds->state;
  for (i = 0; i < n; i++) {
    const PetscBool iscomplex = (i < n - 1 && A[i + 1 + i * ld] != 0.0) ? PETSC_TRUE : PETSC_FALSE;
    PetscInt k = i;
    ierr = DSVectors_NHEP_Eigen_Some(ds, &k, NULL);CHKERRQ(ierr);
    if (st < DS_STATE_CONDENSED) {
      /* not back-transformed: normalize here */
      PetscScalar *Y = ds->mat[DS_MAT_X] + (size_t)i * ld;
      PetscReal norm = 0.0, tmp = 0.0;
      PetscInt r;
      for (r = 0; r < n; r++) norm = hypot(norm, Y[r]);
      if (iscomplex) { for (r = 0; r < n; r++) tmp = hypot(tmp, Y[ld + r]); norm = SlepcAbsEigenvalue(norm, tmp); }
      for (r = 0; r < n; r++) Y[r] /= norm;
      if (iscomplex) for (r = 0; r < n; r++) Y[ld + r] /= norm;
    }
    if (iscomplex) i++;
  }
  return 0;
}

/* dsnhep.c:545-567 */
static PetscErrorCode DSTruncate_NHEP(DS ds, PetscInt n) {
  PetscInt i, newn, ld = ds->ld, l = ds->l;
  PetscScalar *A = ds->mat[DS_MAT_A];
  if (ds->state == DS_STATE_CONDENSED) ds->t = ds->n;
  /* be careful not to break a diagonal 2x2 block */
  if (A[n + (n - 1) * ld] == 0.0) newn = n;
  else newn = (n < ds->n - 1) ? n + 1 : n - 1;
  if (ds->extrarow && ds->k == ds->n) {
    for (i = l; i < newn; i++) A[newn + i * ld] = A[ds->n + i * ld];
    for (i = l; i < ds->n; i++) A[ds->n + i * ld] = 0.0;
  }
  ds->k = 0;
  ds->n = newn;
  return 0;
}

/* dsnhep.c:447-467 */
static PetscErrorCode DSUpdateExtraRow_NHEP(DS ds) {
  PetscInt i, j, n = ds->n, ld = ds->ld;
  PetscScalar *A = ds->mat[DS_MAT_A], *Q = ds->mat[DS_MAT_Q];
  PetscScalar *x = (PetscScalar *)malloc((size_t)2 * ld * sizeof(PetscScalar));
  if (!x) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  PetscScalar *y = x + ld;
  for (i = 0; i < n; i++) x[i] = A[n + i * ld];
  for (j = 0; j < n; j++) { PetscScalar s = 0.0; for (i = 0; i < n; i++) s += Q[i + j * ld] * x[i]; y[j] = s; }
  for (i = 0; i < n; i++) A[n + i * ld] = y[i];
  free(x);
  ds->k = n;
  return 0;
}

/* ---- SVD (dssvd.c) ---------------------------------------------------------------------------------- */
/* dssvd.c:64-98, compact (arrow + upper bidiagonal in T) to dense */
static PetscErrorCode DSSwitchToDense_SVD(DS ds) {
  PetscReal *T = ds->rmat;
  PetscScalar *A = ds->mat[DS_MAT_A];
  PetscInt i, m = ds->m, k = ds->k, ld = ds->ld;
  if (!m) SETERRQ(PETSC_ERR_ARG_WRONG, "m was not set");
  memset(A, 0, (size_t)ld * ld * sizeof(PetscScalar));
  for (i = 0; i < k; i++) { A[i + i * ld] = T[i]; A[i + k * ld] = T[i + ld]; }
  A[k + k * ld] = T[k];
  for (i = k + 1; i < m; i++) { A[i + i * ld] = T[i]; A[i - 1 + i * ld] = T[i - 1 + ld]; }
  return 0;
}

/* dssvd.c:197-290 (DSSolve_SVD_DC); the bidiagonal and the arrow case both go through the
 * dense Jacobi routine */
static PetscErrorCode DSSolve_SVD(DS ds, PetscScalar *wr) {
  PetscErrorCode ierr;
  PetscInt i, l = ds->l, n = ds->n, m = ds->m, ld = ds->ld;
  PetscScalar *A = ds->mat[DS_MAT_A], *U = ds->mat[DS_MAT_U], *VT = ds->mat[DS_MAT_VT];
  PetscReal *d = ds->rmat, *e = ds->rmat + ld;
  if (!ds->compact) SETERRQ(PETSC_ERR_SUP, "Only compact storage is available for DSSVD");
  if (n != m) SETERRQ(PETSC_ERR_SUP, "Only square projected problems are available for DSSVD");
  memset(U, 0, (size_t)ld * ld * sizeof(PetscScalar));
  memset(VT, 0, (size_t)ld * ld * sizeof(PetscScalar));
  for (i = 0; i < l; i++) { U[i + i * ld] = 1.0; VT[i + i * ld] = 1.0; wr[i] = d[i]; }
  if (ds->state > DS_STATE_RAW) {
    /* upper bidiagonal: d on the diagonal, e above it */
    memset(A, 0, (size_t)ld * ld * sizeof(PetscScalar));
    for (i = l; i < n; i++) { A[i + i * ld] = d[i]; if (i + 1 < n) A[i + (i + 1) * ld] = e[i]; }
  } else {
    ierr = DSSwitchToDense_SVD(ds);CHKERRQ(ierr);
  }
  if (n > l) {
    const int info = ksd_gesvd(n - l, A + l + (size_t)l * ld, ld, d + l, U + l + (size_t)l * ld, ld, VT + l + (size_t)l * ld, ld);
    if (info) SETERRQ(PETSC_ERR_LIB, "Error in the dense SVD: %d", info);
  }
  for (i = l; i < n; i++) wr[i] = d[i];
  if (n > 1) memset(e, 0, (size_t)(n - 1) * sizeof(PetscReal));
  return 0;
}

/* dssvd.c:167-195 with DSSortEigenvaluesReal_Private (dspriv.c:242-264) and
 * DSPermuteBoth_Private (dspriv.c:335-362) */
static PetscErrorCode DSSort_SVD(DS ds, PetscScalar *wr) {
  PetscErrorCode ierr;
  PetscInt n = PetscMin(ds->n, ds->m), m = ds->m, l = ds->l, i, j, k, p, result, tmp, *perm = ds->perm, ld = ds->ld;
  PetscReal *d = ds->rmat;
  PetscScalar *U = ds->mat[DS_MAT_U], *VT = ds->mat[DS_MAT_VT];
  for (i = l + 1; i < n; i++) {
    const PetscScalar re = d[perm[i]];
    j = i - 1;
    ierr = ds->cmp(re, 0.0, d[perm[j]], 0.0, &result, ds->cmpctx);CHKERRQ(ierr);
    while (result < 0 && j >= l) {
      tmp = perm[j]; perm[j] = perm[j + 1]; perm[j + 1] = tmp; j--;
      if (j >= l) { ierr = ds->cmp(re, 0.0, d[perm[j]], 0.0, &result, ds->cmpctx);CHKERRQ(ierr); }
    }
  }
  for (i = l; i < n; i++) wr[i] = d[perm[i]];
  for (i = l; i < n; i++) {
    p = perm[i];
    if (p != i) {
      j = i + 1;
      while (perm[j] != i) j++;
      perm[j] = p; perm[i] = i;
      for (k = 0; k < n; k++) { PetscScalar t = U[k + p * ld]; U[k + p * ld] = U[k + i * ld]; U[k + i * ld] = t; }
      for (k = 0; k < m; k++) { PetscScalar t = VT[p + k * ld]; VT[p + k * ld] = VT[i + k * ld]; VT[i + k * ld] = t; }
    }
  }
  for (i = l; i < n; i++) d[i] = wr[i];
  return 0;
}

/* ---- interface (dsops.c) ------------------------------------------------------------------------- */
PetscErrorCode DSSolve(DS ds, PetscScalar *eigr, PetscScalar *eigi) {
  PetscErrorCode ierr;
  DSCheckAlloc(ds);
  if (!eigr) SETERRQ(PETSC_ERR_ARG_NULL, "Null pointer: eigr");
  if (ds->state >= DS_STATE_CONDENSED) return 0;
  if (ds->type == 0) { ierr = DSSolve_HEP_QR(ds, eigr, eigi);CHKERRQ(ierr); }
  else if (ds->type == 2) { ierr = DSSolve_SVD(ds, eigr);CHKERRQ(ierr); }
  else { ierr = DSSolve_NHEP(ds, eigr, eigi);CHKERRQ(ierr); }
  ds->state = DS_STATE_CONDENSED;
  return 0;
}

PetscErrorCode DSSort(DS ds, PetscScalar *eigr, PetscScalar *eigi, PetscScalar *rr, PetscScalar *ri, PetscInt *k) {
  PetscErrorCode ierr;
  PetscInt i;
  DSCheckAlloc(ds);
  DSCheckSolved(ds);
  if (!eigr) SETERRQ(PETSC_ERR_ARG_NULL, "Null pointer: eigr");
  if (ds->state == DS_STATE_TRUNCATED) SETERRQ(PETSC_ERR_ORDER, "Cannot sort a truncated DS");
  if (!ds->cmp) SETERRQ(PETSC_ERR_ORDER, "Must provide a sorting criterion first");
  if (k && !rr) SETERRQ(PETSC_ERR_ARG_WRONG, "Argument k can only be used together with rr");
  if (rr && rr != eigr) SETERRQ(PETSC_ERR_SUP, "Sorting by auxiliary values is not available");
  for (i = 0; i < ds->n; i++) ds->perm[i] = i;
  if (ds->type == 0) { ierr = DSSort_HEP(ds, eigr, eigi);CHKERRQ(ierr); }
  else if (ds->type == 2) { ierr = DSSort_SVD(ds, eigr);CHKERRQ(ierr); }
  else { ierr = DSSort_NHEP_Total(ds, eigr, eigi);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode DSUpdateExtraRow(DS ds) {
  PetscErrorCode ierr;
  DSCheckAlloc(ds);
  if (!ds->extrarow) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Should have called DSSetExtraRow");
  if (ds->type == 2) SETERRQ(PETSC_ERR_SUP, "DSSVD has no extra row");
  if (ds->type == 0) { ierr = DSUpdateExtraRow_HEP(ds);CHKERRQ(ierr); }
  else { ierr = DSUpdateExtraRow_NHEP(ds);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode DSVectors(DS ds, DSMatType mat, PetscInt *j, PetscReal *rnorm) {
  PetscErrorCode ierr;
  DSCheckAlloc(ds);
  if (mat >= DS_NUM_MAT) SETERRQ(PETSC_ERR_ARG_WRONG, "Invalid matrix");
  if (rnorm && !j) SETERRQ(PETSC_ERR_ORDER, "Must give a value of j");
  if (ds->type == 0) { ierr = DSVectors_HEP(ds, mat, j, rnorm);CHKERRQ(ierr); }
  else if (ds->type == 2) {          /* DSVectors_SVD dssvd.c:152-165 */
    if (mat != DS_MAT_U && mat != DS_MAT_VT) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid mat parameter");
    if (rnorm) *rnorm = 0.0;
  } else {
    if (mat != DS_MAT_X) SETERRQ(PETSC_ERR_SUP, "Only right eigenvectors are available");
    if (j) { ierr = DSVectors_NHEP_Eigen_Some(ds, j, rnorm);CHKERRQ(ierr); }
    else { ierr = DSVectors_NHEP_Eigen_All(ds);CHKERRQ(ierr); }
    if (ds->state < DS_STATE_CONDENSED) ds->state = DS_STATE_CONDENSED;
  }
  return 0;
}

PetscErrorCode DSTruncate(DS ds, PetscInt n) {
  PetscErrorCode ierr;
  DSCheckAlloc(ds);
  DSCheckSolved(ds);
  if (n < ds->l || n > ds->n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of n. Must be between l and n");
  if (ds->type == 2) SETERRQ(PETSC_ERR_SUP, "DSSVD cannot be truncated");
  if (ds->type == 0) { ierr = DSTruncate_HEP(ds, n);CHKERRQ(ierr); }
  else { ierr = DSTruncate_NHEP(ds, n);CHKERRQ(ierr); }
  ds->state = DS_STATE_TRUNCATED;
  return 0;
}

PetscErrorCode DSGetMat(DS ds, DSMatType m, PetscScalar *out, PetscInt *rows, PetscInt *cols) {
  PetscInt j;
  DSCheckAlloc(ds);
  if (m >= DS_NUM_MAT || m == DS_MAT_T || !ds->mat[m]) SETERRQ(PETSC_ERR_ARG_WRONG, "Invalid matrix");
  const PetscInt r = PetscMax(ds->n, ds->t), c = ds->n;
  if (rows) *rows = r;
  if (cols) *cols = c;
  if (out) for (j = 0; j < c; j++) memcpy(out + (size_t)j * r, ds->mat[m] + (size_t)j * ds->ld, (size_t)r * sizeof(PetscScalar));
  return 0;
}

PetscErrorCode DSDestroy(DS *ds) {
  if (!ds || !*ds) return 0;
  for (int i = 0; i < DS_NUM_MAT; i++) free((*ds)->mat[i]);
  free((*ds)->rmat);
  free((*ds)->perm);
  free(*ds);
  *ds = NULL;
  return 0;
}

/* ---- sorting criteria (src/sys/slepcsc.c:107-259) ------------------------------------------------- */
#define CMP(name, A, B, LESS, GREATER) \
  PetscErrorCode name(PetscScalar ar, PetscScalar ai, PetscScalar br, PetscScalar bi, PetscInt *result, void *ctx) { \
    const PetscReal a = (A), b = (B); (void)ar; (void)ai; (void)br; (void)bi; (void)ctx; \
    if (a LESS b) *result = 1; else if (a GREATER b) *result = -1; else *result = 0; return 0; }
CMP(SlepcCompareLargestMagnitude, SlepcAbsEigenvalue(ar, ai), SlepcAbsEigenvalue(br, bi), <, >)
CMP(SlepcCompareSmallestMagnitude, SlepcAbsEigenvalue(ar, ai), SlepcAbsEigenvalue(br, bi), >, <)
CMP(SlepcCompareLargestReal, ar, br, <, >)
CMP(SlepcCompareSmallestReal, ar, br, >, <)
CMP(SlepcCompareLargestImaginary, fabs(ai), fabs(bi), <, >)
CMP(SlepcCompareSmallestImaginary, fabs(ai), fabs(bi), >, <)

/* slepcsc.c:310-329: positive real parts first, then by increasing magnitude */
/* slepcsc.c:225-262 */
PetscErrorCode SlepcCompareTargetMagnitude(PetscScalar ar, PetscScalar ai, PetscScalar br, PetscScalar bi, PetscInt *result, void *ctx) {
  const PetscScalar t = *(PetscScalar *)ctx;
  const PetscReal a = SlepcAbsEigenvalue(ar - t, ai), b = SlepcAbsEigenvalue(br - t, bi);
  *result = a > b ? 1 : (a < b ? -1 : 0);
  return 0;
}
PetscErrorCode SlepcCompareTargetReal(PetscScalar ar, PetscScalar ai, PetscScalar br, PetscScalar bi, PetscInt *result, void *ctx) {
  const PetscScalar t = *(PetscScalar *)ctx;
  const PetscReal a = fabs(ar - t), b = fabs(br - t);
  (void)ai; (void)bi;
  *result = a > b ? 1 : (a < b ? -1 : 0);
  return 0;
}

PetscErrorCode SlepcCompareSmallestPosReal(PetscScalar ar, PetscScalar ai, PetscScalar br, PetscScalar bi, PetscInt *result, void *ctx) {
  const PetscBool aisright = ar > 0.0, bisright = br > 0.0;
  (void)ctx;
  if (aisright == bisright) {
    const PetscReal a = SlepcAbsEigenvalue(ar, ai), b = SlepcAbsEigenvalue(br, bi);
    *result = a > b ? 1 : (a < b ? -1 : 0);
  } else *result = aisright ? -1 : 1;
  return 0;
}

PetscErrorCode SlepcSortEigenvalues(SlepcCompareFn cmp, void *ctx, PetscInt n, PetscScalar *eigr, PetscScalar *eigi, PetscInt *perm) {
  PetscErrorCode ierr;
  PetscScalar re, im;
  PetscInt i, j, result, tmp;
  for (i = n - 1; i >= 0; i--) {
    re = eigr[perm[i]];
    im = eigi[perm[i]];
    j = i + 1;
    if (im != 0) { i--; im = eigi[perm[i]]; }       /* complex eigenvalue */
    while (j < n) {
      ierr = cmp(re, im, eigr[perm[j]], eigi[perm[j]], &result, ctx);CHKERRQ(ierr);
      if (result < 0) break;
      /* keep together every complex conjugated eigenpair */
      if (!im) {
        if (eigi[perm[j]] == 0.0) { tmp = perm[j - 1]; perm[j - 1] = perm[j]; perm[j] = tmp; j++; }
        else { tmp = perm[j - 1]; perm[j - 1] = perm[j]; perm[j] = perm[j + 1]; perm[j + 1] = tmp; j += 2; }
      } else {
        if (eigi[perm[j]] == 0.0) { tmp = perm[j - 2]; perm[j - 2] = perm[j]; perm[j] = perm[j - 1]; perm[j - 1] = tmp; j++; }
        else {
          tmp = perm[j - 2]; perm[j - 2] = perm[j]; perm[j] = tmp;
          tmp = perm[j - 1]; perm[j - 1] = perm[j + 1]; perm[j + 1] = tmp;
          j += 2;
        }
      }
    }
  }
  return 0;
}
