/* BV interface layer: creation / registry / active columns / column views
 * (src/sys/classes/bv/interface/bvbasic.c, bvfunc.c, bvregis.c), the dispatch
 * wrappers (bvops.c, bvglobal.c) and Gram-Schmidt (bvorthog.c).  The numerical
 * work is done by the op table of the BV type; this file only holds the host
 * control flow, which is the same for every type. */
#include <stdio.h>

#include "dense.h"
#include "ksimpl.h"

PetscInt64 ks_gs_passes = 0;

/* ---- registry (bvfunc.c:677-684, bvregis.c:45-48) ------------------------------- */
#define KS_MAX_BV_TYPES 16
static struct { char name[32]; PetscErrorCode (*create)(BV); } bv_types[KS_MAX_BV_TYPES];
static int bv_ntypes = 0;
static int bv_registered_builtin = 0;

static void BVRegisterAll(void) {
  if (bv_registered_builtin) return;
  bv_registered_builtin = 1;
  BVRegister("hip", BVCreate_HIP);
}

PetscErrorCode BVRegister(const char *name, PetscErrorCode (*create)(BV)) {
  BVRegisterAll();
  if (!name || !create) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  for (int i = 0; i < bv_ntypes; i++)
    if (!strcmp(bv_types[i].name, name)) { bv_types[i].create = create; return 0; }
  if (bv_ntypes == KS_MAX_BV_TYPES) SETERRQ(PETSC_ERR_MEM, "Too many BV types");
  snprintf(bv_types[bv_ntypes].name, sizeof(bv_types[0].name), "%s", name);
  bv_types[bv_ntypes++].create = create;
  return 0;
}

/* ---- creation, sizes (bvbasic.c:45-72,171-199; bvfunc.c:160-200) ------------------- */
PetscErrorCode BVCreate(void *ctx, BV *bv) {
  BVRegisterAll();
  if (!bv) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  BV b = (BV)calloc(1, sizeof(*b));
  if (!b) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  b->ctx = ctx;
  b->n = -1; b->N = -1;
  b->orthog_type = BV_ORTHOG_CGS;               /* bvfunc.c:184-187 */
  b->orthog_ref = BV_ORTHOG_REFINE_IFNEEDED;
  b->orthog_eta = 0.7071;
  b->ci[0] = b->ci[1] = -1;
  b->rand_seed = 0x12345678;                    /* bvbasic.c:884 */
  *bv = b;
  return 0;
}

static PetscErrorCode BVSetUpStorage(BV bv) {
  PetscErrorCode ierr;
  PetscErrorCode (*r)(BV) = NULL;
  for (int i = 0; i < bv_ntypes; i++) if (!strcmp(bv_types[i].name, bv->type_name)) r = bv_types[i].create;
  if (!r) SETERRQ(PETSC_ERR_ARG_WRONG, "Unable to find requested BV type %s", bv->type_name);
  bv->ci[0] = bv->ci[1] = -bv->nc - 1;
  ierr = (*r)(bv);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVSetType(BV bv, const char *type) {
  PetscErrorCode ierr;
  if (!bv || !type) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!strcmp(bv->type_name, type)) return 0;
  int found = 0;
  for (int i = 0; i < bv_ntypes; i++) if (!strcmp(bv_types[i].name, type)) found = 1;
  if (!found) SETERRQ(PETSC_ERR_ARG_WRONG, "Unable to find requested BV type %s", type);
  if (bv->data) SETERRQ(PETSC_ERR_ORDER, "Cannot change the type of a BV that already has storage");
  snprintf(bv->type_name, sizeof(bv->type_name), "%s", type);
  if (bv->n >= 0) { ierr = BVSetUpStorage(bv);CHKERRQ(ierr); }   /* sizes known: create now, else deferred */
  return 0;
}

PetscErrorCode BVSetSizesFromVec(BV bv, Vec t, PetscInt m) {
  PetscErrorCode ierr;
  if (!bv || !t) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (m <= 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Number of columns %d must be positive", m);
  if (bv->t) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Template vector was already set by a previous call to BVSetSizes/FromVec");
  bv->n = t->n; bv->N = t->N;
  bv->m = m; bv->k = m; bv->l = 0; bv->nc = 0;
  ierr = VecCreateWithOps(t->ops, t->ctx, t->n, t->N, t->row0, PETSC_FALSE, &bv->t);CHKERRQ(ierr);
  if (bv->type_name[0]) { ierr = BVSetUpStorage(bv);CHKERRQ(ierr); }
  return 0;
}

#define BVCheckSizes(bv) do { if (!(bv)) SETERRQ(PETSC_ERR_ARG_NULL, "Null BV"); \
  if (!(bv)->data) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "BV sizes and type have not been defined: call BVSetSizes() and BVSetType()"); } while (0)

PetscErrorCode BVGetSizes(BV bv, PetscInt64 *n, PetscInt64 *N, PetscInt *m) {
  if (!bv) SETERRQ(PETSC_ERR_ARG_NULL, "Null BV");
  if (n) *n = bv->n;
  if (N) *N = bv->N;
  if (m) *m = bv->m;
  return 0;
}

PetscErrorCode BVResize(BV bv, PetscInt m, PetscBool copy) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (m <= 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Number of columns %d must be positive", m);
  if (bv->nc) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Cannot resize a BV with constraints");
  if (bv->m == m) return 0;
  ierr = bv->ops.resize(bv, m, copy);CHKERRQ(ierr);
  free(bv->buffer); bv->buffer = NULL;
  free(bv->h); bv->h = NULL; bv->c = NULL;            /* c points into the allocation of h */
  free(bv->omega); bv->omega = NULL;
  if (bv->cached) { ierr = BVDestroy(&bv->cached);CHKERRQ(ierr); }
  bv->m = m;
  bv->k = PetscMin(bv->k, m);
  bv->l = PetscMin(bv->l, m);
  return 0;
}

PetscErrorCode BVSetActiveColumns(BV bv, PetscInt l, PetscInt k) {
  BVCheckSizes(bv);
  if (k < 0 || k > bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of k (%d). Must be between 0 and m (%d)", k, bv->m);
  if (l < 0 || l > k) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Illegal value of l (%d). Must be between 0 and k (%d)", l, k);
  bv->l = l; bv->k = k;
  return 0;
}
PetscErrorCode BVGetActiveColumns(BV bv, PetscInt *l, PetscInt *k) {
  if (!bv) SETERRQ(PETSC_ERR_ARG_NULL, "Null BV");
  if (l) *l = bv->l;
  if (k) *k = bv->k;
  return 0;
}

PetscErrorCode BVSetOrthogonalization(BV bv, BVOrthogType type, BVOrthogRefineType refine, PetscReal eta, BVOrthogBlockType block) {
  if (!bv) SETERRQ(PETSC_ERR_ARG_NULL, "Null BV");
  if (type != BV_ORTHOG_CGS && type != BV_ORTHOG_MGS) SETERRQ(PETSC_ERR_ARG_WRONG, "Unknown orthogonalization type");
  if (refine < BV_ORTHOG_REFINE_IFNEEDED || refine > BV_ORTHOG_REFINE_ALWAYS) SETERRQ(PETSC_ERR_ARG_WRONG, "Unknown refinement type");
  if (block < BV_ORTHOG_BLOCK_GS || block > BV_ORTHOG_BLOCK_TSQR) SETERRQ(PETSC_ERR_ARG_WRONG, "Unknown block orthogonalization type");
  bv->orthog_type = type; bv->orthog_ref = refine; bv->orthog_block = block;
  if (eta == 0.0) bv->orthog_eta = 0.7071;       /* PETSC_DEFAULT */
  else {
    if (eta <= 0.0 || eta > 1.0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid eta value");
    bv->orthog_eta = eta;
  }
  return 0;
}

/* ---- non-standard inner product (bvimpl.h:107-205, bvbasic.c:481-729) ------------------------ */
/* bvimpl.h:113-134: the square root of z^T B z, signed when B is indefinite */
static PetscErrorCode BV_SafeSqrt(BV bv, PetscScalar alpha, PetscReal *res) {
  if (bv->indef) *res = (alpha < 0.0) ? -sqrt(-alpha) : sqrt(alpha);
  else {
    if (alpha < 0.0) SETERRQ(1, "The inner product is not well defined: indefinite matrix");
    *res = sqrt(alpha);
  }
  return 0;
}

/* bvimpl.h:140-151: Bx = B*x unless it already holds the product of this very vector */
static PetscErrorCode BV_IPMatMult(BV bv, Vec x) {
  PetscErrorCode ierr;
  if (x->id != bv->xid || x->state != bv->xstate) {
    ierr = MatMult(bv->matrix, x, bv->Bx);CHKERRQ(ierr);
    bv->xid = x->id;
    bv->xstate = x->state;
  }
  return 0;
}

/* bvimpl.h:193-205 */
static PetscErrorCode BV_AllocateSignature(BV bv) {
  if (bv->indef && !bv->omega) {
    bv->omega = (PetscReal *)malloc((size_t)(bv->nc + bv->m) * sizeof(PetscReal));
    if (!bv->omega) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    for (PetscInt i = 0; i < bv->nc + bv->m; i++) bv->omega[i] = 1.0;
  }
  return 0;
}

/* ops->dotvec in the inner product of the BV: the type sees B*y (svec.c:131-134) */
static PetscErrorCode BV_DotVecOp(BV X, Vec y, PetscScalar *q) {
  PetscErrorCode ierr;
  Vec z = y;
  if (X->matrix) { ierr = BV_IPMatMult(X, y);CHKERRQ(ierr); z = X->Bx; }
  ierr = X->ops.dotvec(X, z, q);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVSetMatrix(BV bv, Mat B, PetscBool indef) {
  PetscErrorCode ierr;
  if (!bv) SETERRQ(PETSC_ERR_ARG_NULL, "Null BV");
  if (B) {
    if (B->n != B->cn) SETERRQ(PETSC_ERR_ARG_SIZ, "Matrix must be square");
    if (bv->m && bv->n != B->cn) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension BV %lld, Mat %lld", (long long)bv->n, (long long)B->cn);
  }
  bv->matrix = B;
  bv->indef = indef;
  bv->xid = 0; bv->xstate = 0;
  bv->bvstate = -1;
  if (B && !bv->Bx) { ierr = MatCreateVecs(B, &bv->Bx, NULL);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode BVGetMatrix(BV bv, Mat *B, PetscBool *indef) {
  if (!bv) SETERRQ(PETSC_ERR_ARG_NULL, "Null BV");
  if (B) *B = bv->matrix;
  if (indef) *indef = bv->indef;
  return 0;
}

PetscErrorCode BVApplyMatrix(BV bv, Vec x, Vec y) {
  PetscErrorCode ierr;
  if (!bv || !x || !y) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (bv->matrix) {
    ierr = BV_IPMatMult(bv, x);CHKERRQ(ierr);
    ierr = VecCopy(bv->Bx, y);CHKERRQ(ierr);
  } else { ierr = VecCopy(x, y);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode BVGetCachedBV(BV bv, BV *cached) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (!cached) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!bv->cached) {
    ierr = BVCreate(bv->ctx, &bv->cached);CHKERRQ(ierr);
    ierr = BVSetSizesFromVec(bv->cached, bv->t, bv->m);CHKERRQ(ierr);
    ierr = BVSetType(bv->cached, bv->type_name);CHKERRQ(ierr);
    ierr = BVSetOrthogonalization(bv->cached, bv->orthog_type, bv->orthog_ref, bv->orthog_eta, bv->orthog_block);CHKERRQ(ierr);
    bv->bvstate = -1;
  }
  *cached = bv->cached;
  return 0;
}

/* bvimpl.h:157-173: cached = B*X on the active columns, recomputed when X has changed */
static PetscErrorCode BV_IPMatMultBV(BV bv) {
  PetscErrorCode ierr;
  ierr = BVGetCachedBV(bv, &bv->cached);CHKERRQ(ierr);
  ierr = BVSetActiveColumns(bv->cached, bv->l, bv->k);CHKERRQ(ierr);
  if (bv->state != bv->bvstate) {
    if (bv->matrix) { ierr = BVMatMult(bv, bv->matrix, bv->cached);CHKERRQ(ierr); }
    else { ierr = BVCopy(bv, bv->cached);CHKERRQ(ierr); }
    bv->bvstate = bv->state;
  }
  return 0;
}

PetscErrorCode BVApplyMatrixBV(BV X, BV Y) {
  PetscErrorCode ierr;
  BVCheckSizes(X);
  if (Y) {
    if (X->matrix) { ierr = BVMatMult(X, X->matrix, Y);CHKERRQ(ierr); }
    else { ierr = BVCopy(X, Y);CHKERRQ(ierr); }
  } else { ierr = BV_IPMatMultBV(X);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode BVSetSignature(BV bv, const PetscReal *omega) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (!omega) SETERRQ(PETSC_ERR_ARG_NULL, "Null array");
  ierr = BV_AllocateSignature(bv);CHKERRQ(ierr);
  if (bv->indef) for (PetscInt i = 0; i < bv->k; i++) bv->omega[bv->nc + i] = omega[i];
  return 0;
}

PetscErrorCode BVGetSignature(BV bv, PetscReal *omega) {
  BVCheckSizes(bv);
  if (!omega) SETERRQ(PETSC_ERR_ARG_NULL, "Null array");
  for (PetscInt i = 0; i < bv->k; i++) omega[i] = (bv->indef && bv->omega) ? bv->omega[bv->nc + i] : 1.0;
  return 0;
}

PetscErrorCode BVDuplicateResize(BV V, PetscInt m, BV *W) {
  PetscErrorCode ierr;
  BVCheckSizes(V);
  if (!W) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ierr = BVCreate(V->ctx, W);CHKERRQ(ierr);
  ierr = BVSetSizesFromVec(*W, V->t, m);CHKERRQ(ierr);
  ierr = BVSetType(*W, V->type_name);CHKERRQ(ierr);
  ierr = BVSetMatrix(*W, V->matrix, V->indef);CHKERRQ(ierr);
  ierr = BVSetOrthogonalization(*W, V->orthog_type, V->orthog_ref, V->orthog_eta, V->orthog_block);CHKERRQ(ierr);
  (*W)->state++;
  return 0;
}
PetscErrorCode BVDuplicate(BV V, BV *W) { BVCheckSizes(V); return BVDuplicateResize(V, V->m, W); }

/* ---- constraints (bvbasic.c:262-321, bvfunc.c:414-442) ---------------------------------------- */
PetscErrorCode BVGetNumConstraints(BV bv, PetscInt *nc) {
  if (!bv || !nc) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  *nc = bv->nc;
  return 0;
}

static void BVFreeCoefficients(BV bv) {
  free(bv->buffer); bv->buffer = NULL;
  free(bv->h); bv->h = NULL; bv->c = NULL;
  free(bv->omega); bv->omega = NULL;
}

PetscErrorCode BVSetNumConstraints(BV V, PetscInt nc) {
  PetscErrorCode ierr;
  PetscInt total, diff, i;
  Vec x, y;
  BVCheckSizes(V);
  if (nc < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Number of constraints (given %d) cannot be negative", nc);
  if (V->ci[0] != -V->nc - 1 || V->ci[1] != -V->nc - 1) SETERRQ(PETSC_ERR_SUP, "Cannot call BVSetNumConstraints after BVGetColumn");
  diff = nc - V->nc;
  if (!diff) return 0;
  total = V->nc + V->m;
  if (total - nc <= 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Not enough columns for the given nc value");
  if (diff < 0) {             /* lessen constraints, shift contents of BV */
    for (i = 0; i < V->m; i++) {
      ierr = BVGetColumn(V, i, &x);CHKERRQ(ierr);
      ierr = BVGetColumn(V, i + diff, &y);CHKERRQ(ierr);
      ierr = VecCopy(x, y);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V, i, &x);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V, i + diff, &y);CHKERRQ(ierr);
    }
  }
  V->nc = nc;
  V->ci[0] = V->ci[1] = -V->nc - 1;
  V->l = 0;
  V->m = total - nc;
  V->k = V->m;
  BVFreeCoefficients(V);
  V->state++;
  return 0;
}

PetscErrorCode BVInsertConstraints(BV V, PetscInt *nc, Vec *C) {
  PetscErrorCode ierr;
  PetscInt msave;
  BVCheckSizes(V);
  if (!nc) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!*nc) return 0;
  if (*nc < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Number of constraints (given %d) cannot be negative", *nc);
  if (!C || !*C) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (V->nc) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Constraints already present in this BV object");
  if (V->ci[0] != -1 || V->ci[1] != -1) SETERRQ(PETSC_ERR_SUP, "Cannot call BVInsertConstraints after BVGetColumn");
  msave = V->m;
  ierr = BVResize(V, *nc + V->m, PETSC_FALSE);CHKERRQ(ierr);
  ierr = BVInsertVecs(V, 0, nc, C, PETSC_TRUE);CHKERRQ(ierr);
  V->nc = *nc;
  V->m = msave;
  V->l = 0; V->k = msave;
  V->ci[0] = V->ci[1] = -V->nc - 1;
  BVFreeCoefficients(V);          /* sized for the columns as they were counted during the insertion */
  V->state++;
  return 0;
}

/* ---- column views (bvbasic.c:1152-1224) ----------------------------------------------- */
PetscErrorCode BVGetColumn(BV bv, PetscInt j, Vec *v) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (j < 0 && -j > bv->nc) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "You requested constraint %d but only %d are available", -j, bv->nc);
  if (j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "You requested column %d but only %d are available", j, bv->m);
  if (j == bv->ci[0] || j == bv->ci[1]) SETERRQ(PETSC_ERR_SUP, "Column %d already fetched in a previous call to BVGetColumn", j);
  const int l = (bv->ci[0] == -bv->nc - 1) ? 0 : (bv->ci[1] == -bv->nc - 1) ? 1 : -1;    /* BVAvailableVec */
  if (l == -1) SETERRQ(PETSC_ERR_SUP, "Too many requested columns; you must call BVRestoreColumn for one of the previously fetched columns");
  ierr = bv->ops.getcolumn(bv, j, v);CHKERRQ(ierr);
  bv->ci[l] = j;
  bv->st[l] = bv->cv[l]->state;
  *v = bv->cv[l];
  return 0;
}

PetscErrorCode BVRestoreColumn(BV bv, PetscInt j, Vec *v) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (!v || !*v) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (j != bv->ci[0] && j != bv->ci[1]) SETERRQ(PETSC_ERR_ARG_WRONG, "Column %d has not been fetched with a call to BVGetColumn", j);
  const int l = (j == bv->ci[0]) ? 0 : 1;
  if (*v != bv->cv[l]) SETERRQ(PETSC_ERR_ARG_WRONG, "Argument 3 is not the same Vec that was obtained with BVGetColumn");
  if (bv->cv[l]->state != bv->st[l]) bv->state++;           /* the view was modified */
  if (bv->ops.restorecolumn) { ierr = bv->ops.restorecolumn(bv, j, v);CHKERRQ(ierr); }
  bv->ci[l] = -bv->nc - 1;
  *v = NULL;
  return 0;
}

/* bvbasic.c:791-827: host array of (nc+m)*m coefficients, leading dimension nc+m */
PetscErrorCode BVGetBuffer(BV bv, PetscScalar **buffer, PetscInt *ld) {
  BVCheckSizes(bv);
  if (!bv->buffer) {
    bv->buffer = (PetscScalar *)calloc((size_t)(bv->nc + bv->m) * bv->m, sizeof(PetscScalar));
    if (!bv->buffer) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  }
  if (buffer) *buffer = bv->buffer;
  if (ld) *ld = bv->nc + bv->m;
  return 0;
}

/* ---- data movement (bvfunc.c:280-377, bvbasic.c:1540-1650, bvops.c:462-479) -------------- */
PetscErrorCode BVInsertVec(BV bv, PetscInt j, Vec w) {
  PetscErrorCode ierr;
  Vec v;
  BVCheckSizes(bv);
  if (!w) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (w->N != bv->N || w->n != bv->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Vec sizes (global %lld, local %lld) do not match BV sizes (global %lld, local %lld)", (long long)w->N, (long long)w->n, (long long)bv->N, (long long)bv->n);
  if (j < 0 || j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, should be between 0 and %d", j, bv->m - 1);
  ierr = BVGetColumn(bv, j, &v);CHKERRQ(ierr);
  ierr = VecCopy(w, v);CHKERRQ(ierr);
  ierr = BVRestoreColumn(bv, j, &v);CHKERRQ(ierr);
  bv->state++;
  return 0;
}

PetscErrorCode BVInsertVecs(BV V, PetscInt s, PetscInt *m, Vec *W, PetscBool orth) {
  PetscErrorCode ierr;
  PetscInt i, ndep;
  PetscBool lindep;
  PetscReal norm;
  Vec v;
  BVCheckSizes(V);
  if (!m) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!*m) return 0;
  if (*m < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Number of vectors (given %d) cannot be negative", *m);
  if (!W || !*W) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if ((*W)->N != V->N || (*W)->n != V->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Vec sizes do not match BV sizes");
  if (s < 0 || s >= V->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument s has wrong value %d, should be between 0 and %d", s, V->m - 1);
  if (s + (*m) > V->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Too many vectors provided, there is only room for %d", V->m);
  ndep = 0;
  for (i = 0; i < *m; i++) {
    ierr = BVGetColumn(V, s + i - ndep, &v);CHKERRQ(ierr);
    ierr = VecCopy(W[i], v);CHKERRQ(ierr);
    ierr = BVRestoreColumn(V, s + i - ndep, &v);CHKERRQ(ierr);
    if (orth) {
      ierr = BVOrthogonalizeColumn(V, s + i - ndep, NULL, &norm, &lindep);CHKERRQ(ierr);
      if (norm == 0.0 || lindep) ndep++;
      else { ierr = BVScaleColumn(V, s + i - ndep, 1.0 / norm);CHKERRQ(ierr); }
    }
  }
  *m -= ndep;
  V->state++;
  return 0;
}

/* The reference fills the column from a PetscRandom (bvops.c:390-412); here the
 * values are a counter-based hash of (seed, global row), reproducible for any
 * row partition, like its -bv_reproducible_random mode. */
PetscErrorCode BVSetRandomColumn(BV bv, PetscInt j) {
  PetscErrorCode ierr;
  Vec v;
  BVCheckSizes(bv);
  if (j < 0 || j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  ierr = BVGetColumn(bv, j, &v);CHKERRQ(ierr);
  ierr = VecFillHash(v, bv->rand_seed);CHKERRQ(ierr);
  bv->rand_seed = bv->rand_seed * 6364136223846793005ULL + 1442695040888963407ULL;
  ierr = BVRestoreColumn(bv, j, &v);CHKERRQ(ierr);
  bv->state++;
  return 0;
}

PetscErrorCode BVCopy(BV V, BV W) {
  PetscErrorCode ierr;
  BVCheckSizes(V); BVCheckSizes(W);
  if (V->n != W->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension V %lld, W %lld", (long long)V->n, (long long)W->n);
  if (V->k - V->l > W->m - W->l) SETERRQ(PETSC_ERR_ARG_SIZ, "W has %d non-leading columns, not enough to store %d columns", W->m - W->l, V->k - V->l);
  if (!V->n) return 0;
  W->k = W->l + V->k - V->l;
  if (V->indef && V->matrix && V->indef == W->indef && V->matrix == W->matrix && V->omega) {      /* copy signature */
    ierr = BV_AllocateSignature(W);CHKERRQ(ierr);
    memcpy(W->omega + W->nc + W->l, V->omega + V->nc + V->l, (size_t)(V->k - V->l) * sizeof(PetscReal));
  }
  ierr = V->ops.copy(V, W);CHKERRQ(ierr);
  W->state++;
  return 0;
}

PetscErrorCode BVCopyVec(BV V, PetscInt j, Vec w) {
  PetscErrorCode ierr;
  Vec z;
  BVCheckSizes(V);
  if (!w) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (w->n != V->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension V %lld, w %lld", (long long)V->n, (long long)w->n);
  ierr = BVGetColumn(V, j, &z);CHKERRQ(ierr);
  ierr = VecCopy(z, w);CHKERRQ(ierr);
  ierr = BVRestoreColumn(V, j, &z);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVCopyColumn(BV V, PetscInt j, PetscInt i) {
  PetscErrorCode ierr;
  Vec z, w;
  BVCheckSizes(V);
  if (j == i) return 0;
  ierr = BVGetColumn(V, j, &z);CHKERRQ(ierr);
  ierr = BVGetColumn(V, i, &w);CHKERRQ(ierr);
  ierr = VecCopy(z, w);CHKERRQ(ierr);
  ierr = BVRestoreColumn(V, j, &z);CHKERRQ(ierr);
  ierr = BVRestoreColumn(V, i, &w);CHKERRQ(ierr);
  return 0;
}

/* ---- products (bvops.c:60-313) ------------------------------------------------------------- */
PetscErrorCode BVMult(BV Y, PetscScalar alpha, PetscScalar beta, BV X, const PetscScalar *Q, PetscInt ldq) {
  PetscErrorCode ierr;
  BVCheckSizes(Y); BVCheckSizes(X);
  if (X == Y) SETERRQ(PETSC_ERR_ARG_WRONG, "X and Y arguments must be different");
  if (Q) {
    if (ldq < X->k) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat argument has %d rows, should have at least %d", ldq, X->k);
  } else if (X->k - X->l != Y->k - Y->l) SETERRQ(PETSC_ERR_ARG_SIZ, "Y has %d active columns, should match %d active columns in X", Y->k - Y->l, X->k - X->l);
  if (X->n != Y->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension X %lld, Y %lld", (long long)X->n, (long long)Y->n);
  if (!X->n) return 0;
  ierr = Y->ops.mult(Y, alpha, beta, X, Q, ldq);CHKERRQ(ierr);
  Y->state++;
  return 0;
}

PetscErrorCode BVMultVec(BV X, PetscScalar alpha, PetscScalar beta, Vec y, PetscScalar *q) {
  PetscErrorCode ierr;
  BVCheckSizes(X);
  if (!y) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (X->n != y->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension X %lld, y %lld", (long long)X->n, (long long)y->n);
  if (!X->n) return 0;
  if (!q) { ierr = BVGetBuffer(X, &q, NULL);CHKERRQ(ierr); }
  ierr = X->ops.multvec(X, alpha, beta, y, q);CHKERRQ(ierr);
  y->state++;
  return 0;
}

PetscErrorCode BVMultColumn(BV X, PetscScalar alpha, PetscScalar beta, PetscInt j, PetscScalar *q) {
  PetscErrorCode ierr;
  PetscInt ksave;
  Vec y;
  BVCheckSizes(X);
  if (j < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  if (j >= X->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, X->m);
  ksave = X->k;
  X->k = j;
  if (!q) { ierr = BVGetBuffer(X, &q, NULL);CHKERRQ(ierr); }
  ierr = BVGetColumn(X, j, &y);CHKERRQ(ierr);
  ierr = X->ops.multvec(X, alpha, beta, y, q);
  if (!ierr) y->state++;
  X->k = ksave;
  CHKERRQ(ierr);
  ierr = BVRestoreColumn(X, j, &y);CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode BVMultInPlace_Check(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e, PetscBool trans) {
  if (!Q) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix");
  if (s < V->l || s > V->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument s has wrong value %d, should be between %d and %d", s, V->l, V->m);
  if (e < V->l || e > V->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument e has wrong value %d, should be between %d and %d", e, V->l, V->m);
  if (!trans && ldq < V->k) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat argument has %d rows, should have at least %d", ldq, V->k);
  if (trans && e > ldq) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat argument only has %d rows, the requested value of e is larger: %d", ldq, e);
  return 0;
}

PetscErrorCode BVMultInPlace(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e) {
  PetscErrorCode ierr;
  BVCheckSizes(V);
  ierr = BVMultInPlace_Check(V, Q, ldq, s, e, PETSC_FALSE);CHKERRQ(ierr);
  if (s >= e || !V->n) return 0;
  ierr = V->ops.multinplace(V, Q, ldq, s, e);CHKERRQ(ierr);
  V->state++;
  return 0;
}

PetscErrorCode BVMultInPlaceTranspose(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e) {
  PetscErrorCode ierr;
  BVCheckSizes(V);
  ierr = BVMultInPlace_Check(V, Q, ldq, s, e, PETSC_TRUE);CHKERRQ(ierr);
  if (s >= e || !V->n) return 0;
  ierr = V->ops.multinplacetrans(V, Q, ldq, s, e);CHKERRQ(ierr);
  V->state++;
  return 0;
}

/* ---- reductions (bvglobal.c:92-188,324-349,520-545,704-730) ------------------------------------ */
PetscErrorCode BVDot(BV X, BV Y, PetscScalar *M, PetscInt ldm) {
  PetscErrorCode ierr;
  BVCheckSizes(X); BVCheckSizes(Y);
  if (!M) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix");
  if (ldm < Y->k) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat argument has %d rows, should have at least %d", ldm, Y->k);
  if (X->n != Y->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension X %lld, Y %lld", (long long)X->n, (long long)Y->n);
  if (X->matrix != Y->matrix) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "X and Y must have the same inner product matrix");
  if (X->l == X->k || Y->l == Y->k) return 0;
  if (X->matrix) {            /* M = Y^T (B X): B X first, in the cached BV (bvglobal.c:125-130) */
    ierr = BV_IPMatMultBV(X);CHKERRQ(ierr);
    ierr = X->ops.dot(X->cached, Y, M, ldm);CHKERRQ(ierr);
  } else { ierr = X->ops.dot(X, Y, M, ldm);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode BVDotVec(BV X, Vec y, PetscScalar *q) {
  PetscErrorCode ierr;
  BVCheckSizes(X);
  if (!y) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (X->n != y->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension X %lld, y %lld", (long long)X->n, (long long)y->n);
  if (!q) { ierr = BVGetBuffer(X, &q, NULL);CHKERRQ(ierr); }
  ierr = BV_DotVecOp(X, y, q);CHKERRQ(ierr);
  return 0;
}

/* ---- split phase (bvglobal.c:190-310, 732-830) -------------------------------------------- */
PetscErrorCode BVDotVecBegin(BV X, Vec y, PetscScalar *m) {
  PetscErrorCode ierr;
  PetscScalar *vals;
  PetscInt off;
  (void)m;
  BVCheckSizes(X);
  if (!y) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (X->n != y->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension X %lld, y %lld", (long long)X->n, (long long)y->n);
  if (X->matrix) { ierr = BV_IPMatMult(X, y);CHKERRQ(ierr); y = X->Bx; }
  ierr = KSRedBegin_Private(X, X->ops.dotvec_post ? y->ops : NULL, y->ctx, X->k - X->l, 0, &vals);CHKERRQ(ierr);
  if (vals) ierr = X->ops.dotvec(X, y, vals);
  else {
    ierr = X->ops.dotvec_post(X, y, &off);
    if (!ierr) KSRedPosted_Private(off);
  }
  if (ierr) { KSRedAbandon_Private();CHKERRQ(ierr); }
  return 0;
}

/* X must have the active columns it had at the Begin */
PetscErrorCode BVDotVecEnd(BV X, Vec y, PetscScalar *m) {
  PetscErrorCode ierr;
  (void)y;
  BVCheckSizes(X);
  if (!m) { ierr = BVGetBuffer(X, &m, NULL);CHKERRQ(ierr); }
  return KSRedEnd_Private(X, X->k - X->l, m);
}

/* bvglobal.c:361-470: the products of column j with columns l..j-1 */
PetscErrorCode BVDotColumnBegin(BV X, PetscInt j, PetscScalar *m) {
  PetscErrorCode ierr;
  PetscInt ksave;
  Vec y;
  BVCheckSizes(X);
  if (j < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  if (j >= X->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, X->m);
  ksave = X->k;
  X->k = j;
  ierr = BVGetColumn(X, j, &y);
  if (!ierr) {
    ierr = BVDotVecBegin(X, y, m);
    { const PetscErrorCode ierr2 = BVRestoreColumn(X, j, &y); if (!ierr) ierr = ierr2; }
  }
  X->k = ksave;
  CHKERRQ(ierr);
  return 0;
}
PetscErrorCode BVDotColumnEnd(BV X, PetscInt j, PetscScalar *m) {
  PetscErrorCode ierr;
  PetscInt ksave;
  BVCheckSizes(X);
  if (j < 0 || j >= X->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j=%d out of range", j);
  ksave = X->k;
  X->k = j;
  ierr = BVDotVecEnd(X, NULL, m);
  X->k = ksave;
  CHKERRQ(ierr);
  return 0;
}

/* bvglobal.c:482-502,615-700: with a matrix the value posted is z^T B z (the entry belongs to
 * the vector B*z), otherwise the sum of squares as in VecNormBegin/End */
static PetscErrorCode BVNorm_Begin_Private(BV bv, Vec z, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  if (!bv->matrix) return VecNormBegin(z, type, val);
  if (bv->n != z->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension bv %lld, v %lld", (long long)bv->n, (long long)z->n);
  ierr = BV_IPMatMult(bv, z);CHKERRQ(ierr);
  ierr = VecDotBegin(bv->Bx, z, val);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode BVNorm_End_Private(BV bv, Vec z, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  PetscScalar p;
  if (!bv->matrix) return VecNormEnd(z, type, val);
  ierr = VecDotEnd(bv->Bx, z, &p);CHKERRQ(ierr);
  ierr = BV_SafeSqrt(bv, p, val);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode BVNormVecBegin(BV bv, Vec v, NormType type, PetscReal *val) { BVCheckSizes(bv); if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector"); return BVNorm_Begin_Private(bv, v, type, val); }
PetscErrorCode BVNormVecEnd(BV bv, Vec v, NormType type, PetscReal *val) { BVCheckSizes(bv); if (!v || !val) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); return BVNorm_End_Private(bv, v, type, val); }

/* a column view is the same Vec object for different columns, so Begin and End of one
 * column meet on the same entry */
PetscErrorCode BVNormColumnBegin(BV bv, PetscInt j, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  Vec v;
  BVCheckSizes(bv);
  ierr = BVGetColumn(bv, j, &v);CHKERRQ(ierr);
  ierr = BVNorm_Begin_Private(bv, v, type, val);
  { const PetscErrorCode ierr2 = BVRestoreColumn(bv, j, &v); if (!ierr) ierr = ierr2; }
  CHKERRQ(ierr);
  return 0;
}
PetscErrorCode BVNormColumnEnd(BV bv, PetscInt j, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  Vec v;
  BVCheckSizes(bv);
  if (!val) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ierr = BVGetColumn(bv, j, &v);CHKERRQ(ierr);
  ierr = BVNorm_End_Private(bv, v, type, val);
  { const PetscErrorCode ierr2 = BVRestoreColumn(bv, j, &v); if (!ierr) ierr = ierr2; }
  CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVDotColumn(BV X, PetscInt j, PetscScalar *q) {
  PetscErrorCode ierr;
  PetscInt ksave;
  Vec y;
  BVCheckSizes(X);
  if (j < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  if (j >= X->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, X->m);
  ksave = X->k;
  X->k = j;
  if (!q) { ierr = BVGetBuffer(X, &q, NULL);CHKERRQ(ierr); }
  ierr = BVGetColumn(X, j, &y);CHKERRQ(ierr);
  ierr = BV_DotVecOp(X, y, q);
  X->k = ksave;
  CHKERRQ(ierr);
  ierr = BVRestoreColumn(X, j, &y);CHKERRQ(ierr);
  return 0;
}

/* bvorthog.c:176-192: as BVDotColumn but including column j itself */
static PetscErrorCode BVDotColumnInc(BV X, PetscInt j, PetscScalar *q) {
  PetscErrorCode ierr;
  PetscInt ksave;
  Vec y;
  ksave = X->k;
  X->k = j + 1;
  ierr = BVGetColumn(X, j, &y);CHKERRQ(ierr);
  ierr = BV_DotVecOp(X, y, q);
  X->k = ksave;
  CHKERRQ(ierr);
  ierr = BVRestoreColumn(X, j, &y);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVScale(BV bv, PetscScalar alpha) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (!bv->n || (bv->l == bv->k && !bv->nc) || alpha == 1.0) return 0;
  ierr = bv->ops.scale(bv, -1, alpha);CHKERRQ(ierr);
  bv->state++;
  return 0;
}

PetscErrorCode BVScaleColumn(BV bv, PetscInt j, PetscScalar alpha) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (j < 0 || j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  if (!bv->n || alpha == 1.0) return 0;
  ierr = bv->ops.scale(bv, j, alpha);CHKERRQ(ierr);
  bv->state++;
  return 0;
}

/* bvglobal.c:470-480 */
static PetscErrorCode BVNorm_Private(BV bv, Vec z, PetscReal *val) {
  PetscErrorCode ierr;
  PetscScalar p;
  ierr = BV_IPMatMult(bv, z);CHKERRQ(ierr);
  ierr = VecDot(bv->Bx, z, &p);CHKERRQ(ierr);
  ierr = BV_SafeSqrt(bv, p, val);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVNorm(BV bv, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (!val) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (type == NORM_2) SETERRQ(PETSC_ERR_SUP, "Requested norm not available");
  if (bv->matrix) SETERRQ(PETSC_ERR_SUP, "Matrix norm not available for non-standard inner product");
  ierr = bv->ops.norm(bv, -1, type, val);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVNormColumn(BV bv, PetscInt j, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (!val) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (j < 0 || j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  if (bv->matrix) {           /* sqrt(z^T B z), the type argument is ignored (bvglobal.c:722-725) */
    Vec z;
    ierr = BVGetColumn(bv, j, &z);CHKERRQ(ierr);
    ierr = BVNorm_Private(bv, z, val);
    { const PetscErrorCode ierr2 = BVRestoreColumn(bv, j, &z); if (!ierr) ierr = ierr2; }
    CHKERRQ(ierr);
  } else { ierr = bv->ops.norm(bv, j, type, val);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode BVNormVec(BV bv, Vec v, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  BVCheckSizes(bv);
  if (!v || !val) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (bv->n != v->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mismatching local dimension bv %lld, v %lld", (long long)bv->n, (long long)v->n);
  if (bv->matrix) { ierr = BVNorm_Private(bv, v, val);CHKERRQ(ierr); }
  else { ierr = VecNorm(v, type, val);CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode BVMatMult(BV V, Mat A, BV W) {
  PetscErrorCode ierr;
  BVCheckSizes(V); BVCheckSizes(W);
  if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix");
  if (V->n != A->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mat argument has %lld columns, cannot multiply a BV with %lld rows", (long long)A->n, (long long)V->n);
  if (W->n != A->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Mat argument has %lld rows, cannot store the result in a BV with %lld rows", (long long)A->n, (long long)W->n);
  if (V->k - V->l > W->m - W->l) SETERRQ(PETSC_ERR_ARG_SIZ, "W has %d non-leading columns, not enough to store %d columns", W->m - W->l, V->k - V->l);
  W->k = W->l + V->k - V->l;
  ierr = V->ops.matmult(V, A, W);CHKERRQ(ierr);
  W->state++;
  return 0;
}

/* ---- Gram-Schmidt (bvorthog.c) ----------------------------------------------------------------- */
/* coefficient arrays: h == NULL means "column j of the buffer", c == NULL means
 * "column 0 of the buffer" (bvorthog.c:30-156) */
static PetscScalar *HH(BV bv, PetscInt j, PetscScalar *h) { return h ? h : bv->buffer + (size_t)j * (bv->nc + bv->m); }
static PetscScalar *CC(BV bv, PetscScalar *c) { return c ? c : bv->buffer; }

static PetscErrorCode BV_NormVecOrColumn(BV bv, PetscInt j, Vec v, PetscReal *nrm) {
  PetscErrorCode ierr;
  if (v) { ierr = BVNormVec(bv, v, NORM_2, nrm);CHKERRQ(ierr); }
  else { ierr = BVNormColumn(bv, j, NORM_2, nrm);CHKERRQ(ierr); }
  return 0;
}

/* bvorthog.c:197-227 */
static PetscErrorCode BVOrthogonalizeMGS1(BV bv, PetscInt j, Vec v, const PetscBool *which, PetscScalar *h, PetscScalar *c, PetscReal *onrm, PetscReal *nrm) {
  PetscErrorCode ierr;
  PetscInt i;
  PetscScalar dot, *cc = CC(bv, c), *hh = HH(bv, j, h);
  Vec vi, z, w = v;
  if (!v) { ierr = BVGetColumn(bv, j, &w);CHKERRQ(ierr); }
  if (onrm) { ierr = BVNormVec(bv, w, NORM_2, onrm);CHKERRQ(ierr); }
  if (which) for (i = 0; i < bv->nc + j; i++) cc[i] = 0.0;       /* columns left out contribute nothing */
  for (i = -bv->nc; i < j; i++) {
    if (which && i >= 0 && !which[i]) continue;
    ierr = BVGetColumn(bv, i, &vi);CHKERRQ(ierr);
    z = w;
    if (bv->matrix) { ierr = BV_IPMatMult(bv, w);CHKERRQ(ierr); z = bv->Bx; }
    ierr = VecDot(z, vi, &dot);CHKERRQ(ierr);
    cc[bv->nc + i] = dot;
    if (bv->indef) dot /= bv->omega[bv->nc + i];
    ierr = VecAXPY(w, -dot, vi);CHKERRQ(ierr);
    ierr = BVRestoreColumn(bv, i, &vi);CHKERRQ(ierr);
  }
  if (nrm) { ierr = BVNormVec(bv, w, NORM_2, nrm);CHKERRQ(ierr); }
  if (!v) { ierr = BVRestoreColumn(bv, j, &w);CHKERRQ(ierr); }
  for (i = 0; i < bv->nc + j; i++) hh[i] += cc[i];
  ks_gs_passes++;
  return 0;
}

/* bvorthog.c:233-277: |v'| (estimated), |v| and one step of CGS with one global reduction */
static PetscErrorCode BVOrthogonalizeCGS1(BV bv, PetscInt j, Vec v, PetscScalar *h, PetscScalar *c, PetscReal *onorm, PetscReal *norm) {
  PetscErrorCode ierr;
  PetscReal sum, beta = 0.0;
  PetscScalar *cc = CC(bv, c), *hh = HH(bv, j, h);
  PetscInt i;
  bv->k = j;
  if (onorm || norm) {
    if (!v) {
      ierr = BVDotColumnInc(bv, j, c ? c : bv->buffer);CHKERRQ(ierr);
      ierr = BV_SafeSqrt(bv, cc[bv->nc + j], &beta);CHKERRQ(ierr);
    } else {
      ierr = BVDotVec(bv, v, cc);CHKERRQ(ierr);
      ierr = BVNormVec(bv, v, NORM_2, &beta);CHKERRQ(ierr);
    }
  } else {
    if (!v) { ierr = BVDotColumn(bv, j, cc);CHKERRQ(ierr); }
    else { ierr = BVDotVec(bv, v, cc);CHKERRQ(ierr); }
  }
  /* q = v - V h (h/omega with a signature: BV_ApplySignature, bvorthog.c:108-120) */
  if (bv->indef) for (i = 0; i < bv->nc + j; i++) cc[i] /= bv->omega[i];
  if (!v) { ierr = BVMultColumn(bv, -1.0, 1.0, j, cc);CHKERRQ(ierr); }
  else { ierr = BVMultVec(bv, -1.0, 1.0, v, cc);CHKERRQ(ierr); }
  if (bv->indef) for (i = 0; i < bv->nc + j; i++) cc[i] *= bv->omega[i];
  if (onorm) *onorm = beta;
  if (norm) {
    if (bv->indef) { ierr = BV_NormVecOrColumn(bv, j, v, norm);CHKERRQ(ierr); }
    else {
      sum = 0.0;
      for (i = 0; i < bv->nc + j; i++) sum += cc[i] * cc[i];      /* BV_SquareSum */
      *norm = beta * beta - sum;
      if (*norm <= 0.0) { ierr = BV_NormVecOrColumn(bv, j, v, norm);CHKERRQ(ierr); }
      else *norm = sqrt(*norm);
    }
  }
  for (i = 0; i < bv->nc + j; i++) hh[i] += cc[i];               /* BV_AddCoefficients */
  ks_gs_passes++;
  return 0;
}

/* bvorthog.c:290-361 */
static PetscErrorCode BVOrthogonalizeGS(BV bv, PetscInt j, Vec v, const PetscBool *which, PetscReal *norm, PetscBool *lindep) {
  PetscErrorCode ierr;
  PetscScalar *h, *c, *hh;
  PetscReal onrm = 0.0, nrm = 0.0;
  PetscInt k, l, i;
  const PetscBool mgs = (bv->orthog_type == BV_ORTHOG_MGS);
  /* with an indefinite inner product lindep is not computed, and the norm of a column always is */
  const PetscBool dolindep = (!bv->indef && lindep) ? PETSC_TRUE : PETSC_FALSE;
  const PetscBool signature = (bv->indef && !v) ? PETSC_TRUE : PETSC_FALSE;
  if (bv->indef && lindep) *lindep = PETSC_FALSE;
  if (v) { k = bv->k; h = bv->h; c = bv->c; }
  else { k = j; h = NULL; c = NULL; }
#define GS1(on, nr) (mgs ? BVOrthogonalizeMGS1(bv, k, v, which, h, c, on, nr) : BVOrthogonalizeCGS1(bv, k, v, h, c, on, nr))
  hh = HH(bv, k, h);
  for (i = 0; i < bv->nc + k; i++) hh[i] = 0.0;                   /* BV_CleanCoefficients */
  switch (bv->orthog_ref) {
    case BV_ORTHOG_REFINE_IFNEEDED:
      ierr = GS1(&onrm, &nrm);CHKERRQ(ierr);
      l = 1;
      while (l < 3 && nrm && nrm < bv->orthog_eta * onrm) {
        l++;
        if (mgs) onrm = nrm;
        ierr = GS1(mgs ? NULL : &onrm, &nrm);CHKERRQ(ierr);
      }
      if (dolindep) *lindep = !(nrm && nrm >= bv->orthog_eta * onrm);
      break;
    case BV_ORTHOG_REFINE_NEVER:
      ierr = GS1(NULL, NULL);CHKERRQ(ierr);
      if (norm || dolindep || signature) { ierr = BV_NormVecOrColumn(bv, k, v, &nrm);CHKERRQ(ierr); }
      if (dolindep) *lindep = !nrm;
      break;
    case BV_ORTHOG_REFINE_ALWAYS:
      ierr = GS1(NULL, NULL);CHKERRQ(ierr);
      ierr = GS1(dolindep ? &onrm : NULL, (norm || dolindep || signature) ? &nrm : NULL);CHKERRQ(ierr);
      if (dolindep) *lindep = !(nrm && nrm >= bv->orthog_eta * onrm);
      break;
  }
#undef GS1
  if (signature) bv->omega[bv->nc + k] = (nrm < 0.0) ? -1.0 : 1.0;
  if (norm) {
    *norm = nrm;
    if (!v) hh[bv->nc + k] = (dolindep && *lindep) ? 0.0 : nrm;    /* store the norm next to the coefficients */
  }
  return 0;
}

static PetscErrorCode BV_AllocateCoeffs(BV bv) {
  if (!bv->h) {
    bv->h = (PetscScalar *)calloc((size_t)2 * (bv->nc + bv->m), sizeof(PetscScalar));
    if (!bv->h) SETERRQ(PETSC_ERR_MEM, "Out of memory");
    bv->c = bv->h + bv->nc + bv->m;
  }
  return 0;
}

/* bvorthog.c:391-418 */
PetscErrorCode BVOrthogonalizeVec(BV bv, Vec v, PetscScalar *H, PetscReal *norm, PetscBool *lindep) {
  PetscErrorCode ierr;
  PetscInt i, ksave, lsave;
  BVCheckSizes(bv);
  if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  ksave = bv->k; lsave = bv->l;
  bv->l = -bv->nc;
  ierr = BV_AllocateCoeffs(bv);CHKERRQ(ierr);
  ierr = BVGetBuffer(bv, NULL, NULL);CHKERRQ(ierr);
  ierr = BV_AllocateSignature(bv);
  if (!ierr) ierr = BVOrthogonalizeGS(bv, 0, v, NULL, norm, lindep);
  bv->k = ksave; bv->l = lsave;
  CHKERRQ(ierr);
  if (H) for (i = bv->l; i < bv->k; i++) H[i - bv->l] = bv->h[bv->nc + i];     /* BV_StoreCoefficients */
  return 0;
}

/* bvorthog.c:460-485 */
PetscErrorCode BVOrthogonalizeColumn(BV bv, PetscInt j, PetscScalar *H, PetscReal *norm, PetscBool *lindep) {
  PetscErrorCode ierr;
  PetscInt i, ksave, lsave;
  BVCheckSizes(bv);
  if (j < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  if (j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, bv->m);
  ksave = bv->k; lsave = bv->l;
  bv->l = -bv->nc;
  ierr = BVGetBuffer(bv, NULL, NULL);CHKERRQ(ierr);
  ierr = BV_AllocateSignature(bv);
  if (!ierr) ierr = BVOrthogonalizeGS(bv, j, NULL, NULL, norm, lindep);
  bv->k = ksave; bv->l = lsave;
  CHKERRQ(ierr);
  if (H) { PetscScalar *hh = HH(bv, j, NULL); for (i = bv->l; i < j; i++) H[i - bv->l] = hh[bv->nc + i]; }
  bv->state++;
  return 0;
}

/* bvorthog.c:587-633: modified Gram-Schmidt against the columns i < j with which[i] set */
PetscErrorCode BVOrthogonalizeSomeColumn(BV bv, PetscInt j, const PetscBool *which, PetscScalar *H, PetscReal *norm, PetscBool *lindep) {
  PetscErrorCode ierr;
  PetscInt i, ksave, lsave;
  BVCheckSizes(bv);
  if (!which) SETERRQ(PETSC_ERR_ARG_NULL, "Null array");
  if (j < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  if (j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, bv->m);
  if (bv->orthog_type != BV_ORTHOG_MGS) SETERRQ(PETSC_ERR_SUP, "Operation only available for MGS orthogonalization");
  ksave = bv->k; lsave = bv->l;
  bv->l = -bv->nc;
  ierr = BVGetBuffer(bv, NULL, NULL);CHKERRQ(ierr);
  ierr = BV_AllocateSignature(bv);
  if (!ierr) ierr = BVOrthogonalizeGS(bv, j, NULL, which, norm, lindep);
  bv->k = ksave; bv->l = lsave;
  CHKERRQ(ierr);
  if (H) { PetscScalar *hh = HH(bv, j, NULL); for (i = bv->l; i < j; i++) H[i - bv->l] = hh[bv->nc + i]; }
  bv->state++;
  return 0;
}

/* bvorthog.c:523-572 */
PetscErrorCode BVOrthonormalizeColumn(BV bv, PetscInt j, PetscBool replace, PetscReal *norm, PetscBool *lindep) {
  PetscErrorCode ierr;
  PetscReal nrm;
  PetscInt ksave, lsave;
  PetscBool lndep;
  BVCheckSizes(bv);
  if (j < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  if (j >= bv->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, bv->m);
  ksave = bv->k; lsave = bv->l;
  bv->l = -bv->nc;
  ierr = BVGetBuffer(bv, NULL, NULL);CHKERRQ(ierr);
  ierr = BV_AllocateSignature(bv);
  if (!ierr) ierr = BVOrthogonalizeGS(bv, j, NULL, NULL, &nrm, &lndep);
  if (!ierr && replace && (nrm == 0.0 || lndep)) {
    ierr = BVSetRandomColumn(bv, j);
    if (!ierr) ierr = BVOrthogonalizeGS(bv, j, NULL, NULL, &nrm, &lndep);
    if (!ierr && (nrm == 0.0 || lndep)) {
      ierr = BVSetRandomColumn(bv, j);
      if (!ierr) ierr = BVOrthogonalizeGS(bv, j, NULL, NULL, &nrm, &lndep);
    }
  }
  bv->k = ksave; bv->l = lsave;
  CHKERRQ(ierr);
  if (nrm != 1.0 && nrm != 0.0 && bv->n) { ierr = bv->ops.scale(bv, j, 1.0 / nrm);CHKERRQ(ierr); }
  if (norm) *norm = nrm;
  if (lindep) *lindep = lndep;
  bv->state++;
  return 0;
}

PetscErrorCode BVDestroy(BV *bv) {
  PetscErrorCode ierr;
  if (!bv || !*bv) return 0;
  BV b = *bv;
  if (b->data && b->ops.destroy) { ierr = b->ops.destroy(b);CHKERRQ(ierr); }
  ierr = VecDestroy(&b->t);CHKERRQ(ierr);
  ierr = VecDestroy(&b->cv[0]);CHKERRQ(ierr);
  ierr = VecDestroy(&b->cv[1]);CHKERRQ(ierr);
  ierr = VecDestroy(&b->Bx);CHKERRQ(ierr);
  ierr = BVDestroy(&b->cached);CHKERRQ(ierr);
  free(b->buffer);
  free(b->h);
  free(b->omega);
  free(b);
  *bv = NULL;
  return 0;
}

/* ---- block orthogonalisation (bvorthog.c:635-859) ------------------------------------------- */
/* bvorthog.c:635-680 */
static PetscErrorCode BVOrthogonalize_GS(BV V, PetscScalar *r, PetscInt ldr) {
  PetscErrorCode ierr;
  PetscReal norm;
  Vec v, w;
  if (V->matrix) {
    ierr = BVGetCachedBV(V, &V->cached);CHKERRQ(ierr);
    ierr = BVSetActiveColumns(V->cached, V->l, V->k);CHKERRQ(ierr);
  }
  for (PetscInt j = V->l; j < V->k; j++) {
    if (V->matrix && V->orthog_type == BV_ORTHOG_MGS) {      /* fill cached BV */
      ierr = BVGetColumn(V->cached, j, &v);CHKERRQ(ierr);
      ierr = BVGetColumn(V, j, &w);CHKERRQ(ierr);
      ierr = MatMult(V->matrix, w, v);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V, j, &w);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V->cached, j, &v);CHKERRQ(ierr);
    }
    if (r) {
      ierr = BVOrthogonalizeColumn(V, j, r + (size_t)j * ldr + V->l, &norm, NULL);CHKERRQ(ierr);
      r[j + (size_t)j * ldr] = norm;
    } else { ierr = BVOrthogonalizeColumn(V, j, NULL, &norm, NULL);CHKERRQ(ierr); }
    if (!norm) SETERRQ(1, "Breakdown in BVOrthogonalize due to a linearly dependent column");
    if (V->matrix && V->orthog_type == BV_ORTHOG_CGS) {      /* fill cached BV with the product the last pass left */
      ierr = BVGetColumn(V->cached, j, &v);CHKERRQ(ierr);
      ierr = VecCopy(V->Bx, v);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V->cached, j, &v);CHKERRQ(ierr);
    }
    ierr = BVScaleColumn(V, j, 1.0 / norm);CHKERRQ(ierr);
  }
  return 0;
}

/* bvorthog.c:683-765: R = chol(V'V) on the block [l,k), S = inv(R), V <- V S */
static PetscErrorCode BVOrthogonalize_Chol(BV V, PetscScalar *Rin, PetscInt ldrin) {
  PetscErrorCode ierr = 0;
  const PetscInt m = V->k, l = V->l, n = m - l;
  PetscInt i, j, ld = Rin ? ldrin : m;
  PetscScalar *R = Rin, *S = NULL;
  if (!Rin) { R = (PetscScalar *)calloc((size_t)m * m, sizeof(*R)); if (!R) SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  S = (PetscScalar *)calloc((size_t)ld * m, sizeof(*S));
  if (!S) { if (!Rin) free(R); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  ierr = BVDot(V, V, R, ld);
  if (!ierr) {
    /* keep a copy, factor, retry once on a slightly shifted matrix as the reference does */
    for (j = l; j < m; j++) memcpy(S + (size_t)j * ld + l, R + (size_t)j * ld + l, (size_t)n * sizeof(PetscScalar));
    int info = ksd_potrf_upper(n, R + (size_t)l * ld + l, ld);
    if (info) {
      for (j = l; j < m; j++) { memcpy(R + (size_t)j * ld + l, S + (size_t)j * ld + l, (size_t)n * sizeof(PetscScalar)); R[j + (size_t)j * ld] += 50.0 * PETSC_MACHINE_EPSILON; }
      info = ksd_potrf_upper(n, R + (size_t)l * ld + l, ld);
      if (info) ierr = KSError(__func__, __LINE__, 1, "Error in Cholesky factorization, info=%d", info);
    }
  }
  if (!ierr) {
    memset(S, 0, (size_t)ld * m * sizeof(PetscScalar));
    for (j = l; j < m; j++) memcpy(S + (size_t)j * ld + l, R + (size_t)j * ld + l, (size_t)n * sizeof(PetscScalar));
    const int info = ksd_trtri_upper(n, S + (size_t)l * ld + l, ld);
    if (info) ierr = KSError(__func__, __LINE__, 1, "Error in the triangular inverse, info=%d", info);
  }
  if (!ierr) {
    /* zero out entries below the diagonal */
    for (j = l; j < m - 1; j++) for (i = j + 1; i < m; i++) { R[i + (size_t)j * ld] = 0.0; S[i + (size_t)j * ld] = 0.0; }
    ierr = BVMultInPlace(V, S, ld, l, m);
  }
  free(S);
  if (!Rin) free(R);
  CHKERRQ(ierr);
  return 0;
}

PetscErrorCode BVOrthogonalize(BV V, PetscScalar *R, PetscInt ldr) {
  PetscErrorCode ierr;
  BVCheckSizes(V);
  if (R) {
    if (V->l > 0 && V->orthog_block == BV_ORTHOG_BLOCK_GS) SETERRQ(PETSC_ERR_SUP, "Cannot request matrix R in Gram-Schmidt orthogonalization if l>0");
    if (ldr < V->k) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat size %d is smaller than the number of BV active columns %d", ldr, V->k);
  }
  if (V->nc) SETERRQ(PETSC_ERR_SUP, "Not implemented for BV with constraints, use BVOrthogonalizeColumn() instead");
  switch (V->orthog_block) {
    case BV_ORTHOG_BLOCK_GS: ierr = BVOrthogonalize_GS(V, R, ldr);CHKERRQ(ierr); break;
    case BV_ORTHOG_BLOCK_CHOL: ierr = BVOrthogonalize_Chol(V, R, ldr);CHKERRQ(ierr); break;
    default:                                       /* bvorthog.c:851-854, 770-782 */
      if (V->matrix) SETERRQ(PETSC_ERR_SUP, "Orthogonalization method not available for non-standard inner product");
      if (!V->ops.orthogonalize_tsqr) SETERRQ(PETSC_ERR_SUP, "The tall-skinny QR block orthogonalization is not available for BV type %s", V->type_name);
      if (R) for (PetscInt j = 0; j < V->k; j++) memset(R + (size_t)j * ldr, 0, (size_t)V->k * sizeof(PetscScalar));
      ierr = V->ops.orthogonalize_tsqr(V, R, ldr);CHKERRQ(ierr);
      break;
  }
  V->state++;
  return 0;
}
