/* BV type "hip": basis vectors resident in MI355X memory, every op of the table
 * mapped to a libksgpu entry point.  This is the constructor a drop-in registers
 * with BVRegister("hip", BVCreate_HIP); it plays the role BVCreate_Svec
 * (src/sys/classes/bv/impls/svec/svec.c:417-520) plays for the default type. */
#include "ksimpl.h"

#define GBV(bv) ((ksg_bv)(bv)->data)

/* the interface layer owns nc, l, k (svec reads them from the BV at every op): hand them on */
static PetscErrorCode Sync(BV bv) {
  CHKKSG(ksg_bv_set_num_constraints(GBV(bv), bv->nc));
  CHKKSG(ksg_bv_set_active(GBV(bv), bv->l, bv->k));
  return 0;
}

static PetscErrorCode BVMult_HIP(BV Y, PetscScalar alpha, PetscScalar beta, BV X, const PetscScalar *Q, PetscInt ldq) {
  PetscErrorCode ierr;
  ierr = Sync(Y);CHKERRQ(ierr);
  ierr = Sync(X);CHKERRQ(ierr);
  CHKKSG(ksg_bv_mult(GBV(Y), alpha, beta, GBV(X), Q, Q ? ldq : 1));
  return 0;
}
static PetscErrorCode BVMultVec_HIP(BV X, PetscScalar alpha, PetscScalar beta, Vec y, PetscScalar *q) {
  PetscErrorCode ierr;
  ierr = Sync(X);CHKERRQ(ierr);
  CHKKSG(ksg_bv_multvec(GBV(X), alpha, beta, y->array, q));
  return 0;
}
static PetscErrorCode BVMultInPlace_HIP(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e) {
  PetscErrorCode ierr;
  ierr = Sync(V);CHKERRQ(ierr);
  CHKKSG(ksg_bv_multinplace(GBV(V), Q, ldq, s, e, 0));
  return 0;
}
static PetscErrorCode BVMultInPlaceTranspose_HIP(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e) {
  PetscErrorCode ierr;
  ierr = Sync(V);CHKERRQ(ierr);
  CHKKSG(ksg_bv_multinplace(GBV(V), Q, ldq, s, e, 1));
  return 0;
}
static PetscErrorCode BVDot_HIP(BV X, BV Y, PetscScalar *M, PetscInt ldm) {
  PetscErrorCode ierr;
  ierr = Sync(X);CHKERRQ(ierr);
  ierr = Sync(Y);CHKERRQ(ierr);
  CHKKSG(ksg_bv_dot(GBV(X), GBV(Y), M, ldm));
  return 0;
}
static PetscErrorCode BVDotVec_HIP(BV X, Vec y, PetscScalar *q) {
  PetscErrorCode ierr;
  ierr = Sync(X);CHKERRQ(ierr);
  CHKKSG(ksg_bv_dotvec(GBV(X), y->array, q));
  return 0;
}
static PetscErrorCode BVDotVecPost_HIP(BV X, Vec y, PetscInt *offset) {
  PetscErrorCode ierr;
  ierr = Sync(X);CHKERRQ(ierr);
  CHKKSG(ksg_red_post_dotvec(GBV(X), y->array, offset));
  return 0;
}
static PetscErrorCode BVDotVec_Local_HIP(BV X, Vec y, PetscScalar *q) {
  PetscErrorCode ierr;
  ierr = Sync(X);CHKERRQ(ierr);
  CHKKSG(ksg_bv_dotvec_local(GBV(X), y->array, q));
  return 0;
}
static PetscErrorCode BVScale_HIP(BV bv, PetscInt j, PetscScalar alpha) {
  PetscErrorCode ierr;
  ierr = Sync(bv);CHKERRQ(ierr);
  CHKKSG(ksg_bv_scale(GBV(bv), j, alpha));
  return 0;
}
static PetscErrorCode BVNorm_HIP(BV bv, PetscInt j, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  ierr = Sync(bv);CHKERRQ(ierr);
  CHKKSG(ksg_bv_norm(GBV(bv), j, (int)type, val));
  return 0;
}
static PetscErrorCode BVNorm_Local_HIP(BV bv, PetscInt j, NormType type, PetscReal *val) {
  PetscErrorCode ierr;
  ierr = Sync(bv);CHKERRQ(ierr);
  CHKKSG(ksg_bv_norm_local(GBV(bv), j, (int)type, val));
  return 0;
}
static PetscErrorCode BVMatMult_HIP(BV V, Mat A, BV W) {
  PetscErrorCode ierr;
  PetscInt j;
  Vec v, w;
  if (A->ctx == V->ctx && MatGetDeviceHandle_Private(A) && A->n == A->cn) {
    /* device CSR operator: all active columns in one pass over the matrix */
    ierr = Sync(V);CHKERRQ(ierr);
    ierr = Sync(W);CHKERRQ(ierr);
    CHKKSG(ksg_bv_matmult(GBV(V), (ksg_mat)MatGetDeviceHandle_Private(A), GBV(W)));
    return 0;
  }
  if (A->ops.mult && A->ctx == V->ctx && V->data && W->data) {
    /* any other operator on the same context: column by column (svec.c:250-256) */
    for (j = 0; j < V->k - V->l; j++) {
      ierr = BVGetColumn(V, V->l + j, &v);CHKERRQ(ierr);
      ierr = BVGetColumn(W, W->l + j, &w);CHKERRQ(ierr);
      ierr = MatMult(A, v, w);CHKERRQ(ierr);
      ierr = BVRestoreColumn(V, V->l + j, &v);CHKERRQ(ierr);
      ierr = BVRestoreColumn(W, W->l + j, &w);CHKERRQ(ierr);
    }
    return 0;
  }
  SETERRQ(PETSC_ERR_SUP, "BVMatMult needs an operator created on the same context");
}
static PetscErrorCode BVOrthogonalizeTSQR_HIP(BV V, PetscScalar *R, PetscInt ldr) {
  PetscErrorCode ierr;
  ierr = Sync(V);CHKERRQ(ierr);
  CHKKSG(ksg_bv_tsqr(GBV(V), R, R ? ldr : 1));
  return 0;
}
static PetscErrorCode BVCopy_HIP(BV V, BV W) {
  PetscErrorCode ierr;
  ierr = Sync(V);CHKERRQ(ierr);
  ierr = Sync(W);CHKERRQ(ierr);
  CHKKSG(ksg_bv_copy(GBV(V), GBV(W)));
  return 0;
}
static PetscErrorCode BVResize_HIP(BV bv, PetscInt m, PetscBool copy) { CHKKSG(ksg_bv_resize(GBV(bv), m, copy)); return 0; }
static PetscErrorCode BVGetColumn_HIP(BV bv, PetscInt j, Vec *v) {
  PetscErrorCode ierr;
  const int l = (bv->ci[0] == -bv->nc - 1) ? 0 : 1;            /* BVAvailableVec */
  CHKKSG(ksg_bv_set_num_constraints(GBV(bv), bv->nc));
  double *p = ksg_bv_column(GBV(bv), j);
  if (!p) SETERRQ(PETSC_ERR_LIB, "libksgpu: %s", ksg_last_error());
  ierr = VecPlaceArray(bv->cv[l], p);CHKERRQ(ierr);             /* svec.c:322 */
  return 0;
}
static PetscErrorCode BVRestoreColumn_HIP(BV bv, PetscInt j, Vec *v) {
  PetscErrorCode ierr;
  const int l = (j == bv->ci[0]) ? 0 : 1;
  ierr = VecResetArray(bv->cv[l]);CHKERRQ(ierr);                /* svec.c:334 */
  return 0;
}
static PetscErrorCode BVDestroy_HIP(BV bv) { CHKKSG(ksg_bv_destroy(GBV(bv))); bv->data = NULL; return 0; }

PetscErrorCode BVCreate_HIP(BV bv) {
  PetscErrorCode ierr;
  ksg_bv g = NULL;
  if (!bv->t || !bv->t->ctx) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "The template vector has no device context");
  CHKKSG(ksg_bv_create((ksg_ctx)bv->t->ctx, bv->n, bv->N, bv->m, &g));
  bv->data = g;
  bv->ctx = bv->t->ctx;
  ierr = VecCreateWithOps(bv->t->ops, bv->t->ctx, bv->n, bv->N, bv->t->row0, PETSC_FALSE, &bv->cv[0]);CHKERRQ(ierr);
  ierr = VecCreateWithOps(bv->t->ops, bv->t->ctx, bv->n, bv->N, bv->t->row0, PETSC_FALSE, &bv->cv[1]);CHKERRQ(ierr);
  bv->ops.mult = BVMult_HIP;
  bv->ops.multvec = BVMultVec_HIP;
  bv->ops.multinplace = BVMultInPlace_HIP;
  bv->ops.multinplacetrans = BVMultInPlaceTranspose_HIP;
  bv->ops.dot = BVDot_HIP;
  bv->ops.dotvec = BVDotVec_HIP;
  bv->ops.dotvec_local = BVDotVec_Local_HIP;
  bv->ops.scale = BVScale_HIP;
  bv->ops.norm = BVNorm_HIP;
  bv->ops.norm_local = BVNorm_Local_HIP;
  bv->ops.matmult = BVMatMult_HIP;
  bv->ops.copy = BVCopy_HIP;
  bv->ops.resize = BVResize_HIP;
  bv->ops.getcolumn = BVGetColumn_HIP;
  bv->ops.restorecolumn = BVRestoreColumn_HIP;
  bv->ops.destroy = BVDestroy_HIP;
  bv->ops.dotvec_post = BVDotVecPost_HIP;
  bv->ops.orthogonalize_tsqr = BVOrthogonalizeTSQR_HIP;
  return 0;
}
