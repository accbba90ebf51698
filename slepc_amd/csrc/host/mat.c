/* Mat: the operator seen by the solver -- MatMult, MatCreateVecs and the sizes.
 * "aijhip" wraps a libksgpu CSR matrix; a shell matrix calls back into user code
 * (MatCreateShell in the reference's tests/test8.c:57-61). */
#include "ksimpl.h"

static PetscErrorCode MatMult_AIJHIP(Mat A, Vec x, Vec y) {
  CHKKSG(ksg_mat_mult((ksg_mat)A->data, x->array, y->array));
  return 0;
}
static PetscErrorCode MatCreateVecs_AIJHIP(Mat A, Vec *right, Vec *left) {
  PetscErrorCode ierr;
  if (right) { ierr = VecCreateHIP((ksg_ctx)A->ctx, A->n, A->N, A->row0, right);CHKERRQ(ierr); }
  if (left) { ierr = VecCreateHIP((ksg_ctx)A->ctx, A->n, A->N, A->row0, left);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode MatDestroy_AIJHIP(Mat A) { CHKKSG(ksg_mat_destroy((ksg_mat)A->data)); A->data = NULL; return 0; }

PetscErrorCode MatCreateAIJHIP(ksg_ctx ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0, const int64_t *rowptr,
                               const int32_t *colidx, const double *val, Mat *A) {
  if (!ctx || !A) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ksg_mat m = NULL;
  CHKKSG(ksg_mat_create_csr(ctx, n, N, row0, rowptr, colidx, val, &m));
  Mat a = (Mat)calloc(1, sizeof(*a));
  if (!a) { ksg_mat_destroy(m); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  a->ops.mult = MatMult_AIJHIP; a->ops.createvecs = MatCreateVecs_AIJHIP; a->ops.destroy = MatDestroy_AIJHIP;
  a->n = n; a->N = N; a->row0 = row0; a->data = m; a->ctx = ctx;
  *A = a;
  return 0;
}

PetscErrorCode MatCreateShell(void *ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0, void *user,
                              PetscErrorCode (*mult)(Mat, Vec, Vec), PetscErrorCode (*createvecs)(Mat, Vec *, Vec *), Mat *A) {
  if (!A || !mult) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (n < 0 || N < n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid sizes");
  Mat a = (Mat)calloc(1, sizeof(*a));
  if (!a) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  a->ops.mult = mult; a->ops.createvecs = createvecs; a->ops.destroy = NULL;
  a->n = n; a->N = N; a->row0 = row0; a->data = user; a->ctx = ctx;
  *A = a;
  return 0;
}

PetscErrorCode MatShellGetContext(Mat A, void **user) { if (!A || !user) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *user = A->data; return 0; }

PetscErrorCode MatMult(Mat A, Vec x, Vec y) {
  PetscErrorCode ierr;
  if (!A || !x || !y) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (x == y || x->array == y->array) SETERRQ(PETSC_ERR_ARG_WRONG, "x and y must be different vectors");
  if (x->n != A->n || y->n != A->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat local size %lld, vectors %lld and %lld", (long long)A->n, (long long)x->n, (long long)y->n);
  ierr = A->ops.mult(A, x, y);CHKERRQ(ierr);
  y->state++;
  return 0;
}

PetscErrorCode MatCreateVecs(Mat A, Vec *right, Vec *left) {
  PetscErrorCode ierr;
  if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix");
  if (!A->ops.createvecs) SETERRQ(PETSC_ERR_SUP, "This matrix type cannot create vectors");
  ierr = A->ops.createvecs(A, right, left);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatGetSize(Mat A, PetscInt64 *M, PetscInt64 *N) { if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix"); if (M) *M = A->N; if (N) *N = A->N; return 0; }
PetscErrorCode MatGetLocalSize(Mat A, PetscInt64 *m, PetscInt64 *n) { if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix"); if (m) *m = A->n; if (n) *n = A->n; return 0; }
PetscErrorCode MatDestroy(Mat *A) {
  PetscErrorCode ierr;
  if (!A || !*A) return 0;
  if ((*A)->ops.destroy) { ierr = (*A)->ops.destroy(*A);CHKERRQ(ierr); }
  free(*A);
  *A = NULL;
  return 0;
}
