/* Vec: the few PETSc vector operations the hot path calls on BV column views and
 * work vectors, dispatched through an op table so that the storage type decides
 * the implementation (the "hip" type keeps its entries in device memory and maps
 * every operation to libksgpu). */
#include "ksimpl.h"

/* ---- "hip" type --------------------------------------------------------------- */
static PetscErrorCode VecAlloc_HIP(Vec v) { double *p = NULL; CHKKSG(ksg_vec_alloc((ksg_ctx)v->ctx, v->n, &p)); v->own = v->array = p; return 0; }
static PetscErrorCode VecFree_HIP(Vec v) { if (v->own) CHKKSG(ksg_vec_free((ksg_ctx)v->ctx, v->own)); v->own = NULL; return 0; }
static PetscErrorCode VecCopy_HIP(Vec x, Vec y) { CHKKSG(ksg_vec_copy((ksg_ctx)x->ctx, x->array, y->array, x->n)); return 0; }
static PetscErrorCode VecSet_HIP(Vec x, PetscScalar a) { CHKKSG(ksg_vec_set((ksg_ctx)x->ctx, x->array, a, x->n)); return 0; }
static PetscErrorCode VecScale_HIP(Vec x, PetscScalar a) { CHKKSG(ksg_vec_scale((ksg_ctx)x->ctx, x->array, a, x->n)); return 0; }
static PetscErrorCode VecAXPY_HIP(Vec y, PetscScalar a, Vec x) { CHKKSG(ksg_vec_axpy((ksg_ctx)y->ctx, y->array, a, x->array, y->n)); return 0; }
static PetscErrorCode VecDot_HIP(Vec x, Vec y, PetscScalar *r) { CHKKSG(ksg_vec_dot((ksg_ctx)x->ctx, x->array, y->array, x->n, r)); return 0; }
static PetscErrorCode VecNorm2_HIP(Vec x, PetscReal *r) { CHKKSG(ksg_vec_norm2((ksg_ctx)x->ctx, x->array, x->n, r)); return 0; }
static PetscErrorCode VecFillHash_HIP(Vec x, uint64_t seed) { CHKKSG(ksg_vec_fill_hash((ksg_ctx)x->ctx, x->array, x->n, x->row0, seed)); return 0; }
static PetscErrorCode VecToHost_HIP(Vec x, PetscScalar *h) { CHKKSG(ksg_vec_download((ksg_ctx)x->ctx, h, x->array, x->n)); return 0; }
static PetscErrorCode VecFromHost_HIP(Vec x, const PetscScalar *h) { CHKKSG(ksg_vec_upload((ksg_ctx)x->ctx, x->array, h, x->n)); return 0; }

static PetscErrorCode VecRedOpen_HIP(void *ctx) { CHKKSG(ksg_red_open((ksg_ctx)ctx)); return 0; }
static PetscErrorCode VecDotPost_HIP(Vec x, Vec y, PetscInt *offset) { CHKKSG(ksg_red_post_dot((ksg_ctx)x->ctx, x->array, y->array, x->n, offset)); return 0; }
static PetscErrorCode VecRedClose_HIP(void *ctx, PetscScalar *out, PetscInt count) { CHKKSG(ksg_red_close((ksg_ctx)ctx, out, count)); return 0; }

static const struct _VecOps VecOps_HIP = {
  VecAlloc_HIP, VecFree_HIP, VecCopy_HIP, VecSet_HIP, VecScale_HIP, VecAXPY_HIP, VecDot_HIP, VecNorm2_HIP,
  VecFillHash_HIP, VecToHost_HIP, VecFromHost_HIP, VecRedOpen_HIP, VecDotPost_HIP, VecRedClose_HIP
};

PetscErrorCode VecCreateHIP(ksg_ctx ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0, Vec *v) {
  if (!ctx) SETERRQ(PETSC_ERR_ARG_NULL, "Null libksgpu context");
  return VecCreateWithOps(&VecOps_HIP, ctx, n, N, row0, PETSC_TRUE, v);
}

/* ---- generic ---------------------------------------------------------------------- */
PetscErrorCode VecCreateWithOps(const struct _VecOps *ops, void *ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0,
                                PetscBool allocate, Vec *v) {
  PetscErrorCode ierr;
  if (!ops || !v) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (n < 0 || N < n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid sizes: local %lld, global %lld", (long long)n, (long long)N);
  Vec x = (Vec)calloc(1, sizeof(*x));
  if (!x) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  x->ops = ops; x->ctx = ctx; x->n = n; x->N = N; x->row0 = row0;
  if (allocate) { ierr = ops->alloc(x); if (ierr) { free(x); CHKERRQ(ierr); } }
  *v = x;
  return 0;
}

PetscErrorCode VecDuplicate(Vec v, Vec *w) {
  PetscErrorCode ierr;
  if (!v || !w) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ierr = VecCreateWithOps(v->ops, v->ctx, v->n, v->N, v->row0, PETSC_TRUE, w);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode VecDestroy(Vec *v) {
  PetscErrorCode ierr;
  if (!v || !*v) return 0;
  ierr = (*v)->ops->free(*v);CHKERRQ(ierr);
  free(*v);
  *v = NULL;
  return 0;
}

PetscErrorCode VecPlaceArray(Vec v, PetscScalar *array) {
  if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  v->array = array;
  v->state++;
  return 0;
}

PetscErrorCode VecResetArray(Vec v) {
  if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  v->array = v->own;
  return 0;
}

static PetscErrorCode SameLayout(Vec x, Vec y) {
  if (!x || !y) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (x->n != y->n || x->N != y->N) SETERRQ(PETSC_ERR_ARG_INCOMP, "Incompatible vector sizes: local %lld vs %lld", (long long)x->n, (long long)y->n);
  if (x->ops != y->ops) SETERRQ(PETSC_ERR_ARG_INCOMP, "Vectors of different types");
  if (!x->array || !y->array) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Vector has no storage");
  return 0;
}

PetscErrorCode VecCopy(Vec x, Vec y) {
  PetscErrorCode ierr;
  ierr = SameLayout(x, y);CHKERRQ(ierr);
  if (x->array == y->array) return 0;
  ierr = x->ops->copy(x, y);CHKERRQ(ierr);
  y->state++;
  return 0;
}
PetscErrorCode VecSet(Vec x, PetscScalar a) {
  PetscErrorCode ierr;
  if (!x) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  ierr = x->ops->set(x, a);CHKERRQ(ierr);
  x->state++;
  return 0;
}
PetscErrorCode VecScale(Vec x, PetscScalar a) {
  PetscErrorCode ierr;
  if (!x) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  if (a == 1.0) return 0;
  ierr = x->ops->scale(x, a);CHKERRQ(ierr);
  x->state++;
  return 0;
}
PetscErrorCode VecAXPY(Vec y, PetscScalar a, Vec x) {
  PetscErrorCode ierr;
  ierr = SameLayout(x, y);CHKERRQ(ierr);
  if (x == y) SETERRQ(PETSC_ERR_ARG_WRONG, "x and y cannot be the same vector");
  ierr = y->ops->axpy(y, a, x);CHKERRQ(ierr);
  y->state++;
  return 0;
}
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *r) {
  PetscErrorCode ierr;
  ierr = SameLayout(x, y);CHKERRQ(ierr);
  ierr = x->ops->dot(x, y, r);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *r) {
  PetscErrorCode ierr;
  if (!x || !r) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (type != NORM_2 && type != NORM_FROBENIUS) SETERRQ(PETSC_ERR_SUP, "Only the 2-norm is available for vectors");
  ierr = x->ops->norm2(x, r);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecGetSize(Vec x, PetscInt64 *N) { if (!x) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector"); *N = x->N; return 0; }
PetscErrorCode VecGetLocalSize(Vec x, PetscInt64 *n) { if (!x) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector"); *n = x->n; return 0; }
PetscErrorCode VecGetValuesHost(Vec x, PetscScalar *host) {
  PetscErrorCode ierr;
  if (!x || (x->n && !host)) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ierr = x->ops->tohost(x, host);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecSetValuesHost(Vec x, const PetscScalar *host) {
  PetscErrorCode ierr;
  if (!x || (x->n && !host)) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ierr = x->ops->fromhost(x, host);CHKERRQ(ierr);
  x->state++;
  return 0;
}
PetscErrorCode VecFillHash(Vec x, uint64_t seed) {
  PetscErrorCode ierr;
  if (!x) SETERRQ(PETSC_ERR_ARG_NULL, "Null vector");
  ierr = x->ops->fillhash(x, seed);CHKERRQ(ierr);
  x->state++;
  return 0;
}

/* ---- split-phase reductions -------------------------------------------------------------
 * One group at a time: the ...Begin calls queue their reductions in the back end (or compute
 * at once when the type has no post ops), the first ...End closes the group -- one reduction
 * round -- and hands every queued value to its destination. */
#define KS_RED_MAXENTRIES 64
static struct {
  const struct _VecOps *ops; void *ctx; int open, n; PetscInt total;
  struct { PetscScalar *dst; PetscInt count, offset; int root; } e[KS_RED_MAXENTRIES];
} ks_red;

/* opens the group on (ops, ctx) if need be; *batched = whether the back end queues */
PetscErrorCode KSRedJoin_Private(const struct _VecOps *ops, void *ctx, PetscBool *batched) {
  PetscErrorCode ierr;
  *batched = PETSC_FALSE;
  if (!ops || !ops->redopen || !ops->dotpost || !ops->redclose) return 0;
  if (ks_red.open && (ks_red.ops != ops || ks_red.ctx != ctx)) SETERRQ(PETSC_ERR_ORDER, "A group of split-phase reductions is open on another context");
  if (!ks_red.open) {
    ierr = ops->redopen(ctx);CHKERRQ(ierr);
    ks_red.ops = ops; ks_red.ctx = ctx; ks_red.open = 1; ks_red.n = 0; ks_red.total = 0;
  }
  if (ks_red.n >= KS_RED_MAXENTRIES) SETERRQ(PETSC_ERR_SUP, "Too many reductions in one group");
  *batched = PETSC_TRUE;
  return 0;
}

/* records where the `count` values posted at `offset` go (root: deliver the square root) */
void KSRedRecord_Private(PetscScalar *dst, PetscInt count, PetscInt offset, int root) {
  ks_red.e[ks_red.n].dst = dst; ks_red.e[ks_red.n].count = count; ks_red.e[ks_red.n].offset = offset; ks_red.e[ks_red.n].root = root;
  ks_red.n++;
  if (offset + count > ks_red.total) ks_red.total = offset + count;
}

/* drops the open group after a failed post, so that the next group starts clean */
void KSRedAbandon_Private(void) {
  if (!ks_red.open) return;
  ks_red.open = 0;
  (void)ks_red.ops->redclose(ks_red.ctx, NULL, -1);
}

PetscErrorCode KSRedComplete_Private(void) {
  PetscErrorCode ierr;
  if (!ks_red.open) return 0;
  ks_red.open = 0;
  PetscScalar *out = (PetscScalar *)malloc((size_t)PetscMax(ks_red.total, 1) * sizeof(*out));
  if (!out) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  ierr = ks_red.ops->redclose(ks_red.ctx, out, ks_red.total);
  if (!ierr) for (int i = 0; i < ks_red.n; i++)
    for (PetscInt j = 0; j < ks_red.e[i].count; j++) {
      const PetscScalar v = out[ks_red.e[i].offset + j];
      ks_red.e[i].dst[j] = ks_red.e[i].root ? sqrt(v) : v;
    }
  free(out);
  CHKERRQ(ierr);
  return 0;
}

PetscErrorCode VecDotBegin(Vec x, Vec y, PetscScalar *r) {
  PetscErrorCode ierr;
  PetscBool batched;
  PetscInt off;
  if (!x || !y || !r) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (x->n != y->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Incompatible vector local lengths");
  ierr = KSRedJoin_Private(x->ops, x->ctx, &batched);CHKERRQ(ierr);
  if (!batched) return VecDot(x, y, r);
  ierr = x->ops->dotpost(x, y, &off);
  if (ierr) { KSRedAbandon_Private();CHKERRQ(ierr); }
  KSRedRecord_Private(r, 1, off, 0);
  return 0;
}
PetscErrorCode VecDotEnd(Vec x, Vec y, PetscScalar *r) { (void)x; (void)y; (void)r; return KSRedComplete_Private(); }

PetscErrorCode VecNormBegin(Vec x, NormType type, PetscReal *r) {
  PetscErrorCode ierr;
  PetscBool batched;
  PetscInt off;
  if (!x || !r) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (type != NORM_2 && type != NORM_FROBENIUS) SETERRQ(PETSC_ERR_SUP, "Only the 2-norm is available in split phase");
  ierr = KSRedJoin_Private(x->ops, x->ctx, &batched);CHKERRQ(ierr);
  if (!batched) return VecNorm(x, NORM_2, r);
  ierr = x->ops->dotpost(x, x, &off);
  if (ierr) { KSRedAbandon_Private();CHKERRQ(ierr); }
  KSRedRecord_Private(r, 1, off, 1);
  return 0;
}
PetscErrorCode VecNormEnd(Vec x, NormType type, PetscReal *r) { (void)x; (void)type; (void)r; return KSRedComplete_Private(); }
