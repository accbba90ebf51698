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

/* Brackets for code outside the library that reads or writes the device arrays of "hip"
 * vectors on its own (the MatMult of a shell operator): before it, updates the library still
 * holds back are applied (the arrays are then current for work on the context's stream); after
 * it, the library is told that device data changed behind its back.  Other vector types: no-ops. */
PetscErrorCode VecForeignAccessBegin_Private(Vec x) {
  if (x && x->ops == &VecOps_HIP && x->ctx) CHKKSG(ksg_ctx_flush((ksg_ctx)x->ctx));
  return 0;
}
PetscErrorCode VecForeignAccessEnd_Private(Vec y) {
  if (y && y->ops == &VecOps_HIP && y->ctx) CHKKSG(ksg_ctx_touch((ksg_ctx)y->ctx));
  return 0;
}

/* ---- generic ---------------------------------------------------------------------- */
PetscErrorCode VecCreateWithOps(const struct _VecOps *ops, void *ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0,
                                PetscBool allocate, Vec *v) {
  PetscErrorCode ierr;
  if (!ops || !v) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (n < 0 || N < n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid sizes: local %lld, global %lld", (long long)n, (long long)N);
  Vec x = (Vec)calloc(1, sizeof(*x));
  if (!x) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  static PetscInt64 next_id = 0;
  x->ops = ops; x->ctx = ctx; x->n = n; x->N = N; x->row0 = row0; x->id = ++next_id;
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
 * The role of PetscSplitReduction (bvglobal.c:227-241, 276-290): one group at a time.  Every
 * ...Begin appends an entry; on a back end with the post ops its kernel is queued on the
 * device, otherwise the value is computed at once and parked here.  The first ...End
 * completes the group -- one reduction round and one copy for all queued kernels --, and each
 * End hands out the entry it is paired with, in the order of the Begins. */
#define KS_RED_MAXENTRIES 64
#define KS_RED_MAXVALUES  8192
static struct {
  const struct _VecOps *ops; void *ctx;            /* back end with a batch open (NULL: none) */
  int state;                                       /* 0: collecting Begins, 1: handing out at Ends */
  int n, nend; PetscInt nvals, nposted;
  struct { const void *owner; PetscInt count, off, dev_off; int root; } e[KS_RED_MAXENTRIES];
  PetscScalar vals[KS_RED_MAXVALUES];
} ks_red;

static void KSRedReset(void) { ks_red.ops = NULL; ks_red.ctx = NULL; ks_red.state = 0; ks_red.n = ks_red.nend = 0; ks_red.nvals = ks_red.nposted = 0; }

/* drops the group (after a failure), so that the next one starts clean */
void KSRedAbandon_Private(void) {
  if (ks_red.ops) (void)ks_red.ops->redclose(ks_red.ctx, NULL, -1);
  KSRedReset();
}

/* Appends an entry of `count` values for `owner`.  With the post ops of (ops, ctx) present the
 * batch of the back end is opened if need be and *vals = NULL: the caller posts its kernel and
 * reports the index it got with KSRedPosted_Private.  Otherwise *vals is where the caller
 * stores the values it computes now. */
PetscErrorCode KSRedBegin_Private(const void *owner, const struct _VecOps *ops, void *ctx, PetscInt count, int root, PetscScalar **vals) {
  PetscErrorCode ierr;
  const int batched = ops && ops->redopen && ops->dotpost && ops->redclose;
  if (ks_red.state != 0) SETERRQ(PETSC_ERR_ORDER, "Called before all VecxxxEnd() called");
  if (ks_red.n >= KS_RED_MAXENTRIES || ks_red.nvals + count > KS_RED_MAXVALUES) SETERRQ(PETSC_ERR_SUP, "Too many reductions in one group");
  if (batched) {
    if (ks_red.ops && (ks_red.ops != ops || ks_red.ctx != ctx)) SETERRQ(PETSC_ERR_ARG_INCOMP, "Split-phase reductions of one group must live on one context");
    if (!ks_red.ops) { ierr = ops->redopen(ctx);CHKERRQ(ierr); ks_red.ops = ops; ks_red.ctx = ctx; }
  }
  ks_red.e[ks_red.n].owner = owner; ks_red.e[ks_red.n].count = count; ks_red.e[ks_red.n].off = ks_red.nvals;
  ks_red.e[ks_red.n].dev_off = -1; ks_red.e[ks_red.n].root = root;
  *vals = batched ? NULL : ks_red.vals + ks_red.nvals;
  ks_red.nvals += count;
  ks_red.n++;
  return 0;
}

void KSRedPosted_Private(PetscInt dev_off) {
  ks_red.e[ks_red.n - 1].dev_off = dev_off;
  if (dev_off + ks_red.e[ks_red.n - 1].count > ks_red.nposted) ks_red.nposted = dev_off + ks_red.e[ks_red.n - 1].count;
}

/* Hands the next entry to its End: `owner` and `count` must be the ones of the matching Begin */
PetscErrorCode KSRedEnd_Private(const void *owner, PetscInt count, PetscScalar *out) {
  PetscErrorCode ierr;
  if (ks_red.state == 0) {                       /* first End of the group: complete it */
    if (ks_red.ops) {
      PetscScalar *dev = (PetscScalar *)malloc((size_t)PetscMax(ks_red.nposted, 1) * sizeof(*dev));
      if (!dev) { KSRedAbandon_Private(); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
      ierr = ks_red.ops->redclose(ks_red.ctx, dev, ks_red.nposted);
      if (ierr) { free(dev); KSRedAbandon_Private();CHKERRQ(ierr); }
      for (int i = 0; i < ks_red.n; i++)
        if (ks_red.e[i].dev_off >= 0) memcpy(ks_red.vals + ks_red.e[i].off, dev + ks_red.e[i].dev_off, (size_t)ks_red.e[i].count * sizeof(*dev));
      free(dev);
      ks_red.ops = NULL; ks_red.ctx = NULL;
    }
    ks_red.state = 1;
  }
  if (ks_red.nend >= ks_red.n) { KSRedReset(); SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Called VecxxxEnd() more times than VecxxxBegin()"); }
  if (ks_red.e[ks_red.nend].owner != owner || ks_red.e[ks_red.nend].count != count) {
    KSRedReset();
    SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Called xxxEnd() in a different order or with a different object than xxxBegin()");
  }
  for (PetscInt j = 0; j < count; j++) {
    const PetscScalar v = ks_red.vals[ks_red.e[ks_red.nend].off + j];
    out[j] = ks_red.e[ks_red.nend].root ? sqrt(v) : v;
  }
  if (++ks_red.nend == ks_red.n) KSRedReset();    /* all handed out: no outstanding requests */
  return 0;
}

static PetscErrorCode VecDotBegin_Private(Vec x, Vec y, int root) {
  PetscErrorCode ierr;
  PetscScalar *vals;
  PetscInt off;
  ierr = KSRedBegin_Private(x, x->ops, x->ctx, 1, root, &vals);CHKERRQ(ierr);
  if (vals) ierr = VecDot(x, y, vals);
  else {
    ierr = x->ops->dotpost(x, y, &off);
    if (!ierr) KSRedPosted_Private(off);
  }
  if (ierr) { KSRedAbandon_Private();CHKERRQ(ierr); }
  return 0;
}

PetscErrorCode VecDotBegin(Vec x, Vec y, PetscScalar *r) {
  (void)r;
  if (!x || !y) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (x->n != y->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Incompatible vector local lengths");
  return VecDotBegin_Private(x, y, 0);
}
PetscErrorCode VecDotEnd(Vec x, Vec y, PetscScalar *r) {
  (void)y;
  if (!x || !r) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  return KSRedEnd_Private(x, 1, r);
}

/* the sum of squares travels with the group; the root is taken when the value is handed out */
PetscErrorCode VecNormBegin(Vec x, NormType type, PetscReal *r) {
  (void)r;
  if (!x) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (type != NORM_2 && type != NORM_FROBENIUS) SETERRQ(PETSC_ERR_SUP, "Only the 2-norm is available in split phase");
  return VecDotBegin_Private(x, x, 1);
}
PetscErrorCode VecNormEnd(Vec x, NormType type, PetscReal *r) {
  (void)type;
  if (!x || !r) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  return KSRedEnd_Private(x, 1, r);
}
