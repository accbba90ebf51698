/*
 * PETSc side of the drop-in: Vec type "hip", the device vector behind the BV type of
 * bv_hip_petsc.c and the operator of mat_aijhip_petsc.c, for a PETSc 3.7 / SLEPc 3.7 build.
 *
 * NOT compiled in this repository (PETSc is not available here).  The role is the one VECCUDA
 * plays for the reference's GPU path: BVCreate_Svec makes its two column views with
 * VecCreate{Seq,MPI}CUDAWithArray(...,NULL,...) (src/sys/classes/bv/impls/svec/svec.c:458-476)
 * and BVGetColumn_Svec_CUDA re-points them with VecCUDAPlaceArray
 * (src/sys/classes/bv/impls/svec/sveccuda/sveccuda.cu:500-525).
 *
 * What the hot-path host code calls on column views and work vectors, and where it lands:
 *   VecCopy      bvbasic.c:1608,1645; bvfunc.c:300,362     -> ksg_vec_copy
 *   VecAXPY      epssolve.c:531; bvorthog.c:220 (MGS)      -> ksg_vec_axpy
 *   VecScale     epsdefault.c:374                          -> ksg_vec_scale
 *   VecSet       epsdefault.c:398                          -> ksg_vec_set
 *   VecDot       bvglobal.c:477                            -> ksg_vec_dot   (collective, RCCL)
 *   VecNorm      epssolve.c:533; bvglobal.c:593            -> ksg_vec_norm2 (collective, RCCL)
 *   VecSetRandom bvops.c:399-409                           -> ksg_vec_fill_hash
 *   VecDuplicate bvbasic.c:1387 (BVCreateVec)              -> a new vector of this type
 *   VecGetArray / VecGetArrayRead / VecRestoreArray (host access: bv->buffer is a sequential
 *                vector of the template's type read on the host every Gram-Schmidt pass,
 *                bvbasic.c:818-823, bvorthog.c:38,58,77,97,132) -> a host mirror with a flag saying
 *                which copy is current, as VECCUDA keeps (valid_GPU_array): storage on either
 *                side is allocated at first use, so a vector only ever touched through
 *                VecGetArray -- the buffer -- never leaves the host
 *   VecPlaceArray / VecResetArray (svec.c:322,334)         -> swap the device pointer
 *   VecDotBegin/End, VecNormBegin/End (epskrylov.c:107-129, trlanczos.c:201-217)
 *                -> ops->dot_local / norm_local feed PETSc's own PetscSplitReduction
 *
 * The type is not "petscnative": VecGetArray and friends go through ops->getarray /
 * ops->restorearray (vec/vec/interface/rvector.c), which is where the host mirror lives.
 *
 * libksgpu may hold back an update of a basis column until the next call decides how to
 * apply it (include/ksgpu.h, KSG_OPT_LAZY_UPDATE) and keeps reductions it formed ahead of
 * their request (KSG_OPT_FUSE_DOTS).  Everything that reaches a device pointer through this
 * file is bracketed accordingly:
 *   - the ops below hand the pointer straight back to libksgpu, which looks after itself;
 *   - VecHIPGetArrayRead / VecHIPGetArray / VecHIPGetArrayWrite, the PUBLIC accessors for code
 *     outside libksgpu that launches its own kernels (a foreign MatMult), call
 *     ksg_ctx_flush() before handing the pointer out, and the Restore of a writing accessor
 *     calls ksg_ctx_touch();
 *   - VecHIPGetDevicePointer_Private, used by bv_hip_petsc.c and mat_aijhip_petsc.c for
 *     pointers that go straight into a ksg_* call, does neither.
 *
 * Register with
 *     VecRegister("hip", VecCreate_HIP);
 * after PetscInitialize; "aijhip" matrices create their vectors with this type, so user
 * code needs no -vec_type option.
 */
#include <petsc/private/vecimpl.h>
#include "ksgpu.h"

#define VECHIP "hip"
#define KSG(call) do { if ((call)) SETERRQ1(PETSC_COMM_SELF,PETSC_ERR_LIB,"libksgpu: %s",ksg_last_error()); } while (0)

/* which copy of the entries is current */
typedef enum { VECHIP_UNALLOCATED = 0, VECHIP_HOST, VECHIP_DEVICE, VECHIP_BOTH } VecHIPValid;

typedef struct {
  ksg_ctx     ctx;
  PetscScalar *d_own;       /* device storage this vector owns (allocated at first device use) */
  PetscScalar *d;           /* the device array in use: d_own or a placed one                  */
  PetscScalar *d_unplaced;  /* what VecResetArray goes back to                                  */
  PetscBool   placed,noown; /* an array is placed; created WithArray: never allocates its own   */
  PetscScalar *h;           /* host mirror (allocated at the first VecGetArray)                 */
  VecHIPValid valid,valid_unplaced;
  PetscBool   h_out;        /* VecGetArray is outstanding                                       */
} Vec_HIP;

/* ---- one libksgpu context per MPI communicator and rank, shared by Vec, BV and Mat ---- */
static PetscMPIInt KSG_ctx_keyval = MPI_KEYVAL_INVALID;

static PetscMPIInt KSGCtxDelete(MPI_Comm comm,PetscMPIInt keyval,void *attr,void *extra)
{
  ksg_ctx_destroy((ksg_ctx)attr);
  return MPI_SUCCESS;
}

PETSC_EXTERN PetscErrorCode KSGGetContext(MPI_Comm comm,ksg_ctx *ctx)
{
  PetscErrorCode ierr;
  PetscMPIInt    flg,rank,size,lrank;
  MPI_Comm       local;
  void           *attr;
  ksg_ctx        c;
  char           id[128];

  PetscFunctionBegin;
  if (KSG_ctx_keyval == MPI_KEYVAL_INVALID) {
    ierr = MPI_Comm_create_keyval(MPI_COMM_NULL_COPY_FN,KSGCtxDelete,&KSG_ctx_keyval,NULL);CHKERRQ(ierr);
  }
  ierr = MPI_Comm_get_attr(comm,KSG_ctx_keyval,&attr,&flg);CHKERRQ(ierr);
  if (flg) { *ctx = (ksg_ctx)attr; PetscFunctionReturn(0); }
  ierr = MPI_Comm_rank(comm,&rank);CHKERRQ(ierr);
  ierr = MPI_Comm_size(comm,&size);CHKERRQ(ierr);
  /* one GPU per rank of a node: the rank within the shared-memory node picks the device.  The
     split is collective over the communicator it is made from, so it is made from `comm`, whose
     ranks are the ones that call this together (a split of the world communicator would wait for
     ranks outside a sub-communicator for ever); ranks that share a node under `comm` get
     consecutive devices. */
  ierr = MPI_Comm_split_type(comm,MPI_COMM_TYPE_SHARED,0,MPI_INFO_NULL,&local);CHKERRQ(ierr);
  ierr = MPI_Comm_rank(local,&lrank);CHKERRQ(ierr);
  ierr = MPI_Comm_free(&local);CHKERRQ(ierr);
  KSG(ksg_ctx_create((int)lrank,&c));
  if (size > 1) {                 /* RCCL communicator over the ranks of comm (collective) */
    if (!rank) KSG(ksg_comm_unique_id(id));
    ierr = MPI_Bcast(id,128,MPI_BYTE,0,comm);CHKERRQ(ierr);
    KSG(ksg_ctx_comm_init_rccl(c,(int)rank,(int)size,id));
  }
  ierr = MPI_Comm_set_attr(comm,KSG_ctx_keyval,(void*)c);CHKERRQ(ierr);
  *ctx = c;
  PetscFunctionReturn(0);
}

/* ---- device pointer accessors --------------------------------------------------------- */
static PetscErrorCode VecHIPCheck(Vec v,Vec_HIP **h)
{
  PetscErrorCode ierr;
  PetscBool      iship;

  PetscFunctionBegin;
  ierr = PetscObjectTypeCompare((PetscObject)v,VECHIP,&iship);CHKERRQ(ierr);
  if (!iship) SETERRQ(PetscObjectComm((PetscObject)v),PETSC_ERR_ARG_WRONG,"Vector is not of type hip");
  *h = (Vec_HIP*)v->data;
  if ((*h)->h_out) SETERRQ(PetscObjectComm((PetscObject)v),PETSC_ERR_ARG_WRONGSTATE,"VecGetArray() is outstanding on this vector");
  PetscFunctionReturn(0);
}

/* The device array, current: allocated if the vector has none yet, the host copy uploaded if
   that is the newer one.  write: the caller is about to change the device copy. */
static PetscErrorCode VecHIPNeedDevice(Vec v,PetscBool write,PetscScalar **a)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = VecHIPCheck(v,&h);CHKERRQ(ierr);
  if (!h->d) {
    if (h->noown) SETERRQ(PetscObjectComm((PetscObject)v),PETSC_ERR_ARG_WRONGSTATE,"Vector has no array: VecPlaceArray() first");
    KSG(ksg_vec_alloc(h->ctx,v->map->n,&h->d_own));           /* zeroed */
    h->d = h->d_unplaced = h->d_own;
    if (h->valid == VECHIP_UNALLOCATED) h->valid = VECHIP_DEVICE;
  }
  if (h->valid == VECHIP_HOST) {
    KSG(ksg_vec_upload(h->ctx,h->d,h->h,v->map->n));
    h->valid = VECHIP_BOTH;
  }
  if (write) h->valid = VECHIP_DEVICE;
  if (a) *a = h->d;
  PetscFunctionReturn(0);
}

/* for pointers that go straight into a ksg_* call: no flush, no touch.  The caller says
   whether the call will write, and bumps the object state itself when it does. */
PETSC_EXTERN PetscErrorCode VecHIPGetDevicePointer_Private(Vec v,PetscBool write,PetscScalar **a)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecHIPNeedDevice(v,write,a);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode VecHIPGetArrayRead(Vec v,const PetscScalar **a)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = VecHIPNeedDevice(v,PETSC_FALSE,(PetscScalar**)a);CHKERRQ(ierr);
  h = (Vec_HIP*)v->data;
  KSG(ksg_ctx_flush(h->ctx));                     /* the array is current for work on ksg_ctx_stream() */
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode VecHIPRestoreArrayRead(Vec v,const PetscScalar **a)
{
  PetscFunctionBegin;
  if (a) *a = NULL;
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode VecHIPGetArray(Vec v,PetscScalar **a)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecHIPGetArrayRead(v,(const PetscScalar**)a);CHKERRQ(ierr);
  ((Vec_HIP*)v->data)->valid = VECHIP_DEVICE;      /* the caller writes the device copy */
  PetscFunctionReturn(0);
}

/* the caller wrote the array with something other than libksgpu */
PETSC_EXTERN PetscErrorCode VecHIPRestoreArray(Vec v,PetscScalar **a)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = VecHIPCheck(v,&h);CHKERRQ(ierr);
  KSG(ksg_ctx_touch(h->ctx));
  ierr = PetscObjectStateIncrease((PetscObject)v);CHKERRQ(ierr);
  if (a) *a = NULL;
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode VecHIPGetArrayWrite(Vec v,PetscScalar **a)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecHIPGetArrayRead(v,(const PetscScalar**)a);CHKERRQ(ierr);   /* a pending update of v is applied, then overwritten */
  ((Vec_HIP*)v->data)->valid = VECHIP_DEVICE;
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode VecHIPRestoreArrayWrite(Vec v,PetscScalar **a)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecHIPRestoreArray(v,a);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* device counterpart of VecPlaceArray / VecResetArray (VecCUDAPlaceArray, sveccuda.cu:510,522) */
PETSC_EXTERN PetscErrorCode VecHIPPlaceArray(Vec v,const PetscScalar *d)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = VecHIPCheck(v,&h);CHKERRQ(ierr);
  if (h->placed) SETERRQ(PetscObjectComm((PetscObject)v),PETSC_ERR_ARG_WRONGSTATE,"VecHIPPlaceArray() was already called on this vector without VecHIPResetArray()");
  /* the host mirror serves whatever array is in use: entries of the own array that live on the
     host only move to the device before the mirror changes hands */
  if (h->valid == VECHIP_HOST) { ierr = VecHIPNeedDevice(v,PETSC_FALSE,NULL);CHKERRQ(ierr); }
  h->d_unplaced = h->d; h->valid_unplaced = h->valid;
  h->d = (PetscScalar*)d; h->placed = PETSC_TRUE;
  h->valid = VECHIP_DEVICE;                     /* the placed array holds the entries now */
  ierr = PetscObjectStateIncrease((PetscObject)v);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode VecHIPResetArray(Vec v)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = VecHIPCheck(v,&h);CHKERRQ(ierr);
  if (h->placed && h->valid == VECHIP_HOST) {    /* the host mirror of the placed array was written last: bring it home */
    KSG(ksg_vec_upload(h->ctx,h->d,h->h,v->map->n));
  }
  h->d = h->d_unplaced; h->placed = PETSC_FALSE;
  h->valid = h->valid_unplaced == VECHIP_UNALLOCATED ? VECHIP_UNALLOCATED : VECHIP_DEVICE;   /* the mirror no longer holds the own entries */
  ierr = PetscObjectStateIncrease((PetscObject)v);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* ---- op table --------------------------------------------------------------------------- */
#define CTX(v) (((Vec_HIP*)(v)->data)->ctx)

static PetscErrorCode VecCopy_HIP(Vec x,Vec y)
{
  PetscErrorCode ierr;
  PetscScalar    *px,*py;

  PetscFunctionBegin;
  if (x->map->n != y->map->n) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ARG_INCOMP,"Incompatible vector local lengths");
  ierr = VecHIPNeedDevice(x,PETSC_FALSE,&px);CHKERRQ(ierr);
  ierr = VecHIPNeedDevice(y,PETSC_TRUE,&py);CHKERRQ(ierr);
  KSG(ksg_vec_copy(CTX(x),px,py,x->map->n));
  PetscFunctionReturn(0);
}

static PetscErrorCode VecSet_HIP(Vec x,PetscScalar alpha)
{
  PetscErrorCode ierr;
  PetscScalar    *px;

  PetscFunctionBegin;
  ierr = VecHIPNeedDevice(x,PETSC_TRUE,&px);CHKERRQ(ierr);
  KSG(ksg_vec_set(CTX(x),px,alpha,x->map->n));
  PetscFunctionReturn(0);
}

static PetscErrorCode VecScale_HIP(Vec x,PetscScalar alpha)
{
  PetscErrorCode ierr;
  PetscScalar    *px;

  PetscFunctionBegin;
  ierr = VecHIPNeedDevice(x,PETSC_TRUE,&px);CHKERRQ(ierr);
  KSG(ksg_vec_scale(CTX(x),px,alpha,x->map->n));
  PetscFunctionReturn(0);
}

static PetscErrorCode VecAXPY_HIP(Vec y,PetscScalar alpha,Vec x)
{
  PetscErrorCode ierr;
  PetscScalar    *px,*py;

  PetscFunctionBegin;
  if (x->map->n != y->map->n) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ARG_INCOMP,"Incompatible vector local lengths");
  ierr = VecHIPNeedDevice(x,PETSC_FALSE,&px);CHKERRQ(ierr);
  ierr = VecHIPNeedDevice(y,PETSC_TRUE,&py);CHKERRQ(ierr);
  KSG(ksg_vec_axpy(CTX(y),py,alpha,px,y->map->n));
  PetscFunctionReturn(0);
}

/* y = alpha y + x and w = alpha x + y: a scale or copy followed by the update (not on the
   Krylov-Schur path; VecAYPX is what KSP's CG calls under the spectral transformation) */
static PetscErrorCode VecAYPX_HIP(Vec y,PetscScalar alpha,Vec x)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecScale_HIP(y,alpha);CHKERRQ(ierr);
  ierr = VecAXPY_HIP(y,1.0,x);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode VecWAXPY_HIP(Vec w,PetscScalar alpha,Vec x,Vec y)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecCopy_HIP(y,w);CHKERRQ(ierr);
  ierr = VecAXPY_HIP(w,alpha,x);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode VecDotLocal_HIP(Vec x,Vec y,PetscScalar *z)
{
  PetscErrorCode ierr;
  PetscScalar    *px,*py;

  PetscFunctionBegin;
  if (x->map->n != y->map->n) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ARG_INCOMP,"Incompatible vector local lengths");
  ierr = VecHIPNeedDevice(x,PETSC_FALSE,&px);CHKERRQ(ierr);
  ierr = VecHIPNeedDevice(y,PETSC_FALSE,&py);CHKERRQ(ierr);
  KSG(ksg_vec_dot_local(CTX(x),px,py,x->map->n,z));
  PetscFunctionReturn(0);
}

/* real scalars: the conjugate is the value itself, VecTDot = VecDot */
static PetscErrorCode VecDot_HIP(Vec x,Vec y,PetscScalar *z)
{
  PetscErrorCode ierr;
  PetscScalar    *px,*py;

  PetscFunctionBegin;
  if (x->map->n != y->map->n) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ARG_INCOMP,"Incompatible vector local lengths");
  ierr = VecHIPNeedDevice(x,PETSC_FALSE,&px);CHKERRQ(ierr);
  ierr = VecHIPNeedDevice(y,PETSC_FALSE,&py);CHKERRQ(ierr);
  KSG(ksg_vec_dot(CTX(x),px,py,x->map->n,z));              /* sum over the ranks on RCCL */
  PetscFunctionReturn(0);
}

static PetscErrorCode VecNormLocal_HIP(Vec x,NormType type,PetscReal *z)
{
  PetscErrorCode ierr;
  PetscScalar    s;

  PetscFunctionBegin;
  if (type != NORM_2 && type != NORM_FROBENIUS) SETERRQ(PetscObjectComm((PetscObject)x),PETSC_ERR_SUP,"Only the 2-norm is implemented for vectors of type hip");
  ierr = VecDotLocal_HIP(x,x,&s);CHKERRQ(ierr);
  *z = PetscSqrtReal(PetscRealPart(s));
  PetscFunctionReturn(0);
}

static PetscErrorCode VecNorm_HIP(Vec x,NormType type,PetscReal *z)
{
  PetscErrorCode ierr;
  PetscScalar    *px;

  PetscFunctionBegin;
  if (type != NORM_2 && type != NORM_FROBENIUS) SETERRQ(PetscObjectComm((PetscObject)x),PETSC_ERR_SUP,"Only the 2-norm is implemented for vectors of type hip");
  ierr = VecHIPNeedDevice(x,PETSC_FALSE,&px);CHKERRQ(ierr);
  KSG(ksg_vec_norm2(CTX(x),px,x->map->n,z));
  PetscFunctionReturn(0);
}

/* Values that depend only on the global index of an entry (and a seed that moves on with every
   call), so that a random vector is the same for any number of ranks -- what the reference gets
   from -bv_reproducible_random (bvops.c:399-406).  The PetscRandom context only supplies the seed. */
static PetscErrorCode VecSetRandom_HIP(Vec x,PetscRandom r)
{
  PetscErrorCode ierr;
  PetscScalar    *px;
  unsigned long  seed;
  PetscInt       rstart;

  PetscFunctionBegin;
  ierr = PetscRandomGetSeed(r,&seed);CHKERRQ(ierr);
  ierr = PetscRandomSetSeed(r,seed*6364136223846793005UL+1442695040888963407UL);CHKERRQ(ierr);
  ierr = VecGetOwnershipRange(x,&rstart,NULL);CHKERRQ(ierr);
  ierr = VecHIPNeedDevice(x,PETSC_TRUE,&px);CHKERRQ(ierr);
  KSG(ksg_vec_fill_hash(CTX(x),px,x->map->n,rstart,(uint64_t)seed));
  PetscFunctionReturn(0);
}

/* Host access.  PETSc 3.7 reaches this slot from VecGetArray and VecGetArrayRead alike
   (vec/vec/interface/rvector.c, vectors that are not "petscnative"), so every Get counts as a
   write: the host copy is the current one afterwards, and the next device operation uploads it. */
static PetscErrorCode VecGetArray_HIP(Vec v,PetscScalar **a)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = VecHIPCheck(v,&h);CHKERRQ(ierr);
  if (!h->h) { ierr = PetscCalloc1(v->map->n+1,&h->h);CHKERRQ(ierr); }
  if (h->valid == VECHIP_DEVICE) KSG(ksg_vec_download(h->ctx,h->h,h->d,v->map->n));   /* applies a pending update first, synchronises */
  h->valid = VECHIP_HOST;
  h->h_out = PETSC_TRUE;
  *a = h->h;
  PetscFunctionReturn(0);
}

static PetscErrorCode VecRestoreArray_HIP(Vec v,PetscScalar **a)
{
  Vec_HIP *h = (Vec_HIP*)v->data;

  PetscFunctionBegin;
  if (!h->h_out) SETERRQ(PetscObjectComm((PetscObject)v),PETSC_ERR_ARG_WRONGSTATE,"VecGetArray() was not called");
  h->h_out = PETSC_FALSE;
  /* a column view must not keep its entries on the host past BVRestoreColumn: the BV kernels
     read the basis storage directly */
  if (h->placed) {
    KSG(ksg_vec_upload(h->ctx,h->d,h->h,v->map->n));       /* bumps the library's write counter */
    h->valid = VECHIP_BOTH;
  }
  PetscFunctionReturn(0);
}

static PetscErrorCode VecPlaceArray_HIP(Vec v,const PetscScalar *a)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecHIPPlaceArray(v,a);CHKERRQ(ierr);              /* the array of a device vector is a device array */
  PetscFunctionReturn(0);
}

static PetscErrorCode VecResetArray_HIP(Vec v)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecHIPResetArray(v);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode VecGetSize_HIP(Vec v,PetscInt *N) { *N = v->map->N; return 0; }
static PetscErrorCode VecGetLocalSize_HIP(Vec v,PetscInt *n) { *n = v->map->n; return 0; }

static PetscErrorCode VecDestroy_HIP(Vec v)
{
  PetscErrorCode ierr;
  Vec_HIP        *h = (Vec_HIP*)v->data;

  PetscFunctionBegin;
  if (h) {
    if (h->d_own) KSG(ksg_vec_free(h->ctx,h->d_own));
    ierr = PetscFree(h->h);CHKERRQ(ierr);
    ierr = PetscFree(v->data);CHKERRQ(ierr);
  }
  PetscFunctionReturn(0);
}

static PetscErrorCode VecCreate_HIP_Private(Vec v,PetscBool alloc);

static PetscErrorCode VecDuplicate_HIP(Vec v,Vec *w)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecCreate(PetscObjectComm((PetscObject)v),w);CHKERRQ(ierr);
  ierr = PetscLayoutReference(v->map,&(*w)->map);CHKERRQ(ierr);
  ierr = VecCreate_HIP_Private(*w,PETSC_TRUE);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode VecCreate_HIP_Private(Vec v,PetscBool alloc)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = PetscLayoutSetUp(v->map);CHKERRQ(ierr);
  ierr = PetscNewLog(v,&h);CHKERRQ(ierr);
  v->data        = (void*)h;
  v->petscnative = PETSC_FALSE;                /* VecGetArray goes through ops->getarray */
  ierr = KSGGetContext(PetscObjectComm((PetscObject)v),&h->ctx);CHKERRQ(ierr);
  h->noown = (PetscBool)!alloc;                 /* storage itself comes at first use, on the side that uses it */
  h->valid = VECHIP_UNALLOCATED;
  ierr = PetscMemzero(v->ops,sizeof(struct _VecOps));CHKERRQ(ierr);
  v->ops->duplicate     = VecDuplicate_HIP;
  v->ops->duplicatevecs = VecDuplicateVecs_Default;
  v->ops->destroyvecs   = VecDestroyVecs_Default;
  v->ops->copy          = VecCopy_HIP;
  v->ops->set           = VecSet_HIP;
  v->ops->scale         = VecScale_HIP;
  v->ops->axpy          = VecAXPY_HIP;
  v->ops->aypx          = VecAYPX_HIP;
  v->ops->waxpy         = VecWAXPY_HIP;
  v->ops->dot           = VecDot_HIP;
  v->ops->tdot          = VecDot_HIP;
  v->ops->dot_local     = VecDotLocal_HIP;
  v->ops->tdot_local    = VecDotLocal_HIP;
  v->ops->norm          = VecNorm_HIP;
  v->ops->norm_local    = VecNormLocal_HIP;
  v->ops->setrandom     = VecSetRandom_HIP;
  v->ops->getarray      = VecGetArray_HIP;
  v->ops->restorearray  = VecRestoreArray_HIP;
  v->ops->placearray    = VecPlaceArray_HIP;
  v->ops->resetarray    = VecResetArray_HIP;
  v->ops->getsize       = VecGetSize_HIP;
  v->ops->getlocalsize  = VecGetLocalSize_HIP;
  v->ops->destroy       = VecDestroy_HIP;
  ierr = PetscObjectChangeTypeName((PetscObject)v,VECHIP);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* VecSetType(v,"hip") / -vec_type hip */
PETSC_EXTERN PetscErrorCode VecCreate_HIP(Vec v)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecCreate_HIP_Private(v,PETSC_TRUE);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* n local, N global entries (PETSC_DECIDE / PETSC_DETERMINE as for VecCreateMPI) */
PETSC_EXTERN PetscErrorCode VecCreateMPIHIP(MPI_Comm comm,PetscInt n,PetscInt N,Vec *v)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = VecCreate(comm,v);CHKERRQ(ierr);
  ierr = VecSetSizes(*v,n,N);CHKERRQ(ierr);
  ierr = VecCreate_HIP_Private(*v,PETSC_TRUE);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* the reusable column views of the BV: no storage of their own when array is NULL, a device
   array is placed for the time a column is out (svec.c:458-476) */
PETSC_EXTERN PetscErrorCode VecCreateMPIHIPWithArray(MPI_Comm comm,PetscInt bs,PetscInt n,PetscInt N,const PetscScalar *array,Vec *v)
{
  PetscErrorCode ierr;
  Vec_HIP        *h;

  PetscFunctionBegin;
  ierr = VecCreate(comm,v);CHKERRQ(ierr);
  ierr = VecSetSizes(*v,n,N);CHKERRQ(ierr);
  ierr = VecSetBlockSize(*v,bs);CHKERRQ(ierr);
  ierr = VecCreate_HIP_Private(*v,PETSC_FALSE);CHKERRQ(ierr);
  h = (Vec_HIP*)(*v)->data;
  h->d = h->d_unplaced = (PetscScalar*)array;
  if (array) h->valid = VECHIP_DEVICE;
  PetscFunctionReturn(0);
}

/* ---- split-phase bookkeeping shared with the BV type (ops->dotvec_begin / dotvec_end) ----
   One group at a time per context: every Begin posts its kernel into the batch of the context
   (ksg_red_post_*), the first End closes the batch with one all-reduce and the values are
   handed out in posting order -- the role of PetscSplitReduction (bvglobal.c:227-241). */
#define KSG_RED_MAXENTRIES 64
typedef struct { ksg_ctx ctx; PetscInt n,next,count[KSG_RED_MAXENTRIES],offset[KSG_RED_MAXENTRIES],total; PetscBool closed; PetscScalar vals[8192]; } KSGRedQueue;
static KSGRedQueue KSG_red;

PETSC_EXTERN PetscErrorCode KSGRedOpenIfNeeded(ksg_ctx ctx)
{
  PetscFunctionBegin;
  if (KSG_red.ctx && KSG_red.closed && KSG_red.next < KSG_red.n) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ORDER,"Called xxxBegin() before all xxxEnd() of the previous group");
  if (KSG_red.ctx && KSG_red.ctx != ctx) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ARG_INCOMP,"Split-phase reductions of one group must live on one context");
  if (!KSG_red.ctx) { KSG(ksg_red_open(ctx)); KSG_red.ctx = ctx; KSG_red.n = KSG_red.next = KSG_red.total = 0; KSG_red.closed = PETSC_FALSE; }
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode KSGRedPost(MPI_Comm comm,ksg_ctx ctx,PetscInt offset,PetscInt count)
{
  PetscFunctionBegin;
  if (KSG_red.ctx != ctx || KSG_red.closed) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ORDER,"No group of split-phase reductions is open on this context");
  if (KSG_red.n >= KSG_RED_MAXENTRIES) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_SUP,"Too many reductions in one group");
  KSG_red.offset[KSG_red.n] = offset; KSG_red.count[KSG_red.n] = count; KSG_red.n++;
  if (offset+count > KSG_red.total) KSG_red.total = offset+count;
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode KSGRedFetch(MPI_Comm comm,ksg_ctx ctx,PetscInt count,PetscScalar *out)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  if (KSG_red.ctx != ctx) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ORDER,"Called xxxEnd() without xxxBegin()");
  if (!KSG_red.closed) { KSG(ksg_red_close(ctx,KSG_red.vals,(int)KSG_red.total)); KSG_red.closed = PETSC_TRUE; }
  if (KSG_red.next >= KSG_red.n || KSG_red.count[KSG_red.next] != count) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ARG_WRONGSTATE,"Called xxxEnd() in a different order or with a different object than xxxBegin()");
  ierr = PetscMemcpy(out,KSG_red.vals+KSG_red.offset[KSG_red.next],count*sizeof(PetscScalar));CHKERRQ(ierr);
  if (++KSG_red.next == KSG_red.n) KSG_red.ctx = NULL;      /* all handed out: the next Begin opens a new group */
  PetscFunctionReturn(0);
}
