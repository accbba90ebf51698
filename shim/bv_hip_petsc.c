/*
 * PETSc/SLEPc side of the drop-in: BV type "hip" for a real SLEPc 3.7 build.
 *
 * NOT compiled in this repository (PETSc is not available here): this is the
 * binding a SLEPc maintainer adds next to src/sys/classes/bv/impls/svec/ and links
 * against libksgpu.so.  It fills struct _BVOps (include/slepc/private/bvimpl.h:35-64)
 * exactly as BVCreate_Svec does (svec.c:482-517), every slot forwarding to the C ABI
 * of include/ksgpu.h.  Register with
 *     BVRegister("hip", BVCreate_HIP);          (include/slepcbv.h:196)
 * after SlepcInitialize and select with -bv_type hip.
 *
 * Column views are device vectors (with the PETSc generation of the reference the
 * equivalent was VECCUDA and VecCUDAPlaceArray, sveccuda.cu:500-525).  libksgpu may hold an
 * update of a column back until the next call decides how to apply it (include/ksgpu.h,
 * KSG_OPT_LAZY_UPDATE), so the accessors through which code outside this file reaches a
 * view's device pointer must bracket it: readers call ksg_ctx_flush() first, writers call
 * ksg_bv_touch() afterwards.  The VecHIP* names below stand for those bracketed accessors
 * of the view type; the ops in this file use the raw pointer (no bracket needed: they hand
 * it straight back to libksgpu).
 */
#include <slepc/private/bvimpl.h>
#include "ksgpu.h"

typedef struct { ksg_ctx ctx; ksg_bv bv; } BV_HIP;

#define KSG(call) do { if ((call)) SETERRQ1(PETSC_COMM_SELF,PETSC_ERR_LIB,"libksgpu: %s",ksg_last_error()); } while (0)
#define SYNC(bv) KSG(ksg_bv_set_active(((BV_HIP*)(bv)->data)->bv,(bv)->l,(bv)->k))

static PetscErrorCode BVDotVec_HIP(BV X,Vec y,PetscScalar *q)
{
  PetscErrorCode    ierr;
  BV_HIP            *x = (BV_HIP*)X->data;
  const PetscScalar *py;
  PetscScalar       *qq = q;

  PetscFunctionBegin;
  SYNC(X);
  if (!q) { ierr = VecGetArray(X->buffer,&qq);CHKERRQ(ierr); }      /* svec.c:137 */
  ierr = VecHIPGetArrayRead(y,&py);CHKERRQ(ierr);
  KSG(ksg_bv_dotvec(x->bv,py,qq));                                  /* reduction included */
  ierr = VecHIPRestoreArrayRead(y,&py);CHKERRQ(ierr);
  if (!q) { ierr = VecRestoreArray(X->buffer,&qq);CHKERRQ(ierr); }
  PetscFunctionReturn(0);
}

static PetscErrorCode BVMultVec_HIP(BV X,PetscScalar alpha,PetscScalar beta,Vec y,PetscScalar *q)
{
  PetscErrorCode ierr;
  BV_HIP         *x = (BV_HIP*)X->data;
  PetscScalar    *py,*qq = q;

  PetscFunctionBegin;
  SYNC(X);
  if (!q) { ierr = VecGetArray(X->buffer,&qq);CHKERRQ(ierr); }      /* svec.c:60 */
  ierr = VecHIPGetArray(y,&py);CHKERRQ(ierr);
  KSG(ksg_bv_multvec(x->bv,alpha,beta,py,qq));
  ierr = VecHIPRestoreArray(y,&py);CHKERRQ(ierr);
  if (!q) { ierr = VecRestoreArray(X->buffer,&qq);CHKERRQ(ierr); }
  PetscFunctionReturn(0);
}

static PetscErrorCode BVMultInPlace_HIP(BV V,Mat Q,PetscInt s,PetscInt e)
{
  PetscErrorCode ierr;
  PetscScalar    *q;
  PetscInt       ldq;

  PetscFunctionBegin;
  SYNC(V);
  ierr = MatGetSize(Q,&ldq,NULL);CHKERRQ(ierr);
  ierr = MatDenseGetArray(Q,&q);CHKERRQ(ierr);
  KSG(ksg_bv_multinplace(((BV_HIP*)V->data)->bv,q,(int)ldq,(int)s,(int)e,0));   /* svec.c:79 */
  ierr = MatDenseRestoreArray(Q,&q);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVScale_HIP(BV bv,PetscInt j,PetscScalar alpha)
{
  PetscFunctionBegin;
  SYNC(bv);
  KSG(ksg_bv_scale(((BV_HIP*)bv->data)->bv,(int)j,alpha));
  PetscFunctionReturn(0);
}

static PetscErrorCode BVNorm_HIP(BV bv,PetscInt j,NormType type,PetscReal *val)
{
  PetscFunctionBegin;
  SYNC(bv);
  KSG(ksg_bv_norm(((BV_HIP*)bv->data)->bv,(int)j,(int)type,val));   /* NormType values coincide */
  PetscFunctionReturn(0);
}

static PetscErrorCode BVGetColumn_HIP(BV bv,PetscInt j,Vec *v)
{
  PetscErrorCode ierr;
  PetscInt       l = BVAvailableVec;

  PetscFunctionBegin;
  ierr = VecHIPPlaceArray(bv->cv[l],ksg_bv_column(((BV_HIP*)bv->data)->bv,(int)j));CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVRestoreColumn_HIP(BV bv,PetscInt j,Vec *v)
{
  PetscErrorCode ierr;
  PetscInt       l = (j==bv->ci[0])? 0: 1;

  PetscFunctionBegin;
  ierr = VecHIPResetArray(bv->cv[l]);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* mult, multinplacetrans, dot, dotvec_local, norm_local, matmult, copy, resize,
   getarray(read), destroy: the same three lines each, forwarding to ksg_bv_mult,
   ksg_bv_multinplace(...,1), ksg_bv_dot, ksg_bv_dotvec_local, ksg_bv_norm_local,
   ksg_bv_matmult (or MatMult on two column views), ksg_bv_copy, ksg_bv_resize,
   ksg_vec_download into a host mirror, ksg_bv_destroy. */

PETSC_EXTERN PetscErrorCode BVCreate_HIP(BV bv)
{
  PetscErrorCode ierr;
  BV_HIP         *ctx;
  PetscInt       nloc,bs;
  PetscMPIInt    rank,size;

  PetscFunctionBegin;
  ierr = PetscNewLog(bv,&ctx);CHKERRQ(ierr);
  bv->data = (void*)ctx;
  ierr = VecGetLocalSize(bv->t,&nloc);CHKERRQ(ierr);
  ierr = VecGetBlockSize(bv->t,&bs);CHKERRQ(ierr);
  ierr = MPI_Comm_rank(PetscObjectComm((PetscObject)bv),&rank);CHKERRQ(ierr);
  ierr = MPI_Comm_size(PetscObjectComm((PetscObject)bv),&size);CHKERRQ(ierr);
  KSG(ksg_ctx_create(/* device of this rank */ rank % 8,&ctx->ctx));
  if (size>1) {          /* one RCCL communicator per BV communicator */
    char id[128];
    if (!rank) KSG(ksg_comm_unique_id(id));
    ierr = MPI_Bcast(id,128,MPI_BYTE,0,PetscObjectComm((PetscObject)bv));CHKERRQ(ierr);
    KSG(ksg_ctx_comm_init_rccl(ctx->ctx,rank,size,id));
  }
  KSG(ksg_bv_create(ctx->ctx,nloc,bv->N,bv->m,&ctx->bv));
  /* the two reusable column views, as svec.c:458-476 */
  ierr = VecCreateMPIHIPWithArray(PetscObjectComm((PetscObject)bv->t),bs,nloc,PETSC_DECIDE,NULL,&bv->cv[0]);CHKERRQ(ierr);
  ierr = VecCreateMPIHIPWithArray(PetscObjectComm((PetscObject)bv->t),bs,nloc,PETSC_DECIDE,NULL,&bv->cv[1]);CHKERRQ(ierr);

  bv->ops->multvec       = BVMultVec_HIP;
  bv->ops->multinplace   = BVMultInPlace_HIP;
  bv->ops->dotvec        = BVDotVec_HIP;
  bv->ops->scale         = BVScale_HIP;
  bv->ops->norm          = BVNorm_HIP;
  bv->ops->getcolumn     = BVGetColumn_HIP;
  bv->ops->restorecolumn = BVRestoreColumn_HIP;
  /* ... remaining slots as listed above ... */
  PetscFunctionReturn(0);
}
