/*
 * PETSc/SLEPc side of the drop-in: BV type "hip" for a real SLEPc 3.7 build.
 *
 * NOT compiled in this repository (PETSc is not available here): this is the
 * binding a SLEPc maintainer adds next to src/sys/classes/bv/impls/svec/ and links
 * against libksgpu.so.  It fills struct _BVOps (include/slepc/private/bvimpl.h:35-64)
 * exactly as BVCreate_Svec does (svec.c:482-517), every slot that is called without a NULL
 * check (bvops.c, bvglobal.c, bvbasic.c) forwarding to the C ABI
 * of include/ksgpu.h.  Register with
 *     BVRegister("hip", BVCreate_HIP);          (include/slepcbv.h:196)
 * after SlepcInitialize and select with -bv_type hip.
 *
 * Column views are vectors of the device Vec type of vec_hip_petsc.c (with the PETSc
 * generation of the reference the equivalent was VECCUDA and VecCUDAPlaceArray,
 * sveccuda.cu:500-525).  libksgpu may hold an update of a column back until the next call
 * decides how to apply it (include/ksgpu.h, KSG_OPT_LAZY_UPDATE); the public accessors of that
 * Vec type (VecHIPGetArrayRead, ...) bracket foreign code accordingly.  The ops in this file
 * hand every device pointer straight back to libksgpu, which needs no bracket, so they use
 * VecHIPGetDevicePointer_Private.
 */
#include <slepc/private/bvimpl.h>
#include "ksgpu.h"

/* vec_hip_petsc.c */
PETSC_EXTERN PetscErrorCode KSGGetContext(MPI_Comm,ksg_ctx*);
PETSC_EXTERN PetscErrorCode VecHIPGetDevicePointer_Private(Vec,PetscBool write,PetscScalar**);
PETSC_EXTERN PetscErrorCode VecHIPPlaceArray(Vec,const PetscScalar*);
PETSC_EXTERN PetscErrorCode VecHIPResetArray(Vec);
PETSC_EXTERN PetscErrorCode VecCreateMPIHIPWithArray(MPI_Comm,PetscInt,PetscInt,PetscInt,const PetscScalar*,Vec*);
PETSC_EXTERN PetscErrorCode KSGRedOpenIfNeeded(ksg_ctx);

typedef struct { ksg_ctx ctx; ksg_bv bv; } BV_HIP;

#define KSG(call) do { if ((call)) SETERRQ1(PETSC_COMM_SELF,PETSC_ERR_LIB,"libksgpu: %s",ksg_last_error()); } while (0)
/* SLEPc's interface layer owns nc, l and k (svec reads them from the BV at every op) */
#define SYNC(B_) do { KSG(ksg_bv_set_num_constraints(((BV_HIP*)(B_)->data)->bv,(int)(B_)->nc)); \
                      KSG(ksg_bv_set_active(((BV_HIP*)(B_)->data)->bv,(int)(B_)->l,(int)(B_)->k)); } while (0)

static PetscErrorCode BVDotVec_HIP(BV X,Vec y,PetscScalar *q)
{
  PetscErrorCode    ierr;
  BV_HIP            *x = (BV_HIP*)X->data;
  const PetscScalar *py;
  PetscScalar       *qq = q;
  Vec               z = y;

  PetscFunctionBegin;
  SYNC(X);
  if (X->matrix) {                                                   /* svec.c:131-134: B-inner product */
    ierr = BV_IPMatMult(X,y);CHKERRQ(ierr);
    z = X->Bx;
  }
  if (!q) { ierr = VecGetArray(X->buffer,&qq);CHKERRQ(ierr); }      /* svec.c:137 */
  ierr = VecHIPGetDevicePointer_Private(z,PETSC_FALSE,(PetscScalar**)&py);CHKERRQ(ierr);
  KSG(ksg_bv_dotvec(x->bv,py,qq));                                  /* reduction included */
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
  ierr = VecHIPGetDevicePointer_Private(y,PETSC_TRUE,&py);CHKERRQ(ierr);
  KSG(ksg_bv_multvec(x->bv,alpha,beta,py,qq));
  ierr = PetscObjectStateIncrease((PetscObject)y);CHKERRQ(ierr);
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

static PetscErrorCode BVMultInPlaceTranspose_HIP(BV V,Mat Q,PetscInt s,PetscInt e)
{
  PetscErrorCode ierr;
  PetscScalar    *q;
  PetscInt       ldq;

  PetscFunctionBegin;
  SYNC(V);
  ierr = MatGetSize(Q,&ldq,NULL);CHKERRQ(ierr);
  ierr = MatDenseGetArray(Q,&q);CHKERRQ(ierr);
  KSG(ksg_bv_multinplace(((BV_HIP*)V->data)->bv,q,(int)ldq,(int)s,(int)e,1));   /* svec.c:85-100 */
  ierr = MatDenseRestoreArray(Q,&q);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVMult_HIP(BV Y,PetscScalar alpha,PetscScalar beta,BV X,Mat Q)
{
  PetscErrorCode ierr;
  PetscScalar    *q;
  PetscInt       ldq;

  PetscFunctionBegin;
  SYNC(X); SYNC(Y);
  ierr = MatGetSize(Q,&ldq,NULL);CHKERRQ(ierr);
  ierr = MatDenseGetArray(Q,&q);CHKERRQ(ierr);
  KSG(ksg_bv_mult(((BV_HIP*)Y->data)->bv,alpha,beta,((BV_HIP*)X->data)->bv,q,(int)ldq));   /* svec.c:27-49 */
  ierr = MatDenseRestoreArray(Q,&q);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVDot_HIP(BV X,BV Y,Mat M)
{
  PetscErrorCode ierr;
  PetscScalar    *m;
  PetscInt       ldm;

  PetscFunctionBegin;
  SYNC(X); SYNC(Y);
  ierr = MatGetSize(M,&ldm,NULL);CHKERRQ(ierr);
  ierr = MatDenseGetArray(M,&m);CHKERRQ(ierr);
  KSG(ksg_bv_dot(((BV_HIP*)X->data)->bv,((BV_HIP*)Y->data)->bv,m,(int)ldm));               /* svec.c:102-120 */
  ierr = MatDenseRestoreArray(M,&m);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVDotVec_Local_HIP(BV X,Vec y,PetscScalar *m)
{
  PetscErrorCode    ierr;
  const PetscScalar *py;
  Vec               z = y;

  PetscFunctionBegin;
  SYNC(X);
  if (X->matrix) {                                                   /* svec.c:153-156 */
    ierr = BV_IPMatMult(X,y);CHKERRQ(ierr);
    z = X->Bx;
  }
  ierr = VecHIPGetDevicePointer_Private(z,PETSC_FALSE,(PetscScalar**)&py);CHKERRQ(ierr);
  KSG(ksg_bv_dotvec_local(((BV_HIP*)X->data)->bv,py,m));            /* svec.c:145-163: no reduction */
  PetscFunctionReturn(0);
}

/* Split phase (ops->dotvec_begin / dotvec_end, bvimpl.h:43-44, called from bvglobal.c:223-224,
   272-273 in place of PETSc's PetscSplitReduction): the kernel is queued into the batch of
   the context (opened by the first Begin of a group, KSGRedOpenIfNeeded), and the first ...End
   of the group closes it with one all-reduce (ksg_red_close); the values are handed out in
   posting order.  KSGRedPost/KSGRedFetch are that bookkeeping (vec_hip_petsc.c: a queue of
   {offset,count} as PetscSplitReduction keeps). */
PETSC_EXTERN PetscErrorCode KSGRedPost(MPI_Comm,ksg_ctx,PetscInt offset,PetscInt count);
PETSC_EXTERN PetscErrorCode KSGRedFetch(MPI_Comm,ksg_ctx,PetscInt count,PetscScalar *out);

static PetscErrorCode BVDotVec_Begin_HIP(BV X,Vec y,PetscScalar *m)
{
  PetscErrorCode    ierr;
  BV_HIP            *x = (BV_HIP*)X->data;
  const PetscScalar *py;
  Vec               z = y;
  int               offset;

  PetscFunctionBegin;
  SYNC(X);
  if (X->matrix) {
    ierr = BV_IPMatMult(X,y);CHKERRQ(ierr);
    z = X->Bx;
  }
  ierr = VecHIPGetDevicePointer_Private(z,PETSC_FALSE,(PetscScalar**)&py);CHKERRQ(ierr);
  ierr = KSGRedOpenIfNeeded(x->ctx);CHKERRQ(ierr);
  KSG(ksg_red_post_dotvec(x->bv,py,&offset));
  ierr = KSGRedPost(PetscObjectComm((PetscObject)X),x->ctx,offset,X->k-X->l);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVDotVec_End_HIP(BV X,Vec y,PetscScalar *m)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = KSGRedFetch(PetscObjectComm((PetscObject)X),((BV_HIP*)X->data)->ctx,X->k-X->l,m);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVNorm_Local_HIP(BV bv,PetscInt j,NormType type,PetscReal *val)
{
  PetscFunctionBegin;
  SYNC(bv);
  KSG(ksg_bv_norm_local(((BV_HIP*)bv->data)->bv,(int)j,(int)type,val));   /* svec.c:199-214 */
  PetscFunctionReturn(0);
}

/* W = A*V column by column through MatMult on the two column views, as BVMatMult_Svec does
   (svec.c:216-262); an operator of the device type can take ksg_bv_matmult, one pass over the
   matrix for all the columns, instead */
static PetscErrorCode BVMatMult_HIP(BV V,Mat A,BV W)
{
  PetscErrorCode ierr;
  PetscInt       j;
  Vec            v,w;

  PetscFunctionBegin;
  for (j=0;j<V->k-V->l;j++) {
    ierr = BVGetColumn(V,V->l+j,&v);CHKERRQ(ierr);
    ierr = BVGetColumn(W,W->l+j,&w);CHKERRQ(ierr);
    ierr = MatMult(A,v,w);CHKERRQ(ierr);
    ierr = BVRestoreColumn(V,V->l+j,&v);CHKERRQ(ierr);
    ierr = BVRestoreColumn(W,W->l+j,&w);CHKERRQ(ierr);
  }
  PetscFunctionReturn(0);
}

static PetscErrorCode BVCopy_HIP(BV V,BV W)
{
  PetscFunctionBegin;
  SYNC(V); SYNC(W);
  KSG(ksg_bv_copy(((BV_HIP*)V->data)->bv,((BV_HIP*)W->data)->bv));   /* svec.c:264-278 */
  PetscFunctionReturn(0);
}

static PetscErrorCode BVResize_HIP(BV bv,PetscInt m,PetscBool copy)
{
  PetscFunctionBegin;
  KSG(ksg_bv_resize(((BV_HIP*)bv->data)->bv,(int)m,(int)copy));      /* svec.c:280-310 */
  PetscFunctionReturn(0);
}

/* BVGetArray hands out host memory (svec.c:339-402 with the CUDA variant copying back and
   forth, sveccuda.cu:527-590): a host mirror of the whole storage, uploaded again on restore */
static PetscErrorCode BVGetArray_HIP(BV bv,PetscScalar **a)
{
  PetscErrorCode ierr;
  BV_HIP         *ctx = (BV_HIP*)bv->data;
  PetscInt       j;
  int64_t        nloc,ld;

  PetscFunctionBegin;
  KSG(ksg_bv_sizes(ctx->bv,&nloc,NULL,NULL,&ld));
  ierr = PetscMalloc1((bv->nc+bv->m)*bv->n,a);CHKERRQ(ierr);
  for (j=0;j<bv->nc+bv->m;j++) KSG(ksg_vec_download(ctx->ctx,*a+j*bv->n,ksg_bv_column(ctx->bv,(int)(j-bv->nc)),nloc));
  PetscFunctionReturn(0);
}

static PetscErrorCode BVRestoreArray_HIP(BV bv,PetscScalar **a)
{
  PetscErrorCode ierr;
  BV_HIP         *ctx = (BV_HIP*)bv->data;
  PetscInt       j;

  PetscFunctionBegin;
  for (j=0;j<bv->nc+bv->m;j++) KSG(ksg_vec_upload(ctx->ctx,ksg_bv_column(ctx->bv,(int)(j-bv->nc)),*a+j*bv->n,bv->n));
  ierr = PetscFree(*a);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVGetArrayRead_HIP(BV bv,const PetscScalar **a)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = BVGetArray_HIP(bv,(PetscScalar**)a);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVRestoreArrayRead_HIP(BV bv,const PetscScalar **a)
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  ierr = PetscFree(*a);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode BVDestroy_HIP(BV bv)
{
  PetscErrorCode ierr;
  BV_HIP         *ctx = (BV_HIP*)bv->data;

  PetscFunctionBegin;
  KSG(ksg_bv_destroy(ctx->bv));                 /* the context belongs to the communicator (KSGGetContext) */
  ierr = VecDestroy(&bv->cv[0]);CHKERRQ(ierr);
  ierr = VecDestroy(&bv->cv[1]);CHKERRQ(ierr);
  ierr = PetscFree(bv->data);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode BVCreate_HIP(BV bv)
{
  PetscErrorCode ierr;
  BV_HIP         *ctx;
  PetscInt       nloc,bs;

  PetscFunctionBegin;
  ierr = PetscNewLog(bv,&ctx);CHKERRQ(ierr);
  bv->data = (void*)ctx;
  ierr = VecGetLocalSize(bv->t,&nloc);CHKERRQ(ierr);
  ierr = VecGetBlockSize(bv->t,&bs);CHKERRQ(ierr);
  /* the context of this rank on the BV's communicator: device, streams and -- with several
     ranks -- the RCCL communicator, shared with the vectors and the operator */
  ierr = KSGGetContext(PetscObjectComm((PetscObject)bv),&ctx->ctx);CHKERRQ(ierr);
  KSG(ksg_bv_create(ctx->ctx,nloc,bv->N,bv->m,&ctx->bv));
  /* the two reusable column views, as svec.c:458-476 */
  ierr = VecCreateMPIHIPWithArray(PetscObjectComm((PetscObject)bv->t),bs,nloc,PETSC_DECIDE,NULL,&bv->cv[0]);CHKERRQ(ierr);
  ierr = VecCreateMPIHIPWithArray(PetscObjectComm((PetscObject)bv->t),bs,nloc,PETSC_DECIDE,NULL,&bv->cv[1]);CHKERRQ(ierr);

  bv->ops->mult             = BVMult_HIP;
  bv->ops->multvec          = BVMultVec_HIP;
  bv->ops->multinplace      = BVMultInPlace_HIP;
  bv->ops->multinplacetrans = BVMultInPlaceTranspose_HIP;
  bv->ops->dot              = BVDot_HIP;
  bv->ops->dotvec           = BVDotVec_HIP;
  bv->ops->dotvec_local     = BVDotVec_Local_HIP;
  bv->ops->dotvec_begin     = BVDotVec_Begin_HIP;
  bv->ops->dotvec_end       = BVDotVec_End_HIP;
  bv->ops->scale            = BVScale_HIP;
  bv->ops->norm             = BVNorm_HIP;
  bv->ops->norm_local       = BVNorm_Local_HIP;
  bv->ops->matmult          = BVMatMult_HIP;
  bv->ops->copy             = BVCopy_HIP;
  bv->ops->resize           = BVResize_HIP;
  bv->ops->getcolumn        = BVGetColumn_HIP;
  bv->ops->restorecolumn    = BVRestoreColumn_HIP;
  bv->ops->getarray         = BVGetArray_HIP;
  bv->ops->restorearray     = BVRestoreArray_HIP;
  bv->ops->getarrayread     = BVGetArrayRead_HIP;
  bv->ops->restorearrayread = BVRestoreArrayRead_HIP;
  bv->ops->destroy          = BVDestroy_HIP;
  PetscFunctionReturn(0);
}
