/*
 * PETSc side of the drop-in operator: Mat type "aijhip" for a PETSc 3.7 / SLEPc 3.7 build.
 *
 * NOT compiled in this repository (PETSc is not available here): this is the binding a
 * maintainer adds next to the BV type of bv_hip_petsc.c and links against libksgpu.so.
 * The type derives from AIJ, so user code assembles it as always (MatSetValues loop,
 * MatAssemblyBegin/End: src/eps/examples/tutorials/ex2.c:50-66) and selects it with
 * -mat_type aijhip through the MatSetFromOptions call it already makes (ex2.c:52).  At the end
 * of the assembly the local rows go to the device once (ksg_mat_create_csr: diagonal/ghost
 * split and halo plan included); from then on MatMult -- all the solver needs from the
 * operator, src/sys/classes/st/impls/shift/shift.c:38 -- runs in libksgpu, and MatCreateVecs
 * (src/sys/classes/st/interface/stfunc.c:546) hands out device vectors, so the BV template
 * vector and with it every column view is a device vector.
 *
 * Register with
 *     MatRegister("aijhip", MatCreate_AIJHIP);
 * after PetscInitialize.  Vectors are of the device Vec type of vec_hip_petsc.c; the libksgpu
 * context of the matrix is the one of its MPI communicator (one ksg_ctx per rank, shared with
 * the vectors and the BV: KSGGetContext).
 */
#include <petsc/private/matimpl.h>
#include "ksgpu.h"

typedef struct {
  ksg_mat       m;                                  /* device copy of the local rows */
  PetscErrorCode (*assemblyend_aij)(Mat,MatAssemblyType);
  PetscErrorCode (*destroy_aij)(Mat);
} Mat_AIJHIP;

#define KSG(call) do { if ((call)) SETERRQ1(PETSC_COMM_SELF,PETSC_ERR_LIB,"libksgpu: %s",ksg_last_error()); } while (0)

/* vec_hip_petsc.c */
PETSC_EXTERN PetscErrorCode KSGGetContext(MPI_Comm,ksg_ctx*);      /* the per-rank context shared with the BV type */
PETSC_EXTERN PetscErrorCode VecCreateMPIHIP(MPI_Comm,PetscInt,PetscInt,Vec*);
PETSC_EXTERN PetscErrorCode VecHIPGetDevicePointer_Private(Vec,PetscBool write,PetscScalar**);

static PetscErrorCode MatGetHIP(Mat A,Mat_AIJHIP **h)
{
  PetscErrorCode ierr;
  PetscContainer c;

  PetscFunctionBegin;
  ierr = PetscObjectQuery((PetscObject)A,"ksg_aijhip",(PetscObject*)&c);CHKERRQ(ierr);
  if (!c) SETERRQ(PetscObjectComm((PetscObject)A),PETSC_ERR_PLIB,"Not an aijhip matrix");
  ierr = PetscContainerGetPointer(c,(void**)h);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* y = A x on the device (MatMult_SeqAIJ / MatMult_MPIAIJ and their VecScatter replaced) */
static PetscErrorCode MatMult_AIJHIP(Mat A,Vec x,Vec y)
{
  PetscErrorCode    ierr;
  Mat_AIJHIP        *h;
  const PetscScalar *px;
  PetscScalar       *py;

  PetscFunctionBegin;
  ierr = MatGetHIP(A,&h);CHKERRQ(ierr);
  /* both pointers go straight into libksgpu, which applies what it holds back itself */
  ierr = VecHIPGetDevicePointer_Private(x,PETSC_FALSE,(PetscScalar**)&px);CHKERRQ(ierr);
  ierr = VecHIPGetDevicePointer_Private(y,PETSC_TRUE,&py);CHKERRQ(ierr);
  KSG(ksg_mat_mult(h->m,px,py));
  ierr = PetscObjectStateIncrease((PetscObject)y);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode MatCreateVecs_AIJHIP(Mat A,Vec *right,Vec *left)
{
  PetscErrorCode ierr;
  PetscInt       m,n,M,N;

  PetscFunctionBegin;
  ierr = MatGetLocalSize(A,&m,&n);CHKERRQ(ierr);
  ierr = MatGetSize(A,&M,&N);CHKERRQ(ierr);
  if (right) { ierr = VecCreateMPIHIP(PetscObjectComm((PetscObject)A),n,N,right);CHKERRQ(ierr); }
  if (left) { ierr = VecCreateMPIHIP(PetscObjectComm((PetscObject)A),m,M,left);CHKERRQ(ierr); }
  PetscFunctionReturn(0);
}

/* after the AIJ assembly: local rows with global column indices -> device */
static PetscErrorCode MatAssemblyEnd_AIJHIP(Mat A,MatAssemblyType mode)
{
  PetscErrorCode    ierr;
  Mat_AIJHIP        *h;
  ksg_ctx           ctx;
  PetscInt          rstart,rend,cstart,cend,M,N,i,j,ncols;
  const PetscInt    *cols;
  const PetscScalar *vals;
  int64_t           *ia,nz = 0;
  int32_t           *ja;
  double            *va;

  PetscFunctionBegin;
  ierr = MatGetHIP(A,&h);CHKERRQ(ierr);
  ierr = (*h->assemblyend_aij)(A,mode);CHKERRQ(ierr);
  if (mode == MAT_FLUSH_ASSEMBLY) PetscFunctionReturn(0);
  ierr = MatGetOwnershipRange(A,&rstart,&rend);CHKERRQ(ierr);
  ierr = MatGetOwnershipRangeColumn(A,&cstart,&cend);CHKERRQ(ierr);
  ierr = MatGetSize(A,&M,&N);CHKERRQ(ierr);
  ierr = PetscMalloc1(rend-rstart+1,&ia);CHKERRQ(ierr);
  ia[0] = 0;
  for (i=rstart;i<rend;i++) {
    ierr = MatGetRow(A,i,&ncols,NULL,NULL);CHKERRQ(ierr);
    ia[i-rstart+1] = ia[i-rstart]+ncols;
    ierr = MatRestoreRow(A,i,&ncols,NULL,NULL);CHKERRQ(ierr);
  }
  nz = ia[rend-rstart];
  ierr = PetscMalloc2(nz,&ja,nz,&va);CHKERRQ(ierr);
  for (i=rstart;i<rend;i++) {
    ierr = MatGetRow(A,i,&ncols,&cols,&vals);CHKERRQ(ierr);          /* global columns, ascending */
    for (j=0;j<ncols;j++) { ja[ia[i-rstart]+j] = (int32_t)cols[j]; va[ia[i-rstart]+j] = PetscRealPart(vals[j]); }
    ierr = MatRestoreRow(A,i,&ncols,&cols,&vals);CHKERRQ(ierr);
  }
  if (h->m) { KSG(ksg_mat_destroy(h->m)); h->m = NULL; }
  ierr = KSGGetContext(PetscObjectComm((PetscObject)A),&ctx);CHKERRQ(ierr);
  KSG(ksg_mat_create_csr_rect(ctx,rend-rstart,M,rstart,cend-cstart,N,cstart,ia,ja,va,&h->m));
  ierr = PetscFree2(ja,va);CHKERRQ(ierr);
  ierr = PetscFree(ia);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

static PetscErrorCode MatDestroy_AIJHIP(Mat A)
{
  PetscErrorCode ierr;
  Mat_AIJHIP     *h;
  PetscErrorCode (*destroy_aij)(Mat);

  PetscFunctionBegin;
  ierr = MatGetHIP(A,&h);CHKERRQ(ierr);
  if (h->m) KSG(ksg_mat_destroy(h->m));
  destroy_aij = h->destroy_aij;
  ierr = PetscFree(h);CHKERRQ(ierr);
  ierr = PetscObjectCompose((PetscObject)A,"ksg_aijhip",NULL);CHKERRQ(ierr);
  ierr = (*destroy_aij)(A);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

PETSC_EXTERN PetscErrorCode MatCreate_AIJHIP(Mat A)
{
  PetscErrorCode ierr;
  Mat_AIJHIP     *h;
  PetscContainer c;

  PetscFunctionBegin;
  ierr = MatSetType(A,MATAIJ);CHKERRQ(ierr);                       /* seqaij or mpiaij by communicator size */
  ierr = PetscNew(&h);CHKERRQ(ierr);
  h->assemblyend_aij = A->ops->assemblyend;
  h->destroy_aij     = A->ops->destroy;
  ierr = PetscContainerCreate(PetscObjectComm((PetscObject)A),&c);CHKERRQ(ierr);
  ierr = PetscContainerSetPointer(c,h);CHKERRQ(ierr);
  ierr = PetscObjectCompose((PetscObject)A,"ksg_aijhip",(PetscObject)c);CHKERRQ(ierr);
  ierr = PetscContainerDestroy(&c);CHKERRQ(ierr);
  A->ops->assemblyend = MatAssemblyEnd_AIJHIP;
  A->ops->mult        = MatMult_AIJHIP;
  A->ops->getvecs     = MatCreateVecs_AIJHIP;
  A->ops->destroy     = MatDestroy_AIJHIP;
  ierr = PetscObjectChangeTypeName((PetscObject)A,"aijhip");CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
