/* DECLARATIONS-ONLY STAND-IN, used by `make -C shim check` and by nothing else.
 *
 * PETSc and SLEPc are not installed where this repository is built, so the three shim files
 * cannot be compiled for real here.  This header declares -- from the documented PETSc 3.7 /
 * SLEPc 3.7 interfaces, written for this check, not taken from either package -- just the types,
 * macros, structure members and functions the shim files name, so that
 * `gcc -fsyntax-only` can prove two things: the files parse as C, and every identifier they use
 * is declared somewhere with a call shape that fits.  It pins nothing about PETSc: a layout, an
 * argument order or a macro that differs in the real headers shows only in a real build. */
#ifndef KSG_SHIM_CHECK_PETSC_DECLS_H
#define KSG_SHIM_CHECK_PETSC_DECLS_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

typedef int PetscErrorCode;
typedef int PetscInt;
typedef int PetscMPIInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef int64_t PetscObjectState;
#define PETSC_EXTERN extern
#define PETSC_DECIDE    (-1)
#define PETSC_DETERMINE PETSC_DECIDE
#define PetscRealPart(a) (a)
#define PetscSqrtReal(a) sqrt(a)

/* MPI */
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Info;
typedef int MPI_Comm_copy_attr_function(MPI_Comm,int,void*,void*,void*,int*);
typedef int MPI_Comm_delete_attr_function(MPI_Comm,int,void*,void*);
#define MPI_SUCCESS 0
#define MPI_BYTE 1
#define MPI_INFO_NULL 0
#define MPI_KEYVAL_INVALID (-1)
#define MPI_COMM_TYPE_SHARED 1
#define MPI_COMM_NULL_COPY_FN ((MPI_Comm_copy_attr_function*)0)
extern MPI_Comm PETSC_COMM_SELF,PETSC_COMM_WORLD;
int MPI_Comm_rank(MPI_Comm,int*);
int MPI_Comm_size(MPI_Comm,int*);
int MPI_Comm_free(MPI_Comm*);
int MPI_Comm_split_type(MPI_Comm,int,int,MPI_Info,MPI_Comm*);
int MPI_Comm_create_keyval(MPI_Comm_copy_attr_function*,MPI_Comm_delete_attr_function*,int*,void*);
int MPI_Comm_set_attr(MPI_Comm,int,void*);
int MPI_Comm_get_attr(MPI_Comm,int,void*,int*);
int MPI_Bcast(void*,int,MPI_Datatype,int,MPI_Comm);

/* errors */
#define PETSC_ERR_SUP 56
#define PETSC_ERR_LIB 76
#define PETSC_ERR_PLIB 77
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_ARG_INCOMP 75
PetscErrorCode PetscError(MPI_Comm,int,const char*,const char*,PetscErrorCode,int,const char*,...);
#define PetscFunctionBegin do { } while (0)
#define PetscFunctionReturn(a) return (a)
#define CHKERRQ(ierr) do { if (ierr) return PetscError(PETSC_COMM_SELF,__LINE__,__func__,__FILE__,ierr,1," "); } while (0)
#define SETERRQ(comm,code,msg) return PetscError(comm,__LINE__,__func__,__FILE__,code,0,msg)
#define SETERRQ1(comm,code,msg,a) return PetscError(comm,__LINE__,__func__,__FILE__,code,0,msg,a)
#define SETERRQ2(comm,code,msg,a,b) return PetscError(comm,__LINE__,__func__,__FILE__,code,0,msg,a,b)

/* objects */
typedef struct _p_PetscObject *PetscObject;
typedef struct _p_PetscContainer *PetscContainer;
typedef struct _p_PetscRandom *PetscRandom;
typedef struct _p_PetscLayout *PetscLayout;
typedef struct _p_VecScatter *VecScatter;
typedef struct _p_PetscSplitReduction PetscSplitReduction;
typedef struct _p_Vec *Vec;
typedef struct _p_Mat *Mat;
typedef const char *VecType;
typedef const char *MatType;
typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3, NORM_1_AND_2 = 4 } NormType;
typedef enum { NOT_SET_VALUES, INSERT_VALUES, ADD_VALUES } InsertMode;
typedef enum { MAT_FLUSH_ASSEMBLY = 1, MAT_FINAL_ASSEMBLY = 0 } MatAssemblyType;
#define VECCUDA "cuda"
#define MATAIJ  "aij"

struct _p_PetscLayout { PetscInt n,N,rstart,rend,bs; };
struct _p_PetscObject { MPI_Comm comm; char *type_name; PetscObjectState state; };

MPI_Comm PetscObjectComm(PetscObject);
PetscErrorCode PetscObjectStateIncrease(PetscObject);
PetscErrorCode PetscObjectCompose(PetscObject,const char*,PetscObject);
PetscErrorCode PetscObjectQuery(PetscObject,const char*,PetscObject*);
PetscErrorCode PetscObjectChangeTypeName(PetscObject,const char*);
PetscErrorCode PetscObjectTypeCompare(PetscObject,const char*,PetscBool*);
PetscErrorCode PetscContainerCreate(MPI_Comm,PetscContainer*);
PetscErrorCode PetscContainerSetPointer(PetscContainer,void*);
PetscErrorCode PetscContainerGetPointer(PetscContainer,void**);
PetscErrorCode PetscContainerDestroy(PetscContainer*);
PetscErrorCode PetscContainerSetUserDestroy(PetscContainer,PetscErrorCode (*)(void*));
PetscErrorCode PetscLayoutSetUp(PetscLayout);
PetscErrorCode PetscLayoutReference(PetscLayout,PetscLayout*);
PetscErrorCode PetscRandomGetSeed(PetscRandom,unsigned long*);
PetscErrorCode PetscRandomSetSeed(PetscRandom,unsigned long);
PetscErrorCode PetscInitialize(int*,char***,const char*,const char*);
PetscErrorCode PetscMemzero(void*,size_t);
PetscErrorCode PetscMemcpy(void*,const void*,size_t);
PetscErrorCode PetscMallocStandIn(size_t,void*);
PetscErrorCode PetscFreeStandIn(void*);
#define PetscMalloc1(n,p)       PetscMallocStandIn((size_t)(n)*sizeof(**(p)),(void*)(p))
#define PetscCalloc1(n,p)       PetscMallocStandIn((size_t)(n)*sizeof(**(p)),(void*)(p))
#define PetscMalloc2(n,p,m,q)   (PetscMallocStandIn((size_t)(n)*sizeof(**(p)),(void*)(p)) || PetscMallocStandIn((size_t)(m)*sizeof(**(q)),(void*)(q)))
#define PetscNew(p)             PetscMallocStandIn(sizeof(**(p)),(void*)(p))
#define PetscNewLog(o,p)        PetscMallocStandIn(sizeof(**(p)),(void*)(p))
#define PetscFree(p)            (PetscFreeStandIn((void*)(p)),(p) = 0,0)
#define PetscFree2(p,q)         (PetscFreeStandIn((void*)(p)),PetscFreeStandIn((void*)(q)),(p) = 0,(q) = 0,0)

/* Vec: the members of the op table and of the object the shim files touch */
struct _VecOps {
  PetscErrorCode (*duplicate)(Vec,Vec*);
  PetscErrorCode (*duplicatevecs)(Vec,PetscInt,Vec**);
  PetscErrorCode (*destroyvecs)(PetscInt,Vec[]);
  PetscErrorCode (*dot)(Vec,Vec,PetscScalar*);
  PetscErrorCode (*tdot)(Vec,Vec,PetscScalar*);
  PetscErrorCode (*norm)(Vec,NormType,PetscReal*);
  PetscErrorCode (*dot_local)(Vec,Vec,PetscScalar*);
  PetscErrorCode (*tdot_local)(Vec,Vec,PetscScalar*);
  PetscErrorCode (*norm_local)(Vec,NormType,PetscReal*);
  PetscErrorCode (*scale)(Vec,PetscScalar);
  PetscErrorCode (*copy)(Vec,Vec);
  PetscErrorCode (*set)(Vec,PetscScalar);
  PetscErrorCode (*axpy)(Vec,PetscScalar,Vec);
  PetscErrorCode (*aypx)(Vec,PetscScalar,Vec);
  PetscErrorCode (*waxpy)(Vec,PetscScalar,Vec,Vec);
  PetscErrorCode (*setrandom)(Vec,PetscRandom);
  PetscErrorCode (*getsize)(Vec,PetscInt*);
  PetscErrorCode (*getlocalsize)(Vec,PetscInt*);
  PetscErrorCode (*getarray)(Vec,PetscScalar**);
  PetscErrorCode (*restorearray)(Vec,PetscScalar**);
  PetscErrorCode (*placearray)(Vec,const PetscScalar*);
  PetscErrorCode (*resetarray)(Vec);
  PetscErrorCode (*destroy)(Vec);
};
struct _p_Vec { struct _p_PetscObject hdr; struct _VecOps ops[1]; PetscLayout map; void *data; PetscBool petscnative; };

PetscErrorCode VecRegister(const char[],PetscErrorCode (*)(Vec));
PetscErrorCode VecCreate(MPI_Comm,Vec*);
PetscErrorCode VecCreateMPI(MPI_Comm,PetscInt,PetscInt,Vec*);
PetscErrorCode VecSetSizes(Vec,PetscInt,PetscInt);
PetscErrorCode VecSetType(Vec,VecType);
PetscErrorCode VecSetBlockSize(Vec,PetscInt);
PetscErrorCode VecGetBlockSize(Vec,PetscInt*);
PetscErrorCode VecGetSize(Vec,PetscInt*);
PetscErrorCode VecGetLocalSize(Vec,PetscInt*);
PetscErrorCode VecGetOwnershipRange(Vec,PetscInt*,PetscInt*);
PetscErrorCode VecDuplicate(Vec,Vec*);
PetscErrorCode VecDuplicateVecs_Default(Vec,PetscInt,Vec**);
PetscErrorCode VecDestroyVecs_Default(PetscInt,Vec[]);
PetscErrorCode VecDestroy(Vec*);
PetscErrorCode VecGetArray(Vec,PetscScalar**);
PetscErrorCode VecGetArrayRead(Vec,const PetscScalar**);
PetscErrorCode VecRestoreArray(Vec,PetscScalar**);
PetscErrorCode VecRestoreArrayRead(Vec,const PetscScalar**);
PetscErrorCode VecPlaceArray(Vec,const PetscScalar*);
PetscErrorCode VecResetArray(Vec);
PetscErrorCode VecCUDAPlaceArray(Vec,PetscScalar*);
PetscErrorCode VecSet(Vec,PetscScalar);
PetscErrorCode VecSetRandom(Vec,PetscRandom);
PetscErrorCode VecScale(Vec,PetscScalar);
PetscErrorCode VecCopy(Vec,Vec);
PetscErrorCode VecAXPY(Vec,PetscScalar,Vec);
PetscErrorCode VecAYPX(Vec,PetscScalar,Vec);
PetscErrorCode VecDot(Vec,Vec,PetscScalar*);
PetscErrorCode VecTDot(Vec,Vec,PetscScalar*);
PetscErrorCode VecNorm(Vec,NormType,PetscReal*);
PetscErrorCode VecDotBegin(Vec,Vec,PetscScalar*);
PetscErrorCode VecNormBegin(Vec,NormType,PetscReal*);

/* Mat */
struct _MatOps {
  PetscErrorCode (*mult)(Mat,Vec,Vec);
  PetscErrorCode (*getvecs)(Mat,Vec*,Vec*);
  PetscErrorCode (*assemblyend)(Mat,MatAssemblyType);
  PetscErrorCode (*destroy)(Mat);
};
struct _p_Mat { struct _p_PetscObject hdr; struct _MatOps ops[1]; PetscLayout rmap,cmap; void *data; };
PetscErrorCode MatRegister(const char[],PetscErrorCode (*)(Mat));
PetscErrorCode MatSetType(Mat,MatType);
PetscErrorCode MatSetFromOptions(Mat);
PetscErrorCode MatSetValues(Mat,PetscInt,const PetscInt[],PetscInt,const PetscInt[],const PetscScalar[],InsertMode);
PetscErrorCode MatAssemblyBegin(Mat,MatAssemblyType);
PetscErrorCode MatGetSize(Mat,PetscInt*,PetscInt*);
PetscErrorCode MatGetLocalSize(Mat,PetscInt*,PetscInt*);
PetscErrorCode MatGetOwnershipRange(Mat,PetscInt*,PetscInt*);
PetscErrorCode MatGetOwnershipRangeColumn(Mat,PetscInt*,PetscInt*);
PetscErrorCode MatGetRow(Mat,PetscInt,PetscInt*,const PetscInt*[],const PetscScalar*[]);
PetscErrorCode MatRestoreRow(Mat,PetscInt,PetscInt*,const PetscInt*[],const PetscScalar*[]);
PetscErrorCode MatMult(Mat,Vec,Vec);
PetscErrorCode MatMult_SeqAIJ(Mat,Vec,Vec);
PetscErrorCode MatMult_MPIAIJ(Mat,Vec,Vec);
PetscErrorCode MatCreateVecs(Mat,Vec*,Vec*);
PetscErrorCode MatDenseGetArray(Mat,PetscScalar**);
PetscErrorCode MatDenseRestoreArray(Mat,PetscScalar**);
#endif
