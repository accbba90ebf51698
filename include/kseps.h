/*
 * kseps.h -- C ABI of libkseps, the host side of the Krylov-Schur hot path.
 *
 * A small, PETSc-free mirror of the interfaces the reference (SLEPc 3.7.3-dev)
 * exposes on this path: the EPS solver object with its Krylov-Schur drivers,
 * the BV class with its type registry and op table, the DS projected problems
 * (HEP compact, NHEP), and the handful of Vec/Mat entry points the solver
 * calls.  Names, argument meaning and error behaviour follow the reference;
 * every function returns a PetscErrorCode (0 = success) and the message of the
 * last error is available from KSGetLastError().  The compute back end is
 * libksgpu (include/ksgpu.h); there is no CPU implementation of the vector
 * kernels in this library.  Paths in comments are relative to the reference
 * root.
 */
#ifndef KSEPS_H
#define KSEPS_H

#include <stdint.h>

#include "ksgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int PetscErrorCode;
typedef int PetscInt;
typedef int64_t PetscInt64;
typedef double PetscScalar;
typedef double PetscReal;
typedef int PetscBool;
#define PETSC_TRUE 1
#define PETSC_FALSE 0

/* error codes (values of PETSc's petscerror.h) */
#define PETSC_ERR_MEM            55
#define PETSC_ERR_SUP            56
#define PETSC_ERR_ORDER          58
#define PETSC_ERR_ARG_SIZ        60
#define PETSC_ERR_ARG_WRONG      62
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_ARG_INCOMP     75
#define PETSC_ERR_ARG_NULL       85
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_LIB            76
#define PETSC_ERR_PLIB           77
#define PETSC_ERR_USER           83
#define PETSC_ERR_ARG_UNKNOWN_TYPE 86
#define PETSC_ERR_FILE_OPEN      65
#define PETSC_ERR_FILE_READ      66
#define PETSC_ERR_FILE_WRITE     67
#define PETSC_ERR_FILE_UNEXPECTED 79
#define PETSC_ERR_ARG_IDN         61
#define PETSC_ERR_MAT_LU_ZRPVT    71
#define PETSC_ERR_NOT_CONVERGED   82

const char *KSGetLastError(void);

typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3 } NormType;

typedef struct _p_Vec *Vec;
typedef struct _p_Mat *Mat;
typedef struct _p_BV *BV;
typedef struct _p_DS *DS;
typedef struct _p_EPS *EPS;
typedef struct _p_SVD *SVD;
typedef struct _p_ST *ST;

/* ------------------------------------------------------------------ Vec */
/* Op table of a vector type: the PETSc Vec operations the hot-path host code
 * invokes on BV column views (VecCopy bvbasic.c:1645, VecAXPY epssolve.c:531,
 * VecNorm epssolve.c:533, VecDot bvglobal.c:477, VecScale/VecSet
 * epsdefault.c:374,398).  `array` is an address in the memory of the back end
 * (device memory for the "hip" type); the host code never dereferences it. */
struct _VecOps {
  PetscErrorCode (*alloc)(Vec v);                       /* allocate v->array for v->n entries */
  PetscErrorCode (*free)(Vec v);
  PetscErrorCode (*copy)(Vec x, Vec y);
  PetscErrorCode (*set)(Vec x, PetscScalar a);
  PetscErrorCode (*scale)(Vec x, PetscScalar a);
  PetscErrorCode (*axpy)(Vec y, PetscScalar a, Vec x);
  PetscErrorCode (*dot)(Vec x, Vec y, PetscScalar *r);  /* collective */
  PetscErrorCode (*norm2)(Vec x, PetscReal *r);         /* collective */
  PetscErrorCode (*fillhash)(Vec x, uint64_t seed);     /* x[i] = f(seed, row0+i) */
  PetscErrorCode (*tohost)(Vec x, PetscScalar *host);   /* copy the local entries out */
  PetscErrorCode (*fromhost)(Vec x, const PetscScalar *host);
  /* optional (may all be NULL): split-phase reductions of the back end.  redopen starts a
   * batch on the context, dotpost queues x^T y and returns the index of its value, redclose
   * reduces everything queued (BV posts included) at once and delivers the values. */
  PetscErrorCode (*redopen)(void *ctx);
  PetscErrorCode (*dotpost)(Vec x, Vec y, PetscInt *offset);
  PetscErrorCode (*redclose)(void *ctx, PetscScalar *out, PetscInt count);
};
struct _p_Vec {
  const struct _VecOps *ops;
  PetscInt64 n, N, row0;
  PetscScalar *array;       /* current storage */
  PetscScalar *own;         /* storage owned by the vector (NULL for pure views) */
  PetscInt64 state;         /* bumped on every modification */
  void *ctx;                /* back-end context (ksg_ctx for "hip") */
  PetscInt64 id;            /* unique per object (PetscObjectId): with state, the key of the B*x cache of a BV */
};
PetscErrorCode VecCreateHIP(ksg_ctx ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0, Vec *v);
PetscErrorCode VecCreateWithOps(const struct _VecOps *ops, void *ctx, PetscInt64 n, PetscInt64 N,
                                PetscInt64 row0, PetscBool allocate, Vec *v);
PetscErrorCode VecDuplicate(Vec v, Vec *w);
PetscErrorCode VecDestroy(Vec *v);
PetscErrorCode VecPlaceArray(Vec v, PetscScalar *array);   /* svec.c:322 */
PetscErrorCode VecResetArray(Vec v);                       /* svec.c:334 */
PetscErrorCode VecCopy(Vec x, Vec y);
PetscErrorCode VecSet(Vec x, PetscScalar a);
PetscErrorCode VecScale(Vec x, PetscScalar a);
PetscErrorCode VecAXPY(Vec y, PetscScalar a, Vec x);
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *r);
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *r);
PetscErrorCode VecDotBegin(Vec x, Vec y, PetscScalar *r);
PetscErrorCode VecDotEnd(Vec x, Vec y, PetscScalar *r);
PetscErrorCode VecNormBegin(Vec x, NormType type, PetscReal *r);
PetscErrorCode VecNormEnd(Vec x, NormType type, PetscReal *r);
PetscErrorCode VecGetSize(Vec x, PetscInt64 *N);
PetscErrorCode VecGetLocalSize(Vec x, PetscInt64 *n);
PetscErrorCode VecGetValuesHost(Vec x, PetscScalar *host);         /* stands in for VecGetArray */
PetscErrorCode VecSetValuesHost(Vec x, const PetscScalar *host);
PetscErrorCode VecFillHash(Vec x, uint64_t seed);

/* ------------------------------------------------------------------ Mat */
/* The solver needs MatMult (shift.c:38, epssolve.c:527), MatCreateVecs
 * (stfunc.c:546) and the sizes (epssetup.c:135-136). */
struct _MatOps {
  PetscErrorCode (*mult)(Mat A, Vec x, Vec y);
  PetscErrorCode (*createvecs)(Mat A, Vec *right, Vec *left);
  PetscErrorCode (*destroy)(Mat A);
  /* optional (may be NULL): *norm = ||A*xr - kr*xr + ki*xi||_2 in one pass, xi may be NULL;
   * `work` is a scratch vector the operator may use.  EPSComputeError falls back to
   * MatMult + VecAXPY + VecNorm (epssolve.c:506-556) without it. */
  PetscErrorCode (*residualnorm)(Mat A, PetscScalar kr, PetscScalar ki, Vec xr, Vec xi, Vec work, PetscReal *norm);
  /* optional (may be NULL): *AT = A^T as a new matrix (MatTranspose with MAT_INITIAL_MATRIX,
   * what SVDSetUp needs, svdsetup.c:139-148) */
  PetscErrorCode (*transpose)(Mat A, Mat *AT);
};
struct _p_Mat {
  struct _MatOps ops;
  PetscInt64 n, N, row0;    /* rows: local, global, first local */
  void *data;               /* ksg_mat for "aijhip", user context for shells */
  void *ctx;
  PetscInt64 cn, cN, col0;  /* columns (layout of the vector a MatMult takes); equal to the rows for a square operator */
};
/* GPU AIJ operator: the rows [row0,row0+n) in CSR with global column indices
 * (the role of MatCreateAIJ + MatSetValues + MatAssembly with -mat_type set to
 * the device type, ex2.c:50-66). */
PetscErrorCode MatCreateAIJHIP(ksg_ctx ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0,
                               const int64_t *rowptr, const int32_t *colidx, const double *val, Mat *A);
/* The same for a rectangular operator, M x N with rows [row0,row0+m) and the columns
 * [col0,col0+n) local (MatSetSizes(A,m,n,M,N), src/svd/examples/tutorials/ex15.c). */
PetscErrorCode MatCreateAIJHIPRect(ksg_ctx ctx, PetscInt64 m, PetscInt64 M, PetscInt64 row0, PetscInt64 n, PetscInt64 N,
                                   PetscInt64 col0, const int64_t *rowptr, const int32_t *colidx, const double *val, Mat *A);
PetscErrorCode MatTranspose(Mat A, Mat *AT);
/* User operator (MatCreateShell + MatShellSetOperation(MATOP_MULT), tests/test8.c:57-61).
 * createvecs may be NULL when template vectors are supplied by another route. */
PetscErrorCode MatCreateShell(void *ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0, void *user,
                              PetscErrorCode (*mult)(Mat, Vec, Vec),
                              PetscErrorCode (*createvecs)(Mat, Vec *, Vec *), Mat *A);
PetscErrorCode MatShellGetContext(Mat A, void **user);
/* for a rectangular shell: the layout of the vectors MatMult takes, and the routine that
 * builds the transpose (the role of MatShellSetOperation(A,MATOP_TRANSPOSE,...)) */
PetscErrorCode MatShellSetColumnLayout(Mat A, PetscInt64 n, PetscInt64 N, PetscInt64 col0);
PetscErrorCode MatShellSetTranspose(Mat A, PetscErrorCode (*transpose)(Mat, Mat *));
/* PETSc binary matrix files (the format MatLoad reads and MatView writes on a
 * PetscViewerBinary; src/eps/examples/tutorials/ex4.c:56-60 loads the files of
 * share/slepc/datafiles/matrices this way).  KSMatReadPetscBinary returns the whole
 * matrix in CSR on the host, every row in ascending column order; the three arrays are
 * released with KSFree.  MatLoadAIJHIP gives rank `rank` of `nranks` its contiguous row
 * block (the first M mod nranks ranks hold one row more, as MatLoad distributes them)
 * as a device operator. */
PetscErrorCode KSMatReadPetscBinary(const char *filename, PetscInt64 *M, PetscInt64 *N, int64_t **rowptr,
                                    int32_t **colidx, double **val);
PetscErrorCode KSMatWritePetscBinary(const char *filename, PetscInt64 M, PetscInt64 N, const int64_t *rowptr,
                                     const int32_t *colidx, const double *val);
void KSFree(void *p);
PetscErrorCode MatLoadAIJHIP(ksg_ctx ctx, const char *filename, int rank, int nranks, Mat *A);
PetscErrorCode MatMult(Mat A, Vec x, Vec y);
PetscErrorCode MatCreateVecs(Mat A, Vec *right, Vec *left);
PetscErrorCode MatGetSize(Mat A, PetscInt64 *M, PetscInt64 *N);
PetscErrorCode MatGetLocalSize(Mat A, PetscInt64 *m, PetscInt64 *n);
PetscErrorCode MatDestroy(Mat *A);

/* ------------------------------------------------------------------ BV */
typedef enum { BV_ORTHOG_CGS = 0, BV_ORTHOG_MGS = 1 } BVOrthogType;
typedef enum { BV_ORTHOG_REFINE_IFNEEDED = 0, BV_ORTHOG_REFINE_NEVER = 1, BV_ORTHOG_REFINE_ALWAYS = 2 } BVOrthogRefineType;
typedef enum { BV_ORTHOG_BLOCK_GS = 0, BV_ORTHOG_BLOCK_CHOL = 1, BV_ORTHOG_BLOCK_TSQR = 2 } BVOrthogBlockType;

/* Op table of a BV type: struct _BVOps of include/slepc/private/bvimpl.h:35-64.
 * Q matrices are host column-major arrays with leading dimension ldq. */
struct _BVOps {
  PetscErrorCode (*mult)(BV Y, PetscScalar alpha, PetscScalar beta, BV X, const PetscScalar *Q, PetscInt ldq);
  PetscErrorCode (*multvec)(BV X, PetscScalar alpha, PetscScalar beta, Vec y, PetscScalar *q);
  PetscErrorCode (*multinplace)(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e);
  PetscErrorCode (*multinplacetrans)(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e);
  PetscErrorCode (*dot)(BV X, BV Y, PetscScalar *M, PetscInt ldm);
  PetscErrorCode (*dotvec)(BV X, Vec y, PetscScalar *q);
  PetscErrorCode (*dotvec_local)(BV X, Vec y, PetscScalar *q);
  PetscErrorCode (*scale)(BV bv, PetscInt j, PetscScalar alpha);
  PetscErrorCode (*norm)(BV bv, PetscInt j, NormType type, PetscReal *val);
  PetscErrorCode (*norm_local)(BV bv, PetscInt j, NormType type, PetscReal *val);
  PetscErrorCode (*matmult)(BV V, Mat A, BV W);
  PetscErrorCode (*copy)(BV V, BV W);
  PetscErrorCode (*resize)(BV bv, PetscInt m, PetscBool copy);
  PetscErrorCode (*getcolumn)(BV bv, PetscInt j, Vec *v);
  PetscErrorCode (*restorecolumn)(BV bv, PetscInt j, Vec *v);
  PetscErrorCode (*destroy)(BV bv);
  /* optional (may be NULL): the first half of a split-phase dotvec (dotvec_begin of
   * bvimpl.h:35-64) -- queues X(:,l:k)^T y in the batch opened on the context of the template
   * vector and returns the index of its first value */
  PetscErrorCode (*dotvec_post)(BV X, Vec y, PetscInt *offset);
  /* optional (may be NULL): QR factorisation of columns [0,k) by a method that forms no Gram
   * matrix, V <- Q, R (ldr x k, may be NULL) -- what BVOrthogonalize_TSQR (bvorthog.c:770-782)
   * does on the host array of the basis */
  PetscErrorCode (*orthogonalize_tsqr)(BV V, PetscScalar *R, PetscInt ldr);
};

/* struct _p_BV of bvimpl.h:66-105 (the fields this path uses). */
struct _p_BV {
  struct _BVOps ops;
  char type_name[32];
  Vec t;                        /* template vector */
  PetscInt64 n, N;              /* local / global rows */
  PetscInt m, l, k, nc;         /* columns, active range, constraints */
  BVOrthogType orthog_type;
  BVOrthogRefineType orthog_ref;
  PetscReal orthog_eta;
  PetscScalar *buffer;          /* host, (nc+m)*m, column 0 = scratch, column j = h_j (bvbasic.c:791-827) */
  PetscScalar *h, *c;           /* host coefficient arrays for BVOrthogonalizeVec */
  Vec cv[2];                    /* column views */
  PetscInt ci[2];               /* column index of each outstanding view, or -nc-1 */
  PetscInt64 st[2];             /* state of the view when it was handed out */
  PetscInt64 state;
  uint64_t rand_seed;
  void *data;                   /* type-specific (ksg_bv for "hip") */
  void *ctx;
  BVOrthogBlockType orthog_block;   /* method of BVOrthogonalize */
  /* non-standard inner product <x,y> = y^T B x (bvimpl.h:80-88) */
  Mat matrix;                   /* B, borrowed: it must outlive the BV (NULL: standard inner product) */
  PetscBool indef;              /* B is indefinite: signed norms, signature omega */
  Vec Bx;                       /* B*x of the vector whose id/state are xid/xstate (BV_IPMatMult, bvimpl.h:140-151) */
  PetscInt64 xid, xstate;
  BV cached;                    /* B*X of the active columns at state bvstate (BV_IPMatMultBV, bvimpl.h:157-173) */
  PetscInt64 bvstate;
  PetscReal *omega;             /* signature, nc+m entries (indefinite case) */
};

PetscErrorCode BVCreate(void *ctx, BV *bv);
/* Registers a BV type: BVRegister(name, constructor), include/slepcbv.h:196,
 * bvfunc.c:677-684.  The constructor runs when type and sizes are both known
 * (bvbasic.c:64-70,193-197); on entry bv->t, n, N, m, k are set. */
PetscErrorCode BVRegister(const char *name, PetscErrorCode (*create)(BV));
PetscErrorCode BVSetType(BV bv, const char *type);
PetscErrorCode BVSetSizesFromVec(BV bv, Vec t, PetscInt m);
PetscErrorCode BVGetSizes(BV bv, PetscInt64 *n, PetscInt64 *N, PetscInt *m);
PetscErrorCode BVResize(BV bv, PetscInt m, PetscBool copy);
/* Constraints (bvfunc.c:414-442, bvbasic.c:262-321): vectors every orthogonalisation also
 * deflates; stored in front of the columns, reached as columns -nc..-1.  C holds *nc vectors;
 * linearly dependent ones are dropped and *nc is lowered.  Destroys the contents of V. */
PetscErrorCode BVInsertConstraints(BV V, PetscInt *nc, Vec *C);
PetscErrorCode BVSetNumConstraints(BV V, PetscInt nc);
PetscErrorCode BVGetNumConstraints(BV bv, PetscInt *nc);
/* Inner product <x,y> = y^T B x (bvbasic.c:481-646).  B is borrowed, not reference counted:
 * it must outlive the BV.  BVDot, BVDotVec/Column, BVNormVec/Column and every
 * orthogonalisation then work in that inner product; with indef the norms carry the sign of
 * x^T B x and the orthogonalisations maintain the signature (bvorthog.c:255-258,352). */
PetscErrorCode BVSetMatrix(BV bv, Mat B, PetscBool indef);
PetscErrorCode BVGetMatrix(BV bv, Mat *B, PetscBool *indef);
PetscErrorCode BVApplyMatrix(BV bv, Vec x, Vec y);
PetscErrorCode BVApplyMatrixBV(BV X, BV Y);                  /* Y may be NULL: result in the cached BV */
PetscErrorCode BVGetCachedBV(BV bv, BV *cached);
/* signature of the active columns 0..k-1 as a host array of k entries (a sequential Vec in
 * the reference, bvbasic.c:664-729) */
PetscErrorCode BVSetSignature(BV bv, const PetscReal *omega);
PetscErrorCode BVGetSignature(BV bv, PetscReal *omega);
PetscErrorCode BVDuplicate(BV V, BV *W);                     /* bvbasic.c:1432-1485: type, sizes, matrix, orthogonalisation */
PetscErrorCode BVDuplicateResize(BV V, PetscInt m, BV *W);
PetscErrorCode BVSetActiveColumns(BV bv, PetscInt l, PetscInt k);
PetscErrorCode BVGetActiveColumns(BV bv, PetscInt *l, PetscInt *k);
/* bvorthog.c: eta 0 stands for PETSC_DEFAULT (0.7071) */
PetscErrorCode BVSetOrthogonalization(BV bv, BVOrthogType type, BVOrthogRefineType refine, PetscReal eta, BVOrthogBlockType block);
PetscErrorCode BVGetColumn(BV bv, PetscInt j, Vec *v);
PetscErrorCode BVRestoreColumn(BV bv, PetscInt j, Vec *v);
PetscErrorCode BVGetBuffer(BV bv, PetscScalar **buffer, PetscInt *ld);     /* BVGetBufferVec */
PetscErrorCode BVInsertVec(BV bv, PetscInt j, Vec w);
PetscErrorCode BVInsertVecs(BV bv, PetscInt s, PetscInt *m, Vec *W, PetscBool orth);
PetscErrorCode BVSetRandomColumn(BV bv, PetscInt j);
PetscErrorCode BVCopy(BV V, BV W);
PetscErrorCode BVCopyVec(BV V, PetscInt j, Vec w);
PetscErrorCode BVCopyColumn(BV V, PetscInt j, PetscInt i);
PetscErrorCode BVMult(BV Y, PetscScalar alpha, PetscScalar beta, BV X, const PetscScalar *Q, PetscInt ldq);
PetscErrorCode BVMultVec(BV X, PetscScalar alpha, PetscScalar beta, Vec y, PetscScalar *q);
PetscErrorCode BVMultColumn(BV X, PetscScalar alpha, PetscScalar beta, PetscInt j, PetscScalar *q);
PetscErrorCode BVMultInPlace(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e);
PetscErrorCode BVMultInPlaceTranspose(BV V, const PetscScalar *Q, PetscInt ldq, PetscInt s, PetscInt e);
PetscErrorCode BVDot(BV X, BV Y, PetscScalar *M, PetscInt ldm);
PetscErrorCode BVDotVec(BV X, Vec y, PetscScalar *q);
PetscErrorCode BVDotColumn(BV X, PetscInt j, PetscScalar *q);
/* Split-phase reductions (bvglobal.c:190-310,361-470,732-830; VecDotBegin/End of PETSc): all
 * the ...Begin calls of a group first, then the matching ...End calls in the same order, each
 * with the object (and, for a BV, the active columns) of its Begin; the result is written to
 * the array given to the End.  The first End completes the whole group with one reduction
 * round.  PETSC_ERR_ORDER: a Begin after an End of an unfinished group;
 * PETSC_ERR_ARG_WRONGSTATE: an End out of order or without Begin (the group is dropped).
 * Types without the optional post ops compute at Begin and deliver at End. */
PetscErrorCode BVDotVecBegin(BV X, Vec y, PetscScalar *m);
PetscErrorCode BVDotVecEnd(BV X, Vec y, PetscScalar *m);
PetscErrorCode BVDotColumnBegin(BV X, PetscInt j, PetscScalar *m);
PetscErrorCode BVDotColumnEnd(BV X, PetscInt j, PetscScalar *m);
PetscErrorCode BVNormVecBegin(BV bv, Vec v, NormType type, PetscReal *val);
PetscErrorCode BVNormVecEnd(BV bv, Vec v, NormType type, PetscReal *val);
PetscErrorCode BVNormColumnBegin(BV bv, PetscInt j, NormType type, PetscReal *val);
PetscErrorCode BVNormColumnEnd(BV bv, PetscInt j, NormType type, PetscReal *val);
PetscErrorCode BVScale(BV bv, PetscScalar alpha);
PetscErrorCode BVScaleColumn(BV bv, PetscInt j, PetscScalar alpha);
PetscErrorCode BVNorm(BV bv, NormType type, PetscReal *val);
PetscErrorCode BVNormColumn(BV bv, PetscInt j, NormType type, PetscReal *val);
PetscErrorCode BVNormVec(BV bv, Vec v, NormType type, PetscReal *val);
PetscErrorCode BVMatMult(BV V, Mat A, BV W);
PetscErrorCode BVOrthogonalizeVec(BV bv, Vec v, PetscScalar *H, PetscReal *norm, PetscBool *lindep);
PetscErrorCode BVOrthogonalizeColumn(BV bv, PetscInt j, PetscScalar *H, PetscReal *norm, PetscBool *lindep);
/* against the columns i < j with which[i] set only; modified Gram-Schmidt (bvorthog.c:587-633) */
PetscErrorCode BVOrthogonalizeSomeColumn(BV bv, PetscInt j, const PetscBool *which, PetscScalar *H, PetscReal *norm, PetscBool *lindep);
PetscErrorCode BVOrthonormalizeColumn(BV bv, PetscInt j, PetscBool replace, PetscReal *norm, PetscBool *lindep);
/* QR factorisation of the active columns (bvorthog.c:768-859): V <- Q with V0 = Q R.  R is a
 * host column-major array with leading dimension ldr >= k (only the block [l,k) x [l,k) is
 * referenced) or NULL.  Block method as set with BVSetOrthogonalization: column by column
 * Gram-Schmidt (BVOrthogonalize_GS, :635-680) or Cholesky QR, R = chol(V'V), Q = V inv(R)
 * (BVOrthogonalize_Chol, :749-765: one Gram kernel and one block-update kernel).  The
 * tall-skinny QR variant is not available. */
PetscErrorCode BVOrthogonalize(BV V, PetscScalar *R, PetscInt ldr);
PetscErrorCode BVDestroy(BV *bv);
/* Constructor of the device type, registered as "hip" (the role BVCreate_Svec plays
 * for "svec", svec.c:417-520). */
PetscErrorCode BVCreate_HIP(BV bv);

/* ------------------------------------------------------------------ DS */
typedef enum { DS_STATE_RAW = 0, DS_STATE_INTERMEDIATE, DS_STATE_CONDENSED, DS_STATE_TRUNCATED } DSStateType;
typedef enum { DS_MAT_A = 0, DS_MAT_Q, DS_MAT_X, DS_MAT_T, DS_MAT_U, DS_MAT_VT, DS_NUM_MAT } DSMatType;
typedef PetscErrorCode (*SlepcCompareFn)(PetscScalar ar, PetscScalar ai, PetscScalar br, PetscScalar bi, PetscInt *res, void *ctx);

PetscErrorCode DSCreate(DS *ds);
PetscErrorCode DSSetType(DS ds, const char *type);          /* "hep", "nhep" or "svd" */
PetscErrorCode DSSetCompact(DS ds, PetscBool comp);
PetscErrorCode DSSetExtraRow(DS ds, PetscBool ext);
PetscErrorCode DSAllocate(DS ds, PetscInt ld);
PetscErrorCode DSSetSlepcSC(DS ds, SlepcCompareFn cmp, void *ctx);
PetscErrorCode DSSetDimensions(DS ds, PetscInt n, PetscInt m, PetscInt l, PetscInt k);
PetscErrorCode DSGetDimensions(DS ds, PetscInt *n, PetscInt *m, PetscInt *l, PetscInt *k, PetscInt *t);
PetscErrorCode DSGetLeadingDimension(DS ds, PetscInt *ld);
PetscErrorCode DSSetState(DS ds, DSStateType state);
PetscErrorCode DSGetState(DS ds, DSStateType *state);
PetscErrorCode DSGetArray(DS ds, DSMatType m, PetscScalar **a);
PetscErrorCode DSGetArrayReal(DS ds, DSMatType m, PetscReal **a);
PetscErrorCode DSSolve(DS ds, PetscScalar *eigr, PetscScalar *eigi);
PetscErrorCode DSSort(DS ds, PetscScalar *eigr, PetscScalar *eigi, PetscScalar *rr, PetscScalar *ri, PetscInt *k);
PetscErrorCode DSUpdateExtraRow(DS ds);
PetscErrorCode DSVectors(DS ds, DSMatType mat, PetscInt *j, PetscReal *rnorm);
PetscErrorCode DSTruncate(DS ds, PetscInt n);
/* DSGetMat (dsops.c:280-315): copies the rows x cols leading part into `out`
 * (column-major, leading dimension *rows). */
PetscErrorCode DSGetMat(DS ds, DSMatType m, PetscScalar *out, PetscInt *rows, PetscInt *cols);
PetscErrorCode DSDestroy(DS *ds);

PetscErrorCode SlepcCompareLargestMagnitude(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcCompareSmallestMagnitude(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcCompareLargestReal(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcCompareSmallestReal(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcCompareLargestImaginary(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcCompareSmallestImaginary(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcCompareSmallestPosReal(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
/* closest to the target, ctx = PetscScalar *target (slepcsc.c:225-262) */
PetscErrorCode SlepcCompareTargetMagnitude(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcCompareTargetReal(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscInt *, void *);
PetscErrorCode SlepcSortEigenvalues(SlepcCompareFn cmp, void *ctx, PetscInt n, PetscScalar *eigr, PetscScalar *eigi, PetscInt *perm);

/* ------------------------------------------------------------------ EPS */
/* epsopts.c:946-989.  EPS_GHIEP (indefinite B) is not available. */
typedef enum { EPS_HEP = 1, EPS_GHEP = 2, EPS_NHEP = 3, EPS_GNHEP = 4, EPS_PGNHEP = 5 } EPSProblemType;
typedef enum { EPS_LARGEST_MAGNITUDE = 1, EPS_SMALLEST_MAGNITUDE, EPS_LARGEST_REAL, EPS_SMALLEST_REAL,
               EPS_LARGEST_IMAGINARY, EPS_SMALLEST_IMAGINARY, EPS_TARGET_MAGNITUDE, EPS_TARGET_REAL,
               EPS_WHICH_USER = 11 } EPSWhich;
typedef enum { EPS_ERROR_ABSOLUTE = 0, EPS_ERROR_RELATIVE = 1 } EPSErrorType;
typedef enum { EPS_CONVERGED_TOL = 1, EPS_CONVERGED_USER = 2, EPS_DIVERGED_ITS = -1, EPS_DIVERGED_BREAKDOWN = -2,
               EPS_DIVERGED_SYMMETRY_LOST = -3, EPS_CONVERGED_ITERATING = 0 } EPSConvergedReason;
typedef enum { EPS_CONV_ABS = 0, EPS_CONV_REL = 1 } EPSConv;
typedef PetscErrorCode (*EPSMonitorFn)(EPS eps, PetscInt its, PetscInt nconv, PetscScalar *eigr, PetscScalar *eigi,
                                       PetscReal *errest, PetscInt nest, void *ctx);

PetscErrorCode EPSCreate(void *ctx, EPS *eps);                                   /* epsbasic.c:60-80 */
PetscErrorCode EPSSetOperators(EPS eps, Mat A, Mat B);                           /* B == NULL: standard problem */
/* "krylovschur" (default; krylovschur.c, ks-symm.c), "arnoldi" (explicit restart,
 * src/eps/impls/krylov/arnoldi/arnoldi.c) or "lanczos" (explicit restart, symmetric problems,
 * src/eps/impls/krylov/lanczos/lanczos.c; reorthogonalisation chosen with EPSLanczosSetReorthog) */
PetscErrorCode EPSSetType(EPS eps, const char *type);                            /* epsbasic.c */
PetscErrorCode EPSGetType(EPS eps, const char **type);
PetscErrorCode EPSSetProblemType(EPS eps, EPSProblemType type);                  /* epsopts.c:946-989 */
PetscErrorCode EPSSetTarget(EPS eps, PetscScalar target);                        /* epsopts.c (EPSSetTarget): also the default shift of sinvert */
PetscErrorCode EPSGetTarget(EPS eps, PetscScalar *target);
PetscErrorCode EPSGetST(EPS eps, ST *st);
PetscErrorCode EPSSetPurify(EPS eps, PetscBool purify);                          /* epsopts.c:1327 */
/* vectors every basis vector is kept orthogonal to (epssetup.c:264-270; copies are kept) */
PetscErrorCode EPSSetDeflationSpace(EPS eps, PetscInt n, Vec *v);

/* ---- ST: spectral transformation (src/sys/classes/st) ------------------------------------
 * "shift" (default): OP = A - sigma I, or B^-1 (A - sigma B); "sinvert": OP = (A - sigma I)^-1,
 * or (A - sigma B)^-1 B.  The linear systems are solved by "preonly" (dense LU on the host,
 * one process and at most 4096 rows: the stand-in for the reference's KSPPREONLY + PCLU and
 * the default when it applies), "cg" or "gmres" (on the device vectors; gmres is the default
 * otherwise). */
PetscErrorCode STCreate(void *ctx, ST *st);
PetscErrorCode STDestroy(ST *st);
PetscErrorCode STSetType(ST st, const char *type);
PetscErrorCode STGetType(ST st, const char **type);
PetscErrorCode STSetOperators(ST st, Mat A, Mat B);
PetscErrorCode STGetNumMatrices(ST st, PetscInt *n);
PetscErrorCode STSetShift(ST st, PetscScalar shift);
PetscErrorCode STGetShift(ST st, PetscScalar *shift);
PetscErrorCode STSetDefaultShift(ST st, PetscScalar defaultshift);
PetscErrorCode STSetKSPType(ST st, const char *type);
PetscErrorCode STGetKSPType(ST st, const char **type);
PetscErrorCode STSetKSPTolerances(ST st, PetscReal rtol, PetscInt maxit, PetscInt restart);    /* 0 keeps a value */
PetscErrorCode STGetKSPIterations(ST st, PetscInt64 *its, PetscInt64 *solves);
PetscErrorCode STGetBilinearForm(ST st, Mat *B);
PetscErrorCode STSetUp(ST st);
PetscErrorCode STApply(ST st, Vec x, Vec y);                                     /* stsolve.c:44-72 */
PetscErrorCode STMatSolve(ST st, Vec b, Vec x);                                  /* stsles.c */
PetscErrorCode STBackTransform(ST st, PetscInt n, PetscScalar *eigr, PetscScalar *eigi);
PetscErrorCode EPSSetDimensions(EPS eps, PetscInt nev, PetscInt ncv, PetscInt mpd);
PetscErrorCode EPSSetTolerances(EPS eps, PetscReal tol, PetscInt maxits);
PetscErrorCode EPSSetWhichEigenpairs(EPS eps, EPSWhich which);
/* user ordering of the eigenvalues (epsopts.c, EPSSetEigenvalueComparison); sets EPS_WHICH_USER */
PetscErrorCode EPSSetEigenvalueComparison(EPS eps, SlepcCompareFn cmp, void *ctx);
PetscErrorCode EPSSetConvergenceTest(EPS eps, EPSConv conv);                     /* epsopts.c; default relative */
PetscErrorCode EPSSetInitialSpace(EPS eps, PetscInt n, Vec *is);                 /* epssetup.c:449-460 */
/* reorthogonalisation of the "lanczos" type (lanczos.c:740-790; the default is local) */
typedef enum { EPS_LANCZOS_REORTHOG_LOCAL = 0, EPS_LANCZOS_REORTHOG_FULL, EPS_LANCZOS_REORTHOG_SELECTIVE,
               EPS_LANCZOS_REORTHOG_PERIODIC, EPS_LANCZOS_REORTHOG_PARTIAL, EPS_LANCZOS_REORTHOG_DELAYED } EPSLanczosReorthogType;
PetscErrorCode EPSLanczosSetReorthog(EPS eps, EPSLanczosReorthogType reorthog);
PetscErrorCode EPSKrylovSchurSetRestart(EPS eps, PetscReal keep);                /* krylovschur.c:290 */
PetscErrorCode EPSKrylovSchurSetLocking(EPS eps, PetscBool lock);
/* Arnoldi with delayed reorthogonalisation (-eps_arnoldi_delayed, arnoldi.c:168-175;
 * EPSDelayedArnoldi / EPSDelayedArnoldi1 of epskrylov.c:82-233, the latter when the BV
 * refinement is BV_ORTHOG_REFINE_NEVER).  As the reference warns, it may be unstable. */
PetscErrorCode EPSArnoldiSetDelayed(EPS eps, PetscBool delayed);
PetscErrorCode EPSMonitorSet(EPS eps, EPSMonitorFn monitor, void *ctx);
PetscErrorCode EPSGetBV(EPS eps, BV *bv);
PetscErrorCode EPSGetDS(EPS eps, DS *ds);
PetscErrorCode EPSSetUp(EPS eps);                                                /* epssetup.c:100-288 */
PetscErrorCode EPSSolve(EPS eps);                                                /* epssolve.c:63-146 */
PetscErrorCode EPSGetIterationNumber(EPS eps, PetscInt *its);
PetscErrorCode EPSGetConverged(EPS eps, PetscInt *nconv);
PetscErrorCode EPSGetConvergedReason(EPS eps, EPSConvergedReason *reason);
PetscErrorCode EPSGetDimensions(EPS eps, PetscInt *nev, PetscInt *ncv, PetscInt *mpd);
PetscErrorCode EPSGetTolerances(EPS eps, PetscReal *tol, PetscInt *maxits);
PetscErrorCode EPSGetEigenvalue(EPS eps, PetscInt i, PetscScalar *eigr, PetscScalar *eigi);
PetscErrorCode EPSGetEigenvector(EPS eps, PetscInt i, Vec Vr, Vec Vi);
PetscErrorCode EPSGetEigenpair(EPS eps, PetscInt i, PetscScalar *eigr, PetscScalar *eigi, Vec Vr, Vec Vi);
PetscErrorCode EPSGetErrorEstimate(EPS eps, PetscInt i, PetscReal *errest);
PetscErrorCode EPSComputeError(EPS eps, PetscInt i, EPSErrorType type, PetscReal *error);
/* number of Gram-Schmidt passes and operator applications so far (for the byte model) */
PetscErrorCode EPSGetCounters(EPS eps, PetscInt64 *matvecs, PetscInt64 *gs_passes);
PetscErrorCode EPSDestroy(EPS *eps);

/* ------------------------------------------------------------------ SVD */
/* Singular value solvers on the same kernels (include/slepcsvd.h of the reference):
 * "cross" (default, svdsetup.c:112-114; the eigenproblem of A^T A, src/svd/impls/cross/cross.c)
 * "trlanczos" (thick-restart Lanczos bidiagonalisation, src/svd/impls/trlanczos/trlanczos.c),
 * "lanczos" (explicitly restarted Golub-Kahan-Lanczos, src/svd/impls/lanczos/gklanczos.c) and
 * "cyclic" (the eigenproblem of [0 A; A^T 0] through a shell operator on device vectors,
 * src/svd/impls/cyclic/cyclic.c).  The operator may be rectangular; its transpose is
 * built explicitly with MatTranspose. */
typedef enum { SVD_LARGEST = 0, SVD_SMALLEST = 1 } SVDWhich;
typedef enum { SVD_ERROR_ABSOLUTE = 0, SVD_ERROR_RELATIVE = 1 } SVDErrorType;
typedef enum { SVD_CONV_ABS = 0, SVD_CONV_REL = 1 } SVDConv;
typedef enum { SVD_CONVERGED_TOL = 1, SVD_CONVERGED_USER = 2, SVD_DIVERGED_ITS = -1, SVD_DIVERGED_BREAKDOWN = -2,
               SVD_CONVERGED_ITERATING = 0 } SVDConvergedReason;

PetscErrorCode SVDCreate(void *ctx, SVD *svd);                                   /* svdbasic.c */
PetscErrorCode SVDSetOperator(SVD svd, Mat A);                                   /* svdsetup.c:47-60 */
PetscErrorCode SVDSetType(SVD svd, const char *type);
PetscErrorCode SVDGetType(SVD svd, const char **type);
PetscErrorCode SVDSetDimensions(SVD svd, PetscInt nsv, PetscInt ncv, PetscInt mpd);
PetscErrorCode SVDGetDimensions(SVD svd, PetscInt *nsv, PetscInt *ncv, PetscInt *mpd);
PetscErrorCode SVDSetTolerances(SVD svd, PetscReal tol, PetscInt maxits);
PetscErrorCode SVDGetTolerances(SVD svd, PetscReal *tol, PetscInt *maxits);
PetscErrorCode SVDSetWhichSingularTriplets(SVD svd, SVDWhich which);
PetscErrorCode SVDSetConvergenceTest(SVD svd, SVDConv conv);
PetscErrorCode SVDSetInitialSpace(SVD svd, PetscInt n, Vec *is);                 /* svdsetup.c:222-232 */
PetscErrorCode SVDTRLanczosSetOneSide(SVD svd, PetscBool oneside);               /* trlanczos.c:400-426 */
PetscErrorCode SVDLanczosSetOneSide(SVD svd, PetscBool oneside);                 /* gklanczos.c:292-330 */
PetscErrorCode SVDCrossGetEPS(SVD svd, EPS *eps);
PetscErrorCode SVDCyclicGetEPS(SVD svd, EPS *eps);
PetscErrorCode SVDGetBV(SVD svd, BV *V, BV *U);
PetscErrorCode SVDSetUp(SVD svd);                                                /* svdsetup.c:98-205 */
PetscErrorCode SVDSolve(SVD svd);                                                /* svdsolve.c:80-128 */
PetscErrorCode SVDGetIterationNumber(SVD svd, PetscInt *its);
PetscErrorCode SVDGetConverged(SVD svd, PetscInt *nconv);
PetscErrorCode SVDGetConvergedReason(SVD svd, SVDConvergedReason *reason);
/* u has the local rows of the operator as given, v its local columns; either may be NULL */
PetscErrorCode SVDGetSingularTriplet(SVD svd, PetscInt i, PetscReal *sigma, Vec u, Vec v);
PetscErrorCode SVDComputeError(SVD svd, PetscInt i, SVDErrorType type, PetscReal *error);
PetscErrorCode SVDDestroy(SVD *svd);

#ifdef __cplusplus
}
#endif
#endif
