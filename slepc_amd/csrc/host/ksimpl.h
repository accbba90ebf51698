/* Private helpers of libkseps: error propagation in the style of PETSc's
 * SETERRQ/CHKERRQ, allocation, and access to the DS / EPS structures. */
#ifndef KSIMPL_H
#define KSIMPL_H

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "kseps.h"

PetscErrorCode KSError(const char *func, int line, PetscErrorCode code, const char *fmt, ...);
void KSErrorTrace(const char *func, int line);

#define SETERRQ(code, ...) return KSError(__func__, __LINE__, (code), __VA_ARGS__)
#define CHKERRQ(ierr) do { if (ierr) { KSErrorTrace(__func__, __LINE__); return (ierr); } } while (0)
/* a libksgpu call: its message becomes ours */
#define CHKKSG(call) do { int rc_ = (call); if (rc_) return KSError(__func__, __LINE__, PETSC_ERR_LIB, "libksgpu: %s", ksg_last_error()); } while (0)

void *MatGetDeviceHandle_Private(Mat A);          /* mat.c */

/* split-phase reductions (vec.c) */
PetscErrorCode KSRedBegin_Private(const void *owner, const struct _VecOps *ops, void *ctx, PetscInt count, int root, PetscScalar **vals);
void KSRedPosted_Private(PetscInt dev_off);
PetscErrorCode KSRedEnd_Private(const void *owner, PetscInt count, PetscScalar *out);
void KSRedAbandon_Private(void);
PetscErrorCode VecForeignAccessBegin_Private(Vec x);
PetscErrorCode VecForeignAccessEnd_Private(Vec y);

#define PETSC_MACHINE_EPSILON 2.2204460492503131e-16
#define PetscMax(a, b) ((a) < (b) ? (b) : (a))
#define PetscMin(a, b) ((a) < (b) ? (a) : (b))

static inline PetscReal SlepcAbsEigenvalue(PetscScalar re, PetscScalar im) { return hypot(re, im); }

struct _p_DS {
  int type;                 /* 0 hep, 1 nhep, 2 svd */
  PetscBool compact, extrarow;
  PetscInt ld, n, m, l, k, t;
  DSStateType state;
  PetscScalar *mat[DS_NUM_MAT];
  PetscReal *rmat;          /* DS_MAT_T: 3*ld reals */
  PetscInt *perm;
  SlepcCompareFn cmp;
  void *cmpctx;
};

struct _p_EPS {
  void *ctx;
  Mat A;
  BV V;
  DS ds;
  EPSProblemType problem_type;
  int type;                 /* 0 krylovschur (default), 1 arnoldi, 2 lanczos (full reorthogonalisation) */
  PetscBool ishermitian;
  Mat B;                    /* generalized problem A x = lambda B x (NULL: standard) */
  PetscBool isgeneralized, ispositive, purify;
  ST st;
  PetscScalar target;
  SlepcCompareFn cmp0; void *cmp0ctx;   /* the ordering asked for, on values of the original problem */
  void *cmpctx;             /* context of cmp (the ordering the solver uses, on transformed values) */
  Vec *defl; PetscInt nds;  /* deflation space: pending vectors, or the number of constraints held */
  PetscInt nev, ncv, mpd, max_it;
  PetscInt64 n, nloc;
  PetscReal tol;
  EPSWhich which;
  SlepcCompareFn cmp;
  SlepcCompareFn usercmp;   /* EPSSetEigenvalueComparison */
  PetscReal keep;
  PetscBool lock, keep_set;
  EPSConv conv;             /* convergence test: relative (default) or absolute */
  EPSLanczosReorthogType reorthog;   /* Lanczos */
  BV AV;                    /* Lanczos, selective: nearly converged Ritz vectors */
  PetscBool delayed;        /* Arnoldi: delayed reorthogonalisation (arnoldi.c:168-175) */
  Vec *IS; PetscInt nini;
  PetscScalar *eigr, *eigi;
  PetscReal *errest;
  PetscInt *perm;
  PetscInt nconv, its;
  EPSConvergedReason reason;
  int state;                /* 0 initial, 1 setup, 2 solved, 3 eigenvectors */
  EPSMonitorFn monitor; void *monitorctx;
  Vec work[5]; PetscInt nwork;
  PetscInt64 matvecs, gs_passes;
};

PetscBool STIsTrivial_Private(ST st);              /* st.c */
PetscBool STIsShift_Private(ST st);

/* counters shared between bv.c and eps.c */
extern PetscInt64 ks_gs_passes;

#endif
