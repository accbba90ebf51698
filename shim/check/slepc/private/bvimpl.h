/* DECLARATIONS-ONLY STAND-IN for the check of the shim files (`make -C shim check`), written for
 * that check from the documented SLEPc 3.7 BV interface: the members of the op table and of the
 * BV object that bv_hip_petsc.c names, nothing else.  See ../../petsc_decls.h. */
#ifndef KSG_SHIM_CHECK_BVIMPL_H
#define KSG_SHIM_CHECK_BVIMPL_H
#include "../../petsc_decls.h"

typedef struct _p_BV *BV;
struct _BVOps {
  PetscErrorCode (*mult)(BV,PetscScalar,PetscScalar,BV,Mat);
  PetscErrorCode (*multvec)(BV,PetscScalar,PetscScalar,Vec,PetscScalar*);
  PetscErrorCode (*multinplace)(BV,Mat,PetscInt,PetscInt);
  PetscErrorCode (*multinplacetrans)(BV,Mat,PetscInt,PetscInt);
  PetscErrorCode (*dot)(BV,BV,Mat);
  PetscErrorCode (*dotvec)(BV,Vec,PetscScalar*);
  PetscErrorCode (*dotvec_local)(BV,Vec,PetscScalar*);
  PetscErrorCode (*dotvec_begin)(BV,Vec,PetscScalar*);
  PetscErrorCode (*dotvec_end)(BV,Vec,PetscScalar*);
  PetscErrorCode (*scale)(BV,PetscInt,PetscScalar);
  PetscErrorCode (*norm)(BV,PetscInt,NormType,PetscReal*);
  PetscErrorCode (*norm_local)(BV,PetscInt,NormType,PetscReal*);
  PetscErrorCode (*matmult)(BV,Mat,BV);
  PetscErrorCode (*copy)(BV,BV);
  PetscErrorCode (*resize)(BV,PetscInt,PetscBool);
  PetscErrorCode (*getcolumn)(BV,PetscInt,Vec*);
  PetscErrorCode (*restorecolumn)(BV,PetscInt,Vec*);
  PetscErrorCode (*getarray)(BV,PetscScalar**);
  PetscErrorCode (*restorearray)(BV,PetscScalar**);
  PetscErrorCode (*getarrayread)(BV,const PetscScalar**);
  PetscErrorCode (*restorearrayread)(BV,const PetscScalar**);
  PetscErrorCode (*destroy)(BV);
};
struct _p_BV {
  struct _p_PetscObject hdr;
  struct _BVOps ops[1];
  Vec       t;
  PetscInt  n,N,m,l,k,nc;
  Vec       buffer,Bx;
  Vec       cv[2];
  PetscInt  ci[2];
  Mat       matrix;
  void      *data;
};
#define BVAvailableVec (((bv->ci[0]==-bv->nc-1)? 0: (bv->ci[1]==-bv->nc-1)? 1: -1))
PetscErrorCode BV_IPMatMult(BV,Vec);
PetscErrorCode BVRegister(const char[],PetscErrorCode (*)(BV));
PetscErrorCode BVCreateVec(BV,Vec*);
PetscErrorCode BVGetColumn(BV,PetscInt,Vec*);
PetscErrorCode BVRestoreColumn(BV,PetscInt,Vec*);
PetscErrorCode BVGetArray(BV,PetscScalar**);
PetscErrorCode BVRestoreArray(BV,PetscScalar**);
PetscErrorCode SlepcInitialize(int*,char***,const char[],const char[]);
#endif
