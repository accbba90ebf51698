"""ctypes binding of libkseps (the C ABI declared in ``include/kseps.h``): the
host side of the Krylov-Schur path -- EPS, BV, DS, Vec, Mat with the names and
argument meaning of the reference's C interface.

Plumbing only: the numerical work happens in libksgpu on the device, the
control flow in the C host library.  The ctypes mirrors of the C structures
(`VecStruct`, `BVStruct`, ...) exist so that a test can register a BV type of
its own through ``BVRegister`` exactly as a C plug-in would.
"""
import ctypes as C
import os

import numpy as np

from . import ksgpu

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libkseps.so")

c_i64 = C.c_int64
c_dp = C.POINTER(C.c_double)
Err = C.c_int

NORM_1, NORM_2, NORM_FROBENIUS, NORM_INFINITY = 0, 1, 2, 3
EPS_HEP, EPS_GHEP, EPS_NHEP, EPS_GNHEP, EPS_PGNHEP = 1, 2, 3, 4, 5
EPS_TARGET_MAGNITUDE, EPS_TARGET_REAL = 7, 8
EPS_LARGEST_MAGNITUDE, EPS_SMALLEST_MAGNITUDE, EPS_LARGEST_REAL, EPS_SMALLEST_REAL, \
    EPS_LARGEST_IMAGINARY, EPS_SMALLEST_IMAGINARY = 1, 2, 3, 4, 5, 6
WHICH = {"largest_magnitude": 1, "smallest_magnitude": 2, "largest_real": 3, "smallest_real": 4,
         "largest_imaginary": 5, "smallest_imaginary": 6, "target_magnitude": 7, "target_real": 8}
EPS_ERROR_ABSOLUTE, EPS_ERROR_RELATIVE = 0, 1
EPS_CONVERGED_TOL, EPS_DIVERGED_ITS, EPS_DIVERGED_BREAKDOWN = 1, -1, -2
BV_ORTHOG_CGS, BV_ORTHOG_MGS = 0, 1
BV_ORTHOG_REFINE_IFNEEDED, BV_ORTHOG_REFINE_NEVER, BV_ORTHOG_REFINE_ALWAYS = 0, 1, 2
BV_ORTHOG_BLOCK_GS, BV_ORTHOG_BLOCK_CHOL, BV_ORTHOG_BLOCK_TSQR = 0, 1, 2
LANCZOS_REORTHOG = {"local": 0, "full": 1, "selective": 2, "periodic": 3, "partial": 4, "delayed": 5}
DS_STATE_RAW, DS_STATE_INTERMEDIATE, DS_STATE_CONDENSED, DS_STATE_TRUNCATED = 0, 1, 2, 3
DS_MAT_A, DS_MAT_Q, DS_MAT_X, DS_MAT_T = 0, 1, 2, 3


class KsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libkseps error %d: %s" % (code, msg))
        self.code = code


# ---- C structure mirrors -------------------------------------------------------
class VecStruct(C.Structure):
    pass


VecP = C.POINTER(VecStruct)


class VecOps(C.Structure):
    _fields_ = [("alloc", C.CFUNCTYPE(Err, VecP)),
                ("free", C.CFUNCTYPE(Err, VecP)),
                ("copy", C.CFUNCTYPE(Err, VecP, VecP)),
                ("set", C.CFUNCTYPE(Err, VecP, C.c_double)),
                ("scale", C.CFUNCTYPE(Err, VecP, C.c_double)),
                ("axpy", C.CFUNCTYPE(Err, VecP, C.c_double, VecP)),
                ("dot", C.CFUNCTYPE(Err, VecP, VecP, c_dp)),
                ("norm2", C.CFUNCTYPE(Err, VecP, c_dp)),
                ("fillhash", C.CFUNCTYPE(Err, VecP, C.c_uint64)),
                ("tohost", C.CFUNCTYPE(Err, VecP, c_dp)),
                ("fromhost", C.CFUNCTYPE(Err, VecP, c_dp)),
                ("redopen", C.CFUNCTYPE(Err, C.c_void_p)),
                ("dotpost", C.CFUNCTYPE(Err, VecP, VecP, C.POINTER(C.c_int))),
                ("redclose", C.CFUNCTYPE(Err, C.c_void_p, c_dp, C.c_int))]


VecStruct._fields_ = [("ops", C.POINTER(VecOps)), ("n", c_i64), ("N", c_i64), ("row0", c_i64),
                      ("array", C.c_void_p), ("own", C.c_void_p), ("state", c_i64),
                      ("ctx", C.c_void_p), ("id", c_i64)]


class MatStruct(C.Structure):
    pass


MatP = C.POINTER(MatStruct)
MATMULT_FN = C.CFUNCTYPE(Err, MatP, VecP, VecP)
MATVECS_FN = C.CFUNCTYPE(Err, MatP, C.POINTER(VecP), C.POINTER(VecP))
MATDESTROY_FN = C.CFUNCTYPE(Err, MatP)
MATTRANSPOSE_FN = C.CFUNCTYPE(Err, MatP, C.POINTER(MatP))
MATRESNORM_FN = C.CFUNCTYPE(Err, MatP, C.c_double, C.c_double, VecP, VecP, VecP, c_dp)


class MatOps(C.Structure):
    _fields_ = [("mult", MATMULT_FN), ("createvecs", MATVECS_FN), ("destroy", MATDESTROY_FN),
                ("residualnorm", MATRESNORM_FN), ("transpose", MATTRANSPOSE_FN)]


MatStruct._fields_ = [("ops", MatOps), ("n", c_i64), ("N", c_i64), ("row0", c_i64),
                      ("data", C.c_void_p), ("ctx", C.c_void_p), ("cn", c_i64), ("cN", c_i64), ("col0", c_i64)]


class BVStruct(C.Structure):
    pass


BVP = C.POINTER(BVStruct)


class BVOps(C.Structure):
    _fields_ = [("mult", C.CFUNCTYPE(Err, BVP, C.c_double, C.c_double, BVP, c_dp, C.c_int)),
                ("multvec", C.CFUNCTYPE(Err, BVP, C.c_double, C.c_double, VecP, c_dp)),
                ("multinplace", C.CFUNCTYPE(Err, BVP, c_dp, C.c_int, C.c_int, C.c_int)),
                ("multinplacetrans", C.CFUNCTYPE(Err, BVP, c_dp, C.c_int, C.c_int, C.c_int)),
                ("dot", C.CFUNCTYPE(Err, BVP, BVP, c_dp, C.c_int)),
                ("dotvec", C.CFUNCTYPE(Err, BVP, VecP, c_dp)),
                ("dotvec_local", C.CFUNCTYPE(Err, BVP, VecP, c_dp)),
                ("scale", C.CFUNCTYPE(Err, BVP, C.c_int, C.c_double)),
                ("norm", C.CFUNCTYPE(Err, BVP, C.c_int, C.c_int, c_dp)),
                ("norm_local", C.CFUNCTYPE(Err, BVP, C.c_int, C.c_int, c_dp)),
                ("matmult", C.CFUNCTYPE(Err, BVP, MatP, BVP)),
                ("copy", C.CFUNCTYPE(Err, BVP, BVP)),
                ("resize", C.CFUNCTYPE(Err, BVP, C.c_int, C.c_int)),
                ("getcolumn", C.CFUNCTYPE(Err, BVP, C.c_int, C.POINTER(VecP))),
                ("restorecolumn", C.CFUNCTYPE(Err, BVP, C.c_int, C.POINTER(VecP))),
                ("destroy", C.CFUNCTYPE(Err, BVP)),
                ("dotvec_post", C.CFUNCTYPE(Err, BVP, VecP, C.POINTER(C.c_int))),
                ("orthogonalize_tsqr", C.CFUNCTYPE(Err, BVP, c_dp, C.c_int))]


BVStruct._fields_ = [("ops", BVOps), ("type_name", C.c_char * 32), ("t", VecP), ("n", c_i64),
                     ("N", c_i64), ("m", C.c_int), ("l", C.c_int), ("k", C.c_int), ("nc", C.c_int),
                     ("orthog_type", C.c_int), ("orthog_ref", C.c_int), ("orthog_eta", C.c_double),
                     ("buffer", c_dp), ("h", c_dp), ("c", c_dp), ("cv", VecP * 2),
                     ("ci", C.c_int * 2), ("st", c_i64 * 2), ("state", c_i64),
                     ("rand_seed", C.c_uint64), ("data", C.c_void_p), ("ctx", C.c_void_p),
                     ("orthog_block", C.c_int), ("matrix", MatP), ("indef", C.c_int), ("Bx", VecP),
                     ("xid", c_i64), ("xstate", c_i64), ("cached", BVP), ("bvstate", c_i64),
                     ("omega", c_dp)]

BVCREATE_FN = C.CFUNCTYPE(Err, BVP)
COMPARE_FN = C.CFUNCTYPE(Err, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int),
                         C.c_void_p)
MONITOR_FN = C.CFUNCTYPE(Err, C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_int, C.c_void_p)

vp = C.c_void_p
SIGNATURES = {
    "KSGetLastError": (C.c_char_p, []),
    "VecCreateHIP": (Err, [vp, c_i64, c_i64, c_i64, C.POINTER(VecP)]),
    "VecCreateWithOps": (Err, [C.POINTER(VecOps), vp, c_i64, c_i64, c_i64, C.c_int, C.POINTER(VecP)]),
    "VecDuplicate": (Err, [VecP, C.POINTER(VecP)]),
    "VecDestroy": (Err, [C.POINTER(VecP)]),
    "VecPlaceArray": (Err, [VecP, vp]),
    "VecResetArray": (Err, [VecP]),
    "VecCopy": (Err, [VecP, VecP]),
    "VecSet": (Err, [VecP, C.c_double]),
    "VecScale": (Err, [VecP, C.c_double]),
    "VecAXPY": (Err, [VecP, C.c_double, VecP]),
    "VecDot": (Err, [VecP, VecP, c_dp]),
    "VecNorm": (Err, [VecP, C.c_int, c_dp]),
    "VecGetSize": (Err, [VecP, C.POINTER(c_i64)]),
    "VecGetLocalSize": (Err, [VecP, C.POINTER(c_i64)]),
    "VecGetValuesHost": (Err, [VecP, c_dp]),
    "VecSetValuesHost": (Err, [VecP, c_dp]),
    "VecFillHash": (Err, [VecP, C.c_uint64]),
    "MatCreateAIJHIP": (Err, [vp, c_i64, c_i64, c_i64, vp, vp, vp, C.POINTER(MatP)]),
    "MatCreateAIJHIPRect": (Err, [vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, vp, vp, vp, C.POINTER(MatP)]),
    "MatTranspose": (Err, [MatP, C.POINTER(MatP)]),
    "MatCreateShell": (Err, [vp, c_i64, c_i64, c_i64, vp, MATMULT_FN, MATVECS_FN, C.POINTER(MatP)]),
    "MatShellGetContext": (Err, [MatP, C.POINTER(vp)]),
    "MatShellSetColumnLayout": (Err, [MatP, c_i64, c_i64, c_i64]),
    "MatShellSetTranspose": (Err, [MatP, MATTRANSPOSE_FN]),
    "KSMatReadPetscBinary": (Err, [C.c_char_p, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "KSMatWritePetscBinary": (Err, [C.c_char_p, c_i64, c_i64, vp, vp, vp]),
    "KSFree": (None, [vp]),
    "MatLoadAIJHIP": (Err, [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(MatP)]),
    "MatMult": (Err, [MatP, VecP, VecP]),
    "MatCreateVecs": (Err, [MatP, C.POINTER(VecP), C.POINTER(VecP)]),
    "MatGetSize": (Err, [MatP, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "MatGetLocalSize": (Err, [MatP, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "MatDestroy": (Err, [C.POINTER(MatP)]),
    "BVCreate": (Err, [vp, C.POINTER(BVP)]),
    "BVRegister": (Err, [C.c_char_p, BVCREATE_FN]),
    "BVSetType": (Err, [BVP, C.c_char_p]),
    "BVSetSizesFromVec": (Err, [BVP, VecP, C.c_int]),
    "BVGetSizes": (Err, [BVP, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(C.c_int)]),
    "BVResize": (Err, [BVP, C.c_int, C.c_int]),
    "BVSetActiveColumns": (Err, [BVP, C.c_int, C.c_int]),
    "BVInsertConstraints": (Err, [BVP, C.POINTER(C.c_int), C.POINTER(VecP)]),
    "BVSetNumConstraints": (Err, [BVP, C.c_int]),
    "BVGetNumConstraints": (Err, [BVP, C.POINTER(C.c_int)]),
    "BVSetMatrix": (Err, [BVP, MatP, C.c_int]),
    "BVGetMatrix": (Err, [BVP, C.POINTER(MatP), C.POINTER(C.c_int)]),
    "BVApplyMatrix": (Err, [BVP, VecP, VecP]),
    "BVApplyMatrixBV": (Err, [BVP, BVP]),
    "BVGetCachedBV": (Err, [BVP, C.POINTER(BVP)]),
    "BVSetSignature": (Err, [BVP, c_dp]),
    "BVGetSignature": (Err, [BVP, c_dp]),
    "BVDuplicate": (Err, [BVP, C.POINTER(BVP)]),
    "BVDuplicateResize": (Err, [BVP, C.c_int, C.POINTER(BVP)]),
    "BVGetActiveColumns": (Err, [BVP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "BVSetOrthogonalization": (Err, [BVP, C.c_int, C.c_int, C.c_double, C.c_int]),
    "BVOrthogonalize": (Err, [BVP, c_dp, C.c_int]),
    "BVOrthogonalizeSomeColumn": (Err, [BVP, C.c_int, C.POINTER(C.c_int), c_dp, c_dp, C.POINTER(C.c_int)]),
    "BVGetColumn": (Err, [BVP, C.c_int, C.POINTER(VecP)]),
    "BVRestoreColumn": (Err, [BVP, C.c_int, C.POINTER(VecP)]),
    "BVGetBuffer": (Err, [BVP, C.POINTER(c_dp), C.POINTER(C.c_int)]),
    "BVInsertVec": (Err, [BVP, C.c_int, VecP]),
    "BVInsertVecs": (Err, [BVP, C.c_int, C.POINTER(C.c_int), C.POINTER(VecP), C.c_int]),
    "BVSetRandomColumn": (Err, [BVP, C.c_int]),
    "BVCopy": (Err, [BVP, BVP]),
    "BVCopyVec": (Err, [BVP, C.c_int, VecP]),
    "BVCopyColumn": (Err, [BVP, C.c_int, C.c_int]),
    "BVMult": (Err, [BVP, C.c_double, C.c_double, BVP, c_dp, C.c_int]),
    "BVMultVec": (Err, [BVP, C.c_double, C.c_double, VecP, c_dp]),
    "BVMultColumn": (Err, [BVP, C.c_double, C.c_double, C.c_int, c_dp]),
    "BVMultInPlace": (Err, [BVP, c_dp, C.c_int, C.c_int, C.c_int]),
    "BVMultInPlaceTranspose": (Err, [BVP, c_dp, C.c_int, C.c_int, C.c_int]),
    "BVDot": (Err, [BVP, BVP, c_dp, C.c_int]),
    "BVDotVec": (Err, [BVP, VecP, c_dp]),
    "BVDotColumn": (Err, [BVP, C.c_int, c_dp]),
    "BVDotVecBegin": (Err, [BVP, VecP, c_dp]), "BVDotVecEnd": (Err, [BVP, VecP, c_dp]),
    "BVDotColumnBegin": (Err, [BVP, C.c_int, c_dp]), "BVDotColumnEnd": (Err, [BVP, C.c_int, c_dp]),
    "BVNormVecBegin": (Err, [BVP, VecP, C.c_int, c_dp]), "BVNormVecEnd": (Err, [BVP, VecP, C.c_int, c_dp]),
    "BVNormColumnBegin": (Err, [BVP, C.c_int, C.c_int, c_dp]), "BVNormColumnEnd": (Err, [BVP, C.c_int, C.c_int, c_dp]),
    "VecDotBegin": (Err, [VecP, VecP, c_dp]), "VecDotEnd": (Err, [VecP, VecP, c_dp]),
    "VecNormBegin": (Err, [VecP, C.c_int, c_dp]), "VecNormEnd": (Err, [VecP, C.c_int, c_dp]),
    "BVScale": (Err, [BVP, C.c_double]),
    "BVScaleColumn": (Err, [BVP, C.c_int, C.c_double]),
    "BVNorm": (Err, [BVP, C.c_int, c_dp]),
    "BVNormColumn": (Err, [BVP, C.c_int, C.c_int, c_dp]),
    "BVNormVec": (Err, [BVP, VecP, C.c_int, c_dp]),
    "BVMatMult": (Err, [BVP, MatP, BVP]),
    "BVOrthogonalizeVec": (Err, [BVP, VecP, c_dp, c_dp, C.POINTER(C.c_int)]),
    "BVOrthogonalizeColumn": (Err, [BVP, C.c_int, c_dp, c_dp, C.POINTER(C.c_int)]),
    "BVOrthonormalizeColumn": (Err, [BVP, C.c_int, C.c_int, c_dp, C.POINTER(C.c_int)]),
    "BVDestroy": (Err, [C.POINTER(BVP)]),
    "BVCreate_HIP": (Err, [BVP]),
    "DSCreate": (Err, [C.POINTER(vp)]),
    "DSSetType": (Err, [vp, C.c_char_p]),
    "DSSetCompact": (Err, [vp, C.c_int]),
    "DSSetExtraRow": (Err, [vp, C.c_int]),
    "DSAllocate": (Err, [vp, C.c_int]),
    "DSSetSlepcSC": (Err, [vp, COMPARE_FN, vp]),
    "DSSetDimensions": (Err, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "DSGetDimensions": (Err, [vp] + [C.POINTER(C.c_int)] * 5),
    "DSGetLeadingDimension": (Err, [vp, C.POINTER(C.c_int)]),
    "DSSetState": (Err, [vp, C.c_int]),
    "DSGetState": (Err, [vp, C.POINTER(C.c_int)]),
    "DSGetArray": (Err, [vp, C.c_int, C.POINTER(c_dp)]),
    "DSGetArrayReal": (Err, [vp, C.c_int, C.POINTER(c_dp)]),
    "DSSolve": (Err, [vp, c_dp, c_dp]),
    "DSSort": (Err, [vp, c_dp, c_dp, c_dp, c_dp, C.POINTER(C.c_int)]),
    "DSUpdateExtraRow": (Err, [vp]),
    "DSVectors": (Err, [vp, C.c_int, C.POINTER(C.c_int), c_dp]),
    "DSTruncate": (Err, [vp, C.c_int]),
    "DSGetMat": (Err, [vp, C.c_int, c_dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "DSDestroy": (Err, [C.POINTER(vp)]),
    "SlepcCompareLargestMagnitude": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcCompareSmallestMagnitude": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcCompareLargestReal": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcCompareSmallestReal": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcCompareLargestImaginary": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcCompareSmallestImaginary": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcCompareSmallestPosReal": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcSortEigenvalues": (Err, [COMPARE_FN, vp, C.c_int, c_dp, c_dp, C.POINTER(C.c_int)]),
    "EPSCreate": (Err, [vp, C.POINTER(vp)]),
    "EPSSetOperators": (Err, [vp, MatP, MatP]),
    "EPSSetProblemType": (Err, [vp, C.c_int]),
    "EPSSetTarget": (Err, [vp, C.c_double]), "EPSGetTarget": (Err, [vp, c_dp]),
    "EPSGetST": (Err, [vp, C.POINTER(vp)]), "EPSSetPurify": (Err, [vp, C.c_int]),
    "EPSSetDeflationSpace": (Err, [vp, C.c_int, C.POINTER(VecP)]),
    "STCreate": (Err, [vp, C.POINTER(vp)]), "STDestroy": (Err, [C.POINTER(vp)]),
    "STSetType": (Err, [vp, C.c_char_p]), "STGetType": (Err, [vp, C.POINTER(C.c_char_p)]),
    "STSetOperators": (Err, [vp, MatP, MatP]), "STGetNumMatrices": (Err, [vp, C.POINTER(C.c_int)]),
    "STSetShift": (Err, [vp, C.c_double]), "STGetShift": (Err, [vp, c_dp]), "STSetDefaultShift": (Err, [vp, C.c_double]),
    "STSetKSPType": (Err, [vp, C.c_char_p]), "STGetKSPType": (Err, [vp, C.POINTER(C.c_char_p)]),
    "STSetKSPTolerances": (Err, [vp, C.c_double, C.c_int, C.c_int]),
    "STGetKSPIterations": (Err, [vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "STGetBilinearForm": (Err, [vp, C.POINTER(MatP)]), "STSetUp": (Err, [vp]),
    "STApply": (Err, [vp, VecP, VecP]), "STMatSolve": (Err, [vp, VecP, VecP]),
    "STBackTransform": (Err, [vp, C.c_int, c_dp, c_dp]),
    "SlepcCompareTargetMagnitude": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "SlepcCompareTargetReal": (Err, [C.c_double] * 4 + [C.POINTER(C.c_int), vp]),
    "EPSSetType": (Err, [vp, C.c_char_p]),
    "EPSGetType": (Err, [vp, C.POINTER(C.c_char_p)]),
    "EPSSetDimensions": (Err, [vp, C.c_int, C.c_int, C.c_int]),
    "EPSSetTolerances": (Err, [vp, C.c_double, C.c_int]),
    "EPSSetWhichEigenpairs": (Err, [vp, C.c_int]),
    "EPSSetConvergenceTest": (Err, [vp, C.c_int]),
    "EPSSetInitialSpace": (Err, [vp, C.c_int, C.POINTER(VecP)]),
    "EPSKrylovSchurSetRestart": (Err, [vp, C.c_double]),
    "EPSKrylovSchurSetLocking": (Err, [vp, C.c_int]),
    "EPSArnoldiSetDelayed": (Err, [vp, C.c_int]),
    "EPSLanczosSetReorthog": (Err, [vp, C.c_int]),
    "EPSMonitorSet": (Err, [vp, MONITOR_FN, vp]),
    "EPSGetBV": (Err, [vp, C.POINTER(BVP)]),
    "EPSGetDS": (Err, [vp, C.POINTER(vp)]),
    "EPSSetUp": (Err, [vp]),
    "EPSSolve": (Err, [vp]),
    "EPSGetIterationNumber": (Err, [vp, C.POINTER(C.c_int)]),
    "EPSGetConverged": (Err, [vp, C.POINTER(C.c_int)]),
    "EPSGetConvergedReason": (Err, [vp, C.POINTER(C.c_int)]),
    "EPSGetDimensions": (Err, [vp] + [C.POINTER(C.c_int)] * 3),
    "EPSGetTolerances": (Err, [vp, c_dp, C.POINTER(C.c_int)]),
    "EPSGetEigenvalue": (Err, [vp, C.c_int, c_dp, c_dp]),
    "EPSGetEigenvector": (Err, [vp, C.c_int, VecP, VecP]),
    "EPSGetEigenpair": (Err, [vp, C.c_int, c_dp, c_dp, VecP, VecP]),
    "EPSGetErrorEstimate": (Err, [vp, C.c_int, c_dp]),
    "EPSComputeError": (Err, [vp, C.c_int, C.c_int, c_dp]),
    "EPSGetCounters": (Err, [vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "EPSDestroy": (Err, [C.POINTER(vp)]),
    "SVDCreate": (Err, [vp, C.POINTER(vp)]),
    "SVDSetOperator": (Err, [vp, MatP]),
    "SVDSetType": (Err, [vp, C.c_char_p]),
    "SVDGetType": (Err, [vp, C.POINTER(C.c_char_p)]),
    "SVDSetDimensions": (Err, [vp, C.c_int, C.c_int, C.c_int]),
    "SVDGetDimensions": (Err, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "SVDSetTolerances": (Err, [vp, C.c_double, C.c_int]),
    "SVDGetTolerances": (Err, [vp, c_dp, C.POINTER(C.c_int)]),
    "SVDSetWhichSingularTriplets": (Err, [vp, C.c_int]),
    "SVDSetConvergenceTest": (Err, [vp, C.c_int]),
    "SVDSetInitialSpace": (Err, [vp, C.c_int, C.POINTER(VecP)]),
    "SVDTRLanczosSetOneSide": (Err, [vp, C.c_int]),
    "SVDLanczosSetOneSide": (Err, [vp, C.c_int]),
    "SVDCyclicGetEPS": (Err, [vp, C.POINTER(vp)]),
    "EPSSetEigenvalueComparison": (Err, [vp, COMPARE_FN, vp]),
    "SVDCrossGetEPS": (Err, [vp, C.POINTER(vp)]),
    "SVDGetBV": (Err, [vp, C.POINTER(BVP), C.POINTER(BVP)]),
    "SVDSetUp": (Err, [vp]),
    "SVDSolve": (Err, [vp]),
    "SVDGetIterationNumber": (Err, [vp, C.POINTER(C.c_int)]),
    "SVDGetConverged": (Err, [vp, C.POINTER(C.c_int)]),
    "SVDGetConvergedReason": (Err, [vp, C.POINTER(C.c_int)]),
    "SVDGetSingularTriplet": (Err, [vp, C.c_int, c_dp, VecP, VecP]),
    "SVDComputeError": (Err, [vp, C.c_int, C.c_int, c_dp]),
    "SVDDestroy": (Err, [C.POINTER(vp)]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        ksgpu.load()      # libkseps links against libksgpu
        if not os.path.exists(LIB_PATH):
            raise ImportError("libkseps.so not built: run `python __graft_entry__.py build` "
                              "(expected %s)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise KsError(rc, load().KSGetLastError().decode(errors="replace"))


def _dptr(a):
    return a.ctypes.data_as(c_dp)


# ---- thin object wrappers ----------------------------------------------------------
class Vec:
    def __init__(self, p, lib=None):
        self.p = p
        self.lib = lib or load()

    @classmethod
    def create_hip(cls, ctx, n, N=None, row0=0):
        lib = load()
        p = VecP()
        check(lib.VecCreateHIP(ctx.h, n, N if N is not None else n, row0, C.byref(p)))
        return cls(p, lib)

    @property
    def n(self):
        return self.p.contents.n

    def duplicate(self):
        q = VecP()
        check(self.lib.VecDuplicate(self.p, C.byref(q)))
        return Vec(q, self.lib)

    def set_values(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.size == self.n
        check(self.lib.VecSetValuesHost(self.p, _dptr(a)))

    def get_values(self):
        out = np.empty(self.n)
        check(self.lib.VecGetValuesHost(self.p, _dptr(out)))
        return out

    def norm(self):
        r = C.c_double()
        check(self.lib.VecNorm(self.p, NORM_2, C.byref(r)))
        return r.value

    def destroy(self):
        if self.p:
            check(self.lib.VecDestroy(C.byref(self.p)))
            self.p = None


class Mat:
    def __init__(self, p, lib=None, keep=None):
        self.p = p
        self.lib = lib or load()
        self._keep = keep

    @classmethod
    def create_aij_hip(cls, ctx, rowptr, colidx, val, N=None, row0=0):
        lib = load()
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        ci = np.ascontiguousarray(colidx, dtype=np.int32)
        va = np.ascontiguousarray(val, dtype=np.float64)
        n = rp.size - 1
        p = MatP()
        check(lib.MatCreateAIJHIP(ctx.h, n, N if N is not None else n, row0, rp.ctypes.data,
                                  ci.ctypes.data, va.ctypes.data, C.byref(p)))
        return cls(p, lib)

    @classmethod
    def create_aij_hip_rect(cls, ctx, rowptr, colidx, val, ncols, M=None, row0=0, ncols_local=None, col0=0):
        """M x ncols operator from the CSR rows [row0, row0+m); one process: all of it."""
        lib = load()
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        ci = np.ascontiguousarray(colidx, dtype=np.int32)
        va = np.ascontiguousarray(val, dtype=np.float64)
        m = rp.size - 1
        p = MatP()
        check(lib.MatCreateAIJHIPRect(ctx.h, m, M if M is not None else m, row0,
                                      ncols_local if ncols_local is not None else ncols, ncols, col0,
                                      rp.ctypes.data, ci.ctypes.data, va.ctypes.data, C.byref(p)))
        return cls(p, lib)

    def transpose(self):
        p = MatP()
        check(self.lib.MatTranspose(self.p, C.byref(p)))
        return Mat(p, self.lib)

    @classmethod
    def load_aij_hip(cls, ctx, filename, rank=0, nranks=1):
        """MatLoad of a PETSc binary file onto the device (row block of `rank`)."""
        lib = load()
        p = MatP()
        check(lib.MatLoadAIJHIP(ctx.h, os.fsencode(filename), rank, nranks, C.byref(p)))
        return cls(p, lib)

    def create_vecs(self):
        r = VecP()
        check(self.lib.MatCreateVecs(self.p, C.byref(r), None))
        return Vec(r, self.lib)

    def create_vecs_left(self):
        r = VecP()
        check(self.lib.MatCreateVecs(self.p, None, C.byref(r)))
        return Vec(r, self.lib)

    def mult(self, x, y):
        check(self.lib.MatMult(self.p, x.p, y.p))

    @property
    def n(self):
        return self.p.contents.n

    def destroy(self):
        if self.p:
            check(self.lib.MatDestroy(C.byref(self.p)))
            self.p = None


SVD_LARGEST, SVD_SMALLEST = 0, 1
SVD_CONV_ABS, SVD_CONV_REL = 0, 1
SVD_ERROR_ABSOLUTE, SVD_ERROR_RELATIVE = 0, 1
SVD_CONVERGED_TOL, SVD_DIVERGED_ITS = 1, -1


class SVD:
    """``SVDCreate`` ... ``SVDDestroy`` with the argument meaning of the C interface."""

    def __init__(self, ctx_handle=None):
        self.lib = load()
        self.h = vp()
        check(self.lib.SVDCreate(ctx_handle, C.byref(self.h)))

    def set_operator(self, A):
        self._A = A
        check(self.lib.SVDSetOperator(self.h, A.p))

    def set_type(self, name):
        check(self.lib.SVDSetType(self.h, name.encode()))

    def get_type(self):
        s = C.c_char_p()
        check(self.lib.SVDGetType(self.h, C.byref(s)))
        return s.value.decode()

    def set_dimensions(self, nsv, ncv=0, mpd=0):
        check(self.lib.SVDSetDimensions(self.h, nsv, ncv, mpd))

    def set_tolerances(self, tol=0.0, max_it=0):
        check(self.lib.SVDSetTolerances(self.h, tol, max_it))

    def set_which(self, which):
        check(self.lib.SVDSetWhichSingularTriplets(self.h, SVD_SMALLEST if which in ("smallest", SVD_SMALLEST) else SVD_LARGEST))

    def set_convergence_test(self, conv):
        check(self.lib.SVDSetConvergenceTest(self.h, SVD_CONV_ABS if conv in ("abs", SVD_CONV_ABS) else SVD_CONV_REL))

    def set_initial_space(self, vecs):
        arr = (VecP * len(vecs))(*[v.p for v in vecs])
        check(self.lib.SVDSetInitialSpace(self.h, len(vecs), arr))

    def set_oneside(self, oneside=True):
        check(self.lib.SVDTRLanczosSetOneSide(self.h, 1 if oneside else 0))
        check(self.lib.SVDLanczosSetOneSide(self.h, 1 if oneside else 0))

    def set_bv_type(self, name):
        V, U = BVP(), BVP()
        check(self.lib.SVDGetBV(self.h, C.byref(V), C.byref(U)))
        check(self.lib.BVSetType(V, name.encode()))
        check(self.lib.BVSetType(U, name.encode()))

    def solve(self):
        check(self.lib.SVDSolve(self.h))

    def _geti(self, fn):
        v = C.c_int()
        check(fn(self.h, C.byref(v)))
        return v.value

    @property
    def its(self):
        return self._geti(self.lib.SVDGetIterationNumber)

    @property
    def nconv(self):
        return self._geti(self.lib.SVDGetConverged)

    @property
    def reason(self):
        return self._geti(self.lib.SVDGetConvergedReason)

    def dimensions(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.SVDGetDimensions(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def tolerances(self):
        t, m = C.c_double(), C.c_int()
        check(self.lib.SVDGetTolerances(self.h, C.byref(t), C.byref(m)))
        return t.value, m.value

    def get_singular_triplet(self, i, u=None, v=None):
        s = C.c_double()
        check(self.lib.SVDGetSingularTriplet(self.h, i, C.byref(s), u.p if u else None, v.p if v else None))
        return s.value

    def compute_error(self, i, etype=SVD_ERROR_RELATIVE):
        e = C.c_double()
        check(self.lib.SVDComputeError(self.h, i, etype, C.byref(e)))
        return e.value

    def destroy(self):
        if self.h:
            check(self.lib.SVDDestroy(C.byref(self.h)))
            self.h = None


def read_petsc_binary(filename):
    """(M, N, rowptr, colidx, val) of a PETSc binary matrix file, read by the C library."""
    lib = load()
    M, N = c_i64(), c_i64()
    rp, ci, va = vp(), vp(), vp()
    check(lib.KSMatReadPetscBinary(os.fsencode(filename), C.byref(M), C.byref(N), C.byref(rp), C.byref(ci), C.byref(va)))
    try:
        m = M.value
        rowptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(c_i64)), shape=(m + 1,)).copy()
        nz = int(rowptr[m])
        colidx = np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int32)), shape=(max(nz, 1),))[:nz].copy()
        val = np.ctypeslib.as_array(C.cast(va, C.POINTER(C.c_double)), shape=(max(nz, 1),))[:nz].copy()
    finally:
        for q in (rp, ci, va):
            lib.KSFree(q)
    return m, N.value, rowptr, colidx, val


def write_petsc_binary(filename, M, N, rowptr, colidx, val):
    lib = load()
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    check(lib.KSMatWritePetscBinary(os.fsencode(filename), M, N, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))


class ST:
    """The spectral transformation of an eigensolver (borrowed handle: EPSGetST)."""

    def __init__(self, h, lib):
        self.h, self.lib = h, lib

    def set_type(self, name):
        check(self.lib.STSetType(self.h, name.encode()))

    def get_type(self):
        p = C.c_char_p()
        check(self.lib.STGetType(self.h, C.byref(p)))
        return p.value.decode()

    def set_shift(self, sigma):
        check(self.lib.STSetShift(self.h, float(sigma)))

    def get_shift(self):
        v = C.c_double()
        check(self.lib.STGetShift(self.h, C.byref(v)))
        return v.value

    def set_ksp_type(self, name):
        check(self.lib.STSetKSPType(self.h, name.encode()))

    def get_ksp_type(self):
        p = C.c_char_p()
        check(self.lib.STGetKSPType(self.h, C.byref(p)))
        return p.value.decode()

    def set_ksp_tolerances(self, rtol=0.0, maxit=0, restart=0):
        check(self.lib.STSetKSPTolerances(self.h, rtol, maxit, restart))

    def ksp_iterations(self):
        a, b = c_i64(), c_i64()
        check(self.lib.STGetKSPIterations(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value


class EPS:
    """``EPSCreate`` ... ``EPSDestroy`` with the argument meaning of the C interface."""

    def __init__(self, ctx_handle=None):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.EPSCreate(ctx_handle, C.byref(h)))
        self.h = h
        self._keep = []

    def set_operators(self, A, B=None):
        self.A, self.B = A, B
        check(self.lib.EPSSetOperators(self.h, A.p, B.p if B is not None else None))

    def set_target(self, target):
        check(self.lib.EPSSetTarget(self.h, float(target)))

    def set_purify(self, flag):
        check(self.lib.EPSSetPurify(self.h, 1 if flag else 0))

    def set_deflation_space(self, vecs):
        arr = (VecP * max(len(vecs), 1))(*[v.p for v in vecs])
        check(self.lib.EPSSetDeflationSpace(self.h, len(vecs), arr))

    def st(self):
        h = C.c_void_p()
        check(self.lib.EPSGetST(self.h, C.byref(h)))
        return ST(h, self.lib)

    def set_type(self, name):
        check(self.lib.EPSSetType(self.h, name.encode()))

    def get_type(self):
        p = C.c_char_p()
        check(self.lib.EPSGetType(self.h, C.byref(p)))
        return p.value.decode()

    def set_problem_type(self, t):
        check(self.lib.EPSSetProblemType(self.h, t))

    def set_dimensions(self, nev, ncv=0, mpd=0):
        check(self.lib.EPSSetDimensions(self.h, nev, ncv, mpd))

    def set_tolerances(self, tol=0.0, max_it=0):
        check(self.lib.EPSSetTolerances(self.h, tol, max_it))

    def set_which(self, which):
        check(self.lib.EPSSetWhichEigenpairs(self.h, WHICH[which] if isinstance(which, str) else which))

    def set_initial_space(self, vecs):
        arr = (VecP * len(vecs))(*[v.p for v in vecs])
        check(self.lib.EPSSetInitialSpace(self.h, len(vecs), arr))

    def set_restart(self, keep):
        check(self.lib.EPSKrylovSchurSetRestart(self.h, keep))

    def set_locking(self, lock):
        check(self.lib.EPSKrylovSchurSetLocking(self.h, 1 if lock else 0))

    def set_lanczos_reorthog(self, name):
        check(self.lib.EPSLanczosSetReorthog(self.h, LANCZOS_REORTHOG[name]))

    def set_arnoldi_delayed(self, delayed=True):
        check(self.lib.EPSArnoldiSetDelayed(self.h, 1 if delayed else 0))

    def set_bv_type(self, name):
        bv = BVP()
        check(self.lib.EPSGetBV(self.h, C.byref(bv)))
        check(self.lib.BVSetType(bv, name.encode()))

    def bv(self):
        bv = BVP()
        check(self.lib.EPSGetBV(self.h, C.byref(bv)))
        return bv

    def set_monitor(self, fn):
        """fn(its, nconv, eigr, eigi, errest, nest) called after every restart."""
        def _m(eps, its, nconv, eigr, eigi, errest, nest, ctx):
            try:
                fn(its, nconv, np.ctypeslib.as_array(eigr, (nest,)), np.ctypeslib.as_array(eigi, (nest,)),
                   np.ctypeslib.as_array(errest, (nest,)), nest)
                return 0
            except Exception:       # pragma: no cover
                import traceback
                traceback.print_exc()
                return 83
        cb = MONITOR_FN(_m)
        self._keep.append(cb)
        check(self.lib.EPSMonitorSet(self.h, cb, None))

    def setup(self):
        check(self.lib.EPSSetUp(self.h))

    def solve(self):
        check(self.lib.EPSSolve(self.h))
        return self

    def _geti(self, fn):
        v = C.c_int()
        check(fn(self.h, C.byref(v)))
        return v.value

    @property
    def its(self):
        return self._geti(self.lib.EPSGetIterationNumber)

    @property
    def nconv(self):
        return self._geti(self.lib.EPSGetConverged)

    @property
    def reason(self):
        return self._geti(self.lib.EPSGetConvergedReason)

    def dimensions(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.EPSGetDimensions(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_eigenvalue(self, i):
        re, im = C.c_double(), C.c_double()
        check(self.lib.EPSGetEigenvalue(self.h, i, C.byref(re), C.byref(im)))
        return re.value, im.value

    def get_eigenpair(self, i, xr, xi=None):
        re, im = C.c_double(), C.c_double()
        check(self.lib.EPSGetEigenpair(self.h, i, C.byref(re), C.byref(im), xr.p, xi.p if xi else None))
        return re.value, im.value

    def error_estimate(self, i):
        r = C.c_double()
        check(self.lib.EPSGetErrorEstimate(self.h, i, C.byref(r)))
        return r.value

    def compute_error(self, i, etype=EPS_ERROR_RELATIVE):
        r = C.c_double()
        check(self.lib.EPSComputeError(self.h, i, etype, C.byref(r)))
        return r.value

    def counters(self):
        a, b = c_i64(), c_i64()
        check(self.lib.EPSGetCounters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def destroy(self):
        if self.h:
            check(self.lib.EPSDestroy(C.byref(self.h)))
            self.h = None
