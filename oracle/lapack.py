"""ctypes bindings to the LAPACK/BLAS routines the reference's DS class calls.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference reaches these through PETSc's BLAS/LAPACK link
(``include/slepcblaslapack.h``); here they come from the OpenBLAS shared
library that the scipy wheel bundles (LP64 interface, symbols prefixed
``scipy_``).  All matrices are column-major (numpy ``order='F'``) float64.
"""
import ctypes as C
import glob
import os

import numpy as np
import scipy

_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    base = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
    cands = sorted(glob.glob(os.path.join(base, "libscipy_openblas*.so")))
    if not cands:
        raise RuntimeError("oracle: scipy's bundled OpenBLAS not found under " + base)
    _lib = C.CDLL(cands[0])
    return _lib


def _fn(name):
    return getattr(_load(), "scipy_" + name + "_")


_i = C.c_int
_d = C.c_double


def _pi(v):
    return C.byref(_i(int(v)))


def _pd(a):
    assert a.dtype == np.float64
    return a.ctypes.data_as(C.POINTER(_d))


def _ch(c):
    return C.c_char_p(c.encode())


def dlartg(f, g):
    """LAPACK dlartg: plane rotation, returns (c, s, r)."""
    cs, sn, r = _d(), _d(), _d()
    _fn("dlartg")(C.byref(_d(f)), C.byref(_d(g)), C.byref(cs), C.byref(sn), C.byref(r))
    return cs.value, sn.value, r.value


def drot(n, x, y, c, s):
    """BLAS drot on two contiguous 1-D views (in place)."""
    assert x.flags.c_contiguous or x.flags.f_contiguous
    assert y.flags.c_contiguous or y.flags.f_contiguous
    _fn("drot")(_pi(n), _pd(x), _pi(1), _pd(y), _pi(1), C.byref(_d(c)), C.byref(_d(s)))


def dsteqr(compz, n, d, e, Z, ldz):
    """LAPACK dsteqr.  d,e are 1-D views modified in place; Z is a flat view of
    the column-major matrix starting at the (0,0) entry of the block."""
    work = np.zeros(max(1, 2 * n - 2))
    info = _i(0)
    _fn("dsteqr")(_ch(compz), _pi(n), _pd(d), _pd(e), _pd(Z), _pi(ldz), _pd(work),
                  C.byref(info), C.c_size_t(1))
    return info.value


def dgehrd(n, ilo, ihi, A, lda):
    tau = np.zeros(max(1, n))
    lwork = max(1, lda * lda)
    work = np.zeros(lwork)
    info = _i(0)
    _fn("dgehrd")(_pi(n), _pi(ilo), _pi(ihi), _pd(A), _pi(lda), _pd(tau), _pd(work),
                  _pi(lwork), C.byref(info))
    return tau, info.value


def dorghr(n, ilo, ihi, Q, ldq, tau):
    lwork = max(1, ldq * ldq)
    work = np.zeros(lwork)
    info = _i(0)
    _fn("dorghr")(_pi(n), _pi(ilo), _pi(ihi), _pd(Q), _pi(ldq), _pd(tau), _pd(work),
                  _pi(lwork), C.byref(info))
    return info.value


def dhseqr(job, compz, n, ilo, ihi, H, ldh, wr, wi, Z, ldz):
    lwork = max(1, ldh * ldh)
    work = np.zeros(lwork)
    info = _i(0)
    _fn("dhseqr")(_ch(job), _ch(compz), _pi(n), _pi(ilo), _pi(ihi), _pd(H), _pi(ldh),
                  _pd(wr), _pd(wi), _pd(Z), _pi(ldz), _pd(work), _pi(lwork),
                  C.byref(info), C.c_size_t(1), C.c_size_t(1))
    return info.value


def dtrexc(compq, n, T, ldt, Q, ldq, ifst, ilst):
    work = np.zeros(max(1, n))
    info = _i(0)
    i1, i2 = _i(ifst), _i(ilst)
    _fn("dtrexc")(_ch(compq), _pi(n), _pd(T), _pi(ldt), _pd(Q), _pi(ldq), C.byref(i1),
                  C.byref(i2), _pd(work), C.byref(info), C.c_size_t(1))
    return info.value


def dtrevc(side, howmny, select, n, T, ldt, VL, ldvl, VR, ldvr, mm):
    """LAPACK dtrevc.  ``select`` is an int32 array (Fortran LOGICAL) or None."""
    work = np.zeros(max(1, 3 * n))
    info = _i(0)
    mout = _i(0)
    if select is None:
        select = np.zeros(max(1, n), dtype=np.int32)
    sel = np.ascontiguousarray(select, dtype=np.int32)
    _fn("dtrevc")(_ch(side), _ch(howmny), sel.ctypes.data_as(C.POINTER(_i)), _pi(n),
                  _pd(T), _pi(ldt), _pd(VL), _pi(ldvl), _pd(VR), _pi(ldvr), _pi(mm),
                  C.byref(mout), _pd(work), C.byref(info), C.c_size_t(1), C.c_size_t(1))
    select[...] = sel
    return mout.value, info.value
