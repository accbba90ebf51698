"""ctypes binding of ``oracle/ckern.c``: the compiled, row-parallel CPU kernels of the
restatement (CSR product, BV gemv pair, restart product, scale, norm).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the checker's fast back end and the CPU
baseline of ``bench.py``; the product never loads it.
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

from . import bv as bvmod

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libckern.so")
_lib = None
c_i64, c_dp = C.c_int64, C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.ck_max_threads.restype = C.c_int
        lib.ck_spmv.argtypes = [C.c_int, c_i64, vp, vp, vp, vp, vp]
        lib.ck_dotvec.argtypes = [C.c_int, c_i64, C.c_int, vp, c_i64, vp, vp, vp]
        lib.ck_multvec.argtypes = [C.c_int, c_i64, C.c_int, C.c_double, C.c_double, vp, c_i64, vp, vp]
        lib.ck_multinplace.argtypes = [C.c_int, c_i64, C.c_int, vp, c_i64, vp, C.c_int, C.c_int, C.c_int]
        lib.ck_scale.argtypes = [C.c_int, c_i64, C.c_double, vp]
        lib.ck_copy.argtypes = [C.c_int, c_i64, vp, vp]
        lib.ck_sumsq.argtypes = [C.c_int, c_i64, vp, vp]
        lib.ck_sumsq.restype = C.c_double
        lib.ck_triad.argtypes = [C.c_int, c_i64, vp, vp, vp, C.c_double]
        for f in (lib.ck_spmv, lib.ck_dotvec, lib.ck_multvec, lib.ck_multinplace, lib.ck_scale, lib.ck_copy, lib.ck_triad):
            f.restype = None
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data


class CSR:
    """y = A x with the compiled loop; ``bytes`` accumulates the CSR byte model of the products."""

    def __init__(self, kern, rowptr, colidx, val):
        self.k = kern
        self.rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.ci = np.ascontiguousarray(colidx, dtype=np.int32)
        self.va = np.ascontiguousarray(val, dtype=np.float64)
        self.n = self.rp.size - 1
        self.bytes = 0.0
        self.per_product = 12.0 * float(self.rp[-1]) + 8.0 * (self.n + 1) + 16.0 * self.n

    def mult(self, x, y):
        assert x.flags.c_contiguous and y.flags.c_contiguous and x.dtype == np.float64 and y.dtype == np.float64
        self.k.lib.ck_spmv(self.k.nt, self.n, _p(self.rp), _p(self.ci), _p(self.va), _p(x), _p(y))
        self.bytes += self.per_product


class CBV(bvmod.BV):
    """The restatement's BV with the op table of the hot path on the compiled kernels; everything
    else (Gram-Schmidt control flow, buffers, block operations) is inherited."""
    kern = None

    def _colptr(self, j):
        return self.array.ctypes.data + 8 * (self.nc + j) * self.n

    def op_dotvec(self, y, q=None):
        kk = self.k - self.l
        self.bytes += 8.0 * self.n * (kk + 1)
        K = self.kern
        loc = np.zeros(max(kk, 1))
        if self.n and kk:
            assert y.strides == (8,)
            K.lib.ck_dotvec(K.nt, self.n, kk, self._colptr(self.l), self.n, _p(y), _p(loc), _p(K.work(kk)))
        res = self.comm.allreduce_sum(loc[:kk])
        if q is None:
            self.get_buffer()[:kk] = res
            return self.buffer[:kk]
        q[:kk] = res
        return q

    def op_multvec(self, alpha, beta, y, q=None):
        kk = self.k - self.l
        self.bytes += 8.0 * self.n * (kk + 2)
        qq = np.ascontiguousarray(self.get_buffer()[:kk] if q is None else np.asarray(q)[:kk], dtype=np.float64)
        if self.n:
            assert y.strides == (8,)
            K = self.kern
            K.lib.ck_multvec(K.nt, self.n, kk, float(alpha), float(beta), self._colptr(self.l), self.n, _p(qq), _p(y))

    def op_multinplace(self, Q, s, e, trans=False):
        l, k = self.l, self.k
        self.bytes += 8.0 * self.n * ((k - l) + (e - s))
        if not self.n or e <= s:
            return
        B = np.asfortranarray(Q[s:e, l:k].T if trans else Q[l:k, s:e], dtype=np.float64)     # (k-l) x (e-s)
        K = self.kern
        # columns relative to l: the output columns are s-l .. e-l of the active block
        Bfull = np.zeros((k - l, e - l), order="F")
        Bfull[:, s - l:] = B
        K.lib.ck_multinplace(K.nt, self.n, k - l, self._colptr(l), self.n, _p(Bfull), k - l, s - l, e - l)

    def op_scale(self, j, alpha):
        cols = range(self.l, self.k) if j < 0 else [j]
        self.bytes += 16.0 * self.n * len(cols)
        if alpha == 1.0:
            return
        for c in cols:
            self.kern.lib.ck_scale(self.kern.nt, self.n, float(alpha), self._colptr(c))

    def op_norm(self, j, ntype=bvmod.NORM_2):
        if j < 0 or ntype not in (bvmod.NORM_2, bvmod.NORM_FROBENIUS):
            return super().op_norm(j, ntype)
        self.bytes += 8.0 * self.n
        K = self.kern
        loc = np.array([K.lib.ck_sumsq(K.nt, self.n, self._colptr(j), _p(K.work(1)))])
        return float(np.sqrt(self.comm.allreduce_sum(loc)[0]))

    def copy_column(self, j, i):
        if j == i:
            return
        self.bytes += 16.0 * self.n
        self.kern.lib.ck_copy(self.kern.nt, self.n, self._colptr(j), self._colptr(i))


class Kernels:
    def __init__(self, threads=None):
        self.lib = load()
        self.nt = int(threads or self.lib.ck_max_threads())
        self._work = np.zeros(1)

    def work(self, k):
        need = self.nt * max(k, 1)
        if self._work.size < need:
            self._work = np.zeros(need)
        return self._work

    def csr(self, rowptr, colidx, val):
        return CSR(self, rowptr, colidx, val)

    def bv(self, n, m, N=None, comm=None, nc=0):
        b = CBV(n, m, N=N, comm=comm, nc=nc)
        b.kern = self
        return b

    def stream_triad_GBps(self, nbytes=1 << 27, reps=3):
        """Best of `reps` STREAM triads over three arrays of `nbytes` each."""
        n = max(1024, int(nbytes) // 8)
        a, b, c = np.zeros(n), np.ones(n), np.full(n, 2.0)
        best = 0.0
        for _ in range(reps + 1):
            t = time.perf_counter()
            self.lib.ck_triad(self.nt, n, _p(a), _p(b), _p(c), 3.0)
            dt = time.perf_counter() - t
            best = max(best, 24.0 * n / dt / 1e9)
        return best
