"""ctypes binding of libksgpu (the C ABI declared in ``include/ksgpu.h``).

This module is plumbing only: it loads the in-tree shared library, declares the
argument types and turns error codes into exceptions.  There is no CPU
fallback: if the library is missing, importing a symbol from it fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libksgpu.so")

c_i64 = C.c_int64
c_dp = C.POINTER(C.c_double)


class KsgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libksgpu error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    EVENTS = ["dotvec", "multvec", "multinplace", "scale", "norm", "copy", "matmult",
              "mult", "dot", "vecops", "dotvec_fused"]
    _fields_ = [("calls", c_i64 * 11), ("bytes", C.c_double * 11), ("flops", C.c_double * 11),
                ("ms", C.c_double * 11)]

    def as_dict(self):
        return {ev: {"calls": int(self.calls[i]), "bytes": float(self.bytes[i]),
                     "flops": float(self.flops[i]), "ms": float(self.ms[i])}
                for i, ev in enumerate(self.EVENTS)}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(c_i64), C.c_void_p,
                          C.POINTER(c_i64))

# name -> (restype, argtypes); every symbol include/ksgpu.h declares
SIGNATURES = {
    "ksg_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ksg_ctx_destroy": (C.c_int, [C.c_void_p]),
    "ksg_ctx_sync": (C.c_int, [C.c_void_p]),
    "ksg_ctx_flush": (C.c_int, [C.c_void_p]),
    "ksg_ctx_touch": (C.c_int, [C.c_void_p]),
    "ksg_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "ksg_ctx_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ksg_last_error": (C.c_char_p, []),
    "ksg_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ksg_ctx_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ksg_ctx_comm_init_callbacks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, ALLREDUCE_FN,
                                              EXCHANGE_FN, C.c_void_p]),
    "ksg_ctx_comm_finalize": (C.c_int, [C.c_void_p]),
    "ksg_ctx_comm_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ksg_ctx_comm_selftest": (C.c_int, [C.c_void_p]),
    "ksg_vec_alloc": (C.c_int, [C.c_void_p, c_i64, C.POINTER(C.c_void_p)]),
    "ksg_vec_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ksg_vec_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64]),
    "ksg_vec_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64]),
    "ksg_vec_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64]),
    "ksg_vec_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, c_i64]),
    "ksg_vec_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, c_i64]),
    "ksg_vec_axpy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, c_i64]),
    "ksg_vec_dot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, c_dp]),
    "ksg_vec_norm2": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_dp]),
    "ksg_vec_dot_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, c_dp]),
    "ksg_vec_fill_hash": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_uint64]),
    "ksg_bv_create": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_int, C.POINTER(C.c_void_p)]),
    "ksg_bv_destroy": (C.c_int, [C.c_void_p]),
    "ksg_bv_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ksg_bv_set_active": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ksg_bv_sizes": (C.c_int, [C.c_void_p, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(C.c_int),
                               C.POINTER(c_i64)]),
    "ksg_bv_column": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ksg_bv_touch": (C.c_int, [C.c_void_p]),
    "ksg_bv_dotvec": (C.c_int, [C.c_void_p, C.c_void_p, c_dp]),
    "ksg_bv_dotvec_local": (C.c_int, [C.c_void_p, C.c_void_p, c_dp]),
    "ksg_bv_dotvec_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ksg_bv_multvec": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_void_p, c_dp]),
    "ksg_bv_multinplace": (C.c_int, [C.c_void_p, c_dp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ksg_bv_scale": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "ksg_bv_norm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp]),
    "ksg_bv_norm_local": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp]),
    "ksg_bv_copy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ksg_bv_copy_column": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ksg_bv_mult": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_void_p, c_dp, C.c_int]),
    "ksg_bv_dot": (C.c_int, [C.c_void_p, C.c_void_p, c_dp, C.c_int]),
    "ksg_bv_tsqr": (C.c_int, [C.c_void_p, c_dp, C.c_int]),
    "ksg_mat_hot_columns": (C.c_int, [C.c_void_p, C.POINTER(c_i64)]),
    "ksg_bv_set_num_constraints": (C.c_int, [C.c_void_p, C.c_int]),
    "ksg_bv_matmult": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ksg_mat_create_csr": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_void_p)]),
    "ksg_mat_destroy": (C.c_int, [C.c_void_p]),
    "ksg_mat_create_csr_rect": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(C.c_void_p)]),
    "ksg_mat_col_sizes": (C.c_int, [C.c_void_p, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64)]),
    "ksg_mat_get_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ksg_mat_transpose": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ksg_red_open": (C.c_int, [C.c_void_p]),
    "ksg_red_post_dotvec": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "ksg_red_post_dot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.POINTER(C.c_int)]),
    "ksg_red_close": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "ksg_mat_mult": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ksg_mat_residual_norm": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_double)]),
    "ksg_mat_sizes": (C.c_int, [C.c_void_p, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64),
                                C.POINTER(c_i64)]),
    "ksg_mat_info": (C.c_int, [C.c_void_p, C.POINTER(c_i64)]),
    "ksg_mat_traffic_model": (C.c_int, [C.c_void_p, c_dp, c_dp]),
    "ksg_mat_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "ksg_bins_host_check": (C.c_int, [c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_i64, C.c_void_p, C.c_void_p,
                                      C.POINTER(c_i64)]),
    "ksg_stats_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ksg_stats_get": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "ksg_stats_reset": (C.c_int, [C.c_void_p]),
}

_lib = None


def load():
    """Loads libksgpu.so (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libksgpu.so not built: run `python __graft_entry__.py build` "
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
        raise KsgError(rc, load().ksg_last_error().decode(errors="replace"))


def _dptr(a):
    return a.ctypes.data_as(c_dp)


OPT_FUSE_DOTS = 1
OPT_DOT_GRID = 2
OPT_MV_GRID = 3
OPT_LAZY_UPDATE = 4
OPT_HOT_COLS = 5
OPT_BINS = 7


def bins_host_check(rowptr, colidx, val, x, ncols=None, has_diag=True, dshift=0):
    """y = A x formed on the host the way the binned product forms it (no device needed), and
    {binned entries, runs, row blocks, value mode, entries left in the CSR}."""
    import numpy as np
    lib = load()
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    n = len(rp) - 1
    y = np.empty(n)
    info = (c_i64 * 6)()
    check(lib.ksg_bins_host_check(n, len(xx) if ncols is None else ncols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data,
                                  1 if has_diag else 0, dshift, xx.ctypes.data, y.ctypes.data, info))
    return y, dict(zip(("binned", "runs", "row_blocks", "value_mode", "csr_entries", "diagonal_folded"), (int(v) for v in info)))


class Context:
    """A libksgpu context on one HIP device."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.ksg_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device
        self._keep = []

    def close(self):
        if self.h:
            self.lib.ksg_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(self.lib.ksg_ctx_sync(self.h))

    # -- communicator -----------------------------------------------------------
    def comm_init_rccl(self, rank, nranks, uid):
        buf = C.create_string_buffer(bytes(uid), 128)
        check(self.lib.ksg_ctx_comm_init_rccl(self.h, rank, nranks, buf))

    def comm_finalize(self):
        check(self.lib.ksg_ctx_comm_finalize(self.h))

    def comm_selftest(self):
        check(self.lib.ksg_ctx_comm_selftest(self.h))

    def unique_id(self):
        buf = C.create_string_buffer(128)
        check(self.lib.ksg_comm_unique_id(buf))
        return buf.raw

    def comm_init_callbacks(self, rank, nranks, allreduce, exchange, on_device=False):
        """allreduce(addr, count, op) and exchange(send_addr, sendoff, recv_addr, recvoff)
        are Python callables working on raw addresses (host unless on_device)."""
        def _ar(user, buf, count, op):
            try:
                allreduce(buf, count, op)
                return 0
            except Exception:       # pragma: no cover - reported through the error code
                import traceback
                traceback.print_exc()
                return 1

        def _ex(user, sbuf, soff, rbuf, roff):
            try:
                so = [soff[i] for i in range(nranks + 1)]
                ro = [roff[i] for i in range(nranks + 1)]
                exchange(sbuf, so, rbuf, ro)
                return 0
            except Exception:       # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        cb1, cb2 = ALLREDUCE_FN(_ar), EXCHANGE_FN(_ex)
        self._keep += [cb1, cb2]
        check(self.lib.ksg_ctx_comm_init_callbacks(self.h, rank, nranks, 1 if on_device else 0,
                                                   cb1, cb2, None))

    # -- vectors ------------------------------------------------------------------
    def vec_alloc(self, n):
        p = C.c_void_p()
        check(self.lib.ksg_vec_alloc(self.h, n, C.byref(p)))
        return p.value

    def vec_free(self, p):
        check(self.lib.ksg_vec_free(self.h, p))

    def upload(self, dptr, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        check(self.lib.ksg_vec_upload(self.h, dptr, a.ctypes.data, a.size))

    def download(self, dptr, n):
        out = np.empty(n, dtype=np.float64)
        check(self.lib.ksg_vec_download(self.h, out.ctypes.data, dptr, n))
        return out

    def vec_copy(self, x, y, n):
        check(self.lib.ksg_vec_copy(self.h, x, y, n))

    def vec_set(self, x, a, n):
        check(self.lib.ksg_vec_set(self.h, x, a, n))

    def vec_scale(self, x, a, n):
        check(self.lib.ksg_vec_scale(self.h, x, a, n))

    def vec_axpy(self, y, a, x, n):
        check(self.lib.ksg_vec_axpy(self.h, y, a, x, n))

    def vec_dot(self, x, y, n):
        r = C.c_double()
        check(self.lib.ksg_vec_dot(self.h, x, y, n, C.byref(r)))
        return r.value

    def vec_dot_local(self, x, y, n):
        r = C.c_double()
        check(self.lib.ksg_vec_dot_local(self.h, x, y, n, C.byref(r)))
        return r.value

    def vec_norm2(self, x, n):
        r = C.c_double()
        check(self.lib.ksg_vec_norm2(self.h, x, n, C.byref(r)))
        return r.value

    def vec_fill_hash(self, x, n, row0=0, seed=0x12345678):
        check(self.lib.ksg_vec_fill_hash(self.h, x, n, row0, seed))

    # -- statistics ---------------------------------------------------------------
    def stats(self):
        s = Stats()
        check(self.lib.ksg_stats_get(self.h, C.byref(s)))
        return s.as_dict()

    def stats_reset(self):
        check(self.lib.ksg_stats_reset(self.h))

    def set_option(self, option, value):
        check(self.lib.ksg_ctx_set_option(self.h, option, int(value)))

    def enable_timing(self, on=True):
        check(self.lib.ksg_stats_enable_timing(self.h, 1 if on else 0))


class DeviceBV:
    """Basis vectors in device memory (``ksg_bv_*``)."""

    def __init__(self, ctx, n, m, N=None):
        self.ctx = ctx
        self.lib = ctx.lib
        h = C.c_void_p()
        check(self.lib.ksg_bv_create(ctx.h, n, N if N is not None else n, m, C.byref(h)))
        self.h = h
        self.n, self.m = n, m
        self.l, self.k = 0, m

    def destroy(self):
        if self.h:
            self.lib.ksg_bv_destroy(self.h)
            self.h = None

    def set_active(self, l, k):
        check(self.lib.ksg_bv_set_active(self.h, l, k))
        self.l, self.k = l, k

    def column(self, j):
        p = self.lib.ksg_bv_column(self.h, j)
        if not p:
            raise KsgError(1, self.lib.ksg_last_error().decode())
        return p

    def touch(self):
        check(self.lib.ksg_bv_touch(self.h))

    def set_column(self, j, arr):
        self.ctx.upload(self.column(j), arr)

    def get_column(self, j):
        return self.ctx.download(self.column(j), self.n)

    def to_numpy(self):
        return np.stack([self.get_column(j) for j in range(self.m)], axis=1) if self.m else None

    def from_numpy(self, A):
        for j in range(A.shape[1]):
            self.set_column(j, A[:, j])

    def dotvec(self, y_dev, local=False):
        q = np.zeros(max(self.k - self.l, 1))
        fn = self.lib.ksg_bv_dotvec_local if local else self.lib.ksg_bv_dotvec
        check(fn(self.h, y_dev, _dptr(q)))
        return q[:self.k - self.l]

    def dotvec_noalloc(self, y_dev, q_dev):
        """Local products into the device array q_dev; no allocation, no synchronisation."""
        check(self.lib.ksg_bv_dotvec_noalloc(self.h, y_dev, q_dev))

    def dotvec_cols(self, l, k, j):
        """Products of columns [l,k) with column j."""
        self.set_active(l, k)
        return self.dotvec(self.column(j))

    def multvec(self, alpha, beta, y_dev, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        check(self.lib.ksg_bv_multvec(self.h, alpha, beta, y_dev, _dptr(q)))

    def multinplace(self, Q, s, e, trans=False):
        Qf = np.asfortranarray(Q, dtype=np.float64)
        check(self.lib.ksg_bv_multinplace(self.h, _dptr(Qf), Qf.shape[0], s, e, 1 if trans else 0))

    def scale(self, j, alpha):
        check(self.lib.ksg_bv_scale(self.h, j, alpha))

    def norm(self, j, ntype=1, local=False):
        r = C.c_double()
        fn = self.lib.ksg_bv_norm_local if local else self.lib.ksg_bv_norm
        check(fn(self.h, j, ntype, C.byref(r)))
        return r.value

    def copy_to(self, W):
        check(self.lib.ksg_bv_copy(self.h, W.h))

    def copy_column(self, j, i):
        check(self.lib.ksg_bv_copy_column(self.h, j, i))

    def mult(self, alpha, beta, X, Q):
        if Q is None:
            check(self.lib.ksg_bv_mult(self.h, alpha, beta, X.h, None, 1))
        else:
            Qf = np.asfortranarray(Q, dtype=np.float64)
            check(self.lib.ksg_bv_mult(self.h, alpha, beta, X.h, _dptr(Qf), Qf.shape[0]))

    def dot(self, Y, M):
        """M(Y.l:Y.k, X.l:X.k) = Y^T X with X = self; M is a Fortran-ordered array."""
        assert M.flags.f_contiguous
        check(self.lib.ksg_bv_dot(self.h, Y.h, _dptr(M), M.shape[0]))
        return M

    def matmult(self, A, W):
        check(self.lib.ksg_bv_matmult(self.h, A.h, W.h))

    def tsqr(self, R=None):
        """Columns [0,k) <- Q of their QR factorisation; R (Fortran-ordered, at least k x k) receives the triangle."""
        if R is None:
            check(self.lib.ksg_bv_tsqr(self.h, None, 1))
        else:
            assert R.flags.f_contiguous
            check(self.lib.ksg_bv_tsqr(self.h, _dptr(R), R.shape[0]))
        return R


class DeviceCSR:
    """Row block of a CSR matrix in device memory (``ksg_mat_*``)."""

    def __init__(self, ctx, rowptr, colidx, val, N=None, row0=0, cols=None):
        """``cols`` = (nc_local, NC, col0) for an operator whose columns are laid out
        differently from its rows (rectangular); default: square, columns like rows."""
        self.ctx = ctx
        self.lib = ctx.lib
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        ci = np.ascontiguousarray(colidx, dtype=np.int32)
        va = np.ascontiguousarray(val, dtype=np.float64)
        n = rp.size - 1
        h = C.c_void_p()
        if cols is None:
            check(self.lib.ksg_mat_create_csr(ctx.h, n, N if N is not None else n, row0, rp.ctypes.data,
                                              ci.ctypes.data, va.ctypes.data, C.byref(h)))
        else:
            check(self.lib.ksg_mat_create_csr_rect(ctx.h, n, N if N is not None else n, row0, cols[0], cols[1], cols[2],
                                                   rp.ctypes.data, ci.ctypes.data, va.ctypes.data, C.byref(h)))
        self._adopt(h)

    def _adopt(self, h):
        self.h = h
        n, N, row0, nnz = c_i64(), c_i64(), c_i64(), c_i64()
        check(self.lib.ksg_mat_sizes(h, C.byref(n), C.byref(N), C.byref(row0), C.byref(nnz)))
        nc, NC, col0 = c_i64(), c_i64(), c_i64()
        check(self.lib.ksg_mat_col_sizes(h, C.byref(nc), C.byref(NC), C.byref(col0)))
        self.n, self.N, self.row0, self.nnz = n.value, N.value, row0.value, nnz.value
        self.nc, self.NC, self.col0 = nc.value, NC.value, col0.value

    def transpose(self):
        """A^T as a new operator (collective)."""
        h = C.c_void_p()
        check(self.lib.ksg_mat_transpose(self.h, C.byref(h)))
        t = DeviceCSR.__new__(DeviceCSR)
        t.ctx, t.lib = self.ctx, self.lib
        t._adopt(h)
        return t

    def get_csr(self):
        """(rowptr, colidx, val) of the local rows, global column indices."""
        rp = np.zeros(self.n + 1, dtype=np.int64)
        ci = np.zeros(max(self.nnz, 1), dtype=np.int32)
        va = np.zeros(max(self.nnz, 1), dtype=np.float64)
        check(self.lib.ksg_mat_get_csr(self.h, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))
        return rp, ci[:self.nnz], va[:self.nnz]

    def destroy(self):
        if self.h:
            self.lib.ksg_mat_destroy(self.h)
            self.h = None

    def mult(self, x_dev, y_dev):
        check(self.lib.ksg_mat_mult(self.h, x_dev, y_dev))

    def residual_norm(self, kr, ki, x_dev, xi_dev=None, work_dev=None):
        """||A x - kr x + ki xi||_2 (device pointers; xi and work may be None)."""
        r = C.c_double()
        check(self.lib.ksg_mat_residual_norm(self.h, kr, ki, x_dev, xi_dev, work_dev, C.byref(r)))
        return r.value

    def hot_columns(self):
        n = c_i64()
        check(self.lib.ksg_mat_hot_columns(self.h, C.byref(n)))
        return n.value

    def traffic_model(self):
        """(executed, csr_model) bytes of one product on this rank."""
        a, b = C.c_double(), C.c_double()
        check(self.lib.ksg_mat_traffic_model(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def describe(self):
        buf = C.create_string_buffer(1024)
        check(self.lib.ksg_mat_describe(self.h, buf, 1024))
        return buf.value.decode()

    def info(self):
        a = (c_i64 * 4)()
        check(self.lib.ksg_mat_info(self.h, a))
        return {"stream_tiles": a[0], "group_tiles": a[1], "long_rows": a[2], "ghosts": a[3]}
