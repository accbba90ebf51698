"""TEST-ONLY plug-ins for the host solver: a Vec type, a BV type and a shell
operator whose kernels are the numpy oracle.

They are registered through the same seams a C plug-in would use
(``VecCreateWithOps``, ``BVRegister``, ``MatCreateShell``), which lets the
"not gpu" tests drive the C host logic (Gram-Schmidt control flow, Krylov-Schur
restarts, DS) on a machine without a GPU.  Nothing here is reachable from the
product: the package never imports ``tests`` or ``oracle``.
"""
import ctypes as C

import numpy as np

from oracle import bv as obv
from oracle.eps import splitmix64_uniform
from slepc_amd import kseps as K

_lib = K.load()
_arrays = {}          # address -> numpy array owned by a "numpy" Vec
_bvs = {}             # id -> oracle BV
_next_id = [1]


class Comm:
    """Reduction hook shared by all plug-in objects (identity on one rank)."""
    rank, size = 0, 1

    def allreduce_sum(self, a):
        return a

    def allreduce_max(self, a):
        return a


COMM = [Comm()]


def _view(vecp):
    v = vecp.contents
    if not v.n:
        return np.zeros(0)
    return np.ctypeslib.as_array((C.c_double * v.n).from_address(v.array))


def _guard(fn):
    def wrapped(*a):
        try:
            r = fn(*a)
            return 0 if r is None else r
        except Exception:       # pragma: no cover - surfaced through the error code
            import traceback
            traceback.print_exc()
            return K.__dict__.get("PETSC_ERR_USER", 83)
    return wrapped


# ---- Vec type "numpy" -----------------------------------------------------------
@_guard
def _v_alloc(v):
    a = np.zeros(max(v.contents.n, 1))
    _arrays[a.ctypes.data] = a
    v.contents.array = a.ctypes.data
    v.contents.own = a.ctypes.data


@_guard
def _v_free(v):
    if v.contents.own:
        _arrays.pop(v.contents.own, None)
        v.contents.own = None


@_guard
def _v_copy(x, y):
    _view(y)[:] = _view(x)


@_guard
def _v_set(x, a):
    _view(x)[:] = a


@_guard
def _v_scale(x, a):
    _view(x)[:] *= a


@_guard
def _v_axpy(y, a, x):
    _view(y)[:] += a * _view(x)


@_guard
def _v_dot(x, y, r):
    r[0] = float(COMM[0].allreduce_sum(np.array([_view(x) @ _view(y)]))[0])


@_guard
def _v_norm2(x, r):
    xv = _view(x)
    r[0] = float(np.sqrt(COMM[0].allreduce_sum(np.array([xv @ xv]))[0]))


@_guard
def _v_fillhash(x, seed):
    v = x.contents
    g = np.arange(v.row0, v.row0 + v.n, dtype=np.uint64)
    _view(x)[:] = splitmix64_uniform(g, seed=seed)


@_guard
def _v_tohost(x, h):
    n = x.contents.n
    if n:
        np.ctypeslib.as_array(h, (n,))[:] = _view(x)


@_guard
def _v_fromhost(x, h):
    n = x.contents.n
    if n:
        _view(x)[:] = np.ctypeslib.as_array(h, (n,))


_F = dict(K.VecOps._fields_)
VEC_OPS = K.VecOps(_F["alloc"](_v_alloc), _F["free"](_v_free), _F["copy"](_v_copy), _F["set"](_v_set),
                   _F["scale"](_v_scale), _F["axpy"](_v_axpy), _F["dot"](_v_dot), _F["norm2"](_v_norm2),
                   _F["fillhash"](_v_fillhash), _F["tohost"](_v_tohost), _F["fromhost"](_v_fromhost))


def vec_create(n, N=None, row0=0):
    p = K.VecP()
    K.check(_lib.VecCreateWithOps(C.byref(VEC_OPS), None, n, N if N is not None else n, row0, 1, C.byref(p)))
    return K.Vec(p, _lib)


# ---- BV type "oracle" -------------------------------------------------------------
def _obv(bvp):
    b = bvp.contents
    o = _bvs[b.data]
    if o.nc != b.nc:                      # the interface layer owns nc (svec reads bv->nc at every op)
        total = o.nc + o.m
        o.nc, o.m = b.nc, total - b.nc
    o.l, o.k = b.l, b.k
    return o


def _q(ptr, n):
    return np.ctypeslib.as_array(ptr, (max(n, 1),))[:n]


@_guard
def _b_mult(Y, alpha, beta, X, Q, ldq):
    oy, ox = _obv(Y), _obv(X)
    Qm = None
    if Q:
        ncols = Y.contents.k
        Qm = np.ctypeslib.as_array(Q, (ncols * ldq,)).reshape((ldq, ncols), order="F")
    oy.op_mult(alpha, beta, ox, Qm)


@_guard
def _b_multvec(X, alpha, beta, y, q):
    o = _obv(X)
    o.op_multvec(alpha, beta, _view(y), _q(q, o.k - o.l))


def _mip(V, Q, ldq, s, e, trans):
    o = _obv(V)
    ncols = max(o.k, e)
    Qm = np.ctypeslib.as_array(Q, (ncols * ldq,)).reshape((ldq, ncols), order="F")
    o.op_multinplace(Qm, s, e, trans)


@_guard
def _b_multinplace(V, Q, ldq, s, e):
    _mip(V, Q, ldq, s, e, False)


@_guard
def _b_multinplacetrans(V, Q, ldq, s, e):
    _mip(V, Q, ldq, s, e, True)


@_guard
def _b_dot(X, Y, M, ldm):
    ox, oy = _obv(X), _obv(Y)
    Mm = np.ctypeslib.as_array(M, (ldm * ox.k,)).reshape((ldm, ox.k), order="F")
    ox.op_dot(oy, Mm)


@_guard
def _b_dotvec(X, y, q):
    o = _obv(X)
    o.op_dotvec(_view(y), _q(q, o.k - o.l))


@_guard
def _b_dotvec_local(X, y, q):
    o = _obv(X)
    _q(q, o.k - o.l)[:] = o.array[:, o.nc + o.l:o.nc + o.k].T @ _view(y)


@_guard
def _b_scale(bv, j, alpha):
    _obv(bv).op_scale(j, alpha)


@_guard
def _b_norm(bv, j, t, val):
    val[0] = _obv(bv).op_norm(j, t)


@_guard
def _b_matmult(V, A, W):
    ov, ow = _obv(V), _obv(W)
    mv = _shell_ctx[A.contents.data]
    for j in range(ov.k - ov.l):
        mv(ov.col(ov.l + j), ow.col(ow.l + j))


@_guard
def _b_copy(V, W):
    ov, ow = _obv(V), _obv(W)
    ow.array[:, ow.nc + ow.l:ow.nc + ow.l + ov.k - ov.l] = ov.active()


@_guard
def _b_resize(bv, m, copy):
    b = bv.contents
    old = _bvs[b.data]
    new = obv.BV(old.n, m, N=old.N, comm=COMM[0])
    if copy:
        mc = min(m, old.m)
        new.array[:, :mc] = old.array[:, :mc]
    _bvs[b.data] = new


@_guard
def _b_getcolumn(bv, j, v):
    b = bv.contents
    l = 0 if b.ci[0] == -b.nc - 1 else 1
    col = _obv(bv).col(j)
    K.check(_lib.VecPlaceArray(b.cv[l], col.ctypes.data))


@_guard
def _b_restorecolumn(bv, j, v):
    b = bv.contents
    l = 0 if j == b.ci[0] else 1
    K.check(_lib.VecResetArray(b.cv[l]))


@_guard
def _b_destroy(bv):
    _bvs.pop(bv.contents.data, None)
    bv.contents.data = None


_B = dict(K.BVOps._fields_)
_BV_CBS = {
    "mult": _B["mult"](_b_mult), "multvec": _B["multvec"](_b_multvec),
    "multinplace": _B["multinplace"](_b_multinplace),
    "multinplacetrans": _B["multinplacetrans"](_b_multinplacetrans), "dot": _B["dot"](_b_dot),
    "dotvec": _B["dotvec"](_b_dotvec), "dotvec_local": _B["dotvec_local"](_b_dotvec_local),
    "scale": _B["scale"](_b_scale), "norm": _B["norm"](_b_norm), "norm_local": _B["norm_local"](_b_norm),
    "matmult": _B["matmult"](_b_matmult), "copy": _B["copy"](_b_copy), "resize": _B["resize"](_b_resize),
    "getcolumn": _B["getcolumn"](_b_getcolumn), "restorecolumn": _B["restorecolumn"](_b_restorecolumn),
    "destroy": _B["destroy"](_b_destroy),
}


@_guard
def _bv_create_oracle(bv):
    """The constructor a plug-in registers with BVRegister (model: svec.c:417-520)."""
    b = bv.contents
    t = b.t.contents
    ident = _next_id[0]
    _next_id[0] += 1
    _bvs[ident] = obv.BV(b.n, b.m, N=b.N, comm=COMM[0])
    b.data = ident
    for i in range(2):
        p = K.VecP()
        K.check(_lib.VecCreateWithOps(t.ops, None, t.n, t.N, t.row0, 0, C.byref(p)))
        b.cv[i] = p
    for name, cb in _BV_CBS.items():
        setattr(b.ops, name, cb)


_CREATE_CB = K.BVCREATE_FN(_bv_create_oracle)
K.check(_lib.BVRegister(b"oracle", _CREATE_CB))


# ---- shell operator --------------------------------------------------------------
_shell_ctx = {}
_shell_keep = []


def shell_mat(matvec, n, N=None, row0=0):
    """MatCreateShell with MATOP_MULT = matvec(x, y) on numpy views and
    MatCreateVecs yielding "numpy" vectors."""
    ident = _next_id[0]
    _next_id[0] += 1
    _shell_ctx[ident] = matvec

    @_guard
    def _mult(A, x, y):
        _shell_ctx[A.contents.data](_view(x), _view(y))

    @_guard
    def _vecs(A, right, left):
        a = A.contents
        for out, (n_, N_, r0_) in ((right, (a.cn, a.cN, a.col0)), (left, (a.n, a.N, a.row0))):
            if out:
                p = K.VecP()
                K.check(_lib.VecCreateWithOps(C.byref(VEC_OPS), None, n_, N_, r0_, 1, C.byref(p)))
                out[0] = p

    cb1, cb2 = K.MATMULT_FN(_mult), K.MATVECS_FN(_vecs)
    _shell_keep.extend([cb1, cb2])
    p = K.MatP()
    K.check(_lib.MatCreateShell(None, n, N if N is not None else n, row0, ident, cb1, cb2, C.byref(p)))
    return K.Mat(p, _lib)


def shell_mat_rect(matvec, matvec_t, M, N):
    """A rectangular M x N shell (one process) that can be transposed: MatTranspose
    returns the N x M shell of matvec_t."""
    A = shell_mat(matvec, M)
    K.check(_lib.MatShellSetColumnLayout(A.p, N, N, 0))

    @_guard
    def _transpose(Ap, out):
        T = shell_mat_rect(matvec_t, matvec, N, M)
        _shell_keep.append(T)
        out[0] = T.p
        T.p = None                      # owned by the caller of MatTranspose from here on

    cb = K.MATTRANSPOSE_FN(_transpose)
    _shell_keep.append(cb)
    K.check(_lib.MatShellSetTranspose(A.p, cb))
    return A


def bv_create(n, m, N=None, row0=0, bvtype=b"oracle"):
    """BVCreate + BVSetSizesFromVec + BVSetType on "numpy" vectors; returns (BVP, template Vec)."""
    t = vec_create(n, N, row0)
    bv = K.BVP()
    K.check(_lib.BVCreate(None, C.byref(bv)))
    K.check(_lib.BVSetSizesFromVec(bv, t.p, m))
    K.check(_lib.BVSetType(bv, bvtype))
    return bv, t


def bv_array(bv):
    """The oracle storage behind an "oracle" BV (n x (nc+m), column-major)."""
    return _bvs[bv.contents.data].array
