"""Shared test helpers: the reference's BV and EPS test programs restated against
the C host interface (libkseps), parametrised by back end:

* ``"oracle"`` -- the test-only numpy plug-in (tests/oracle_plugin.py), no GPU;
* ``"hip"``    -- the product path: BV type "hip" + device vectors (needs a GPU).
"""
import ctypes as C
import json
import os

import numpy as np

from slepc_amd import kseps as K

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS_MACH = np.finfo(float).eps


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


class Backend:
    def __init__(self, kind, ctx=None):
        self.kind = kind
        self.ctx = ctx
        self.lib = K.load()
        if kind == "oracle":
            from tests import oracle_plugin
            self.plug = oracle_plugin

    def vec(self, n, N=None, row0=0):
        if self.kind == "oracle":
            return self.plug.vec_create(n, N, row0)
        return K.Vec.create_hip(self.ctx, n, N, row0)

    def bv(self, n, m):
        t = self.vec(n)
        bv = K.BVP()
        K.check(self.lib.BVCreate(None, C.byref(bv)))
        K.check(self.lib.BVSetSizesFromVec(bv, t.p, m))
        K.check(self.lib.BVSetType(bv, b"oracle" if self.kind == "oracle" else b"hip"))
        t.destroy()
        return bv

    def mat_csr(self, rp, ci, va):
        if self.kind == "oracle":
            from scipy.sparse import csr_matrix
            A = csr_matrix((va, ci, rp), shape=(len(rp) - 1, len(rp) - 1))

            def mv(x, y):
                y[:] = A @ x
            return self.plug.shell_mat(mv, len(rp) - 1)
        return K.Mat.create_aij_hip(self.ctx, rp, ci, va)

    def mat_csr_rect(self, rp, ci, va, ncols):
        """M x ncols operator (one process)."""
        M = len(rp) - 1
        if self.kind == "oracle":
            from scipy.sparse import csr_matrix
            A = csr_matrix((va, ci, rp), shape=(M, ncols))
            AT = csr_matrix(A.T)

            def mv(x, y):
                y[:] = A @ x

            def mvt(x, y):
                y[:] = AT @ x
            return self.plug.shell_mat_rect(mv, mvt, M, ncols)
        return K.Mat.create_aij_hip_rect(self.ctx, rp, ci, va, ncols)

    # -- BV helpers over the C interface --
    def set_col(self, bv, j, arr):
        v = K.VecP()
        K.check(self.lib.BVGetColumn(bv, j, C.byref(v)))
        a = np.ascontiguousarray(arr, dtype=np.float64)
        K.check(self.lib.VecSetValuesHost(v, a.ctypes.data_as(K.c_dp)))
        K.check(self.lib.BVRestoreColumn(bv, j, C.byref(v)))

    def get_col(self, bv, j):
        v = K.VecP()
        K.check(self.lib.BVGetColumn(bv, j, C.byref(v)))
        out = np.empty(bv.contents.n)
        K.check(self.lib.VecGetValuesHost(v, out.ctypes.data_as(K.c_dp)))
        K.check(self.lib.BVRestoreColumn(bv, j, C.byref(v)))
        return out

    def to_numpy(self, bv):
        return np.stack([self.get_col(bv, j) for j in range(bv.contents.m)], axis=1)

    def destroy_bv(self, bv):
        K.check(self.lib.BVDestroy(C.byref(bv)))


def dp(a):
    return a.ctypes.data_as(K.c_dp)


def fill_test1(be, bv, n, k):
    """X[i+j, j] = 3i + j - 2, i < 4 (bv/examples/tests/test1.c:66-77)."""
    for j in range(k):
        col = np.zeros(n)
        for i in range(4):
            if i + j < n:
                col[i + j] = 3 * i + j - 2
        be.set_col(bv, j, col)


def run_bv_test1(be):
    """bv/examples/tests/test1.c against output/test1_1_svec.out."""
    g = golden("bv_test1.json")
    lib = be.lib
    n, k, l = 10, 5, 3
    X, Y = be.bv(n, k), be.bv(n, l)
    fill_test1(be, X, n, k)
    for j in range(l):
        be.set_col(Y, j, np.full(n, (j + 1) / 4.0))
    Q = np.asfortranarray(np.array([[2.0 if i < j else -0.5 for j in range(l)] for i in range(k)]))
    K.check(lib.BVMult(Y, 2.0, 1.0, X, dp(Q), k))
    assert np.array_equal(be.to_numpy(Y), np.array(g["Y_after_mult"]).reshape(l, n).T)
    z = np.array([2.0 * (-0.5) ** i for i in range(k)])
    v = K.VecP()
    K.check(lib.BVGetColumn(Y, 0, C.byref(v)))
    K.check(lib.BVMultVec(X, -1.0, 1.0, v, dp(z)))
    K.check(lib.BVRestoreColumn(Y, 0, C.byref(v)))
    assert np.array_equal(be.to_numpy(Y), np.array(g["Y_after_multvec"]).reshape(l, n).T)
    M = np.zeros((l, k), order="F")
    K.check(lib.BVDot(X, Y, dp(M), l))
    assert np.array_equal(M, np.array(g["M"]))
    zz = np.zeros(k)
    K.check(lib.BVGetColumn(Y, 0, C.byref(v)))
    K.check(lib.BVDotVec(X, v, dp(zz)))
    K.check(lib.BVRestoreColumn(Y, 0, C.byref(v)))
    assert np.array_equal(zz, np.array(g["z"]))
    K.check(lib.BVMultInPlace(X, dp(Q), k, 1, l))
    K.check(lib.BVScale(X, 2.0))
    Xh = be.to_numpy(X)
    assert np.array_equal(Xh, np.array(g["X_after_multinplace"]).reshape(k, n).T)
    nrm = C.c_double()
    K.check(lib.BVNormColumn(X, 0, K.NORM_2, C.byref(nrm)))
    assert "%g" % nrm.value == g["norm_X0"]
    K.check(lib.BVNorm(X, K.NORM_FROBENIUS, C.byref(nrm)))
    assert "%g" % nrm.value == g["norm_fro"]
    assert list(Xh[0, :]) == g["first_row"]
    be.destroy_bv(X)
    be.destroy_bv(Y)


def run_bv_test2(be, otype=K.BV_ORTHOG_CGS):
    """bv/examples/tests/test2.c (column orthogonalisation and BVOrthogonalizeVec)
    against output/test2_1.out."""
    g = golden("bv_misc.json")
    lib = be.lib
    n, k = 20, 8
    X = be.bv(n, k)
    K.check(lib.BVSetOrthogonalization(X, otype, K.BV_ORTHOG_REFINE_IFNEEDED, 0.0, K.BV_ORTHOG_BLOCK_GS))
    for j in range(k):
        col = np.zeros(n)
        for i in range(n // 2 + 1):
            if i + j < n:
                col[i + j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
        be.set_col(X, j, col)
    X0 = be.to_numpy(X)
    nrm = C.c_double()
    for j in range(k):
        K.check(lib.BVOrthogonalizeColumn(X, j, None, C.byref(nrm), None))
        K.check(lib.BVScaleColumn(X, j, 1.0 / nrm.value))
    M = np.zeros((k, k), order="F")
    K.check(lib.BVDot(X, X, dp(M), k))
    assert np.abs(M - np.eye(k)).sum(axis=0).max() < 100 * EPS_MACH          # "Level of orthogonality < 100*eps"
    assert g["test2_lines"][0] == "Level of orthogonality < 100*eps"
    # the basis spans the same space as the original columns: X0 = X (X^T X0)
    Xh = be.to_numpy(X)
    assert np.linalg.norm(X0 - Xh @ (Xh.T @ X0)) < 100 * EPS_MACH           # "Residual ||X-QR|| < 100*eps"
    e = be.vec(n)
    K.check(lib.VecSet(e.p, 1.0))
    K.check(lib.BVOrthogonalizeVec(X, e.p, None, C.byref(nrm), None))
    assert "%g" % nrm.value == g["test2_norm_ones"]
    e.destroy()
    be.destroy_bv(X)


def run_bv_test4(be, trans=False):
    """bv/examples/tests/test4.c -n 18 -kx 12 -ky 8 (lx=3, ly=2) against output/test4_1.out."""
    g = golden("bv_misc.json")
    lib = be.lib
    n, kx, lx, ky, ly = 18, 12, 3, 8, 2
    X = be.bv(n, kx + 2)
    K.check(lib.BVSetActiveColumns(X, lx, kx))
    fill_test1(be, X, n, kx + 2)
    Y = be.bv(n, ky + 1)
    K.check(lib.BVSetActiveColumns(Y, ly, ky))
    for j in range(ky + 1):
        be.set_col(Y, j, np.full(n, (j + 1) / 4.0))
    Q = np.asfortranarray(np.array([[2.0 if i < j else -0.5 for j in range(ky)] for i in range(kx)]))
    K.check(lib.BVResize(X, kx + 4, 1))
    K.check(lib.BVMult(Y, 2.0, 0.5, X, dp(Q), kx))
    z = np.array([2.0 * (-0.5) ** i for i in range(kx - lx)])
    v = K.VecP()
    K.check(lib.BVGetColumn(Y, 0, C.byref(v)))
    K.check(lib.BVMultVec(X, -1.0, 1.0, v, dp(z)))
    K.check(lib.BVRestoreColumn(Y, 0, C.byref(v)))
    M = np.zeros((ky, kx), order="F")
    K.check(lib.BVDot(X, Y, dp(M), ky))
    zz = np.zeros(kx - lx)
    K.check(lib.BVGetColumn(Y, 0, C.byref(v)))
    K.check(lib.BVDotVec(X, v, dp(zz)))
    K.check(lib.BVRestoreColumn(Y, 0, C.byref(v)))
    if trans:
        Qt = np.asfortranarray(Q.T)
        K.check(lib.BVMultInPlaceTranspose(X, dp(Qt), ky, lx + 1, ky))
    else:
        K.check(lib.BVMultInPlace(X, dp(Q), kx, lx + 1, ky))
    K.check(lib.BVScale(X, 2.0))
    nrm = C.c_double()
    K.check(lib.BVNormColumn(X, lx, K.NORM_2, C.byref(nrm)))
    assert "%g" % nrm.value == g["test4_norm_X3"]
    K.check(lib.BVNorm(X, K.NORM_FROBENIUS, C.byref(nrm)))
    assert "%g" % nrm.value == g["test4_norm_fro"]
    be.destroy_bv(X)
    be.destroy_bv(Y)


def run_split_phase(be, n=1000, m=7):
    """A group of split-phase reductions (BVDotVecBegin/End, BVDotColumnBegin/End,
    BVNormColumnBegin/End, BVNormVecBegin/End, VecDotBegin/End, VecNormBegin/End:
    bvglobal.c:190-310, 732-830) delivers what the immediate calls deliver, to every
    destination, whatever the order of the End calls; a second group starts clean."""
    lib = be.lib
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, m))
    X = be.bv(n, m)
    for j in range(m):
        be.set_col(X, j, A[:, j])
    y = be.vec(n)
    yh = rng.standard_normal(n)
    K.check(lib.VecSetValuesHost(y.p, dp(yh)))
    for rep in range(2):
        K.check(lib.BVSetActiveColumns(X, 1, m - 1 - rep))
        k = m - 2 - rep
        ref_dv, ref_dc = np.zeros(k), np.zeros(3)
        ref_nc, ref_nv, ref_d, ref_n = (C.c_double() for _ in range(4))
        K.check(lib.BVDotVec(X, y.p, dp(ref_dv)))
        K.check(lib.BVNormColumn(X, 2, K.NORM_2, C.byref(ref_nc)))
        K.check(lib.BVNormVec(X, y.p, K.NORM_2, C.byref(ref_nv)))
        K.check(lib.VecDot(y.p, y.p, C.byref(ref_d)))
        K.check(lib.VecNorm(y.p, K.NORM_2, C.byref(ref_n)))
        K.check(lib.BVDotColumn(X, 4, dp(ref_dc)))
        dv, dc = np.full(k, np.nan), np.full(3, np.nan)
        nc, nv, d, nn = (C.c_double(float("nan")) for _ in range(4))
        K.check(lib.BVDotVecBegin(X, y.p, dp(dv)))
        K.check(lib.BVNormColumnBegin(X, 2, K.NORM_2, C.byref(nc)))
        K.check(lib.BVNormVecBegin(X, y.p, K.NORM_2, C.byref(nv)))
        K.check(lib.VecDotBegin(y.p, y.p, C.byref(d)))
        K.check(lib.VecNormBegin(y.p, K.NORM_2, C.byref(nn)))
        K.check(lib.BVDotColumnBegin(X, 4, dp(dc)))
        assert lib.BVNormColumnEnd(X, 2, K.NORM_2, C.byref(nc)) != 0      # not the next in order
        if rep == 0:        # the failed End dropped the group: post it again
            K.check(lib.BVDotVecBegin(X, y.p, dp(dv)))
            K.check(lib.BVNormColumnBegin(X, 2, K.NORM_2, C.byref(nc)))
            K.check(lib.BVNormVecBegin(X, y.p, K.NORM_2, C.byref(nv)))
            K.check(lib.VecDotBegin(y.p, y.p, C.byref(d)))
            K.check(lib.VecNormBegin(y.p, K.NORM_2, C.byref(nn)))
            K.check(lib.BVDotColumnBegin(X, 4, dp(dc)))
        else:
            continue
        K.check(lib.BVDotVecEnd(X, y.p, dp(dv)))
        assert lib.BVDotVecBegin(X, y.p, dp(dv)) != 0                     # Begin after an End of an unfinished group
        K.check(lib.BVNormColumnEnd(X, 2, K.NORM_2, C.byref(nc)))
        K.check(lib.BVNormVecEnd(X, y.p, K.NORM_2, C.byref(nv)))
        K.check(lib.VecDotEnd(y.p, y.p, C.byref(d)))
        K.check(lib.VecNormEnd(y.p, K.NORM_2, C.byref(nn)))
        K.check(lib.BVDotColumnEnd(X, 4, dp(dc)))
        assert lib.VecDotEnd(y.p, y.p, C.byref(d)) != 0                   # more Ends than Begins
        assert np.array_equal(dv, ref_dv) and np.array_equal(dc, ref_dc)
        assert d.value == ref_d.value
        for got, ref in ((nc, ref_nc), (nv, ref_nv), (nn, ref_n)):
            assert abs(got.value - ref.value) <= 4 * np.finfo(float).eps * ref.value
        assert np.allclose(dv, A[:, 1:1 + k].T @ yh, rtol=1e-12, atol=1e-12)
        assert np.allclose(dc, A[:, 1:4].T @ A[:, 4], rtol=1e-12, atol=1e-12)
        assert np.isclose(nc.value, np.linalg.norm(A[:, 2]), rtol=1e-14)
    y.destroy()
    be.destroy_bv(X)


def run_bv_test13(be):
    """bv/examples/tests/test13.c (q == NULL: coefficients in the BV buffer)."""
    g = golden("bv_misc.json")
    lib = be.lib
    n, k = 10, 5
    X = be.bv(n, k)
    fill_test1(be, X, n, k)
    K.check(lib.BVDotColumn(X, 2, None))
    K.check(lib.BVMultColumn(X, -1.0, 1.0, 2, None))
    nrm = C.c_double()
    K.check(lib.BVNorm(X, K.NORM_FROBENIUS, C.byref(nrm)))
    assert "%g" % nrm.value == g["test13_norm_fro"]
    be.destroy_bv(X)


def solve_eps(be, csr, hermitian=True, nev=1, ncv=0, mpd=0, tol=0.0, max_it=0, which=None, v0=None,
              keep=None, lock=True, want_vectors=False, solver=None, delayed=False, orthog_refine=None, reorthog=None):
    """EPSCreate / EPSSetOperators / ... / EPSSolve / EPSGetEigenpair / EPSComputeError."""
    rp, ci, va = csr
    A = be.mat_csr(rp, ci, va)
    eps = K.EPS(be.ctx.h if be.ctx else None)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_HEP if hermitian else K.EPS_NHEP)
    if solver:
        eps.set_type(solver)
    if reorthog:
        eps.set_lanczos_reorthog(reorthog)
    if delayed:
        eps.set_arnoldi_delayed(True)
    if orthog_refine is not None:
        K.check(eps.lib.BVSetOrthogonalization(eps.bv(), K.BV_ORTHOG_CGS, orthog_refine, 0.0, K.BV_ORTHOG_BLOCK_GS))
    eps.set_dimensions(nev, ncv, mpd)
    eps.set_tolerances(tol, max_it)
    if which:
        eps.set_which(which)
    if keep:
        eps.set_restart(keep)
    eps.set_locking(lock)
    if be.kind == "oracle":
        eps.set_bv_type("oracle")
    vs = None
    if v0 is not None:
        vs = A.create_vecs()
        vs.set_values(v0)
        eps.set_initial_space([vs])
    eps.solve()
    res = {"its": eps.its, "nconv": eps.nconv, "reason": eps.reason, "values": [], "errors": [],
           "vectors": [], "dims": eps.dimensions()}
    xr, xi = A.create_vecs(), A.create_vecs()
    for i in range(eps.nconv):
        re, im = eps.get_eigenpair(i, xr, xi)
        res["values"].append((re, im))
        if i < max(nev, 1) + 2:
            res["errors"].append(eps.compute_error(i))
            if want_vectors:
                res["vectors"].append((xr.get_values(), xi.get_values()))
    for o in (xr, xi):
        o.destroy()
    if vs:
        vs.destroy()
    eps.destroy()
    A.destroy()
    return res


def run_bv_test11(be, block, n=20, l=2, k=8):
    """src/sys/classes/bv/examples/tests/test11.c with the standard inner product (golden
    output test11_1.out: "Level of orthogonality < 100*eps" for -bv_orthog_block gs and chol),
    plus the factor R against the oracle when no leading columns are kept."""
    from oracle import bv as obv
    lib = be.lib
    X0 = np.zeros((n, k))
    if n == 20:
        for j in range(k):
            for i in range(n // 2 + 1):
                if i + j < n:
                    X0[i + j, j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
        tol = 100 * np.finfo(float).eps
    else:                                   # a well-conditioned tall block for the kernels at full width
        X0 = np.random.default_rng(n).standard_normal((n, k))
        tol = 1e-12
    for (ll, want_R) in ((l, False), (0, True)):
        Y = be.bv(n, k)
        K.check(lib.BVSetOrthogonalization(Y, K.BV_ORTHOG_CGS, K.BV_ORTHOG_REFINE_IFNEEDED, 0.0, block))
        for j in range(k):
            be.set_col(Y, j, X0[:, j])
        O = obv.BV(n, k)
        O.array[:, :] = X0
        if block == K.BV_ORTHOG_BLOCK_GS:          # GS requires the leading columns to be orthogonal
            for j in range(ll):
                K.check(lib.BVOrthonormalizeColumn(Y, j, 0, None, None))
                O.orthonormalize_column(j, False)
        K.check(lib.BVSetActiveColumns(Y, ll, k))
        O.set_active_columns(ll, k)
        R = np.zeros((k, k), order="F")
        K.check(lib.BVOrthogonalize(Y, dp(R) if want_R else None, k))
        Ro = obv.orthogonalize_block(O, "gs" if block == K.BV_ORTHOG_BLOCK_GS else "chol", want_R=True)
        Q = be.to_numpy(Y)
        M = np.eye(k)
        M[ll:, ll:] = Q[:, ll:].T @ Q[:, ll:]
        assert np.abs(M - np.eye(k)).sum(axis=0).max() < tol      # MatNorm(M,NORM_1)
        assert np.abs(Q - O.array).max() < 1e-11
        if want_R:
            assert np.abs(np.tril(R, -1)).max() == 0.0
            assert np.abs(Q @ R - X0).max() < 1e-12 and np.abs(R - Ro).max() < 1e-10
        be.destroy_bv(Y)


def solve_svd(be, csr, ncols, nsv=1, ncv=0, mpd=0, tol=0.0, max_it=0, which="largest", solver=None, oneside=False,
              conv=None, v0=None, want_vectors=False):
    """SVDCreate / SVDSetOperator / ... / SVDSolve / SVDGetSingularTriplet / SVDComputeError."""
    rp, ci, va = csr
    A = be.mat_csr_rect(rp, ci, va, ncols)
    svd = K.SVD(be.ctx.h if be.ctx else None)
    svd.set_operator(A)
    if solver:
        svd.set_type(solver)
    svd.set_dimensions(nsv, ncv, mpd)
    svd.set_tolerances(tol, max_it)
    svd.set_which(which)
    if conv:
        svd.set_convergence_test(conv)
    if oneside:
        svd.set_oneside(True)
    if be.kind == "oracle":
        svd.set_bv_type("oracle")
    vs = None
    if v0 is not None:
        vs = A.create_vecs()
        vs.set_values(v0)
        svd.set_initial_space([vs])
    svd.solve()
    res = {"its": svd.its, "nconv": svd.nconv, "reason": svd.reason, "sigma": [], "errors": [], "vectors": [],
           "type": svd.get_type(), "dims": svd.dimensions(), "tols": svd.tolerances()}
    u, v = A.create_vecs_left(), A.create_vecs()
    for i in range(svd.nconv):
        res["sigma"].append(svd.get_singular_triplet(i, u, v))
        if i < nsv + 2:
            res["errors"].append(svd.compute_error(i))
            if want_vectors:
                res["vectors"].append((u.get_values(), v.get_values()))
    for o in (u, v):
        o.destroy()
    if vs:
        vs.destroy()
    svd.destroy()
    A.destroy()
    return res
