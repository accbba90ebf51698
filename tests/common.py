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
              keep=None, lock=True, want_vectors=False, solver=None, delayed=False, orthog_refine=None, reorthog=None,
              csr_b=None, problem=None, st_type=None, sigma=None, target=None, ksp=None, deflation=None, purify=None,
              ksp_restart=0):
    """EPSCreate / EPSSetOperators / ... / EPSSolve / EPSGetEigenpair / EPSComputeError.
    csr_b: the matrix B of a generalized problem; problem: EPS_GHEP, ...; st_type, sigma, ksp:
    the spectral transformation and its linear solver; deflation: vectors of EPSSetDeflationSpace."""
    rp, ci, va = csr
    A = be.mat_csr(rp, ci, va)
    B = be.mat_csr(*csr_b) if csr_b is not None else None
    eps = K.EPS(be.ctx.h if be.ctx else None)
    eps.set_operators(A, B)
    if problem is not None:
        eps.set_problem_type(problem)
    elif B is None:
        eps.set_problem_type(K.EPS_HEP if hermitian else K.EPS_NHEP)
    st = eps.st()
    if st_type:
        st.set_type(st_type)
    if sigma is not None:
        st.set_shift(sigma)
    if target is not None:
        eps.set_target(target)
    if ksp:
        st.set_ksp_type(ksp)
    if ksp_restart:
        st.set_ksp_tolerances(0.0, 0, ksp_restart)
    if purify is not None:
        eps.set_purify(purify)
    dvecs = []
    for d in (deflation or []):
        v = A.create_vecs()
        v.set_values(d)
        dvecs.append(v)
    if dvecs:
        eps.set_deflation_space(dvecs)
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
           "vectors": [], "dims": eps.dimensions(), "ksp": st.get_ksp_type(), "linear": st.ksp_iterations()}
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
    for v in dvecs:
        v.destroy()
    eps.destroy()
    A.destroy()
    if B is not None:
        B.destroy()
    return res


def golden_matrix(name):
    """One of the reference's data files kept as a fixture (share/slepc/datafiles/matrices)."""
    _, _, rp, ci, va = K.read_petsc_binary(os.path.join(GOLDEN, "matrices", name + ".petsc"))
    return rp, ci, va


def diag_csr(d):
    n = len(d)
    return np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32), np.asarray(d, dtype=np.float64)


def check_values(got, want, rtol=1e-9):
    """Eigenvalues (re, im) against printed golden values "re", "re+imi" (5 decimals)."""
    assert len(got) >= len(want)
    for (re, im), w in zip(got, want):
        m = __import__("re").match(r"^(-?[0-9.]+)(?:([+-][0-9.]+)i)?$", w)
        wr, wi = float(m.group(1)), float(m.group(2) or 0.0)
        scale = max(abs(complex(wr, wi)), 1e-300)
        assert abs(re - wr) <= 0.6e-5 + rtol * scale and abs(im - wi) <= 0.6e-5 + rtol * scale, (re, im, w)


def run_eps_ex13(be, ksp=None, nulldim=0):
    """src/eps/examples/tutorials/ex13.c (generalized symmetric problem, shift-and-invert at
    target 0, -eps_nev 4 -eps_ncv 22) against output/ex13_1.out."""
    from slepc_amd import matgen as mg
    g = golden("eps_tests.json")["ex13"]
    n = 10
    N = n * n
    d = np.full(N, 4.0)
    d[:nulldim] = 0.0
    res = solve_eps(be, mg.laplacian2d(n, n), nev=4, ncv=22, csr_b=diag_csr(d), problem=K.EPS_GHEP,
                    st_type="sinvert", target=0.0, which="target_magnitude", ksp=ksp, want_vectors=True)
    assert res["reason"] == 1 and res["nconv"] >= 4
    if nulldim == 0:
        check_values(res["values"][:4], g)
    assert max(res["errors"][:4]) < 1e-7
    return res


def run_eps_ex7(be, ksp=None):
    """src/eps/examples/tutorials/ex7.c on the reference's files bfw62a/bfw62b (-eps_nev 4; a
    generalized non-symmetric problem with the default transformation, B^-1 A) against
    output/ex7_1.out."""
    g = golden("eps_tests.json")["ex7"]
    res = solve_eps(be, golden_matrix("bfw62a"), nev=4, csr_b=golden_matrix("bfw62b"), ksp=ksp)
    assert res["reason"] == 1 and res["nconv"] >= 4
    check_values(res["values"][:4], g)
    assert max(res["errors"][:4]) < 1e-7
    return res


def petsc_g(x):
    """%g as PetscPrintf prints it: a value without fractional part gets a trailing dot."""
    t = "%g" % x
    return t if ("." in t or "e" in t or "n" in t) else t + "."


def tridiag_csr(n):
    """B = tridiag(-1, 2, -1) (bv tests test3.c:51-56, test11.c:119-124)."""
    rp, ci, va = [0], [], []
    for i in range(n):
        for j, v in ((i - 1, -1.0), (i, 2.0), (i + 1, -1.0)):
            if 0 <= j < n:
                ci.append(j)
                va.append(v)
        rp.append(len(ci))
    return np.array(rp, dtype=np.int64), np.array(ci, dtype=np.int32), np.array(va)


def _level_of_orthogonality(M, target):
    nrm = np.abs(M - target).sum(axis=0).max()          # MatNorm(M,NORM_1)
    return "Level of orthogonality < 100*eps" if nrm < 100 * np.finfo(float).eps else "Level of orthogonality: %g" % nrm


def run_bv_test3(be, otype=K.BV_ORTHOG_CGS):
    """bv/examples/tests/test3.c (non-standard inner product) against output/test3_1.out."""
    g = golden("bv_misc.json")["test3_lines"]
    lib = be.lib
    n, k = 10, 5
    B = be.mat_csr(*tridiag_csr(n))
    X = be.bv(n, k)
    K.check(lib.BVSetOrthogonalization(X, otype, K.BV_ORTHOG_REFINE_IFNEEDED, 0.0, K.BV_ORTHOG_BLOCK_GS))
    K.check(lib.BVSetMatrix(X, B.p, 0))
    fill_test1(be, X, n, k)
    out = []
    nrm = C.c_double()
    K.check(lib.BVNormColumn(X, 0, K.NORM_2, C.byref(nrm)))
    out.append("B-Norm of X[0] = " + petsc_g(nrm.value))
    for j in range(k):
        K.check(lib.BVOrthogonalizeColumn(X, j, None, C.byref(nrm), None))
        K.check(lib.BVScaleColumn(X, j, 1.0 / nrm.value))
    M = np.zeros((k, k), order="F")
    K.check(lib.BVDot(X, X, dp(M), k))
    out.append(_level_of_orthogonality(M, np.eye(k)))
    v = K.VecP()
    K.check(lib.BVGetColumn(X, 0, C.byref(v)))
    K.check(lib.BVNormVecBegin(X, v, K.NORM_1, C.byref(nrm)))
    K.check(lib.BVNormVecEnd(X, v, K.NORM_1, C.byref(nrm)))
    K.check(lib.BVRestoreColumn(X, 0, C.byref(v)))
    out.append("B-Norm of X[0] = " + petsc_g(float("%.12g" % nrm.value)))
    assert out == g, out
    # the product really is the B-inner one
    Q = be.to_numpy(X)
    Bd = 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    assert np.abs(Q.T @ Bd @ Q - np.eye(k)).max() < 1e-13
    be.destroy_bv(X)
    B.destroy()


def run_bv_test5(be):
    """bv/examples/tests/test5.c (indefinite inner product, -bv_orthog_refine always) against
    output/test5_1.out; the checks the program makes itself (BVApplyMatrix twice = identity for
    the involutory permutation, BVApplyMatrixBV) are made too."""
    g = golden("bv_misc.json")["test5_lines"]
    lib = be.lib
    n, k = 10, 5
    rp = np.arange(n + 1, dtype=np.int64)
    ci = np.arange(n - 1, -1, -1, dtype=np.int32)
    B = be.mat_csr(rp, ci, np.ones(n))
    X = be.bv(n, k)
    K.check(lib.BVSetOrthogonalization(X, K.BV_ORTHOG_CGS, K.BV_ORTHOG_REFINE_ALWAYS, 0.0, K.BV_ORTHOG_BLOCK_GS))
    K.check(lib.BVSetMatrix(X, B.p, 1))
    for j in range(k):
        col = -np.ones(n)
        for i in range(4):
            if i + j < n:
                col[i + j] = 3 * i + j - 2
        be.set_col(X, j, col)
    X0 = be.to_numpy(X)
    out = []
    nrm = C.c_double()
    K.check(lib.BVNormColumn(X, 0, K.NORM_2, C.byref(nrm)))
    out.append("B-Norm of X[0] = " + petsc_g(nrm.value))
    for j in range(k):
        K.check(lib.BVOrthogonalizeColumn(X, j, None, C.byref(nrm), None))
        K.check(lib.BVScaleColumn(X, j, 1.0 / nrm.value))
    Y = K.BVP()
    K.check(lib.BVDuplicate(X, C.byref(Y)))
    K.check(lib.BVCopy(X, Y))
    M = np.zeros((k, k), order="F")
    K.check(lib.BVDot(Y, Y, dp(M), k))
    omega = np.zeros(k)
    K.check(lib.BVGetSignature(Y, dp(omega)))
    assert set(omega) <= {-1.0, 1.0} and (omega < 0).any()
    out.append(_level_of_orthogonality(M, np.diag(omega)))
    assert out == g, out
    K.check(lib.BVSetSignature(Y, dp(omega)))
    om2 = np.zeros(k)
    K.check(lib.BVGetSignature(Y, dp(om2)))
    assert np.array_equal(om2, omega)
    # BVApplyMatrix twice gives the vector back; BVApplyMatrixBV gives B*X
    t, w = be.vec(n), be.vec(n)
    v = K.VecP()
    K.check(lib.BVGetColumn(X, 0, C.byref(v)))
    K.check(lib.BVApplyMatrix(X, v, w.p))
    K.check(lib.BVApplyMatrix(X, w.p, t.p))
    K.check(lib.VecAXPY(t.p, -1.0, v))
    K.check(lib.BVRestoreColumn(X, 0, C.byref(v)))
    K.check(lib.VecNorm(t.p, K.NORM_2, C.byref(nrm)))
    assert abs(nrm.value) <= 10 * np.finfo(float).eps
    K.check(lib.BVApplyMatrixBV(X, Y))
    Q = be.to_numpy(X)
    assert np.array_equal(be.to_numpy(Y), Q[::-1, :])
    # against the CPU restatement, column for column
    from oracle import bv as obv
    O = obv.BV(n, k)
    O.array[:, :] = X0
    O.orthog_ref = obv.REFINE_ALWAYS
    O.set_matrix(lambda x: x[::-1].copy(), True)
    for j in range(k):
        onrm, _ = O.orthogonalize_column(j)
        O.scale_column(j, 1.0 / onrm)
    assert np.abs(Q - O.array).max() < 1e-13 and np.array_equal(O.get_signature(), omega)
    t.destroy()
    w.destroy()
    be.destroy_bv(Y)
    be.destroy_bv(X)
    B.destroy()


def run_bv_test6(be, otype=K.BV_ORTHOG_CGS, n=20, k=8, nc=2, dependent=False):
    """bv/examples/tests/test6.c (orthogonalisation with constraints) against
    output/test6_1.out; with ``dependent`` a repeated constraint must be dropped."""
    g = golden("bv_misc.json")["test6_lines"]
    lib = be.lib
    X = be.bv(n, k)
    K.check(lib.BVSetOrthogonalization(X, otype, K.BV_ORTHOG_REFINE_IFNEEDED, 0.0, K.BV_ORTHOG_BLOCK_GS))
    cvecs, Cm = [], []
    for j in range(nc):
        c = np.zeros(n)
        c[:(j if (dependent and j == nc - 1) else j + 1)] = 1.0
        Cm.append(c)
        v = be.vec(n)
        K.check(lib.VecSetValuesHost(v.p, dp(c)))
        cvecs.append(v)
    arr = (K.VecP * nc)(*[v.p for v in cvecs])
    ncio = C.c_int(nc)
    K.check(lib.BVInsertConstraints(X, C.byref(ncio), arr))
    assert ncio.value == (nc - 1 if dependent else nc)
    got = C.c_int()
    K.check(lib.BVGetNumConstraints(X, C.byref(got)))
    assert got.value == ncio.value
    assert lib.BVInsertConstraints(X, C.byref(ncio), arr) != 0          # only once
    for v in cvecs:
        v.destroy()
    X0 = np.zeros((n, k))
    for j in range(k):
        for i in range(n // 2 + 1):
            if i + j < n:
                X0[i + j, j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
        be.set_col(X, j, X0[:, j])
    nrm = C.c_double()
    for j in range(k):
        K.check(lib.BVOrthogonalizeColumn(X, j, None, C.byref(nrm), None))
        K.check(lib.BVScaleColumn(X, j, 1.0 / nrm.value))
    M = np.zeros((k, k), order="F")
    K.check(lib.BVDot(X, X, dp(M), k))
    assert [_level_of_orthogonality(M, np.eye(k))] == g
    Q = be.to_numpy(X)
    # the constraints are deflated, and reachable as columns -nc..-1
    cons = np.stack([be.get_col(X, -1 - i) for i in range(ncio.value)], axis=1)
    assert np.abs(cons.T @ Q).max() < 1e-14 and np.abs(cons.T @ cons - np.eye(ncio.value)).max() < 1e-14
    assert np.abs(np.array(Cm).dot(Q)).max() < 1e-13
    from oracle import bv as obv
    O = obv.BV(n, k)
    O.orthog_type = obv.MGS if otype == K.BV_ORTHOG_MGS else obv.CGS
    assert O.insert_constraints(Cm) == ncio.value
    O.array[:, O.nc:O.nc + k] = X0
    for j in range(k):
        onrm, _ = O.orthogonalize_column(j)
        O.scale_column(j, 1.0 / onrm)
    assert np.abs(Q - O.array[:, O.nc:O.nc + k]).max() < 1e-13
    be.destroy_bv(X)


def run_bv_test11_matrix(be, block, n=20, l=2, k=8):
    """The second half of bv/examples/tests/test11.c: block orthogonalisation in the inner
    product of B = tridiag(-1,2,-1), the cached B*X, against the last two lines of
    output/test11_1.out."""
    g = golden("bv_misc.json")["test11_lines"][1:]
    lib = be.lib
    B = be.mat_csr(*tridiag_csr(n))
    X, Y = be.bv(n, k), be.bv(n, k)
    for bv in (X, Y):
        K.check(lib.BVSetOrthogonalization(bv, K.BV_ORTHOG_CGS, K.BV_ORTHOG_REFINE_IFNEEDED, 0.0, block))
        for j in range(k):
            col = np.zeros(n)
            for i in range(n // 2 + 1):
                if i + j < n:
                    col[i + j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
            be.set_col(bv, j, col)
        K.check(lib.BVSetActiveColumns(bv, l, k))
    Z = K.BVP()
    K.check(lib.BVDuplicate(X, C.byref(Z)))
    K.check(lib.BVSetActiveColumns(Z, l, k))
    K.check(lib.BVSetMatrix(X, B.p, 0))
    K.check(lib.BVSetMatrix(Y, B.p, 0))
    if block == K.BV_ORTHOG_BLOCK_GS:
        for j in range(l):
            K.check(lib.BVOrthonormalizeColumn(Y, j, 0, None, None))
    K.check(lib.BVOrthogonalize(Y, None, 0))
    cached = K.BVP()
    K.check(lib.BVGetCachedBV(Y, C.byref(cached)))
    K.check(lib.BVMatMult(X, B.p, Z))
    K.check(lib.BVMult(Z, -1.0, 1.0, cached, None, 0))
    nrm = C.c_double()
    K.check(lib.BVNorm(Z, K.NORM_FROBENIUS, C.byref(nrm)))
    out = ["Residual ||cached-BX|| < 100*eps" if nrm.value < 100 * np.finfo(float).eps else "Residual ||cached-BX||: %g" % nrm.value]
    M = np.asfortranarray(np.eye(k))
    K.check(lib.BVDot(Y, Y, dp(M), k))
    out.append(_level_of_orthogonality(M, np.eye(k)))
    assert out == g, out
    assert lib.BVNorm(Y, K.NORM_FROBENIUS, C.byref(nrm)) != 0        # no matrix norm in a B-inner product
    be.destroy_bv(Z)
    be.destroy_bv(X)
    be.destroy_bv(Y)
    B.destroy()


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


def run_bv_test11_tsqr(be, n=20, l=2, k=8):
    """src/sys/classes/bv/examples/tests/test11.c with -bv_orthog_block tsqr (golden output
    test11_3_tsqr.out: "Level of orthogonality < 100*eps"; the inner-product half is skipped
    there, test11.c:129).  The factorisation covers the columns from the first one whatever the
    leading index (bvorthog.c:778); Q and R are also compared with numpy's Householder QR."""
    lib = be.lib
    X0 = np.zeros((n, k))
    if n == 20:
        for j in range(k):
            for i in range(n // 2 + 1):
                if i + j < n:
                    X0[i + j, j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
        tol = 100 * np.finfo(float).eps
    else:
        X0 = np.random.default_rng(n).standard_normal((n, k))
        tol = 1e-12
    Qo, Ro = np.linalg.qr(X0)
    s = np.where(np.diag(Ro) < 0, -1.0, 1.0)
    Qo, Ro = Qo * s, (Ro.T * s).T
    for want_R in (False, True):
        Y = be.bv(n, k)
        K.check(lib.BVSetOrthogonalization(Y, K.BV_ORTHOG_CGS, K.BV_ORTHOG_REFINE_IFNEEDED, 0.0, K.BV_ORTHOG_BLOCK_TSQR))
        for j in range(k):
            be.set_col(Y, j, X0[:, j])
        K.check(lib.BVSetActiveColumns(Y, l, k))
        R = np.full((k, k), 7.0, order="F")
        K.check(lib.BVOrthogonalize(Y, dp(R) if want_R else None, k))
        Q = be.to_numpy(Y)
        M = np.eye(k)
        M[l:, l:] = Q[:, l:].T @ Q[:, l:]
        assert np.abs(M - np.eye(k)).sum(axis=0).max() < tol      # MatNorm(M,NORM_1)
        assert np.abs(Q - Qo).max() < 1e-11
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
