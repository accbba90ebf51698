"""Singular value solvers on the kernels of the eigensolver: "cross" (the eigenproblem of
A^T A) and "trlanczos" (thick-restart Lanczos bidiagonalisation, two-sided and one-sided).

* the oracle (oracle/svd.py) against the reference's golden outputs:
  src/svd/examples/tutorials/output/{ex8_1,ex14_1,ex15_1}.out and
  src/svd/examples/tests/output/{test1_1,test3_1,test4_1}.out;
* the C host driver (slepc_amd/csrc/host/svd.c) against the oracle driver, both on the numpy
  BV/Vec/Mat plug-in (same decisions, values to rounding) -- no GPU needed;
* on the GPU: the product path (BV "hip", rectangular CSR operator and its device-built
  transpose) against the oracle and the golden values.
"""
import os

import numpy as np
import pytest
from scipy.sparse import csr_matrix, lil_matrix

from oracle import petscbin, svd as osvd
from slepc_amd import kseps as K
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATDIR = os.path.join(ROOT, "tests", "golden", "matrices")

# golden values, as printed by the reference (tests/golden/make_golden.py: svd_tests)
G = common.golden("svd_tests.json")
EX8 = (G["ex8"]["sigma_1"], G["ex8"]["sigma_n"], G["ex8"]["cond"])       # ex8_1.out = test1_1.out
EX14 = G["ex14"]["values"]                                                # ex14_1.out (cross, 3 iterations)
TEST3 = G["test3"]["values"]                                              # test3_1.out
TEST4 = G["test4"]["values"][0]                                           # test4_1.out


def grcar(N, M):
    """ex8.c:62-69 (square) / test3.c:51-58 (N x M): -1 on the subdiagonal, 1 on the diagonal
    and the three superdiagonals, rows cut at the last column."""
    A = lil_matrix((N, M))
    for i in range(N):
        if i == 0:
            cols, vals = range(0, min(4, M - i)), (1, 1, 1, 1)
        else:
            cnt = min(5, M - i + 1)
            cols, vals = range(i - 1, i - 1 + max(cnt, 0)), (-1, 1, 1, 1, 1)
        for c, v in zip(cols, vals):
            A[i, c] = v
    return csr_matrix(A)


def lauchli(n, mu):
    """ex15.c:60-72: (n+1) x n, first row ones, mu below."""
    A = lil_matrix((n + 1, n))
    A[0, :] = 1.0
    for i in range(1, n + 1):
        A[i, i - 1] = mu
    return csr_matrix(A)


def bidiag(m, n):
    """test4.c:60-67: m x n, 1 on the diagonal, 2 above it."""
    A = lil_matrix((m, n))
    for i in range(m):
        if i < n - 1:
            A[i, i], A[i, i + 1] = 1.0, 2.0
        elif i == n - 1:
            A[i, i] = 1.0
    return csr_matrix(A)


def rdb200():
    M, N, rp, ci, va = petscbin.read(os.path.join(MATDIR, "rdb200.petsc"))
    return csr_matrix((va, ci, rp), shape=(M, N))


def _csr(A):
    A = csr_matrix(A)
    A.sort_indices()
    return (A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)), A.shape[1]


def _oracle(A, **kw):
    A = csr_matrix(A)
    AT = csr_matrix(A.T)

    def mu(x, y):
        y[:] = A @ x

    def mt(x, y):
        y[:] = AT @ x
    return osvd.SVD(mu, mt, A.shape[0], A.shape[1], **kw).solve()


# ---- the oracle against the golden files -------------------------------------------------
@pytest.mark.parametrize("solver,oneside", [("cross", False), ("trlanczos", False), ("trlanczos", True),
                                            ("lanczos", False), ("lanczos", True), ("cyclic", False)])
def test_oracle_goldens(solver, oneside):
    A = grcar(30, 30)
    kw = dict(nsv=1, solver=solver, oneside=oneside, tol=1e-6, max_it=1000)      # test1.c:68-69
    s1 = _oracle(A, **kw).get_singular_triplet(0)[0]
    sn = _oracle(A, which="smallest", **kw).get_singular_triplet(0)[0]
    assert ("%.4f" % s1, "%.4f" % sn, "%.4f" % (s1 / sn)) == EX8
    s = _oracle(grcar(35, 30), nsv=4, solver=solver, oneside=oneside, v0=np.ones(30))
    assert ["%.5f" % s.get_singular_triplet(i)[0] for i in range(4)] == TEST3
    assert all(s.compute_error(i) < 5e-8 for i in range(4))
    s = _oracle(bidiag(20, 22), nsv=1, solver=solver, oneside=oneside, conv="abs")   # a wide operator
    sg, u, v = s.get_singular_triplet(0)
    assert "%.5f" % sg == TEST4 and u.shape == (20,) and v.shape == (22,)


def test_oracle_goldens_file_and_lauchli():
    s = _oracle(rdb200(), nsv=4)                                   # runex14_1: default type cross
    assert ["%.5f" % s.get_singular_triplet(i)[0] for i in range(4)] == EX14 and s.its == G["ex14"]["its"]
    g = G["ex15"]
    s = _oracle(lauchli(g["n"], g["mu"]), nsv=1, solver="trlanczos")             # ex15_1.out
    assert s.its == g["its"] and s.nconv == g["nconv"]
    assert ["%.6f" % s.get_singular_triplet(i)[0] for i in range(s.nconv)] == g["sigma"]


# ---- C driver against the oracle driver ------------------------------------------------------
CASES = {
    "grcar_cross": (lambda: grcar(40, 40), dict(nsv=3, solver="cross")),
    "grcar_rect_trl": (lambda: grcar(35, 30), dict(nsv=4, solver="trlanczos")),
    "grcar_rect_trl_oneside": (lambda: grcar(35, 30), dict(nsv=4, solver="trlanczos", oneside=True)),
    "grcar_trl_smallest": (lambda: grcar(30, 30), dict(nsv=1, solver="trlanczos", which="smallest", tol=1e-6, max_it=1000)),
    "bidiag_wide_trl_abs": (lambda: bidiag(20, 22), dict(nsv=2, solver="trlanczos", conv="abs")),
    "bidiag_wide_cross": (lambda: bidiag(20, 22), dict(nsv=2, solver="cross")),
    "grcar_rect_lanczos": (lambda: grcar(35, 30), dict(nsv=4, solver="lanczos")),
    "grcar_rect_lanczos_oneside": (lambda: grcar(35, 30), dict(nsv=4, solver="lanczos", oneside=True)),
    "bidiag_wide_lanczos": (lambda: bidiag(20, 22), dict(nsv=2, solver="lanczos", conv="abs")),
    "grcar_rect_cyclic": (lambda: grcar(35, 30), dict(nsv=4, solver="cyclic")),
    "grcar_cyclic_smallest": (lambda: grcar(30, 30), dict(nsv=1, solver="cyclic", which="smallest", tol=1e-6, max_it=1000)),
    "bidiag_wide_cyclic": (lambda: bidiag(20, 22), dict(nsv=2, solver="cyclic")),
    "rdb200_cross": (rdb200, dict(nsv=4, solver="cross")),
    "rdb200_trl": (rdb200, dict(nsv=4, solver="trlanczos")),
}


def _check_against_oracle(be, name):
    gen, kw = CASES[name]
    A = gen()
    M, N = A.shape
    v0 = None
    if M >= N:
        v0 = np.cos(np.arange(1, N + 1, dtype=float))
    csr, ncols = _csr(A)
    r = common.solve_svd(be, csr, ncols, v0=v0, want_vectors=True, **kw)
    okw = dict(kw)
    if "tol" not in okw:
        okw["tol"] = None
    o = _oracle(A, v0=v0, **okw)
    assert r["type"] == kw["solver"]
    assert (r["its"], r["nconv"]) == (o.its, o.nconv) and r["reason"] == K.SVD_CONVERGED_TOL
    assert r["dims"] == (o.nsv, o.ncv, o.mpd) and r["tols"] == (o.tol, o.max_it)
    for i in range(kw["nsv"]):
        sg, u, v = o.get_singular_triplet(i)
        assert abs(r["sigma"][i] - sg) <= 1e-10 * sg
        assert r["errors"][i] < 1e-6
        gu, gv = r["vectors"][i]
        s = np.sign(gv @ v)
        assert np.abs(s * gv - v).max() < 1e-6 and np.abs(s * gu - u).max() < 1e-6
        # the two halves of an eigenvector of [0 A; A^T 0] have norm 1/sqrt(2) only as far as it has converged
        ntol = 1e-7 if kw["solver"] == "cyclic" else 1e-12
        assert abs(np.linalg.norm(gu) - 1) < ntol and abs(np.linalg.norm(gv) - 1) < ntol


@pytest.fixture(scope="module")
def be_cpu():
    return common.Backend("oracle")


@pytest.mark.parametrize("name", list(CASES))
def test_host_driver_matches_oracle(be_cpu, name):
    _check_against_oracle(be_cpu, name)


def test_ds_svd_against_numpy():
    """DSSVD in compact storage: arrow + upper bidiagonal (state raw) and plain bidiagonal
    (state intermediate), locked leading part, sorted by the criterion."""
    lib = K.load()
    import ctypes as C
    rng = np.random.default_rng(3)
    for n, l, k, state in ((9, 0, 0, K.DS_STATE_INTERMEDIATE), (12, 2, 6, K.DS_STATE_RAW), (7, 0, 3, K.DS_STATE_RAW)):
        ds = K.vp()
        K.check(lib.DSCreate(C.byref(ds)))
        K.check(lib.DSSetType(ds, b"svd"))
        K.check(lib.DSSetCompact(ds, 1))
        K.check(lib.DSAllocate(ds, n + 1))
        cmpf = C.cast(lib.SlepcCompareLargestReal, K.COMPARE_FN)
        K.check(lib.DSSetSlepcSC(ds, cmpf, None))
        K.check(lib.DSSetDimensions(ds, n, n, l, k))
        T = K.c_dp()
        K.check(lib.DSGetArrayReal(ds, K.DS_MAT_T, C.byref(T)))
        ld = n + 1
        d = rng.uniform(0.5, 2.0, n)
        e = rng.uniform(-1.0, 1.0, n)
        e[:l] = 0.0
        for i in range(n):
            T[i] = d[i]
            T[ld + i] = e[i] if i < n - 1 else 0.0
        B = np.diag(d)
        for i in range(l, k):
            B[i, k] = e[i]
        for i in range(max(k, l) + 1, n):
            B[i - 1, i] = e[i - 1]
        if state != K.DS_STATE_RAW:
            for i in range(l, n - 1):
                B[i, i + 1] = e[i]
        K.check(lib.DSSetState(ds, state))
        w = np.zeros(n)
        K.check(lib.DSSolve(ds, w.ctypes.data_as(K.c_dp), None))
        K.check(lib.DSSort(ds, w.ctypes.data_as(K.c_dp), None, None, None, None))
        ref = np.linalg.svd(B[l:, l:], compute_uv=False)
        assert np.allclose(w[l:], ref, rtol=1e-13, atol=1e-14) and np.array_equal(w[:l], d[:l])
        U = np.zeros(n * n)
        VT = np.zeros(n * n)
        r_, c_ = C.c_int(), C.c_int()
        K.check(lib.DSGetMat(ds, 4, U.ctypes.data_as(K.c_dp), C.byref(r_), C.byref(c_)))     # DS_MAT_U
        K.check(lib.DSGetMat(ds, 5, VT.ctypes.data_as(K.c_dp), C.byref(r_), C.byref(c_)))    # DS_MAT_VT
        U = U.reshape(n, n).T
        VT = VT.reshape(n, n).T
        assert np.abs(U @ np.diag(w) @ VT - B).max() < 1e-13
        assert np.abs(U.T @ U - np.eye(n)).max() < 1e-13 and np.abs(VT @ VT.T - np.eye(n)).max() < 1e-13
        K.check(lib.DSDestroy(C.byref(ds)))


# ---- the product path on the GPU ---------------------------------------------------------------
@pytest.fixture(scope="module")
def be_gpu(gpu_ctx):
    return common.Backend("hip", gpu_ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_svd_matches_oracle(be_gpu, name):
    _check_against_oracle(be_gpu, name)


@pytest.mark.gpu
def test_gpu_svd_goldens(be_gpu):
    csr, nc = _csr(grcar(30, 30))
    for solver in ("cross", "trlanczos", "lanczos", "cyclic"):
        kw = dict(nsv=1, solver=solver, tol=1e-6, max_it=1000)
        s1 = common.solve_svd(be_gpu, csr, nc, **kw)["sigma"][0]
        sn = common.solve_svd(be_gpu, csr, nc, which="smallest", **kw)["sigma"][0]
        assert ("%.4f" % s1, "%.4f" % sn, "%.4f" % (s1 / sn)) == EX8
    csr, nc = _csr(rdb200())
    r = common.solve_svd(be_gpu, csr, nc, nsv=4)
    assert ["%.5f" % x for x in r["sigma"][:4]] == EX14 and r["its"] == G["ex14"]["its"] and r["type"] == G["ex14"]["type"]
    g = G["ex15"]
    csr, nc = _csr(lauchli(g["n"], g["mu"]))
    r = common.solve_svd(be_gpu, csr, nc, nsv=1, solver="trlanczos")
    assert r["its"] == g["its"] and r["nconv"] == g["nconv"]
    assert ["%.6f" % x for x in r["sigma"]] == g["sigma"]


@pytest.mark.gpu
def test_gpu_transpose_and_rectangular_spmv(gpu_ctx):
    """ksg_mat_create_csr_rect / ksg_mat_transpose / ksg_mat_get_csr: index arrays bit-exact
    against scipy's transpose, products against the host."""
    from slepc_amd import ksgpu
    rng = np.random.default_rng(8)
    for M, N, dens in ((700, 300, 0.02), (129, 1000, 0.01), (64, 64, 0.2), (5, 3000, 0.3)):
        D = np.where(rng.random((M, N)) < dens, rng.standard_normal((M, N)), 0.0)
        D[M // 2, :] = 0.0                                 # an empty row
        A = csr_matrix(D)
        A.sort_indices()
        dA = ksgpu.DeviceCSR(gpu_ctx, A.indptr, A.indices, A.data, cols=(N, N, 0))
        assert (dA.n, dA.N, dA.nc, dA.NC) == (M, M, N, N)
        rp, ci, va = dA.get_csr()
        assert np.array_equal(rp, A.indptr) and np.array_equal(ci, A.indices) and va.tobytes() == A.data.tobytes()
        dT = dA.transpose()
        T = csr_matrix(A.T)
        T.sort_indices()
        assert (dT.n, dT.nc) == (N, M)
        rp, ci, va = dT.get_csr()
        assert np.array_equal(rp, T.indptr) and np.array_equal(ci, T.indices) and va.tobytes() == T.data.tobytes()
        x, y = rng.standard_normal(N), rng.standard_normal(M)
        dx, dy, dax, daty = gpu_ctx.vec_alloc(N), gpu_ctx.vec_alloc(M), gpu_ctx.vec_alloc(M), gpu_ctx.vec_alloc(N)
        gpu_ctx.upload(dx, x)
        gpu_ctx.upload(dy, y)
        dA.mult(dx, dax)
        dT.mult(dy, daty)
        ax, aty = gpu_ctx.download(dax, M), gpu_ctx.download(daty, N)
        assert np.abs(ax - A @ x).max() <= 1e-13 * max(1.0, np.abs(A @ x).max())
        assert np.abs(aty - T @ y).max() <= 1e-13 * max(1.0, np.abs(T @ y).max())
        for p in (dx, dy, dax, daty):
            gpu_ctx.vec_free(p)
        dT.destroy()
        dA.destroy()


EX8_1 = G["ex8"]["text"]
EX14_1 = G["ex14"]["text"]


def _run_example(name, *args):
    import subprocess
    exe = os.path.join(ROOT, "examples", "bin", name)
    if not os.path.exists(exe):
        pytest.fail("%s not built: run `python __graft_entry__.py build`" % exe)
    r = subprocess.run([exe, *args], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    return r.stdout.decode()


@pytest.mark.gpu
def test_examples_match_reference_output():
    """runex8_1 (./ex8) and runex14_1 (./ex14 -file rdb200.petsc -svd_nsv 4 -terse) of
    src/svd/examples/tutorials/makefile:55-63, and ex8 with the other solver."""
    assert _run_example("ex8") == EX8_1
    assert _run_example("ex8", "-svd_type", "trlanczos") == EX8_1
    assert _run_example("ex8", "-svd_type", "trlanczos", "-svd_trlanczos_oneside") == EX8_1
    assert _run_example("ex8", "-svd_type", "lanczos") == EX8_1
    assert _run_example("ex8", "-svd_type", "cyclic") == EX8_1
    assert _run_example("ex14", "-file", os.path.join(MATDIR, "rdb200.petsc"), "-svd_nsv", "4", "-terse") == EX14_1
