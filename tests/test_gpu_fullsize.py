"""Checks at the full size of the benchmark configuration (4096 x 4096 grid, 16.7M rows),
where the oracle cannot run a whole solve in test time: known answers and
size-independent properties for the kernels, and a short oracle-vs-product run of the
first restarts with a narrower basis."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = 4096


@pytest.fixture(scope="module")
def lap(gpu_ctx):
    from slepc_amd import ksgpu, matgen
    rp, ci, va = matgen.laplacian2d(G)
    A = ksgpu.DeviceCSR(gpu_ctx, rp, ci, va)
    yield A, (rp, ci, va)
    A.destroy()


def _stencil_apply(x):
    """The 5-point operator of ex2.c:56-63 applied to a grid function, no matrix."""
    X = x.reshape(G, G)
    Y = 4.0 * X
    Y[1:, :] -= X[:-1, :]
    Y[:-1, :] -= X[1:, :]
    Y[:, 1:] -= X[:, :-1]
    Y[:, :-1] -= X[:, 1:]
    return Y.reshape(-1)


def test_spmv_known_answers_fullsize(gpu_ctx, lap):
    A, _ = lap
    n = G * G
    xd, yd = gpu_ctx.vec_alloc(n), gpu_ctx.vec_alloc(n)
    # A * 1: 0 inside, 1 on edges, 2 in corners -- exact
    gpu_ctx.vec_set(xd, 1.0, n)
    A.mult(xd, yd)
    y = gpu_ctx.download(yd, n).reshape(G, G)
    assert np.all(y[1:-1, 1:-1] == 0.0)
    assert np.all(y[0, 1:-1] == 1.0) and np.all(y[-1, 1:-1] == 1.0)
    assert np.all(y[1:-1, 0] == 1.0) and np.all(y[1:-1, -1] == 1.0)
    assert y[0, 0] == y[0, -1] == y[-1, 0] == y[-1, -1] == 2.0
    # integer-valued grid function: exact whatever the order of the sums
    x = (np.arange(n, dtype=np.int64) * 2654435761 % 13 - 6).astype(np.float64)
    gpu_ctx.upload(xd, x)
    A.mult(xd, yd)
    assert np.array_equal(gpu_ctx.download(yd, n), _stencil_apply(x))
    # linearity: A(2x) = 2 A x bit for bit (scaling by 2 is exact)
    r = np.random.default_rng(0).standard_normal(n)
    gpu_ctx.upload(xd, r)
    A.mult(xd, yd)
    y1 = gpu_ctx.download(yd, n)
    gpu_ctx.upload(xd, 2.0 * r)
    A.mult(xd, yd)
    assert np.array_equal(gpu_ctx.download(yd, n), 2.0 * y1)
    gpu_ctx.vec_free(xd)
    gpu_ctx.vec_free(yd)


def test_gram_schmidt_properties_fullsize(gpu_ctx):
    """CGS2 through the op table on 12 hashed columns of 16.7M rows: orthonormal to
    working accuracy, coefficients reproduce the input (V R = A), and the products the
    fused kernel leaves behind equal a fresh reduction to rounding."""
    from slepc_amd import ksgpu
    n, m = G * G, 12
    V = ksgpu.DeviceBV(gpu_ctx, n, m + 1)
    for j in range(m + 1):
        gpu_ctx.vec_fill_hash(V.column(j), n, 0, 77 + j)
    a_last = V.get_column(m - 1)
    R = np.zeros((m, m))
    for j in range(m):
        h = np.zeros(j)
        for _ in range(2):
            c = V.dotvec_cols(0, j + 1, j)
            if j:
                V.set_active(0, j)
                V.multvec(-1.0, 1.0, V.column(j), c[:j])
                h += c[:j]
        c = V.dotvec_cols(0, j + 1, j)
        if j:
            # the same products once more, this time from the reduction kernel
            gpu_ctx.set_option(ksgpu.OPT_FUSE_DOTS, 0)
            c2 = V.dotvec_cols(0, j + 1, j)
            gpu_ctx.set_option(ksgpu.OPT_FUSE_DOTS, 1)
            assert np.abs(c - c2).max() <= 1e-12 * max(1.0, np.abs(c2).max())
        nrm = np.sqrt(c[j])
        V.scale(j, 1.0 / nrm)
        R[:j, j] = h
        R[j, j] = nrm
    V.set_active(0, m)
    W = ksgpu.DeviceBV(gpu_ctx, n, m)
    V.copy_to(W)
    M = np.zeros((m, m), order="F")
    V.dot(W, M)
    assert np.abs(M - np.eye(m)).max() < 5e-13
    # last column of V R equals the hashed input column
    y = gpu_ctx.vec_alloc(n)
    V.multvec(1.0, 0.0, y, np.ascontiguousarray(R[:, m - 1]))
    assert np.abs(gpu_ctx.download(y, n) - a_last).max() < 1e-11
    gpu_ctx.vec_free(y)
    W.destroy()
    V.destroy()


def test_first_restarts_match_oracle_fullsize(gpu_ctx, lap):
    """The first two restarts of the benchmark's own configuration -- 16.7M rows, nev=32, ncv=64,
    default start vector on both sides -- product against oracle: the 32 leading Ritz values
    and their error estimates agree after each restart.  This runs the 65-column reduction,
    the fused and deferred updates and the 64 -> 32+ column restart product at full size
    against the CPU restatement (on its compiled kernels, 16 threads)."""
    from oracle import ckern, eps as oeps
    from slepc_amd import kseps as K
    _, (rp, ci, va) = lap
    n = G * G
    its, nev = 2, 32
    A = K.Mat.create_aij_hip(gpu_ctx, rp, ci, va)
    eps = K.EPS(gpu_ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_HEP)
    eps.set_dimensions(nev)
    eps.set_tolerances(0.0, its)
    assert eps.dimensions()[1] in (0, 64)
    got = {}
    eps.set_monitor(lambda i, nconv, er, ei, ee, nest: got.__setitem__(i, (er[:nev].copy(), ee[:nev].copy(), nest)))
    eps.solve()
    assert eps.dimensions()[1] == 64
    kern = ckern.Kernels(16)
    want = {}
    o = oeps.EPS(kern.csr(rp, ci, va).mult, n, hermitian=True, nev=nev, max_it=its, bv_factory=kern.bv,
                 monitor=lambda i, nconv, er, ei, ee, nv: want.__setitem__(i, (er[:nev].copy(), ee[:nev].copy(), nv)))
    o.solve()
    assert o.ncv == 64
    assert sorted(got) == sorted(want) == [1, 2]
    for i in (1, 2):
        assert got[i][2] == want[i][2] == 64
        assert np.abs(got[i][0] - want[i][0]).max() <= 1e-10 * 8.0
        assert np.abs(got[i][1] - want[i][1]).max() <= 1e-7 * max(want[i][1].max(), 1e-30) + 1e-12
    eps.destroy()
    A.destroy()


def test_block_product_and_residual_fullsize(gpu_ctx, lap):
    """At 16.7M rows: the one-pass block product equals one product per column bit for bit and
    reproduces the stencil on integer data; the store-free residual norm equals the norm of
    A x - k x formed on the host."""
    from slepc_amd import ksgpu
    A, _ = lap
    n = G * G
    m = 5
    V, W = ksgpu.DeviceBV(gpu_ctx, n, m), ksgpu.DeviceBV(gpu_ctx, n, m)
    for j in range(m):
        gpu_ctx.vec_fill_hash(V.column(j), n, 0, 77 + j)
    x_int = (np.arange(n, dtype=np.int64) * 40503 % 17 - 8).astype(np.float64)
    gpu_ctx.upload(V.column(2), x_int)
    V.matmult(A, W)
    yd = gpu_ctx.vec_alloc(n)
    for j in range(m):
        A.mult(V.column(j), yd)
        assert np.array_equal(gpu_ctx.download(yd, n), gpu_ctx.download(W.column(j), n))
    assert np.array_equal(gpu_ctx.download(W.column(2), n), _stencil_apply(x_int))
    # residual of column 0 with a shift: against the host in extended precision
    x = gpu_ctx.download(V.column(0), n)
    ax = gpu_ctx.download(W.column(0), n)
    k = 3.75
    r = ax.astype(np.longdouble) - np.longdouble(k) * x
    want = float(np.sqrt(np.sum(r * r)))
    got = A.residual_norm(k, 0.0, V.column(0))
    assert abs(got - want) <= 1e-12 * want
    gpu_ctx.vec_free(yd)
    V.destroy()
    W.destroy()


def test_rmat_scale22_solve_matches_oracle(gpu_ctx):
    """A power-law operator large enough to be held in the binned form by the library's own choice
    (R-MAT scale 22: 4.2M rows, 6.9e7 entries, x = 32 MiB), solved to 1e-10 for 16 pairs on the device
    and by the CPU restatement on its compiled kernels: eigenvalues to 1e-10, residuals, restart count."""
    from oracle import ckern, eps as oeps
    from slepc_amd import ksgpu, kseps as K, matgen
    rp, ci, va = matgen.rmat_laplacian(22)
    n = len(rp) - 1
    nev, tol = 16, 1e-10
    A = K.Mat.create_aij_hip(gpu_ctx, rp, ci, va)
    dev = ksgpu.DeviceCSR.__new__(ksgpu.DeviceCSR)
    dev.ctx, dev.lib, dev.h = gpu_ctx, gpu_ctx.lib, A.p.contents.data
    assert dev.describe().startswith("binned part"), dev.describe()
    eps = K.EPS(gpu_ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_HEP)
    eps.set_dimensions(nev)
    eps.set_tolerances(tol, 200)
    eps.solve()
    assert eps.reason == K.EPS_CONVERGED_TOL and eps.nconv >= nev
    got = [eps.get_eigenvalue(i)[0] for i in range(nev)]
    errs = [eps.compute_error(i) for i in range(nev)]
    kern = ckern.Kernels(16)
    o = oeps.EPS(kern.csr(rp, ci, va).mult, n, hermitian=True, nev=nev, tol=tol, max_it=200, bv_factory=kern.bv)
    o.solve()
    assert o.nconv >= nev and abs(o.its - eps.its) <= 1
    for i in range(nev):
        lam = o.get_eigenvalue(i)[0]
        assert abs(got[i] - lam) <= 1e-10 * abs(lam), (i, got[i], lam)
    assert max(errs) < 10 * tol
    eps.destroy()
    A.destroy()


def test_markov3000_first_restarts_match_oracle(gpu_ctx):
    """BASELINE configuration 4 at its full size -- ex5's Markov chain, m = 3000 (4.5M rows, 1.8e7 entries),
    nonsymmetric Krylov-Schur, largest real part, nev = 16 (ncv = 32), start vector e0 + e1 (ex5.c:86-94):
    the Ritz values (real and imaginary parts) and error estimates the product reports after each of the
    first three restarts against the CPU restatement on its compiled kernels."""
    from oracle import ckern, eps as oeps
    from slepc_amd import kseps as K, matgen
    m, nev, its = 3000, 16, 3
    rp, ci, va = matgen.markov(m)
    n = len(rp) - 1
    assert n == m * (m + 1) // 2
    start = np.zeros(n)
    start[:2] = 1.0
    A = K.Mat.create_aij_hip(gpu_ctx, rp, ci, va)
    eps = K.EPS(gpu_ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_NHEP)
    eps.set_which(K.EPS_LARGEST_REAL)
    eps.set_dimensions(nev)
    eps.set_tolerances(0.0, its)
    v0 = A.create_vecs()
    v0.set_values(start)
    eps.set_initial_space([v0])
    got = {}
    eps.set_monitor(lambda i, nconv, er, ei, ee, nest: got.__setitem__(i, (er[:nev].copy(), ei[:nev].copy(), ee[:nev].copy(), nconv)))
    eps.solve()
    assert eps.dimensions()[1] == 32
    kern = ckern.Kernels(16)
    want = {}
    o = oeps.EPS(kern.csr(rp, ci, va).mult, n, hermitian=False, nev=nev, max_it=its, which="largest_real", v0=start, bv_factory=kern.bv,
                 monitor=lambda i, nconv, er, ei, ee, nv: want.__setitem__(i, (er[:nev].copy(), ei[:nev].copy(), ee[:nev].copy(), nconv)))
    o.solve()
    assert o.ncv == 32 and sorted(got) == sorted(want) == [1, 2, 3]
    for i in (1, 2, 3):
        assert got[i][3] == want[i][3]
        assert np.abs(got[i][0] - want[i][0]).max() <= 1e-10 and np.abs(got[i][1] - want[i][1]).max() <= 1e-10
        assert np.abs(got[i][2] - want[i][2]).max() <= 1e-6 * max(want[i][2].max(), 1e-30) + 1e-12
    v0.destroy()
    eps.destroy()
    A.destroy()


def test_lap3d_256_first_restarts_match_oracle(gpu_ctx):
    """The 3-D 7-point Laplacian of BASELINE configuration 3 on a 256^3 grid (16.7M rows, the rows one of
    eight GPUs owns at 512^3), nev = 32, ncv = 64, default start vector: the Ritz values and error
    estimates after each of the first two restarts against the CPU restatement on its compiled kernels."""
    from oracle import ckern, eps as oeps
    from slepc_amd import kseps as K, matgen
    g, nev, its = 256, 32, 2
    rp, ci, va = matgen.laplacian3d(g, g, g)
    n = g ** 3
    A = K.Mat.create_aij_hip(gpu_ctx, rp, ci, va)
    eps = K.EPS(gpu_ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_HEP)
    eps.set_dimensions(nev)
    eps.set_tolerances(0.0, its)
    got = {}
    eps.set_monitor(lambda i, nconv, er, ei, ee, nest: got.__setitem__(i, (er[:nev].copy(), ee[:nev].copy(), nest)))
    eps.solve()
    kern = ckern.Kernels(16)
    want = {}
    o = oeps.EPS(kern.csr(rp, ci, va).mult, n, hermitian=True, nev=nev, max_it=its, bv_factory=kern.bv,
                 monitor=lambda i, nconv, er, ei, ee, nv: want.__setitem__(i, (er[:nev].copy(), ee[:nev].copy(), nv)))
    o.solve()
    assert sorted(got) == sorted(want) == [1, 2]
    for i in (1, 2):
        assert got[i][2] == want[i][2] == 64
        assert np.abs(got[i][0] - want[i][0]).max() <= 1e-10 * 12.0
        assert np.abs(got[i][1] - want[i][1]).max() <= 1e-7 * max(want[i][1].max(), 1e-30) + 1e-12
    eps.destroy()
    A.destroy()


def test_lap3d_512_first_restarts_match_oracle(gpu_ctx):
    """BASELINE configuration 3 at its full size on one GPU -- the 134M-row 7-point Laplacian on the 512^3
    grid, nev = 32, ncv = 64, default start vector: Ritz values and error estimates after each of the first
    two restarts against those of the CPU restatement on its compiled kernels, kept as a fixture
    (tests/golden/make_first_restarts_fixture.py: 70 GB of basis on the host and more than a minute on 16
    threads, so it is not run inside the test).  Offsets beyond 2^31 elements, the row-code product at
    134M rows and the 64 -> 32+ column restart product are all on this path."""
    import json
    from slepc_amd import kseps as K, matgen
    g, nev, its = 512, 32, 2
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lap3d512_oracle_first_restarts.json")) as f:
        want = json.load(f)
    rp, ci, va = matgen.laplacian3d(g, g, g)
    assert want["rows"] == g ** 3 and want["entries"] == int(rp[-1]) and want["nev"] == nev
    A = K.Mat.create_aij_hip(gpu_ctx, rp, ci, va)
    eps = K.EPS(gpu_ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_HEP)
    eps.set_dimensions(nev)
    eps.set_tolerances(0.0, its)
    got = {}
    eps.set_monitor(lambda i, nconv, er, ei, ee, nest: got.__setitem__(i, (er[:nev].copy(), ee[:nev].copy(), nest, nconv)))
    eps.solve()
    eps.destroy()
    A.destroy()
    assert sorted(got) == [1, 2] and sorted(want["restarts"]) == ["1", "2"]
    for i in (1, 2):
        w = want["restarts"][str(i)]
        wre, wee = np.array(w["re"]), np.array(w["errest"])
        assert got[i][2] == w["nest"] == 64 and got[i][3] == w["nconv"]
        assert np.abs(got[i][0] - wre).max() <= 1e-10 * 12.0
        assert np.abs(got[i][1] - wee).max() <= 1e-7 * max(wee.max(), 1e-30) + 1e-12
