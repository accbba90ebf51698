"""The compiled kernels of the CPU restatement (oracle/ckern.c) against its numpy form, and a
whole solve on them against the same solve on numpy: they are the CPU baseline bench.py times,
so they must compute what the pinned oracle computes."""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

from oracle import bv as obv, ckern, eps as oeps
from slepc_amd import matgen as mg


@pytest.fixture(scope="module")
def kern():
    ckern.build()
    return ckern.Kernels(3)


def test_spmv_bit_exact(kern):
    for csr in (mg.laplacian2d(37, 29), mg.markov(23), mg.rmat_laplacian(10)):
        rp, ci, va = csr
        n = len(rp) - 1
        x = np.random.default_rng(1).standard_normal(n)
        y = np.zeros(n)
        kern.csr(rp, ci, va).mult(x, y)
        ref = np.array([sum(va[p] * x[ci[p]] for p in range(rp[i], rp[i + 1])) if rp[i + 1] > rp[i] else 0.0
                        for i in range(n)])
        # every row's products added in storage order, products rounded before the sum
        want = np.zeros(n)
        for i in range(n):
            s = 0.0
            for p in range(rp[i], rp[i + 1]):
                s = s + va[p] * x[ci[p]]
            want[i] = s
        assert np.array_equal(y, want) and np.allclose(y, ref)


@pytest.mark.parametrize("n,m,l,k", [(1000, 9, 0, 9), (4099, 33, 2, 30), (7, 5, 1, 4), (1, 3, 0, 3)])
def test_bv_ops_match_numpy(kern, n, m, l, k):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, m))
    G, O = kern.bv(n, m), obv.BV(n, m)
    G.array[:, :] = A
    O.array[:, :] = A
    for b in (G, O):
        b.set_active_columns(l, k)
    y = rng.standard_normal(n)
    q1, q2 = np.zeros(k - l), np.zeros(k - l)
    G.op_dotvec(y, q1)
    O.op_dotvec(y, q2)
    assert np.abs(q1 - q2).max() <= 1e-13 * np.sqrt(n) * 10
    y1, y2 = y.copy(), y.copy()
    G.op_multvec(-0.5, 2.0, y1, q2)
    O.op_multvec(-0.5, 2.0, y2, q2)
    assert np.abs(y1 - y2).max() <= 1e-12 * max(1.0, np.abs(y2).max())
    y1[:] = np.nan
    G.op_multvec(1.0, 0.0, y1, q2)
    O.op_multvec(1.0, 0.0, y2, q2)
    assert np.abs(y1 - y2).max() <= 1e-12 * max(1.0, np.abs(y2).max())
    Q = rng.standard_normal((m, m))
    s, e = l, max(l + 1, k - 1)
    G.op_multinplace(Q, s, e)
    O.op_multinplace(Q, s, e)
    assert np.abs(G.array - O.array).max() <= 1e-12 * max(1.0, np.abs(O.array).max())
    G.op_scale(l, -2.5)
    O.op_scale(l, -2.5)
    G.op_scale(-1, 0.5)
    O.op_scale(-1, 0.5)
    G.copy_column(l, m - 1)
    O.copy_column(l, m - 1)
    assert np.abs(G.array - O.array).max() <= 1e-12 * max(1.0, np.abs(O.array).max())
    ng, no = G.op_norm(l), O.op_norm(l)
    assert abs(ng - no) <= 1e-13 * no
    assert G.bytes == O.bytes                      # the same byte model


@pytest.mark.parametrize("herm", [True, False])
def test_solve_on_compiled_kernels(kern, herm):
    rp, ci, va = mg.laplacian2d(60, 41) if herm else mg.markov(40)
    n = len(rp) - 1
    S = csr_matrix((va, ci, rp))
    v0 = oeps.splitmix64_uniform(np.arange(n, dtype=np.uint64))
    kw = dict(hermitian=herm, nev=5, tol=1e-10, v0=v0, which=None if herm else "largest_real")
    ref = oeps.EPS(lambda x, y: y.__setitem__(slice(None), S @ x), n, **kw).solve()
    got = oeps.EPS(kern.csr(rp, ci, va).mult, n, bv_factory=kern.bv, **kw).solve()
    assert got.nconv >= 5 and ref.nconv >= 5
    for i in range(5):
        a, b = got.get_eigenvalue(i), ref.get_eigenvalue(i)
        assert abs(a[0] - b[0]) <= 1e-10 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-10
