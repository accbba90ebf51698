"""Parity of the HIP CSR SpMV (through the C ABI) against the CPU oracle.

Index handling must be exact: on matrices whose rows all take the
one-lane-per-row path the result must be bit-identical to a sequential CSR
loop; grouped and long-row paths sum in a different order and are held to
1e-13 relative to sum |a_ij x_j|.
"""
import numpy as np
import pytest

from oracle import matgen as og

pytestmark = pytest.mark.gpu


def _run(ctx, rp, ci, va, x):
    from slepc_amd import ksgpu
    n = len(rp) - 1
    A = ksgpu.DeviceCSR(ctx, rp, ci, va)
    xd, yd = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.upload(xd, x)
    ctx.upload(yd, np.full(n, np.nan))
    A.mult(xd, yd)
    y = ctx.download(yd, n)
    info = A.info()
    ctx.vec_free(xd)
    ctx.vec_free(yd)
    A.destroy()
    return y, info


@pytest.mark.parametrize("gen", ["lap1d", "lap2d", "lap3d", "markov"])
def test_stencils_bit_exact(gpu_ctx, gen):
    from slepc_amd import matgen as mg
    rp, ci, va = {"lap1d": lambda: mg.laplacian1d(100003), "lap2d": lambda: mg.laplacian2d(301, 217),
                  "lap3d": lambda: mg.laplacian3d(41, 37, 29), "markov": lambda: mg.markov(300)}[gen]()
    n = len(rp) - 1
    x = np.random.default_rng(0).standard_normal(n)
    y, info = _run(gpu_ctx, rp, ci, va, x)
    yo = og.csr_matvec_fast(rp, ci, va, x, np.zeros(n))
    assert info["long_rows"] == 0
    if info["group_tiles"] == 0:
        assert np.array_equal(y, yo)
    else:   # only the last, short tile may be grouped
        assert info["group_tiles"] <= 1
        assert np.abs(y - yo).max() <= 1e-13 * np.abs(yo).max()


def test_golden_bv_test7(gpu_ctx):
    """Reference bv test7: 1-D Laplacian times staircase vectors = e_0 + e_j - e_{j+1};
    the printed error is exactly 0."""
    from slepc_amd import ksgpu, matgen as mg
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bv_misc.json")))
    n, k = 10, 5
    rp, ci, va = mg.laplacian1d(n)
    A = ksgpu.DeviceCSR(gpu_ctx, rp, ci, va)
    V, W = ksgpu.DeviceBV(gpu_ctx, n, k), ksgpu.DeviceBV(gpu_ctx, n, k)
    Vh = np.zeros((n, k))
    for j in range(k):
        Vh[:j + 1, j] = 1.0
    V.from_numpy(Vh)
    V.matmult(A, W)
    Wh = W.to_numpy()
    for j in range(k):
        e = np.zeros(n)
        e[0] += 1.0
        e[j] += 1.0
        e[j + 1] -= 1.0
        assert "%g" % np.linalg.norm(Wh[:, j] - e) == g["test7_errors"][0].rstrip(".")
    for o in (A, V, W):
        o.destroy()


def test_irregular_rows(gpu_ctx):
    """R-MAT graph Laplacian: grouped tiles and long rows."""
    from slepc_amd import matgen as mg
    rp, ci, va = mg.rmat_laplacian(16)
    n = len(rp) - 1
    assert np.diff(rp).max() > 4096          # longer than one LDS tile
    x = np.random.default_rng(1).standard_normal(n)
    y, info = _run(gpu_ctx, rp, ci, va, x)
    assert info["long_rows"] > 0 and info["group_tiles"] > 0
    yo = og.csr_matvec_fast(rp, ci, va, x, np.zeros(n))
    absA = np.abs(va)
    bound = og.csr_matvec_fast(rp, ci, absA, np.abs(x), np.zeros(n))
    assert np.all(np.abs(y - yo) <= 1e-13 * np.maximum(bound, 1.0))
    # integer data: exact whatever the summation order
    xi = np.arange(n, dtype=np.float64) % 7 - 3
    y, _ = _run(gpu_ctx, rp, ci, va, xi)
    assert np.array_equal(y, og.csr_matvec_fast(rp, ci, va, xi, np.zeros(n)))


def test_edge_cases(gpu_ctx):
    from slepc_amd import ksgpu
    # empty rows, a single dense row, a 1x1 matrix
    rp = np.array([0, 0, 3, 3, 4], dtype=np.int64)
    ci = np.array([0, 1, 3, 2], dtype=np.int32)
    va = np.array([1.0, 2.0, 3.0, -1.0])
    x = np.array([1.0, 10.0, 100.0, 1000.0])
    y, _ = _run(gpu_ctx, rp, ci, va, x)
    assert np.array_equal(y, [0.0, 3021.0, 0.0, -100.0])
    y, _ = _run(gpu_ctx, np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int32), np.array([2.5]), np.array([4.0]))
    assert np.array_equal(y, [10.0])
    with pytest.raises(ksgpu.KsgError):
        ksgpu.DeviceCSR(gpu_ctx, np.array([0, 1], dtype=np.int64), np.array([5], dtype=np.int32), np.array([1.0]))


@pytest.mark.parametrize("gen", ["lap2d", "markov", "rmat"])
def test_fused_residual_norm(gpu_ctx, gen):
    """ksg_mat_residual_norm (one pass, nothing stored) against MatMult + AXPY + norm on the
    host in extended precision: real shift, complex pair, zero shift."""
    from scipy.sparse import csr_matrix
    from slepc_amd import ksgpu, matgen as mg
    if gen == "lap2d":
        rp, ci, va = mg.laplacian2d(211, 97)
    elif gen == "markov":
        rp, ci, va = mg.markov(70)
    else:
        rp, ci, va = mg.rmat_laplacian(14, 8, seed=3)      # hub rows: grouped and long-row tiles
    n = len(rp) - 1
    ctx = gpu_ctx
    A = ksgpu.DeviceCSR(ctx, rp, ci, va)
    rng = np.random.default_rng(11)
    xr, xi = rng.standard_normal(n), rng.standard_normal(n)
    dx, dxi = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.upload(dx, xr)
    ctx.upload(dxi, xi)
    S = csr_matrix((va, ci, rp), shape=(n, n))
    axr = (S @ xr).astype(np.longdouble)
    for kr, ki, use_xi in ((3.25, 0.0, False), (-1.5, 0.75, True), (0.0, 0.0, False)):
        ref = axr - np.longdouble(kr) * xr
        if use_xi:
            ref = ref + np.longdouble(ki) * xi
        want = float(np.sqrt(np.sum(ref * ref)))
        got = A.residual_norm(kr, ki, dx, dxi if use_xi else None)
        assert abs(got - want) <= 1e-12 * want
        # the same call twice gives the same bits (fixed reduction order)
        assert got == A.residual_norm(kr, ki, dx, dxi if use_xi else None)
    ctx.vec_free(dx)
    ctx.vec_free(dxi)
    A.destroy()


@pytest.mark.parametrize("gen", ["lap2d", "lap3d", "markov", "rmat"])
def test_block_product_equals_column_products(gpu_ctx, gen):
    """BVMatMult as one pass over the matrix for all the columns: bit for bit the result of
    one MatMult per column (stream, grouped and long-row tiles), active ranges honoured."""
    from slepc_amd import ksgpu, matgen as mg
    rp, ci, va = {"lap2d": lambda: mg.laplacian2d(301, 217), "lap3d": lambda: mg.laplacian3d(41, 37, 29),
                  "markov": lambda: mg.markov(200), "rmat": lambda: mg.rmat_laplacian(16)}[gen]()
    n = len(rp) - 1
    A = ksgpu.DeviceCSR(gpu_ctx, rp, ci, va)
    rng = np.random.default_rng(4)
    m = 21
    V, W = ksgpu.DeviceBV(gpu_ctx, n, m), ksgpu.DeviceBV(gpu_ctx, n, m)
    Vh = rng.standard_normal((n, m))
    V.from_numpy(Vh)
    yd = gpu_ctx.vec_alloc(n)
    for (l, k) in ((0, m), (2, 10), (5, 6), (3, 20)):
        W.from_numpy(np.full((n, m), np.nan))
        V.set_active(l, k)
        W.set_active(l, k)
        V.matmult(A, W)
        Wh = W.to_numpy()
        assert np.isnan(Wh[:, :l]).all() and np.isnan(Wh[:, k:]).all()
        for j in range(l, k):
            A.mult(V.column(j), yd)
            assert np.array_equal(gpu_ctx.download(yd, n), Wh[:, j]), (gen, l, k, j)
    gpu_ctx.vec_free(yd)
    for o in (A, V, W):
        o.destroy()


def test_packed_hot_columns_bit_exact(gpu_ctx):
    """A power-law operator with the x entries of its most referenced columns packed
    (KSG_OPT_HOT_COLS): the product, the store-free residual, the block product, the CSR read
    back and the transpose are bit for bit those of the same operator without the packing."""
    from slepc_amd import ksgpu, matgen as mg
    ctx = gpu_ctx
    rp, ci, va = mg.rmat_laplacian(15, 8, seed=7)
    n = len(rp) - 1
    rng = np.random.default_rng(2)
    x = rng.standard_normal(n)
    xd, yd, zd = ctx.vec_alloc(n), ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.upload(xd, x)
    ctx.set_option(ksgpu.OPT_HOT_COLS, 0)
    A0 = ksgpu.DeviceCSR(ctx, rp, ci, va)
    assert A0.hot_columns() == 0
    ctx.set_option(ksgpu.OPT_HOT_COLS, 1024)
    A1 = ksgpu.DeviceCSR(ctx, rp, ci, va)
    ctx.set_option(ksgpu.OPT_HOT_COLS, 1048576)
    nhot = A1.hot_columns()
    assert 0 < nhot <= 1024
    # a stencil has no hot columns whatever the limit
    ctx.set_option(ksgpu.OPT_HOT_COLS, 64)
    L = ksgpu.DeviceCSR(ctx, *mg.laplacian2d(60, 60))
    ctx.set_option(ksgpu.OPT_HOT_COLS, 1048576)
    assert L.hot_columns() == 0
    L.destroy()
    A0.mult(xd, yd)
    ctx.upload(zd, np.full(n, np.nan))
    A1.mult(xd, zd)
    y0, y1 = ctx.download(yd, n), ctx.download(zd, n)
    assert np.array_equal(y0, y1)
    from scipy.sparse import csr_matrix
    ref = csr_matrix((va, ci, rp), shape=(n, n)) @ x
    assert np.abs(y1 - ref).max() <= 1e-12 * np.abs(ref).max()
    # a second product after x changed picks the new hot entries up
    x2 = rng.standard_normal(n)
    ctx.upload(xd, x2)
    A0.mult(xd, yd)
    A1.mult(xd, zd)
    assert np.array_equal(ctx.download(yd, n), ctx.download(zd, n))
    assert A0.residual_norm(1.75, 0.0, xd) == A1.residual_norm(1.75, 0.0, xd)
    r0, c0, v0 = A0.get_csr()
    r1, c1, v1 = A1.get_csr()
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1) and np.array_equal(v0, v1) and np.array_equal(c1, ci)
    T0, T1 = A0.transpose(), A1.transpose()
    for a, b in zip(T0.get_csr(), T1.get_csr()):
        assert np.array_equal(a, b)
    m = 5
    V, W0, W1 = (ksgpu.DeviceBV(ctx, n, m) for _ in range(3))
    V.from_numpy(rng.standard_normal((n, m)))
    V.matmult(A0, W0)
    V.matmult(A1, W1)
    assert np.array_equal(W0.to_numpy(), W1.to_numpy())
    for o in (A0, A1, T0, T1, V, W0, W1):
        o.destroy()
    for p in (xd, yd, zd):
        ctx.vec_free(p)


@pytest.mark.parametrize("ndistinct", [1, 3, 256, 257])
def test_value_table_spmv(gpu_ctx, ndistinct):
    """An operator with at most 256 distinct values is multiplied through one-byte indices into a
    value table, one with more through the values themselves: both give the bits of a sequential
    CSR loop (rows short enough for the one-lane-per-row path), the residual variant and the
    round trip through ksg_mat_get_csr included."""
    from slepc_amd import ksgpu
    rng = np.random.default_rng(ndistinct)
    n, per = 3000, 5
    rp = np.arange(0, (n + 1) * per, per, dtype=np.int64)
    ci = np.sort(rng.integers(0, n, (n, per)), axis=1).astype(np.int32).ravel()
    pool = rng.standard_normal(ndistinct)
    pool[0] = -0.0 if ndistinct > 1 else 1.5                 # told apart by bits: -0.0 and 0.0 differ
    if ndistinct > 2:
        pool[1] = 0.0
    va = pool[rng.integers(0, ndistinct, n * per)]
    va[:ndistinct] = pool                                     # every value occurs
    x = rng.standard_normal(n)
    A = ksgpu.DeviceCSR(gpu_ctx, rp, ci, va)
    xd, yd = gpu_ctx.vec_alloc(n), gpu_ctx.vec_alloc(n)
    gpu_ctx.upload(xd, x)
    A.mult(xd, yd)
    y = gpu_ctx.download(yd, n)
    ref = np.zeros(n)
    for r in range(n):
        s = 0.0
        for p in range(rp[r], rp[r + 1]):
            s += va[p] * x[ci[p]]
        ref[r] = s
    assert np.array_equal(y, ref)
    rp2, ci2, va2 = A.get_csr()
    assert np.array_equal(va2.view(np.uint64), va.view(np.uint64)) and np.array_equal(ci2, ci)
    nrm = A.residual_norm(0.75, 0.0, xd)
    assert abs(nrm - np.linalg.norm(ref - 0.75 * x)) < 1e-12 * np.linalg.norm(ref)
    A.destroy()
    gpu_ctx.vec_free(xd)
    gpu_ctx.vec_free(yd)


@pytest.mark.parametrize("hot", [0, 256])
def test_value_table_with_dense_diagonal(gpu_ctx, hot):
    """Off-diagonal entries with few distinct values and an arbitrary diagonal (graph Laplacians,
    shifted stencils): the entry stream carries one-byte indices, the diagonal is a dense vector.
    Bit for bit the sequential CSR loop on short rows, the CSR read back unchanged; with packed
    hot columns the diagonal entries keep their own column."""
    from slepc_amd import ksgpu, matgen as mg
    ctx = gpu_ctx
    rng = np.random.default_rng(11)
    # a 2-D stencil whose diagonal is replaced by random values: more than 256 distinct values
    rp, ci, va = mg.laplacian2d(70, 61)
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    va = va.copy()
    va[ci == rows] = rng.standard_normal(n) * 3.0
    va[(ci != rows) & (rng.random(va.size) < 0.3)] = 0.25
    x = rng.standard_normal(n)
    xd, yd = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.upload(xd, x)
    A = ksgpu.DeviceCSR(ctx, rp, ci, va)
    assert "the diagonal as a dense vector" in A.describe()
    A.mult(xd, yd)
    y = ctx.download(yd, n)
    ref = np.zeros(n)
    for r in range(n):
        s = 0.0
        for p in range(rp[r], rp[r + 1]):
            s += va[p] * x[ci[p]]
        ref[r] = s
    assert np.array_equal(y, ref)
    r2, c2, v2 = A.get_csr()
    assert np.array_equal(c2, ci) and np.array_equal(v2.view(np.uint64), va.view(np.uint64))
    assert abs(A.residual_norm(0.5, 0.0, xd) - np.linalg.norm(ref - 0.5 * x)) < 1e-12 * np.linalg.norm(ref)
    A.destroy()
    # the graph Laplacian of a power-law graph: -1 off the diagonal, degrees on it; grouped and
    # long-row tiles, packed columns or not: the same bits as the operator with its values streamed
    rp, ci, va = mg.rmat_laplacian(14, 8, seed=3)
    n2 = len(rp) - 1
    va = va.copy()
    va[ci == np.repeat(np.arange(n2), np.diff(rp))] += rng.random(n2)      # a shifted Laplacian: every diagonal value its own
    x2 = rng.standard_normal(n2)
    x2d, y2d, z2d = ctx.vec_alloc(n2), ctx.vec_alloc(n2), ctx.vec_alloc(n2)
    ctx.upload(x2d, x2)
    ctx.set_option(ksgpu.OPT_HOT_COLS, hot)
    L = ksgpu.DeviceCSR(ctx, rp, ci, va)
    # the same matrix with one off-diagonal value perturbed in 300 places: no table fits, the values are streamed
    vb = va.copy()
    rows2 = np.repeat(np.arange(n2), np.diff(rp))
    off = np.flatnonzero(ci != rows2)
    pick = rng.choice(off, 300, replace=False)
    vb[pick] = -1.0 - 1e-3 * np.arange(1, 301)
    Lb = ksgpu.DeviceCSR(ctx, rp, ci, vb)
    ctx.set_option(ksgpu.OPT_HOT_COLS, 1048576)
    assert "the diagonal as a dense vector" in L.describe() and "8-byte stream" in Lb.describe()
    assert (L.hot_columns() > 0) == (hot > 0)
    L.mult(x2d, y2d)
    yl = ctx.download(y2d, n2)
    from scipy.sparse import csr_matrix
    want = csr_matrix((va, ci, rp), shape=(n2, n2)) @ x2
    assert np.abs(yl - want).max() <= 1e-13 * np.abs(want).max() * 10
    # bit-equality of the two value paths where the values agree: rows none of whose entries was perturbed
    Lb.mult(x2d, z2d)
    yb = ctx.download(z2d, n2)
    touched = np.zeros(n2, dtype=bool)
    touched[rows2[pick]] = True
    assert np.array_equal(yl[~touched], yb[~touched])
    r3, c3, v3 = L.get_csr()
    assert np.array_equal(c3, ci) and np.array_equal(v3, va)
    for o in (L, Lb):
        o.destroy()
    for p in (xd, yd, x2d, y2d, z2d):
        ctx.vec_free(p)


@pytest.mark.parametrize("W,H,E,T,values", [(1024, 256, 4096, 2, "laplacian"), (64, 64, 1 << 20, 1, "laplacian"),
                                             (16384, 2048, 65536, 4, "laplacian"), (2048, 512, 8192, 3, "few"),
                                             (512, 128, 2048, 2, "any"), (4096, 100, 700, 2, "zeros"),
                                             (256, 8192, 300, 5, "nodiag"), (8192, 1024, 500, 4, "any")])
def test_binned_product(gpu_ctx, W, H, E, T, values):
    """A power-law operator whose well filled (row block, column block) cells are held in the binned
    form (KSG_OPT_BINS): the product agrees with the plain operator's and the sequential CSR loop's
    to rounding (a row's sum is formed from its CSR entries first, its binned entries after), for one
    value, few values (one-byte indices) and arbitrary values, with explicit zeros among the entries,
    with rows longer than a row block may be (cut into pieces) and with a diagonal that is not
    complete (its entries are then ordinary entries); the entries read back and the transpose are exactly the input's; residual norm and block product
    go through the same path; the second product gives the same bits as the first; an operator that
    is not power-law is left as it is."""
    import os
    from scipy.sparse import csr_matrix
    from slepc_amd import ksgpu, matgen as mg
    ctx = gpu_ctx
    rp, ci, va = mg.rmat_laplacian(14, 8, seed=11)
    n = len(rp) - 1
    rng = np.random.default_rng(7)
    offd = ci != np.repeat(np.arange(n), np.diff(rp))
    va = va.copy()
    if values == "few":
        va[offd] = rng.choice(np.array([-1.0, -0.5, 2.25, 1e-3]), size=int(offd.sum()))
    elif values == "any":
        va[offd] = rng.standard_normal(int(offd.sum()))
    elif values == "zeros":
        va[offd] = rng.choice(np.array([-1.0, 0.0]), size=int(offd.sum()))
    elif values == "nodiag":                               # rows 3 and 700 lose their diagonal entry
        keep = np.ones(len(ci), dtype=bool)
        for r in (3, 700):
            keep[rp[r]:rp[r + 1]] = ci[rp[r]:rp[r + 1]] != r
        rows = np.repeat(np.arange(n), np.diff(rp))[keep]
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int64)
        ci, va = ci[keep], va[keep]
    env = {"KSG_BINS_W": W, "KSG_BINS_H": H, "KSG_BINS_E": E, "KSG_BINS_T": T, "KSG_BINS_TASK_GROUPS": 16}
    old = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update({k: str(v) for k, v in env.items()})
        ctx.set_option(ksgpu.OPT_BINS, 1)
        A0 = ksgpu.DeviceCSR(ctx, rp, ci, va)
        ctx.set_option(ksgpu.OPT_BINS, 2)
        A1 = ksgpu.DeviceCSR(ctx, rp, ci, va)
        L = ksgpu.DeviceCSR(ctx, *mg.laplacian2d(70, 70))
    finally:
        ctx.set_option(ksgpu.OPT_BINS, 0)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert "binned part" not in A0.describe() and "binned part" not in L.describe()
    assert "binned part" in A1.describe(), A1.describe()
    want_values = {"laplacian": "values: one", "few": "1-byte indices", "zeros": "1-byte indices", "any": "8-byte stream",
                   "nodiag": "1-byte indices"}[values]       # -1 and the degrees of the binned diagonal entries: few values
    assert want_values in A1.describe().split("binned part")[1], A1.describe()
    assert ("the diagonal as a dense vector" in A1.describe()) == (values != "nodiag")
    L.destroy()
    M = csr_matrix((va, ci, rp), shape=(n, n))
    xd, yd, zd, wd = (ctx.vec_alloc(n) for _ in range(4))
    for trial in range(2):
        x = rng.standard_normal(n)
        ctx.upload(xd, x)
        ctx.upload(zd, np.full(n, np.nan))
        A0.mult(xd, yd)
        A1.mult(xd, zd)
        y0, y1 = ctx.download(yd, n), ctx.download(zd, n)
        ref = M @ x
        scale = np.abs(M) @ np.abs(x) + 1e-300
        assert np.all(np.abs(y1 - ref) <= 1e-14 * scale) and np.all(np.abs(y1 - y0) <= 1e-14 * scale)
        ctx.upload(zd, np.full(n, np.nan))
        A1.mult(xd, zd)
        assert np.array_equal(ctx.download(zd, n), y1)        # the additions come in a fixed order
    if values in ("laplacian", "zeros", "nodiag"):
        # integer-valued data: every product and partial sum is exact, so the re-ordered sums of the
        # binned form must equal the sequential CSR loop bit for bit -- any entry lost, doubled or
        # attributed to the wrong row or column shows
        xi = (np.arange(n, dtype=np.int64) * 2654435761 % 1021 - 510).astype(np.float64)
        ctx.upload(xd, xi)
        ctx.upload(zd, np.full(n, np.nan))
        A1.mult(xd, zd)
        want = np.zeros(n)
        from oracle.matgen import csr_matvec_fast
        csr_matvec_fast(rp, ci, va, xi, want)
        assert np.array_equal(ctx.download(zd, n), want)
        ctx.upload(xd, x)
    r1, c1, v1 = A1.get_csr()
    assert np.array_equal(r1, rp) and np.array_equal(c1, ci) and np.array_equal(v1, va)
    T0, T1 = A0.transpose(), A1.transpose()
    for a, b in zip(T0.get_csr(), T1.get_csr()):
        assert np.array_equal(a, b)
    want = np.linalg.norm(ref - 1.75 * x)
    assert abs(A1.residual_norm(1.75, 0.0, xd, None, wd) - want) <= 1e-13 * want
    with pytest.raises(ksgpu.KsgError):
        A1.residual_norm(1.75, 0.0, xd)                    # the binned part needs a work vector
    m = 3
    V, Wb = ksgpu.DeviceBV(ctx, n, m), ksgpu.DeviceBV(ctx, n, m)
    Va = rng.standard_normal((n, m))
    V.from_numpy(Va)
    V.matmult(A1, Wb)
    Wa = Wb.to_numpy()
    for j in range(m):
        assert np.all(np.abs(Wa[:, j] - M @ Va[:, j]) <= 1e-14 * (np.abs(M) @ np.abs(Va[:, j]) + 1e-300))
    ex, cm = A1.traffic_model()
    assert ex > 0 and cm > 0
    for o in (A0, A1, T0, T1, V, Wb):
        o.destroy()
    for p_ in (xd, yd, zd, wd):
        ctx.vec_free(p_)


@pytest.mark.parametrize("values", ["one", "few", "any"])
def test_binned_product_whole_steps_integer_exact(gpu_ctx, values):
    """The binned product with the library's own task sizes on an operator large enough that the
    expand pass runs its main loop (two whole steps of 16 waves x 8 groups in flight per workgroup:
    tasks of several hundred groups), for the three ways the values are held: on integer-valued data
    every product and partial sum is exact, so the result must equal the sequential CSR loop bit for
    bit, and the rows read back must be the input's."""
    from oracle.matgen import csr_matvec_fast
    from slepc_amd import ksgpu, matgen as mg
    ctx = gpu_ctx
    rp, ci, va = mg.rmat_laplacian(20)
    n = len(rp) - 1
    rng = np.random.default_rng(99)
    offd = ci != np.repeat(np.arange(n), np.diff(rp))
    va = va.copy()
    if values == "few":
        va[offd] = rng.choice(np.array([-1.0, -2.0, 3.0, 5.0]), size=int(offd.sum()))
    elif values == "any":
        va[offd] = rng.integers(-1000, 1001, size=int(offd.sum())).astype(np.float64)
    try:
        ctx.set_option(ksgpu.OPT_BINS, 2)
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
    finally:
        ctx.set_option(ksgpu.OPT_BINS, 0)
    d = A.describe()
    assert "binned part" in d, d
    assert {"one": "values: one", "few": "1-byte indices", "any": "8-byte stream"}[values] in d.split("binned part")[1], d
    xi = (np.arange(n, dtype=np.int64) * 2654435761 % 1021 - 510).astype(np.float64)
    want = np.zeros(n)
    csr_matvec_fast(rp, ci, va, xi, want)
    xd, yd = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.upload(xd, xi)
    for _ in range(2):
        ctx.upload(yd, np.full(n, np.nan))
        A.mult(xd, yd)
        assert np.array_equal(ctx.download(yd, n), want)
    r1, c1, v1 = A.get_csr()
    assert np.array_equal(r1, rp) and np.array_equal(c1, ci) and np.array_equal(v1, va)
    A.destroy()
    ctx.vec_free(xd)
    ctx.vec_free(yd)


def test_binned_random_blocks(gpu_ctx):
    """Random small operators with skewed columns, hub rows, empty rows, with and without a complete
    diagonal, square and rectangular, under random settings of the tuning variables: whatever form
    the library chooses under KSG_OPT_BINS = always, the product is A x and the rows read back are
    the input's."""
    import os
    from scipy.sparse import csr_matrix
    from slepc_amd import ksgpu
    ctx = gpu_ctx
    rng = np.random.default_rng(4242)
    keys = ("KSG_BINS_W", "KSG_BINS_H", "KSG_BINS_E", "KSG_BINS_T", "KSG_BINS_TASK_GROUPS")
    old = {k: os.environ.get(k) for k in keys}
    binned = 0
    try:
        ctx.set_option(ksgpu.OPT_BINS, 2)
        for trial in range(40):
            n = int(rng.integers(64, 3000))
            nc = n if trial % 3 else int(rng.integers(64, 4000))
            nnz_t = int(rng.choice([2.0, 6.0, 20.0]) * n)
            rows = rng.integers(0, n, nnz_t)
            cols = np.minimum((nc * rng.random(nnz_t) ** rng.choice([2.0, 4.0, 6.0])).astype(np.int64), nc - 1)
            if trial % 4 == 0:
                extra = rng.integers(0, nc, int(rng.integers(200, 3000)))
                rows = np.concatenate([rows, np.full(len(extra), int(rng.integers(0, n)))])
                cols = np.concatenate([cols, extra])
            if nc == n and trial % 2:
                rows = np.concatenate([rows, np.arange(n)])
                cols = np.concatenate([cols, np.arange(n)])
            key = np.unique(rows * nc + cols)
            rows, cols = key // nc, (key % nc).astype(np.int32)
            kind = int(rng.integers(0, 3))
            vals = rng.standard_normal(len(key)) if kind == 0 else (np.full(len(key), -1.0) if kind == 1 else rng.choice(np.array([0.5, -2.0, 0.0]), len(key)))
            rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int64)
            os.environ.update({"KSG_BINS_W": str(rng.choice([64, 256, 1024])), "KSG_BINS_H": str(rng.choice([64, 100, 256])),
                               "KSG_BINS_E": str(rng.choice([40, 200, 5000])), "KSG_BINS_T": str(rng.choice([1, 2, 3])),
                               "KSG_BINS_TASK_GROUPS": str(rng.choice([1, 2, 9]))})
            A = ksgpu.DeviceCSR(ctx, rp, cols, vals, cols=None if nc == n else (nc, nc, 0))
            binned += "binned part" in A.describe()
            x = rng.standard_normal(nc)
            xd, yd = ctx.vec_alloc(nc), ctx.vec_alloc(n)
            ctx.upload(xd, x)
            ctx.upload(yd, np.full(n, np.nan))
            A.mult(xd, yd)
            y = ctx.download(yd, n)
            M = csr_matrix((vals, cols, rp), shape=(n, nc))
            assert np.all(np.abs(y - M @ x) <= 1e-13 * (np.abs(M) @ np.abs(x) + 1e-300)), (trial, A.describe())
            r1, c1, v1 = A.get_csr()
            assert np.array_equal(r1, rp) and np.array_equal(c1, cols) and np.array_equal(v1, vals), (trial, A.describe())
            A.destroy()
            ctx.vec_free(xd)
            ctx.vec_free(yd)
    finally:
        ctx.set_option(ksgpu.OPT_BINS, 0)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert binned >= 10, binned


def test_binned_random_large_blocks(gpu_ctx):
    """Random operators of 10^5 .. 10^6 rows and 10^6 .. 10^7 entries with skewed columns, hub rows
    (some longer than a row block may be: cut into pieces), empty rows, square and rectangular, integer
    values of the three kinds, under the library's own task sizes and some odd ones: with integer x the
    binned product must equal the sequential CSR loop bit for bit (every workgroup of the expand pass runs
    whole double steps, single steps and a ragged end here), and the rows read back are the input's."""
    import os
    from oracle.matgen import csr_matvec_fast
    from slepc_amd import ksgpu
    ctx = gpu_ctx
    rng = np.random.default_rng(20261)
    keys = ("KSG_BINS_W", "KSG_BINS_H", "KSG_BINS_E", "KSG_BINS_T", "KSG_BINS_TASK_GROUPS")
    old = {k: os.environ.get(k) for k in keys}
    binned = 0
    try:
        ctx.set_option(ksgpu.OPT_BINS, 2)
        for trial in range(6):
            n = int(rng.integers(100_000, 1_000_000))
            nc = n if trial % 3 else int(rng.integers(100_000, 1_500_000))
            nnz_t = int(rng.choice([6.0, 12.0, 20.0]) * n)
            rows = np.minimum((n * rng.random(nnz_t) ** rng.choice([1.0, 2.0, 3.0])).astype(np.int64), n - 1)
            cols = np.minimum((nc * rng.random(nnz_t) ** rng.choice([3.0, 4.0, 6.0])).astype(np.int64), nc - 1)
            hubs = rng.integers(0, n, 3)                                   # three hub rows, the first very long
            for h, length in zip(hubs, (int(min(nc, 400_000)), 60_000, 5_000)):
                extra = rng.choice(nc, size=length, replace=False)
                rows = np.concatenate([rows, np.full(length, h)])
                cols = np.concatenate([cols, extra])
            if nc == n and trial % 2:
                rows = np.concatenate([rows, np.arange(n)])
                cols = np.concatenate([cols, np.arange(n)])
            key = np.unique(rows * nc + cols)
            rows, cols = key // nc, (key % nc).astype(np.int32)
            kind = trial % 3
            vals = (rng.integers(-1000, 1001, len(key)).astype(np.float64) if kind == 0 else
                    np.full(len(key), -1.0) if kind == 1 else rng.choice(np.array([3.0, -2.0, 0.0, 7.0]), len(key)))
            rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int64)
            for k in keys:
                os.environ.pop(k, None)
            if trial >= 3:                                                 # odd shapes: short tasks, low rows blocks, small E (more pieces)
                os.environ.update({"KSG_BINS_W": str(rng.choice([4096, 16384])), "KSG_BINS_H": str(rng.choice([1000, 4096])),
                                   "KSG_BINS_E": str(rng.choice([30_000, 100_000])), "KSG_BINS_T": str(rng.choice([1, 2, 4])),
                                   "KSG_BINS_TASK_GROUPS": str(rng.choice([129, 300, 777]))})
            A = ksgpu.DeviceCSR(ctx, rp, cols, vals, cols=None if nc == n else (nc, nc, 0))
            binned += "binned part" in A.describe()              # the library keeps a block whose columns are not skewed enough as it is
            xi = (np.arange(nc, dtype=np.int64) * 2654435761 % 1021 - 510).astype(np.float64)
            want = np.zeros(n)
            csr_matvec_fast(rp, cols, vals, xi, want)
            xd, yd = ctx.vec_alloc(nc), ctx.vec_alloc(n)
            ctx.upload(xd, xi)
            ctx.upload(yd, np.full(n, np.nan))
            A.mult(xd, yd)
            got = ctx.download(yd, n)
            assert np.array_equal(got, want), (trial, int((got != want).sum()), A.describe())
            r1, c1, v1 = A.get_csr()
            assert np.array_equal(r1, rp) and np.array_equal(c1, cols) and np.array_equal(v1, vals), (trial, A.describe())
            A.destroy()
            ctx.vec_free(xd)
            ctx.vec_free(yd)
    finally:
        ctx.set_option(ksgpu.OPT_BINS, 0)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert binned >= 4, binned


@pytest.mark.parametrize("gen", ["lap1d", "lap2d", "lap3d", "band"])
def test_row_code_kernel_equals_tile_kernel(gpu_ctx, gen):
    """Stencils and other banded operators with few distinct (column - row, value) pairs are
    multiplied by the row-code kernel (one byte per entry, coalesced x loads); the result is bit
    for bit the tile kernel's and the sequential CSR loop's, whatever the XCD chunking; the
    store-free residual agrees to rounding; operators without that form keep the tile kernel."""
    import os
    from slepc_amd import ksgpu, matgen as mg
    ctx = gpu_ctx
    rng = np.random.default_rng(21)
    if gen == "band":
        n = 20011
        offs = np.array([-700, -7, -1, 0, 2, 9, 513])
        pool = np.array([0.5, -1.25, 3.0, -0.0, 7.5])
        rows, cols, vals = [], [], []
        for r in range(n):
            keep = offs[(r + offs >= 0) & (r + offs < n) & (rng.random(offs.size) < 0.92)]
            rows += [r] * keep.size
            cols += list(r + keep)
            vals += list(pool[(r + keep) % pool.size])
        counts = np.bincount(rows, minlength=n)
        rp = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        ci, va = np.array(cols, dtype=np.int32), np.array(vals)
    else:
        rp, ci, va = {"lap1d": lambda: mg.laplacian1d(50021), "lap2d": lambda: mg.laplacian2d(301, 217),
                      "lap3d": lambda: mg.laplacian3d(41, 37, 29)}[gen]()
    n = len(rp) - 1
    x = rng.standard_normal(n)
    xd, yd = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.upload(xd, x)
    results = {}
    for key, env in (("tiles", {"KSG_SPMV_ROWCODE": "0"}), ("codes", {}), ("codes_c8", {"KSG_SPMV_ROWCODE_CHUNK": "8"}),
                     ("codes_c1", {"KSG_SPMV_ROWCODE_CHUNK": "1"})):
        for k in ("KSG_SPMV_ROWCODE", "KSG_SPMV_ROWCODE_CHUNK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
        for k in env:
            os.environ.pop(k, None)
        assert ("row codes" in A.describe()) == (key != "tiles"), A.describe()
        ctx.upload(yd, np.full(n, np.nan))
        A.mult(xd, yd)
        results[key] = (ctx.download(yd, n), A.residual_norm(1.5, 0.0, xd), A.traffic_model())
        r2, c2, v2 = A.get_csr()
        assert np.array_equal(c2, ci) and np.array_equal(v2.view(np.uint64), va.view(np.uint64))
        A.destroy()
    ref = og.csr_matvec_fast(rp, ci, va, x, np.zeros(n))
    for key in ("codes", "codes_c8", "codes_c1"):
        assert np.array_equal(results[key][0], ref)
        assert abs(results[key][1] - results["tiles"][1]) <= 1e-13 * results["tiles"][1]
        assert results[key][2][0] < results["tiles"][2][0]             # fewer bytes by the model, the same CSR figure
        assert results[key][2][1] == results["tiles"][2][1]
    if info_grouped := (np.abs(results["tiles"][0] - ref).max() > 0):
        assert np.abs(results["tiles"][0] - ref).max() <= 1e-13 * np.abs(ref).max()
    # a nonsymmetric chain with hundreds of distinct values has no row-code form
    M = ksgpu.DeviceCSR(ctx, *mg.markov(120))
    assert "row tiles" in M.describe()
    M.destroy()
    ctx.vec_free(xd)
    ctx.vec_free(yd)


@pytest.mark.parametrize("case", ["stencil", "rmat_plain", "rmat_binned", "random_values"])
def test_setup_same_for_any_host_thread_count(gpu_ctx, case):
    """The set-up of an operator (validation, column counts, packed columns, value tables, row codes,
    the binned build) runs its passes on several host threads from a million entries on
    (KSG_HOST_THREADS): the operator described, the entries read back and every bit of the product
    are the same for 1 and 7 threads, and when the power-law test starts from a sample of the entries
    (as it does for very large blocks); a column index out of range is reported as with one thread."""
    import os
    from slepc_amd import ksgpu, matgen as mg
    ctx = gpu_ctx
    rng = np.random.default_rng(3)
    if case == "stencil":
        rp, ci, va = mg.laplacian2d(600, 600)
    else:
        rp, ci, va = mg.rmat_laplacian(16, 10, seed=4)
        if case == "random_values":
            va = rng.standard_normal(len(va))
    n = len(rp) - 1
    assert rp[-1] > 1 << 20
    x = rng.standard_normal(n)
    xd, yd = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.upload(xd, x)
    old = os.environ.get("KSG_HOST_THREADS")
    got = []
    try:
        ctx.set_option(ksgpu.OPT_BINS, 2 if case in ("rmat_binned", "random_values") else 1)
        for nt, sample_from in ((1, None), (7, None), (7, 40000)):
            os.environ["KSG_HOST_THREADS"] = str(nt)
            # the power-law test of very large blocks starts from a sample of the entries: here on this size
            os.environ.pop("KSG_SAMPLE_MIN_NNZ", None)
            if sample_from:
                os.environ["KSG_SAMPLE_MIN_NNZ"] = str(sample_from)
            A = ksgpu.DeviceCSR(ctx, rp, ci, va)
            ctx.upload(yd, np.full(n, np.nan))
            A.mult(xd, yd)
            got.append((A.describe(), ctx.download(yd, n), A.get_csr()))
            A.destroy()
        bad = ci.copy()
        bad[len(bad) // 2 + 11] = n
        bad[len(bad) - 5] = -1
        with pytest.raises(ksgpu.KsgError, match="out of range at entry %d" % (len(bad) // 2 + 11)):
            ksgpu.DeviceCSR(ctx, rp, bad, va)
    finally:
        ctx.set_option(ksgpu.OPT_BINS, 0)
        os.environ.pop("KSG_SAMPLE_MIN_NNZ", None)
        if old is None:
            os.environ.pop("KSG_HOST_THREADS", None)
        else:
            os.environ["KSG_HOST_THREADS"] = old
    assert got[0][0] == got[1][0] == got[2][0], (got[0][0], got[1][0], got[2][0])
    assert np.array_equal(got[0][1], got[2][1])
    assert {"stencil": "row codes", "rmat_plain": "packed hot columns", "rmat_binned": "binned part",
            "random_values": "binned part"}[case] in got[0][0], got[0][0]
    assert np.array_equal(got[0][1], got[1][1])
    for a, b in zip(got[0][2], got[1][2]):
        assert np.array_equal(a, b)
    assert np.array_equal(got[0][2][1], ci) and np.array_equal(got[0][2][2], va)
    ctx.vec_free(xd)
    ctx.vec_free(yd)


