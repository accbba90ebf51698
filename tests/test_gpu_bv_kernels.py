"""Parity of the HIP BV kernels (through the C ABI) against the CPU oracle.

Floating-point bar: every result within 1e-12 relative to the magnitude of the
data (different summation order than BLAS, fp64 throughout); integer-valued
inputs must come out exact.
"""
import numpy as np
import pytest

from oracle import bv as obv

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _mk(ctx, n, m, seed=0):
    from slepc_amd import ksgpu
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, m))
    G = ksgpu.DeviceBV(ctx, n, m)
    G.from_numpy(A)
    O = obv.BV(n, m)
    O.array[:, :] = A
    return G, O, rng


def _close(a, b, scale=None):
    a, b = np.asarray(a), np.asarray(b)
    s = scale if scale is not None else max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    return np.abs(a - b).max() <= RTOL * s if a.size else True


@pytest.mark.parametrize("n,m,l,k", [(1000, 5, 0, 5), (4097, 33, 0, 33), (100003, 65, 3, 65),
                                      (1, 3, 0, 3), (63, 17, 2, 9), (200000, 20, 0, 17), (7, 1, 0, 1)])
def test_dotvec(gpu_ctx, n, m, l, k):
    G, O, rng = _mk(gpu_ctx, n, m)
    y = rng.standard_normal(n)
    yd = gpu_ctx.vec_alloc(n)
    gpu_ctx.upload(yd, y)
    G.set_active(l, k)
    O.set_active_columns(l, k)
    q = G.dotvec(yd)
    qo = np.zeros(k - l)
    O.op_dotvec(y, qo)
    assert _close(q, qo, scale=np.sqrt(n) * 10)
    # a column of the BV itself as y (the Gram-Schmidt case)
    q2 = G.dotvec(G.column(k - 1))
    O.op_dotvec(O.col(k - 1), qo)
    assert _close(q2, qo, scale=n)
    gpu_ctx.vec_free(yd)
    G.destroy()


@pytest.mark.parametrize("n,m,l,k,alpha,beta", [(1000, 5, 0, 5, -1.0, 1.0), (4097, 33, 0, 32, 1.0, 0.0),
                                                 (100003, 65, 3, 64, -1.0, 1.0), (1, 3, 0, 3, 2.0, 0.5),
                                                 (50001, 140, 0, 139, -0.5, 2.0), (63, 4, 1, 1, 3.0, 2.0)])
def test_multvec(gpu_ctx, n, m, l, k, alpha, beta):
    G, O, rng = _mk(gpu_ctx, n, m, seed=1)
    q = rng.standard_normal(max(k - l, 1))
    G.set_active(l, k)
    O.set_active_columns(l, k)
    j = m - 1
    ycol = O.col(j).copy()
    if beta == 0.0:
        G.set_column(j, np.full(n, np.nan))     # beta = 0 must not read y
    G.multvec(alpha, beta, G.column(j), q)
    O.op_multvec(alpha, beta, ycol, q[:k - l])
    assert _close(G.get_column(j), ycol, scale=10 * np.sqrt(max(k - l, 1)))
    G.destroy()


@pytest.mark.parametrize("n,m,l,k,s,e", [(1000, 5, 0, 5, 1, 3), (4097, 65, 0, 64, 0, 33), (100001, 65, 10, 64, 10, 40),
                                          (33, 80, 0, 80, 0, 80), (2, 9, 2, 8, 3, 8), (5000, 150, 5, 140, 5, 100),
                                          # one launch of the matrix-core kernel, in place: 33..64 output columns,
                                          # input widths that are no multiple of 4, row counts that leave ragged tiles
                                          (100003, 65, 0, 64, 0, 64), (70001, 66, 1, 64, 1, 50), (4160, 40, 3, 38, 3, 38),
                                          (131, 64, 0, 61, 0, 47), (64, 20, 0, 17, 2, 17), (1, 7, 0, 7, 0, 7)])
@pytest.mark.parametrize("trans", [False, True])
def test_multinplace(gpu_ctx, n, m, l, k, s, e, trans):
    G, O, rng = _mk(gpu_ctx, n, m, seed=2)
    Q = np.asfortranarray(rng.standard_normal((m, m)))
    G.set_active(l, k)
    O.set_active_columns(l, k)
    G.multinplace(Q, s, e, trans)
    O.op_multinplace(Q, s, e, trans)
    assert _close(G.to_numpy(), O.array, scale=10 * np.sqrt(k - l))
    G.destroy()


def test_scale_norm_copy(gpu_ctx):
    n, m = 30001, 9
    G, O, rng = _mk(gpu_ctx, n, m, seed=3)
    G.set_active(2, 7)
    O.set_active_columns(2, 7)
    for nt in (obv.NORM_1, obv.NORM_2, obv.NORM_FROBENIUS, obv.NORM_INFINITY):
        assert abs(G.norm(-1, nt) - O.op_norm(-1, nt)) <= 1e-12 * O.op_norm(-1, nt)
        assert abs(G.norm(4, nt) - O.op_norm(4, nt)) <= 1e-12 * O.op_norm(4, nt)
    G.scale(3, -2.5)
    O.op_scale(3, -2.5)
    G.scale(-1, 0.5)
    O.op_scale(-1, 0.5)
    G.scale(8, 0.0)
    O.op_scale(8, 0.0)
    assert np.array_equal(G.to_numpy(), O.array)          # scaling is exact in fp64
    G.copy_column(1, 6)
    O.copy_column(1, 6)
    assert np.array_equal(G.to_numpy(), O.array)
    G.destroy()


def test_golden_bv_test1(gpu_ctx):
    """The reference's bv test1 (n=10, k=5, l=3) on the device: exact small-integer
    results of BVMult, BVMultVec, BVDot, BVDotVec, BVMultInPlace+BVScale, norms."""
    import json, os
    from slepc_amd import ksgpu
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bv_test1.json")))
    n, k, l = 10, 5, 3
    X = ksgpu.DeviceBV(gpu_ctx, n, k)
    Y = ksgpu.DeviceBV(gpu_ctx, n, l)
    Xh = np.zeros((n, k))
    for j in range(k):
        for i in range(4):
            if i + j < n:
                Xh[i + j, j] = 3 * i + j - 2
    X.from_numpy(Xh)
    Y.from_numpy(np.stack([np.full(n, (j + 1) / 4.0) for j in range(l)], axis=1))
    Q = np.asfortranarray(np.array([[2.0 if i < j else -0.5 for j in range(l)] for i in range(k)]))
    Y.mult(2.0, 1.0, X, Q)
    assert np.array_equal(Y.to_numpy(), np.array(g["Y_after_mult"]).reshape(l, n).T)
    z = np.array([2.0 * (-0.5) ** i for i in range(k)])
    X.multvec(-1.0, 1.0, Y.column(0), z)
    assert np.array_equal(Y.to_numpy(), np.array(g["Y_after_multvec"]).reshape(l, n).T)
    M = np.zeros((l, k), order="F")
    X.dot(Y, M)
    assert np.array_equal(M, np.array(g["M"]))
    assert np.array_equal(X.dotvec(Y.column(0)), np.array(g["z"]))
    X.set_active(0, k)
    X.multinplace(Q, 1, l)
    X.scale(-1, 2.0)
    assert np.array_equal(X.to_numpy(), np.array(g["X_after_multinplace"]).reshape(k, n).T)
    assert "%g" % X.norm(0, obv.NORM_2) == g["norm_X0"]
    assert "%g" % X.norm(-1, obv.NORM_FROBENIUS) == g["norm_fro"]
    X.destroy()
    Y.destroy()


def test_bv_mult_dot_general(gpu_ctx):
    from slepc_amd import ksgpu
    n = 20011
    rng = np.random.default_rng(5)
    Xa, Ya = rng.standard_normal((n, 12)), rng.standard_normal((n, 7))
    X, Y = ksgpu.DeviceBV(gpu_ctx, n, 12), ksgpu.DeviceBV(gpu_ctx, n, 7)
    X.from_numpy(Xa)
    Y.from_numpy(Ya)
    OX, OY = obv.BV(n, 12), obv.BV(n, 7)
    OX.array[:] = Xa
    OY.array[:] = Ya
    X.set_active(2, 11)
    Y.set_active(1, 6)
    OX.set_active_columns(2, 11)
    OY.set_active_columns(1, 6)
    Q = np.asfortranarray(rng.standard_normal((12, 7)))
    Y.mult(0.5, -1.5, X, Q)
    OY.op_mult(0.5, -1.5, OX, Q)
    assert _close(Y.to_numpy(), OY.array, scale=50)
    M = np.zeros((7, 12), order="F")
    Mo = np.zeros((7, 12), order="F")
    X.dot(Y, M)
    OX.op_dot(OY, Mo)
    assert _close(M, Mo, scale=n)
    X.destroy()
    Y.destroy()


@pytest.mark.parametrize("n,m,k", [(1000, 5, 4), (4097, 34, 33), (100003, 65, 64), (2, 3, 2), (1, 3, 2),
                                    (63, 18, 9), (50001, 130, 128), (200000, 101, 100), (30000, 140, 139)])
@pytest.mark.parametrize("lazy", [0, 1])
def test_multvec_fused_dots(gpu_ctx, n, m, k, lazy):
    """The Gram-Schmidt update of column k by columns [0,k) also yields the next
    pass's reduction; the dotvec that follows must return it (same values as
    a fresh reduction, to rounding) and anything that writes in between must
    invalidate it.  lazy=0: the update is launched at once and leaves the products
    behind; lazy=1: the update waits and the request for the products launches it."""
    from slepc_amd import ksgpu
    gpu_ctx.set_option(ksgpu.OPT_LAZY_UPDATE, lazy)
    try:
        _fused_dots_case(gpu_ctx, n, m, k, lazy)
    finally:
        gpu_ctx.set_option(ksgpu.OPT_LAZY_UPDATE, 1)


def _fused_dots_case(gpu_ctx, n, m, k, lazy):
    G, O, rng = _mk(gpu_ctx, n, m, seed=3)
    q = rng.standard_normal(k) * 0.1
    G.set_active(0, k)
    O.set_active_columns(0, k)
    ycol = O.col(k)
    gpu_ctx.stats_reset()
    G.multvec(-1.0, 1.0, G.column(k), q)
    O.op_multvec(-1.0, 1.0, ycol, q)
    if not lazy:
        assert _close(G.get_column(k), ycol, scale=10 * np.sqrt(k))
    G.set_active(0, k + 1)
    O.set_active_columns(0, k + 1)
    got = G.dotvec(G.column(k))
    want = np.zeros(k + 1)
    O.op_dotvec(ycol, want)
    st = gpu_ctx.stats()
    assert _close(G.get_column(k), ycol, scale=10 * np.sqrt(k))
    fused = k <= 128                         # beyond 64 columns also a lazy update is launched at once
    assert st["dotvec_fused"]["calls"] == (1 if fused else 0)
    assert st["dotvec"]["calls"] == (0 if fused else 1)
    assert _close(got, want, scale=n)
    # a second identical request is still valid
    assert _close(G.dotvec(G.column(k)), want, scale=n)
    # different active range: a real reduction
    G.set_active(1 if k > 1 else 0, k + 1)
    O.set_active_columns(1 if k > 1 else 0, k + 1)
    want2 = np.zeros(k + 1 - (1 if k > 1 else 0))
    O.op_dotvec(ycol, want2)
    assert _close(G.dotvec(G.column(k)), want2, scale=n)
    # a write in between invalidates
    G.set_active(0, k)
    O.set_active_columns(0, k)
    G.multvec(-1.0, 1.0, G.column(k), q)
    O.op_multvec(-1.0, 1.0, ycol, q)
    G.scale(k, 0.5)
    ycol *= 0.5
    G.set_active(0, k + 1)
    O.set_active_columns(0, k + 1)
    before = gpu_ctx.stats()["dotvec_fused"]["calls"]
    O.op_dotvec(ycol, want)
    assert _close(G.dotvec(G.column(k)), want, scale=n)
    assert gpu_ctx.stats()["dotvec_fused"]["calls"] == before
    # a foreign write announced with touch() invalidates too
    G.set_active(0, k)
    G.multvec(-1.0, 1.0, G.column(k), q)
    G.touch()
    G.set_active(0, k + 1)
    G.dotvec(G.column(k))
    assert gpu_ctx.stats()["dotvec_fused"]["calls"] == before
    G.destroy()


def test_fused_dots_option_off(gpu_ctx):
    from slepc_amd import ksgpu
    G, O, rng = _mk(gpu_ctx, 5000, 9, seed=4)
    q = rng.standard_normal(8)
    gpu_ctx.set_option(ksgpu.OPT_FUSE_DOTS, 0)
    try:
        gpu_ctx.stats_reset()
        G.set_active(0, 8)
        G.multvec(-1.0, 1.0, G.column(8), q)
        G.set_active(0, 9)
        G.dotvec(G.column(8))
        st = gpu_ctx.stats()
        assert st["dotvec_fused"]["calls"] == 0 and st["dotvec"]["calls"] == 1
    finally:
        gpu_ctx.set_option(ksgpu.OPT_FUSE_DOTS, 1)
    G.destroy()


def _cgs2_columns(ctx, A, lazy, passes=2, peek=None, estimate=False):
    """Classical Gram-Schmidt with `passes` passes per column, driven through the C ABI the
    way BVOrthogonalizeGS drives the op table; returns the basis and the statistics."""
    from slepc_amd import ksgpu
    n, m = A.shape
    ctx.set_option(ksgpu.OPT_LAZY_UPDATE, 1 if lazy else 0)
    G = ksgpu.DeviceBV(ctx, n, m)
    G.from_numpy(A)
    ctx.stats_reset()
    seen = {}
    try:
        G.scale(0, 1.0 / np.sqrt(G.dotvec_cols(0, 1, 0)[0]))
        for j in range(1, m):
            for p in range(passes):
                c = G.dotvec_cols(0, j + 1, j)
                G.set_active(0, j)
                G.multvec(-1.0, 1.0, G.column(j), c[:j])
                if peek is not None and p == 0 and j == peek:
                    seen[j] = G.get_column(j)          # forces the deferred update out
            if estimate:
                # norm from the coefficients of the last pass, as bvorthog.c:268-273 does:
                # the scaling follows the update directly
                G.scale(j, 1.0 / np.sqrt(c[j] - np.sum(c[:j] ** 2)))
            else:
                G.touch()          # norm from a fresh reduction in every mode (same order of sums)
                c = G.dotvec_cols(0, j + 1, j)
                G.scale(j, 1.0 / np.sqrt(c[j]))
        out = G.to_numpy()
        st = ctx.stats()
    finally:
        ctx.set_option(ksgpu.OPT_LAZY_UPDATE, 1)
        G.destroy()
    return out, st, seen


@pytest.mark.parametrize("n,m", [(100003, 40), (4096, 66), (2, 2), (777, 5)])
def test_lazy_update_bitwise(gpu_ctx, n, m):
    """Deferring the store of the first update changes which kernel writes the column,
    never a bit of what is written."""
    rng = np.random.default_rng(11)
    A = rng.standard_normal((n, m))
    eager, st_e, _ = _cgs2_columns(gpu_ctx, A, lazy=False)
    lazy, st_l, _ = _cgs2_columns(gpu_ctx, A, lazy=True)
    assert np.array_equal(eager, lazy)
    # the reference's flow (estimated norm, scaling right after the second update): the two
    # updates and the scaling become one pass, still the same bits
    e2, st_e2, _ = _cgs2_columns(gpu_ctx, A, lazy=False, estimate=True)
    l2, st_l2, _ = _cgs2_columns(gpu_ctx, A, lazy=True, estimate=True)
    assert np.array_equal(e2, l2)
    if m > 3:
        assert st_l2["scale"]["bytes"] < st_e2["scale"]["bytes"]
        assert st_l2["scale"]["calls"] == st_e2["scale"]["calls"]
    # the basis is orthonormal to working accuracy
    if n > m:
        assert np.abs(lazy.T @ lazy - np.eye(m)).max() < 1e-12
    # from the third column on the first update of each column is deferred: fewer bytes
    if m > 3:
        assert st_l["multvec"]["bytes"] < st_e["multvec"]["bytes"]
        assert st_l["multvec"]["calls"] == st_e["multvec"]["calls"]
    assert st_l["dotvec_fused"]["calls"] == st_e["dotvec_fused"]["calls"] > 0


def test_lazy_update_flush_paths(gpu_ctx):
    rng = np.random.default_rng(12)
    A = rng.standard_normal((50001, 12))
    eager, _, seen_e = _cgs2_columns(gpu_ctx, A, lazy=False, peek=7)
    lazy, st, seen_l = _cgs2_columns(gpu_ctx, A, lazy=True, peek=7)
    assert np.array_equal(seen_e[7], seen_l[7])
    assert np.array_equal(eager, lazy)
    # single-pass Gram-Schmidt: the prediction turns deferral off after one miss
    e1, st_e1, _ = _cgs2_columns(gpu_ctx, A, lazy=False, passes=1)
    l1, st_l1, _ = _cgs2_columns(gpu_ctx, A, lazy=True, passes=1)
    assert np.array_equal(e1, l1)
    assert st_l1["multvec"]["calls"] == st_e1["multvec"]["calls"]
    # three passes per column: the third update finds two pending ones
    e3, _, _ = _cgs2_columns(gpu_ctx, A, lazy=False, passes=3, estimate=True)
    l3, _, _ = _cgs2_columns(gpu_ctx, A, lazy=True, passes=3, estimate=True)
    assert np.array_equal(e3, l3)


@pytest.mark.parametrize("n,mx,my,lx,kx,ly,ky,alpha,beta", [
    (30001, 70, 40, 3, 70, 0, 40, 1.0, 0.0),        # two output panels, ragged tiles
    (4096, 65, 65, 0, 65, 0, 65, -0.5, 1.0),        # wider than 64 both ways
    (1001, 9, 33, 1, 9, 2, 33, 2.0, -1.0),
    (2, 3, 2, 0, 3, 0, 2, 1.0, 0.5),
    (5000, 8, 8, 2, 2, 0, 8, 1.0, 0.0),              # no active columns of X: Y = beta*Y
    # the matrix-core product out of place: 33..64 output columns with and without beta, ragged
    # input widths and row tiles, and two panels of it (at most 64 input, more than 64 output columns)
    (20011, 64, 64, 0, 64, 0, 64, 1.0, 0.0),
    (4133, 40, 50, 3, 40, 1, 49, -1.5, 0.75),
    (10000, 30, 150, 0, 30, 0, 150, 1.0, -1.0),
    (65, 64, 100, 0, 61, 10, 100, 2.0, 0.0),
])
def test_bv_mult_dot_block_kernels(gpu_ctx, n, mx, my, lx, kx, ly, ky, alpha, beta):
    """BVMult / BVDot through the block kernels (restart kernel out of place with beta,
    register-tiled Gram kernel) against the oracle's GEMMs."""
    from slepc_amd import ksgpu
    rng = np.random.default_rng(21)
    Xa, Ya = rng.standard_normal((n, mx)), rng.standard_normal((n, my))
    X, Y = ksgpu.DeviceBV(gpu_ctx, n, mx), ksgpu.DeviceBV(gpu_ctx, n, my)
    X.from_numpy(Xa)
    Y.from_numpy(Ya)
    OX, OY = obv.BV(n, mx), obv.BV(n, my)
    OX.array[:] = Xa
    OY.array[:] = Ya
    for b, o, l, k in ((X, OX, lx, kx), (Y, OY, ly, ky)):
        b.set_active(l, k)
        o.set_active_columns(l, k)
    M = np.zeros((my, mx), order="F")
    Mo = np.zeros((my, mx), order="F")
    X.dot(Y, M)
    OX.op_dot(OY, Mo)
    assert _close(M, Mo, scale=n)
    Q = np.asfortranarray(rng.standard_normal((mx, my)))
    if beta == 0.0:
        nanY = Ya.copy()
        nanY[:, ly:ky] = np.nan                       # beta = 0 must not read Y
        Y.from_numpy(nanY)
    Y.mult(alpha, beta, X, Q)
    OY.op_mult(alpha, beta, OX, Q)
    assert _close(Y.to_numpy(), OY.array, scale=10 * np.sqrt(max(kx - lx, 1)))
    X.destroy()
    Y.destroy()


@pytest.mark.parametrize("n,k", [(4097, 9), (30000, 33), (5, 2)])
def test_unaligned_vector_fallback(gpu_ctx, n, k):
    """A vector that starts 8 bytes off a 16-byte boundary takes the scalar-load kernels."""
    G, O, rng = _mk(gpu_ctx, n, k, seed=9)
    base = gpu_ctx.vec_alloc(n + 1)
    yd = base + 8                                    # device address, 8 bytes further
    y = rng.standard_normal(n)
    gpu_ctx.upload(yd, y)
    q = G.dotvec(yd)
    qo = np.zeros(k)
    O.op_dotvec(y, qo)
    assert _close(q, qo, scale=np.sqrt(n) * 10)
    coef = rng.standard_normal(k)
    G.multvec(-0.5, 2.0, yd, coef)
    O.op_multvec(-0.5, 2.0, y, coef)
    assert _close(gpu_ctx.download(yd, n), y, scale=10 * np.sqrt(k))
    gpu_ctx.vec_free(base)
    G.destroy()


def test_empty_local_block(gpu_ctx):
    """A rank that owns no rows (more ranks than rows): every op is a no-op that still
    returns its zeros, so the collective call sequence stays the same on all ranks."""
    from slepc_amd import ksgpu
    G = ksgpu.DeviceBV(gpu_ctx, 0, 5, N=3)
    G.set_active(0, 4)
    y = G.column(4)
    assert np.array_equal(G.dotvec(y), np.zeros(4))
    G.multvec(-1.0, 1.0, y, np.ones(4))
    G.set_active(0, 5)
    assert np.array_equal(G.dotvec(y), np.zeros(5))
    assert G.norm(2, 1) == 0.0 and G.norm(-1, 2) == 0.0
    G.scale(1, 3.0)
    G.multinplace(np.asfortranarray(np.eye(5)), 0, 3)
    W = ksgpu.DeviceBV(gpu_ctx, 0, 5, N=3)
    G.copy_to(W)
    M = np.ones((5, 5), order="F")
    G.dot(W, M)
    assert np.array_equal(M, np.zeros((5, 5)))
    W.mult(1.0, 0.0, G, np.asfortranarray(np.eye(5)))
    G.destroy()
    W.destroy()


def test_posted_reductions(gpu_ctx):
    """One batch of posted reductions (the device side of BVDotVecBegin/End and
    VecDotBegin/End, bvglobal.c:190-310): the values arrive in posting order and are the
    ones the immediate calls return, bit for bit; counts are checked; an abandoned batch
    leaves the context usable."""
    import ctypes as C
    from slepc_amd import ksgpu
    lib = ksgpu.load()
    n, m = 50021, 12
    G, O, rng = _mk(gpu_ctx, n, m, seed=9)
    y = rng.standard_normal(n)
    yd = gpu_ctx.vec_alloc(n)
    gpu_ctx.upload(yd, y)
    G.set_active(2, 11)
    ref1 = G.dotvec(yd)
    ref2 = gpu_ctx.vec_dot(yd, yd, n)
    G.set_active(0, 5)
    ref3 = G.dotvec(G.column(7))
    off = [C.c_int(-1) for _ in range(3)]
    ksgpu.check(lib.ksg_red_open(gpu_ctx.h))
    G.set_active(2, 11)
    ksgpu.check(lib.ksg_red_post_dotvec(G.h, yd, C.byref(off[0])))
    ksgpu.check(lib.ksg_red_post_dot(gpu_ctx.h, yd, yd, n, C.byref(off[1])))
    G.set_active(0, 5)
    ksgpu.check(lib.ksg_red_post_dotvec(G.h, G.column(7), C.byref(off[2])))
    assert [o.value for o in off] == [0, 9, 10]
    out = (C.c_double * 15)()
    assert lib.ksg_red_close(gpu_ctx.h, out, 14) != 0          # 15 were posted; the batch stays open
    ksgpu.check(lib.ksg_red_close(gpu_ctx.h, out, 15))
    got = np.array(out[:])
    assert np.array_equal(got[:9], ref1) and got[9] == ref2 and np.array_equal(got[10:], ref3)
    assert lib.ksg_red_close(gpu_ctx.h, out, 0) != 0           # nothing open
    assert lib.ksg_red_post_dot(gpu_ctx.h, yd, yd, n, None) != 0
    # abandoned batch, then a fresh one
    ksgpu.check(lib.ksg_red_open(gpu_ctx.h))
    assert lib.ksg_red_open(gpu_ctx.h) != 0
    ksgpu.check(lib.ksg_red_post_dot(gpu_ctx.h, yd, yd, n, None))
    ksgpu.check(lib.ksg_red_close(gpu_ctx.h, None, -1))
    ksgpu.check(lib.ksg_red_open(gpu_ctx.h))
    ksgpu.check(lib.ksg_red_post_dot(gpu_ctx.h, yd, yd, n, C.byref(off[0])))
    ksgpu.check(lib.ksg_red_close(gpu_ctx.h, out, 1))
    assert off[0].value == 0 and out[0] == ref2
    gpu_ctx.vec_free(yd)
    G.destroy()


@pytest.mark.parametrize("n,mx,my,lx,kx,ly,ky", [(4096, 20, 13, 0, 20, 0, 13), (10007, 40, 33, 3, 37, 1, 33),
                                                  (50021, 70, 66, 0, 70, 2, 66), (17, 65, 65, 0, 65, 0, 65),
                                                  (3, 9, 12, 0, 9, 0, 12), (100003, 64, 64, 0, 64, 0, 64)])
def test_dot_matrix_core_kernel(gpu_ctx, n, mx, my, lx, kx, ly, ky):
    """BVDot with more than one register tile of results runs on the matrix cores
    (16/32/64-wide blocks, several blocks, row counts that are not multiples of 16):
    M(ly+i, lx+j) = Y_i^T X_j against the host, integer data exact; X with itself
    (only the upper blocks are computed, the rest mirrored) symmetric to the bit."""
    from slepc_amd import ksgpu
    rng = np.random.default_rng(n + mx)
    Xa, Ya = rng.standard_normal((n, mx)), rng.standard_normal((n, my))
    X, Y = ksgpu.DeviceBV(gpu_ctx, n, mx), ksgpu.DeviceBV(gpu_ctx, n, my)
    X.from_numpy(Xa)
    Y.from_numpy(Ya)
    X.set_active(lx, kx)
    Y.set_active(ly, ky)
    M = np.full((my, mx), np.nan, order="F")
    X.dot(Y, M)
    ref = Ya[:, ly:ky].T @ Xa[:, lx:kx]
    assert np.abs(M[ly:ky, lx:kx] - ref).max() <= 1e-12 * np.sqrt(n) * 10
    assert np.isnan(M[:ly, :]).all() and np.isnan(M[:, :lx]).all() and np.isnan(M[ky:, :]).all() and np.isnan(M[:, kx:]).all()
    S = np.zeros((mx, mx), order="F")
    X.dot(X, S)
    G = S[lx:kx, lx:kx]
    assert np.array_equal(G, G.T)
    assert np.abs(G - Xa[:, lx:kx].T @ Xa[:, lx:kx]).max() <= 1e-12 * np.sqrt(n) * 10
    # small integers: every product and sum is exact whatever the order
    Xi, Yi = rng.integers(-3, 4, (n, mx)).astype(float), rng.integers(-3, 4, (n, my)).astype(float)
    X.from_numpy(Xi)
    Y.from_numpy(Yi)
    X.dot(Y, M)
    assert np.array_equal(M[ly:ky, lx:kx], Yi[:, ly:ky].T @ Xi[:, lx:kx])
    X.destroy()
    Y.destroy()


def _qr_pos(A):
    """Householder QR of numpy with the diagonal of R made positive (the unique factorisation)."""
    Q, R = np.linalg.qr(A)
    s = np.where(np.diag(R) < 0, -1.0, 1.0)
    return Q * s, (R.T * s).T


@pytest.mark.parametrize("n,m,k,cond", [(20, 8, 8, 1.0), (1000, 5, 5, 1.0), (30011, 40, 40, 1.0), (100003, 65, 64, 1.0),
                                         (64, 64, 64, 1.0), (4097, 33, 17, 1e9), (200000, 16, 16, 1e10), (65, 3, 1, 1.0)])
def test_tsqr(gpu_ctx, n, m, k, cond):
    """ksg_bv_tsqr against numpy's Householder QR: Q orthogonal to a few eps whatever the
    conditioning (columns scaled over `cond` and mixed), V = Q R, R upper triangular with a
    positive diagonal, Q and R equal to the unique factorisation up to the conditioning; columns
    beyond k untouched."""
    from slepc_amd import ksgpu
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((n, m))
    if cond > 1.0:
        # singular values from 1 down to 1/cond
        U, _ = np.linalg.qr(rng.standard_normal((n, k)))
        W, _ = np.linalg.qr(rng.standard_normal((k, k)))
        A[:, :k] = (U * np.logspace(0, -np.log10(cond), k)) @ W.T
    G = ksgpu.DeviceBV(gpu_ctx, n, m)
    G.from_numpy(A)
    G.set_active(min(2, k - 1), k)                      # the leading index is ignored, as in the reference
    R = np.full((m, m), np.nan, order="F")
    G.tsqr(R)
    Q = G.to_numpy()
    eps = np.finfo(float).eps
    assert np.abs(Q[:, :k].T @ Q[:, :k] - np.eye(k)).sum(axis=0).max() < 100 * eps * max(1.0, np.sqrt(k))
    assert np.array_equal(Q[:, k:], A[:, k:])
    Rk = R[:k, :k]
    assert np.abs(np.tril(Rk, -1)).max() == 0.0 and (np.diag(Rk) > 0).all()
    assert np.isnan(R[k:, :]).all() and np.isnan(R[:, k:]).all()
    scale = np.abs(A[:, :k]).max()
    assert np.abs(Q[:, :k] @ Rk - A[:, :k]).max() < 50 * eps * scale * np.sqrt(k)
    Qo, Ro = _qr_pos(A[:, :k])
    assert np.abs(Rk - Ro).max() < 1e-13 * cond * np.abs(Ro).max() * k
    if cond == 1.0:
        assert np.abs(Q[:, :k] - Qo).max() < 1e-11
    G.destroy()


def test_tsqr_errors(gpu_ctx):
    """Numerically dependent columns, too many columns and constraints are refused; the basis
    stays usable."""
    from slepc_amd import ksgpu
    rng = np.random.default_rng(5)
    A = rng.standard_normal((500, 70))
    A[:, 3] = A[:, 1] - 2.0 * A[:, 2]
    G = ksgpu.DeviceBV(gpu_ctx, 500, 70)
    G.from_numpy(A)
    G.set_active(0, 6)
    with pytest.raises(ksgpu.KsgError):
        G.tsqr()
    G.set_active(0, 70)
    with pytest.raises(ksgpu.KsgError):
        G.tsqr()
    G.from_numpy(A)
    G.set_active(0, 3)
    G.tsqr()
    Q = G.to_numpy()[:, :3]
    assert np.abs(Q.T @ Q - np.eye(3)).max() < 1e-14
    G.destroy()
