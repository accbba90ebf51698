"""The allocation-free reduction (``ksg_bv_dotvec_noalloc``) against ``ksg_bv_dotvec_local``.

Both run the same kernels with the same launch geometry, so the bar is equality of
the bits.  Two calls in a row share the fixed device scratch array; the second one
is smaller, so stale partial sums of the first would show.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(ctx, n, m, l, k, seed):
    from slepc_amd import ksgpu
    rng = np.random.default_rng(seed)
    G = ksgpu.DeviceBV(ctx, n, m)
    G.from_numpy(rng.standard_normal((n, m)))
    G.set_active(l, k)
    yd = ctx.vec_alloc(max(n, 1))
    ctx.upload(yd, rng.standard_normal(max(n, 1)))
    return G, yd


def test_noalloc_matches_local_twice(gpu_ctx):
    ctx = gpu_ctx
    qd = ctx.vec_alloc(128)
    G1, y1 = _case(ctx, 300001, 65, 0, 65, 11)      # 65 columns, every row workgroup busy
    G2, y2 = _case(ctx, 4099, 20, 3, 12, 12)        # smaller: fewer columns, fewer partial sums
    want1 = G1.dotvec(y1, local=True)
    want2 = G2.dotvec(y2, local=True)
    ctx.vec_set(qd, np.nan, 128)
    G1.dotvec_noalloc(y1, qd)
    got1 = ctx.download(qd, 65)
    ctx.vec_set(qd, np.nan, 128)
    G2.dotvec_noalloc(y2, qd)
    got2 = ctx.download(qd, 128)
    assert np.array_equal(got1, want1)
    assert np.array_equal(got2[:9], want2)
    assert np.isnan(got2[9:]).all()                 # nothing written past the k-l results
    # and once more the large one: the array is reused, not consumed
    G1.dotvec_noalloc(y1, qd)
    assert np.array_equal(ctx.download(qd, 65), want1)
    # a column of the basis itself as y, right after an update of it is still pending
    # (the lazy Gram-Schmidt path): the pending update is applied first
    G1.set_active(0, 64)
    q = want1[:64] * 1e-3
    G1.multvec(-1.0, 1.0, G1.column(64), q)
    G1.dotvec_noalloc(G1.column(64), qd)
    got = ctx.download(qd, 64)
    ref = G1.dotvec(G1.column(64), local=True)
    assert np.array_equal(got, ref)
    for p in (y1, y2, qd):
        ctx.vec_free(p)
    G1.destroy()
    G2.destroy()


def test_noalloc_limits(gpu_ctx):
    from slepc_amd import ksgpu
    ctx = gpu_ctx
    G, yd = _case(ctx, 70000, 140, 0, 140, 13)
    qd = ctx.vec_alloc(256)
    # 140 columns x 256 partial sums fit (35840 doubles); 16x the row workgroups do not
    G.dotvec_noalloc(yd, qd)
    assert np.array_equal(ctx.download(qd, 140), G.dotvec(yd, local=True))
    ctx.set_option(ksgpu.OPT_DOT_GRID, 4096)
    G2, y2 = _case(ctx, 1200000, 65, 0, 65, 14)
    with pytest.raises(ksgpu.KsgError) as ei:
        G2.dotvec_noalloc(y2, qd)
    assert ei.value.code == 5 and "scratch" in str(ei.value)
    ctx.set_option(ksgpu.OPT_DOT_GRID, 256)
    G2.dotvec_noalloc(y2, qd)
    assert np.array_equal(ctx.download(qd, 65), G2.dotvec(y2, local=True))
    # empty active range and empty vectors
    G.set_active(5, 5)
    G.dotvec_noalloc(yd, qd)
    for p in (yd, y2, qd):
        ctx.vec_free(p)
    G.destroy()
    G2.destroy()
