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


def test_noalloc_under_stream_capture(gpu_ctx):
    """The entry point inside hipStreamBeginCapture ... hipStreamEndCapture on the library's stream, in the
    global capture mode in which any hipMalloc (from any thread) is an error that ends the capture: the
    call is captured, the graph replays it, and the result is X^T y -- against numpy to rounding and
    against the allocating path bit for bit.  Free device memory is the same before and after."""
    import ctypes as C
    from slepc_amd import ksgpu
    ctx = gpu_ctx
    hip = C.CDLL("libamdhip64.so")
    n, m = 200003, 40
    rng = np.random.default_rng(21)
    Xa, ya = rng.standard_normal((n, m)), rng.standard_normal(n)
    G = ksgpu.DeviceBV(ctx, n, m)
    G.from_numpy(Xa)
    G.set_active(2, 37)
    yd, qd = ctx.vec_alloc(n), ctx.vec_alloc(64)
    ctx.upload(yd, ya)
    want = G.dotvec(yd, local=True)
    ctx.vec_set(qd, np.nan, 64)
    ctx.sync()
    stream = C.c_void_p(ctx.lib.ksg_ctx_stream(ctx.h))

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    before = free_bytes()
    assert hip.hipStreamBeginCapture(stream, 0) == 0          # hipStreamCaptureModeGlobal
    try:
        G.dotvec_noalloc(yd, qd)                              # raises if anything in it is illegal under capture
    finally:
        graph = C.c_void_p()
        rc_end = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert rc_end == 0 and graph.value
    assert free_bytes() == before                            # the captured call took no device memory
    assert np.isnan(ctx.download(qd, 64)).all()              # captured, not executed
    gexec = C.c_void_p()
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
    for _ in range(2):
        ctx.vec_set(qd, np.nan, 64)
        ctx.sync()
        assert hip.hipGraphLaunch(gexec, stream) == 0
        ctx.sync()
        got = ctx.download(qd, 64)
        assert np.array_equal(got[:35], want)
        assert np.isnan(got[35:]).all()
    ref = Xa[:, 2:37].T @ ya
    assert np.all(np.abs(got[:35] - ref) <= 1e-12 * (np.abs(Xa[:, 2:37]).T @ np.abs(ya)))
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    # outside a capture: free memory unchanged by a call
    b2 = free_bytes()
    G.dotvec_noalloc(yd, qd)
    ctx.sync()
    assert free_bytes() == b2
    for p in (yd, qd):
        ctx.vec_free(p)
    G.destroy()
