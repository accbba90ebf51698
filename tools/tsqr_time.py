import sys, time
sys.path.insert(0, '.')
import numpy as np
from slepc_amd import ksgpu
ctx = ksgpu.Context(0)
n = 1 << 24
for k in (8, 16, 32, 64):
    V = ksgpu.DeviceBV(ctx, n, k)
    for j in range(k):
        ctx.vec_fill_hash(V.column(j), n, 0, 100 + j)
    ctx.sync()
    R = np.zeros((k, k), order="F")
    ctx.enable_timing(True); ctx.stats_reset()
    t0 = time.perf_counter(); V.tsqr(R); ctx.sync(); t1 = time.perf_counter()
    st = ctx.stats(); ctx.enable_timing(False)
    print("   device ms: triangles %.1f, V R^-1 %.1f" % (st["dot"]["ms"], st["multinplace"]["ms"]))
    M = np.zeros((k, k), order="F"); V.dot(V, M)
    print("tsqr n=%d k=%d: %.1f ms (two sweeps), |Q^TQ-I|max=%.2e, R11=%.6g" % (n, k, 1e3 * (t1 - t0), np.abs(M - np.eye(k)).max(), R[0, 0]), flush=True)
    V.destroy()
