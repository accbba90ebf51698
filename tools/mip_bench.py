"""Times ksg_bv_multinplace at the benchmark shape (16.7 M rows): V(:,0:ne) = V(:,0:K) Q.
Environment: KSG_MIP_VALU=1 (vector kernel), KSG_MIP_ROWS=2|4, KSG_MIP_GRID=<workgroups>."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slepc_amd import ksgpu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16777216
ctx = ksgpu.Context(0)
ctx.enable_timing(True)
rng = np.random.default_rng(0)
V = ksgpu.DeviceBV(ctx, n, 65)
for j in range(65):
    ctx.vec_fill_hash(V.column(j), n, 0, 100 + j)
for K, ne in ((64, 32), (64, 48), (64, 64), (48, 24), (33, 17)):
    Q = np.asfortranarray(np.linalg.qr(rng.standard_normal((K, K)))[0])
    V.set_active(0, K)
    for _ in range(2):
        V.multinplace(Q, 0, ne)
    ctx.sync()
    ctx.stats_reset()
    reps = 6
    t0 = time.perf_counter()
    for _ in range(reps):
        V.multinplace(Q, 0, ne)
    ctx.sync()
    wall = (time.perf_counter() - t0) / reps * 1e3
    st = ctx.stats()["multinplace"]
    sec = st["ms"] * 1e-3
    print("K=%2d ne=%2d  %.3f ms per call (events), %.3f ms wall, %.0f GB/s, %.1f TFLOP/s"
          % (K, ne, st["ms"] / st["calls"], wall, st["bytes"] / sec / 1e9, st["flops"] / sec / 1e12), flush=True)
V.destroy()
ctx.close()
