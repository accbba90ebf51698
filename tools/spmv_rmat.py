#!/usr/bin/env python3
"""SpMV rate on an R-MAT graph Laplacian (power-law rows and columns) for several limits of
the packed hot columns (KSG_OPT_HOT_COLS; 0 = no packing).
Usage: spmv_rmat.py [scale=22] [limits=0,65536,262144,524288]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    limits = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,65536,262144,524288").split(",")]
    cache = "/tmp/rmat_laplacian_%d.npz" % scale          # several profiler passes share one generation
    if os.path.exists(cache):
        z = np.load(cache)
        rp, ci, va = z["rp"], z["ci"], z["va"]
    else:
        rp, ci, va = matgen.rmat_laplacian(scale)
        if os.environ.get("KSG_RMAT_CACHE"):
            np.savez(cache, rp=rp, ci=ci, va=va)
    n = len(rp) - 1
    lens = np.diff(rp)
    print("n=%d nnz=%d max row %d, rows > 32: %d, > 256: %d, > 4096: %d" % (
        n, rp[-1], lens.max(), (lens > 32).sum(), (lens > 256).sum(), (lens > 4096).sum()), flush=True)
    ctx = ksgpu.Context(0)
    x, y = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.vec_fill_hash(x, n, 0, 5)
    y0 = None
    for lim in limits:
        ctx.set_option(ksgpu.OPT_HOT_COLS, lim)
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
        A.mult(x, y)
        ctx.sync()
        ctx.enable_timing(True)
        ctx.stats_reset()
        for _ in range(int(os.environ.get("KSG_SPMV_REPS", "20"))):
            A.mult(x, y)
        ctx.sync()
        s = ctx.stats()["matmult"]
        ctx.enable_timing(False)
        yb = ctx.download(y, n)
        if y0 is None:
            y0 = yb
        print("hot limit %7d packed %7d  %.3f ms  %.1f GB/s  %.1f Gnnz/s  same bits as the first: %s" % (
            lim, A.hot_columns(), s["ms"] / s["calls"], s["bytes"] / s["ms"] / 1e6,
            rp[-1] * s["calls"] / s["ms"] / 1e6, np.array_equal(yb, y0)), flush=True)
        A.destroy()


if __name__ == "__main__":
    main()
