#!/usr/bin/env python3
"""SpMV time on an R-MAT graph Laplacian against the number of column groups the operator is held
in (KSG_OPT_COL_GROUPS; 1 = the whole operator in one block).
Usage: spmv_colgroups.py [scale=24] [groups=1,2,4,8] [limit of packed columns per group=1048576,...]
KSG_RANK_OF=r/P: only the rows that rank r of P owns under the entry-balanced split, as a rectangular
operator over all columns -- the shape of one rank's work in a P-rank run (its ghost entries included).
KSG_SWEEP_E=a,b,...: every setting once per value of KSG_BINS_E."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    groups = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
    hots = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1048576").split(",")]
    t0 = time.perf_counter()
    NC = 1 << scale
    cols = None
    if os.environ.get("KSG_RANK_OF"):
        r, P = (int(v) for v in os.environ["KSG_RANK_OF"].split("/"))
        starts = matgen.balanced_row_blocks(matgen.rmat_row_weights(scale), P, 0.0)
        rp, ci, va = matgen.rmat_laplacian(scale, row0=int(starts[r]), row1=int(starts[r + 1]))
        cols = (NC, NC, 0)
        va = np.full_like(va, -1.0)          # a ghost block holds off-diagonal entries only: one value
        print("rows %d..%d of rank %d of %d" % (starts[r], starts[r + 1], r, P))
    else:
        rp, ci, va = matgen.rmat_laplacian(scale)
    n = len(rp) - 1
    print("scale %d: n=%d nnz=%d, generated in %.0f s" % (scale, n, rp[-1], time.perf_counter() - t0), flush=True)
    ctx = ksgpu.Context(0)
    x, y = ctx.vec_alloc(NC), ctx.vec_alloc(n)
    ctx.vec_fill_hash(x, NC, 0, 5)
    y0 = None
    # KSG_SWEEP_E=65536,131072,...: the same operator once per value of KSG_BINS_E (with KSG_SPMV_BINS=2)
    sweep = [v for v in os.environ.get("KSG_SWEEP_E", "").split(",") if v]
    for g, hot, e in [(g, h, e) for g in groups for h in hots for e in (sweep or [None])]:
        if e is not None:
            os.environ["KSG_BINS_E"] = e
            print("KSG_BINS_E=%s" % e, flush=True)
        ctx.set_option(ksgpu.OPT_COL_GROUPS, g)
        ctx.set_option(ksgpu.OPT_HOT_COLS, hot)
        t0 = time.perf_counter()
        A = ksgpu.DeviceCSR(ctx, rp, ci, va, cols=cols)
        tc = time.perf_counter() - t0
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
        ex, cm = A.traffic_model()
        print("groups %3d, at most %7d packed columns each: %.3f ms per product, %.2f TB/s of the CSR model, %.2f TB/s of the bytes it moves, %.1f G entries/s; "
              "packed %d; set up in %.0f s; max difference to the first %.2e of max |y|" % (
                  g, hot, s["ms"] / s["calls"], cm * s["calls"] / s["ms"] / 1e9, ex * s["calls"] / s["ms"] / 1e9,
                  rp[-1] * s["calls"] / s["ms"] / 1e6, A.hot_columns(), tc, np.abs(yb - y0).max() / np.abs(y0).max()), flush=True)
        A.destroy()


if __name__ == "__main__":
    main()
