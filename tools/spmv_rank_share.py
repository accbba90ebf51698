#!/usr/bin/env python3
"""What the product of one rank's share of an R-MAT graph Laplacian costs, for the two partitions of
`bench.py --workload rmat --gpus N`: the rank's rows as a rectangular one-rank operator (its rows x all
columns; local and ghost block in one), timed alone on the device.
Usage: spmv_rank_share.py [scale=26] [ranks=8] [rank ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    which = [int(v) for v in sys.argv[3:]] or [0, P - 1]
    N = 1 << scale
    ctx = ksgpu.Context(0)
    new_of_old, _, hstarts = matgen.hashed_relabeling(N, P)
    cstarts = matgen.balanced_row_blocks(matgen.rmat_row_weights(scale), P, 0.0)
    for name, starts, relabel in (("hashed", hstarts, new_of_old), ("contiguous by entries", cstarts, None)):
        for r in which:
            row0, row1 = int(starts[r]), int(starts[r + 1])
            t0 = time.time()
            rp, ci, va = matgen.rmat_laplacian(scale, row0=row0, row1=row1, relabel=relabel)
            n, nnz = row1 - row0, int(rp[-1])
            def timed(A, ncols):
                x, y = ctx.vec_alloc(ncols), ctx.vec_alloc(n)
                ctx.vec_fill_hash(x, ncols, 0, 7)
                for _ in range(3):
                    A.mult(x, y)
                ctx.sync()
                ctx.enable_timing(True)
                ctx.stats_reset()
                for _ in range(10):
                    A.mult(x, y)
                ctx.sync()
                st = ctx.stats()["matmult"]
                ctx.enable_timing(False)
                ctx.vec_free(x)
                ctx.vec_free(y)
                return st["ms"] / st["calls"]
            A = ksgpu.DeviceCSR(ctx, rp, ci, va, cols=(N, N, 0))      # n rows, all N columns: a rectangular one-rank operator
            ms = timed(A, N)
            print("%s, rank %d of %d: %d rows, %d entries: %.3f ms per product = %.2f ps per entry (generated and set up in %.0f s); %s" % (
                name, r, P, n, nnz, ms, 1e9 * ms / nnz, time.time() - t0, A.describe()[:160]), flush=True)
            A.destroy()
            # the two blocks a rank of the distributed operator multiplies: its own columns (square), and the columns
            # of the other ranks it refers to, numbered in their order as in the ghost buffer (rectangular)
            rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
            own = (ci >= row0) & (ci < row1)
            def block(mask, cols):
                cnt = np.bincount(rows[mask], minlength=n)
                return np.concatenate(([0], np.cumsum(cnt))).astype(np.int64), cols.astype(np.int32), va[mask]
            lrp, lci, lva = block(own, ci[own] - row0)
            ghost_cols = np.unique(ci[~own])
            grp, gci, gva = block(~own, np.searchsorted(ghost_cols, ci[~own]))
            del rows, own, rp, ci, va
            L = ksgpu.DeviceCSR(ctx, lrp, lci, lva)
            msl = timed(L, n)
            G = ksgpu.DeviceCSR(ctx, grp, gci, gva, cols=(len(ghost_cols), len(ghost_cols), 0))
            msg = timed(G, len(ghost_cols))
            print("    as the distributed operator holds it: local block %d entries %.3f ms, ghost block %d entries over %d ghost columns %.3f ms: %.3f ms together; ghost: %s" % (
                int(lrp[-1]), msl, int(grp[-1]), len(ghost_cols), msg, msl + msg, G.describe()[:120]), flush=True)
            L.destroy()
            G.destroy()


if __name__ == "__main__":
    main()
