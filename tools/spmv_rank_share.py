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
            A = ksgpu.DeviceCSR(ctx, rp, ci, va, cols=(N, N, 0))      # n rows, all N columns: a rectangular one-rank operator
            del rp, ci, va
            x, y = ctx.vec_alloc(N), ctx.vec_alloc(n)
            ctx.vec_fill_hash(x, N, 0, 7)
            for _ in range(3):
                A.mult(x, y)
            ctx.sync()
            ctx.enable_timing(True)
            ctx.stats_reset()
            for _ in range(10):
                A.mult(x, y)
            ctx.sync()
            s = ctx.stats()["matmult"]
            ctx.enable_timing(False)
            ms = s["ms"] / s["calls"]
            print("%s, rank %d of %d: %d rows, %d entries: %.3f ms per product = %.2f ps per entry (generated and set up in %.0f s); %s" % (
                name, r, P, n, nnz, ms, 1e9 * ms / nnz, time.time() - t0, A.describe()[:160]), flush=True)
            A.destroy()
            ctx.vec_free(x)
            ctx.vec_free(y)


if __name__ == "__main__":
    main()
