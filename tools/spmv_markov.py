#!/usr/bin/env python3
"""SpMV time of ex5's Markov chain operator (general CSR: the tile kernel) against its size.
Usage: spmv_markov.py [m ...]   (rows = m(m+1)/2)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slepc_amd import ksgpu, matgen


def main():
    ctx = ksgpu.Context(0)
    for m in [int(v) for v in sys.argv[1:]] or [3000]:
        rp, ci, va = matgen.markov(m)
        n, nnz = len(rp) - 1, int(rp[-1])
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
        x, y = ctx.vec_alloc(n), ctx.vec_alloc(n)
        ctx.vec_fill_hash(x, n, 0, 7)
        for _ in range(3):
            A.mult(x, y)
        ctx.sync()
        ctx.enable_timing(True)
        ctx.stats_reset()
        for _ in range(50):
            A.mult(x, y)
        ctx.sync()
        s = ctx.stats()["matmult"]
        ctx.enable_timing(False)
        ex, cm = A.traffic_model()
        ms = s["ms"] / s["calls"]
        print("m=%d: %d rows, %d entries: %.4f ms per product, %.0f GB/s of the bytes it moves, %.0f GB/s of the CSR model; %s" % (
            m, n, nnz, ms, ex / ms / 1e6, cm / ms / 1e6, A.describe()[:90]), flush=True)
        A.destroy()
        ctx.vec_free(x)
        ctx.vec_free(y)


if __name__ == "__main__":
    main()
