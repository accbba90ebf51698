#!/usr/bin/env python3
"""SpMV rate on an R-MAT graph Laplacian (power-law rows).  Usage: spmv_rmat.py [scale=22]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    rp, ci, va = matgen.rmat_laplacian(scale)
    n = len(rp) - 1
    lens = np.diff(rp)
    print("n=%d nnz=%d max row %d, rows > 32: %d, > 256: %d, > 4096: %d" % (
        n, rp[-1], lens.max(), (lens > 32).sum(), (lens > 256).sum(), (lens > 4096).sum()), flush=True)
    ctx = ksgpu.Context(0)
    x, y = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.vec_fill_hash(x, n, 0, 5)
    for bal in ("0", "1"):
        os.environ["KSG_SPMV_BALANCE"] = bal
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
        A.mult(x, y)
        ctx.sync()
        ctx.enable_timing(True)
        ctx.stats_reset()
        for _ in range(10):
            A.mult(x, y)
        ctx.sync()
        s = ctx.stats()["matmult"]
        print("balance=%s info=%s  %.3f ms  %.1f GB/s  %.1f Gnnz/s" % (
            bal, A.info(), s["ms"] / s["calls"], s["bytes"] / s["ms"] / 1e6, rp[-1] * s["calls"] / s["ms"] / 1e6), flush=True)
        ctx.enable_timing(False)
        yb = ctx.download(y, n)
        if bal == "0":
            y0 = yb
        else:
            print("max |y_balanced - y_plain| = %.3e (scale %.3e)" % (np.abs(yb - y0).max(), np.abs(y0).max()))
        A.destroy()


if __name__ == "__main__":
    main()
