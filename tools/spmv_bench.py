#!/usr/bin/env python3
"""SpMV timing on the stencil operators of the benchmark configurations.

    python tools/spmv_bench.py ex2:4096 lap3d:256 lap3d:512

Prints the HIP-event time per product, the algorithmic GB/s and a checksum of y
(equal between runs with different kernel options when the results are bit-identical)."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen


def main():
    ctx = ksgpu.Context(0)
    for spec in sys.argv[1:] or ["ex2:4096"]:
        kind, g = spec.split(":")
        g = int(g)
        t0 = time.time()
        rp, ci, va = matgen.laplacian2d(g) if kind == "ex2" else matgen.laplacian3d(g, g, g)
        n = len(rp) - 1
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
        nnz = int(rp[-1])
        del rp, ci, va
        x, y = ctx.vec_alloc(n), ctx.vec_alloc(n)
        ctx.vec_fill_hash(x, n, 0, 7)
        for _ in range(3):
            A.mult(x, y)
        ctx.sync()
        ctx.enable_timing(True)
        ctx.stats_reset()
        reps = 20
        for _ in range(reps):
            A.mult(x, y)
        ctx.sync()
        s = ctx.stats()["matmult"]
        ctx.enable_timing(False)
        h = ctx.download(y, n)
        print("%-10s n=%d nnz=%d  %.4f ms  %.1f GB/s  sha=%s  (setup %.1fs)" % (
            spec, n, nnz, s["ms"] / s["calls"], s["bytes"] / s["ms"] / 1e6,
            hashlib.sha1(h.tobytes()).hexdigest()[:12], time.time() - t0), flush=True)
        A.destroy()
        ctx.vec_free(x)
        ctx.vec_free(y)


if __name__ == "__main__":
    main()
