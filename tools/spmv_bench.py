#!/usr/bin/env python3
"""SpMV timing on the stencil operators of the benchmark configurations.

    python tools/spmv_bench.py ex2:4096 lap3d:256 lap3d:512 lap1d:134217728 lap2d:512:262144

Prints the HIP-event time per product, the algorithmic GB/s and a checksum of y
(equal between runs with different kernel options when the results are bit-identical)."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen
if os.environ.get("KSGPU_LIB"):                       # a build of the device library with other compile-time settings
    ksgpu.LIB_PATH = os.environ["KSGPU_LIB"]


def main():
    ctx = ksgpu.Context(0)
    for spec in sys.argv[1:] or ["ex2:4096"]:
        kind, g = spec.split(":", 1)
        t0 = time.time()
        if kind == "lap1d":                                  # lap1d:N -- three entries per row, offsets +-1
            rp, ci, va = matgen.laplacian1d(int(g))
        elif kind == "lap2d":                                # lap2d:n:m -- n x m grid, offsets +-1, +-n
            a, b = (int(v) for v in g.split(":"))
            rp, ci, va = matgen.laplacian2d(a, b)
        else:
            g = int(g)
            rp, ci, va = matgen.laplacian2d(g) if kind == "ex2" else matgen.laplacian3d(g, g, g)
        n = len(rp) - 1
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
        nnz = int(rp[-1])
        del rp, ci, va
        # KSG_VEC_SKEW=a,b: x and y start a and b bytes into their allocations (where three streams of
        # the same power-of-two length meet the memory channels)
        skew = [int(v) for v in os.environ.get("KSG_VEC_SKEW", "0,0").split(",")]
        x0, y0 = ctx.vec_alloc(n + skew[0] // 8), ctx.vec_alloc(n + skew[1] // 8)
        x, y = x0 + skew[0], y0 + skew[1]
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
        ctx.vec_free(x0)
        ctx.vec_free(y0)


if __name__ == "__main__":
    main()
