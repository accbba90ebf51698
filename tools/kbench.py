#!/usr/bin/env python3
"""Per-kernel micro-benchmark: achieved algorithmic GB/s of each BV kernel and of
the SpMV at the benchmark shapes (HIP-event timing from the library's own
statistics).  Usage: python tools/kbench.py [grid]  (default 4096 -> 16.7M rows)."""
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ctx = ksgpu.Context(0)
    n = g * g
    m = 65
    t0 = time.time()
    rp, ci, va = matgen.laplacian2d(g)
    A = ksgpu.DeviceCSR(ctx, rp, ci, va)
    print("matrix: n=%d nnz=%d gen+upload %.1fs info=%s" % (n, rp[-1], time.time() - t0, A.info()), flush=True)
    V = ksgpu.DeviceBV(ctx, n, m)
    for j in range(m):
        ctx.vec_fill_hash(V.column(j), n, 0, 1000 + j)
    ctx.sync()
    ctx.enable_timing(True)
    res = {}

    def run(name, ev, fn):
        fn()
        ctx.sync()
        ctx.stats_reset()
        for _ in range(reps):
            fn()
        ctx.sync()
        s = ctx.stats()[ev]
        gbs = s["bytes"] / (s["ms"] * 1e-3) / 1e9 if s["ms"] else 0.0
        res[name] = {"ms": s["ms"] / s["calls"], "GBps": gbs}
        print("%-22s %9.3f ms  %8.1f GB/s" % (name, s["ms"] / s["calls"], gbs), flush=True)

    run("spmv", "matmult", lambda: A.mult(V.column(0), V.column(64)))
    for J in (8, 16, 33, 48, 64):
        V.set_active(0, J + 1)
        run("dotvec k=%d" % (J + 1), "dotvec", lambda: V.dotvec(V.column(J)))
        V.set_active(0, J)
        q = np.full(J, 1e-3)
        run("multvec k=%d" % J, "multvec", lambda: V.multvec(-1.0, 1.0, V.column(J), q))
    V.set_active(0, 64)
    Q = np.asfortranarray(np.eye(64) + 1e-3)
    run("multinplace 64x32", "multinplace", lambda: V.multinplace(Q, 0, 32))
    run("multinplace 64x48", "multinplace", lambda: V.multinplace(Q, 0, 48))
    # block operations (not on the Krylov-Schur path): Gram matrix and out-of-place product
    W = ksgpu.DeviceBV(ctx, n, 64)
    for kk in (16, 32, 64):
        V.set_active(0, kk)
        W.set_active(0, kk)
        M = np.zeros((64, m), order="F")
        run("dot %dx%d" % (kk, kk), "dot", lambda: V.dot(W, M))
        run("dot %dx%d, V with itself" % (kk, kk), "dot", lambda: V.dot(V, M))
        Qm = np.asfortranarray(np.eye(m, 64) * 0.5 + 1e-3)
        run("mult %dx%d" % (kk, kk), "mult", lambda: W.mult(1.0, 0.0, V, Qm))
    # block product W = A V: one pass over the matrix for all the columns
    for kk in (1, 8, 32):
        V.set_active(0, kk)
        W.set_active(0, kk)
        run("matmult %d cols" % kk, "matmult", lambda: V.matmult(A, W))
    W.destroy()
    V.set_active(0, 64)
    run("scale", "scale", lambda: V.scale(3, 1.0000001))
    run("norm", "norm", lambda: V.norm(3, 1))
    run("copy", "copy", lambda: V.copy_column(3, 64))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
