#!/usr/bin/env python3
"""Restart timing of the singular value solvers on the benchmark operator (the 2-D 5-point
Laplacian of ex2, g x g grid): K restarts of each solver with the library's own per-kernel
statistics, as bench.py reports them for the eigensolver.
Usage: python tools/svdbench.py [grid=4096] [nsv=32] [restarts=3]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, kseps as K, matgen


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    nsv = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    restarts = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = ksgpu.Context(0)
    rp, ci, va = matgen.laplacian2d(g)
    n = g * g
    A = K.Mat.create_aij_hip(ctx, rp, ci, va)
    del rp, ci, va
    out = {"grid": g, "rows": n, "nsv": nsv, "restarts": restarts, "solvers": {}}
    for name, stype, oneside in (("trlanczos", "trlanczos", False), ("trlanczos one-sided", "trlanczos", True),
                                 ("lanczos", "lanczos", False), ("cross", "cross", False), ("cyclic", "cyclic", False)):
        svd = K.SVD(ctx.h)
        svd.set_operator(A)
        svd.set_type(stype)
        svd.set_dimensions(nsv)
        svd.set_tolerances(1e-8, restarts + 1)        # the first restart builds the whole basis: not timed apart
        if oneside:
            svd.set_oneside(True)
        K.check(svd.lib.SVDSetUp(svd.h))              # transpose built here, outside the timing
        ctx.sync()
        ctx.stats_reset()
        ctx.enable_timing(True)
        t0 = time.perf_counter()
        svd.solve()
        ctx.sync()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        ctx.enable_timing(False)
        nbytes = sum(v["bytes"] for v in st.values())
        kern = {k: {"calls": v["calls"], "ms": round(v["ms"], 2), "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] else None}
                for k, v in st.items() if v["calls"]}
        out["solvers"][name] = {"its": svd.its, "wall_s": round(dt, 3), "ms_per_restart": round(1e3 * dt / max(svd.its, 1), 2),
                                "GBps": round(nbytes / dt / 1e9, 1), "ncv": svd.dimensions()[1], "kernels": kern}
        print("%-20s its %d  %.1f ms/restart  %.0f GB/s (ncv %d)" % (name, svd.its, 1e3 * dt / max(svd.its, 1), nbytes / dt / 1e9,
                                                                    svd.dimensions()[1]), flush=True)
        svd.destroy()
    print(json.dumps(out))
    A.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
