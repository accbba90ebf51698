#!/usr/bin/env python3
"""SpMV time on an R-MAT graph Laplacian: the plain operator (column groups as chosen by size) against
the binned form (KSG_OPT_BINS) for several settings of its tuning variables.
Usage: spmv_bins.py [scale=24] [settings ...]   a setting is W=..,H=..,E=..,T=..,TASK_GROUPS=.. (KSG_BINS_*);
"plain" = the operator without a binned part."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu, matgen
if os.environ.get("KSGPU_LIB"):                       # a build of the device library with other compile-time settings
    ksgpu.LIB_PATH = os.environ["KSGPU_LIB"]


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    settings = sys.argv[2:] or ["plain", "W=16384,H=2048,T=4"]
    t0 = time.perf_counter()
    cache = "/tmp/rmat_laplacian_%d.npz" % scale          # several profiler passes share one generation
    if os.path.exists(cache):
        z = np.load(cache)
        rp, ci, va = z["rp"], z["ci"], z["va"]
    else:
        rp, ci, va = matgen.rmat_laplacian(scale)
        if os.environ.get("KSG_RMAT_CACHE"):
            np.savez(cache, rp=rp, ci=ci, va=va)
    n = len(rp) - 1
    print("scale %d: n=%d nnz=%d, generated in %.0f s" % (scale, n, rp[-1], time.perf_counter() - t0), flush=True)
    ctx = ksgpu.Context(0)
    x, y = ctx.vec_alloc(n), ctx.vec_alloc(n)
    ctx.vec_fill_hash(x, n, 0, 5)
    y0 = None
    for setting in settings:
        for k in ("W", "H", "E", "T", "TASK_GROUPS", "THREADS", "ALIGN"):
            os.environ.pop("KSG_BINS_" + k, None)
        if setting == "plain":
            ctx.set_option(ksgpu.OPT_BINS, 1)
        else:
            ctx.set_option(ksgpu.OPT_BINS, 2)
            for kv in setting.split(","):
                k, v = kv.split("=")
                os.environ["KSG_BINS_" + k] = v
        t0 = time.perf_counter()
        A = ksgpu.DeviceCSR(ctx, rp, ci, va)
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
        print("%-40s %.3f ms per product, %.2f TB/s of the CSR model, %.2f TB/s of the bytes it moves, %.1f G entries/s; "
              "set up in %.0f s; max difference to the first %.2e of max |y|" % (
                  setting, s["ms"] / s["calls"], cm * s["calls"] / s["ms"] / 1e9, ex * s["calls"] / s["ms"] / 1e9,
                  rp[-1] * s["calls"] / s["ms"] / 1e6, tc, np.abs(yb - y0).max() / np.abs(y0).max()), flush=True)
        print("    " + A.describe(), flush=True)
        A.destroy()


if __name__ == "__main__":
    main()
