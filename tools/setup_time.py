#!/usr/bin/env python3
"""Set-up time of an operator on the device (validation, choice of the form, conversion, upload), with the
stages the library reports under KSG_SETUP_TIMING=1.
Usage: setup_time.py lap2d:4096 | lap3d:512 | rmat:24 ...   (KSG_HOST_THREADS sets the host threads)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KSG_SETUP_TIMING", "1")
from slepc_amd import ksgpu, matgen


def main():
    ctx = ksgpu.Context(0)
    for spec in sys.argv[1:]:
        kind, size = spec.split(":")
        size = int(size)
        t0 = time.perf_counter()
        if kind == "lap2d":
            rp, ci, va = matgen.laplacian2d(size, size)
        elif kind == "lap3d":
            rp, ci, va = matgen.laplacian3d(size, size, size)
        else:
            rp, ci, va = matgen.rmat_laplacian(size)
        print("%s: n=%d nnz=%d, generated in %.1f s" % (spec, len(rp) - 1, rp[-1], time.perf_counter() - t0), flush=True)
        for rep in range(2):
            t0 = time.perf_counter()
            A = ksgpu.DeviceCSR(ctx, rp, ci, va)
            ctx.sync()
            print("%s: set up in %.2f s (%d host threads)" % (spec, time.perf_counter() - t0, int(os.environ.get("KSG_HOST_THREADS", "0")) or 16), flush=True)
            A.destroy()
        del rp, ci, va


if __name__ == "__main__":
    main()
