#!/usr/bin/env python3
"""Sweep of the row-grid bounds of the reduction and update kernels (the options
KSG_OPT_DOT_GRID / KSG_OPT_MV_GRID) at the benchmark shape.
Usage: python tools/ksweep.py [grid=4096]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slepc_amd import ksgpu


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reps = 5
    ctx = ksgpu.Context(0)
    n = g * g
    m = 66
    V = ksgpu.DeviceBV(ctx, n, m)
    for j in range(m):
        ctx.vec_fill_hash(V.column(j), n, 0, 1000 + j)
    ctx.sync()
    ctx.enable_timing(True)

    def run(ev, fn):
        fn()
        ctx.sync()
        ctx.stats_reset()
        for _ in range(reps):
            fn()
        ctx.sync()
        s = ctx.stats()[ev]
        return s["bytes"] / (s["ms"] * 1e-3) / 1e9

    grids = [256, 384, 512, 640, 768, 1024, 1536, 2048, 4096]
    print("dotvec GB/s; columns = grid bound " + " ".join("%6d" % x for x in grids))
    for k in (5, 9, 13, 17, 25, 33, 34, 41, 49, 57, 65):
        V.set_active(0, k)
        row = []
        for gr in grids:
            ctx.set_option(ksgpu.OPT_DOT_GRID, gr)
            row.append(run("dotvec", lambda: V.dotvec(V.column(65))))
        print("dot k=%3d                        " % k + " ".join("%6.0f" % x for x in row), flush=True)
    ctx.set_option(ksgpu.OPT_DOT_GRID, 256)
    print("multvec (fused) GB/s")
    for k in (4, 8, 12, 16, 24, 32, 33, 40, 48, 56, 64):
        V.set_active(0, k)
        q = np.full(k, 1e-3)
        row = []
        for gr in grids:
            ctx.set_option(ksgpu.OPT_MV_GRID, gr)
            row.append(run("multvec", lambda: V.multvec(-1.0, 1.0, V.column(k), q)))
        print("mv  k=%3d                        " % k + " ".join("%6.0f" % x for x in row), flush=True)


if __name__ == "__main__":
    main()
