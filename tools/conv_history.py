#!/usr/bin/env python3
"""Restart-by-restart convergence history of the product (GPU) next to the CPU restatement of
the reference path (oracle, on its compiled kernels) on the same operator, start vector and
settings: number of converged pairs and the error estimate of the first unconverged pair after
every restart.  Shows whether a slow or stalling solve is the algorithm's doing (both sides
alike) or the product's.

    python tools/conv_history.py ex2 1024            # 2-D Laplacian 1024^2, nev=32
    python tools/conv_history.py markov 500          # ex5 Markov chain m=500, nev=16, largest real
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    workload, size = sys.argv[1], int(sys.argv[2])
    max_it = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    from oracle import ckern, eps as oeps
    from slepc_amd import ksgpu, kseps as K, matgen as mg
    herm = workload != "markov"
    if workload == "ex2":
        csr, nev, name = mg.laplacian2d(size), 32, "ex2 2-D 5-point Laplacian %dx%d" % (size, size)
    elif workload == "lap3d":
        csr, nev, name = mg.laplacian3d(size), 32, "3-D 7-point Laplacian %d^3" % size
    else:
        csr, nev, name = mg.markov(size), 16, "ex5 Markov chain m=%d" % size
    rp, ci, va = csr
    n = len(rp) - 1
    if herm:
        v0 = oeps.splitmix64_uniform(np.arange(n, dtype=np.uint64))
    else:
        v0 = np.zeros(n)
        v0[:2] = 1.0
    ncv = max(2 * nev, nev + 15)
    cap = max_it or max(100, 2 * n // ncv)

    def rec(hist):
        def mon(its, nconv, er, ei, ee, nv):
            k = min(nconv, nv - 1)
            hist.append((int(its), int(nconv), float(er[k]), float(ee[k])))
        return mon

    # product
    ctx = ksgpu.Context(0)
    A = K.Mat.create_aij_hip(ctx, rp, ci, va)
    eps = K.EPS(ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_HEP if herm else K.EPS_NHEP)
    if not herm:
        eps.set_which(K.EPS_LARGEST_REAL)
    eps.set_dimensions(nev)
    eps.set_tolerances(0.0, cap)
    eps.set_bv_type("hip")
    vs = A.create_vecs()
    vs.set_values(v0)
    eps.set_initial_space([vs])
    hg = []
    eps.set_monitor(rec(hg))
    t = time.perf_counter()
    eps.solve()
    tg = time.perf_counter() - t
    got = {"its": eps.its, "nconv": eps.nconv, "reason": eps.reason, "seconds": round(tg, 2),
           "values": [eps.get_eigenvalue(i) for i in range(min(eps.nconv, nev))]}
    print("product: %d restarts, %d converged, reason %d, %.1f s" % (eps.its, eps.nconv, eps.reason, tg), flush=True)
    # oracle
    kern = ckern.Kernels(min(16, os.cpu_count() or 1))
    ho = []
    t = time.perf_counter()
    o = oeps.EPS(kern.csr(rp, ci, va).mult, n, hermitian=herm, nev=nev, v0=v0, max_it=cap, bv_factory=kern.bv,
                 which=None if herm else "largest_real", monitor=rec(ho)).solve()
    to = time.perf_counter() - t
    want = {"its": o.its, "nconv": o.nconv, "reason": o.reason, "seconds": round(to, 2),
            "values": [o.get_eigenvalue(i) for i in range(min(o.nconv, nev))]}
    print("oracle:  %d restarts, %d converged, reason %d, %.1f s" % (o.its, o.nconv, o.reason, to), flush=True)

    def first_at(h, k):
        return next((r[0] for r in h if r[1] >= k), None)
    m = min(len(hg), len(ho))
    same_nconv = sum(1 for a, b in zip(hg[:m], ho[:m]) if a[1] == b[1])
    # relative difference of the error estimate of the first unconverged pair while both sides
    # have the same number of converged pairs
    rel = [abs(a[3] - b[3]) / max(b[3], 1e-300) for a, b in zip(hg[:m], ho[:m]) if a[1] == b[1] and b[3] > 0]
    lead = next((i for i, (a, b) in enumerate(zip(hg[:m], ho[:m])) if a[1] != b[1]), m)
    vals = min(len(got["values"]), len(want["values"]))
    dv = max((abs(got["values"][i][0] - want["values"][i][0]) / abs(want["values"][i][0]) for i in range(vals)), default=None)
    out = {"workload": "%s (%d rows), nev=%d, ncv=%d, tol=1e-8, restart cap %d (krylovschur.c:89)" % (name, n, nev, ncv, cap),
           "product": got, "oracle": want,
           "restarts_with_equal_nconv": same_nconv, "restarts_compared": m,
           "first_restart_with_different_nconv": lead if lead < m else None,
           "errest_rel_diff_first_50_restarts_max": max(rel[:50]) if rel else None,
           "errest_rel_diff_median": float(np.median(rel)) if rel else None,
           "restart_of_kth_converged_pair": {str(k): [first_at(hg, k), first_at(ho, k)] for k in (1, 2, 4, 8, 16, 24, 32) if k <= nev},
           "max_rel_eigenvalue_difference": dv,
           "history_every_25th_restart": [(a, b) for i, (a, b) in enumerate(zip(hg[:m], ho[:m])) if i % 25 == 24 or i == m - 1]}
    print(json.dumps(out))
    vs.destroy()
    eps.destroy()
    A.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
