#!/usr/bin/env python3
"""Converged eigenvalues of BASELINE configurations 4 and 5 from the CPU restatement of the
reference path (``oracle.eps.EPS``: the control flow of krylovschur.c:169-288 / ks-symm.c:30-111,
on the compiled row-parallel kernels of ``oracle/ckern.c``), written as fixtures the GPU tests
compare the product with (``tests/test_gpu_converged.py``).  CPU only; run in the build container:

    python tests/golden/make_converged_fixtures.py markov 3000 [threads [output directory]]
    python tests/golden/make_converged_fixtures.py rmat 24 [threads [output directory]]

(The scale-26 fixture was made by the same script on the host cores of the GPU box -- no GPU involved -- because
the build container's sessions did not last the twelve minutes it takes; the output directory is for that.)

* markov m: ex5's Markov chain (ex5.c:150-193), nonsymmetric, largest real part, nev = 16
  (ncv = 32), start vector e0 + e1 (ex5.c:86-94), tol 1e-12 -- the tolerance at which the
  eigenvalues of a nonsymmetric problem are determined to the 1e-10 the comparison asks for.
* rmat s: graph Laplacian of the symmetrised R-MAT graph of scale s (edge factor 8,
  (a,b,c) = (0.57,0.19,0.19), seed 0x5EED; ``slepc_amd.matgen.rmat_laplacian``), symmetric,
  nev = 32 (ncv = 64), hashed start vector, tol 1e-10.

The fixture records the matrix it was made from (rows, entries, a checksum of the column
indices), so a test that generates a different matrix fails on that and not on eigenvalues.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402


def csr_checksum(rp, ci):
    """Order-dependent 64-bit checksum of the pattern: sum over entries of (position+1)*(column+1) mod 2^64."""
    total = np.uint64(0)
    step = 1 << 24
    with np.errstate(over="ignore"):
        for a in range(0, ci.size, step):
            c = ci[a:a + step].astype(np.uint64) + np.uint64(1)
            p = np.arange(a, a + c.size, dtype=np.uint64) + np.uint64(1)
            total += np.sum(c * p, dtype=np.uint64)
    return int(total)


def main():
    kind, size = sys.argv[1], int(sys.argv[2])
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else min(16, os.cpu_count() or 1)
    from oracle import ckern, eps as oeps
    from slepc_amd import matgen as mg
    t = time.perf_counter()
    if kind == "markov":
        rp, ci, va = mg.markov(size)
        n = len(rp) - 1
        herm, nev, tol, which = False, 16, 1e-12, "largest_real"
        v0 = np.zeros(n)
        v0[:2] = 1.0
        name = "ex5 Markov chain m=%d" % size
        start = "e0 + e1 (ex5.c:86-94)"
    elif kind == "rmat":
        rp, ci, va = mg.rmat_laplacian(size, progress=lambda s: print(s, flush=True))
        n = len(rp) - 1
        herm, nev, tol, which = True, 32, 1e-10, None
        v0 = oeps.splitmix64_uniform(np.arange(n, dtype=np.uint64))
        name = "R-MAT scale %d graph Laplacian" % size
        start = "2 u(splitmix64(0x12345678 ^ g)) - 1 of the global row g"
    else:
        raise SystemExit(__doc__)
    print("%s: %d rows, %d entries, generated in %.1f s" % (name, n, int(rp[-1]), time.perf_counter() - t), flush=True)
    chk = csr_checksum(rp, ci)
    ncv = max(2 * nev, nev + 15)
    cap = max(100, 2 * n // ncv)                      # krylovschur.c:89
    kern = ckern.Kernels(threads)
    A = kern.csr(rp, ci, va)
    hist = []
    t0 = time.perf_counter()

    def mon(its, nconv, er, ei, ee, nv):
        if not hist or hist[-1][1] != nconv:
            hist.append((int(its), int(nconv)))
        if its % 100 == 0 or kind == "rmat":
            k = min(nconv, nv - 1)
            print("restart %d: nconv %d, first unconverged %.15g (errest %.3e), %.0f s" % (
                its, nconv, er[k], ee[k], time.perf_counter() - t0), flush=True)

    o = oeps.EPS(A.mult, n, hermitian=herm, nev=nev, tol=tol, v0=v0, max_it=cap, which=which,
                 monitor=mon, bv_factory=kern.bv)
    o.solve()
    secs = time.perf_counter() - t0
    nc = min(o.nconv, nev + 1)
    vals = [list(map(float, o.get_eigenvalue(i))) for i in range(nc)]
    errs = [float(o.compute_error(i)) for i in range(min(o.nconv, nev))]
    out = {"workload": name, "rows": n, "entries": int(rp[-1]), "colidx_checksum": chk,
           "problem": "HEP" if herm else "NHEP, largest real part", "nev": nev, "ncv": ncv, "tol": tol,
           "start_vector": start, "restart_cap": cap,
           "restarts": int(o.its), "nconv": int(o.nconv), "reason": int(o.reason),
           "eigenvalues": vals, "relative_residuals": errs,
           "nconv_history": hist,
           "made_by": "tests/golden/make_converged_fixtures.py %s %d: oracle.eps.EPS on oracle/ckern.c, %d threads, %.0f s" % (
               kind, size, threads, secs)}
    path = os.path.join(sys.argv[4] if len(sys.argv) > 4 else HERE, "converged_%s%d.json" % (kind, size))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %s: %d restarts, %d converged, max residual %.3e, %.0f s" % (path, o.its, o.nconv, max(errs or [0]), secs))


if __name__ == "__main__":
    main()
