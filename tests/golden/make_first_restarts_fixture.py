#!/usr/bin/env python3
"""Ritz values and error estimates after each of the first restarts of BASELINE configuration 3 at
its full size -- the 7-point Laplacian on the 512^3 grid, nev = 32, ncv = 64, hashed start vector
-- from the CPU restatement of the reference path (``oracle.eps.EPS``: krylovschur.c:169-288 /
ks-symm.c:30-111, on the compiled kernels of ``oracle/ckern.c``), written as the fixture
``tests/test_gpu_fullsize.py::test_lap3d_512_first_restarts_match_oracle`` compares the product with.
CPU only; 70 GB of basis on the host and about a minute on 16 threads, so it is run on the host cores
of the GPU box:

    python tests/golden/make_first_restarts_fixture.py [grid=512 [restarts=2 [threads=16 [output directory]]]]
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    its = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    out_dir = sys.argv[4] if len(sys.argv) > 4 else HERE
    from oracle import ckern, eps as oeps
    from slepc_amd import matgen
    nev = 32
    rp, ci, va = matgen.laplacian3d(g, g, g)
    n = g ** 3
    kern = ckern.Kernels(threads)
    rec = {}
    t0 = time.perf_counter()

    def mon(i, nconv, er, ei, ee, nv):
        rec[str(int(i))] = {"nconv": int(nconv), "nest": int(nv), "re": [float(v) for v in er[:nev]],
                            "errest": [float(v) for v in ee[:nev]]}
        print("restart %d after %.0f s" % (i, time.perf_counter() - t0), flush=True)

    o = oeps.EPS(kern.csr(rp, ci, va).mult, n, hermitian=True, nev=nev, max_it=its, bv_factory=kern.bv, monitor=mon)
    o.solve()
    out = {"workload": "3-D 7-point Laplacian %dx%dx%d (%d rows), symmetric Krylov-Schur nev=%d" % (g, g, g, n, nev),
           "rows": n, "entries": int(rp[-1]), "nev": nev, "ncv": 64,
           "start_vector": "2 u(splitmix64(0x12345678 ^ g)) - 1 of the global row g (the solver's default)",
           "restarts": rec,
           "made_by": "tests/golden/make_first_restarts_fixture.py %d %d: oracle.eps.EPS on oracle/ckern.c, %d threads, %.0f s" % (
               g, its, threads, time.perf_counter() - t0)}
    path = os.path.join(out_dir, "lap3d%d_oracle_first_restarts.json" % g)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
