"""Converged solves of BASELINE configurations 4 and 5 at (or, for the R-MAT graph, near) their own size
against the CPU restatement of the reference path: the fixtures under tests/golden/converged_*.json hold
the eigenvalues `oracle.eps.EPS` converged to on its compiled kernels (made in the build container by
tests/golden/make_converged_fixtures.py, which also records a checksum of the matrix); the product solves
the same problem -- same matrix, start vector, nev, ncv, tolerance, restart cap -- on the device.
Bars: eigenvalues to 1e-10 relative (what the reference's own criterion comes to for a solve against its
CPU path), residuals ||A x - k x|| / |k| < 10 tol."""
import importlib.util
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _checksum(rp, ci):
    spec = importlib.util.spec_from_file_location("make_converged_fixtures", os.path.join(GOLD, "make_converged_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.csr_checksum(rp, ci)


def test_markov_3000_converged_eigenvalues(gpu_ctx):
    """ex5's Markov chain, m = 3000 (4.5M rows), nonsymmetric Krylov-Schur, largest real part, nev = 16, ncv = 32,
    start vector e0 + e1 (ex5.c:86-94), tol 1e-12: some 8000 restarts (the top of the spectrum has gaps of 6e-7)."""
    from slepc_amd import kseps as K, matgen
    fx = _fixture("converged_markov3000.json")
    rp, ci, va = matgen.markov(3000)
    n = len(rp) - 1
    assert n == fx["rows"] and int(rp[-1]) == fx["entries"] and _checksum(rp, ci) == fx["colidx_checksum"]
    A = K.Mat.create_aij_hip(gpu_ctx, rp, ci, va)
    eps = K.EPS(gpu_ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_NHEP)
    eps.set_which(K.EPS_LARGEST_REAL)
    eps.set_dimensions(fx["nev"])
    eps.set_tolerances(fx["tol"], fx["restart_cap"])
    v0 = A.create_vecs()
    start = np.zeros(n)
    start[:2] = 1.0
    v0.set_values(start)
    eps.set_initial_space([v0])
    eps.solve()
    assert eps.dimensions()[1] == fx["ncv"]
    assert eps.reason == K.EPS_CONVERGED_TOL and eps.nconv >= fx["nev"]
    # the same algorithm on the same data: the restart counts differ only through rounding
    assert abs(eps.its - fx["restarts"]) <= 0.05 * fx["restarts"], (eps.its, fx["restarts"])
    for i in range(fx["nev"]):
        kr, ki = eps.get_eigenvalue(i)
        wr, wi = fx["eigenvalues"][i]
        assert abs(complex(kr, ki) - complex(wr, wi)) <= 1e-10 * abs(complex(wr, wi)), (i, kr, ki, wr, wi)
    errs = [eps.compute_error(i) for i in range(fx["nev"])]
    assert max(errs) < 10 * fx["tol"], errs
    v0.destroy()
    eps.destroy()
    A.destroy()


@pytest.mark.parametrize("scale", [24, 26])
def test_rmat_converged_eigenvalues(gpu_ctx, scale):
    """The graph Laplacian of the symmetrised R-MAT graph of scale 24 (16.7M rows, 2.8e8 entries) and of scale 26
    (67.1M rows, 1.13e9 entries: BASELINE configuration 5's own matrix), nev = 32, ncv = 64, tol 1e-10, hashed start
    vector; the operator is binned by the library's own choice.  Generating the scale-26 matrix takes two minutes."""
    from slepc_amd import ksgpu, kseps as K, matgen
    fx = _fixture("converged_rmat%d.json" % scale)
    rp, ci, va = matgen.rmat_laplacian(scale)
    n = len(rp) - 1
    assert n == fx["rows"] and int(rp[-1]) == fx["entries"] and _checksum(rp, ci) == fx["colidx_checksum"]
    A = K.Mat.create_aij_hip(gpu_ctx, rp, ci, va)
    del rp, ci, va
    dev = ksgpu.DeviceCSR.__new__(ksgpu.DeviceCSR)
    dev.ctx, dev.lib, dev.h = gpu_ctx, gpu_ctx.lib, A.p.contents.data
    assert dev.describe().startswith("binned part"), dev.describe()
    eps = K.EPS(gpu_ctx.h)
    eps.set_operators(A)
    eps.set_problem_type(K.EPS_HEP)
    eps.set_dimensions(fx["nev"])
    eps.set_tolerances(fx["tol"], fx["restart_cap"])
    eps.solve()
    assert eps.dimensions()[1] == fx["ncv"]
    assert eps.reason == K.EPS_CONVERGED_TOL and eps.nconv >= fx["nev"]
    assert abs(eps.its - fx["restarts"]) <= 1, (eps.its, fx["restarts"])
    for i in range(fx["nev"]):
        lam = eps.get_eigenvalue(i)[0]
        want = fx["eigenvalues"][i][0]
        assert abs(lam - want) <= 1e-10 * abs(want), (i, lam, want)
    errs = [eps.compute_error(i) for i in range(fx["nev"])]
    assert max(errs) < 10 * fx["tol"], errs
    eps.destroy()
    A.destroy()
