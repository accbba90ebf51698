"""CPU oracle for the Krylov-Schur hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in this package is part of the product:
it may be imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker (or as
the thing timed as the CPU baseline), never as a compute path of
``slepc_amd``.

What it is: a restatement, in numpy + LAPACK, of the reference's CPU code
path for ``EPSSolve`` with the Krylov-Schur solver (SLEPc 3.7.3-dev), each
function citing the reference ``file:line`` it follows.  The reference
itself cannot be built in this environment (it needs PETSc, which is not in
the tree), so the oracle is pinned against the golden outputs that the
reference's own test-suite holds (``tests/golden/*.json``, extracted from
``src/**/examples/**/output/*.out`` by ``tests/golden/make_golden.py``) and
against analytic spectra.

The dense LAPACK routines the reference calls (``steqr``, ``lartg``, ``rot``,
``gehrd``, ``orghr``, ``hseqr``, ``trexc``, ``trevc``) are called here too,
through ctypes on the OpenBLAS that scipy bundles (``oracle/lapack.py``).
"""
