"""The C example programs (examples/ex2.c, examples/ex5.c: the reference's tutorials written
against include/kseps.h) run with the arguments of the reference's own test targets
(src/eps/examples/tutorials/makefile: runex2_1 ``-n 72 -eps_nev 4 -eps_ncv 20``, runex5_1
``-eps_nev 4 -eps_largest_real``) and must print what the reference's golden files
hold (output/ex2_1.out, output/ex5_1.out), line for line."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EX2_1 = """
2-D Laplacian Eigenproblem, N=5184 (72x72 grid)

 Solution method: krylovschur

 Number of requested eigenvalues: 4
 All requested eigenvalues computed up to the required tolerance:
     7.99630, 7.99074, 7.98519, 7.98150

"""

EX5_1 = """
Markov Model, N=120 (m=15)

 Solution method: krylovschur

 Number of requested eigenvalues: 4
 All requested eigenvalues computed up to the required tolerance:
     1.00000, 0.97137, 0.90423, 0.85714

"""


def _run(name, *args):
    exe = os.path.join(ROOT, "examples", "bin", name)
    if not os.path.exists(exe):
        pytest.fail("%s not built: run `python __graft_entry__.py build`" % exe)
    r = subprocess.run([exe, *args], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    return r.stdout.decode()


def test_ex2_matches_reference_output():
    assert _run("ex2", "-n", "72", "-eps_nev", "4", "-eps_ncv", "20") == EX2_1


def test_ex5_matches_reference_output():
    assert _run("ex5", "-eps_nev", "4", "-eps_largest_real") == EX5_1


def test_ex2_arnoldi_same_eigenvalues():
    """runtest-style: the same problem with -eps_type arnoldi prints the same eigenvalues."""
    out = _run("ex2", "-n", "30", "-m", "20", "-eps_nev", "3", "-eps_type", "arnoldi")
    ref = _run("ex2", "-n", "30", "-m", "20", "-eps_nev", "3")
    assert "Solution method: arnoldi" in out and "Solution method: krylovschur" in ref
    assert out.splitlines()[-2] == ref.splitlines()[-2] and "All requested eigenvalues computed" in out


EX13_1 = """
Generalized Symmetric Eigenproblem, N=100 (10x10 grid), null(B)=0

 Solution method: krylovschur

 Number of requested eigenvalues: 4
 All requested eigenvalues computed up to the required tolerance:
     0.04051, 0.09963, 0.09963, 0.15875

"""


def test_ex13_matches_reference_output():
    """runex13_1: ``-eps_nev 4 -eps_ncv 22`` (output/ex13_1.out), with each linear solver."""
    assert _run("ex13", "-eps_nev", "4", "-eps_ncv", "22") == EX13_1
    assert _run("ex13", "-eps_nev", "4", "-eps_ncv", "22", "-st_ksp_type", "cg") == EX13_1


def test_ex7_matches_reference_output():
    """runex7_1: ``-f1 bfw62a.petsc -f2 bfw62b.petsc -eps_nev 4`` (output/ex7_1.out).  The
    restart and linear-iteration counts depend on the start vector (a PETSc random vector in
    the reference) and one eigenvalue sits within 1e-11 of a rounding boundary of the five
    printed decimals, so the values are compared as numbers."""
    import re
    mat = os.path.join(ROOT, "tests", "golden", "matrices")
    out = _run("ex7", "-f1", os.path.join(mat, "bfw62a.petsc"), "-f2", os.path.join(mat, "bfw62b.petsc"), "-eps_nev", "4")
    lines = out.splitlines()
    assert lines[1] == "Generalized eigenproblem stored in file." and lines[3] == " Reading REAL matrices from binary files..."
    assert re.match(r" Number of iterations of the method: \d+$", lines[4])
    m = re.match(r" Number of linear iterations of the method: (\d+)$", lines[5])
    assert m and int(m.group(1)) > 0
    assert lines[6:11] == [" Solution method: krylovschur", "", " Number of requested eigenvalues: 4",
                           " Stopping condition: tol=1e-08, maxit=100",
                           " All requested eigenvalues computed up to the required tolerance:"]
    got = [complex(v.replace("i", "j")) for v in lines[11].strip().split(", ")]
    want = [complex(-243874.97870, 6999.66927), complex(-243874.97870, -6999.66927), -212991.49278, -199807.74659]
    assert all(abs(a - b) <= 1.1e-5 for a, b in zip(got, want))
