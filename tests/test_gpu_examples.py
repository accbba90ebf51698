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
