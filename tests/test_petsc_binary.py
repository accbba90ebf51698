"""PETSc binary matrix files (the wire format either side of the operator: MatLoad /
MatView on a binary viewer).  The C reader of the host library against the numpy
restatement on the data files the reference ships for its ex4/ex7 tests
(share/slepc/datafiles/matrices, copied as fixtures under tests/golden/matrices), the
writer round trip, the error behaviour, and the reference's golden output of ex4
(src/eps/examples/tutorials/output/ex4_1.out) reproduced by the oracle solver."""
import os
import subprocess

import numpy as np
import pytest
from scipy.sparse import csr_matrix

from oracle import eps as oeps, petscbin
from slepc_amd import kseps as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATDIR = os.path.join(ROOT, "tests", "golden", "matrices")

# (rows, nonzeros) as the readme beside the reference's files states them
FILES = {"rdb200": (200, 1120), "bfw62a": (62, 450), "bfw62b": (62, 342)}

EX4_1 = """
Eigenproblem stored in file.

 Reading REAL matrix from a binary file...
 Number of iterations of the method: 5
 Solution method: krylovschur

 Number of requested eigenvalues: 4
 Stopping condition: tol=1e-08, maxit=100
 All requested eigenvalues computed up to the required tolerance:
     -35.00752, -34.10419, -33.20131, -32.68111

"""


@pytest.mark.parametrize("name", list(FILES))
def test_c_reader_matches_numpy_reader(name):
    path = os.path.join(MATDIR, name + ".petsc")
    M, N, rp, ci, va = K.read_petsc_binary(path)
    oM, oN, orp, oci, ova = petscbin.read(path)
    assert (M, N) == (oM, oN) == (FILES[name][0],) * 2 and rp[-1] == FILES[name][1]
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
    assert va.tobytes() == ova.tobytes()
    for r in range(M):
        assert np.all(np.diff(ci[rp[r]:rp[r + 1]]) > 0)


def test_write_read_round_trip_and_row_sorting(tmp_path):
    rng = np.random.default_rng(5)
    M, N = 37, 41
    dense = np.where(rng.random((M, N)) < 0.15, rng.standard_normal((M, N)), 0.0)
    dense[3, :] = 0.0                                   # an empty row
    A = csr_matrix(dense)
    A.sort_indices()
    # rows written in descending column order: the reader must hand them back ascending
    rp = A.indptr.astype(np.int64)
    ci = np.concatenate([A.indices[rp[r]:rp[r + 1]][::-1] for r in range(M)]).astype(np.int32)
    va = np.concatenate([A.data[rp[r]:rp[r + 1]][::-1] for r in range(M)])
    p1, p2 = str(tmp_path / "c.petsc"), str(tmp_path / "np.petsc")
    K.write_petsc_binary(p1, M, N, rp, ci, va)
    petscbin.write(p2, M, N, rp, ci, va)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    m, n, rp2, ci2, va2 = K.read_petsc_binary(p1)
    assert (m, n) == (M, N) and np.array_equal(rp2, rp)
    assert np.array_equal(ci2, A.indices) and va2.tobytes() == A.data.tobytes()
    # empty matrix
    K.write_petsc_binary(p1, 0, 0, np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0))
    assert K.read_petsc_binary(p1)[:2] == (0, 0)


def test_reader_errors(tmp_path):
    good = open(os.path.join(MATDIR, "bfw62b.petsc"), "rb").read()

    def code(data):
        p = str(tmp_path / "bad.petsc")
        open(p, "wb").write(data)
        with pytest.raises(K.KsError) as e:
            K.read_petsc_binary(p)
        return e.value.code

    assert code(b"\x00\x00\x00\x01" + good[4:]) == 79          # wrong class id: PETSC_ERR_FILE_UNEXPECTED
    assert code(good[:-5]) == 66                                # truncated: PETSC_ERR_FILE_READ
    assert code(good[:10]) == 66
    bad = bytearray(good)
    off = 16 + 4 * 62
    bad[off:off + 4] = (62).to_bytes(4, "big")                  # a column index outside the matrix
    assert code(bytes(bad)) == 79
    bad = bytearray(good)
    bad[16:20] = (7).to_bytes(4, "big") if int.from_bytes(good[16:20], "big") != 7 else (8).to_bytes(4, "big")
    assert code(bytes(bad)) == 79                               # row lengths disagree with the header
    with pytest.raises(K.KsError) as e:
        K.read_petsc_binary(str(tmp_path / "missing.petsc"))
    assert e.value.code == 65                                   # PETSC_ERR_FILE_OPEN


def test_oracle_reproduces_ex4_golden():
    """runex4_1: ./ex4 -file rdb200.petsc -eps_nev 4 -terse (tutorials/makefile:188)."""
    M, N, rp, ci, va = petscbin.read(os.path.join(MATDIR, "rdb200.petsc"))
    A = csr_matrix((va, ci, rp), shape=(M, N))

    def mv(x, y):
        y[:] = A @ x
    e = oeps.EPS(mv, M, hermitian=False, nev=4).solve()
    assert e.its == 5 and e.nconv >= 4
    assert ", ".join("%.5f" % e.get_eigenvalue(i)[0] for i in range(4)) in EX4_1
    assert all(e.get_eigenvalue(i)[1] == 0.0 for i in range(4))


@pytest.mark.gpu
def test_ex4_matches_reference_output():
    exe = os.path.join(ROOT, "examples", "bin", "ex4")
    if not os.path.exists(exe):
        pytest.fail("%s not built: run `python __graft_entry__.py build`" % exe)
    r = subprocess.run([exe, "-file", os.path.join(MATDIR, "rdb200.petsc"), "-eps_nev", "4", "-terse"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    assert r.stdout.decode() == EX4_1


@pytest.mark.gpu
def test_mat_load_on_device(gpu_ctx):
    """MatLoad onto the device: the loaded operator applied to a vector equals the host product."""
    path = os.path.join(MATDIR, "bfw62a.petsc")
    M, N, rp, ci, va = petscbin.read(path)
    A = K.Mat.load_aij_hip(gpu_ctx, path)
    x, y = A.create_vecs(), A.create_vecs()
    xs = np.arange(1, M + 1, dtype=np.float64) / 7.0
    x.set_values(xs)
    A.mult(x, y)
    ref = np.array([np.sum(va[rp[r]:rp[r + 1]] * xs[ci[rp[r]:rp[r + 1]]]) for r in range(M)])
    assert np.abs(y.get_values() - ref).max() <= 1e-13 * np.abs(ref).max()
    with pytest.raises(K.KsError):
        K.Mat.load_aij_hip(gpu_ctx, os.path.join(MATDIR, "nothing.petsc"))
    for v in (x, y):
        v.destroy()
    A.destroy()
