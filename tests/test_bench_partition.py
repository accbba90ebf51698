"""How bench.py cuts its workloads over the ranks (no GPU): the 512^3 Laplacian in slabs of
planes -- the configuration the strong-scaling target is quoted on --, the R-MAT operator in
row blocks balanced by entries, and the launcher's handling of a failing rank."""
import os
import subprocess
import sys

import numpy as np
import pytest

from slepc_amd import matgen as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("size", [1, 2, 4, 8])
def test_lap3d_512_slab_partition(size):
    g = 512
    plane = g * g
    args = bench.parse_args(["--gpus", str(size)] + (["--workload", "lap3d"] if size == 1 else []))
    assert args.workload == "lap3d" and args.grid == 512 and args.nev == 32 and not args.weak
    covered = 0
    for rank in range(size):
        name, N, row0, row1, dims, part = bench.workload_rows(args, size, rank)
        assert N == g ** 3 and dims == (g, g, g) and name.startswith("3-D 7-point Laplacian 512x512x512 (134217728 rows)")
        assert row0 == covered and (row1 - row0) == (g // size) * plane        # 64 planes per GPU at 8
        covered = row1
        # ghosts: exactly one plane of x from each neighbouring rank.  The rows of the two
        # boundary planes are the only ones that reference columns outside the block.
        need = set()
        for k0 in {row0, row1 - plane}:
            rp, ci, va = mg.laplacian3d(g, g, g, k0, k0 + plane)
            out = ci[(ci < row0) | (ci >= row1)]
            need.update(np.unique(out).tolist())
        need = np.array(sorted(need), dtype=np.int64)
        below, above = need[need < row0], need[need >= row1]
        assert below.size == (plane if rank > 0 else 0) and above.size == (plane if rank < size - 1 else 0)
        if below.size:
            assert below[0] == row0 - plane and below[-1] == row0 - 1
        if above.size:
            assert above[0] == row1 and above[-1] == row1 + plane - 1
        # an interior plane references nothing outside
        if row1 - row0 > 2 * plane:
            rp, ci, va = mg.laplacian3d(g, g, g, row0 + plane, row0 + 2 * plane)
            assert ci.min() >= row0 and ci.max() < row1
    assert covered == g ** 3


def test_default_workloads():
    a = bench.parse_args([])
    assert (a.workload, a.grid, a.nev, a.gpus) == ("ex2", 4096, 32, 1)
    a = bench.parse_args(["--gpus", "8"])
    assert (a.workload, a.grid, a.nev) == ("lap3d", 512, 32)
    a = bench.parse_args(["--workload", "markov"])
    assert (a.grid, a.nev) == (3000, 16)
    name, N, r0, r1, dims, part = bench.workload_rows(bench.parse_args(["--gpus", "2", "--workload", "ex2", "--weak", "--grid", "100"]), 2, 1)
    assert dims == (100, 200) and (r0, r1) == (10000, 20000) and N == 20000


def test_slab_rows_uneven():
    # 10 planes over 4 ranks: 3, 3, 2, 2
    got = [mg.slab_rows((7, 5, 10), 4, r) for r in range(4)]
    assert got == [(0, 105), (105, 210), (210, 280), (280, 350)]


@pytest.mark.parametrize("scale,ranks", [(14, 2), (16, 3), (17, 8)])
def test_rmat_blocks_balanced_by_entries(scale, ranks):
    rp, ci, va = mg.rmat_laplacian(scale)
    w = mg.rmat_row_weights(scale)
    starts = mg.balanced_row_blocks(w, ranks)
    assert starts[0] == 0 and starts[-1] == 1 << scale and all(a < b for a, b in zip(starts, starts[1:]))
    nnz = np.array([rp[starts[p + 1]] - rp[starts[p]] for p in range(ranks)], dtype=np.float64)
    assert nnz.max() / nnz.mean() < 1.12                      # the expected counts track the real ones
    # PETSc's equal-row split puts most entries of a power-law operator on the first rank
    eq = np.array([rp[mg.split_rows(1 << scale, ranks, p)[1]] - rp[mg.split_rows(1 << scale, ranks, p)[0]] for p in range(ranks)], dtype=np.float64)
    assert eq.max() / eq.mean() > nnz.max() / nnz.mean() + 0.2
    # with a cost per row the blocks move towards equal rows
    st2 = mg.balanced_row_blocks(w, ranks, row_weight=float(w.mean()))
    rows = np.diff(starts).astype(np.float64)
    rows2 = np.diff(st2).astype(np.float64)
    assert rows2.max() / rows2.mean() < rows.max() / rows.mean()
    # the model's total is the number of entries before duplicates are removed
    assert abs(w.sum() - (2 * 8 + 1) * (1 << scale)) < 1e-6 * w.sum()


def test_launcher_reports_a_failing_rank():
    """Without a GPU every rank fails when it opens its device: the launcher stops and returns
    non-zero instead of hanging or printing a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = "-1"
    env["ROCR_VISIBLE_DEVICES"] = "-1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "16", "--steps", "1", "--warmup", "1",
                        "--transport", "gloo", "--share-device"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180, env=env)
    assert r.returncode != 0
    assert b"stopping the other ranks" in r.stderr or b"exited with status" in r.stderr
    assert not [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]


@pytest.mark.parametrize("scale,ranks", [(14, 3), (16, 8)])
def test_rmat_hashed_partition_is_a_relabelling_and_balances_rows_and_entries(scale, ranks):
    """`bench.py --workload rmat --gpus N` (default partition): the vertices are dealt to the ranks by a
    hash of their number and relabelled rank by rank in their original order.  The generated operator is
    P A P^T of the natural one bit for bit, every rank has N/ranks rows to a percent and its share of the
    entries to a few percent (contiguous blocks of the natural order balance one of the two only), and
    the start vector of the relabelled problem is the default one permuted."""
    from scipy.sparse import csr_matrix
    N = 1 << scale
    new_of_old, old_of_new, starts = mg.hashed_relabeling(N, ranks)
    assert np.array_equal(np.sort(new_of_old), np.arange(N)) and np.array_equal(new_of_old[old_of_new], np.arange(N))
    assert starts[0] == 0 and starts[-1] == N and len(starts) == ranks + 1
    for r in range(ranks):                                   # a rank's vertices keep their order
        assert np.all(np.diff(old_of_new[starts[r]:starts[r + 1]]) > 0)
    rp, ci, va = mg.rmat_laplacian(scale)
    A = csr_matrix((va, ci, rp), shape=(N, N))
    B = A[old_of_new][:, old_of_new].tocsr()
    B.sort_indices()
    rp2, ci2, va2 = mg.rmat_laplacian(scale, relabel=new_of_old)
    assert np.array_equal(B.indptr, rp2) and np.array_equal(B.indices, ci2) and np.array_equal(B.data, va2)
    rows = np.diff(starts)
    ent = np.array([rp2[starts[r + 1]] - rp2[starts[r]] for r in range(ranks)])
    assert rows.max() <= 1.02 * N / ranks and rows.min() >= 0.98 * N / ranks
    assert ent.max() <= 1.12 * ent.mean(), ent                # the largest hub row alone is 1-2 % of a share at these scales
    cont = mg.balanced_row_blocks(mg.rmat_row_weights(scale), ranks, 0.0)
    assert np.diff(cont).max() > 2.0 * N / ranks               # what balancing the entries of the natural order does to the rows
    # the default start vector of the natural numbering, permuted
    v_nat = mg.hashed_start_vector(np.arange(N))
    assert np.array_equal(mg.hashed_start_vector(old_of_new), v_nat[old_of_new])
