"""World-size-2 runs on the CPU (gloo): the C host driver with row-partitioned
vectors must reproduce the single-rank result -- same eigenvalues to 1e-10,
residuals within tolerance, eigenvector equal up to sign -- for every partition,
because the start vector depends only on the global row index."""
import numpy as np
import pytest

from tests.dist_util import run_ranks


@pytest.mark.parametrize("case,nev", [("lap2d", 6), ("markov", 5)])
def test_two_ranks_match_one(case, nev):
    one = run_ranks("cpu", case, 1)
    two = run_ranks("cpu", case, 2)
    assert one["reason"] == two["reason"] == 1
    assert min(one["nconv"], two["nconv"]) >= nev
    for a, b in zip(one["values"][:nev], two["values"][:nev]):
        assert abs(a[0] - b[0]) <= 1e-10 * max(abs(a[0]), 1e-300) and abs(a[1] - b[1]) <= 1e-10
    assert max(two["errors"]) < 1e-8
    x1, x2 = np.array(one["x0"]), np.array(two["x0"])
    s = np.sign(x1 @ x2)
    assert np.abs(s * x2 - x1).max() < 1e-6


def test_three_ranks_uneven_rows():
    one = run_ranks("cpu", "lap3d", 1)
    three = run_ranks("cpu", "lap3d", 3)
    for a, b in zip(one["values"][:5], three["values"][:5]):
        assert abs(a[0] - b[0]) <= 1e-10 * abs(a[0])


@pytest.mark.parametrize("ranks", [2, 3])
def test_rmat_rows_generated_by_all_ranks_together(ranks):
    """`bench.py --workload rmat --gpus N`: every rank draws 1/N of the edge list (chunks of the one
    random stream, each reached by skipping ahead) and the edges travel to the owners of their rows;
    the rows are bit for bit those of the single-process generator."""
    r = run_ranks("cpu", "rmat_gen", ranks)
    assert r["same"] == 1 and r["ranks"] == ranks and sum(r["rows"]) == 1 << 12
