"""The binned form of power-law operators, checked on the host: ksg_bins_host_check builds the same
arrays the device gets and walks them the way the two kernels do."""
import os

import numpy as np
import pytest
from scipy.sparse import csr_matrix

from slepc_amd import ksgpu, matgen as mg


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update({k: str(v) for k, v in env.items()})
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("W,H,E,T", [(64, 64, 1 << 20, 1), (1024, 256, 4096, 2), (16384, 2048, 65536, 4),
                                     (4096, 100, 700, 2), (512, 8192, 300, 3)])
@pytest.mark.parametrize("values", ["laplacian", "few", "any", "zeros"])
def test_binned_layout_on_host(W, H, E, T, values):
    rp, ci, va = mg.rmat_laplacian(13, 8, seed=3)
    n = len(rp) - 1
    rng = np.random.default_rng(17)
    offd = ci != np.repeat(np.arange(n), np.diff(rp))
    va = va.copy()
    if values == "few":
        va[offd] = rng.choice(np.array([-1.0, -0.5, 2.25, 1e-3]), size=int(offd.sum()))
    elif values == "any":
        va[offd] = rng.standard_normal(int(offd.sum()))
    elif values == "zeros":
        va[offd] = rng.choice(np.array([-1.0, 0.0]), size=int(offd.sum()))
    x = rng.standard_normal(n)
    y, info = _with_env({"KSG_BINS_W": W, "KSG_BINS_H": H, "KSG_BINS_E": E, "KSG_BINS_T": T, "KSG_BINS_TASK_GROUPS": 7},
                        lambda: ksgpu.bins_host_check(rp, ci, va, x))
    M = csr_matrix((va, ci, rp), shape=(n, n))
    assert info["binned"] > 0, info
    assert info["diagonal_folded"] == 1                  # one diagonal entry per row: a dense vector in the reduce pass
    assert info["binned"] + info["csr_entries"] + n == rp[-1]
    assert info["value_mode"] == {"laplacian": 0, "few": 1, "zeros": 1, "any": 2}[values]
    assert np.all(np.abs(y - M @ x) <= 1e-14 * (np.abs(M) @ np.abs(x) + 1e-300))


def test_binned_layout_rectangular_and_sparse():
    """A block without a diagonal (a ghost block's shape), columns not a multiple of the block
    width; an operator whose cells are all thin is left as a CSR."""
    rng = np.random.default_rng(2)
    rp, ci, va = mg.rmat_laplacian(12, 8, seed=5)
    n = len(rp) - 1
    keep = ci < n - 37                                   # cut the last columns off
    rows = np.repeat(np.arange(n), np.diff(rp))[keep]
    rp2 = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int64)
    ci2, va2 = ci[keep], va[keep]
    x = rng.standard_normal(n - 37)
    y, info = _with_env({"KSG_BINS_W": 256, "KSG_BINS_H": 128, "KSG_BINS_T": 2},
                        lambda: ksgpu.bins_host_check(rp2, ci2, va2, x, has_diag=False))
    M = csr_matrix((va2, ci2, rp2), shape=(n, n - 37))
    assert info["binned"] > 0 and info["diagonal_folded"] == 0
    assert info["binned"] + info["csr_entries"] == rp2[-1]
    assert np.all(np.abs(y - M @ x) <= 1e-14 * (np.abs(M) @ np.abs(x) + 1e-300))
    # a square block whose diagonal is not complete: its entries are ordinary entries
    hole = np.ones(len(ci), dtype=bool)
    hole[rp[5]:rp[5 + 1]] = ci[rp[5]:rp[6]] != 5
    rows3 = np.repeat(np.arange(n), np.diff(rp))[hole]
    rp3 = np.concatenate([[0], np.cumsum(np.bincount(rows3, minlength=n))]).astype(np.int64)
    x3 = rng.standard_normal(n)
    y3, info = _with_env({"KSG_BINS_W": 256, "KSG_BINS_H": 128, "KSG_BINS_T": 2}, lambda: ksgpu.bins_host_check(rp3, ci[hole], va[hole], x3))
    M3 = csr_matrix((va[hole], ci[hole], rp3), shape=(n, n))
    assert info["binned"] > 0 and info["diagonal_folded"] == 0 and info["binned"] + info["csr_entries"] == rp3[-1]
    assert np.all(np.abs(y3 - M3 @ x3) <= 1e-14 * (np.abs(M3) @ np.abs(x3) + 1e-300))
    lp, lc, lv = mg.laplacian2d(60, 60)
    xl = rng.standard_normal(3600)
    yl, info = _with_env({"KSG_BINS_W": 64, "KSG_BINS_H": 64, "KSG_BINS_T": 500}, lambda: ksgpu.bins_host_check(lp, lc, lv, xl))
    assert info["binned"] == 0 and np.array_equal(yl, csr_matrix((lv, lc, lp)) @ xl)


def test_binned_layout_random_blocks():
    """Random small blocks -- empty rows, rows of one entry, rows longer than a row block may be, columns
    beyond the last full column block, square with and without a complete diagonal, rectangular -- under
    random settings of the tuning variables: the walk of the binned arrays gives A x."""
    rng = np.random.default_rng(12345)
    built = 0
    for trial in range(120):
        n = int(rng.integers(1, 400))
        nc = n if trial % 3 else int(rng.integers(1, 600))
        # columns drawn with a skew so that some cells fill up
        dens = rng.choice([0.5, 2.0, 6.0, 20.0])
        nnz_t = int(min(n * nc, max(1, dens * n)))
        rows = rng.integers(0, n, nnz_t)
        cols = np.minimum((nc * rng.random(nnz_t) ** rng.choice([1.0, 2.0, 4.0])).astype(np.int64), nc - 1)
        if trial % 4 == 0:                                   # one hub row
            hub = int(rng.integers(0, n))
            extra = rng.integers(0, nc, int(rng.integers(50, 900)))
            rows = np.concatenate([rows, np.full(len(extra), hub)])
            cols = np.concatenate([cols, extra])
        if nc == n and trial % 2:                            # a complete diagonal
            rows = np.concatenate([rows, np.arange(n)])
            cols = np.concatenate([cols, np.arange(n)])
        key = np.unique(rows * nc + cols)
        rows, cols = key // nc, key % nc
        kind = int(rng.integers(0, 3))
        vals = rng.standard_normal(len(key)) if kind == 0 else (np.full(len(key), -1.0) if kind == 1 else rng.choice(np.array([0.5, -2.0, 0.0]), len(key)))
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int64)
        x = rng.standard_normal(nc)
        env = {"KSG_BINS_W": int(rng.choice([64, 128, 256, 1024])), "KSG_BINS_H": int(rng.choice([64, 100, 256])),
               "KSG_BINS_E": int(rng.choice([40, 200, 5000])), "KSG_BINS_T": int(rng.choice([1, 2, 3])),
               "KSG_BINS_TASK_GROUPS": int(rng.choice([1, 2, 9]))}
        y, info = _with_env(env, lambda: ksgpu.bins_host_check(rp, cols.astype(np.int32), vals, x, ncols=nc, has_diag=(nc == n)))
        M = csr_matrix((vals, cols, rp), shape=(n, nc))
        assert np.all(np.abs(y - M @ x) <= 1e-13 * (np.abs(M) @ np.abs(x) + 1e-300)), (trial, env, info)
        if info["binned"]:
            built += 1
            assert info["binned"] + info["csr_entries"] + (n if info["diagonal_folded"] else 0) == rp[-1]
    assert built > 40


@pytest.mark.parametrize("values", ["laplacian", "few", "any"])
def test_binned_layout_same_for_any_thread_count(values):
    """The build runs its passes over the entries on several host threads (KSG_HOST_THREADS) from a
    million entries on: the arrays -- and with them every bit of the product -- are the same for
    1, 3 and 8 threads, with long rows cut into pieces and a few values or arbitrary ones."""
    rp, ci, va = mg.rmat_laplacian(16, 10, seed=9)
    n = len(rp) - 1
    assert rp[-1] > 1 << 20
    rng = np.random.default_rng(5)
    offd = ci != np.repeat(np.arange(n), np.diff(rp))
    va = va.copy()
    if values == "few":
        va[offd] = rng.choice(np.array([-1.0, -0.5, 2.25, 1e-3]), size=int(offd.sum()))
    elif values == "any":
        va[offd] = rng.standard_normal(int(offd.sum()))
    x = rng.standard_normal(n)
    out = []
    for nt in (1, 3, 8):
        out.append(_with_env({"KSG_HOST_THREADS": nt, "KSG_BINS_W": 2048, "KSG_BINS_H": 512, "KSG_BINS_E": 3000, "KSG_BINS_T": 3},
                             lambda: ksgpu.bins_host_check(rp, ci, va, x)))
    M = csr_matrix((va, ci, rp), shape=(n, n))
    assert out[0][1]["binned"] > rp[-1] // 2
    assert np.all(np.abs(out[0][0] - M @ x) <= 1e-14 * (np.abs(M) @ np.abs(x) + 1e-300))
    for y, info in out[1:]:
        assert info == out[0][1]
        assert np.array_equal(y, out[0][0])
