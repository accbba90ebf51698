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
