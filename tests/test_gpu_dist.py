"""Row-partitioned runs of the product path.  The box has one GPU, so two or three
ranks share device 0 and the collectives of libksgpu are served by gloo callbacks
(host buffers); the partitioning, the diagonal/ghost split, the halo gather lists
and the ghost-block kernel are the ones the RCCL transport uses.  RCCL itself is
exercised on a single-rank communicator."""
import numpy as np
import pytest

from tests.dist_util import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,nev,ranks", [("lap2d", 6, 2), ("markov", 5, 2), ("lap3d", 5, 3), ("rmat", 4, 2), ("lap3d", 5, 8)])
def test_ranks_match_single(case, nev, ranks):
    one = run_ranks("gpu", case, 1)
    many = run_ranks("gpu", case, ranks)
    assert one["reason"] == many["reason"] == 1
    for a, b in zip(one["values"][:nev], many["values"][:nev]):
        assert abs(a[0] - b[0]) <= 1e-10 * max(abs(a[0]), 1e-300) and abs(a[1] - b[1]) <= 1e-10
    assert max(many["errors"]) < 1e-8
    x1, x2 = np.array(one["x0"]), np.array(many["x0"])
    s = np.sign(x1 @ x2)
    assert np.abs(s * x2 - x1).max() < 1e-6


@pytest.mark.parametrize("case,ranks", [("rmat_nnz", 2), ("rmat_nnz", 3), ("rmat_cost", 3), ("rmat_nnz", 8)])
def test_rmat_balanced_blocks_match_single(case, ranks):
    """The power-law operator split into row blocks balanced by entries (and by entries plus a
    cost per row): eigenvalues to 1e-10 of the single-rank run, and one product against the
    whole matrix on the host.  Most entries of every rank are ghost entries here; they go through
    the tile kernel with its own packed columns."""
    import re
    from scipy.sparse import csr_matrix
    from slepc_amd import matgen as mg
    one = run_ranks("gpu", case, 1)
    many = run_ranks("gpu", case, ranks)
    assert one["reason"] == many["reason"] == 1
    for a, b in zip(one["values"][:4], many["values"][:4]):
        assert abs(a[0] - b[0]) <= 1e-10 * abs(a[0])
    assert max(many["errors"]) < 1e-8
    x1, x2 = np.array(one["x0"]), np.array(many["x0"])
    assert np.abs(np.sign(x1 @ x2) * x2 - x1).max() < 1e-6
    rp, ci, va = mg.rmat_laplacian(13)
    N = 1 << 13
    g = np.arange(N)
    want = csr_matrix((va, ci, rp), shape=(N, N)) @ (np.cos(0.37 * g) + (g % 7 == 0))
    for res in (one, many):
        assert np.abs(np.array(res["y"]) - want).max() <= 1e-13 * np.abs(want).max()
    # every rank received ghosts and multiplied a ghost block; columns were packed in diagonal
    # and in ghost blocks (a rank whose block of columns is small packs nothing)
    ghosts = [int(re.search(r"ghost entries received per product: (\d+)", d).group(1)) for d in many["variant"]]
    assert min(ghosts) > 0
    assert all(int(re.search(r"ghost block: (\d+) entries", d).group(1)) > 0 for d in many["variant"])
    assert sum(int(re.search(r"packed hot columns: (\d+)", d).group(1)) for d in many["variant"]) > 0
    assert sum(int(re.search(r"(\d+) packed\)", d).group(1)) for d in many["variant"]) > 0


def test_ghost_columns_numbered_either_way():
    """The columns a rank's ghost block uses are numbered through a mark per column of the operator or,
    for operators with far more columns than the block has entries, by sorting them: both give the same
    halo plan and the same bits."""
    a = run_ranks("gpu", "rmat_nnz", 3)
    b = run_ranks("gpu", "rmat_nnz", 3, extra_env={"KSG_GHOST_SORT": "1"})
    assert a["reason"] == b["reason"] == 1
    assert a["variant"] == b["variant"]
    assert np.array_equal(np.array(a["y"]), np.array(b["y"]))
    assert a["values"] == b["values"]


@pytest.mark.parametrize("ranks", [1, 3, 8])
def test_rmat_product_exact_on_integer_data(ranks):
    """R-MAT scale 20 (10^6 rows, 1.7e7 entries) on 1, 3 and 8 ranks sharing the device, binned forms
    forced with the library's own cell and task sizes (the expand pass of the local and of the ghost
    blocks runs its main loop, the ghost block's tasks in the stages of the halo rounds): with
    integer-valued x and values every rank's rows of y equal the sequential CSR loop bit for bit."""
    import re
    r = run_ranks("gpu", "rmat_exact", ranks, extra_env={"KSG_SPMV_BINS": "2"})
    assert r["same"] == 1 and r["ranks"] == ranks, r["variant"]
    assert any(d.startswith("binned part") for d in r["variant"]), r["variant"]
    if ranks > 1:
        binned_ghost = [re.search(r"ghost block: (\d+) entries in (\d+) rows, (\d+) of them binned", d) for d in r["variant"]]
        assert any(m and int(m.group(3)) > 0 for m in binned_ghost), r["variant"]


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_rmat_binned_blocks_match_single(ranks):
    """The same split with both blocks of every rank in the binned form (KSG_OPT_BINS forced through
    the environment, small cells): the ghost block's x is the ghost buffer and its sums are added into
    the rows it touches; eigenvalues, the product and the rows read back against the plain single rank."""
    import re
    from scipy.sparse import csr_matrix
    from slepc_amd import matgen as mg
    env = {"KSG_SPMV_BINS": "2", "KSG_BINS_W": "512", "KSG_BINS_H": "128", "KSG_BINS_E": "1024", "KSG_BINS_T": "2", "KS_CHECK_CSR": "1"}
    one = run_ranks("gpu", "rmat_nnz", 1)
    many = run_ranks("gpu", "rmat_nnz", ranks, extra_env=env)
    assert one["reason"] == many["reason"] == 1 and many["csr_ok"]
    for a, b in zip(one["values"][:4], many["values"][:4]):
        assert abs(a[0] - b[0]) <= 1e-10 * abs(a[0])
    assert max(many["errors"]) < 1e-8
    rp, ci, va = mg.rmat_laplacian(13)
    N = 1 << 13
    g = np.arange(N)
    M = csr_matrix((va, ci, rp), shape=(N, N))
    xg = np.cos(0.37 * g) + (g % 7 == 0)
    assert np.all(np.abs(np.array(many["y"]) - M @ xg) <= 1e-14 * (np.abs(M) @ np.abs(xg)))
    # a rank whose block has fewer than 4096 columns (the hub rows' rank) keeps the plain forms
    assert any(d.startswith("binned part") for d in many["variant"]), many["variant"]
    binned_ghost = [re.search(r"ghost block: (\d+) entries in (\d+) rows, (\d+) of them binned", d) for d in many["variant"]]
    assert any(m and int(m.group(3)) > 0.5 * int(m.group(1)) for m in binned_ghost), many["variant"]


def test_halo_in_rounds_gives_the_same_bits():
    """A binned ghost block gets its halo in rounds -- piece s of every source's share of the ghost buffer
    in round s, all pairs of ranks at once -- and starts the expand tasks of a piece when it has arrived;
    the sums are formed afterwards as ever, so the product and the solve are bit for bit those of the
    single exchange (KSG_HALO_ROUNDS=0), for the default number of rounds and for an odd one."""
    env = {"KSG_SPMV_BINS": "2", "KSG_BINS_W": "512", "KSG_BINS_H": "128", "KSG_BINS_E": "1024", "KSG_BINS_T": "2"}
    for ranks, rounds in ((2, None), (3, "7"), (8, None)):
        a = run_ranks("gpu", "rmat_nnz", ranks, extra_env=dict(env, KSG_HALO_ROUNDS=rounds) if rounds else env)
        b = run_ranks("gpu", "rmat_nnz", ranks, extra_env=dict(env, KSG_HALO_ROUNDS="0"))
        assert any("halo in rounds" in d for d in a["variant"]), a["variant"]
        assert not any("halo in rounds" in d for d in b["variant"])
        assert a["reason"] == b["reason"] == 1
        assert np.array_equal(np.array(a["y"]), np.array(b["y"]))
        assert a["values"] == b["values"]


@pytest.mark.parametrize("case,ranks", [("svd_tall", 2), ("svd_wide", 3)])
def test_svd_ranks_match_single(case, ranks):
    """Rectangular operator with rows and columns split over the ranks, its transpose built
    by the exchange of entries, and the singular value solvers on top."""
    one = run_ranks("gpu", case, 1)
    many = run_ranks("gpu", case, ranks)
    assert one["transpose_err"] < 1e-12 and many["transpose_err"] < 1e-12
    for key in ("trlanczos", "trlanczos_oneside", "cross"):
        a, b = one[key], many[key]
        assert a["reason"] == b["reason"] == 1 and a["nconv"] >= 4 and b["nconv"] >= 4
        for i in range(4):
            assert abs(a["sigma"][i] - one["dense_sigma"][i]) <= 1e-10 * one["dense_sigma"][i]
            assert abs(b["sigma"][i] - one["dense_sigma"][i]) <= 1e-10 * one["dense_sigma"][i]
        assert max(b["errors"]) < 1e-7
        for w in ("u0", "v0"):
            x1, x2 = np.array(a[w]), np.array(b[w])
            s = np.sign(x1 @ x2)
            assert np.abs(s * x2 - x1).max() < 1e-6


@pytest.mark.parametrize("ranks", [2, 3])
def test_bv_inner_product_and_constraints_ranks_match_single(ranks):
    """Constraints, Gram-Schmidt in the inner product of a distributed matrix, the block Gram
    matrix and a group of split-phase reductions give the single-rank result on 2 and 3 ranks
    (the reference runs bv tests test3 and test6 on two processes)."""
    one = run_ranks("gpu", "bv_ip", 1)
    many = run_ranks("gpu", "bv_ip", ranks)
    assert one["nc"] == many["nc"] == 2
    Q1, Qm = np.array(one["Q"]), np.array(many["Q"])
    assert np.abs(Q1 - Qm).max() < 1e-12
    assert abs(one["bnorm0"] - many["bnorm0"]) < 1e-12 * one["bnorm0"]
    assert np.abs(np.array(one["norms"]) - np.array(many["norms"])).max() < 1e-12
    assert np.abs(np.array(one["split"]) - np.array(many["split"])).max() < 1e-11
    G = np.array(many["gram"])
    assert np.abs(G - np.eye(G.shape[0])).max() < 1e-13
    # B-orthonormal against the matrix itself, constraints deflated
    n = Q1.shape[1]
    B = 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    W = Qm @ B @ Qm.T
    assert np.abs(W - np.eye(W.shape[0])).max() < 1e-12
    # the tall-skinny QR: orthonormal columns, and the one factorisation with a positive
    # diagonal whatever the number of ranks
    T1, Tm = np.array(one["tsqr_Q"]), np.array(many["tsqr_Q"])
    assert np.abs(Tm @ Tm.T - np.eye(Tm.shape[0])).max() < 1e-14
    assert np.abs(T1 - Tm).max() < 1e-12
    R1, Rm = np.array(one["tsqr_R"]), np.array(many["tsqr_R"])
    assert np.abs(R1 - Rm).max() < 1e-12 * np.abs(R1).max() and (np.diag(Rm) > 0).all()
    g = np.arange(n, dtype=float)
    Y0 = np.array([np.cos((j + 2) * (g + 1) / 13.0) + 0.25 * (g == 5 * j) for j in range(Tm.shape[0])]).T
    assert np.abs(Tm.T @ Rm - Y0).max() < 1e-12


@pytest.mark.parametrize("case,ranks", [("gen_sinvert", 2), ("gen_shift", 3)])
def test_generalized_problem_ranks_match_single(case, ranks):
    """A generalized symmetric problem through shift-and-invert (CG) and through the default
    transformation (GMRES) with the operators split by rows: the single-rank result, and the
    dense solution of the pencil."""
    import scipy.linalg as sl
    from scipy.sparse import csr_matrix
    from slepc_amd import matgen as mg
    one = run_ranks("gpu", case, 1)
    many = run_ranks("gpu", case, ranks)
    rp, ci, va = mg.laplacian2d(19, 14)
    A = csr_matrix((va, ci, rp)).toarray()
    d = 2.0 + np.sin(np.arange(19 * 14))
    lam = sl.eigh(A, np.diag(d), eigvals_only=True)
    want = lam[:4] if case == "gen_sinvert" else lam[::-1][:4]
    for res in (one, many):
        assert res["reason"] == 1 and res["nconv"] >= 4 and res["ksp"] == ("cg" if case == "gen_sinvert" else "gmres")
        assert np.abs(np.array([v[0] for v in res["values"]]) - want).max() < 1e-8
        assert max(res["errors"]) < 1e-8
    x1, xm = np.array(one["x0"]), np.array(many["x0"])
    assert np.abs(np.sign(x1 @ xm) * xm - x1).max() < 1e-6
    assert abs((x1 * d) @ x1 - 1.0) < 1e-8                       # B-normalised


def test_rccl_single_rank_communicator():
    """dlopen of RCCL, ncclCommInitRank and the all-reduce on the compute stream."""
    from slepc_amd import ksgpu
    ctx = ksgpu.Context(0)
    ctx.comm_init_rccl(0, 1, ctx.unique_id())
    n = 100003
    V = ksgpu.DeviceBV(ctx, n, 3)
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, 3))
    V.from_numpy(A)
    q = V.dotvec(V.column(2))
    assert np.allclose(q, A.T @ A[:, 2], rtol=1e-12, atol=1e-9)
    assert abs(V.norm(1, 1) - np.linalg.norm(A[:, 1])) < 1e-10
    # the library's self-test: a sum, and a ring exchange (here to itself) through ncclSend / ncclRecv
    # in one group on the communication stream, the calls of the halo exchange
    ctx.comm_selftest()
    # leaving the communicator: single-rank again, reductions stay local
    ctx.comm_finalize()
    q2 = V.dotvec(V.column(2))
    assert np.array_equal(q, q2)
    V.destroy()
    ctx.close()
