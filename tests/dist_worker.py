"""One rank of a multi-process solve (launched by tests/test_dist*.py through
``torch.distributed.run``-style environment variables).

mode cpu: C host driver + test-only oracle plug-in, reductions over gloo.
mode gpu: the product path; every rank opens device 0 and the collectives of
          libksgpu are served by gloo callbacks on host buffers (RCCL refuses two
          ranks on one device, and the test box has one GPU).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    """Runs the case, then leaves the process group in order: a rank that exits with gloo's
    worker threads still alive can die in their destructors (std::terminate) after its work is done."""
    try:
        run()
    finally:
        import torch.distributed as td
        if td.is_available() and td.is_initialized():
            td.destroy_process_group()


def run():
    mode, case, out = sys.argv[1], sys.argv[2], sys.argv[3]
    hot_limit = None
    from slepc_amd import dist as kdist, kseps as K, matgen as mg
    rank, _, size = kdist.env_rank()
    dist = kdist.init_process_group() if size > 1 else None

    if case == "rmat_gen":
        return rmat_gen_main(out, rank, size, dist)
    if case == "rmat_exact":
        return rmat_exact_main(mode, out, rank, size, dist)
    if case.startswith("svd_"):
        return svd_main(mode, case, out, rank, size, dist)
    if case == "bv_ip":
        return bv_ip_main(mode, out, rank, size, dist)
    if case.startswith("gen_"):
        return generalized_main(mode, case, out, rank, size, dist)
    if case == "lap2d":
        gx, gy = 61, 47
        N = gx * gy
        row0, row1 = mg.split_rows(N, size, rank)
        csr = mg.laplacian2d(gx, gy, row0, row1)
        herm, kw = True, dict(nev=6, tol=1e-10)
    elif case == "lap3d":
        g = (17, 13, 11)
        N = g[0] * g[1] * g[2]
        row0, row1 = mg.split_rows(N, size, rank)
        csr = mg.laplacian3d(*g, row0=row0, row1=row1)
        herm, kw = True, dict(nev=5, tol=1e-10)
    elif case == "markov":
        m = 40
        N = m * (m + 1) // 2
        row0, row1 = mg.split_rows(N, size, rank)
        csr = mg.markov(m, row0, row1)
        herm, kw = False, dict(nev=5, tol=1e-12, which="largest_real")
    elif case == "rmat":
        N = 1 << 11
        row0, row1 = mg.split_rows(N, size, rank)
        csr = mg.rmat_laplacian(11, row0=row0, row1=row1)
        herm, kw = True, dict(nev=4, tol=1e-10)
    elif case in ("rmat_nnz", "rmat_cost"):
        # row blocks balanced by entries (rmat_nnz) or by entries plus a cost per row (rmat_cost),
        # small limit on the packed columns so that both blocks of every rank pack some
        scale = 13
        N = 1 << scale
        starts = mg.balanced_row_blocks(mg.rmat_row_weights(scale), size, 0.0 if case == "rmat_nnz" else 12.0)
        row0, row1 = starts[rank], starts[rank + 1]
        csr = mg.rmat_laplacian(scale, row0=row0, row1=row1)
        herm, kw = True, dict(nev=4, tol=1e-10)
        hot_limit = 256
    else:
        raise ValueError(case)
    n = row1 - row0
    rp, ci, va = csr

    if mode == "cpu":
        import torch
        from scipy.sparse import csr_matrix
        from tests import oracle_plugin as plug

        class GlooComm:
            def __init__(self):
                self.rank, self.size = rank, size

            def _red(self, a, op):
                t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
                if size > 1:
                    dist.all_reduce(t, op=op)
                return t.numpy()

            def allreduce_sum(self, a):
                return self._red(a, dist.ReduceOp.SUM if dist else None)

            def allreduce_max(self, a):
                return self._red(a, dist.ReduceOp.MAX if dist else None)

        plug.COMM[0] = GlooComm()
        A = csr_matrix((va, ci, rp), shape=(n, N))
        counts = [mg.split_rows(N, size, r)[1] - mg.split_rows(N, size, r)[0] for r in range(size)]

        def mv(x, y):
            if size > 1:
                parts = [torch.zeros(c, dtype=torch.float64) for c in counts]
                dist.all_gather(parts, torch.from_numpy(x.copy())) if len(set(counts)) == 1 else \
                    [dist.broadcast(parts[r] if r != rank else parts[r].copy_(torch.from_numpy(x)), src=r) for r in range(size)]
                xf = torch.cat(parts).numpy()
            else:
                xf = x
            y[:] = A @ xf

        Am = plug.shell_mat(mv, n, N, row0)
        eps = K.EPS(None)
        bvtype = "oracle"
        ctx = None
    else:
        from slepc_amd import ksgpu
        ctx = ksgpu.Context(0)
        if size > 1:
            kdist.init_comm(ctx, transport=os.environ.get("KS_TRANSPORT", "gloo"))
        if hot_limit is not None:
            ctx.set_option(ksgpu.OPT_HOT_COLS, hot_limit)
        Am = K.Mat.create_aij_hip(ctx, rp, ci, va, N=N, row0=row0)
        eps = K.EPS(ctx.h)
        bvtype = "hip"

    eps.set_operators(Am)
    eps.set_problem_type(K.EPS_HEP if herm else K.EPS_NHEP)
    eps.set_dimensions(kw["nev"])
    eps.set_tolerances(kw["tol"])
    if "which" in kw:
        eps.set_which(kw["which"])
    eps.set_bv_type(bvtype)
    if not herm:
        v0 = Am.create_vecs()
        g = np.arange(row0, row1)
        v0.set_values(((g == 0) | (g == 1)).astype(np.float64))
        eps.set_initial_space([v0])
    eps.solve()
    res = {"its": eps.its, "nconv": eps.nconv, "reason": eps.reason,
           "values": [eps.get_eigenvalue(i) for i in range(eps.nconv)],
           "errors": [eps.compute_error(i) for i in range(min(eps.nconv, kw["nev"]))]}
    xr = Am.create_vecs()
    eps.get_eigenpair(0, xr)
    local = xr.get_values()
    if size > 1:
        gathered = [None] * size
        dist.all_gather_object(gathered, local)
        res["x0"] = np.concatenate(gathered).tolist()
    else:
        res["x0"] = local.tolist()
    if mode == "gpu":
        # one product against the whole matrix on the host, and what the operator packed
        from slepc_amd import ksgpu  # noqa: F811
        dev = ksgpu.DeviceCSR.__new__(ksgpu.DeviceCSR)
        dev.ctx, dev.lib, dev.h = ctx, ctx.lib, Am.p.contents.data
        res["variant"] = dev.describe()
        if os.environ.get("KS_CHECK_CSR"):
            # the rows read back from the device (both blocks, binned parts included) are the input's
            dev._adopt(dev.h)
            r2, c2, v2 = dev.get_csr()
            ok = bool(np.array_equal(r2, rp) and np.array_equal(c2, ci) and np.array_equal(v2, va))
            if size > 1:
                oks = [None] * size
                dist.all_gather_object(oks, ok)
                ok = all(oks)
            res["csr_ok"] = ok
        res["rows"] = [row0, row1]
        xv, yv = Am.create_vecs(), Am.create_vecs()
        xv.set_values(np.cos(0.37 * np.arange(row0, row1)) + (np.arange(row0, row1) % 7 == 0))
        Am.mult(xv, yv)
        yl = yv.get_values()
        if size > 1:
            parts = [None] * size
            dist.all_gather_object(parts, yl)
            yl = np.concatenate(parts)
            descr = [None] * size
            dist.all_gather_object(descr, res["variant"])
            res["variant"] = descr
        res["y"] = yl.tolist()
        for o in (xv, yv):
            o.destroy()
    if rank == 0:
        with open(out, "w") as f:
            json.dump(res, f)
    if dist:
        dist.barrier()
    eps.destroy()
    xr.destroy()
    Am.destroy()
    if ctx:
        ctx.close()


def rmat_gen_main(out, rank, size, dist):
    """Every rank draws its share of the chunks of the R-MAT edge list and receives the edges of its
    rows: the rows must be those a single process generates, whatever the chunk size."""
    import numpy as np
    import torch
    from slepc_amd import matgen as mg
    scale = 12
    starts = mg.balanced_row_blocks(mg.rmat_row_weights(scale), size, 4.0)
    same = 1
    for chunk in (1 << 12, 1 << 13, 1 << 24):
        got = mg.rmat_laplacian_distributed(scale, starts, rank, dist, chunk=chunk)
        want = mg.rmat_laplacian(scale, row0=starts[rank], row1=starts[rank + 1], chunk=chunk)
        same &= int(all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(got, want)))
    # the hashed partition: the relabelled graph P A P^T, each rank its block of it
    new_of_old, old_of_new, hstarts = mg.hashed_relabeling(1 << scale, size)
    got_h = mg.rmat_laplacian_distributed(scale, hstarts, rank, dist, relabel=new_of_old)
    want_h = mg.rmat_laplacian(scale, row0=int(hstarts[rank]), row1=int(hstarts[rank + 1]), relabel=new_of_old)
    same &= int(all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(got_h, want_h)))
    whole = mg.rmat_laplacian(scale)                       # the chunk size does not change the matrix either
    a, b = starts[rank], starts[rank + 1]
    same &= int(np.array_equal(got[0], whole[0][a:b + 1] - whole[0][a]) and np.array_equal(got[1], whole[1][whole[0][a]:whole[0][b]]))
    flag = torch.tensor([same], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        with open(out, "w") as f:
            json.dump({"same": int(flag[0]), "ranks": size, "rows": [int(starts[r + 1] - starts[r]) for r in range(size)]}, f)


def rmat_exact_main(mode, out, rank, size, dist):
    """The distributed product of an R-MAT graph Laplacian of 10^6 rows with the library's own cell and
    task sizes (both blocks of a rank binned where it chooses so: KSG_SPMV_BINS=2 from the test) on
    integer-valued data: every rank's rows of y must equal the sequential CSR loop over its rows bit
    for bit, and the rows read back from the device are the input's."""
    import numpy as np
    import torch
    from oracle.matgen import csr_matvec_fast
    from slepc_amd import kseps as K, ksgpu, matgen as mg, dist as kdist
    assert mode == "gpu"
    scale = int(os.environ.get("KS_EXACT_SCALE", "20"))
    N = 1 << scale
    starts = mg.balanced_row_blocks(mg.rmat_row_weights(scale), size, 0.0)
    row0, row1 = int(starts[rank]), int(starts[rank + 1])
    rp, ci, va = mg.rmat_laplacian(scale, row0=row0, row1=row1)
    ctx = ksgpu.Context(0)
    if size > 1:
        kdist.init_comm(ctx, transport=os.environ.get("KS_TRANSPORT", "gloo"))
    Am = K.Mat.create_aij_hip(ctx, rp, ci, va, N=N, row0=row0)
    xg = (np.arange(N, dtype=np.int64) * 2654435761 % 1021 - 510).astype(np.float64)
    want = np.zeros(row1 - row0)
    csr_matvec_fast(rp, ci, va, xg, want)
    xv, yv = Am.create_vecs(), Am.create_vecs()
    xv.set_values(xg[row0:row1])
    same = 1
    for _ in range(2):
        yv.set_values(np.full(row1 - row0, np.nan))
        Am.mult(xv, yv)
        same &= int(np.array_equal(yv.get_values(), want))
    dev = ksgpu.DeviceCSR.__new__(ksgpu.DeviceCSR)
    dev.ctx, dev.lib, dev.h = ctx, ctx.lib, Am.p.contents.data
    descr = dev.describe()
    dev._adopt(dev.h)
    r2, c2, v2 = dev.get_csr()
    same &= int(np.array_equal(r2, rp) and np.array_equal(c2, ci) and np.array_equal(v2, va))
    flag = torch.tensor([same], dtype=torch.int64)
    variants = [descr]
    if size > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        variants = [None] * size
        dist.all_gather_object(variants, descr)
    if rank == 0:
        with open(out, "w") as f:
            json.dump({"same": int(flag[0]), "ranks": size, "variant": variants}, f)
    if dist:
        dist.barrier()
    for o in (xv, yv):
        o.destroy()
    Am.destroy()
    ctx.close()


def bv_ip_main(mode, out, rank, size, dist):
    """Rows of a BV and of its inner-product matrix split over the ranks (the reference runs
    its bv tests test3 and test6 on two processes): constraints, Gram-Schmidt in the B-inner
    product, the block Gram matrix and a group of split-phase reductions."""
    import ctypes as C
    from slepc_amd import kseps as K, ksgpu, matgen as mg, dist as kdist
    from tests.common import tridiag_csr, dp
    assert mode == "gpu"
    N, k, nc = 211, 7, 2
    row0, row1 = mg.split_rows(N, size, rank)
    n = row1 - row0
    rp, ci, va = tridiag_csr(N)
    lo, hi = rp[row0], rp[row1]
    ctx = ksgpu.Context(0)
    if size > 1:
        kdist.init_comm(ctx, transport=os.environ.get("KS_TRANSPORT", "gloo"))
    B = K.Mat.create_aij_hip(ctx, rp[row0:row1 + 1] - lo, ci[lo:hi], va[lo:hi], N=N, row0=row0)
    lib = K.load()
    t = K.Vec.create_hip(ctx, n, N, row0)
    t.set_values(np.cos(np.arange(row0, row1) / 3.0))
    X = K.BVP()
    K.check(lib.BVCreate(None, C.byref(X)))
    K.check(lib.BVSetSizesFromVec(X, t.p, k))
    K.check(lib.BVSetType(X, b"hip"))
    K.check(lib.BVSetMatrix(X, B.p, 0))
    g = np.arange(row0, row1, dtype=float)
    cons = []
    for j in range(nc):
        v = K.Vec.create_hip(ctx, n, N, row0)
        v.set_values(np.cos((j + 1) * g / 7.0))
        cons.append(v)
    ncio = C.c_int(nc)
    K.check(lib.BVInsertConstraints(X, C.byref(ncio), (K.VecP * nc)(*[v.p for v in cons])))
    col = K.VecP()

    def set_col(j, a):
        K.check(lib.BVGetColumn(X, j, C.byref(col)))
        K.check(lib.VecSetValuesHost(col, dp(np.ascontiguousarray(a))))
        K.check(lib.BVRestoreColumn(X, j, C.byref(col)))

    def get_col(j):
        a = np.empty(n)
        K.check(lib.BVGetColumn(X, j, C.byref(col)))
        K.check(lib.VecGetValuesHost(col, dp(a)))
        K.check(lib.BVRestoreColumn(X, j, C.byref(col)))
        return a
    for j in range(k):
        set_col(j, np.sin((j + 1) * (g + 1) / 11.0) + (g == 3 * j))
    nrm = C.c_double()
    K.check(lib.BVNormColumn(X, 0, K.NORM_2, C.byref(nrm)))
    res = {"nc": ncio.value, "bnorm0": nrm.value, "norms": []}
    for j in range(k):
        K.check(lib.BVOrthogonalizeColumn(X, j, None, C.byref(nrm), None))
        K.check(lib.BVScaleColumn(X, j, 1.0 / nrm.value))
        res["norms"].append(nrm.value)
    M = np.zeros((k, k), order="F")
    K.check(lib.BVDot(X, X, dp(M), k))
    res["gram"] = M.tolist()
    # one group: coefficients against the first columns, a B-norm and a plain dot product
    dv, d, bn = np.zeros(3), C.c_double(), C.c_double()
    K.check(lib.BVSetActiveColumns(X, 0, 3))
    K.check(lib.BVDotVecBegin(X, t.p, dp(dv)))
    K.check(lib.BVNormColumnBegin(X, 5, K.NORM_2, C.byref(bn)))
    K.check(lib.VecDotBegin(t.p, t.p, C.byref(d)))
    K.check(lib.BVDotVecEnd(X, t.p, dp(dv)))
    K.check(lib.BVNormColumnEnd(X, 5, K.NORM_2, C.byref(bn)))
    K.check(lib.VecDotEnd(t.p, t.p, C.byref(d)))
    K.check(lib.BVSetActiveColumns(X, 0, k))
    res["split"] = dv.tolist() + [bn.value, d.value]
    cols = [get_col(j) for j in range(-ncio.value, k)]
    if size > 1:
        full = []
        for a in cols:
            parts = [None] * size
            dist.all_gather_object(parts, a)
            full.append(np.concatenate(parts))
        cols = full
    res["Q"] = [a.tolist() for a in cols]
    # tall-skinny QR of a second basis (standard inner product, no constraints): the reference
    # runs bv test11 with -bv_orthog_block tsqr on two processes
    kq = 6
    Y = K.BVP()
    K.check(lib.BVCreate(None, C.byref(Y)))
    K.check(lib.BVSetSizesFromVec(Y, t.p, kq))
    K.check(lib.BVSetType(Y, b"hip"))
    K.check(lib.BVSetOrthogonalization(Y, K.BV_ORTHOG_CGS, K.BV_ORTHOG_REFINE_IFNEEDED, 0.0, K.BV_ORTHOG_BLOCK_TSQR))
    for j in range(kq):
        K.check(lib.BVGetColumn(Y, j, C.byref(col)))
        K.check(lib.VecSetValuesHost(col, dp(np.ascontiguousarray(np.cos((j + 2) * (g + 1) / 13.0) + 0.25 * (g == 5 * j)))))
        K.check(lib.BVRestoreColumn(Y, j, C.byref(col)))
    Rq = np.zeros((kq, kq), order="F")
    K.check(lib.BVOrthogonalize(Y, dp(Rq), kq))
    qcols = []
    for j in range(kq):
        a = np.empty(n)
        K.check(lib.BVGetColumn(Y, j, C.byref(col)))
        K.check(lib.VecGetValuesHost(col, dp(a)))
        K.check(lib.BVRestoreColumn(Y, j, C.byref(col)))
        if size > 1:
            parts = [None] * size
            dist.all_gather_object(parts, a)
            a = np.concatenate(parts)
        qcols.append(a.tolist())
    res["tsqr_Q"], res["tsqr_R"] = qcols, Rq.tolist()
    K.check(lib.BVDestroy(C.byref(Y)))
    if rank == 0:
        with open(out, "w") as f:
            json.dump(res, f)
    if dist:
        dist.barrier()
    K.check(lib.BVDestroy(C.byref(X)))
    for v in cons + [t]:
        v.destroy()
    B.destroy()
    ctx.close()


def generalized_main(mode, case, out, rank, size, dist):
    """ex13-like generalized symmetric problem (B diagonal, varying) on row-distributed operators:
    gen_sinvert -- shift-and-invert at a target inside the spectrum's lower end, A - sigma B merged
    on every rank from its rows, CG/GMRES with the halo exchange in every product;
    gen_shift   -- the default transformation, systems with B."""
    from slepc_amd import kseps as K, ksgpu, matgen as mg, dist as kdist
    assert mode == "gpu"
    gx, gy = 19, 14
    N = gx * gy
    row0, row1 = mg.split_rows(N, size, rank)
    n = row1 - row0
    rp, ci, va = mg.laplacian2d(gx, gy, row0, row1)
    d = 2.0 + np.sin(np.arange(row0, row1))
    ctx = ksgpu.Context(0)
    if size > 1:
        kdist.init_comm(ctx, transport=os.environ.get("KS_TRANSPORT", "gloo"))
    A = K.Mat.create_aij_hip(ctx, rp, ci, va, N=N, row0=row0)
    B = K.Mat.create_aij_hip(ctx, np.arange(n + 1, dtype=np.int64), np.arange(row0, row1, dtype=np.int32), d, N=N, row0=row0)
    eps = K.EPS(ctx.h)
    eps.set_operators(A, B)
    eps.set_problem_type(K.EPS_GHEP)
    eps.set_dimensions(4)
    eps.set_tolerances(1e-10)
    st = eps.st()
    if case == "gen_sinvert":
        st.set_type("sinvert")
        eps.set_target(0.0)
        eps.set_which("target_magnitude")
        st.set_ksp_type("cg")
    else:
        st.set_ksp_type("gmres")
    eps.solve()
    res = {"its": eps.its, "nconv": eps.nconv, "reason": eps.reason, "ksp": st.get_ksp_type(),
           "values": [eps.get_eigenvalue(i) for i in range(min(eps.nconv, 4))],
           "errors": [eps.compute_error(i) for i in range(min(eps.nconv, 4))]}
    xr = A.create_vecs()
    eps.get_eigenpair(0, xr)
    local = xr.get_values()
    if size > 1:
        gathered = [None] * size
        dist.all_gather_object(gathered, local)
        local = np.concatenate(gathered)
    res["x0"] = local.tolist()
    if rank == 0:
        with open(out, "w") as f:
            json.dump(res, f)
    if dist:
        dist.barrier()
    xr.destroy()
    eps.destroy()
    A.destroy()
    B.destroy()
    ctx.close()


def svd_main(mode, case, out, rank, size, dist):
    """Singular values of a rectangular operator: rows and columns both split over the ranks,
    the transpose built by the library (an exchange of entries between the ranks)."""
    from scipy.sparse import csr_matrix
    from slepc_amd import kseps as K, ksgpu, matgen as mg, dist as kdist
    assert mode == "gpu"
    M, N = (420, 300) if case.endswith("tall") else (260, 410)
    rng = np.random.default_rng(17)                       # the same matrix on every rank
    D = np.where(rng.random((M, N)) < 0.03, rng.standard_normal((M, N)), 0.0)
    D[np.arange(min(M, N)), np.arange(min(M, N))] += 3.0 + np.arange(min(M, N)) / 50.0
    A = csr_matrix(D)
    A.sort_indices()
    row0, row1 = mg.split_rows(M, size, rank)
    col0, col1 = mg.split_rows(N, size, rank)
    L = A[row0:row1]
    ctx = ksgpu.Context(0)
    if size > 1:
        kdist.init_comm(ctx, transport=os.environ.get("KS_TRANSPORT", "gloo"))
    Am = K.Mat.create_aij_hip_rect(ctx, L.indptr, L.indices, L.data, N, M=M, row0=row0, ncols_local=col1 - col0, col0=col0)
    res = {}
    # the transposed operator against the host, entry for entry
    T = Am.transpose()
    y = T.create_vecs_left()
    x = T.create_vecs()
    xs = np.sin(np.arange(M, dtype=float))
    x.set_values(xs[row0:row1])
    T.mult(x, y)
    ref = (csr_matrix(A.T) @ xs)[col0:col1]
    res["transpose_err"] = float(np.abs(y.get_values() - ref).max())
    for o in (x, y):
        o.destroy()
    T.destroy()
    for solver, oneside in (("trlanczos", False), ("trlanczos", True), ("cross", False)):
        svd = K.SVD(ctx.h)
        svd.set_operator(Am)
        svd.set_type(solver)
        svd.set_dimensions(4)
        svd.set_tolerances(1e-10)
        if oneside:
            svd.set_oneside(True)
        svd.solve()
        key = solver + ("_oneside" if oneside else "")
        res[key] = {"its": svd.its, "nconv": svd.nconv, "reason": svd.reason,
                    "sigma": [svd.get_singular_triplet(i) for i in range(min(svd.nconv, 4))],
                    "errors": [svd.compute_error(i) for i in range(min(svd.nconv, 4))]}
        u, v = Am.create_vecs_left(), Am.create_vecs()
        svd.get_singular_triplet(0, u, v)
        lu, lv = u.get_values(), v.get_values()
        if size > 1:
            gu, gv = [None] * size, [None] * size
            dist.all_gather_object(gu, lu)
            dist.all_gather_object(gv, lv)
            lu, lv = np.concatenate(gu), np.concatenate(gv)
        res[key]["u0"], res[key]["v0"] = lu.tolist(), lv.tolist()
        for o in (u, v):
            o.destroy()
        svd.destroy()
    if rank == 0:
        ref = np.linalg.svd(D, compute_uv=False)[:4]
        res["dense_sigma"] = ref.tolist()
        with open(out, "w") as f:
            json.dump(res, f)
    if dist:
        dist.barrier()
    Am.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
