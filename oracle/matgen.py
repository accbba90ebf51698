"""Loop-for-loop restatement of the matrix generators of the reference's example
programs, producing CSR with sorted column indices per row (what PETSc's AIJ
assembly yields).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Pure-Python loops: use
for small sizes only; the vectorised generators of ``slepc_amd.matgen`` are
checked against these bit for bit.
"""
import numpy as np


def _assemble(n, rows):
    """rows: list of dict col->val (INSERT_VALUES semantics).  Returns
    (rowptr int64, colidx int32, val float64) with sorted columns."""
    rowptr = np.zeros(n + 1, dtype=np.int64)
    cols, vals = [], []
    for i, r in enumerate(rows):
        ks = sorted(r)
        cols.extend(ks)
        vals.extend(r[c] for c in ks)
        rowptr[i + 1] = rowptr[i] + len(ks)
    return rowptr, np.asarray(cols, dtype=np.int32), np.asarray(vals, dtype=np.float64)


def laplacian1d(n):
    """``src/eps/examples/tutorials/ex1.c:54-58``."""
    rows = []
    for i in range(n):
        r = {}
        if i > 0:
            r[i - 1] = -1.0
        if i < n - 1:
            r[i + 1] = -1.0
        r[i] = 2.0
        rows.append(r)
    return _assemble(n, rows)


def laplacian2d(n, m=None):
    """``src/eps/examples/tutorials/ex2.c:56-63`` (N = n*m, II = i*n + j)."""
    if m is None:
        m = n
    N = n * m
    rows = []
    for II in range(N):
        i = II // n
        j = II - i * n
        r = {}
        if i > 0:
            r[II - n] = -1.0
        if i < m - 1:
            r[II + n] = -1.0
        if j > 0:
            r[II - 1] = -1.0
        if j < n - 1:
            r[II + 1] = -1.0
        r[II] = 4.0
        rows.append(r)
    return _assemble(N, rows)


def laplacian3d(mx, my=None, mz=None):
    """``src/eps/examples/tutorials/ex19.c:70-86`` in natural ordering
    idx = i + mx*(j + my*k)."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    N = mx * my * mz
    rows = []
    for k in range(mz):
        for j in range(my):
            for i in range(mx):
                def idx(a, b, c):
                    return a + mx * (b + my * c)
                r = {idx(i, j, k): 6.0}
                if k > 0:
                    r[idx(i, j, k - 1)] = -1.0
                if j > 0:
                    r[idx(i, j - 1, k)] = -1.0
                if i > 0:
                    r[idx(i - 1, j, k)] = -1.0
                if i < mx - 1:
                    r[idx(i + 1, j, k)] = -1.0
                if j < my - 1:
                    r[idx(i, j + 1, k)] = -1.0
                if k < mz - 1:
                    r[idx(i, j, k + 1)] = -1.0
                rows.append(r)
    return _assemble(N, rows)


def markov(m):
    """``MatMarkovModel``, ``src/eps/examples/tutorials/ex5.c:150-193``
    (N = m(m+1)/2)."""
    N = m * (m + 1) // 2
    cst = 0.5 / float(m - 1)
    rows = [dict() for _ in range(N)]
    ix = 0
    for i in range(1, m + 1):
        jmax = m - i + 1
        for j in range(1, jmax + 1):
            ix += 1
            r = rows[ix - 1]
            if j != jmax:
                pd = cst * float(i + j - 1)
                r[ix] = 2 * pd if i == 1 else pd
                r[ix + jmax - 1] = 2 * pd if j == 1 else pd
            pu = 0.5 - cst * float(i + j - 3)
            if j > 1:
                r[ix - 2] = pu
            if i > 1:
                r[ix - jmax - 2] = pu
    return _assemble(N, rows)


def csr_matvec(rowptr, colidx, val, x, y):
    """y = A x with the summation order of a sequential CSR loop (row by row,
    entries in storage order) -- what ``MatMult_SeqAIJ`` does."""
    n = len(rowptr) - 1
    for i in range(n):
        s = 0.0
        for p in range(rowptr[i], rowptr[i + 1]):
            s += val[p] * x[colidx[p]]
        y[i] = s
    return y


def csr_matvec_fast(rowptr, colidx, val, x, y):
    """Same result as :func:`csr_matvec` through scipy's sequential CSR kernel
    (identical loop order); usable at benchmark sizes."""
    from scipy.sparse import _sparsetools
    n = len(rowptr) - 1
    y[:] = 0.0
    ip = rowptr if rowptr.dtype == colidx.dtype else rowptr.astype(colidx.dtype)
    _sparsetools.csr_matvec(n, len(x), ip, colidx, val, x, y)
    return y
