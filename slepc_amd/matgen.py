"""Synthetic operators of the benchmark configurations, generated as row blocks
of CSR (int64 row pointers, int32 global column indices sorted within each row,
float64 values) -- the layout ``ksg_mat_create_csr`` takes.

These follow the matrix-filling loops of the reference's example programs
(``src/eps/examples/tutorials/ex1.c:54-58``, ``ex2.c:56-63``, ``ex19.c:70-86``,
``ex5.c:150-193``) but are vectorised with numpy so that the 16.7M-row and
134M-row cases are practical; every generator takes the range of rows
``[row0, row1)`` owned by the caller, like ``MatGetOwnershipRange`` does there.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _threads():
    """Host threads for the generators' passes (numpy releases the interpreter lock in them):
    KSG_GEN_THREADS, else OMP_NUM_THREADS when it asks for more than one (the launcher of
    torch.distributed sets it to 1 when it finds it unset), else the machine's cores shared among
    the ranks of this node; at most 16."""
    for name in ("KSG_GEN_THREADS", "OMP_NUM_THREADS"):
        try:
            n = int(os.environ.get(name, "0"))
        except ValueError:
            n = 0
        if n > 1 or (n == 1 and name == "KSG_GEN_THREADS"):
            return min(16, n)
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    return max(1, min(16, (os.cpu_count() or 1) // ranks))


def _finish(n_local, rows, cols, vals):
    """Sorts entries by (row, col) and builds the row pointer."""
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = np.zeros(n_local + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n_local), out=rowptr[1:])
    return rowptr, cols.astype(np.int32), vals.astype(np.float64)


def _stencil(n_local, offsets):
    """offsets: list of (global column array, mask array, value), given in
    ascending column order.  Builds the CSR without a sort."""
    counts = np.zeros(n_local, dtype=np.int64)
    for _, mask, _ in offsets:
        counts += mask
    rowptr = np.zeros(n_local + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colidx = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    pos = rowptr[:-1].copy()
    for col, mask, v in offsets:
        p = pos[mask]
        colidx[p] = col[mask]
        val[p] = v
        pos += mask
    return rowptr, colidx, val


def laplacian1d(n, row0=0, row1=None):
    """1-D Laplacian tridiag(-1, 2, -1) of order n (ex1.c:54-58)."""
    row1 = n if row1 is None else row1
    i = np.arange(row0, row1, dtype=np.int64)
    return _stencil(row1 - row0, [(i - 1, i > 0, -1.0), (i, np.ones_like(i, dtype=bool), 2.0),
                                  (i + 1, i < n - 1, -1.0)])


def laplacian2d(n, m=None, row0=0, row1=None):
    """2-D 5-point Laplacian on an n x m grid, II = i*n + j (ex2.c:56-63)."""
    m = n if m is None else m
    N = n * m
    row1 = N if row1 is None else row1
    II = np.arange(row0, row1, dtype=np.int64)
    i = II // n
    j = II - i * n
    return _stencil(row1 - row0, [(II - n, i > 0, -1.0), (II - 1, j > 0, -1.0),
                                  (II, np.ones_like(II, dtype=bool), 4.0),
                                  (II + 1, j < n - 1, -1.0), (II + n, i < m - 1, -1.0)])


def laplacian3d(mx, my=None, mz=None, row0=0, row1=None):
    """3-D 7-point Laplacian, natural ordering idx = i + mx*(j + my*k)
    (ex19.c:70-86)."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    N = mx * my * mz
    row1 = N if row1 is None else row1
    idx = np.arange(row0, row1, dtype=np.int64)
    i = idx % mx
    j = (idx // mx) % my
    k = idx // (mx * my)
    one = np.ones_like(idx, dtype=bool)
    return _stencil(row1 - row0, [(idx - mx * my, k > 0, -1.0), (idx - mx, j > 0, -1.0),
                                  (idx - 1, i > 0, -1.0), (idx, one, 6.0),
                                  (idx + 1, i < mx - 1, -1.0), (idx + mx, j < my - 1, -1.0),
                                  (idx + mx * my, k < mz - 1, -1.0)])


def markov(m, row0=0, row1=None):
    """Markov-chain random walk on a triangular grid, N = m(m+1)/2
    (MatMarkovModel, ex5.c:150-193)."""
    N = m * (m + 1) // 2
    row1 = N if row1 is None else row1
    cst = 0.5 / float(m - 1)
    # (i, j) of every row, 1-based, rows numbered ix-1 in the order of the loops
    jmax_all = m - np.arange(1, m + 1, dtype=np.int64) + 1
    starts = np.concatenate(([0], np.cumsum(jmax_all)))            # first ix-1 of each i
    r = np.arange(row0, row1, dtype=np.int64)
    i = np.searchsorted(starts, r, side="right")                    # 1-based i
    j = r - starts[i - 1] + 1
    jmax = m - i + 1
    ix = r + 1
    pd = cst * (i + j - 1).astype(np.float64)
    pu = 0.5 - cst * (i + j - 3).astype(np.float64)
    notlast = j != jmax
    rows = [r[notlast], r[notlast], r[j > 1], r[i > 1]]
    cols = [ix[notlast], (ix + jmax - 1)[notlast], (ix - 2)[j > 1], (ix - jmax - 2)[i > 1]]
    vals = [np.where(i == 1, 2 * pd, pd)[notlast], np.where(j == 1, 2 * pd, pd)[notlast],
            pu[j > 1], pu[i > 1]]
    rows = np.concatenate(rows) - row0
    return _finish(row1 - row0, rows, np.concatenate(cols), np.concatenate(vals))


def rmat_chunk_edges(scale, chunk_index, edge_factor=8, a=0.57, b=0.19, c=0.19, seed=0x5EED, chunk=1 << 24):
    """Edges number chunk_index*chunk .. of the R-MAT edge list (source, destination; self-loops
    removed), each chunk from its own place in the one Philox stream the whole list is drawn from:
    chunk k starts scale*chunk*k draws into it, and the generator skips there (4 draws per counter
    step), so any subset of the chunks can be drawn on its own -- by another rank, in any order --
    and the list is the same."""
    N = 1 << scale
    ne = edge_factor * N
    done = chunk_index * chunk
    cnt = min(chunk, ne - done)
    if cnt <= 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    bg = np.random.Philox(key=seed)
    bg.advance(chunk_index * scale * chunk // 4)
    rng = np.random.Generator(bg)
    src = np.zeros(cnt, dtype=np.int64)
    dst = np.zeros(cnt, dtype=np.int64)
    for _ in range(scale):
        u = rng.random(cnt)
        right = u >= a + c          # quadrant b or d: column bit set
        down = ((u >= a) & (u < a + c)) | (u >= a + b + c)   # quadrant c or d: row bit set
        src = (src << 1) | down
        dst = (dst << 1) | right
    keep = src != dst
    return src[keep], dst[keep]


def rmat_num_chunks(scale, edge_factor=8, chunk=1 << 24):
    return -(-(edge_factor << scale) // chunk)


def _rmat_bucket_edges(N, row0, row1, nbuck):
    return np.array([(row0 + (row1 - row0) * i // nbuck) * N for i in range(nbuck + 1)], dtype=np.int64)


def _rmat_split_keys(key, edges):
    """The keys cut into the row ranges edges[i] .. edges[i+1] (a list of arrays, any order within)."""
    nbuck = len(edges) - 1
    if nbuck == 1:
        return [key]
    which = (np.searchsorted(edges, key, side="right") - 1).astype(np.uint8 if nbuck <= 256 else np.uint16)   # small integers: a radix sort
    order = np.argsort(which, kind="stable")
    key = key[order]
    cuts = np.searchsorted(which[order], np.arange(nbuck + 1))
    return [key[cuts[i]:cuts[i + 1]] for i in range(nbuck)]


def _rmat_csr_from_keys(pieces, N, row0, row1, progress=None):
    """CSR of rows [row0, row1) of the graph Laplacian from the keys source*N + destination of the
    directed edges of those rows (duplicates allowed, any order).  `pieces`: a list of lists of key
    arrays, pieces[j][i] holding keys of row range i (`_rmat_split_keys`), or one array of keys."""
    if isinstance(pieces, np.ndarray):
        nbuck = max(1, min(64, pieces.size >> 24))
        pieces = [_rmat_split_keys(pieces, _rmat_bucket_edges(N, row0, row1, nbuck))]
    nbuck = len(pieces[0])

    # duplicates removed row range by row range (the same sorted result as one np.unique over
    # everything), the ranges on host threads
    def dedup(i):
        return np.unique(np.concatenate([p[i] for p in pieces]))
    with ThreadPoolExecutor(_threads()) as ex:
        parts = list(ex.map(dedup, range(nbuck)))
    del pieces
    if progress:
        progress("R-MAT: duplicates removed in %d row ranges" % nbuck)
    key = np.concatenate(parts) if nbuck > 1 else parts[0]
    del parts
    # key is sorted by (row, column): the CSR follows without another sort, the diagonal entry
    # of each row (its degree) slipped in at its place among the columns
    n_local = row1 - row0
    src = key // N
    dst = (key - src * N).astype(np.int32)
    counts = np.bincount(src - row0, minlength=n_local)
    del src
    rowptr = np.zeros(n_local + 1, dtype=np.int64)
    np.cumsum(counts + 1, out=rowptr[1:])
    g = np.arange(row0, row1, dtype=np.int64)
    pos = np.searchsorted(key, g * N + g) + np.arange(n_local, dtype=np.int64)
    del key
    total = int(rowptr[-1])
    diag = np.zeros(total, dtype=bool)
    diag[pos] = True
    colidx = np.empty(total, dtype=np.int32)
    colidx[pos] = g
    colidx[~diag] = dst
    del dst
    val = np.full(total, -1.0)
    val[pos] = counts
    if progress:
        progress("R-MAT: CSR of %d rows, %d entries built" % (n_local, total))
    return rowptr, colidx, val


def rmat_laplacian(scale, edge_factor=8, a=0.57, b=0.19, c=0.19, seed=0x5EED, row0=0, row1=None,
                   chunk=1 << 24, progress=None, relabel=None):
    """Graph Laplacian D - A of a symmetrised R-MAT graph with 2**scale vertices:
    power-law row lengths, unit weights, duplicates and self-loops removed.
    `progress(text)` is called every few seconds of work (large scales take minutes)."""
    N = 1 << scale
    row1 = N if row1 is None else row1
    ne = edge_factor * N
    # row ranges the duplicates are removed in: every chunk's keys are cut into them where they are
    # drawn, so that nothing but the chunks and the ranges -- both on host threads -- is ever sorted
    expected = 2 * ne * (row1 - row0) // N
    edges = _rmat_bucket_edges(N, row0, row1, max(1, min(64, expected >> 23)))

    def chunk_keys(k):
        src, dst = rmat_chunk_edges(scale, k, edge_factor, a, b, c, seed, chunk)
        if relabel is not None:
            src, dst = relabel[src], relabel[dst]
        s2 = np.concatenate((src, dst))
        d2 = np.concatenate((dst, src))
        if row0 != 0 or row1 != N:
            mine = (s2 >= row0) & (s2 < row1)
            s2, d2 = s2[mine], d2[mine]
        return _rmat_split_keys(s2 * N + d2, edges)
    nch = rmat_num_chunks(scale, edge_factor, chunk)
    with ThreadPoolExecutor(_threads()) as ex:             # the chunks are independent: any order, the same list
        keys = list(ex.map(chunk_keys, range(nch)))
    if progress:
        progress("R-MAT: %d edges drawn in %d chunks" % (ne, nch))
    return _rmat_csr_from_keys(keys, N, row0, row1, progress)


def rmat_laplacian_distributed(scale, starts, rank, dist, edge_factor=8, a=0.57, b=0.19, c=0.19, seed=0x5EED,
                               chunk=1 << 24, progress=None, relabel=None):
    """The rows starts[rank] .. starts[rank+1] of `rmat_laplacian(scale)` when all ranks of the
    torch.distributed group `dist` call this together: every rank draws a share of the chunks of
    the edge list (chunk k on rank k mod P), sorts its edges by the rank that owns their row and
    sends them there, so nobody draws the whole list.  The matrix is the one a single process
    generates (duplicates are removed after the exchange, and sorting makes the order of arrival
    irrelevant)."""
    import torch
    N = 1 << scale
    P = len(starts) - 1
    bounds = np.asarray(starts, dtype=np.int64)
    nch = rmat_num_chunks(scale, edge_factor, chunk)

    def draw(k):                                            # chunk k, its keys sorted by the rank that owns their row
        src, dst = rmat_chunk_edges(scale, k, edge_factor, a, b, c, seed, chunk)
        if relabel is not None:
            src, dst = relabel[src], relabel[dst]
        s2 = np.concatenate((src, dst))
        d2 = np.concatenate((dst, src))
        owner = (np.searchsorted(bounds, s2, side="right") - 1).astype(np.uint16)     # small integers: a radix sort
        order = np.argsort(owner, kind="stable")
        key = (s2 * N + d2)[order]
        cuts = np.searchsorted(owner[order], np.arange(P + 1))
        return [key[cuts[p]:cuts[p + 1]] for p in range(P)]
    mine = list(range(rank, nch, P))
    with ThreadPoolExecutor(max(1, min(_threads(), len(mine) or 1))) as ex:
        drawn = list(ex.map(draw, mine))
    if progress:
        progress("R-MAT: %d of %d chunks drawn on rank %d" % (len(mine), nch, rank))
    send = [np.concatenate([d[p] for d in drawn]) if drawn else np.zeros(0, dtype=np.int64) for p in range(P)]
    del drawn
    counts = torch.tensor([len(x) for x in send], dtype=torch.int64)
    allc = [torch.zeros(P, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(allc, counts)
    recv = [np.empty(int(allc[p][rank]), dtype=np.int64) for p in range(P)]
    recv[rank] = send[rank]
    reqs = []
    for p in range(P):
        if p == rank:
            continue
        if recv[p].size:
            reqs.append(dist.irecv(torch.from_numpy(recv[p]), src=p))
        if send[p].size:
            reqs.append(dist.isend(torch.from_numpy(send[p]), dst=p))
    for r in reqs:
        r.wait()
    del send
    nkeys = sum(x.size for x in recv)
    if progress:
        progress("R-MAT: rank %d received %d edges of its rows" % (rank, nkeys))
    # what every source sent is cut into the row ranges the duplicates are removed in, on host threads
    row0, row1 = int(starts[rank]), int(starts[rank + 1])
    edges = _rmat_bucket_edges(N, row0, row1, max(1, min(64, nkeys >> 23, max(1, row1 - row0))))
    with ThreadPoolExecutor(_threads()) as ex:
        pieces = list(ex.map(lambda x: _rmat_split_keys(x, edges), recv))
    del recv
    return _rmat_csr_from_keys(pieces, N, row0, row1, progress)


def _splitmix64(z):
    z = z + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hashed_start_vector(rows, seed=0x12345678):
    """The solver's default start vector at the global rows `rows`: 2 u - 1 with u the top 53 bits of
    splitmix64(row ^ seed) (what `ksg_vec_fill_hash` puts there; EPSGetStartVector's random vector,
    epssolve.c:675-710, made a function of the global row so that it does not depend on the partition)."""
    with np.errstate(over="ignore"):
        z = _splitmix64(np.asarray(rows, dtype=np.uint64) ^ np.uint64(seed))
    return 2.0 * ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)) - 1.0


def hashed_relabeling(N, nranks, seed=0x0B1A5ED):
    """A relabelling of the N vertices of a graph that deals them to `nranks` ranks by a hash of their
    number and keeps each rank's vertices in their original order: (new label of every vertex, the
    vertex behind every new label, the ranks' first new labels).  Contiguous blocks of the relabelled
    operator P A P^T then hold about the same number of rows AND of entries whatever the degree
    distribution (contiguous blocks of a power-law graph in its natural order cannot have both: the
    hubs sit together), while within a rank's share the hub columns stay together, which the binned
    product needs."""
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(N, dtype=np.uint64) ^ np.uint64(seed))
    owner = (h % np.uint64(nranks)).astype(np.uint8 if nranks <= 256 else np.uint16)
    del h
    old_of_new = np.argsort(owner, kind="stable")           # small integers: a radix sort
    starts = np.concatenate(([0], np.cumsum(np.bincount(owner, minlength=nranks)))).astype(np.int64)
    del owner
    new_of_old = np.empty(N, dtype=np.int64)
    new_of_old[old_of_new] = np.arange(N, dtype=np.int64)
    return new_of_old, old_of_new, starts


def slab_rows(dims, nranks, rank):
    """Row block of `rank` when a grid operator in natural ordering (fastest index first,
    ex2.c:56-63, ex19.c:70-86) is split into slabs along its slowest dimension: whole grid
    lines (2-D) or planes (3-D) per rank, as even as PetscSplitOwnership makes them.  Every
    rank then needs exactly one line / plane of x from each neighbour."""
    lead = int(np.prod(dims[:-1]))
    k0, k1 = split_rows(dims[-1], nranks, rank)
    return k0 * lead, k1 * lead


def rmat_row_weights(scale, edge_factor=8, a=0.57, b=0.19, c=0.19):
    """Expected number of entries of every row of `rmat_laplacian` (before duplicates are
    removed): an edge has its source bit set with probability c+d and its destination bit with
    b+d at every level, independently across levels, and every edge is stored in both
    directions; plus the diagonal."""
    N = 1 << scale
    d = 1.0 - a - b - c
    v = np.arange(N, dtype=np.uint32) if scale <= 32 else np.arange(N, dtype=np.uint64)
    ones = np.zeros(N, dtype=np.int8)
    for bit in range(scale):
        ones += ((v >> bit) & 1).astype(np.int8)
    ps, pd = c + d, b + d
    k = np.arange(scale + 1, dtype=np.float64)
    table = edge_factor * N * (ps ** k * (1 - ps) ** (scale - k) + pd ** k * (1 - pd) ** (scale - k)) + 1.0
    return table[ones]


def balanced_row_blocks(weights, nranks, row_weight=0.0):
    """Contiguous row blocks of (nearly) equal total weight: starts[p] is the first row of rank p,
    starts[nranks] = N.  weights[i] is the cost of row i in matrix entries (its number of entries,
    exact or expected); `row_weight` adds a cost per row in the same unit, for the work that grows
    with the rows whatever they hold (the basis-vector kernels: with it at 0 the split balances the
    entries only, and a power-law operator leaves some ranks with several times the rows of others)."""
    w = np.asarray(weights, dtype=np.float64) + float(row_weight)
    cum = np.cumsum(w)
    total = cum[-1]
    starts = [0]
    for p in range(1, nranks):
        starts.append(max(starts[-1], int(np.searchsorted(cum, total * p / nranks, side="left")) + 1))
    starts.append(len(w))
    return [min(s, len(w)) for s in starts]


def split_rows(N, nranks, rank):
    """Contiguous row blocks as PETSc's default layout (PetscSplitOwnership):
    the first N % nranks ranks get one extra row."""
    base, extra = divmod(N, nranks)
    row0 = rank * base + min(rank, extra)
    return row0, row0 + base + (1 if rank < extra else 0)
