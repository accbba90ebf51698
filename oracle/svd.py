"""CPU restatement of the reference's singular value solvers that run on the same
kernels as the eigensolver (test infrastructure: only the tests use this module).

* ``SVD(..., solver="trlanczos")``: thick-restart Lanczos bidiagonalisation,
  ``SVDSolve_TRLanczos`` (src/svd/impls/trlanczos/trlanczos.c:272-385) with the two-sided
  recurrence ``SVDTwoSideLanczos`` (src/svd/impls/lanczos/gklanczos.c:67-107) or the
  one-sided one with classical Gram-Schmidt ``SVDOneSideTRLanczosCGS``
  (trlanczos.c:134-270);
* ``solver="cross"`` (the reference's default, svdsetup.c:112-114): the eigenproblem of
  A^T A, ``SVDSetUp_Cross``/``SVDSolve_Cross`` (src/svd/impls/cross/cross.c:118-200);
* the projected problem ``DSSVD`` (src/sys/classes/ds/impls/svd/dssvd.c) with numpy's SVD
  (LAPACK gesdd, the routine the reference calls).

One process only.  The operator is given as two callbacks, y = A x and y = A^T x.
"""
import numpy as np

from . import ds as dsmod
from .bv import BV, REFINE_IFNEEDED, REFINE_NEVER, REFINE_ALWAYS
from .eps import EPS, splitmix64_uniform

ITERATING, CONVERGED_TOL, DIVERGED_ITS = 0, 1, -1
EPSILON = np.finfo(float).eps


class DSSVD:
    """``DSSVD`` in compact storage: T[0:ld] the diagonal, T[ld:2ld] the arrow / upper
    off-diagonal (dssvd.c:39-62)."""

    def __init__(self, ld, compare):
        self.ld = ld
        self.compare = compare
        self.T = np.zeros(3 * ld)
        self.A = np.zeros((ld, ld))
        self.U = np.zeros((ld, ld))
        self.VT = np.zeros((ld, ld))
        self.perm = np.zeros(ld, dtype=int)
        self.n = self.m = self.l = self.k = 0
        self.state = dsmod.STATE_RAW

    def set_dimensions(self, n, m, l, k):
        self.n, self.m, self.l, self.k = n, m, l, k

    def set_state(self, state):
        self.state = state

    def _to_dense(self):
        """``DSSwitchFormat_SVD`` to dense storage (dssvd.c:87-98)."""
        A, T, ld, k, m = self.A, self.T, self.ld, self.k, self.m
        A[:, :] = 0.0
        for i in range(k):
            A[i, i] = T[i]
            A[i, k] = T[i + ld]
        A[k, k] = T[k]
        for i in range(k + 1, m):
            A[i, i] = T[i]
            A[i - 1, i] = T[i - 1 + ld]

    def solve(self, wr):
        """``DSSolve_SVD_DC`` (dssvd.c:197-290)."""
        n, m, l, ld = self.n, self.m, self.l, self.ld
        d, e = self.T[:ld], self.T[ld:2 * ld]
        U, VT = self.U, self.VT
        U[:, :] = 0.0
        VT[:, :] = 0.0
        for i in range(l):
            U[i, i] = 1.0
            VT[i, i] = 1.0
            wr[i] = d[i]
        n3 = n - l
        m3 = m - l
        if self.state > dsmod.STATE_RAW:
            B = np.zeros((n3, n3))              # upper bidiagonal of bdsdc("U")
            for i in range(n3):
                B[i, i] = d[l + i]
                if i + 1 < n3:
                    B[i, i + 1] = e[l + i]
            u, s, vt = np.linalg.svd(B)
            U[l:l + n3, l:l + n3] = u
            VT[l:l + n3, l:l + n3] = vt
            d[l:l + n3] = s
        else:
            self._to_dense()
            u, s, vt = np.linalg.svd(self.A[l:l + n3, l:l + m3])
            U[l:l + n3, l:l + n3] = u
            VT[l:l + m3, l:l + m3] = vt
            d[l:l + len(s)] = s
        for i in range(l, min(n, m)):
            wr[i] = d[i]
        e[:max(n - 1, 0)] = 0.0
        self.state = dsmod.STATE_CONDENSED

    def sort(self, wr):
        """``DSSort_SVD`` (dssvd.c:167-195): ``DSSortEigenvaluesReal_Private``
        (dspriv.c:242-264) and ``DSPermuteBoth_Private`` (dspriv.c:335-362)."""
        l, ld = self.l, self.ld
        n = min(self.n, self.m)
        d = self.T[:ld]
        perm = self.perm
        for i in range(n):
            perm[i] = i
        for i in range(l + 1, n):
            re = d[perm[i]]
            j = i - 1
            result = self.compare(re, 0.0, d[perm[j]], 0.0)
            while result < 0 and j >= l:
                perm[j], perm[j + 1] = perm[j + 1], perm[j]
                j -= 1
                if j >= l:
                    result = self.compare(re, 0.0, d[perm[j]], 0.0)
        for i in range(l, n):
            wr[i] = d[perm[i]]
        U, VT, m = self.U, self.VT, self.m
        for i in range(l, n):
            p = perm[i]
            if p != i:
                j = i + 1
                while perm[j] != i:
                    j += 1
                perm[j] = p
                perm[i] = i
                U[:n, [p, i]] = U[:n, [i, p]]
                VT[[p, i], :m] = VT[[i, p], :m]
        for i in range(l, n):
            d[i] = wr[i]


class SVD:
    """``SVDCreate`` ... ``SVDSolve`` ... ``SVDGetSingularTriplet`` for an M x N operator."""

    def __init__(self, mult, mult_t, M, N, nsv=1, ncv=0, mpd=0, tol=None, max_it=0, which="largest",
                 solver="cross", oneside=False, conv="rel", v0=None, orthog_refine=REFINE_IFNEEDED):
        if solver not in ("cross", "trlanczos", "lanczos", "cyclic"):
            raise ValueError("unknown solver " + solver)
        self.M, self.N = M, N
        self.swapped = M < N                      # svdsetup.c:139-158: the solver works with the tall one
        if self.swapped:
            self._A, self._AT = mult_t, mult
            self.rows, self.cols = N, M
        else:
            self._A, self._AT = mult, mult_t
            self.rows, self.cols = M, N
        self.nsv, self.ncv, self.mpd = nsv, ncv, mpd
        self.tol, self.max_it = tol, max_it        # tol None: the default (1e-8, svdsetup.c:163)
        self.which = which
        self.solver = solver
        self.oneside = oneside
        self.conv = conv
        self.v0 = v0                              # right initial vector of the operator as given
        self.orthog_refine = orthog_refine
        self.reason = ITERATING
        self.its = 0
        self.nconv = 0
        self.nmult = 0

    def _matmult(self, trans, x, y):
        """``SVDMatMult`` (include/slepc/private/svdimpl.h)."""
        self.nmult += 1
        (self._AT if trans else self._A)(x, y)

    # -- setup ------------------------------------------------------------------------
    def setup(self):
        mn = min(self.M, self.N)
        if self.ncv > mn:
            self.ncv = mn
        if self.nsv > mn:
            self.nsv = mn
        if self.ncv and self.nsv > self.ncv:
            raise ValueError("nsv bigger than ncv")
        if self.solver == "trlanczos":
            self._setup_trlanczos()
        elif self.solver == "lanczos":
            self._setup_trlanczos()        # SVDSetUp_Lanczos (gklanczos.c:46-64) sets the same things up
        elif self.solver == "cyclic":
            self._setup_cyclic()
        else:
            self._setup_cross()
        return self

    def _dimensions_default(self):
        """``SVDSetDimensions_Default`` (svdsetup.c:295-315)."""
        N = self.cols
        if self.ncv:
            if self.ncv < self.nsv:
                raise ValueError("The value of ncv must be at least nsv")
        elif self.mpd:
            self.ncv = min(N, self.nsv + self.mpd)
        else:
            if self.nsv < 500:
                self.ncv = min(N, max(2 * self.nsv, 10))
            else:
                self.mpd = 500
                self.ncv = min(N, self.nsv + self.mpd)
        if not self.mpd:
            self.mpd = self.ncv

    def _setup_trlanczos(self):
        """``SVDSetUp_TRLanczos`` (trlanczos.c:59-75)."""
        self._dimensions_default()
        if self.tol is None:
            self.tol = 1e-8
        if self.ncv > self.nsv + self.mpd:
            raise ValueError("The value of ncv must not be larger than nev+mpd")
        if not self.max_it:
            self.max_it = max(self.cols // self.ncv, 100)
        ncv = self.ncv
        self.V = BV(self.cols, ncv + 1)
        self.U = BV(self.rows, ncv + 1)
        self.V.orthog_ref = self.U.orthog_ref = self.orthog_refine
        self.sigma = np.zeros(ncv + 1)
        self.errest = np.zeros(ncv + 1)
        self.perm = np.arange(ncv + 1)
        cmpf = dsmod.COMPARE["largest_real" if self.which == "largest" else "smallest_real"]
        self.ds = DSSVD(ncv, cmpf)
        if self.v0 is not None and not self.swapped:
            # BVInsertVecs(svd->V,0,&k,svd->IS,PETSC_TRUE), svdsetup.c:183-189
            self.V.col(0)[:] = self.v0
            self.V.orthonormalize_column(0, True)
            self.nini = 1
        else:
            self.nini = 0

    def _setup_cross(self):
        """``SVDSetUp_Cross`` (cross.c:118-172): EPS on the shell matrix A^T A."""
        w = np.zeros(self.rows)

        def cross_mult(x, y):
            self._matmult(False, x, w)
            self._matmult(True, w, y)
        self.eps = EPS(cross_mult, self.cols, hermitian=True, nev=self.nsv, ncv=self.ncv, mpd=self.mpd,
                       tol=1e-8 / 10.0 if self.tol is None else self.tol, max_it=self.max_it,
                       which="largest_real" if self.which == "largest" else "smallest_real",
                       v0=None if self.swapped else self.v0)
        self.eps.conv = self.conv                  # EPSSetConvergenceTest, cross.c:147-154
        self.eps.setup()
        if self.tol is None:
            self.tol = 1e-8
        self.ncv, self.mpd, self.max_it = self.eps.ncv, self.eps.mpd, self.eps.max_it
        self.V = BV(self.cols, self.ncv)
        self.U = None
        self.sigma = np.zeros(self.ncv)
        self.errest = np.zeros(self.ncv)
        self.perm = np.arange(self.ncv)

    def _setup_cyclic(self):
        """``SVDSetUp_Cyclic`` (src/svd/impls/cyclic/cyclic.c:81-226), shell variant: EPS on
        H(A) = [0 A; A^T 0] of order rows+cols; its positive eigenvalues are the singular
        values, its eigenvectors hold [u; v]/sqrt(2)."""
        m, n = self.rows, self.cols

        def cyclic_mult(x, y):
            self._matmult(False, x[m:], y[:m])
            self._matmult(True, x[:m], y[m:])

        def smallest_pos_real(ar, ai, br, bi):
            """``SlepcCompareSmallestPosReal`` (slepcsc.c)."""
            aisright, bisright = ar > 0.0, br > 0.0
            if aisright == bisright:
                if aisright:
                    return -1 if ar < br else (1 if ar > br else 0)
                return 1 if ar < br else (-1 if ar > br else 0)       # both non-positive: the larger first
            return -1 if aisright else 1
        v0 = None
        if self.v0 is not None and not self.swapped:
            v0 = np.concatenate([np.zeros(m), self.v0])
        which = "largest_real"
        if self.which != "largest":
            dsmod.COMPARE["_smallest_pos_real"] = smallest_pos_real
            which = "_smallest_pos_real"
        self.eps = EPS(cyclic_mult, m + n, hermitian=True, nev=self.nsv, ncv=self.ncv, mpd=self.mpd,
                       tol=1e-8 / 10.0 if self.tol is None else self.tol, max_it=self.max_it, which=which, v0=v0)
        self.eps.conv = self.conv
        self.eps.setup()
        self.ncv, self.mpd, self.max_it = min(self.eps.ncv, min(self.M, self.N)), self.eps.mpd, self.eps.max_it
        if self.tol is None:
            self.tol = 1e-8
        self.V = BV(n, self.eps.ncv)
        self.U = BV(m, self.eps.ncv)
        self.sigma = np.zeros(self.eps.ncv)
        self.errest = np.zeros(self.eps.ncv)
        self.perm = np.arange(self.eps.ncv)

    # -- solve ------------------------------------------------------------------------
    def solve(self):
        """``SVDSolve`` (svdsolve.c:80-128)."""
        self.setup()
        self.its = 0
        self.nconv = 0
        if self.solver == "trlanczos":
            self._solve_trlanczos()
            self._have_left = True
        elif self.solver == "lanczos":
            self._solve_lanczos()
            self._have_left = not self.oneside
        elif self.solver == "cyclic":
            self._solve_cyclic()
            self._have_left = True
        else:
            self._solve_cross()
            self._have_left = False
        order = np.argsort(self.sigma[:self.nconv], kind="stable")
        if self.which == "smallest":
            self.perm[:self.nconv] = order
        else:
            self.perm[:self.nconv] = order[::-1]
        return self

    def _converged(self, sigma, res):
        return res if self.conv == "abs" else res / sigma

    def _two_side_lanczos(self, alpha, beta, k, n):
        """``SVDTwoSideLanczos`` (gklanczos.c:67-107)."""
        V, U = self.V, self.U
        self._matmult(False, V.col(k), U.col(k))
        alpha[k], _ = U.orthogonalize_column(k)
        U.scale_column(k, 1.0 / alpha[k])
        for i in range(k + 1, n):
            self._matmult(True, U.col(i - 1), V.col(i))
            beta[i - 1], _ = V.orthogonalize_column(i)
            V.scale_column(i, 1.0 / beta[i - 1])
            self._matmult(False, V.col(i), U.col(i))
            alpha[i], _ = U.orthogonalize_column(i)
            U.scale_column(i, 1.0 / alpha[i])
        self._matmult(True, U.col(n - 1), V.col(n))
        beta[n - 1], _ = V.orthogonalize_column(n)

    def _orthogonalize_cgs(self, i, h, a):
        """``SVDOrthogonalizeCGS`` (trlanczos.c:134-173); h holds i+1 values for "if needed"."""
        V = self.V
        refine, eta = V.orthog_ref, V.orthog_eta
        if refine == REFINE_NEVER:
            return V.norm_column(i)
        if refine == REFINE_ALWAYS:
            V.set_active_columns(0, i)
            hh = V.dot_column(i)
            V.mult_column(-1.0, 1.0, i, hh.copy())
            return V.norm_column(i)
        dot = h[i]
        onorm = np.sqrt(dot) / a
        s = 0.0
        for j in range(i):
            s += h[j] * h[j]
        norm = dot / (a * a) - s
        if norm > 0.0:
            norm = np.sqrt(norm)
        else:
            norm = V.norm_column(i)
        if norm < eta * onorm:
            V.set_active_columns(0, i)
            hh = V.dot_column(i)
            V.mult_column(-1.0, 1.0, i, hh.copy())
            norm = V.norm_column(i)
        return norm

    def _one_side_cgs(self, alpha, beta, nconv, l, n):
        """``SVDOneSideTRLanczosCGS`` (trlanczos.c:175-270): only V is orthogonalised; the
        norm of the previous left vector travels with the Gram-Schmidt coefficients."""
        V, U = self.V, self.U
        k = nconv + l
        refine = V.orthog_ref
        self._matmult(False, V.col(k), U.col(k))
        if l > 0:
            U.mult_column(-1.0, 1.0, k, beta[nconv:nconv + l].copy())
        for i in range(k + 1, n + 1):
            self._matmult(True, U.col(i - 1), V.col(i))
            a = U.norm_column(i - 1)
            if refine == REFINE_IFNEEDED:
                V.set_active_columns(0, i + 1)
                work = np.array(V.dotvec(V.col(i)), dtype=float)
                V.set_active_columns(0, i)
            else:
                V.set_active_columns(0, i)
                work = np.array(V.dot_column(i), dtype=float)
            U.scale_column(i - 1, 1.0 / a)
            work[:i] = work[:i] / a
            V.mult_column(-1.0, 1.0 / a, i, work[:i].copy())
            b = self._orthogonalize_cgs(i, work, a)
            if i < n:
                V.scale_column(i, 1.0 / b)
                if abs(b) < 10 * EPSILON:
                    raise ArithmeticError("Recurrence generated a zero vector; use a two-sided variant")
                self._matmult(False, V.col(i), U.col(i))
                U.col(i)[:] -= b * U.col(i - 1)
            alpha[i - 1] = a
            beta[i - 1] = b
        V.set_active_columns(nconv, n)

    def _solve_trlanczos(self):
        """``SVDSolve_TRLanczos`` (trlanczos.c:272-385)."""
        V, U, ds = self.V, self.U, self.ds
        ld = ds.ld
        w = np.zeros(ld)
        if not self.nini:
            V.col(0)[:] = splitmix64_uniform(np.arange(self.cols, dtype=np.uint64))
            V.orthonormalize_column(0, True)
        l = 0
        while self.reason == ITERATING:
            self.its += 1
            nv = min(self.nconv + self.mpd, self.ncv)
            V.set_active_columns(self.nconv, nv)
            U.set_active_columns(self.nconv, nv)
            alpha, beta = ds.T[:ld], ds.T[ld:2 * ld]
            if self.oneside:
                self._one_side_cgs(alpha, beta, self.nconv, l, nv)
            else:
                self._two_side_lanczos(alpha, beta, self.nconv + l, nv)
            lastbeta = beta[nv - 1]
            V.scale_column(nv, 1.0 / lastbeta)

            ds.set_dimensions(nv, nv, self.nconv, self.nconv + l)
            ds.set_state(dsmod.STATE_INTERMEDIATE if l == 0 else dsmod.STATE_RAW)
            ds.solve(w)
            ds.sort(w)

            k = 0
            conv = True
            for i in range(self.nconv, nv):
                self.sigma[i] = w[i]
                beta[i] = ds.U[nv - 1, i] * lastbeta
                resnorm = abs(beta[i])
                self.errest[i] = self._converged(self.sigma[i], resnorm)
                if conv:
                    if self.errest[i] < self.tol:
                        k += 1
                    else:
                        conv = False

            if self.nconv + k >= self.nsv:
                self.reason = CONVERGED_TOL
            elif self.its >= self.max_it:
                self.reason = DIVERGED_ITS
            l = 0 if self.reason != ITERATING else max((nv - self.nconv - k) // 2, 0)

            # V <- V VT^T, U <- U Uds on columns nconv .. nconv+k+l (active nconv..nv)
            V.op_multinplace(ds.VT[:nv, :nv].copy(), self.nconv, self.nconv + k + l, True)
            U.op_multinplace(ds.U[:nv, :nv].copy(), self.nconv, self.nconv + k + l, False)
            if self.reason == ITERATING:
                V.copy_column(nv, self.nconv + k + l)
            self.nconv += k
        if self.oneside:
            for i in range(self.nconv):
                U.orthonormalize_column(i, False)

    def _one_side_lanczos(self, alpha, beta, u, u1, k, n):
        """``SVDOneSideLanczos`` (gklanczos.c:109-169): the left vectors are not kept, only the
        current and the previous one."""
        V = self.V
        bvl, bvk = V.l, V.k
        self._matmult(False, V.col(k), u)
        for i in range(k + 1, n + 1):
            self._matmult(True, u, V.col(i))
            a = np.sqrt(float(V.comm.allreduce_sum(np.array([u @ u]))[0]))
            V.set_active_columns(0, i)
            work = np.array(V.dot_column(i), dtype=float)
            u *= 1.0 / a
            V.mult_column(-1.0 / a, 1.0 / a, i, work.copy())
            work = np.array(V.dot_column(i), dtype=float)
            V.mult_column(-1.0, 1.0, i, work.copy())
            b = V.norm_column(i)
            if i < n:
                if abs(b) < 10 * EPSILON:
                    raise ArithmeticError("Recurrence generated a zero vector; use a two-sided variant")
                V.scale_column(i, 1.0 / b)
                self._matmult(False, V.col(i), u1)
                u1 -= b * u
                u, u1 = u1, u
            alpha[i - 1] = a
            beta[i - 1] = b
        V.set_active_columns(bvl, bvk)

    def _solve_lanczos(self):
        """``SVDSolve_Lanczos`` (gklanczos.c:171-275): explicit restart with the first
        unconverged right singular vector."""
        V, U, ds = self.V, self.U, self.ds
        ld = ds.ld
        w = np.zeros(ld)
        u = np.zeros(self.rows)
        u1 = np.zeros(self.rows)
        if not self.nini:
            V.col(0)[:] = splitmix64_uniform(np.arange(self.cols, dtype=np.uint64))
            V.scale_column(0, 1.0 / V.norm_column(0))
        while self.reason == ITERATING:
            self.its += 1
            nv = min(self.nconv + self.mpd, self.ncv)
            V.set_active_columns(self.nconv, nv)
            alpha, beta = ds.T[:ld], ds.T[ld:2 * ld]
            if self.oneside:
                self._one_side_lanczos(alpha, beta, u, u1, self.nconv, nv)
            else:
                U.set_active_columns(self.nconv, nv)
                self._two_side_lanczos(alpha, beta, self.nconv, nv)
            lastbeta = beta[nv - 1]
            ds.set_dimensions(nv, nv, self.nconv, 0)
            ds.set_state(dsmod.STATE_INTERMEDIATE)
            ds.solve(w)
            ds.sort(w)
            k = 0
            conv = True
            for i in range(self.nconv, nv):
                self.sigma[i] = w[i]
                resnorm = abs(ds.U[nv - 1, i]) * lastbeta
                self.errest[i] = self._converged(self.sigma[i], resnorm)
                if conv:
                    if self.errest[i] < self.tol:
                        k += 1
                    else:
                        conv = False
            if self.nconv + k >= self.nsv:
                self.reason = CONVERGED_TOL
            elif self.its >= self.max_it:
                self.reason = DIVERGED_ITS
            if self.reason == ITERATING:
                # restart vector: the first unconverged right vector, V(:,nconv:nv) * VT(k+nconv, nconv:nv)^T
                V.mult_column(1.0, 0.0, nv, ds.VT[k + self.nconv, self.nconv:nv].copy())
            V.op_multinplace(ds.VT[:nv, :nv].copy(), self.nconv, self.nconv + k, True)
            if not self.oneside:
                U.op_multinplace(ds.U[:nv, :nv].copy(), self.nconv, self.nconv + k, False)
            if self.reason == ITERATING:
                V.copy_column(nv, self.nconv + k)
            self.nconv += k

    def _solve_cyclic(self):
        """``SVDSolve_Cyclic`` (cyclic.c:228-267).  The eigenvector of H(A) of unit norm holds
        u/sqrt(2) and v/sqrt(2): both halves are scaled by sqrt(2) (the reference revision at
        hand divides instead, which leaves vectors of norm 1/2)."""
        e = self.eps
        e.solve()
        self.its, self.reason = e.its, e.reason
        m = self.rows
        j = 0
        for i in range(e.nconv):
            lam, _ = e.get_eigenvalue(i)
            if lam > 0.0:
                xr, _ = e.get_eigenvector(i)
                self.sigma[j] = lam
                self.U.col(j)[:] = xr[:m] * np.sqrt(2.0)
                self.V.col(j)[:] = xr[m:] * np.sqrt(2.0)
                j += 1
        self.nconv = j

    def _solve_cross(self):
        """``SVDSolve_Cross`` (cross.c:174-200)."""
        e = self.eps
        e.solve()
        self.nconv, self.its, self.reason = e.nconv, e.its, e.reason
        for i in range(self.nconv):
            lam, _ = e.get_eigenvalue(i)
            xr, _ = e.get_eigenvector(i)
            self.V.col(i)[:] = xr
            if lam < 0.0:
                raise ArithmeticError("Negative eigenvalue computed by EPS")
            self.sigma[i] = np.sqrt(lam)

    # -- post-solve ---------------------------------------------------------------------
    def _compute_vectors(self):
        """``SVDComputeVectors`` (svdsolve.c:26-57): u_j = A v_j, orthonormalised."""
        if self._have_left:
            return
        self._have_left = True
        self.U = BV(self.rows, self.ncv)
        for j in range(self.nconv):
            self._matmult(False, self.V.col(j), self.U.col(j))
            self.U.orthonormalize_column(j, False)

    def get_singular_triplet(self, i, vectors=True):
        """``SVDGetSingularTriplet`` (svdsolve.c:256-280) -> (sigma, u, v) of the operator as
        given (u has M entries, v has N)."""
        p = self.perm[i]
        sigma = self.sigma[p]
        if not vectors:
            return sigma, None, None
        self._compute_vectors()
        u, v = self.U.col(p).copy(), self.V.col(p).copy()
        if self.swapped:
            u, v = v, u
        return sigma, u, v

    def compute_error(self, i, relative=True):
        """``SVDComputeError`` (svdsolve.c:355-379): sqrt(||A v - s u||^2 + ||A^T u - s v||^2)."""
        sigma, u, v = self.get_singular_triplet(i)
        mult, mult_t = (self._AT, self._A) if self.swapped else (self._A, self._AT)
        x = np.zeros(self.M)
        y = np.zeros(self.N)
        mult(v, x)
        mult_t(u, y)
        n1 = np.linalg.norm(x - sigma * u)
        n2 = np.linalg.norm(y - sigma * v)
        err = np.hypot(n1, n2)
        return err / sigma if relative else err
