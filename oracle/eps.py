"""CPU restatement of ``EPSSolve`` with the Krylov-Schur solver (symmetric and
non-symmetric variants) as the reference runs it on the CPU.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``src/eps/impls/krylov/krylovschur/ks-symm.c:30-111``,
``krylovschur.c:73-288``, ``src/eps/impls/krylov/epskrylov.c:39-74,252-327,347-376``,
``src/eps/interface/epssolve.c:63-146,334-462,506-710``,
``epssetup.c:466-490`` and ``epsdefault.c:136-338``.
"""
import numpy as np

from . import bv as bvmod
from . import ds as dsmod
from .bv import BV, SelfComm, EPSILON, REFINE_IFNEEDED, REFINE_NEVER, REFINE_ALWAYS

ITERATING = 0
CONVERGED_TOL = 2
DIVERGED_ITS = -3
DIVERGED_BREAKDOWN = -4


def splitmix64_uniform(g, seed=0x12345678):
    """Counter-based start vector: value in [-1,1) from the *global* row index,
    so that it is identical for every partition of the rows."""
    z = (np.asarray(g, dtype=np.uint64) ^ np.uint64(seed)) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return 2.0 * u - 1.0


class EPS:
    """Standard eigenproblem ``A x = lambda x`` solved by Krylov-Schur.

    ``matvec(x, y)`` computes y = A x on the local rows (the ``STApply`` ->
    ``STApply_Shift`` -> ``MatMult`` hop of stsolve.c:44-72 / shift.c:27-41 with
    sigma = 0).
    """

    def __init__(self, matvec, n_local, N=None, hermitian=True, nev=1, ncv=0, mpd=0,
                 tol=1e-8, max_it=0, which=None, keep=0.5, lock=True, comm=None,
                 row0=0, v0=None, monitor=None, solver="krylovschur", delayed=False,
                 orthog_refine=REFINE_IFNEEDED, reorthog="local", Bmatvec=None, positive=True,
                 st="shift", sigma=None, target=0.0, solve=None, purify=True, deflation=None, bv_factory=None):
        """Generalized problem: ``Bmatvec(x, y)`` computes y = B x.  ``st`` is the spectral
        transformation ("shift", "sinvert"), ``sigma`` its shift (sinvert: the target when not
        given, sinvert.c:112) and ``solve(b, x)`` solves with the matrix the transformation needs
        -- B for shift, A - sigma B for sinvert -- the KSPSolve the reference delegates to PETSc
        (stsles.c).  ``deflation``: vectors of EPSSetDeflationSpace."""
        self.matvec = matvec
        self.bv_factory = bv_factory or BV           # storage type of the basis (-bv_type)
        self.Bmatvec = Bmatvec
        self.generalized = Bmatvec is not None
        self.positive = self.generalized and positive and (hermitian or positive == "pgnhep")
        if st not in ("shift", "sinvert"):
            raise ValueError("unknown ST type " + st)
        self.st = st
        self.sigma = (target if st == "sinvert" else 0.0) if sigma is None else sigma
        self.target = target
        self.lin_solve = solve
        self.purify = purify
        self.deflation = deflation or []
        if (self.generalized or st == "sinvert") and solve is None:
            raise ValueError("a linear solver is needed")
        self.n = int(n_local)
        self.N = int(N if N is not None else n_local)
        self.hermitian = hermitian
        if solver not in ("krylovschur", "arnoldi", "lanczos"):
            raise ValueError("unknown solver " + solver)
        self.solver = solver
        self.conv = "rel"                            # EPSSetConvergenceTest
        self.delayed = delayed                       # EPSArnoldiSetDelayed (arnoldi.c:168-175)
        self.orthog_refine = orthog_refine           # -bv_orthog_refine
        if reorthog not in ("local", "full", "selective", "periodic", "partial", "delayed"):
            raise ValueError("unknown reorthogonalization " + reorthog)
        self.reorthog = reorthog                     # EPSLanczosSetReorthog; the reference's default is local
        self.nev, self.ncv, self.mpd = nev, ncv, mpd
        self.tol = tol                               # default 1e-8: epssetup.c:184
        self.max_it = max_it
        self.which = which
        self.keep = keep
        self.lock = lock
        self.comm = comm if comm is not None else SelfComm()
        self.row0 = row0
        self.v0 = v0
        self.monitor = monitor
        self.reason = ITERATING
        self.its = 0
        self.nconv = 0
        self.nmatvec = 0

    # -- setup: EPSSetUp_KrylovSchur (krylovschur.c:73-167) ----------------
    def setup(self):
        nev = self.nev
        if self.ncv:                                  # EPSSetDimensions_Default epssetup.c:466-490
            if self.ncv < nev + 1 and not (self.ncv == nev and self.ncv == self.N):
                raise ValueError("The value of ncv must be at least nev+1")
        elif self.mpd:
            self.ncv = min(self.N, nev + self.mpd)
        else:
            if nev < 500:
                self.ncv = min(self.N, max(2 * nev, nev + 15))
            else:
                self.mpd = 500
                self.ncv = min(self.N, nev + self.mpd)
        if not self.mpd:
            self.mpd = self.ncv
        if self.ncv > self.nev + self.mpd:
            raise ValueError("The value of ncv must not be larger than nev+mpd")
        if not self.max_it:
            self.max_it = max(100, 2 * self.N // self.ncv)   # krylovschur.c:89
        if not self.which:                                   # epsdefault.c:243-253
            self.which = "target_magnitude" if self.st == "sinvert" else "largest_magnitude"
        if not self.keep:
            self.keep = 0.5                                  # krylovschur.c:103
        ncv = self.ncv
        self.V = self.bv_factory(self.n, ncv + 1, N=self.N, comm=self.comm)   # EPSAllocateSolution(eps,1)
        self.V.orthog_ref = self.orthog_refine
        if self.solver == "lanczos" and self.reorthog != "full":
            self.V.orthog_type = bvmod.MGS            # EPSSetUp_Lanczos lanczos.c:77-81
        if self.solver == "lanczos" and self.reorthog == "selective":
            self.AV = BV(self.n, ncv + 1, N=self.N, comm=self.comm)
            self.AV.orthog_type, self.AV.orthog_ref = bvmod.MGS, self.orthog_refine
        self.eigr = np.zeros(ncv + 1)
        self.eigi = np.zeros(ncv + 1)
        self.errest = np.zeros(ncv + 1)
        self.perm = np.arange(ncv + 1)
        # purification only where it makes sense (epssetup.c:169-177)
        if self.purify and (not self.generalized or (not self.hermitian and not self.positive)):
            self.purify = False
        # EPS_SetInnerProduct (epsimpl.h:146-161)
        if self.positive or (self.generalized and self.hermitian):
            def bmul(x):
                y = np.zeros(self.n)
                self.Bmatvec(x, y)
                return y
            self.V.set_matrix(bmul, indef=not self.positive)
        if self.deflation:                                   # epssetup.c:263-270
            self.V.insert_constraints(self.deflation)
        self.cmp0 = self._raw_compare()
        trivial = self.st == "shift" and self.sigma == 0.0 and not self.generalized
        if trivial:
            cmpf = self.cmp0
        else:
            # eigenvalues of the transformed operator are ordered by what they are in the original
            # problem (SlepcSCCompare with map = SlepcMap_ST, slepcsc.c:51-81; epssetup.c:239-248)
            def cmpf(ar, ai, br, bi):
                ar, ai = self._back_transform(ar, ai)
                br, bi = self._back_transform(br, bi)
                return self.cmp0(ar, ai, br, bi)
        self.cmpf = cmpf
        if self.hermitian and self.solver in ("krylovschur", "lanczos"):
            self.ds = dsmod.DSHEP(ncv + 1, cmpf)
        else:
            self.ds = dsmod.DSNHEP(ncv + 1, cmpf)

    # -- helpers --------------------------------------------------------------
    def _raw_compare(self):
        if self.which == "target_magnitude":                 # slepcsc.c:261-274
            t = self.target
            return lambda ar, ai, br, bi: (lambda a, b: 1 if a > b else (-1 if a < b else 0))(
                dsmod.abs_eig(ar - t, ai), dsmod.abs_eig(br - t, bi))
        if self.which == "target_real":                      # slepcsc.c:276-289
            t = self.target
            return lambda ar, ai, br, bi: (lambda a, b: 1 if a > b else (-1 if a < b else 0))(abs(ar - t), abs(br - t))
        return dsmod.COMPARE[self.which]

    def _back_transform(self, re, im):
        """``STBackTransform_Shift`` (shift.c:60-70) / ``_Sinvert`` (sinvert.c:59-82)."""
        if self.st == "shift":
            return re + self.sigma, im
        if im == 0:
            return 1.0 / re + self.sigma, im
        t = re * re + im * im
        return re / t + self.sigma, -im / t

    def _apply(self, x, y):
        """``STApply``: shift.c:27-41, sinvert.c:26-40."""
        self.nmatvec += 1
        if self.st == "shift":
            if self.generalized:                             # y = B^-1 (A - sigma B) x
                w = np.zeros(self.n)
                self.matvec(x, w)
                if self.sigma != 0.0:
                    bx = np.zeros(self.n)
                    self.Bmatvec(x, bx)
                    w -= self.sigma * bx
                self.lin_solve(w, y)
            else:
                self.matvec(x, y)
                if self.sigma != 0.0:
                    y -= self.sigma * x
        else:
            if self.generalized:                             # y = (A - sigma B)^-1 B x
                w = np.zeros(self.n)
                self.Bmatvec(x, w)
                self.lin_solve(w, y)
            else:
                self.lin_solve(np.array(x), y)

    def _random_column(self, j):
        g = np.arange(self.row0, self.row0 + self.n, dtype=np.uint64)
        return splitmix64_uniform(g, seed=0x12345678 + 7919 * (j + 1) + self.its)

    def _get_start_vector(self, i):
        """``EPSGetStartVector`` (epssolve.c:675-710).  Returns breakdown flag."""
        V = self.V
        if i > 0 or self.v0 is None:
            if i == 0:
                g = np.arange(self.row0, self.row0 + self.n, dtype=np.uint64)
                V.col(0)[:] = splitmix64_uniform(g)
            else:
                V.col(i)[:] = self._random_column(i)
        else:
            V.col(0)[:] = self.v0
        norm, lindep = V.orthogonalize_column(i)
        if lindep or norm == 0.0:
            if i == 0:
                raise ArithmeticError("Initial vector is zero or belongs to the deflation space")
            return True
        V.scale_column(i, 1.0 / norm)
        return False

    def _full_lanczos(self, alpha, beta, k, m):
        """``EPSFullLanczos`` (epskrylov.c:347-376).  Returns (M, breakdown)."""
        V = self.V
        V.set_active_columns(0, m)
        M = m
        breakdown = False
        for j in range(k, m):
            self._apply(V.col(j), V.col(j + 1))
            nrm, breakdown = V.orthonormalize_column(j + 1, False)
            beta[j] = nrm
            if breakdown:
                M = j + 1
                break
        nc = V.nc
        ld = nc + V.m
        a = V.get_buffer()
        for j in range(k, M):
            alpha[j] = a[nc + j + (j + 1) * ld]
        return M, breakdown

    def _basic_arnoldi(self, H, k, m):
        """``EPSBasicArnoldi`` (epskrylov.c:39-74).  Returns (M, beta, breakdown)."""
        V = self.V
        V.set_active_columns(0, m)
        M = m
        beta = 0.0
        breakdown = False
        for j in range(k, m):
            self._apply(V.col(j), V.col(j + 1))
            beta, breakdown = V.orthonormalize_column(j + 1, False)
            if breakdown:
                M = j + 1
                break
        nc = V.nc
        ld = nc + V.m
        a = V.get_buffer()
        for j in range(k, M):
            H[:j + 2, j] = a[nc + (j + 1) * ld: nc + (j + 1) * ld + j + 2]
        return M, beta, breakdown

    def _vdot(self, x, y):
        return float(self.V.comm.allreduce_sum(np.array([x @ y]))[0])

    def _delayed_arnoldi(self, H, k, m):
        """``EPSDelayedArnoldi`` (epskrylov.c:82-186): the reorthogonalisation of column j
        and the normalisation of column j-1 are postponed to step j+1, so that every step
        needs one round of reductions only.  Returns (M, beta, breakdown)."""
        V = self.V
        u = np.zeros(self.n)
        t = np.zeros(self.n)
        lhh = np.zeros(m + 1)
        V.set_active_columns(0, m)
        for j in range(k, m):
            self._apply(V.col(j), V.col(j + 1))
            H[:j + 1, j] = V.dot_column(j + 1)
            if j > k:
                lhh[:j] = V.dot_column(j)
                dot = self._vdot(V.col(j), V.col(j))
            if j > k + 1:
                norm2 = np.sqrt(self._vdot(u, u))
            if j > k:
                norm1 = np.sqrt(dot)
                H[:j, j] = H[:j, j] / norm1
                H[j, j] = H[j, j] / dot
                t[:] = V.col(j)
                V.scale_column(j, 1.0 / norm1)
                V.scale_column(j + 1, 1.0 / norm1)
            V.mult_column(-1.0, 1.0, j + 1, H[:j + 1, j].copy())
            if j > k:
                V.set_active_columns(0, j)
                V.multvec(-1.0, 1.0, t, lhh[:j].copy())
                V.set_active_columns(0, m)
                H[:j, j - 1] += lhh[:j]
            if j > k + 1:
                V.col(j - 1)[:] = u
                V.scale_column(j - 1, 1.0 / norm2)
                H[j - 1, j - 2] = norm2
            if j < m - 1:
                u[:] = t
        norm2 = np.sqrt(self._vdot(t, t))
        t *= 1.0 / norm2
        V.col(m - 1)[:] = t
        H[m - 1, m - 2] = norm2
        lhh[:m] = V.dot_column(m)
        V.mult_column(-1.0, 1.0, m, lhh[:m].copy())
        H[:m, m - 1] += lhh[:m]
        beta = V.norm_column(m)
        V.scale_column(m, 1.0 / beta)
        return m, beta, False

    def _delayed_arnoldi1(self, H, k, m):
        """``EPSDelayedArnoldi1`` (epskrylov.c:192-233): delayed normalisation only, no
        reorthogonalisation (chosen when the BV refinement is "never")."""
        V = self.V
        beta = 0.0
        V.set_active_columns(0, m)
        for j in range(k, m):
            self._apply(V.col(j), V.col(j + 1))
            H[:j + 1, j] = V.dot_column(j + 1)
            if j > k:
                dot = self._vdot(V.col(j), V.col(j))
                norm = np.sqrt(dot)
                V.scale_column(j, 1.0 / norm)
                H[j, j - 1] = norm
                H[:j, j] = H[:j, j] / norm
                H[j, j] = H[j, j] / dot
                V.scale_column(j + 1, 1.0 / norm)
                beta = norm
            V.mult_column(-1.0, 1.0, j + 1, H[:j + 1, j].copy())
        return m, beta, False

    def _krylov_convergence(self, kini, nits, beta):
        """``EPSKrylovConvergence`` (epskrylov.c:252-327) with the relative test
        ``EPSConvergedRelative`` (epsdefault.c:258-266); trivial region, no true
        residual."""
        marker = -1
        k = kini
        while k < kini + nits:
            re, im = self.eigr[k], self.eigi[k]
            if self.st == "shift":                           # epskrylov.c:285-287
                re, im = self._back_transform(re, im)
            newk, resnorm = self.ds.vectors_rnorm(k)
            resnorm *= beta
            # EPSConvergedRelative / EPSConvergedAbsolute (epsdefault.c:258-280)
            self.errest[k] = resnorm if self.conv == "abs" else resnorm / dsmod.abs_eig(re, im)
            if marker == -1 and self.errest[k] >= self.tol:
                marker = k
            if newk == k + 1:
                self.errest[k + 1] = self.errest[k]
                k += 1
            if marker != -1:
                break
            k += 1
        if marker != -1:
            k = marker
        return k

    def _stopping(self, k):
        """``EPSStoppingBasic`` (epsdefault.c:324-338)."""
        if k >= self.nev:
            return CONVERGED_TOL
        if self.its >= self.max_it:
            return DIVERGED_ITS
        return ITERATING

    # -- solvers -----------------------------------------------------------------
    def _solve_symm(self):
        """``EPSSolve_KrylovSchur_Symm`` (ks-symm.c:30-111)."""
        ds, V = self.ds, self.V
        ld = ds.ld
        self._get_start_vector(0)
        l = 0
        while self.reason == ITERATING:
            self.its += 1
            nv = min(self.nconv + self.mpd, self.ncv)
            a = ds.T[:ld]
            b = ds.T[ld:2 * ld]
            nv, breakdown = self._full_lanczos(a, b, self.nconv + l, nv)
            beta = b[nv - 1]
            ds.set_dimensions(nv, self.nconv, self.nconv + l)
            ds.set_state(dsmod.STATE_INTERMEDIATE if l == 0 else dsmod.STATE_RAW)
            V.set_active_columns(self.nconv, nv)

            ds.solve(self.eigr)
            ds.sort(self.eigr)
            ds.update_extra_row()

            k = self._krylov_convergence(self.nconv, nv - self.nconv, beta)
            self.reason = self._stopping(k)
            nconv = k

            if self.reason != ITERATING or breakdown:
                l = 0
            else:
                l = max(1, int((nv - k) * self.keep))
            if not self.lock and l > 0:
                l += k
                k = 0
            if self.reason == ITERATING:
                if breakdown:
                    if k < self.nev:
                        if self._get_start_vector(k):
                            self.reason = DIVERGED_BREAKDOWN
                else:
                    ds.truncate(k + l)
            Q = ds.get_mat_Q()
            V.mult_in_place(Q, self.nconv, k + l)
            if self.reason == ITERATING and not breakdown:
                V.copy_column(nv, k + l)
            self.nconv = k
            if self.monitor:
                self.monitor(self.its, nconv, self.eigr, self.eigi, self.errest, nv)

    def _solve_default(self):
        """``EPSSolve_KrylovSchur_Default`` (krylovschur.c:169-288), Ritz extraction."""
        ds, V = self.ds, self.V
        self._get_start_vector(0)
        l = 0
        while self.reason == ITERATING:
            self.its += 1
            nv = min(self.nconv + self.mpd, self.ncv)
            S = ds.A
            nv, beta, breakdown = self._basic_arnoldi(S, self.nconv + l, nv)
            ds.set_dimensions(nv, self.nconv, self.nconv + l)
            ds.set_state(dsmod.STATE_INTERMEDIATE if l == 0 else dsmod.STATE_RAW)
            V.set_active_columns(self.nconv, nv)

            ds.solve(self.eigr, self.eigi)
            ds.sort(self.eigr, self.eigi)

            k = self._krylov_convergence(self.nconv, nv - self.nconv, beta)
            self.reason = self._stopping(k)
            nconv = k

            if self.reason != ITERATING or breakdown:
                l = 0
            else:
                l = max(1, int((nv - k) * self.keep))
                if S[k + l, k + l - 1] != 0.0:
                    if k + l < nv - 1:
                        l = l + 1
                    else:
                        l = l - 1
            if not self.lock and l > 0:
                l += k
                k = 0
            if self.reason == ITERATING:
                if breakdown:
                    if k < self.nev:
                        if self._get_start_vector(k):
                            self.reason = DIVERGED_BREAKDOWN
                else:
                    Q = ds.Q
                    for i in range(k, k + l):
                        S[k + l, i] = Q[nv - 1, i] * beta
            Qm = ds.get_mat("Q")
            V.mult_in_place(Qm, self.nconv, k + l)
            if self.reason == ITERATING and not breakdown:
                V.copy_column(nv, k + l)
            self.nconv = k
            if self.monitor:
                self.monitor(self.its, nconv, self.eigr, self.eigi, self.errest, nv)
        ds.set_dimensions(self.nconv, 0, 0)
        ds.set_state(dsmod.STATE_RAW)

    def _solve_arnoldi(self):
        """``EPSSolve_Arnoldi`` (src/eps/impls/krylov/arnoldi/arnoldi.c:78-154): explicitly
        restarted Arnoldi with locking, Ritz extraction, non-delayed orthogonalisation.  The
        projected problem is always the non-symmetric one (``EPSSetUp_Arnoldi``, :58)."""
        ds, V = self.ds, self.V
        self._get_start_vector(0)
        while self.reason == ITERATING:
            self.its += 1
            nv = min(self.nconv + self.mpd, self.ncv)
            ds.set_dimensions(nv, self.nconv, 0)
            H = ds.A
            if not self.delayed:
                nv, beta, breakdown = self._basic_arnoldi(H, self.nconv, nv)
            elif self.V.orthog_ref == REFINE_NEVER:
                nv, beta, breakdown = self._delayed_arnoldi1(H, self.nconv, nv)
            else:
                nv, beta, breakdown = self._delayed_arnoldi(H, self.nconv, nv)
            ds.set_dimensions(nv, self.nconv, 0)
            ds.set_state(dsmod.STATE_INTERMEDIATE)
            V.set_active_columns(self.nconv, nv)

            ds.solve(self.eigr, self.eigi)
            ds.sort(self.eigr, self.eigi)
            ds.update_extra_row()

            k = self._krylov_convergence(self.nconv, nv - self.nconv, beta)
            Qm = ds.get_mat("Q")
            V.mult_in_place(Qm, self.nconv, min(k + 1, nv))
            self.reason = self._stopping(k)
            if self.reason == ITERATING and breakdown:
                if self._get_start_vector(k):
                    self.reason = DIVERGED_BREAKDOWN
            self.nconv = k
            if self.monitor:
                self.monitor(self.its, self.nconv, self.eigr, self.eigi, self.errest, nv)
        ds.set_dimensions(self.nconv, 0, 0)
        ds.set_state(dsmod.STATE_RAW)

    # -- Lanczos recurrences with cheaper reorthogonalisation (lanczos.c:110-541) ------------
    def _local_lanczos(self, alpha, beta, k, m):
        """``EPSLocalLanczos`` (lanczos.c:110-148): against the two previous vectors (and the
        locked ones) only."""
        V = self.V
        which = [True] * k + [False] * (m + 1 - k)
        hwork = np.zeros(m + 1)
        V.set_active_columns(0, m)
        M, breakdown = m, False
        for j in range(k, m):
            self._apply(V.col(j), V.col(j + 1))
            which[j] = True
            if j - 2 >= k:
                which[j - 2] = False
            beta[j], breakdown = V.orthogonalize_some_column(j + 1, which, hwork)
            alpha[j] = hwork[j]
            if breakdown:
                M = j + 1
                break
            V.scale_column(j + 1, 1.0 / beta[j])
        return M, breakdown

    def _selective_lanczos(self, alpha, beta, k, m, anorm):
        """``EPSSelectiveLanczos`` (lanczos.c:213-288): local reorthogonalisation plus
        orthogonalisation against the nearly converged Ritz vectors."""
        V, AV = self.V, self.AV
        sqrteps = np.sqrt(np.finfo(float).eps)
        which = [True] * k + [False] * (m + 1 - k)
        hwork = np.zeros(m + 1)
        nritz = 0
        M, breakdown = m, False
        for j in range(k, m):
            V.set_active_columns(0, m)
            self._apply(V.col(j), V.col(j + 1))
            which[j] = True
            if j - 2 >= k:
                which[j - 2] = False
            norm, breakdown = V.orthogonalize_some_column(j + 1, which, hwork)
            alpha[j] = hwork[j]
            beta[j] = norm
            if breakdown:
                M = j + 1
                break
            n = j - k + 1
            T = np.diag(alpha[k:k + n]) + np.diag(beta[k:k + n - 1], 1) + np.diag(beta[k:k + n - 1], -1)
            ritz, Y = np.linalg.eigh(T)                       # DenseTridiagonal (stevr)
            for i in range(n):
                anorm = max(anorm, abs(ritz[i]))
            nritzo = sum(1 for i in range(n) if norm * abs(Y[n - 1, i]) < sqrteps * anorm)
            if nritzo > nritz:
                nritz = 0
                for i in range(n):
                    if norm * abs(Y[n - 1, i]) < sqrteps * anorm:
                        V.set_active_columns(k, k + n)
                        V.multvec(1.0, 0.0, AV.col(nritz), Y[:, i].copy())
                        nritz += 1
            if nritz > 0:
                AV.set_active_columns(0, nritz)
                norm, breakdown = AV.orthogonalize_vec(V.col(j + 1))
                if breakdown:
                    M = j + 1
                    break
            V.scale_column(j + 1, 1.0 / norm)
        return M, breakdown

    @staticmethod
    def _update_omega(omega, omega_old, j, alpha, beta, eps1, anorm):
        """``update_omega`` (lanczos.c:290-318); beta is indexed as the reference's beta-1."""
        T = eps1 * anorm
        binv = 1.0 / beta[j + 1]
        omega_old[0] = beta[1] * omega[1] + (alpha[0] - alpha[j]) * omega[0] - beta[j] * omega_old[0]
        omega_old[0] = binv * (omega_old[0] + T) if omega_old[0] > 0 else binv * (omega_old[0] - T)
        for k in range(1, j - 1):
            omega_old[k] = beta[k + 1] * omega[k + 1] + (alpha[k] - alpha[j]) * omega[k] + beta[k] * omega[k - 1] - beta[j] * omega_old[k]
            omega_old[k] = binv * (omega_old[k] + T) if omega_old[k] > 0 else binv * (omega_old[k] - T)
        omega_old[j - 1] = binv * T
        for k in range(j):
            omega[k] = omega_old[k]
            omega_old[k] = omega[k]
        omega[j] = eps1

    @staticmethod
    def _compute_int(which, off, mu, j, delta, eta):
        """``compute_int`` (lanczos.c:320-357) on which[off:off+j]."""
        found, maxpos, mx = False, 0, 0.0
        for i in range(j):
            if abs(mu[i]) >= delta:
                which[off + i] = True
                found = True
            else:
                which[off + i] = False
            if abs(mu[i]) > mx:
                maxpos, mx = i, abs(mu[i])
        if not found:
            which[off + maxpos] = True
        for i in range(j):
            if which[off + i]:
                for kk in range(i, -1, -1):
                    if abs(mu[kk]) < eta or which[off + kk]:
                        break
                    which[off + kk] = True
                for kk in range(i + 1, j):
                    if abs(mu[kk]) < eta or which[off + kk]:
                        break
                    which[off + kk] = True

    def _partial_lanczos(self, alpha, beta, k, m, anorm):
        """``EPSPartialLanczos`` (lanczos.c:359-468): periodic or partial reorthogonalisation
        steered by the omega recurrence that estimates the loss of orthogonality."""
        V = self.V
        macheps = np.finfo(float).eps
        omega = np.zeros(m + 2)
        omega_old = np.zeros(m + 2)
        which = [True] * k + [False] * (m + 1 - k)
        which2 = [False] * (m + 1)
        hwork = np.zeros(m + 1)
        reorth = force_reorth = fro = False
        estimate_anorm = False
        eps1 = np.sqrt(float(self.N)) * macheps / 2
        delta = np.sqrt(macheps) / np.sqrt(float(self.ncv))
        eta = macheps ** 0.75 / np.sqrt(float(self.ncv))
        if anorm < 0.0:
            anorm = 1.0
            estimate_anorm = True
        V.set_active_columns(0, m)
        M, breakdown = m, False
        for j in range(k, m):
            self._apply(V.col(j), V.col(j + 1))
            if fro:
                norm, breakdown = V.orthogonalize_column(j + 1, hwork)
                alpha[j] = hwork[j]
            else:
                which[j] = True
                if j - 2 >= k:
                    which[j - 2] = False
                norm, breakdown = V.orthogonalize_some_column(j + 1, which, hwork)
                alpha[j] = hwork[j]
                beta[j] = norm
                if estimate_anorm:
                    anorm = max(anorm, abs(alpha[j]) + norm + beta[j - 1]) if j > k else max(anorm, abs(alpha[j]) + norm)
                reorth = False
                if j > k:
                    # the reference passes alpha, beta-1 with absolute index j
                    bsh = np.concatenate(([0.0], beta))          # bsh[i] = beta[i-1]
                    self._update_omega(omega, omega_old, j, alpha, bsh, eps1, anorm)
                    for i in range(j - k):
                        if abs(omega[i]) > delta:
                            reorth = True
                if reorth or force_reorth:
                    for i in range(k):
                        which2[i] = False
                    for i in range(k, j + 1):
                        which2[i] = True
                    if self.reorthog == "periodic":
                        force_reorth = not force_reorth
                        for i in range(j - k):
                            omega[i] = eps1
                    else:
                        if force_reorth:
                            force_reorth = False
                        else:
                            force_reorth = True
                            self._compute_int(which2, k, omega, j - k, delta, eta)
                            for i in range(j - k):
                                if which2[i + k]:
                                    omega[i] = eps1
                    norm, breakdown = V.orthogonalize_some_column(j + 1, which2, hwork)
            if breakdown or norm < self.N * anorm * macheps:
                M = j + 1
                breakdown = True if breakdown else breakdown
                break
            if not fro and norm * delta < anorm * eps1:
                fro = True
            beta[j] = norm
            V.scale_column(j + 1, 1.0 / norm)
        return M, breakdown

    def _basic_lanczos(self, alpha, beta, k, m, anorm):
        """``EPSBasicLanczos`` (lanczos.c:470-541)."""
        r = self.reorthog
        if r == "local":
            return self._local_lanczos(alpha, beta, k, m)
        if r == "full":
            return self._full_lanczos(alpha, beta, k, m)
        if r == "selective":
            return self._selective_lanczos(alpha, beta, k, m, anorm)
        if r in ("periodic", "partial"):
            return self._partial_lanczos(alpha, beta, k, m, anorm)
        n = m
        T = np.zeros((n + 1, n + 1))
        if self.V.orthog_ref == REFINE_NEVER:
            _, betam, breakdown = self._delayed_arnoldi1(T, k, n)
        else:
            _, betam, breakdown = self._delayed_arnoldi(T, k, n)
        for i in range(k, n - 1):
            alpha[i] = T[i, i]
            beta[i] = T[i + 1, i]
        alpha[n - 1] = T[n - 1, n - 1]
        beta[n - 1] = betam
        return n, breakdown

    def _solve_lanczos(self):
        """``EPSSolve_Lanczos`` (src/eps/impls/krylov/lanczos/lanczos.c:543-726) with full
        reorthogonalisation (``EPSBasicLanczos`` -> ``EPSFullLanczos``, :499-541): explicit
        restart with the wanted unconverged Ritz vector, converged vectors locked in front."""
        ds, V = self.ds, self.V
        ds.extrarow = False
        ncv = self.ncv
        ritz = np.zeros(ncv + 1)
        bnd = np.zeros(ncv + 1)
        conv = ["N"] * (ncv + 1)
        perm = np.zeros(ncv + 1, dtype=int)
        cmpf = self.cmpf
        macheps = np.finfo(float).eps
        self._get_start_vector(0)
        anorm = -1.0
        nconv = 0
        while self.reason == ITERATING:
            self.its += 1
            n = min(nconv + self.mpd, ncv)
            d, e = ds.d, ds.e
            n, breakdown = self._basic_lanczos(d, e, nconv, n, anorm)
            beta = e[n - 1]
            ds.set_dimensions(n, nconv, 0)
            ds.set_state(dsmod.STATE_INTERMEDIATE)
            V.set_active_columns(nconv, n)
            ds.solve(ritz)
            ds.sort(ritz)
            for i in range(nconv, n):
                anorm = max(anorm, abs(ritz[i]))
            Y = ds.Q
            for i in range(nconv, n):
                resnorm = beta * abs(Y[n - 1, i]) + macheps * anorm
                bnd[i] = resnorm / abs(ritz[i])                    # EPSConvergedRelative
                conv[i] = "C" if bnd[i] < self.tol else "N"
            if self.reorthog == "local":                     # purge repeated ritz values (:598-603)
                for i in range(nconv + 1, n):
                    if conv[i] == "C" and abs((ritz[i] - ritz[i - 1]) / ritz[i]) < self.tol:
                        conv[i] = "R"
            if not breakdown:
                restart = nconv
                while restart < n and conv[restart] != "N":
                    restart += 1
                if restart >= n:
                    breakdown = True
                else:
                    for i in range(restart + 1, n):
                        if conv[i] == "N" and cmpf(ritz[restart], 0.0, ritz[i], 0.0) > 0:
                            restart = i
                    V.set_active_columns(nconv, n)
                    V.mult_column(1.0, 0.0, n, np.ascontiguousarray(Y[nconv:n, restart]))
                    V.set_active_columns(nconv, n)
            # converged Ritz pairs first
            for i in range(nconv, n):
                perm[i] = i
            k = nconv
            while k < n:
                if conv[perm[k]] != "C":
                    j = k + 1
                    while j < n and conv[perm[j]] != "C":
                        j += 1
                    if j >= n:
                        break
                    perm[k], perm[j] = perm[j], perm[k]
                k += 1
            for i in range(nconv, k):
                x = perm[i]
                if x != i:
                    j = i + 1
                    while perm[j] != i:
                        j += 1
                    ritz[x], ritz[i] = ritz[i], ritz[x]
                    bnd[x], bnd[i] = bnd[i], bnd[x]
                    conv[x], conv[i] = conv[i], conv[x]
                    perm[j] = x
                    perm[i] = i
                    tmp = Y[:n, x].copy()
                    Y[:n, x] = Y[:n, i]
                    Y[:n, i] = tmp
            V.mult_in_place(ds.get_mat_Q(), nconv, k)
            if self.reorthog == "local":                     # purge spurious ritz values (:661-695)
                w = np.zeros(self.n)
                for i in range(nconv, k):
                    vi = V.col(i)
                    vi *= 1.0 / np.sqrt(float(V.comm.allreduce_sum(np.array([vi @ vi]))[0]))
                    self._apply(vi, w)
                    w -= ritz[i] * vi
                    norm = np.sqrt(float(V.comm.allreduce_sum(np.array([w @ w]))[0]))
                    bnd[i] = norm / abs(ritz[i])
                    if bnd[i] >= self.tol:
                        conv[i] = "S"
                i = nconv
                while i < k:
                    if conv[i] != "C":
                        j = i + 1
                        while j < k and conv[j] != "C":
                            j += 1
                        if j >= k:
                            break
                        ritz[j], ritz[i] = ritz[i], ritz[j]
                        bnd[j], bnd[i] = bnd[i], bnd[j]
                        conv[j], conv[i] = conv[i], conv[j]
                        tmp = V.col(i).copy()
                        V.col(i)[:] = V.col(j)
                        V.col(j)[:] = tmp
                    i += 1
                k = i
            for i in range(nconv, n):
                self.eigr[i] = ritz[i]
                self.errest[i] = bnd[i]
            nconv = k
            if self.monitor:
                self.monitor(self.its, nconv, self.eigr, self.eigi, self.errest, n)
            self.reason = self._stopping(nconv)
            if self.reason == ITERATING:
                V.copy_column(n, nconv)
                if self.reorthog == "local" and not breakdown:
                    _, breakdown = V.orthonormalize_column(nconv, False)
                if breakdown:
                    if self._get_start_vector(nconv):
                        self.reason = DIVERGED_BREAKDOWN
        self.nconv = nconv

    def solve(self):
        """``EPSSolve`` (epssolve.c:63-146)."""
        self.setup()
        self.nconv = 0
        self.its = 0
        self.reason = ITERATING
        if self.solver == "arnoldi":
            self._solve_arnoldi()
        elif self.solver == "lanczos":
            if not self.hermitian:
                raise ValueError("Requested method is only available for Hermitian problems")
            self._solve_lanczos()
        elif self.hermitian:
            self._solve_symm()
        else:
            self._solve_default()
        self._vectors_done = False
        for i in range(self.nconv):                          # EPSBackTransform_Default (epsdefault.c:27-34)
            self.eigr[i], self.eigi[i] = self._back_transform(self.eigr[i], self.eigi[i])
        # reorder conjugate eigenvalues, positive imaginary part first (:106-120)
        i = 0
        while i < self.nconv - 1:
            if self.eigi[i] != 0:
                if self.eigi[i] < 0:
                    self.eigi[i] = -self.eigi[i]
                    self.eigi[i + 1] = -self.eigi[i + 1]
                    self.compute_vectors()
                    self.V.scale_column(i + 1, -1.0)
                i += 1
            i += 1
        self.perm = np.arange(self.ncv + 1)
        dsmod.sort_eigenvalues(self.cmp0, self.nconv, self.eigr, self.eigi, self.perm)
        return self

    # -- post-solve ------------------------------------------------------------------
    def compute_vectors(self):
        """``EPSComputeVectors`` -> ``EPSComputeVectors_Hermitian`` (no-op for a
        standard HEP, epsdefault.c:40-74) or ``EPSComputeVectors_Schur``
        (epsdefault.c:136-205)."""
        if self._vectors_done:
            return
        self._vectors_done = True
        if self.hermitian and self.solver in ("krylovschur", "lanczos"):
            if self.purify:                                  # epsdefault.c:50-56
                self._purify(self.nconv)
                for i in range(self.nconv):
                    self.V.scale_column(i, 1.0 / self.V.norm_column(i))
            return
        n = self.ds.n
        if n == 0:
            return
        self.ds.vectors_all()
        Z = self.ds.get_mat("X")
        self.V.set_active_columns(0, n)
        self.V.mult_in_place(Z, 0, n)
        if self.purify:                                      # epsdefault.c:168-205
            self._purify(self.nconv)

            def nrm2(v):
                return float(np.sqrt(self.comm.allreduce_sum(np.array([v @ v]))[0]))
            i = 0
            while i < n:
                if self.eigi[i] != 0.0 and i + 1 < n:
                    t = 1.0 / dsmod.abs_eig(nrm2(self.V.col(i)), nrm2(self.V.col(i + 1)))
                    self.V.col(i)[:] *= t
                    self.V.col(i + 1)[:] *= t
                    i += 1
                else:
                    nr = nrm2(self.V.col(i))
                    if nr != 0.0:
                        self.V.col(i)[:] /= nr
                i += 1

    def _purify(self, k):
        """``EPS_Purify`` (epsimpl.h:166-183)."""
        for i in range(k):
            v = self.V.col(i).copy()
            self._apply(v, self.V.col(i))

    def get_eigenvalue(self, i):
        """``EPSGetEigenvalue`` (epssolve.c:370-390)."""
        k = self.perm[i]
        return self.eigr[k], self.eigi[k]

    def get_eigenvector(self, i):
        """``EPSGetEigenvector`` (epssolve.c:425-462): returns (xr, xi)."""
        self.compute_vectors()
        k = self.perm[i]
        if self.eigi[k] > 0:
            return self.V.col(k).copy(), self.V.col(k + 1).copy()
        if self.eigi[k] < 0:
            return self.V.col(k - 1).copy(), -self.V.col(k).copy()
        return self.V.col(k).copy(), np.zeros(self.n)

    def compute_error(self, i, relative=True):
        """``EPSComputeError`` (epssolve.c:580-649) with
        ``EPSComputeResidualNorm_Private`` (:506-556)."""
        kr, ki = self.get_eigenvalue(i)
        xr, xi = self.get_eigenvector(i)
        u = np.zeros(self.n)

        def nrm2(v):
            return float(np.sqrt(self.comm.allreduce_sum(np.array([v @ v]))[0]))

        def bmul(x):
            if not self.generalized:
                return x
            y = np.zeros(self.n)
            self.Bmatvec(x, y)
            return y

        if ki == 0 or abs(ki) < abs(kr * EPSILON):
            self.matvec(xr, u)
            if abs(kr) > EPSILON:
                u -= kr * bmul(xr)
            err = nrm2(u)
        else:
            self.matvec(xr, u)
            bxr, bxi = bmul(xr), bmul(xi)
            if dsmod.abs_eig(kr, ki) > EPSILON:
                u -= kr * bxr
                u += ki * bxi
            nr = nrm2(u)
            self.matvec(xi, u)
            if dsmod.abs_eig(kr, ki) > EPSILON:
                u -= kr * bxi
                u -= ki * bxr
            ni = nrm2(u)
            err = dsmod.abs_eig(nr, ni)
        if relative:
            err /= dsmod.abs_eig(kr, ki)
        return err
