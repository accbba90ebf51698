"""CPU restatement of the reference's BV class (svec storage) -- the part of
it that the Krylov-Schur solver exercises.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Storage follows ``src/sys/classes/bv/impls/svec/svec.c``: all columns in one
column-major array, constraints first, element (r, column j) at
``base[(nc+j)*n + r]``.  Kernels follow ``src/sys/classes/bv/interface/bvblas.c``
(BLAS gemv/gemm through numpy); Gram-Schmidt follows
``src/sys/classes/bv/interface/bvorthog.c``.
"""
import numpy as np

EPSILON = np.finfo(np.float64).eps

CGS, MGS = 0, 1
REFINE_IFNEEDED, REFINE_NEVER, REFINE_ALWAYS = 0, 1, 2
NORM_1, NORM_2, NORM_FROBENIUS, NORM_INFINITY = 0, 1, 2, 3


class SelfComm:
    """Single-rank communicator: reductions are the identity."""
    rank = 0
    size = 1

    def allreduce_sum(self, a):
        return a

    def allreduce_max(self, a):
        return a


class BV:
    """Basis vectors, svec layout (``svec.c:417-520``).  ``n`` = local rows,
    ``m`` = columns, ``nc`` = constraints; active columns are ``[l,k)``
    (``bvbasic.c:405-425``)."""

    def __init__(self, n, m, N=None, comm=None, nc=0):
        self.n = int(n)
        self.N = int(N if N is not None else n)
        self.m = int(m)
        self.nc = int(nc)
        self.l = 0
        self.k = self.m
        self.comm = comm if comm is not None else SelfComm()
        self.array = np.zeros((self.n, self.nc + self.m), order="F")
        # defaults: bvfunc.c:184-187
        self.orthog_type = CGS
        self.orthog_ref = REFINE_IFNEEDED
        self.orthog_eta = 0.7071
        self.buffer = None
        # non-standard inner product <x,y> = y^T B x (bvimpl.h:80-88): matvec(x) -> B x on the
        # local rows, indef = B indefinite, omega = signature (nc+m entries), cached = B*X
        self.matrix = None
        self.indef = False
        self.omega = None
        self.Bx = None
        self.cached = None
        # counters used by the byte/flop model of the benchmark
        self.npasses = 0
        self.bytes = 0.0     # algorithmic bytes moved by the op table (same model as the device library)

    # ---- views -------------------------------------------------------
    def col(self, j):
        """Column view, ``svec.c:312-324`` (``pv+(nc+j)*n``)."""
        return self.array[:, self.nc + j]

    def active(self):
        return self.array[:, self.nc + self.l:self.nc + self.k]

    def set_active_columns(self, l, k):
        """``bvbasic.c:405-425``."""
        self.l, self.k = int(l), int(k)

    def get_buffer(self):
        """``bvbasic.c:806-827``: length (nc+m)*m, leading dimension nc+m."""
        if self.buffer is None:
            self.buffer = np.zeros((self.nc + self.m) * self.m)
        return self.buffer

    # ---- inner product matrix, constraints (bvbasic.c:262-321,481-729; bvfunc.c:414-442) ----
    def set_matrix(self, matvec, indef=False):
        """``BVSetMatrix`` (bvbasic.c:481-504): ``matvec(x)`` returns B x for the local rows."""
        self.matrix = matvec
        self.indef = bool(indef)

    def _safe_sqrt(self, alpha):
        """``BV_SafeSqrt`` (bvimpl.h:113-134)."""
        if self.indef:
            return -np.sqrt(-alpha) if alpha < 0.0 else np.sqrt(alpha)
        if alpha < 0.0:
            raise ArithmeticError("The inner product is not well defined: indefinite matrix")
        return np.sqrt(alpha)

    def _ipmatmult(self, x):
        """``BV_IPMatMult`` (bvimpl.h:140-151), without the cache (same values)."""
        self.Bx = np.array(self.matrix(x), dtype=np.float64)
        return self.Bx

    def _allocate_signature(self):
        """``BV_AllocateSignature`` (bvimpl.h:193-205)."""
        if self.indef and self.omega is None:
            self.omega = np.ones(self.nc + self.m)

    def get_signature(self):
        """``BVGetSignature`` (bvbasic.c:707-729)."""
        if self.indef and self.omega is not None:
            return self.omega[self.nc:self.nc + self.k].copy()
        return np.ones(self.k)

    def set_signature(self, omega):
        """``BVSetSignature`` (bvbasic.c:664-687)."""
        self._allocate_signature()
        if self.indef:
            self.omega[self.nc:self.nc + self.k] = omega[:self.k]

    def apply_matrix(self, x):
        """``BVApplyMatrix`` (bvbasic.c:551-566)."""
        return self._ipmatmult(x).copy() if self.matrix is not None else np.array(x, dtype=np.float64)

    def _ipmatmult_bv(self):
        """``BV_IPMatMultBV`` (bvimpl.h:157-173): cached = B*X on the active columns."""
        if self.cached is None:
            self.cached = BV(self.n, self.m, N=self.N, comm=self.comm)
        self.cached.set_active_columns(self.l, self.k)
        for j in range(self.l, self.k):
            self.cached.col(j)[:] = self.matrix(self.col(j)) if self.matrix is not None else self.col(j)
        return self.cached

    def dot(self, Y, M):
        """``BVDot`` (bvglobal.c:92-137): M = Y^T B X through the cached B X."""
        if self.matrix is not None:
            return self._ipmatmult_bv().op_dot(Y, M)
        return self.op_dot(Y, M)

    def norm_vec(self, v):
        """``BVNormVec`` / ``BVNorm_Private`` (bvglobal.c:470-480,570-597)."""
        if self.matrix is not None:
            p = float(self.comm.allreduce_sum(np.array([self._ipmatmult(v) @ v]))[0])
            return float(self._safe_sqrt(p))
        return float(np.sqrt(self.comm.allreduce_sum(np.array([v @ v]))[0]))

    def insert_constraints(self, C):
        """``BVInsertConstraints`` (bvfunc.c:414-442): the vectors of C, orthonormalised, become
        the constraints; linearly dependent ones are dropped.  Returns the number kept.
        Destroys the contents of the columns."""
        if self.nc:
            raise ValueError("Constraints already present in this BV object")
        nc, msave = len(C), self.m
        if not nc:
            return 0
        self.m = nc + msave                                  # BVResize(V, nc+m, FALSE)
        self.array = np.zeros((self.n, self.m), order="F")
        self.l, self.k = 0, self.m
        self.buffer = None
        self.omega = None
        ndep = 0                                             # BVInsertVecs(V,0,&nc,C,TRUE), bvfunc.c:331-377
        for i in range(nc):
            j = i - ndep
            self.col(j)[:] = C[i]
            norm, lindep = self.orthogonalize_column(j)
            if norm == 0.0 or lindep:
                ndep += 1
            else:
                self.scale_column(j, 1.0 / norm)
        self.nc = nc - ndep
        self.m = msave
        self.l, self.k = 0, msave
        self.buffer = None
        self.omega = None
        return self.nc

    # ---- op table (svec.c) -------------------------------------------
    def op_dotvec(self, y, q=None):
        """``BVDotVec_Svec`` (svec.c:122-143) -> ``BVDotVec_BLAS_Private``
        (bvblas.c:242-266): q = X(:,l:k)^T y, then SUM-allreduce."""
        kk = self.k - self.l
        self.bytes += 8.0 * self.n * (kk + 1)
        X = self.array[:, self.nc + self.l:self.nc + self.k]
        loc = X.T @ y if self.n else np.zeros(kk)
        res = self.comm.allreduce_sum(np.asarray(loc, dtype=np.float64))
        if q is None:
            self.get_buffer()[:kk] = res
            return self.buffer[:kk]
        q[:kk] = res
        return q

    def op_multvec(self, alpha, beta, y, q=None):
        """``BVMultVec_Svec`` (svec.c:51-66) -> gemv 'N' (bvblas.c:65-76):
        y = beta*y + alpha*X(:,l:k) q."""
        kk = self.k - self.l
        self.bytes += 8.0 * self.n * (kk + 2)
        qq = self.get_buffer()[:kk] if q is None else np.asarray(q)[:kk]
        X = self.array[:, self.nc + self.l:self.nc + self.k]
        if self.n:
            if beta == 0.0:
                y[:] = alpha * (X @ qq)
            else:
                if beta != 1.0:
                    y *= beta
                y += alpha * (X @ qq)

    def op_multinplace(self, Q, s, e, trans=False):
        """``BVMultInPlace_Svec`` (svec.c:68-83) ->
        ``BVMultInPlace_BLAS_Private`` (bvblas.c:83-119):
        V(:,s:e) = V(:,l:k) * Q(l:k, s:e)   (Q indexed absolutely)."""
        l, k = self.l, self.k
        self.bytes += 8.0 * self.n * ((k - l) + (e - s))
        A = self.array[:, self.nc + l:self.nc + k]
        if trans:
            B = Q[s:e, l:k].T
        else:
            B = Q[l:k, s:e]
        if self.n and e > s:
            self.array[:, self.nc + s:self.nc + e] = A @ B

    def op_scale(self, j, alpha):
        """``BVScale_Svec`` (svec.c:165-180), ``BVScale_BLAS_Private``
        (bvblas.c:271-285)."""
        tgt = self.active() if j < 0 else self.array[:, self.nc + j:self.nc + j + 1]
        self.bytes += 16.0 * tgt.size
        if alpha == 0.0:
            tgt[...] = 0.0
        elif alpha != 1.0:
            tgt *= alpha

    def op_norm(self, j, ntype=NORM_2):
        """``BVNorm_Svec`` (svec.c:182-197), ``BVNorm_LAPACK_Private``
        (bvblas.c:290-340)."""
        A = self.active() if j < 0 else self.array[:, self.nc + j:self.nc + j + 1]
        self.bytes += 8.0 * A.size
        if ntype in (NORM_2, NORM_FROBENIUS):
            loc = np.array([float(np.sum(A * A))])
            return float(np.sqrt(self.comm.allreduce_sum(loc)[0]))
        if ntype == NORM_1:
            cs = self.comm.allreduce_sum(np.abs(A).sum(axis=0))
            return float(cs.max()) if cs.size else 0.0
        if ntype == NORM_INFINITY:
            loc = np.array([np.abs(A).sum(axis=1).max() if A.size else 0.0])
            return float(self.comm.allreduce_max(loc)[0])
        raise ValueError("norm type")

    def op_mult(self, alpha, beta, X, Q):
        """``BVMult_Svec`` (svec.c:27-49): Y = beta*Y + alpha*X*Q, with
        ``Q + Y->l*ldq + X->l`` and sizes (X.k-X.l) x (Y.k-Y.l)."""
        Y = self
        Ya = Y.active()
        Xa = X.active()
        if Q is None:
            if beta != 1.0:
                Ya *= beta
            Ya += alpha * Xa
            return
        B = Q[X.l:X.k, Y.l:Y.k]
        if beta == 0.0:
            Ya[...] = alpha * (Xa @ B)
        else:
            Ya *= beta
            Ya += alpha * (Xa @ B)

    def op_dot(self, Y, M):
        """``BVDot_Svec`` (svec.c:102-120): M(Y.l:Y.k, X.l:X.k) = Y^T X."""
        X = self
        loc = Y.active().T @ X.active()
        tot = self.comm.allreduce_sum(np.ascontiguousarray(loc).ravel()).reshape(loc.shape)
        M[Y.l:Y.k, X.l:X.k] = tot
        return M

    # ---- interface wrappers ------------------------------------------
    def dotvec(self, y, q=None):
        """``BVDotVec`` (bvglobal.c:168-188); with a matrix the type works on B y (svec.c:131-134)."""
        return self.op_dotvec(self._ipmatmult(y) if self.matrix is not None else y, q)

    def multvec(self, alpha, beta, y, q=None):
        """``BVMultVec`` (bvops.c:125-149)."""
        self.op_multvec(alpha, beta, y, q)

    def dot_column(self, j, q=None):
        """``BVDotColumn`` (bvglobal.c:324-349): k is set to j for the call."""
        ksave = self.k
        self.k = j
        r = self.dotvec(self.col(j), q)
        self.k = ksave
        return r

    def _dot_column_inc(self, j, q=None):
        """``BVDotColumnInc`` (bvorthog.c:176-192): also column j itself."""
        ksave = self.k
        self.k = j + 1
        r = self.dotvec(self.col(j), q)
        self.k = ksave
        return r

    def mult_column(self, alpha, beta, j, q=None):
        """``BVMultColumn`` (bvops.c:179-206)."""
        ksave = self.k
        self.k = j
        self.op_multvec(alpha, beta, self.col(j), q)
        self.k = ksave

    def mult_in_place(self, Q, s, e):
        """``BVMultInPlace`` (bvops.c:234-263)."""
        if s >= e:
            return
        self.op_multinplace(Q, s, e, False)

    def norm_column(self, j, ntype=NORM_2):
        """``BVNormColumn`` (bvglobal.c:704-730): sqrt(z^T B z) with a matrix, whatever the type."""
        if self.matrix is not None:
            return self.norm_vec(self.col(j))
        return self.op_norm(j, ntype)

    def norm(self, ntype=NORM_FROBENIUS):
        """``BVNorm`` (bvglobal.c:520-545): norm of the active block."""
        return self.op_norm(-1, ntype)

    def scale_column(self, j, alpha):
        """``BVScaleColumn`` (bvops.c:367-387)."""
        if alpha == 1.0:
            return
        self.op_scale(j, alpha)

    def scale(self, alpha):
        """``BVScale`` (bvops.c:330-350)."""
        if alpha == 1.0:
            return
        self.op_scale(-1, alpha)

    def copy_column(self, j, i):
        """``BVCopyColumn`` (bvbasic.c:1628-1650)."""
        if j == i:
            return
        self.bytes += 16.0 * self.n
        self.col(i)[:] = self.col(j)

    def insert_vec(self, j, w):
        """``BVInsertVec`` (bvfunc.c:280-305)."""
        self.col(j)[:] = w

    # ---- Gram-Schmidt (bvorthog.c) -------------------------------------
    def _hcol(self, j):
        ld = self.nc + self.m
        return self.get_buffer()[j * ld:(j + 1) * ld]

    def _cgs1(self, j, want_norms):
        """``BVOrthogonalizeCGS1`` (bvorthog.c:233-277) for a column (v==NULL,
        h==c==NULL -> coefficients live in the buffer: column 0 = scratch c,
        column j = accumulated h)."""
        nc = self.nc
        buf = self.get_buffer()
        self.k = j
        self.npasses += 1
        onorm = norm = None
        if want_norms:
            self._dot_column_inc(j, None)               # :243
            beta = self._safe_sqrt(buf[nc + j])         # BV_SquareRoot :126-136
        else:
            self.dot_column(j, None)                    # :250
        if self.indef:                                  # BV_ApplySignature :108-120, :255
            buf[:nc + j] /= self.omega[:nc + j]
        self.mult_column(-1.0, 1.0, j, None)            # :256
        if self.indef:                                  # :258
            buf[:nc + j] *= self.omega[:nc + j]
        if want_norms:
            onorm = beta
            if self.indef:                              # :264-265
                norm = self.norm_column(j, NORM_2)
            else:
                ssum = 0.0                              # BV_SquareSum :89-101
                for i in range(nc + j):
                    ssum += buf[i] * buf[i]
                norm = beta * beta - ssum               # :269
                if norm <= 0.0:
                    norm = self.norm_column(j, NORM_2)  # :270-271
                else:
                    norm = np.sqrt(norm)
        h = self._hcol(j)                               # BV_AddCoefficients :50-64
        for i in range(nc + j):
            h[i] += buf[i]
        return onorm, norm

    def _mgs1(self, j, want_onorm, want_norm):
        """``BVOrthogonalizeMGS1`` (bvorthog.c:197-227), column variant; ``self._which``
        (set by ``orthogonalize_some_column``) restricts the columns used."""
        nc = self.nc
        buf = self.get_buffer()
        w = self.col(j)
        which = getattr(self, "_which", None)
        onrm = nrm = None
        if want_onorm:
            onrm = self.norm_vec(w)
        if which is not None:
            buf[:nc + j] = 0.0
        for i in range(-nc, j):
            if which is not None and i >= 0 and not which[i]:
                continue
            vi = self.col(i)
            z = self._ipmatmult(w) if self.matrix is not None else w     # :212-215
            dot = float(self.comm.allreduce_sum(np.array([z @ vi]))[0])
            buf[nc + i] = dot
            if self.indef:                                               # :219
                dot /= self.omega[nc + i]
            w -= dot * vi
        if want_norm:
            nrm = self.norm_vec(w)
        h = self._hcol(j)
        for i in range(nc + j):
            h[i] += buf[i]
        return onrm, nrm

    def _gs(self, j, want_norm=True, want_lindep=True):
        """``BVOrthogonalizeGS`` (bvorthog.c:290-361), column variant.  Returns (norm, lindep);
        with an indefinite inner product lindep is not computed, the norm always is, and its
        sign goes to the signature (:311-316,352)."""
        nc = self.nc
        mgs = self.orthog_type == MGS
        self._allocate_signature()
        signature = self.indef
        if self.indef:
            want_lindep = False
        k = j
        h = self._hcol(k)
        for i in range(nc + k):                          # BV_CleanCoefficients :30-44
            h[i] = 0.0
        lindep = False
        nrm = None
        if self.orthog_ref == REFINE_IFNEEDED:           # :322-333
            if mgs:
                onrm, nrm = self._mgs1(k, True, True)
            else:
                onrm, nrm = self._cgs1(k, True)
            l = 1
            while l < 3 and nrm and nrm < self.orthog_eta * onrm:
                l += 1
                if mgs:
                    onrm = nrm
                    _, nrm = self._mgs1(k, False, True)
                else:
                    onrm, nrm = self._cgs1(k, True)
            if want_lindep:
                lindep = not (nrm and nrm >= self.orthog_eta * onrm)
        elif self.orthog_ref == REFINE_NEVER:            # :335-343
            if mgs:
                self._mgs1(k, False, False)
            else:
                self._cgs1(k, False)
            if want_norm or want_lindep or signature:
                nrm = self.norm_column(k, NORM_2)
            if want_lindep:
                lindep = not nrm
        else:                                            # REFINE_ALWAYS :345-350
            if mgs:
                self._mgs1(k, False, False)
                onrm, nrm = self._mgs1(k, want_lindep, want_norm or want_lindep or signature)
            else:
                self._cgs1(k, False)
                onrm, nrm = self._cgs1(k, want_norm or want_lindep or signature)
            if want_lindep:
                lindep = not (nrm and nrm >= self.orthog_eta * onrm)
        if signature:                                    # :352
            self.omega[nc + k] = -1.0 if nrm < 0.0 else 1.0
        if want_norm:                                    # :353-358
            h[nc + k] = 0.0 if (want_lindep and lindep) else nrm
        return nrm, lindep

    def orthogonalize_column(self, j, H=None):
        """``BVOrthogonalizeColumn`` (bvorthog.c:460-485).  Returns (norm, lindep);
        if H is given, H[0:j-l] receives the coefficients (BV_StoreCoefficients
        :142-156)."""
        ksave, lsave = self.k, self.l
        self.l = -self.nc
        nrm, lindep = self._gs(j, True, True)
        self.k, self.l = ksave, lsave
        if H is not None:
            h = self._hcol(j)
            for i in range(self.l, j):
                H[i - self.l] = h[self.nc + i]
        return nrm, lindep

    def orthogonalize_vec(self, v):
        """``BVOrthogonalizeVec`` (bvorthog.c:391-418) with modified Gram-Schmidt: v against the
        active columns [0,k) in place, refinement as configured.  Returns (norm, lindep)."""
        if self.orthog_type != MGS:
            raise ValueError("only the MGS variant is restated")
        k = self.k

        def vnorm():
            return float(np.sqrt(self.comm.allreduce_sum(np.array([v @ v]))[0]))

        def mgs1(want_onorm, want_norm):
            onrm = vnorm() if want_onorm else None
            for i in range(-self.nc, k):
                vi = self.col(i)
                dot = float(self.comm.allreduce_sum(np.array([v @ vi]))[0])
                v[:] -= dot * vi
            return onrm, (vnorm() if want_norm else None)
        if self.orthog_ref == REFINE_IFNEEDED:
            onrm, nrm = mgs1(True, True)
            l = 1
            while l < 3 and nrm and nrm < self.orthog_eta * onrm:
                l += 1
                onrm = nrm
                _, nrm = mgs1(False, True)
            lindep = not (nrm and nrm >= self.orthog_eta * onrm)
        elif self.orthog_ref == REFINE_NEVER:
            mgs1(False, False)
            nrm = vnorm()
            lindep = not nrm
        else:
            mgs1(False, False)
            onrm, nrm = mgs1(True, True)
            lindep = not (nrm and nrm >= self.orthog_eta * onrm)
        return nrm, lindep

    def orthogonalize_some_column(self, j, which, H=None):
        """``BVOrthogonalizeSomeColumn`` (bvorthog.c:587-633): modified Gram-Schmidt against
        the columns i < j with which[i] set (and the constraints).  Returns (norm, lindep)."""
        if self.orthog_type != MGS:
            raise ValueError("Operation only available for MGS orthogonalization")
        ksave, lsave = self.k, self.l
        self.l = -self.nc
        self._which = which
        try:
            nrm, lindep = self._gs(j, True, True)
        finally:
            self._which = None
        self.k, self.l = ksave, lsave
        if H is not None:
            h = self._hcol(j)
            for i in range(self.l, j):
                H[i - self.l] = h[self.nc + i]
        return nrm, lindep

    def orthonormalize_column(self, j, replace=False, rand=None):
        """``BVOrthonormalizeColumn`` (bvorthog.c:523-572).  Returns (norm, lindep)."""
        ksave, lsave = self.k, self.l
        self.l = -self.nc
        nrm, lndep = self._gs(j, True, True)
        if replace and (nrm == 0.0 or lndep):
            for _ in range(2):
                self.col(j)[:] = rand(j)
                nrm, lndep = self._gs(j, True, True)
                if not (nrm == 0.0 or lndep):
                    break
        self.k, self.l = ksave, lsave
        if nrm != 1.0 and nrm != 0.0:
            if self.n:
                self.op_scale(j, 1.0 / nrm)
        return nrm, lndep


def orthogonalize_block(V, method="gs", want_R=False):
    """``BVOrthogonalize`` (bvorthog.c:768-859) of the active columns of the oracle BV ``V``:
    "gs" column by column (``BVOrthogonalize_GS``, :635-680) or "chol" (``BVOrthogonalize_Chol``
    with ``MatCholeskyFactorInvert``, :683-765).  Returns R (k x k, only the block [l,k) set)
    when asked for."""
    l, k = V.l, V.k
    R = np.zeros((k, k))
    if method == "gs":
        for j in range(l, k):
            H = np.zeros(k)
            norm, _ = V.orthogonalize_column(j, H)
            if not norm:
                raise ArithmeticError("Breakdown in BVOrthogonalize due to a linearly dependent column")
            R[l:j, j] = H[:j - l]
            R[j, j] = norm
            V.scale_column(j, 1.0 / norm)
    elif method == "chol":
        V.op_dot(V, R)
        G = R[l:k, l:k]
        try:
            U = np.linalg.cholesky(G).T
        except np.linalg.LinAlgError:
            U = np.linalg.cholesky(G + 50.0 * EPSILON * np.eye(k - l)).T
        R[l:k, l:k] = U
        S = np.zeros((k, k))
        S[l:k, l:k] = np.triu(np.linalg.inv(U))
        V.op_multinplace(S, l, k, False)
    else:
        raise ValueError(method)
    return R if want_R else None
