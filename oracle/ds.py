"""CPU restatement of the reference's DS class for the two projected problems
Krylov-Schur uses: DSHEP (compact arrow + tridiagonal storage, extra row) and
DSNHEP.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``src/sys/classes/ds/impls/hep/dshep.c``,
``src/sys/classes/ds/impls/nhep/dsnhep.c``,
``src/sys/classes/ds/interface/dsops.c`` and ``dspriv.c``; the LAPACK routines
are the ones the reference calls (``oracle/lapack.py``).
"""
import numpy as np

from . import lapack

STATE_RAW, STATE_INTERMEDIATE, STATE_CONDENSED, STATE_TRUNCATED = 0, 1, 2, 3


def abs_eig(re, im):
    """``SlepcAbsEigenvalue`` (include/slepcmath.h): hypot."""
    return float(np.hypot(re, im))


# ---- sorting criteria: src/sys/slepcsc.c:173-223 -----------------------
def cmp_largest_magnitude(ar, ai, br, bi):
    a, b = abs_eig(ar, ai), abs_eig(br, bi)
    return 1 if a < b else (-1 if a > b else 0)


def cmp_smallest_magnitude(ar, ai, br, bi):
    a, b = abs_eig(ar, ai), abs_eig(br, bi)
    return 1 if a > b else (-1 if a < b else 0)


def cmp_largest_real(ar, ai, br, bi):
    return 1 if ar < br else (-1 if ar > br else 0)


def cmp_smallest_real(ar, ai, br, bi):
    return 1 if ar > br else (-1 if ar < br else 0)


def cmp_largest_imaginary(ar, ai, br, bi):
    a, b = abs(ai), abs(bi)
    return 1 if a < b else (-1 if a > b else 0)


def cmp_smallest_imaginary(ar, ai, br, bi):
    a, b = abs(ai), abs(bi)
    return 1 if a > b else (-1 if a < b else 0)


COMPARE = {
    "largest_magnitude": cmp_largest_magnitude,
    "smallest_magnitude": cmp_smallest_magnitude,
    "largest_real": cmp_largest_real,
    "smallest_real": cmp_smallest_real,
    "largest_imaginary": cmp_largest_imaginary,
    "smallest_imaginary": cmp_smallest_imaginary,
}


def sort_eigenvalues(compare, n, eigr, eigi, perm):
    """``SlepcSortEigenvalues`` (slepcsc.c:107-159): insertion sort of perm that
    keeps complex-conjugate pairs together."""
    i = n - 1
    while i >= 0:
        re = eigr[perm[i]]
        im = eigi[perm[i]]
        j = i + 1
        if im != 0:
            i -= 1
            im = eigi[perm[i]]
        while j < n:
            if compare(re, im, eigr[perm[j]], eigi[perm[j]]) < 0:
                break
            if not im:
                if eigi[perm[j]] == 0.0:
                    perm[j - 1], perm[j] = perm[j], perm[j - 1]
                    j += 1
                else:
                    tmp = perm[j - 1]
                    perm[j - 1] = perm[j]
                    perm[j] = perm[j + 1]
                    perm[j + 1] = tmp
                    j += 2
            else:
                if eigi[perm[j]] == 0.0:
                    tmp = perm[j - 2]
                    perm[j - 2] = perm[j]
                    perm[j] = perm[j - 1]
                    perm[j - 1] = tmp
                    j += 1
                else:
                    perm[j - 2], perm[j] = perm[j], perm[j - 2]
                    perm[j - 1], perm[j + 1] = perm[j + 1], perm[j - 1]
                    j += 2
        i -= 1
    return perm


class DS:
    """Common state of a DS object (``dsops.c:139-177``)."""

    def __init__(self, ld, compare):
        self.ld = ld
        self.n = self.l = self.k = self.t = 0
        self.state = STATE_RAW
        self.compare = compare
        self.perm = np.arange(ld)
        self.extrarow = False

    def set_dimensions(self, n, l, k):
        """``DSSetDimensions`` (dsops.c:139-177)."""
        self.n, self.t, self.l, self.k = n, n, l, k

    def set_state(self, state):
        self.state = state


class DSHEP(DS):
    """DSHEP with compact storage and extra row (what
    ``EPSSetUp_KrylovSchur`` configures, krylovschur.c:141-148).
    T[0:ld] = diagonal, T[ld:2ld] = arrow / off-diagonal entries."""

    def __init__(self, ld, compare):
        super().__init__(ld, compare)
        self.extrarow = True
        self.T = np.zeros(3 * ld)
        self.Q = np.zeros((ld, ld), order="F")

    @property
    def d(self):
        return self.T[:self.ld]

    @property
    def e(self):
        return self.T[self.ld:2 * self.ld]

    def _arrow_tridiag(self, n, d, e, Q, off):
        """``ArrowTridiag`` (dshep.c:254-301): Rutishauser reduction of the
        arrowhead block to tridiagonal, rotations accumulated in Q.  d, e are
        views starting at l; Q is the flat column-major array, ``off`` the
        offset of the (l,l) entry."""
        ld = self.ld
        if n <= 2:
            return
        Qf = Q.reshape(-1, order="F")

        def colview(c, length):
            s = off + c * ld
            return Qf[s:s + length]

        for j in range(n - 2):
            temp = e[j + 1]
            c, s, r = lapack.dlartg(temp, e[j])
            e[j + 1] = r
            s = -s
            temp = d[j + 1]
            e[j] = c * s * (temp - d[j])
            d[j + 1] = s * s * d[j] + c * c * temp
            d[j] = c * c * d[j] + s * s * temp
            j2 = j + 2
            lapack.drot(j2, colview(j, j2), colview(j + 1, j2), c, s)
            for i in range(j - 1, -1, -1):
                offv = -s * e[i]
                e[i] = c * e[i]
                temp = e[i + 1]
                c, s, r = lapack.dlartg(temp, offv)
                e[i + 1] = r
                s = -s
                temp = (d[i] - d[i + 1]) * s - 2.0 * c * e[i]
                p = s * temp
                d[i + 1] += p
                d[i] -= p
                e[i] = -e[i] - c * temp
                lapack.drot(j2, colview(i, j2), colview(i + 1, j2), c, s)

    def solve(self, wr):
        """``DSSolve`` (dsops.c:500-518) -> ``DSSolve_HEP_QR`` (dshep.c:420-471)
        with ``DSIntermediate_HEP`` (dshep.c:306-359), compact branch."""
        if self.state >= STATE_CONDENSED:
            return
        n, l, ld = self.n, self.l, self.ld
        n1 = self.k - l + 1
        n2 = n - self.k - 1
        n3 = n1 + n2
        off = l + l * ld
        d, e = self.d, self.e
        self.Q[...] = 0.0
        for i in range(n):
            self.Q[i, i] = 1.0
        Qf = self.Q.reshape(-1, order="F")
        assert np.shares_memory(Qf, self.Q)
        if self.state < STATE_INTERMEDIATE:
            self._arrow_tridiag(n1, d[l:], e[l:], self.Q, off)
        for i in range(l):
            wr[i] = d[i]
        info = lapack.dsteqr("V", n3, d[l:], e[l:], Qf[off:], ld)
        if info:
            raise ArithmeticError("Error in Lapack xSTEQR %d" % info)
        for i in range(l, n):
            wr[i] = d[i]
        e[:max(n - 1, 0)] = 0.0
        self.state = STATE_CONDENSED

    def sort(self, wr):
        """``DSSort`` (dsops.c:553-576) -> ``DSSort_HEP`` (dshep.c:361-387) with
        ``DSSortEigenvaluesReal_Private`` (dspriv.c:242-264) and
        ``DSPermuteColumns_Private`` (dspriv.c:288-309)."""
        n, l = self.n, self.l
        d = self.d
        perm = self.perm
        for i in range(n):
            perm[i] = i
        nn = self.t
        for i in range(l + 1, nn):
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
        Q = self.Q
        for i in range(l, n):
            p = perm[i]
            if p != i:
                j = i + 1
                while perm[j] != i:
                    j += 1
                perm[j] = p
                perm[i] = i
                tmp = Q[:n, p].copy()
                Q[:n, p] = Q[:n, i]
                Q[:n, i] = tmp
        for i in range(l, n):
            d[i] = wr[i]

    def update_extra_row(self):
        """``DSUpdateExtraRow_HEP`` (dshep.c:389-418), compact branch."""
        n = self.n
        e = self.e
        beta = e[n - 1]
        for i in range(n):
            e[i] = beta * self.Q[n - 1, i]
        self.k = n

    def vectors_rnorm(self, j):
        """``DSVectors_HEP`` (dshep.c:170-205): residual estimate of Ritz pair j,
        abs(Q[n-1,j]); returns (newj, rnorm)."""
        return j, abs(self.Q[self.n - 1, j])

    def truncate(self, n):
        """``DSTruncate`` (dsops.c:232-251) -> ``DSTruncate_HEP`` (dshep.c:689-704)."""
        if self.state == STATE_CONDENSED:
            self.t = self.n
        self.k = n if self.extrarow else 0
        self.n = n
        self.state = STATE_TRUNCATED

    def get_mat_Q(self):
        """``DSGetMat`` (dsops.c:280-315): rows = max(n,t), cols = n."""
        rows = max(self.n, self.t)
        return np.array(self.Q[:rows, :self.n], order="F", copy=True)


class DSNHEP(DS):
    """DSNHEP (``dsnhep.c``): A holds H / the real Schur form, Q the Schur vectors."""

    def __init__(self, ld, compare):
        super().__init__(ld, compare)
        self.A = np.zeros((ld, ld), order="F")
        self.Q = np.zeros((ld, ld), order="F")
        self.X = np.zeros((ld, ld), order="F")

    def solve(self, wr, wi):
        """``DSSolve_NHEP`` (dsnhep.c:469-543)."""
        if self.state >= STATE_CONDENSED:
            return
        n, ld = self.n, self.ld
        ilo = self.l + 1
        A, Q = self.A, self.Q
        Q[...] = 0.0
        for i in range(n):
            Q[i, i] = 1.0
        if n == 1:
            wr[0] = A[0, 0]
            wi[0] = 0.0
            self.state = STATE_CONDENSED
            return
        Af = A.reshape(-1, order="F")
        Qf = Q.reshape(-1, order="F")
        if self.state < STATE_INTERMEDIATE:
            tau, info = lapack.dgehrd(n, ilo, n, Af, ld)
            if info:
                raise ArithmeticError("Error in Lapack xGEHRD %d" % info)
            for j in range(n - 1):
                for i in range(j + 2, n):
                    Q[i, j] = A[i, j]
                    A[i, j] = 0.0
            info = lapack.dorghr(n, ilo, n, Qf, ld, tau)
            if info:
                raise ArithmeticError("Error in Lapack xORGHR %d" % info)
        info = lapack.dhseqr("S", "V", n, ilo, n, Af, ld, wr, wi, Qf, ld)
        j = 0
        while j < self.l:
            if j == n - 1 or A[j + 1, j] == 0.0:
                wr[j] = A[j, j]
                wi[j] = 0.0
            else:
                wr[j] = A[j, j]
                wr[j + 1] = A[j, j]
                wi[j] = np.sqrt(abs(A[j + 1, j])) * np.sqrt(abs(A[j, j + 1]))
                wi[j + 1] = -wi[j]
                j += 1
            j += 1
        if info:
            raise ArithmeticError("Error in Lapack xHSEQR %d" % info)
        self.state = STATE_CONDENSED

    def update_extra_row(self):
        """``DSUpdateExtraRow_NHEP`` (dsnhep.c:434-456): row n of A times Q."""
        n = self.n
        x = self.A[n, :n].copy()
        self.A[n, :n] = self.Q[:n, :n].T @ x
        self.k = n

    def sort(self, wr, wi):
        """``DSSort_NHEP_Total`` (dsnhep.c:347-432): selection sort with trexc."""
        n, ld = self.n, self.ld
        T, Q = self.A, self.Q
        Tf = T.reshape(-1, order="F")
        Qf = Q.reshape(-1, order="F")
        for i in range(n):
            self.perm[i] = i
        i = self.l
        while i < n - 1:
            re, im = wr[i], wi[i]
            pos = 0
            j = i + 1
            if im != 0:
                j = i + 2
            while j < n:
                if self.compare(re, im, wr[j], wi[j]) > 0:
                    re, im = wr[j], wi[j]
                    pos = j
                if wi[j] != 0:
                    j += 1
                j += 1
            if pos:
                info = lapack.dtrexc("V", n, Tf, ld, Qf, ld, pos + 1, i + 1)
                if info:
                    raise ArithmeticError("Error in Lapack xTREXC %d" % info)
                j = i
                while j < n:
                    wr[j] = T[j, j]
                    if j < n - 1 and T[j + 1, j] != 0.0:
                        wi[j] = np.sqrt(abs(T[j + 1, j])) * np.sqrt(abs(T[j, j + 1]))
                        wr[j + 1] = wr[j]
                        wi[j + 1] = -wi[j]
                        j += 1
                    else:
                        wi[j] = 0.0
                    j += 1
            if wi[i] != 0:
                i += 1
            i += 1

    def vectors_rnorm(self, k):
        """``DSVectors_NHEP_Eigen_Some`` (dsnhep.c:123-192): k-th right eigenvector
        in X(:,k) [and X(:,k+1)], returns (newk, abs(last component))."""
        n, ld = self.n, self.ld
        A, Q, X = self.A, self.Q, self.X
        select = np.zeros(n, dtype=np.int32)
        select[k] = 1
        iscomplex = k < n - 1 and A[k + 1, k] != 0.0
        mm = 2 if iscomplex else 1
        if iscomplex:
            select[k + 1] = 1
        Af = A.reshape(-1, order="F")
        Xf = X.reshape(-1, order="F")
        Y = Xf[k * ld:]
        mout, info = lapack.dtrevc("R", "S", select, n, Af, ld, Y, ld, Y, ld, mm)
        if info:
            raise ArithmeticError("Error in Lapack xTREVC %d" % info)
        if mout != mm:
            raise ArithmeticError("Inconsistent arguments")
        if self.state >= STATE_CONDENSED:
            w0 = X[:n, k].copy()
            X[:n, k] = Q[:n, :n] @ w0
            if iscomplex:
                w1 = X[:n, k + 1].copy()
                X[:n, k + 1] = Q[:n, :n] @ w1
            norm = float(np.linalg.norm(X[:n, k]))
            if iscomplex:
                norm = abs_eig(norm, float(np.linalg.norm(X[:n, k + 1])))
            X[:n, k] *= 1.0 / norm
            if iscomplex:
                X[:n, k + 1] *= 1.0 / norm
        if iscomplex:
            return k + 1, abs_eig(X[n - 1, k], X[n - 1, k + 1])
        return k, abs(X[n - 1, k])

    def vectors_all(self):
        """``DSVectors_NHEP_Eigen_All`` (dsnhep.c:194-254): all right eigenvectors
        into X, back-transformed with Q when the DS has been solved."""
        n, ld = self.n, self.ld
        A, X = self.A, self.X
        Af = A.reshape(-1, order="F")
        Xf = X.reshape(-1, order="F")
        if self.state >= STATE_CONDENSED:
            back = "B"
            X[...] = self.Q
        else:
            back = "A"
        mout, info = lapack.dtrevc("R", back, None, n, Af, ld, Xf, ld, Xf, ld, n)
        if info:
            raise ArithmeticError("Error in Lapack xTREVC %d" % info)
        i = 0
        while i < n:
            iscomplex = i < n - 1 and A[i + 1, i] != 0.0
            norm = float(np.linalg.norm(X[:n, i]))
            if iscomplex:
                norm = abs_eig(norm, float(np.linalg.norm(X[:n, i + 1])))
            X[:n, i] *= 1.0 / norm
            if iscomplex:
                X[:n, i + 1] *= 1.0 / norm
                i += 1
            i += 1
        if self.state < STATE_CONDENSED:
            self.state = STATE_CONDENSED

    def truncate(self, n):
        """``DSTruncate_NHEP`` (dsnhep.c:545-567)."""
        if self.state == STATE_CONDENSED:
            self.t = self.n
        A = self.A
        if A[n, n - 1] == 0.0:
            newn = n
        else:
            newn = n + 1 if n < self.n - 1 else n - 1
        if self.extrarow and self.k == self.n:
            for i in range(self.l, newn):
                A[newn, i] = A[self.n, i]
            for i in range(self.l, self.n):
                A[self.n, i] = 0.0
        self.k = 0
        self.n = newn
        self.state = STATE_TRUNCATED

    def get_mat(self, which="Q"):
        """``DSGetMat`` (dsops.c:280-315)."""
        rows = max(self.n, self.t)
        M = self.Q if which == "Q" else self.X
        return np.array(M[:rows, :self.n], order="F", copy=True)
