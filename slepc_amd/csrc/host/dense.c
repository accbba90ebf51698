/*
 * Small dense kernels for the projected eigenproblem (order <= ncv+1), written
 * without LAPACK so that the host solver has no external dependency.  Each
 * routine provides what the corresponding LAPACK routine provides to the
 * reference's DS class (src/sys/classes/ds/impls/hep/dshep.c and
 * impls/nhep/dsnhep.c): lartg, rot, steqr, gehrd+orghr, hseqr, trexc, trevc.
 * All matrices are column-major doubles.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dense.h"

#define A_(i, j) A[(size_t)(i) + (size_t)(j) * lda]
#define Z_(i, j) Z[(size_t)(i) + (size_t)(j) * ldz]
#define T_(i, j) T[(size_t)(i) + (size_t)(j) * ldt]
#define Q_(i, j) Q[(size_t)(i) + (size_t)(j) * ldq]

static double sign_of(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

/* Plane rotation: [c s; -s c] [f; g] = [r; 0]. */
void ksd_lartg(double f, double g, double *c, double *s, double *r) {
  if (g == 0.0) { *c = 1.0; *s = 0.0; *r = f; return; }
  if (f == 0.0) { *c = 0.0; *s = 1.0; *r = g; return; }
  double rr = hypot(f, g);
  double cc = f / rr, ss = g / rr;
  if (fabs(f) > fabs(g) && cc < 0.0) { cc = -cc; ss = -ss; rr = -rr; }
  *c = cc; *s = ss; *r = rr;
}

/* x <- c*x + s*y ; y <- c*y - s*x  on n contiguous entries. */
void ksd_rot(int n, double *x, double *y, double c, double s) {
  for (int i = 0; i < n; i++) {
    const double t = c * x[i] + s * y[i];
    y[i] = c * y[i] - s * x[i];
    x[i] = t;
  }
}

/* Symmetric tridiagonal eigenproblem by implicit QL with Wilkinson shifts.
 * d[n], e[n-1] (e destroyed); Z (n rows used, n columns) is multiplied from the
 * right by the rotations, so on entry it holds the basis in which the
 * tridiagonal matrix is expressed.  Eigenvalues return in d (unsorted).
 * Returns 0, or i>0 if eigenvalue i failed to converge. */
int ksd_steqr(int n, double *d, double *e, double *Z, int ldz, int nrows) {
  if (n <= 1) return 0;
  double *ee = (double *)malloc(sizeof(double) * (size_t)n);
  if (!ee) return -1;
  for (int i = 0; i < n - 1; i++) ee[i] = e[i];
  ee[n - 1] = 0.0;
  for (int l = 0; l < n; l++) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; m++) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(ee[m]) <= DBL_EPSILON * dd) break;
      }
      if (m != l) {
        if (iter++ == 60) { free(ee); return l + 1; }
        double g = (d[l + 1] - d[l]) / (2.0 * ee[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + ee[l] / (g + sign_of(r, g));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; i--) {
          double f = s * ee[i];
          const double b = c * ee[i];
          r = hypot(f, g);
          ee[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= p; ee[m] = 0.0; break; }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          for (int k = 0; k < nrows; k++) {
            f = Z_(k, i + 1);
            Z_(k, i + 1) = s * Z_(k, i) + c * f;
            Z_(k, i) = c * Z_(k, i) - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        ee[l] = g;
        ee[m] = 0.0;
      }
    } while (m != l);
  }
  for (int i = 0; i < n - 1; i++) e[i] = 0.0;
  free(ee);
  return 0;
}

/* Householder vector for x (length m): H = I - tau v v^T with v[0] = 1 maps x to
 * beta e_1.  v overwrites x[1..], returns beta, tau. */
static void house(int m, double *x, double *beta, double *tau) {
  if (m <= 1) { *tau = 0.0; *beta = m ? x[0] : 0.0; return; }
  double xnorm = 0.0;
  for (int i = 1; i < m; i++) xnorm = hypot(xnorm, x[i]);
  if (xnorm == 0.0) { *tau = 0.0; *beta = x[0]; return; }
  const double alpha = x[0];
  const double b = -sign_of(hypot(alpha, xnorm), alpha);
  *tau = (b - alpha) / b;
  const double sc = 1.0 / (alpha - b);
  for (int i = 1; i < m; i++) x[i] *= sc;
  *beta = b;
}

/* Reduction to upper Hessenberg form of the trailing block that starts at row and
 * column ilo (0-based), with the orthogonal factor formed explicitly:
 * on exit A := Q^T A Q is Hessenberg in rows/columns ilo..n-1 and Q (n x n) holds
 * the product of the reflectors (identity outside the block).  The first ilo
 * columns of A are assumed upper triangular and are not touched. */
int ksd_gehrd_orghr(int n, int ilo, double *A, int lda, double *Q, int ldq) {
  for (int j = 0; j < n; j++) {
    for (int i = 0; i < n; i++) Q_(i, j) = 0.0;
    Q_(j, j) = 1.0;
  }
  if (n - ilo < 3) return 0;
  double *v = (double *)malloc(sizeof(double) * (size_t)n);
  if (!v) return -1;
  for (int c = ilo; c < n - 2; c++) {
    const int m = n - c - 1;                 /* reflector acts on rows c+1..n-1 */
    double beta, tau;
    for (int i = 0; i < m; i++) v[i] = A_(c + 1 + i, c);
    house(m, v, &beta, &tau);
    v[0] = 1.0;
    A_(c + 1, c) = beta;
    for (int i = c + 2; i < n; i++) A_(i, c) = 0.0;
    if (tau == 0.0) continue;
    /* A <- H A on rows c+1.., columns c+1.. */
    for (int j = c + 1; j < n; j++) {
      double s = 0.0;
      for (int i = 0; i < m; i++) s += v[i] * A_(c + 1 + i, j);
      s *= tau;
      for (int i = 0; i < m; i++) A_(c + 1 + i, j) -= s * v[i];
    }
    /* A <- A H on all rows, columns c+1.. */
    for (int i = 0; i < n; i++) {
      double s = 0.0;
      for (int j = 0; j < m; j++) s += A_(i, c + 1 + j) * v[j];
      s *= tau;
      for (int j = 0; j < m; j++) A_(i, c + 1 + j) -= s * v[j];
    }
    /* Q <- Q H */
    for (int i = 0; i < n; i++) {
      double s = 0.0;
      for (int j = 0; j < m; j++) s += Q_(i, c + 1 + j) * v[j];
      s *= tau;
      for (int j = 0; j < m; j++) Q_(i, c + 1 + j) -= s * v[j];
    }
  }
  free(v);
  return 0;
}

/* Schur factorisation of a real 2x2 matrix [a b; c d] in standard form: returns
 * the rotation (cs, sn) and the new entries such that either c = 0 (two real
 * eigenvalues) or a = d and b*c < 0 (complex conjugate pair). */
void ksd_lanv2(double *a, double *b, double *c, double *d, double *rt1r, double *rt1i,
               double *rt2r, double *rt2i, double *cs, double *sn) {
  const double multpl = 4.0;
  const double eps = DBL_EPSILON;
  if (*c == 0.0) {
    *cs = 1.0; *sn = 0.0;
  } else if (*b == 0.0) {
    /* swap rows and columns */
    *cs = 0.0; *sn = 1.0;
    const double temp = *d;
    *d = *a; *a = temp;
    *b = -*c; *c = 0.0;
  } else if ((*a - *d) == 0.0 && sign_of(1.0, *b) != sign_of(1.0, *c)) {
    *cs = 1.0; *sn = 0.0;
  } else {
    const double temp = *a - *d;
    double p = 0.5 * temp;
    const double bcmax = fmax(fabs(*b), fabs(*c));
    const double bcmis = fmin(fabs(*b), fabs(*c)) * sign_of(1.0, *b) * sign_of(1.0, *c);
    double scale = fmax(fabs(p), bcmax);
    double z = (p / scale) * p + (bcmax / scale) * bcmis;
    if (z >= multpl * eps) {
      /* real eigenvalues: make the block upper triangular */
      z = p + sign_of(sqrt(scale) * sqrt(z), p);
      *a = *d + z;
      *d = *d - (bcmax / z) * bcmis;
      const double tau = hypot(*c, z);
      *cs = z / tau;
      *sn = *c / tau;
      *b = *b - *c;
      *c = 0.0;
    } else {
      /* complex or almost equal real eigenvalues: make the diagonal equal */
      const double sigma = *b + *c;
      const double tau = hypot(sigma, temp);
      *cs = sqrt(0.5 * (1.0 + fabs(sigma) / tau));
      *sn = -(p / (tau * (*cs))) * sign_of(1.0, sigma);
      const double aa = *a * (*cs) + *b * (*sn);
      const double bb = -*a * (*sn) + *b * (*cs);
      const double cc = *c * (*cs) + *d * (*sn);
      const double dd = -*c * (*sn) + *d * (*cs);
      *a = aa * (*cs) + cc * (*sn);
      *b = bb * (*cs) + dd * (*sn);
      *c = -aa * (*sn) + cc * (*cs);
      *d = -bb * (*sn) + dd * (*cs);
      const double tmp = 0.5 * (*a + *d);
      *a = tmp; *d = tmp;
      if (*c != 0.0) {
        if (*b != 0.0) {
          if (sign_of(1.0, *b) == sign_of(1.0, *c)) {
            /* real eigenvalues after all: reduce to upper triangular */
            const double sab = sqrt(fabs(*b)), sac = sqrt(fabs(*c));
            p = sign_of(sab * sac, *c);
            const double tau2 = 1.0 / sqrt(fabs(*b + *c));
            *a = tmp + p;
            *d = tmp - p;
            *b = *b - *c;
            *c = 0.0;
            const double cs1 = sab * tau2, sn1 = sac * tau2;
            const double t = (*cs) * cs1 - (*sn) * sn1;
            *sn = (*cs) * sn1 + (*sn) * cs1;
            *cs = t;
          }
        } else {
          *b = -*c; *c = 0.0;
          const double t = *cs;
          *cs = -*sn; *sn = t;
        }
      }
    }
  }
  *rt1r = *a; *rt2r = *d;
  if (*c == 0.0) { *rt1i = 0.0; *rt2i = 0.0; }
  else { *rt1i = sqrt(fabs(*b)) * sqrt(fabs(*c)); *rt2i = -*rt1i; }
}

/* Apply the rotation of a standardised 2x2 block at (k,k) to the rest of T and to Z:
 * rows k,k+1 for columns > k+1, columns k,k+1 for rows < k, and Z columns. */
static void apply_block_rot(int n, double *T, int ldt, double *Z, int ldz, int nz, int k, double cs, double sn) {
  for (int j = k + 2; j < n; j++) {
    const double t1 = T_(k, j), t2 = T_(k + 1, j);
    T_(k, j) = cs * t1 + sn * t2;
    T_(k + 1, j) = cs * t2 - sn * t1;
  }
  for (int i = 0; i < k; i++) {
    const double t1 = T_(i, k), t2 = T_(i, k + 1);
    T_(i, k) = cs * t1 + sn * t2;
    T_(i, k + 1) = cs * t2 - sn * t1;
  }
  if (Z)
    for (int i = 0; i < nz; i++) {
      const double t1 = Z_(i, k), t2 = Z_(i, k + 1);
      Z_(i, k) = cs * t1 + sn * t2;
      Z_(i, k + 1) = cs * t2 - sn * t1;
    }
}

/* Real Schur form of an upper Hessenberg matrix by the Francis double-shift QR
 * iteration: H := Z_new^T H Z_new quasi-triangular with standardised 2x2 blocks,
 * Z := Z * Z_new (Z must hold an orthogonal matrix on entry; n rows).  Only the
 * block ilo..n-1 (0-based) is iterated on; rows above it and the columns are
 * updated so that the whole n x n matrix stays similar.  Returns 0 or the
 * (1-based) index of an eigenvalue that did not converge. */
int ksd_hseqr(int n, int ilo, double *T, int ldt, double *wr, double *wi, double *Z, int ldz) {
  const double smlnum = DBL_MIN * ((double)n / DBL_EPSILON);
  const int itmax = 30 * (n > 10 ? n : 10);
  for (int i = 0; i < ilo; i++) { wr[i] = T_(i, i); wi[i] = 0.0; }
  if (n - ilo == 1) { wr[ilo] = T_(ilo, ilo); wi[ilo] = 0.0; return 0; }
  for (int j = ilo; j < n - 2; j++) { if (j + 2 < n) T_(j + 2, j) = 0.0; if (j + 3 < n) T_(j + 3, j) = 0.0; }
  int i = n - 1;
  while (i >= ilo) {
    int l = ilo, its;
    int converged = 0;
    for (its = 0; its <= itmax; its++) {
      /* look for a negligible subdiagonal element */
      int k;
      for (k = i; k > l; k--) {
        if (fabs(T_(k, k - 1)) <= smlnum) break;
        double tst = fabs(T_(k - 1, k - 1)) + fabs(T_(k, k));
        if (tst == 0.0) {
          if (k - 2 >= ilo) tst += fabs(T_(k - 1, k - 2));
          if (k + 1 < n) tst += fabs(T_(k + 1, k));
        }
        if (fabs(T_(k, k - 1)) <= DBL_EPSILON * tst) {
          const double ab = fmax(fabs(T_(k, k - 1)), fabs(T_(k - 1, k)));
          const double ba = fmin(fabs(T_(k, k - 1)), fabs(T_(k - 1, k)));
          const double aa = fmax(fabs(T_(k, k)), fabs(T_(k - 1, k - 1) - T_(k, k)));
          const double bb = fmin(fabs(T_(k, k)), fabs(T_(k - 1, k - 1) - T_(k, k)));
          const double s = aa + ab;
          if (ba * (ab / s) <= fmax(smlnum, DBL_EPSILON * (bb * (aa / s)))) break;
        }
      }
      l = k;
      if (l > ilo) T_(l, l - 1) = 0.0;
      if (l >= i - 1) { converged = 1; break; }

      double h11, h21, h12, h22;
      if (its == 10) {
        const double s = fabs(T_(l + 1, l)) + fabs(T_(l + 2, l + 1));
        h11 = 0.75 * s + T_(l, l); h12 = -0.4375 * s; h21 = s; h22 = h11;
      } else if (its == 20) {
        const double s = fabs(T_(i, i - 1)) + fabs(T_(i - 1, i - 2));
        h11 = 0.75 * s + T_(i, i); h12 = -0.4375 * s; h21 = s; h22 = h11;
      } else {
        h11 = T_(i - 1, i - 1); h21 = T_(i, i - 1); h12 = T_(i - 1, i); h22 = T_(i, i);
      }
      double rt1r, rt1i, rt2r, rt2i;
      const double s = fabs(h11) + fabs(h12) + fabs(h21) + fabs(h22);
      if (s == 0.0) { rt1r = rt1i = rt2r = rt2i = 0.0; }
      else {
        h11 /= s; h21 /= s; h12 /= s; h22 /= s;
        const double tr = 0.5 * (h11 + h22);
        const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
        const double rtdisc = sqrt(fabs(det));
        if (det >= 0.0) { rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i; }
        else {
          rt1r = tr + rtdisc; rt2r = tr - rtdisc;
          if (fabs(rt1r - h22) <= fabs(rt2r - h22)) { rt1r *= s; rt2r = rt1r; }
          else { rt2r *= s; rt1r = rt2r; }
          rt1i = rt2i = 0.0;
        }
      }
      /* look for two consecutive small subdiagonal elements */
      int m;
      double v[3];
      for (m = i - 2; m >= l; m--) {
        const double h21s0 = T_(m + 1, m);
        double ss = fabs(T_(m, m) - rt2r) + fabs(rt2i) + fabs(h21s0);
        const double h21s = T_(m + 1, m) / ss;
        v[0] = h21s * T_(m, m + 1) + (T_(m, m) - rt1r) * ((T_(m, m) - rt2r) / ss) - rt1i * (rt2i / ss);
        v[1] = h21s * (T_(m, m) + T_(m + 1, m + 1) - rt1r - rt2r);
        v[2] = h21s * T_(m + 2, m + 1);
        ss = fabs(v[0]) + fabs(v[1]) + fabs(v[2]);
        v[0] /= ss; v[1] /= ss; v[2] /= ss;
        if (m == l) break;
        const double h00 = fabs(T_(m - 1, m - 1)), h11a = fabs(T_(m, m)), h22a = fabs(T_(m + 1, m + 1));
        if (fabs(T_(m, m - 1)) * (fabs(v[1]) + fabs(v[2])) <= DBL_EPSILON * fabs(v[0]) * (h00 + h11a + h22a)) break;
      }
      /* double-shift QR step */
      for (int k2 = m; k2 <= i - 1; k2++) {
        const int nr = (3 < i - k2 + 1) ? 3 : i - k2 + 1;
        if (k2 > m) { v[0] = T_(k2, k2 - 1); v[1] = T_(k2 + 1, k2 - 1); if (nr == 3) v[2] = T_(k2 + 2, k2 - 1); }
        double beta, tau;
        house(nr, v, &beta, &tau);
        if (k2 > m) {
          T_(k2, k2 - 1) = beta;
          T_(k2 + 1, k2 - 1) = 0.0;
          if (k2 < i - 1) T_(k2 + 2, k2 - 1) = 0.0;
        } else if (m > l) {
          T_(k2, k2 - 1) *= (1.0 - tau);
        }
        const double v2 = v[1], t1 = tau, t2 = tau * v2;
        if (nr == 3) {
          const double v3 = v[2], t3 = tau * v3;
          for (int j = k2; j < n; j++) {
            const double sum = T_(k2, j) + v2 * T_(k2 + 1, j) + v3 * T_(k2 + 2, j);
            T_(k2, j) -= sum * t1; T_(k2 + 1, j) -= sum * t2; T_(k2 + 2, j) -= sum * t3;
          }
          const int jmax = (k2 + 3 < i) ? k2 + 3 : i;
          for (int j = 0; j <= jmax; j++) {
            const double sum = T_(j, k2) + v2 * T_(j, k2 + 1) + v3 * T_(j, k2 + 2);
            T_(j, k2) -= sum * t1; T_(j, k2 + 1) -= sum * t2; T_(j, k2 + 2) -= sum * t3;
          }
          if (Z) for (int j = 0; j < n; j++) {
            const double sum = Z_(j, k2) + v2 * Z_(j, k2 + 1) + v3 * Z_(j, k2 + 2);
            Z_(j, k2) -= sum * t1; Z_(j, k2 + 1) -= sum * t2; Z_(j, k2 + 2) -= sum * t3;
          }
        } else if (nr == 2) {
          for (int j = k2; j < n; j++) {
            const double sum = T_(k2, j) + v2 * T_(k2 + 1, j);
            T_(k2, j) -= sum * t1; T_(k2 + 1, j) -= sum * t2;
          }
          for (int j = 0; j <= i; j++) {
            const double sum = T_(j, k2) + v2 * T_(j, k2 + 1);
            T_(j, k2) -= sum * t1; T_(j, k2 + 1) -= sum * t2;
          }
          if (Z) for (int j = 0; j < n; j++) {
            const double sum = Z_(j, k2) + v2 * Z_(j, k2 + 1);
            Z_(j, k2) -= sum * t1; Z_(j, k2 + 1) -= sum * t2;
          }
        }
      }
    }
    if (!converged) return i + 1;
    if (l == i) {
      wr[i] = T_(i, i); wi[i] = 0.0;
    } else {
      double cs, sn;
      ksd_lanv2(&T_(i - 1, i - 1), &T_(i - 1, i), &T_(i, i - 1), &T_(i, i), &wr[i - 1], &wi[i - 1], &wr[i], &wi[i], &cs, &sn);
      apply_block_rot(n, T, ldt, Z, ldz, n, i - 1, cs, sn);
    }
    i = l - 1;
  }
  return 0;
}

/* ---- swapping adjacent diagonal blocks of a quasi-triangular matrix ------------ */

/* Solve the small dense system M x = b (order m <= 4) with complete pivoting. */
static int solve_small(int m, double *M, double *b) {
  int cp[4];
  for (int k = 0; k < m; k++) {
    int pi = k, pj = k;
    double big = 0.0;
    for (int i = k; i < m; i++)
      for (int j = k; j < m; j++)
        if (fabs(M[i + 4 * j]) > big) { big = fabs(M[i + 4 * j]); pi = i; pj = j; }
    if (big == 0.0) return 1;
    if (pi != k) { for (int j = 0; j < m; j++) { double t = M[k + 4 * j]; M[k + 4 * j] = M[pi + 4 * j]; M[pi + 4 * j] = t; } double t = b[k]; b[k] = b[pi]; b[pi] = t; }
    if (pj != k) for (int i = 0; i < m; i++) { double t = M[i + 4 * k]; M[i + 4 * k] = M[i + 4 * pj]; M[i + 4 * pj] = t; }
    cp[k] = pj;
    for (int i = k + 1; i < m; i++) {
      const double f = M[i + 4 * k] / M[k + 4 * k];
      for (int j = k + 1; j < m; j++) M[i + 4 * j] -= f * M[k + 4 * j];
      b[i] -= f * b[k];
    }
  }
  for (int k = m - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < m; j++) s -= M[k + 4 * j] * b[j];
    b[k] = s / M[k + 4 * k];
  }
  for (int k = m - 1; k >= 0; k--)
    if (cp[k] != k) { double t = b[k]; b[k] = b[cp[k]]; b[cp[k]] = t; }
  return 0;
}

/* Swap the adjacent diagonal blocks T11 (order n1, at j1) and T22 (order n2) of
 * the quasi-triangular T by an orthogonal similarity, updating Q (nq rows). */
static int laexc(int n, double *T, int ldt, double *Q, int ldq, int nq, int j1, int n1, int n2) {
  if (n1 == 0 || n2 == 0 || j1 + n1 >= n) return 0;
  if (n1 == 1 && n2 == 1) {
    const double t11 = T_(j1, j1), t22 = T_(j1 + 1, j1 + 1);
    double cs, sn, r;
    ksd_lartg(T_(j1, j1 + 1), t22 - t11, &cs, &sn, &r);
    for (int j = j1 + 2; j < n; j++) {
      const double a = T_(j1, j), b = T_(j1 + 1, j);
      T_(j1, j) = cs * a + sn * b; T_(j1 + 1, j) = cs * b - sn * a;
    }
    for (int i = 0; i < j1; i++) {
      const double a = T_(i, j1), b = T_(i, j1 + 1);
      T_(i, j1) = cs * a + sn * b; T_(i, j1 + 1) = cs * b - sn * a;
    }
    T_(j1, j1) = t22; T_(j1 + 1, j1 + 1) = t11;
    if (Q) for (int i = 0; i < nq; i++) {
      const double a = Q_(i, j1), b = Q_(i, j1 + 1);
      Q_(i, j1) = cs * a + sn * b; Q_(i, j1 + 1) = cs * b - sn * a;
    }
    return 0;
  }
  /* general case: nd = n1+n2 in {3,4}.  Solve T11 X - X T22 = T12, then the
   * orthogonal U whose first n2 columns span [ -X ; I ] swaps the blocks. */
  const int nd = n1 + n2;
  double D[16], X[4], M[16];
  for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) D[i + 4 * j] = T_(j1 + i, j1 + j);
  /* unknowns X(i,j), i<n1, j<n2, ordered column-major */
  const int nx = n1 * n2;
  memset(M, 0, sizeof(M));
  for (int j = 0; j < n2; j++)
    for (int i = 0; i < n1; i++) {
      const int row = i + n1 * j;
      for (int p = 0; p < n1; p++) M[row + 4 * (p + n1 * j)] += D[i + 4 * p];                 /* T11 X */
      for (int p = 0; p < n2; p++) M[row + 4 * (i + n1 * p)] -= D[(n1 + p) + 4 * (n1 + j)];   /* - X T22 */
      X[row] = D[i + 4 * (n1 + j)];
    }
  if (solve_small(nx, M, X)) return 1;
  /* W = [ -X ; I ] (nd x n2); Householder QR gives U (nd x nd) */
  double W[8], U[16];
  for (int j = 0; j < n2; j++) {
    for (int i = 0; i < n1; i++) W[i + 4 * j] = -X[i + n1 * j];
    for (int i = 0; i < n2; i++) W[n1 + i + 4 * j] = (i == j) ? 1.0 : 0.0;
  }
  for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) U[i + 4 * j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < n2; c++) {
    double v[4], beta, tau;
    const int m = nd - c;
    for (int i = 0; i < m; i++) v[i] = W[c + i + 4 * c];
    house(m, v, &beta, &tau);
    v[0] = 1.0;
    for (int j = c; j < n2; j++) {
      double s = 0.0;
      for (int i = 0; i < m; i++) s += v[i] * W[c + i + 4 * j];
      s *= tau;
      for (int i = 0; i < m; i++) W[c + i + 4 * j] -= s * v[i];
    }
    for (int i = 0; i < nd; i++) {
      double s = 0.0;
      for (int j = 0; j < m; j++) s += U[i + 4 * (c + j)] * v[j];
      s *= tau;
      for (int j = 0; j < m; j++) U[i + 4 * (c + j)] -= s * v[j];
    }
  }
  /* check: (U^T D U)(n2.., 0..n2-1) must be negligible */
  double Dn[16], tmp[16];
  for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) { double s = 0.0; for (int p = 0; p < nd; p++) s += D[i + 4 * p] * U[p + 4 * j]; tmp[i + 4 * j] = s; }
  for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) { double s = 0.0; for (int p = 0; p < nd; p++) s += U[p + 4 * i] * tmp[p + 4 * j]; Dn[i + 4 * j] = s; }
  double dnorm = 0.0, low = 0.0;
  for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) dnorm = fmax(dnorm, fabs(D[i + 4 * j]));
  for (int j = 0; j < n2; j++) for (int i = n2; i < nd; i++) low = fmax(low, fabs(Dn[i + 4 * j]));
  if (low > 10.0 * DBL_EPSILON * fmax(dnorm, DBL_MIN / DBL_EPSILON)) return 1;   /* blocks too close to swap */
  /* apply U to T (rows j1.., columns j1..) and Q */
  double col[4];
  for (int j = j1; j < n; j++) {               /* T <- U^T T on rows j1..j1+nd-1 */
    for (int i = 0; i < nd; i++) col[i] = T_(j1 + i, j);
    for (int i = 0; i < nd; i++) { double s = 0.0; for (int p = 0; p < nd; p++) s += U[p + 4 * i] * col[p]; T_(j1 + i, j) = s; }
  }
  for (int i = 0; i < j1 + nd; i++) {           /* T <- T U on columns j1..j1+nd-1 */
    for (int p = 0; p < nd; p++) col[p] = T_(i, j1 + p);
    for (int j = 0; j < nd; j++) { double s = 0.0; for (int p = 0; p < nd; p++) s += col[p] * U[p + 4 * j]; T_(i, j1 + j) = s; }
  }
  if (Q) for (int i = 0; i < nq; i++) {
    for (int p = 0; p < nd; p++) col[p] = Q_(i, j1 + p);
    for (int j = 0; j < nd; j++) { double s = 0.0; for (int p = 0; p < nd; p++) s += col[p] * U[p + 4 * j]; Q_(i, j1 + j) = s; }
  }
  for (int j = 0; j < n2; j++) for (int i = n2; i < nd; i++) T_(j1 + i, j1 + j) = 0.0;
  /* restore the standard form of the swapped 2x2 blocks */
  double wr1, wi1, wr2, wi2, cs, sn;
  if (n2 == 2) {
    ksd_lanv2(&T_(j1, j1), &T_(j1, j1 + 1), &T_(j1 + 1, j1), &T_(j1 + 1, j1 + 1), &wr1, &wi1, &wr2, &wi2, &cs, &sn);
    apply_block_rot(n, T, ldt, Q, ldq, nq, j1, cs, sn);
  }
  if (n1 == 2) {
    const int j3 = j1 + n2;
    ksd_lanv2(&T_(j3, j3), &T_(j3, j3 + 1), &T_(j3 + 1, j3), &T_(j3 + 1, j3 + 1), &wr1, &wi1, &wr2, &wi2, &cs, &sn);
    apply_block_rot(n, T, ldt, Q, ldq, nq, j3, cs, sn);
  }
  return 0;
}

/* Move the diagonal block that starts at row ifst to row ilst (0-based) by
 * swapping adjacent blocks, keeping T quasi-triangular and Q T Q^T invariant. */
int ksd_trexc(int n, double *T, int ldt, double *Q, int ldq, int nq, int ifst, int ilst) {
  if (n <= 1) return 0;
  if (ifst > 0 && T_(ifst, ifst - 1) != 0.0) ifst--;
  int nbf = 1;
  if (ifst < n - 1 && T_(ifst + 1, ifst) != 0.0) nbf = 2;
  if (ilst > 0 && T_(ilst, ilst - 1) != 0.0) ilst--;
  int nbl = 1;
  if (ilst < n - 1 && T_(ilst + 1, ilst) != 0.0) nbl = 2;
  if (ifst == ilst) return 0;
  if (ifst < ilst) {
    if (nbf == 2 && nbl == 1) ilst--;
    if (nbf == 1 && nbl == 2) ilst++;
    int here = ifst;
    while (here < ilst) {
      int nbnext = 1;
      if (here + nbf + 1 < n && T_(here + nbf + 1, here + nbf) != 0.0) nbnext = 2;
      if (laexc(n, T, ldt, Q, ldq, nq, here, nbf, nbnext)) return 1;
      here += nbnext;
      if (nbf == 2 && T_(here + 1, here) == 0.0) {
        /* the 2x2 block split into two real eigenvalues: carry only the first on */
        nbf = 1;
      }
    }
  } else {
    int here = ifst;
    while (here > ilst) {
      int nbnext = 1;
      if (here >= 2 && T_(here - 1, here - 2) != 0.0) nbnext = 2;
      if (laexc(n, T, ldt, Q, ldq, nq, here - nbnext, nbnext, nbf)) return 1;
      here -= nbnext;
      if (nbf == 2 && T_(here + 1, here) == 0.0) nbf = 1;
    }
  }
  return 0;
}

/* ---- eigenvectors of a quasi-triangular matrix ------------------------------------ */

/* Right eigenvector(s) of the upper quasi-triangular T for the diagonal block that
 * contains row/column k.  For a real eigenvalue x (n entries) is returned in xr;
 * for a complex pair at (k,k+1) the vector for the eigenvalue with positive
 * imaginary part is returned as xr + i*xi.  Entries below the block are zero.
 * The vectors are scaled so that the largest component has magnitude one.
 * Returns 1 if the block is 2x2 (complex), else 0. */
int ksd_trevc_right(int n, const double *T, int ldt, int k, double *xr, double *xi) {
  double tnorm = 0.0;
  for (int j = 0; j < n; j++) { double s = 0.0; for (int i = 0; i <= (j + 1 < n ? j + 1 : j); i++) s += fabs(T_(i, j)); tnorm = fmax(tnorm, s); }
  const double smin_base = fmax(DBL_EPSILON * tnorm, DBL_MIN / DBL_EPSILON);
  if (k > 0 && T_(k, k - 1) != 0.0) k--;
  const int cplx = (k < n - 1 && T_(k + 1, k) != 0.0);
  for (int i = 0; i < n; i++) { xr[i] = 0.0; if (xi) xi[i] = 0.0; }
  if (!cplx) {
    const double wr = T_(k, k);
    const double smin = fmax(DBL_EPSILON * fabs(wr), smin_base);
    xr[k] = 1.0;
    for (int i = 0; i < k; i++) xr[i] = -T_(i, k);
    int j = k - 1;
    while (j >= 0) {
      if (j > 0 && T_(j, j - 1) != 0.0) {
        /* 2x2 diagonal block at (j-1, j) */
        double a = T_(j - 1, j - 1) - wr, b = T_(j - 1, j), c = T_(j, j - 1), d = T_(j, j) - wr;
        double det = a * d - b * c;
        if (fabs(det) < smin * smin) det = smin * smin;
        const double r1 = xr[j - 1], r2 = xr[j];
        const double x1 = (d * r1 - b * r2) / det, x2 = (a * r2 - c * r1) / det;
        xr[j - 1] = x1; xr[j] = x2;
        for (int i = 0; i < j - 1; i++) xr[i] -= x1 * T_(i, j - 1) + x2 * T_(i, j);
        j -= 2;
      } else {
        double d = T_(j, j) - wr;
        if (fabs(d) < smin) d = smin;
        const double x1 = xr[j] / d;
        xr[j] = x1;
        for (int i = 0; i < j; i++) xr[i] -= x1 * T_(i, j);
        j -= 1;
      }
      /* rescale to keep the growth bounded */
      double big = 0.0;
      for (int i = 0; i <= k; i++) big = fmax(big, fabs(xr[i]));
      if (big > 1e150) for (int i = 0; i <= k; i++) xr[i] /= big;
    }
    double emax = 0.0;
    for (int i = 0; i <= k; i++) emax = fmax(emax, fabs(xr[i]));
    if (emax > 0.0) for (int i = 0; i <= k; i++) xr[i] /= emax;
    return 0;
  }
  /* complex pair: block [a b; c a] with b*c < 0, eigenvalue wr + i*wi, wi > 0 */
  const double wr = T_(k, k);
  const double wi = sqrt(fabs(T_(k, k + 1))) * sqrt(fabs(T_(k + 1, k)));
  const double smin = fmax(DBL_EPSILON * (fabs(wr) + fabs(wi)), smin_base);
  if (fabs(T_(k, k + 1)) >= fabs(T_(k + 1, k))) { xr[k] = 1.0; xi[k + 1] = wi / T_(k, k + 1); }
  else { xr[k] = -wi / T_(k + 1, k); xi[k + 1] = 1.0; }
  xr[k + 1] = 0.0; xi[k] = 0.0;
  for (int i = 0; i < k; i++) { xr[i] = -xr[k] * T_(i, k); xi[i] = -xi[k + 1] * T_(i, k + 1); }
  int j = k - 1;
  while (j >= 0) {
    if (j > 0 && T_(j, j - 1) != 0.0) {
      /* solve the 2x2 complex system [(a - w) b; c (d - w)] x = rhs, w = wr + i wi */
      const double a_r = T_(j - 1, j - 1) - wr, d_r = T_(j, j) - wr, b = T_(j - 1, j), c = T_(j, j - 1);
      /* det = (a_r - i wi)(d_r - i wi) - b c */
      double det_r = a_r * d_r - wi * wi - b * c, det_i = -wi * (a_r + d_r);
      double dm = det_r * det_r + det_i * det_i;
      if (dm < smin * smin * smin * smin) { det_r = smin * smin; det_i = 0.0; dm = det_r * det_r; }
      const double r1r = xr[j - 1], r1i = xi[j - 1], r2r = xr[j], r2i = xi[j];
      /* x1 = ((d - w) r1 - b r2)/det ; x2 = ((a - w) r2 - c r1)/det */
      const double n1r = d_r * r1r + wi * r1i - b * r2r, n1i = d_r * r1i - wi * r1r - b * r2i;
      const double n2r = a_r * r2r + wi * r2i - c * r1r, n2i = a_r * r2i - wi * r2r - c * r1i;
      const double x1r = (n1r * det_r + n1i * det_i) / dm, x1i = (n1i * det_r - n1r * det_i) / dm;
      const double x2r = (n2r * det_r + n2i * det_i) / dm, x2i = (n2i * det_r - n2r * det_i) / dm;
      xr[j - 1] = x1r; xi[j - 1] = x1i; xr[j] = x2r; xi[j] = x2i;
      for (int i = 0; i < j - 1; i++) {
        xr[i] -= x1r * T_(i, j - 1) + x2r * T_(i, j);
        xi[i] -= x1i * T_(i, j - 1) + x2i * T_(i, j);
      }
      j -= 2;
    } else {
      double dr = T_(j, j) - wr, di = -wi;
      double dm = dr * dr + di * di;
      if (dm < smin * smin) { dr = smin; di = 0.0; dm = dr * dr; }
      const double rr = xr[j], ri = xi[j];
      const double x1r = (rr * dr + ri * di) / dm, x1i = (ri * dr - rr * di) / dm;
      xr[j] = x1r; xi[j] = x1i;
      for (int i = 0; i < j; i++) { xr[i] -= x1r * T_(i, j); xi[i] -= x1i * T_(i, j); }
      j -= 1;
    }
    double big = 0.0;
    for (int i = 0; i <= k + 1; i++) big = fmax(big, fabs(xr[i]) + fabs(xi[i]));
    if (big > 1e150) for (int i = 0; i <= k + 1; i++) { xr[i] /= big; xi[i] /= big; }
  }
  double emax = 0.0;
  for (int i = 0; i <= k + 1; i++) emax = fmax(emax, fabs(xr[i]) + fabs(xi[i]));
  if (emax > 0.0) for (int i = 0; i <= k + 1; i++) { xr[i] /= emax; xi[i] /= emax; }
  return 1;
}

/* ---- singular value decomposition -------------------------------------------------------
 * One-sided Jacobi (Hestenes): plane rotations from the right make the columns of G = A V
 * mutually orthogonal; then s_j = ||g_j||, u_j = g_j / s_j.  Takes the place of LAPACK's
 * gesdd/bdsdc in the projected problem of the singular value solvers
 * (src/sys/classes/ds/impls/svd/dssvd.c:197-290); the matrices are of order <= ncv. */
int ksd_gesvd(int n, const double *A, int lda, double *s, double *U, int ldu, double *VT, int ldvt) {
  if (n <= 0) return 0;
  double *G = (double *)malloc((size_t)n * n * sizeof(double));
  double *V = (double *)calloc((size_t)n * n, sizeof(double));
  int *ord = (int *)malloc((size_t)n * sizeof(int));
  if (!G || !V || !ord) { free(G); free(V); free(ord); return -1; }
  int i, j, p, q, sweep, rotated = 1;
  for (j = 0; j < n; j++) { memcpy(G + (size_t)j * n, A + (size_t)j * lda, (size_t)n * sizeof(double)); V[j + (size_t)j * n] = 1.0; }
  const double eps = 2.2204460492503131e-16;
  for (sweep = 0; sweep < 60 && rotated; sweep++) {
    rotated = 0;
    for (p = 0; p < n - 1; p++) {
      for (q = p + 1; q < n; q++) {
        double *gp = G + (size_t)p * n, *gq = G + (size_t)q * n;
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (i = 0; i < n; i++) { alpha += gp[i] * gp[i]; beta += gq[i] * gq[i]; gamma += gp[i] * gq[i]; }
        if (gamma == 0.0 || fabs(gamma) <= eps * sqrt(alpha * beta)) continue;
        rotated = 1;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        double *vp = V + (size_t)p * n, *vq = V + (size_t)q * n;
        for (i = 0; i < n; i++) {
          const double a = gp[i], b = gq[i];
          gp[i] = c * a - sn * b; gq[i] = sn * a + c * b;
          const double va = vp[i], vb = vq[i];
          vp[i] = c * va - sn * vb; vq[i] = sn * va + c * vb;
        }
      }
    }
  }
  if (rotated) { free(G); free(V); free(ord); return 1; }       /* no convergence */
  /* singular values, descending */
  for (j = 0; j < n; j++) {
    double nrm = 0.0;
    for (i = 0; i < n; i++) nrm = hypot(nrm, G[i + (size_t)j * n]);
    s[j] = nrm; ord[j] = j;
  }
  for (i = 1; i < n; i++) {                                     /* insertion sort of ord by s, stable */
    const int o = ord[i];
    for (j = i - 1; j >= 0 && s[ord[j]] < s[o]; j--) ord[j + 1] = ord[j];
    ord[j + 1] = o;
  }
  double smax = s[ord[0]];
  int rank = 0;
  for (j = 0; j < n; j++) {
    const int o = ord[j];
    for (i = 0; i < n; i++) VT[j + (size_t)i * ldvt] = V[i + (size_t)o * n];
    if (s[o] > (double)n * eps * smax && s[o] > 0.0) {
      for (i = 0; i < n; i++) U[i + (size_t)j * ldu] = G[i + (size_t)o * n] / s[o];
      rank = j + 1;
    }
  }
  /* columns of U for (numerically) zero singular values: complete to an orthonormal basis */
  for (j = rank; j < n; j++) {
    int e;
    for (e = 0; e < n; e++) {
      double nrm = 0.0;
      for (i = 0; i < n; i++) U[i + (size_t)j * ldu] = (i == e) ? 1.0 : 0.0;
      for (int pass = 0; pass < 2; pass++)
        for (q = 0; q < j; q++) {
          double d = 0.0;
          for (i = 0; i < n; i++) d += U[i + (size_t)q * ldu] * U[i + (size_t)j * ldu];
          for (i = 0; i < n; i++) U[i + (size_t)j * ldu] -= d * U[i + (size_t)q * ldu];
        }
      for (i = 0; i < n; i++) nrm = hypot(nrm, U[i + (size_t)j * ldu]);
      if (nrm > 0.5 / sqrt((double)n)) { for (i = 0; i < n; i++) U[i + (size_t)j * ldu] /= nrm; break; }
    }
  }
  {
    double *tmp = (double *)malloc((size_t)n * sizeof(double));
    if (!tmp) { free(G); free(V); free(ord); return -1; }
    for (j = 0; j < n; j++) tmp[j] = s[ord[j]];
    memcpy(s, tmp, (size_t)n * sizeof(double));
    free(tmp);
  }
  free(G); free(V); free(ord);
  return 0;
}

/* ---- Cholesky factor and triangular inverse (potrf "U", trtri "U","N") ---------------------- */
int ksd_potrf_upper(int n, double *A, int lda) {
  for (int j = 0; j < n; j++) {
    double d = A[j + (size_t)j * lda];
    for (int k = 0; k < j; k++) d -= A[k + (size_t)j * lda] * A[k + (size_t)j * lda];
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    A[j + (size_t)j * lda] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[j + (size_t)i * lda];
      for (int k = 0; k < j; k++) s -= A[k + (size_t)j * lda] * A[k + (size_t)i * lda];
      A[j + (size_t)i * lda] = s / d;
    }
  }
  return 0;
}

int ksd_trtri_upper(int n, double *A, int lda) {
  for (int j = 0; j < n; j++) {
    if (A[j + (size_t)j * lda] == 0.0) return j + 1;
    A[j + (size_t)j * lda] = 1.0 / A[j + (size_t)j * lda];
    const double ajj = -A[j + (size_t)j * lda];
    /* column j above the diagonal: x = T(0:j,0:j) * a(0:j,j), T the inverse computed so far */
    for (int i = 0; i < j; i++) {
      double s = 0.0;
      for (int k = i; k < j; k++) s += A[i + (size_t)k * lda] * A[k + (size_t)j * lda];
      A[i + (size_t)j * lda] = s;          /* row i only needs entries k >= i, which are still the originals below row i */
    }
    for (int i = 0; i < j; i++) A[i + (size_t)j * lda] *= ajj;
  }
  return 0;
}

/* ---- LU with partial pivoting (the direct solve behind a spectral transformation of a small
 * problem; the reference uses PETSc's PCLU there) -------------------------------------------- */
int ksd_getrf(int n, double *A, int lda, int *piv) {
  for (int k = 0; k < n; k++) {
    int p = k;
    double big = fabs(A[k + (size_t)k * lda]);
    for (int i = k + 1; i < n; i++) { const double v = fabs(A[i + (size_t)k * lda]); if (v > big) { big = v; p = i; } }
    piv[k] = p;
    if (big == 0.0) return k + 1;
    if (p != k) for (int j = 0; j < n; j++) { const double t = A[k + (size_t)j * lda]; A[k + (size_t)j * lda] = A[p + (size_t)j * lda]; A[p + (size_t)j * lda] = t; }
    const double inv = 1.0 / A[k + (size_t)k * lda];
    for (int i = k + 1; i < n; i++) A[i + (size_t)k * lda] *= inv;
    for (int j = k + 1; j < n; j++) {
      const double akj = A[k + (size_t)j * lda];
      if (akj != 0.0) for (int i = k + 1; i < n; i++) A[i + (size_t)j * lda] -= A[i + (size_t)k * lda] * akj;
    }
  }
  return 0;
}

void ksd_getrs(int n, const double *LU, int lda, const int *piv, double *b) {
  for (int k = 0; k < n; k++) { const int p = piv[k]; if (p != k) { const double t = b[k]; b[k] = b[p]; b[p] = t; } }
  for (int k = 0; k < n; k++) { const double bk = b[k]; if (bk != 0.0) for (int i = k + 1; i < n; i++) b[i] -= LU[i + (size_t)k * lda] * bk; }
  for (int k = n - 1; k >= 0; k--) {
    b[k] /= LU[k + (size_t)k * lda];
    const double bk = b[k];
    if (bk != 0.0) for (int i = 0; i < k; i++) b[i] -= LU[i + (size_t)k * lda] * bk;
  }
}
