/* LAPACK-free dense kernels for the projected problem (see dense.c). */
#ifndef KS_DENSE_H
#define KS_DENSE_H

void ksd_lartg(double f, double g, double *c, double *s, double *r);
void ksd_rot(int n, double *x, double *y, double c, double s);
int ksd_steqr(int n, double *d, double *e, double *Z, int ldz, int nrows);
int ksd_gehrd_orghr(int n, int ilo, double *A, int lda, double *Q, int ldq);
void ksd_lanv2(double *a, double *b, double *c, double *d, double *rt1r, double *rt1i,
               double *rt2r, double *rt2i, double *cs, double *sn);
int ksd_hseqr(int n, int ilo, double *T, int ldt, double *wr, double *wi, double *Z, int ldz);
int ksd_trexc(int n, double *T, int ldt, double *Q, int ldq, int nq, int ifst, int ilst);
int ksd_trevc_right(int n, const double *T, int ldt, int k, double *xr, double *xi);

/* A = U diag(s) VT for a square n x n matrix (column-major), singular values descending;
 * U and VT are n x n orthogonal.  A is not modified. */
int ksd_gesvd(int n, const double *A, int lda, double *s, double *U, int ldu, double *VT, int ldvt);

/* upper Cholesky factor of the symmetric positive definite n x n matrix in A (upper triangle
 * referenced), in place; returns k > 0 if the leading minor of order k is not positive */
int ksd_potrf_upper(int n, double *A, int lda);
/* inverse of an upper triangular matrix, in place; returns k > 0 for a zero on the diagonal */
int ksd_trtri_upper(int n, double *A, int lda);
/* LU factorisation with partial pivoting of the n x n matrix A (column-major), in place;
 * piv[k] = row exchanged with row k.  Returns k > 0 when the pivot of step k is exactly zero. */
int ksd_getrf(int n, double *A, int lda, int *piv);
/* solves A x = b with the factors of ksd_getrf; b is overwritten by x */
void ksd_getrs(int n, const double *LU, int lda, const int *piv, double *b);

#endif
