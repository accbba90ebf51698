/* Mat: the operator seen by the solver -- MatMult, MatCreateVecs and the sizes.
 * "aijhip" wraps a libksgpu CSR matrix; a shell matrix calls back into user code
 * (MatCreateShell in the reference's tests/test8.c:57-61). */
#include <stdio.h>

#include "ksimpl.h"

static PetscErrorCode MatMult_AIJHIP(Mat A, Vec x, Vec y) {
  CHKKSG(ksg_mat_mult((ksg_mat)A->data, x->array, y->array));
  return 0;
}
static PetscErrorCode MatCreateVecs_AIJHIP(Mat A, Vec *right, Vec *left) {
  PetscErrorCode ierr;
  if (right) { ierr = VecCreateHIP((ksg_ctx)A->ctx, A->cn, A->cN, A->col0, right);CHKERRQ(ierr); }
  if (left) { ierr = VecCreateHIP((ksg_ctx)A->ctx, A->n, A->N, A->row0, left);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode MatResidualNorm_AIJHIP(Mat A, PetscScalar kr, PetscScalar ki, Vec xr, Vec xi, Vec work, PetscReal *norm) {
  CHKKSG(ksg_mat_residual_norm((ksg_mat)A->data, kr, ki, xr->array, xi ? xi->array : NULL, work ? work->array : NULL, norm));
  return 0;
}
static PetscErrorCode MatDestroy_AIJHIP(Mat A) { CHKKSG(ksg_mat_destroy((ksg_mat)A->data)); A->data = NULL; return 0; }

static PetscErrorCode MatTranspose_AIJHIP(Mat A, Mat *AT);

/* wraps a libksgpu matrix */
static PetscErrorCode MatWrap_AIJHIP(ksg_ctx ctx, ksg_mat m, Mat *A) {
  Mat a = (Mat)calloc(1, sizeof(*a));
  if (!a) { ksg_mat_destroy(m); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  a->ops.mult = MatMult_AIJHIP; a->ops.createvecs = MatCreateVecs_AIJHIP; a->ops.destroy = MatDestroy_AIJHIP;
  a->ops.transpose = MatTranspose_AIJHIP;
  a->data = m; a->ctx = ctx;
  if (ksg_mat_sizes(m, &a->n, &a->N, &a->row0, NULL) || ksg_mat_col_sizes(m, &a->cn, &a->cN, &a->col0)) {
    ksg_mat_destroy(m); free(a);
    SETERRQ(PETSC_ERR_LIB, "libksgpu: %s", ksg_last_error());
  }
  /* the residual of an eigenpair only makes sense for a square operator */
  a->ops.residualnorm = (a->N == a->cN && a->n == a->cn) ? MatResidualNorm_AIJHIP : NULL;
  *A = a;
  return 0;
}

static PetscErrorCode MatTranspose_AIJHIP(Mat A, Mat *AT) {
  ksg_mat t = NULL;
  CHKKSG(ksg_mat_transpose((ksg_mat)A->data, &t));
  return MatWrap_AIJHIP((ksg_ctx)A->ctx, t, AT);
}

PetscErrorCode MatCreateAIJHIP(ksg_ctx ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0, const int64_t *rowptr,
                               const int32_t *colidx, const double *val, Mat *A) {
  if (!ctx || !A) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ksg_mat m = NULL;
  CHKKSG(ksg_mat_create_csr(ctx, n, N, row0, rowptr, colidx, val, &m));
  return MatWrap_AIJHIP(ctx, m, A);
}

PetscErrorCode MatCreateAIJHIPRect(ksg_ctx ctx, PetscInt64 m, PetscInt64 M, PetscInt64 row0, PetscInt64 n, PetscInt64 N,
                                   PetscInt64 col0, const int64_t *rowptr, const int32_t *colidx, const double *val, Mat *A) {
  if (!ctx || !A) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  ksg_mat mm = NULL;
  CHKKSG(ksg_mat_create_csr_rect(ctx, m, M, row0, n, N, col0, rowptr, colidx, val, &mm));
  return MatWrap_AIJHIP(ctx, mm, A);
}

PetscErrorCode MatTranspose(Mat A, Mat *AT) {
  PetscErrorCode ierr;
  if (!A || !AT) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (!A->ops.transpose) SETERRQ(PETSC_ERR_SUP, "This matrix type cannot be transposed");
  ierr = A->ops.transpose(A, AT);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode MatCreateShell(void *ctx, PetscInt64 n, PetscInt64 N, PetscInt64 row0, void *user,
                              PetscErrorCode (*mult)(Mat, Vec, Vec), PetscErrorCode (*createvecs)(Mat, Vec *, Vec *), Mat *A) {
  if (!A || !mult) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (n < 0 || N < n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid sizes");
  Mat a = (Mat)calloc(1, sizeof(*a));
  if (!a) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  a->ops.mult = mult; a->ops.createvecs = createvecs; a->ops.destroy = NULL; a->ops.residualnorm = NULL; a->ops.transpose = NULL;
  a->n = n; a->N = N; a->row0 = row0; a->data = user; a->ctx = ctx;
  a->cn = n; a->cN = N; a->col0 = row0;
  *A = a;
  return 0;
}

/* a shell whose MatMult takes vectors of another layout than its rows (rectangular), and
 * the routine that builds its transpose (MatShellSetOperation(A,MATOP_TRANSPOSE,...)) */
PetscErrorCode MatShellSetColumnLayout(Mat A, PetscInt64 n, PetscInt64 N, PetscInt64 col0) {
  if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix");
  if (n < 0 || N < n || col0 < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Invalid sizes");
  A->cn = n; A->cN = N; A->col0 = col0;
  return 0;
}
PetscErrorCode MatShellSetTranspose(Mat A, PetscErrorCode (*transpose)(Mat, Mat *)) {
  if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix");
  A->ops.transpose = transpose;
  return 0;
}

/* the libksgpu matrix behind an "aijhip" operator, NULL for any other type */
void *MatGetDeviceHandle_Private(Mat A) { return (A && A->ops.mult == MatMult_AIJHIP) ? A->data : NULL; }

PetscErrorCode MatShellGetContext(Mat A, void **user) { if (!A || !user) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument"); *user = A->data; return 0; }

PetscErrorCode MatMult(Mat A, Vec x, Vec y) {
  PetscErrorCode ierr;
  if (!A || !x || !y) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (x == y || x->array == y->array) SETERRQ(PETSC_ERR_ARG_WRONG, "x and y must be different vectors");
  if (x->n != A->cn || y->n != A->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat local size %lld x %lld, vectors %lld and %lld", (long long)A->n, (long long)A->cn, (long long)x->n, (long long)y->n);
  if (A->ops.mult == MatMult_AIJHIP) { ierr = A->ops.mult(A, x, y);CHKERRQ(ierr); }
  else {
    /* user code: it may read x->array and write y->array with kernels or copies of its own,
     * so x must hold every update the device library defers, and the library must learn that
     * y changed (it keeps reductions it formed ahead of their request) */
    ierr = VecForeignAccessBegin_Private(x);CHKERRQ(ierr);
    ierr = A->ops.mult(A, x, y);CHKERRQ(ierr);
    ierr = VecForeignAccessEnd_Private(y);CHKERRQ(ierr);
  }
  y->state++;
  return 0;
}

PetscErrorCode MatCreateVecs(Mat A, Vec *right, Vec *left) {
  PetscErrorCode ierr;
  if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix");
  if (!A->ops.createvecs) SETERRQ(PETSC_ERR_SUP, "This matrix type cannot create vectors");
  ierr = A->ops.createvecs(A, right, left);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatGetSize(Mat A, PetscInt64 *M, PetscInt64 *N) { if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix"); if (M) *M = A->N; if (N) *N = A->cN; return 0; }
PetscErrorCode MatGetLocalSize(Mat A, PetscInt64 *m, PetscInt64 *n) { if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "Null matrix"); if (m) *m = A->n; if (n) *n = A->cn; return 0; }
PetscErrorCode MatDestroy(Mat *A) {
  PetscErrorCode ierr;
  if (!A || !*A) return 0;
  if ((*A)->ops.destroy) { ierr = (*A)->ops.destroy(*A);CHKERRQ(ierr); }
  free(*A);
  *A = NULL;
  return 0;
}

/* ---- PETSc binary matrices --------------------------------------------------------
 * The format PetscViewerBinary writes for an AIJ matrix and MatLoad reads (the files of
 * share/slepc/datafiles/matrices, loaded by src/eps/examples/tutorials/ex4.c:56-60):
 * big-endian throughout -- int32 class id 1211216 (MAT_FILE_CLASSID), int32 rows M,
 * int32 columns N, int32 nonzeros nz, int32 row lengths [M], int32 column indices [nz]
 * (global, 0-based), float64 values [nz]. */
#define KS_MAT_FILE_CLASSID 1211216

static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
static double be64f(const unsigned char *p) {
  uint64_t u = 0;
  for (int i = 0; i < 8; i++) u = (u << 8) | p[i];
  double d;
  memcpy(&d, &u, sizeof(d));
  return d;
}

struct ks_entry { int32_t col; double val; };
static int ks_entry_cmp(const void *a, const void *b) {
  const int32_t ca = ((const struct ks_entry *)a)->col, cb = ((const struct ks_entry *)b)->col;
  return (ca > cb) - (ca < cb);
}

static PetscErrorCode read_exact(FILE *f, void *buf, size_t bytes, const char *what, const char *filename) {
  if (fread(buf, 1, bytes, f) != bytes) SETERRQ(PETSC_ERR_FILE_READ, "%s: file ends inside the %s", filename, what);
  return 0;
}

PetscErrorCode KSMatReadPetscBinary(const char *filename, PetscInt64 *M, PetscInt64 *N, int64_t **rowptr,
                                    int32_t **colidx, double **val) {
  PetscErrorCode ierr = 0;
  if (!filename || !M || !N || !rowptr || !colidx || !val) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  *rowptr = NULL; *colidx = NULL; *val = NULL;
  FILE *f = fopen(filename, "rb");
  if (!f) SETERRQ(PETSC_ERR_FILE_OPEN, "Cannot open %s", filename);
  unsigned char hdr[16], *raw = NULL;
  int64_t *rp = NULL; int32_t *ci = NULL; double *va = NULL; struct ks_entry *tmp = NULL;
  ierr = read_exact(f, hdr, sizeof(hdr), "header", filename);
  if (ierr) goto fail;
  if (be32(hdr) != KS_MAT_FILE_CLASSID) { ierr = KSError(__func__, __LINE__, PETSC_ERR_FILE_UNEXPECTED, "%s: not a PETSc binary matrix (class id %u)", filename, be32(hdr)); goto fail; }
  {
    const int32_t m = (int32_t)be32(hdr + 4), n = (int32_t)be32(hdr + 8), nz = (int32_t)be32(hdr + 12);
    if (m < 0 || n < 0) { ierr = KSError(__func__, __LINE__, PETSC_ERR_FILE_UNEXPECTED, "%s: negative dimensions", filename); goto fail; }
    if (nz < 0) { ierr = KSError(__func__, __LINE__, PETSC_ERR_SUP, "%s: dense-format matrix files are not supported", filename); goto fail; }
    const size_t rawbytes = (size_t)m * 4 > (size_t)nz * 8 ? (size_t)m * 4 : (size_t)nz * 8;
    raw = (unsigned char *)malloc(rawbytes ? rawbytes : 1);
    rp = (int64_t *)malloc(((size_t)m + 1) * sizeof(*rp));
    ci = (int32_t *)malloc(((size_t)nz ? (size_t)nz : 1) * sizeof(*ci));
    va = (double *)malloc(((size_t)nz ? (size_t)nz : 1) * sizeof(*va));
    if (!raw || !rp || !ci || !va) { ierr = KSError(__func__, __LINE__, PETSC_ERR_MEM, "Out of memory"); goto fail; }
    ierr = read_exact(f, raw, (size_t)m * 4, "row lengths", filename);
    if (ierr) goto fail;
    rp[0] = 0;
    for (int32_t r = 0; r < m; r++) {
      const int32_t len = (int32_t)be32(raw + 4 * (size_t)r);
      if (len < 0) { ierr = KSError(__func__, __LINE__, PETSC_ERR_FILE_UNEXPECTED, "%s: negative length of row %d", filename, r); goto fail; }
      rp[r + 1] = rp[r] + len;
    }
    if (rp[m] != nz) { ierr = KSError(__func__, __LINE__, PETSC_ERR_FILE_UNEXPECTED, "%s: row lengths add up to %lld, header says %d nonzeros", filename, (long long)rp[m], nz); goto fail; }
    ierr = read_exact(f, raw, (size_t)nz * 4, "column indices", filename);
    if (ierr) goto fail;
    for (int32_t e = 0; e < nz; e++) {
      ci[e] = (int32_t)be32(raw + 4 * (size_t)e);
      if (ci[e] < 0 || ci[e] >= n) { ierr = KSError(__func__, __LINE__, PETSC_ERR_FILE_UNEXPECTED, "%s: column index %d outside [0,%d)", filename, ci[e], n); goto fail; }
    }
    ierr = read_exact(f, raw, (size_t)nz * 8, "values", filename);
    if (ierr) goto fail;
    for (int32_t e = 0; e < nz; e++) va[e] = be64f(raw + 8 * (size_t)e);
    /* AIJ storage keeps every row in ascending column order (MatSetValues sorts on
     * insertion); files written from an assembled matrix already are */
    for (int32_t r = 0; r < m; r++) {
      int sorted = 1;
      for (int64_t e = rp[r] + 1; e < rp[r + 1]; e++) if (ci[e] <= ci[e - 1]) { sorted = 0; break; }
      if (sorted) continue;
      const size_t len = (size_t)(rp[r + 1] - rp[r]);
      struct ks_entry *t2 = (struct ks_entry *)realloc(tmp, len * sizeof(*tmp));
      if (!t2) { ierr = KSError(__func__, __LINE__, PETSC_ERR_MEM, "Out of memory"); goto fail; }
      tmp = t2;
      for (size_t e = 0; e < len; e++) { tmp[e].col = ci[rp[r] + (int64_t)e]; tmp[e].val = va[rp[r] + (int64_t)e]; }
      qsort(tmp, len, sizeof(*tmp), ks_entry_cmp);
      for (size_t e = 1; e < len; e++)
        if (tmp[e].col == tmp[e - 1].col) { ierr = KSError(__func__, __LINE__, PETSC_ERR_FILE_UNEXPECTED, "%s: row %d holds column %d twice", filename, r, tmp[e].col); goto fail; }
      for (size_t e = 0; e < len; e++) { ci[rp[r] + (int64_t)e] = tmp[e].col; va[rp[r] + (int64_t)e] = tmp[e].val; }
    }
    *M = m; *N = n;
  }
  free(raw); free(tmp); fclose(f);
  *rowptr = rp; *colidx = ci; *val = va;
  return 0;
fail:
  free(raw); free(tmp); free(rp); free(ci); free(va); fclose(f);
  return ierr;
}

void KSFree(void *p) { free(p); }

static void put_be32(unsigned char *p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }

/* The writer of the same format (what MatView does on a binary viewer). */
PetscErrorCode KSMatWritePetscBinary(const char *filename, PetscInt64 M, PetscInt64 N, const int64_t *rowptr,
                                     const int32_t *colidx, const double *val) {
  if (!filename || !rowptr || (M > 0 && rowptr[M] > 0 && (!colidx || !val))) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (M < 0 || N < 0 || M > INT32_MAX || N > INT32_MAX || rowptr[M] > INT32_MAX)
    SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "The format holds 32-bit sizes; got %lld x %lld with %lld nonzeros", (long long)M, (long long)N, (long long)rowptr[M]);
  const int64_t nz = rowptr[M];
  const size_t bytes = 16 + 4 * (size_t)M + 12 * (size_t)nz;
  unsigned char *buf = (unsigned char *)malloc(bytes), *p = buf;
  if (!buf) SETERRQ(PETSC_ERR_MEM, "Out of memory");
  put_be32(p, KS_MAT_FILE_CLASSID); put_be32(p + 4, (uint32_t)M); put_be32(p + 8, (uint32_t)N); put_be32(p + 12, (uint32_t)nz);
  p += 16;
  for (PetscInt64 r = 0; r < M; r++, p += 4) put_be32(p, (uint32_t)(rowptr[r + 1] - rowptr[r]));
  for (int64_t e = 0; e < nz; e++, p += 4) put_be32(p, (uint32_t)colidx[e]);
  for (int64_t e = 0; e < nz; e++, p += 8) {
    uint64_t u;
    memcpy(&u, &val[e], sizeof(u));
    for (int i = 0; i < 8; i++) p[i] = (unsigned char)(u >> (56 - 8 * i));
  }
  FILE *f = fopen(filename, "wb");
  if (!f) { free(buf); SETERRQ(PETSC_ERR_FILE_OPEN, "Cannot open %s for writing", filename); }
  const size_t w = fwrite(buf, 1, bytes, f);
  const int rc = fclose(f);
  free(buf);
  if (w != bytes || rc) SETERRQ(PETSC_ERR_FILE_WRITE, "Short write to %s", filename);
  return 0;
}

PetscErrorCode MatLoadAIJHIP(ksg_ctx ctx, const char *filename, int rank, int nranks, Mat *A) {
  PetscErrorCode ierr;
  PetscInt64 M, N;
  int64_t *rp; int32_t *ci; double *va;
  if (!ctx || !A) SETERRQ(PETSC_ERR_ARG_NULL, "Null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Rank %d of %d", rank, nranks);
  ierr = KSMatReadPetscBinary(filename, &M, &N, &rp, &ci, &va);CHKERRQ(ierr);
  /* contiguous row blocks, the first M mod nranks ranks one row longer (PetscSplitOwnership) */
  const PetscInt64 base = M / nranks, extra = M % nranks;
  const PetscInt64 row0 = rank * base + (rank < extra ? rank : extra), n = base + (rank < extra ? 1 : 0);
  int64_t *lrp = (int64_t *)malloc(((size_t)n + 1) * sizeof(*lrp));
  if (!lrp) { free(rp); free(ci); free(va); SETERRQ(PETSC_ERR_MEM, "Out of memory"); }
  for (PetscInt64 r = 0; r <= n; r++) lrp[r] = rp[row0 + r] - rp[row0];
  /* columns (the layout of x) are split the same way */
  const PetscInt64 cbase = N / nranks, cextra = N % nranks;
  const PetscInt64 col0 = rank * cbase + (rank < cextra ? rank : cextra), cn = cbase + (rank < cextra ? 1 : 0);
  ierr = MatCreateAIJHIPRect(ctx, n, M, row0, cn, N, col0, lrp, ci + rp[row0], va + rp[row0], A);
  free(lrp); free(rp); free(ci); free(va);
  CHKERRQ(ierr);
  return 0;
}
