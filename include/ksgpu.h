/*
 * ksgpu.h -- C ABI of libksgpu, the MI355X (gfx950) device library behind the
 * Krylov-Schur hot path: basis-vector (BV) kernels and the CSR operator.
 *
 * Every entry point takes plain pointers and sizes and returns an int error
 * code (0 = success, see KSG_ERR_*); ksg_last_error() gives the message of the
 * last failure on the calling thread.  Device pointers are raw HIP device
 * addresses.  All work is enqueued on the context's stream and is in program
 * order; entry points that return host data (dotvec, norm, dot) have completed
 * when they return.  One host thread per context; collective entry points must
 * be called by every rank in the same order.
 *
 * Each declaration cites the interface of the reference (SLEPc 3.7.3-dev, paths
 * relative to the reference root) that it stands in for.
 */
#ifndef KSGPU_H
#define KSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSG_OK             0
#define KSG_ERR_ARG        1   /* bad argument (range, NULL, alignment)            */
#define KSG_ERR_HIP        2   /* a HIP runtime call failed                        */
#define KSG_ERR_MEM        3   /* host or device allocation failed                 */
#define KSG_ERR_COMM       4   /* communicator / RCCL failure                      */
#define KSG_ERR_UNSUPPORTED 5

/* norm types: values of PETSc's NormType as used by BVNorm (bvglobal.c:520-545) */
#define KSG_NORM_1         0
#define KSG_NORM_2         1
#define KSG_NORM_FROBENIUS 2
#define KSG_NORM_INFINITY  3

typedef struct ksg_ctx_s *ksg_ctx;
typedef struct ksg_bv_s  *ksg_bv;
typedef struct ksg_mat_s *ksg_mat;

/* ------------------------------------------------------------------ context */

/* Binds to HIP device `device`, creates the compute and communication streams
 * and the small pinned/device reduction buffers.  Stands in for the per-process
 * set-up PETSc does for VECCUDA (SlepcInitialize, src/sys/slepcinit.c:249). */
int ksg_ctx_create(int device, ksg_ctx *ctx);
int ksg_ctx_destroy(ksg_ctx ctx);
/* Blocks until everything enqueued on the context has finished. */
int ksg_ctx_sync(ksg_ctx ctx);
/* Enqueues whatever update of a column is still pending (KSG_OPT_LAZY_UPDATE) without
 * waiting for it: afterwards work enqueued on ksg_ctx_stream() by the caller sees every
 * column up to date.  Every other entry point does this itself where it matters. */
int ksg_ctx_flush(ksg_ctx ctx);
/* Must be called after device vectors of this context were written by anything other than
 * this library (a kernel or copy of the caller, e.g. the MatMult of a user-defined operator):
 * applies a pending update and drops reductions formed ahead of their request
 * (KSG_OPT_FUSE_DOTS).  ksg_bv_touch is the same for a caller that holds the BV. */
int ksg_ctx_touch(ksg_ctx ctx);
/* The HIP stream (hipStream_t) all kernels are launched on. */
void *ksg_ctx_stream(ksg_ctx ctx);
/* Tuning switches.  KSG_OPT_FUSE_DOTS (default 1): ksg_bv_multvec on the column that
 * follows the active ones also forms that column's products with the active columns
 * and with itself, and the ksg_bv_dotvec that asks for them next costs no pass over
 * the basis.  No counterpart in the reference (its two GEMVs are separate BLAS calls,
 * bvblas.c:65-76,242-266); results are the same reductions in a different order.
 * KSG_OPT_LAZY_UPDATE (default 1): such an update is not launched at once.  If the caller
 * next asks for the updated column's products (a refinement pass) they are formed in one
 * read-only pass and the column stays pending; a second update joins the first; the scaling
 * that normalises the column is folded in, and one pass applies everything with a single
 * store of the column.  Any other call that touches device data applies what is pending
 * first.  The stored values are bit for bit those of separate launches.
 * KSG_OPT_DOT_GRID / KSG_OPT_MV_GRID: upper bound on the workgroups along the rows of
 * the reduction (default 256, one per CU) and update (default 1024) kernels.
 * KSG_OPT_HOT_COLS (default 1048576 = 8 MiB of x; 0 disables): a CSR operator created
 * afterwards whose local column block has more than four times that many columns (262144 =
 * 2 MiB, half of one XCD's L2, are tried for a smaller one), with columns referenced at least four times as often as the average that together
 * hold 15 % or more of the entries (power-law graphs), keeps a packed copy of the x entries of
 * its most referenced columns: each product gathers them into a dense array that stays in cache,
 * and the matrix entries of those columns index it instead of x.  Same values, same order
 * of sums: results are bit for bit those without the packing.
 * KSG_OPT_BINS (default 0 = by size; 1 = never; 2 = always): a block of a power-law operator
 * created afterwards -- the local block of a rank, and the ghost block of a distributed operator,
 * whose x is the ghost buffer -- (x of 16 MiB or more when chosen by size) keeps the entries of its
 * well filled (row block, column block) cells in a binned form: one pass looks x up in LDS slices
 * and writes the products in the order of the row blocks, a second pass sums them per row block in
 * LDS and adds the diagonal term; the remaining entries stay a CSR over the rows they touch and
 * are added afterwards.  The gathers from a large x, one memory request each, become two streams.
 * A row's sum is formed from its binned entries by
 * column block, its diagonal entry, and its remaining entries: the same products added in another
 * order, the same order in every product. */
enum { KSG_OPT_FUSE_DOTS = 1, KSG_OPT_DOT_GRID = 2, KSG_OPT_MV_GRID = 3, KSG_OPT_LAZY_UPDATE = 4, KSG_OPT_HOT_COLS = 5,
       KSG_OPT_BINS = 7 };
int ksg_ctx_set_option(ksg_ctx ctx, int option, int value);
const char *ksg_last_error(void);

/* Communicator.  The reference reduces with MPI_Allreduce on PetscObjectComm(bv)
 * (bvblas.c:260,305; sveccuda.cu:341); here rank r of `nranks` joins an RCCL
 * communicator built from a 128-byte unique id that rank 0 obtained from
 * ksg_comm_unique_id() and distributed out of band. */
int ksg_comm_unique_id(void *id128);
int ksg_ctx_comm_init_rccl(ksg_ctx ctx, int rank, int nranks, const void *id128);
/* Alternative transport: the caller supplies the two collectives (used with
 * torch.distributed, and by the CPU tests).  `buf` is a device pointer when
 * `on_device` was 1 at registration, otherwise a host pointer. */
typedef int (*ksg_allreduce_fn)(void *user, double *buf, int count, int op /*0 sum, 1 max*/);
typedef int (*ksg_exchange_fn)(void *user, const double *sendbuf, const int64_t *sendoff,
                               double *recvbuf, const int64_t *recvoff);
int ksg_ctx_comm_init_callbacks(ksg_ctx ctx, int rank, int nranks, int on_device,
                                ksg_allreduce_fn allreduce, ksg_exchange_fn exchange, void *user);
/* Leaves the communicator (collective, like ncclCommDestroy / MPI_Comm_free). */
int ksg_ctx_comm_finalize(ksg_ctx ctx);
int ksg_ctx_comm_rank(ksg_ctx ctx, int *rank, int *nranks);
/* Collective self-test of the communicator: a sum over the ranks with a known answer and a ring
 * exchange (every rank sends to the next and receives from the previous one) through the path the
 * halo exchange of MatMult uses.  The reference relies on MPI being there and correct; a launcher
 * calls this once after ksg_ctx_comm_init_* so that a transport that came up wrong fails here. */
int ksg_ctx_comm_selftest(ksg_ctx ctx);

/* --------------------------------------------------------- device vectors */
/* The PETSc Vec operations that hot-path host code invokes on BV columns
 * (VecCopy bvbasic.c:1645, VecAXPY epssolve.c:531, VecNorm epssolve.c:533,
 * VecDot bvglobal.c:477, VecScale/VecSet epsdefault.c:374,398). */
int ksg_vec_alloc(ksg_ctx ctx, int64_t n, double **x);
int ksg_vec_free(ksg_ctx ctx, double *x);
int ksg_vec_upload(ksg_ctx ctx, double *x_dev, const double *x_host, int64_t n);
int ksg_vec_download(ksg_ctx ctx, double *x_host, const double *x_dev, int64_t n);
int ksg_vec_copy(ksg_ctx ctx, const double *x, double *y, int64_t n);
int ksg_vec_set(ksg_ctx ctx, double *x, double alpha, int64_t n);
int ksg_vec_scale(ksg_ctx ctx, double *x, double alpha, int64_t n);
int ksg_vec_axpy(ksg_ctx ctx, double *y, double alpha, const double *x, int64_t n);
int ksg_vec_dot(ksg_ctx ctx, const double *x, const double *y, int64_t n, double *result); /* collective */
int ksg_vec_norm2(ksg_ctx ctx, const double *x, int64_t n, double *result);                 /* collective */
/* local rows only, no reduction over the ranks (VecDot_Seq under VecDot_MPI; the dot_local /
 * norm_local slots that VecDotBegin / VecNormBegin call before PETSc's own merged MPI_Allreduce) */
int ksg_vec_dot_local(ksg_ctx ctx, const double *x, const double *y, int64_t n, double *result);
/* x[i] = 2*u(splitmix64(seed ^ (row0+i))) - 1: start vector that depends only on
 * the global row index (stands in for BVSetRandomColumn, bvops.c:462-479). */
int ksg_vec_fill_hash(ksg_ctx ctx, double *x, int64_t n, int64_t row0, uint64_t seed);

/* ---------------------------------------------------------------- BV */
/* Storage as in BVSVEC (svec.c:417-520): column-major, constraints first,
 * element (r, column j) at base[(nc+j)*ld + r]; ld >= n is padded so that every
 * column starts on a 256-byte boundary.  Ops act on the active columns [l,k)
 * (BVSetActiveColumns, bvbasic.c:405-425); l may be -nc. */
int ksg_bv_create(ksg_ctx ctx, int64_t n_local, int64_t N, int m, ksg_bv *bv);   /* BVCreate_Svec svec.c:417 */
int ksg_bv_destroy(ksg_bv bv);                                                  /* BVDestroy_Svec svec.c:404 */
int ksg_bv_resize(ksg_bv bv, int m, int copy);                                  /* BVResize_Svec svec.c:280 */
int ksg_bv_set_active(ksg_bv bv, int l, int k);
/* The first nc of the nc+m columns held become constraints with the indices -nc..-1
 * (BVInsertConstraints bvfunc.c:433-439, BVSetNumConstraints bvbasic.c:276-294, storage
 * svec.c:41,322: constraints first); m becomes nc_old+m_old-nc.  No data moves. */
int ksg_bv_set_num_constraints(ksg_bv bv, int nc);
int ksg_bv_sizes(ksg_bv bv, int64_t *n_local, int64_t *N, int *m, int64_t *ld);
/* Device address of column j (BVGetColumn_Svec, svec.c:312-324: pv+(nc+j)*n).  Every
 * entry point of this library sees the column up to date; code of the caller that reads
 * it through this pointer on its own must call ksg_ctx_flush first (a kernel on
 * ksg_ctx_stream) or ksg_ctx_sync (anything else, e.g. hipMemcpy on another stream), which
 * apply an update deferred under KSG_OPT_LAZY_UPDATE. */
double *ksg_bv_column(ksg_bv bv, int j);
/* Must be called after column storage was written by anything other than this
 * library (a kernel of the caller, hipMemcpy) through a pointer from ksg_bv_column:
 * the counterpart of the state bump in BVRestoreColumn (bvbasic.c:1212-1215). */
int ksg_bv_touch(ksg_bv bv);
/* q[i] = sum_r X[r,l+i]*y[r], i<k-l, summed over ranks; q is a HOST array
 * (BVDotVec_Svec svec.c:122 -> BVDotVec_BLAS_Private bvblas.c:242). */
int ksg_bv_dotvec(ksg_bv X, const double *y_dev, double *q_host);
int ksg_bv_dotvec_local(ksg_bv X, const double *y_dev, double *q_host);          /* svec.c:145 */
/* No counterpart in the reference.  For callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the products
 * of ksg_bv_dotvec_local, q[i] = sum over the LOCAL rows of X[r,l+i]*y[r], written to the
 * DEVICE array q_dev (k-l doubles) by work enqueued on ksg_ctx_stream(); the call returns
 * without synchronising and calls no hipMalloc/hipFree.  The per-workgroup partial sums go to a
 * fixed 1 MiB device array (131072 doubles per GPU, held until the process ends) instead of the
 * context's growable workspace: (k-l) x partial sums per column must fit, i.e. up to 128 active
 * columns at the default KSG_OPT_DOT_GRID; more is refused with KSG_ERR_UNSUPPORTED.  The
 * kernels and the order of sums are those of ksg_bv_dotvec_local: the values are bit for bit
 * the same.  All calls on one GPU share that array: calls that overlap on one GPU (two
 * contexts, two streams) race; calls on one context are ordered by its stream. */
int ksg_bv_dotvec_noalloc(ksg_bv X, const double *y_dev, double *q_dev);
/* Split-phase reductions (the role of BVDotVecBegin/End, BVNormVecBegin/End, VecDotBegin/End,
 * bvglobal.c:190-310, 732-830, which the delayed Arnoldi and the one-sided bidiagonalisations
 * use to pay one reduction round per step, epskrylov.c:107-129, trlanczos.c:201-217).
 * ksg_red_open starts a batch; each post launches its reduction into the next free slot of a
 * device buffer and returns without synchronising (*offset = index of its first value);
 * ksg_red_close does one all-reduce over all posted values and delivers them to the host in
 * posting order.  Between open and close other operations may run; one batch per context.
 * ksg_red_close with count < 0 abandons the batch (after a failed post). */
int ksg_red_open(ksg_ctx ctx);
int ksg_red_post_dotvec(ksg_bv X, const double *y_dev, int *offset);     /* X(:,l:k)^T y: k-l values */
int ksg_red_post_dot(ksg_ctx ctx, const double *x_dev, const double *y_dev, int64_t n, int *offset);  /* x^T y: 1 value */
int ksg_red_close(ksg_ctx ctx, double *out_host, int count);
/* y = beta*y + alpha*X(:,l:k)*q; q is a HOST array of k-l coefficients
 * (BVMultVec_Svec svec.c:51 -> BVMultVec_BLAS_Private bvblas.c:65). */
int ksg_bv_multvec(ksg_bv X, double alpha, double beta, double *y_dev, const double *q_host);
/* V(:,s:e) = V(:,l:k)*Q(l:k,s:e) (or Q^T when trans); Q is a HOST column-major
 * array with leading dimension ldq, indexed absolutely
 * (BVMultInPlace_Svec svec.c:68 -> BVMultInPlace_BLAS_Private bvblas.c:83). */
int ksg_bv_multinplace(ksg_bv V, const double *Q_host, int ldq, int s, int e, int trans);
/* j<0: scale active block, else column j (BVScale_Svec svec.c:165). */
int ksg_bv_scale(ksg_bv bv, int j, double alpha);
/* j<0: norm of the active block, else of column j; collective
 * (BVNorm_Svec svec.c:182 -> BVNorm_LAPACK_Private bvblas.c:290). */
int ksg_bv_norm(ksg_bv bv, int j, int type, double *val);
int ksg_bv_norm_local(ksg_bv bv, int j, int type, double *val);                  /* svec.c:199 */
/* W(:,lw:kw) = V(:,lv:kv) (BVCopy_Svec svec.c:263). */
int ksg_bv_copy(ksg_bv V, ksg_bv W);
/* V(:,i) = V(:,j) (BVCopyColumn bvbasic.c:1628: VecCopy on two column views). */
int ksg_bv_copy_column(ksg_bv V, int j, int i);
/* Y = beta*Y + alpha*X*Q, Q host array ldq x *, NULL = identity pattern (AXPY)
 * (BVMult_Svec svec.c:27). */
int ksg_bv_mult(ksg_bv Y, double alpha, double beta, ksg_bv X, const double *Q_host, int ldq);
/* M(ly:ky, lx:kx) = Y^T X, M host array with leading dimension ldm; collective
 * (BVDot_Svec svec.c:102). */
int ksg_bv_dot(ksg_bv X, ksg_bv Y, double *M_host, int ldm);
/* Tall-skinny QR of columns [0,k): V = Q R with Q^T Q = I left in V and R (upper triangular,
 * positive diagonal, host array ldr x k, may be NULL); collective.  At most 64 columns, no
 * constraints; numerically dependent columns are an error.  (BVOrthogonalize_TSQR
 * bvorthog.c:770 -> BVOrthogonalize_LAPACK_Private bvblas.c:345: geqrf/orgqr on the local
 * rows plus a QR of the stacked triangles, two processes at most; here a Householder
 * reduction tree on the device for R, any number of ranks, and Q = V R^-1 in two sweeps.) */
int ksg_bv_tsqr(ksg_bv V, double *R_host, int ldr);
/* W(:,lw+j) = A*V(:,lv+j), column by column (BVMatMult_Svec svec.c:216,250-256). */
int ksg_bv_matmult(ksg_bv V, ksg_mat A, ksg_bv W);

/* ---------------------------------------------------------------- Mat */
/* Local row block [row0,row0+n_local) of an N x N CSR ("AIJ") matrix: rowptr has
 * n_local+1 entries starting at 0, colidx holds GLOBAL column indices.  All
 * three arrays are host arrays and are copied.  With more than one rank the
 * matrix is split into the diagonal block and the ghost block and the gather
 * lists of the halo exchange are built (collective), as MatMPIAIJ does inside
 * PETSc.  Stands in for MatCreateAIJ + MatSetValues + MatAssembly as used by
 * the examples (ex2.c:50-66).  From a million entries on, the passes over the entries
 * (validation, column counts, value tables, row codes, the binned build) run on host
 * threads: KSG_HOST_THREADS of them, else OMP_NUM_THREADS, else the machine's, at most
 * 16; the operator built is the same for any number.  KSG_SETUP_TIMING=1 prints the
 * stages on stderr. */
int ksg_mat_create_csr(ksg_ctx ctx, int64_t n_local, int64_t N, int64_t row0,
                       const int64_t *rowptr, const int32_t *colidx, const double *val,
                       ksg_mat *A);
/* The same for an operator whose columns are distributed differently from its rows (a
 * rectangular matrix, as the singular value solvers take: MatSetSizes(A,m,n,M,N) of
 * src/svd/examples/tutorials/ex15.c): the vector x of ksg_mat_mult has nc_local entries, the
 * columns [col0, col0+nc_local) of the NC global ones. */
int ksg_mat_create_csr_rect(ksg_ctx ctx, int64_t n_local, int64_t N, int64_t row0,
                            int64_t nc_local, int64_t NC, int64_t col0,
                            const int64_t *rowptr, const int32_t *colidx, const double *val,
                            ksg_mat *A);
int ksg_mat_col_sizes(ksg_mat A, int64_t *nc_local, int64_t *NC, int64_t *col0);
/* The local rows back on the host with global column indices; rowptr has n_local+1
 * entries, colidx and val nnz_local (ksg_mat_sizes). */
int ksg_mat_get_csr(ksg_mat A, int64_t *rowptr, int32_t *colidx, double *val);
/* AT = A^T as a new operator with rows and columns exchanged, rank p holding the rows that
 * were its columns (MatTranspose as SVDSetUp builds svd->AT, svdsetup.c:139-148).
 * Collective. */
int ksg_mat_transpose(ksg_mat A, ksg_mat *AT);
int ksg_mat_destroy(ksg_mat A);
/* y = A*x on device vectors of n_local entries; collective when nranks > 1
 * (MatMult as called by STApply_Shift, shift.c:38). */
int ksg_mat_mult(ksg_mat A, const double *x_dev, double *y_dev);
/* *norm = ||A*x - kr*x + ki*xi||_2 over all ranks; xi may be NULL (then ki is ignored).
 * The residual of an eigenpair as EPSComputeResidualNorm_Private forms it
 * (src/eps/interface/epssolve.c:506-556: MatMult, VecAXPY, VecNorm) in one pass over the
 * matrix that stores nothing.  With nranks > 1 `work` (n_local entries, distinct from x and
 * xi) receives A*x first; with one rank it is not touched and may be NULL.  Collective. */
int ksg_mat_residual_norm(ksg_mat A, double kr, double ki, const double *x_dev, const double *xi_dev,
                          double *work_dev, double *norm);
int ksg_mat_sizes(ksg_mat A, int64_t *n_local, int64_t *N, int64_t *row0, int64_t *nnz_local);
/* Kernel-selection report: number of row tiles handled by the row-per-thread
 * stream path, the grouped (several lanes per row) path and the long-row path,
 * and the number of ghost entries received per multiply. */
int ksg_mat_info(ksg_mat A, int64_t info[4]);
/* Bytes one ksg_mat_mult moves on this rank: *executed by the model of the kernel variant that
 * runs (value table, 2-byte row offsets, packed columns, ghost block, halo pack: what the
 * statistics of KSG_EV_MATMULT are charged with), *csr_model by the plain CSR layout of PETSc's
 * AIJ (12 bytes per entry, 8 per row pointer, x and y once).  Either pointer may be NULL. */
int ksg_mat_traffic_model(ksg_mat A, double *executed, double *csr_model);
/* One line of text naming the kernel variant and data layout the operator's products use. */
int ksg_mat_describe(ksg_mat A, char *buf, size_t len);
/* Host-side check of the binned layout (KSG_OPT_BINS), no device needed and no counterpart in the
 * reference: y = A x for the CSR block given (has_diag: row r's entry in column r + dshift is the
 * operator's diagonal), formed as the library forms it -- the binned part and the diagonal through
 * the same walk of the same arrays as the two kernels, then the remaining entries row by row in
 * storage order.  info = {binned entries, runs, row blocks, value mode (0 one value, 1 one-byte
 * indices, 2 eight-byte values), entries left in the CSR, 1 if the diagonal is folded into the
 * reduce pass}; info[0] == 0: fewer than half of the entries fall into well filled cells and the
 * block would stay a plain CSR. */
int ksg_bins_host_check(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *colidx, const double *val,
                        int has_diag, int64_t dshift, const double *x, double *y, int64_t info[6]);
/* number of columns whose x entries the operator keeps packed (KSG_OPT_HOT_COLS), 0 if none */
int ksg_mat_hot_columns(ksg_mat A, int64_t *nhot);

/* --------------------------------------------------------- statistics */
/* Events, named after the reference's PetscLog events (bvfunc.c:78-93,
 * stfunc.c:75-82). */
enum {
  KSG_EV_DOTVEC = 0, KSG_EV_MULTVEC, KSG_EV_MULTINPLACE, KSG_EV_SCALE, KSG_EV_NORM,
  KSG_EV_COPY, KSG_EV_MATMULT, KSG_EV_MULT, KSG_EV_DOT, KSG_EV_VECOPS,
  KSG_EV_DOTVEC_FUSED,            /* dotvec requests answered by the preceding fused multvec: no bytes, no time */
  KSG_EV_COUNT
};
typedef struct {
  int64_t calls[KSG_EV_COUNT];
  double  bytes[KSG_EV_COUNT];   /* algorithmic bytes moved (model of the byte counts) */
  double  flops[KSG_EV_COUNT];
  double  ms[KSG_EV_COUNT];      /* device time from HIP events; 0 unless timing is on */
} ksg_stats;
int ksg_stats_enable_timing(ksg_ctx ctx, int on);
int ksg_stats_get(ksg_ctx ctx, ksg_stats *out);
int ksg_stats_reset(ksg_ctx ctx);

#ifdef __cplusplus
}
#endif
#endif
