/* Internal declarations shared by the translation units of libksgpu. */
#ifndef KSG_INTERNAL_H
#define KSG_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "ksgpu.h"

#define KSG_BLOCK      256          /* threads per workgroup (4 wave64)            */
#define KSG_MAXGRID    2048         /* 256 CUs x 8 workgroups: grid-stride beyond   */
#define KSG_QMAX       128          /* coefficients passed by value to one launch   */
#define KSG_RED_MAX    8192         /* doubles in the reduction buffers             */
#define KSG_DOT_GRID   256          /* row workgroups of the read-only reductions   */
#define KSG_MV_GRID    1024         /* row workgroups of the GEMV (read + write)    */

/* row tiles of the CSR product kernel */
#ifndef KSG_TILE_NNZ
#define KSG_TILE_NNZ  2048     /* products held in LDS per workgroup (16 KiB: 8 workgroups per CU) */
#endif
#ifndef KSG_TILE_ROWS
#define KSG_TILE_ROWS 512      /* rows per stream tile: a multiple of the workgroup size */
#endif
#define KSG_PC_ROWS    256     /* rows per workgroup step of the row-code kernel */

void ksg_set_error(const char *fmt, ...);

#define KSG_FAIL(code, ...) do { ksg_set_error(__VA_ARGS__); return (code); } while (0)
#define KSG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    ksg_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    return KSG_ERR_HIP; } } while (0)
#define KSG_CHK(call) do { int r_ = (call); if (r_) return r_; } while (0)

enum { KSG_COMM_NONE = 0, KSG_COMM_RCCL = 1, KSG_COMM_CALLBACKS = 2 };

/* ---- host threads for the set-up of an operator (the passes over its entries) ---- */

/* KSG_HOST_THREADS, else OMP_NUM_THREADS, else the machine's hardware threads; at most 16. */
int ksg_host_threads(void);

/* fn(t, T) for t = 0 .. T-1, each on a host thread of its own (t = 0 on the caller's).  Parts for
 * which no thread could be started run on the caller's thread afterwards. */
template <class F>
static inline void ksg_parallel(int T, F fn) {
  if (T <= 1) { fn(0, 1); return; }
  std::vector<std::thread> th;
  th.reserve((size_t)T - 1);
  int started = 1;
  try {
    for (; started < T; started++) { const int t = started; th.emplace_back([&fn, t, T] { fn(t, T); }); }
  } catch (...) { }
  fn(0, T);
  for (auto &x : th) x.join();
  for (int t = started; t < T; t++) fn(t, T);
}

/* First row of part t when rows 0 .. nrows-1 are cut into T parts of about as many entries each. */
static inline int64_t ksg_split_rows(const int64_t *rowptr, int64_t nrows, int t, int T) {
  if (t <= 0) return 0;
  if (t >= T) return nrows;
  const int64_t want = rowptr[0] + (rowptr[nrows] - rowptr[0]) / T * t;
  return (int64_t)(std::lower_bound(rowptr, rowptr + nrows + 1, want) - rowptr);
}

/* Allocator whose elements start out uninitialised: resize() of a vector of several GiB touches
 * no page, the threads that fill it do. */
template <class T>
struct ksg_noinit_alloc {
  typedef T value_type;
  ksg_noinit_alloc() noexcept {}
  template <class U> ksg_noinit_alloc(const ksg_noinit_alloc<U> &) noexcept {}
  T *allocate(size_t n) { return static_cast<T *>(::operator new(n * sizeof(T))); }
  void deallocate(T *p, size_t) noexcept { ::operator delete(p); }
  template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const ksg_noinit_alloc<U> &) const noexcept { return true; }
  template <class U> bool operator!=(const ksg_noinit_alloc<U> &) const noexcept { return false; }
};
template <class T> using ksg_hvec = std::vector<T, ksg_noinit_alloc<T>>;

/* KSG_SETUP_TIMING=1: the stages of an operator's set-up, in seconds, on stderr. */
struct ksg_phase_timer {
  const char *who;
  bool on;
  std::chrono::steady_clock::time_point t;
  explicit ksg_phase_timer(const char *w) : who(w), on(getenv("KSG_SETUP_TIMING") && atoi(getenv("KSG_SETUP_TIMING")) > 0),
                                           t(std::chrono::steady_clock::now()) {}
  void lap(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "%s: %-28s %.3f s\n", who, what, std::chrono::duration<double>(now - t).count());
    t = now;
  }
};

/* Up to 257 distinct 64-bit patterns and the number each was given: open addressing over 1024 slots. */
struct ksg_value_set {
  uint64_t key[1024];
  int num[1024];
  ksg_value_set() { for (int i = 0; i < 1024; i++) num[i] = -1; }
  static unsigned slot(uint64_t bits) { return (unsigned)((bits * 0x9E3779B97F4A7C15ull) >> 54); }
  int find(uint64_t bits) const {
    unsigned h = slot(bits);
    while (num[h] >= 0 && key[h] != bits) h = (h + 1) & 1023;
    return num[h];
  }
  /* the number of `bits`; a pattern not seen before gets `next` */
  int find_or_add(uint64_t bits, int next) {
    unsigned h = slot(bits);
    while (num[h] >= 0 && key[h] != bits) h = (h + 1) & 1023;
    if (num[h] < 0) { key[h] = bits; num[h] = next; }
    return num[h];
  }
};

struct ksg_timer { hipEvent_t e0, e1; int ev; };

/* Result of the reduction a fused multvec computed ahead of the request for it. */
struct ksg_spec {
  bool valid;
  uint64_t gen;              /* value of ksg_ctx_s::gen when it was produced */
  const double *X, *y;       /* first active column, updated vector */
  int64_t ld, n;
  int k;                     /* active columns; d_spec holds k+1 values */
};

/* A Gram-Schmidt update y <- beta*y + X q whose products are known (ksg_spec) but whose
 * result has not been written yet: it is applied together with the next update of the
 * same column, or on its own as soon as anything else needs device data. */
struct ksg_pending {
  bool active;
  const double *X;
  double *y;
  int64_t ld, n;
  int k;
  int stage;                 /* 1: one update pending, 2: a second one on top of it */
  double beta, beta2;
  double q[64], q2[64];      /* coefficients, already multiplied by alpha */
};

struct ksg_ctx_s {
  int device;
  hipStream_t stream;        /* compute */
  hipStream_t comm_stream;   /* halo pack + exchange */
  hipEvent_t ev_x_ready, ev_halo_done;
  std::vector<hipEvent_t> ev_peer;   /* one per round of a halo exchange in rounds: that round's pieces have arrived */
  double *d_red;             /* device reduction buffer, KSG_RED_MAX doubles */
  double *h_red;             /* pinned host mirror */
  double *d_partial;         /* per-workgroup partial sums */
  double *d_scratch;         /* this device's copy of the fixed scratch array (ksg_scratch_bind), not owned */
  size_t partial_cap;        /* doubles */
  double *d_q;               /* device copy of small host matrices (Q for multinplace) */
  size_t q_cap;
  double *d_work;            /* scratch panel for out-of-place fallbacks */
  size_t work_cap;
  double *h_q;               /* pinned staging of the small host matrices */
  size_t hq_cap;
  double *d_spec;            /* KSG_QMAX+1 doubles: reduction left by a fused multvec */
  double *d_batch;           /* KSG_RED_MAX doubles: results of posted reductions awaiting one all-reduce */
  int batch_open, batch_count;
  ksg_spec spec;
  uint64_t gen;              /* bumped by every operation that may write device data */
  int fuse_dots;             /* 0 disables the fused reduction (ksg_ctx_set_option) */
  int dot_grid, mv_grid;     /* row workgroups of the reduction / update kernels */
  ksg_pending pend;
  int lazy_update;           /* 0: never defer the store of a Gram-Schmidt update */
  int hot_cols;              /* most x entries a new operator may keep packed (0: none) */
  int bins;                  /* binned form of a new power-law operator: 0 = chosen by size, 1 = never, 2 = always */
  /* communicator */
  int comm_kind, rank, nranks, cb_on_device;
  void *rccl_comm;
  ksg_allreduce_fn cb_allreduce;
  ksg_exchange_fn cb_exchange;
  void *cb_user;
  /* statistics */
  ksg_stats stats;
  int timing;
  std::vector<ksg_timer> timers;   /* event pairs of operations not yet read back */
  std::vector<hipEvent_t> event_pool;
};

struct ksg_bv_s {
  ksg_ctx ctx;
  int64_t n, N, ld;
  int m, nc, l, k;
  double *d;
};

/* one CSR block on the device, with its row tiles */
struct ksg_csr_dev {
  int64_t nrows, nnz;
  int64_t *rowptr;   /* nrows+1 */
  int32_t *colidx;
  double *val;
  int32_t *tiles;    /* ntiles+1 row boundaries */
  int64_t *tile_p;   /* rowptr at those boundaries */
  uint16_t *rowoff;  /* nrows + ntiles entries: tile i holds, from index tiles[i] + i on, the
                      * entry offsets of its rows relative to the tile's first entry (one more
                      * than it has rows); 2 bytes per row instead of the 8 of rowptr */
  uint8_t *vidx;     /* nnz one-byte indices into vtab, when the matrix has at most 256 distinct values */
  double *vtab;      /* 256 doubles */
  int nvt;           /* 1 when vidx/vtab are there; 2: the table covers the off-diagonal entries (at most
                      * 255 distinct values) and the index 255 marks a diagonal entry, whose value is dvals[row] */
  double *dvals;     /* nrows diagonal values (nvt == 2) */
  int32_t dshift;    /* column index of row r's diagonal entry minus r (nvt == 2) */
  int64_t ntiles;
  int64_t n_stream, n_group, n_long;
  /* row-code form (banded operators with few distinct (column - row, value) pairs and short
   * rows -- stencils): one byte per entry naming the pair, stored by entry position across the
   * rows (in one or two 8-byte words per row), and the two tables */
  int pc_w;          /* longest row, 0: the block has no row-code form */
  int pc_chunk;      /* consecutive 256-row blocks that go to the same XCD */
  uint64_t *pc_words;  /* 1 or 2 words per row: byte 7 of the first is the row's length, the rest its codes */
  int32_t *pc_off;   /* 256 column offsets */
  double *pc_val;    /* 256 values */
};

/* What the binned part of a power-law block leaves: the entries of thin cells, as a CSR over only
 * the rows that have such entries (packed hot columns: ncols + slot). */
struct ksg_rest {
  ksg_csr_dev blk;
  int64_t ncols, nrows;
  int32_t *d_rows;           /* local row of every row of blk                  */
  int64_t nhot;
  std::vector<int32_t> hot_cols;
  int32_t *d_hot_idx;
  double *d_xhot;
};

/* The binned part of a power-law block (ksg_bins.hip): entries of well filled (row block, column
 * block) cells, held in the order the expand pass streams them (by column block) and, for their
 * rows, in the order the reduce pass streams the products (by row block). */
struct ksg_bins {
  int64_t nent;              /* slots: the sums the expand pass writes, one per stretch of entries of one row, the fill of the runs included; 0: the block has no binned part */
  int64_t nreal;             /* matrix entries of the binned part                               */
  int64_t nent1;             /* entries of the expand stream, fill included (a multiple of 64)  */
  int64_t nseg;              /* runs: the entries of one (row block, column block) cell, consecutive in both orders */
  int64_t nparts;            /* parts: the pieces of the runs that lie in one group of 64 entries */
  int64_t nbins;             /* row blocks (all of them: the reduce pass stores every row of y)  */
  int64_t ntasks;            /* workgroups of the expand pass                                    */
  int64_t nrows;             /* rows of the block                                                */
  int64_t dshift;            /* column of row r's diagonal entry minus r (dv != NULL)            */
  int W, H;                  /* columns per column block, most rows per row block                */
  int threads;               /* threads of an expand workgroup                                   */
  int vmode;                 /* 0: one value (vconst), 1: one-byte indices into vtab, 2: 8-byte values */
  double vconst;
  uint16_t *col16;           /* nent1: column within the column block (W: a fill entry, x = 0)   */
  uint8_t *vidx;             /* nent1 (vmode 1)                                                  */
  double *vtab;              /* 256 (vmode 1)                                                    */
  double *val;               /* nent1 (vmode 2)                                                  */
  uint32_t *gfirst;          /* per group of 64 entries: the part its first entry belongs to     */
  uint64_t *runmask;         /* per group: bit l = entry l opens a part (a run starts there)     */
  uint64_t *rowmask;         /* per group: bit l = entry l opens a stretch (another row, a part, a row of 16 lanes) */
  int64_t *O;                /* per part: the slot of its first stretch                          */
  int64_t *tasks;            /* 3 per task: column block, first group, end group                 */
  int64_t *task_stage;       /* host array, nstages + 1: the tasks of stage s of the column stages the block was built with are task_stage[s] .. task_stage[s+1] (NULL: one stage) */
  int nstages;
  uint16_t *rowloc;          /* nent: row within the row block of every slot, in the reduce order */
  int64_t *binp0, *binp1;    /* nbins each: the products of a row block, in the reduce order; the blocks with the most products first */
  int32_t *bin_r0, *bin_nr;  /* first row and row count of every row block                       */
  int64_t nlong, npieces;    /* rows longer than a unit of the reduce pass may be, and the pieces they are cut into */
  int32_t *long_rows, *long_first, *piece_row;   /* those rows; their pieces long_first[j] .. long_first[j+1]; the row of every piece */
  double *part;              /* npieces sums, added to their rows after the reduce pass             */
  double *dv;                /* nrows: the block's diagonal, folded into the reduce pass (NULL: the block has none, or not one per row) */
  double *xe;                /* nent + 64 sums                                                   */
};

struct ksg_mat_s {
  ksg_ctx ctx;
  int64_t n, N, row0, nnz;
  int64_t nc, NC, col0;      /* column layout: x has nc local entries, columns [col0,col0+nc) are local */
  std::vector<int64_t> cstarts;  /* first column of every rank, nranks+1 (empty with one rank) */
  std::vector<int32_t> ghosts;   /* global column of every ghost-buffer slot */
  ksg_csr_dev diag;          /* columns local (col - col0); entries of a packed column: nc + its slot */
  ksg_bins bins;             /* the binned part of the local block (power-law operators); diag is then empty */
  std::vector<ksg_rest> rest;      /* what the binned part left of the local block (none or one) */
  ksg_bins gbins;            /* the same for the ghost block: its x is the ghost buffer, its rows the list o_rows */
  std::vector<ksg_rest> grest;     /* what is left of a binned ghost block (d_rows: local rows of y) */
  /* packed copy of the x entries of the most referenced columns (power-law operators) */
  int64_t nhot;
  std::vector<int32_t> hot_cols; /* local column of every slot                  */
  int32_t *d_hot_idx;        /* the same on the device                         */
  double *d_xhot;            /* nhot doubles, refreshed by every product       */
  /* ghost block: a CSR over only the rows that have ghost entries (row i of it is local row
   * o_rows[i]); its column indices are slots of the ghost buffer, or -- for its own most
   * referenced columns -- ghost count + slot of a packed array, as in the diagonal block */
  int64_t no_rows;           /* number of such rows                            */
  int32_t *o_rows;           /* their local row indices                        */
  ksg_csr_dev ghost;
  int64_t o_nnz;
  int64_t g_nhot;
  std::vector<int32_t> g_hot_cols;
  int32_t *d_g_hot_idx;
  double *d_g_xhot;
  /* halo plan */
  int64_t nghost, nsend;
  double *d_ghost, *d_send;  /* device staging buffers                         */
  double *h_ghost, *h_send;  /* pinned host staging (callback transport)       */
  int32_t *d_send_idx;       /* local indices to gather into d_send            */
  std::vector<int64_t> sendoff, recvoff;   /* nranks+1 each                    */
  int halo_rounds;           /* rounds the halo travels in (0: one exchange); the ghost block's expand tasks start as their pieces arrive */
};

/* statistics / timing brackets */
int ksg_op_begin(ksg_ctx ctx, int ev);            /* applies a pending update first */
#define KSG_OPB_NOFLUSH 1
#define KSG_OPB_NOBUMP  2
int ksg_op_begin_ex(ksg_ctx ctx, int ev, int flags);
int ksg_flush_pending(ksg_ctx ctx);               /* ksg_bv.hip */
int ksg_op_end(ksg_ctx ctx, int ev, double bytes, double flops);

/* communicator primitives (ksg_comm.hip) */
int ksg_comm_allreduce(ksg_ctx ctx, double *d_buf, double *h_out, int count, int op);
int ksg_comm_exchange(ksg_ctx ctx, ksg_mat A);
/* the same exchange in A->halo_rounds rounds, piece s of every source's share of the ghost buffer in
 * round s (all pairs of ranks at once); ev_peer[s] is recorded on the communication stream when
 * round s has arrived */
int ksg_comm_exchange_rounds(ksg_ctx ctx, ksg_mat A);
int ksg_comm_allreduce_host(ksg_ctx ctx, double *h_buf, int count, int op);
int ksg_comm_exchange_host(ksg_ctx ctx, const double *send, const int64_t *sendoff,
                           double *recv, const int64_t *recvoff);

/* binned part of a power-law block (ksg_bins.hip) */
/* colstage (may be NULL): the stage in which every column of x becomes available, ascending
 * column boundaries paired with stage numbers (column c with bound[i] <= c < bound[i+1] has stage
 * stage[i]); the expand tasks are then grouped by the last stage their column block needs */
struct ksg_colstages { std::vector<int64_t> bound; std::vector<int> stage; int nstages; };
int ksg_bins_build(ksg_ctx c, ksg_bins *B, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *colidx,
                   const double *val, bool has_diag, int64_t dshift, std::vector<int32_t> &lrows, std::vector<int64_t> &lrp,
                   std::vector<int32_t> &lci, std::vector<double> &lva, const int32_t *rowmap, const ksg_colstages *colstage = NULL);
int ksg_bins_mult(const ksg_bins *B, const double *x, int64_t ncols, double *y, hipStream_t st, const int32_t *d_rowmap);
/* the two halves of ksg_bins_mult: the expand tasks of one stage (stage < 0: all), then the sums */
int ksg_bins_expand(const ksg_bins *B, const double *x, int64_t ncols, hipStream_t st, int stage);
int ksg_bins_finish(const ksg_bins *B, const double *x, double *y, hipStream_t st, const int32_t *d_rowmap);
double ksg_bins_bytes(const ksg_bins *B);
int ksg_bins_to_host(const ksg_bins *B, std::vector<int32_t> &rows, std::vector<int32_t> &cols, std::vector<double> &vals);
void ksg_bins_free(ksg_bins *B);
/* indices of the entries' values in a table of their distinct bit patterns; the number of distinct
 * values or -1 for more than `limit` (ksg_spmv.hip) */
int ksg_build_value_table(int64_t nnz, const double *val, const uint8_t *skip, size_t limit, std::vector<double> &tab,
                          ksg_hvec<uint8_t> &idx);

/* workspace helpers */
int ksg_scratch_bind(ksg_ctx ctx);                /* ksg_bv.hip; the context's device must be current */
int ksg_ensure_partial(ksg_ctx ctx, size_t doubles);
int ksg_ensure_q(ksg_ctx ctx, size_t doubles);
int ksg_ensure_work(ksg_ctx ctx, size_t doubles);

/* reductions used by several files: sum over `nb` partials per column */
int ksg_launch_dotvec(ksg_ctx ctx, const double *X, int64_t ld, int k, const double *y,
                      int64_t n, double *d_out);
/* out[col] = sum of partial[col*nb .. col*nb+nb) in a fixed order */
int ksg_launch_reduce_sum(ksg_ctx ctx, const double *partial, int nb, int ncols, double *d_out);
int ksg_launch_sumsq(ksg_ctx ctx, const double *X, int64_t ld, int ncols, int64_t n, double *d_out);

static inline int ksg_grid_for(int64_t work_items) {
  int64_t g = (work_items + KSG_BLOCK - 1) / KSG_BLOCK;
  if (g < 1) g = 1;
  if (g > KSG_MAXGRID) g = KSG_MAXGRID;
  return (int)g;
}

#endif
