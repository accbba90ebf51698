/* CSR operator: y = A x, the MatMult that STApply_Shift issues once per Krylov
 * step (src/sys/classes/st/impls/shift/shift.c:38).  In the reference this is
 * PETSc's MatMult_SeqAIJ / MatMult_MPIAIJ (or cuSPARSE); here it is a
 * hand-written gfx950 kernel family selected per row tile by nnz/row:
 *
 *   stream tile  : up to KSG_TILE_ROWS consecutive rows whose entries fit one LDS
 *                  tile.  The workgroup streams val/colidx of the whole tile
 *                  with coalesced loads, gathers x through L2, parks the
 *                  products in LDS, then
 *                    - many short rows: one lane per row sums its products in
 *                      storage order (bit-identical to a sequential CSR loop);
 *                    - few longer rows: a power-of-two group of lanes per row
 *                      sums strided and combines with wave shuffles;
 *   long row     : a row longer than one LDS tile gets the whole workgroup.
 *
 * With several ranks the matrix is split like PETSc's MPIAIJ into the diagonal
 * block (local columns) and the ghost block; the ghost values are gathered,
 * exchanged on the communication stream while the diagonal block is multiplied
 * on the compute stream, and then the ghost block is added. */
#include <algorithm>
#include <stdlib.h>

#include "ksg_internal.h"


__device__ __forceinline__ double wave_sum_width(double v, int width) {
  for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ void fence_d(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void fence_i(int32_t &v) { asm volatile("" : "+v"(v)); }

#define KSG_TILE_PER_THREAD (KSG_TILE_NNZ / KSG_BLOCK)    /* entries of a tile per thread: 8 */

/* RES = 2: y[rowmap[row]] += (A x)_row -- the ghost block of a distributed operator, whose rows
 * are a subset of the local rows, added to what the diagonal block left in y.
 * RES = 0: y = A x.  RES = 1: nothing is stored; every row's sum goes into the residual
 * r = (A x)_row - kr x_row + ki xi_row (xi may be NULL) and the workgroup leaves the sum
 * of r^2 over its rows in partial[blockIdx.x] -- the norm EPSComputeResidualNorm_Private
 * (src/eps/interface/epssolve.c:506-556) needs, without materialising A x. */
template <int RES>
__device__ __forceinline__ void spmv_emit(double s, int row, const double *__restrict__ x, double *__restrict__ y,
                                          double kr, double ki, const double *__restrict__ xi, double &acc, bool nt,
                                          const int32_t *__restrict__ rowmap) {
  if (RES == 2) {
    y[rowmap[row]] += s;               /* every row of the block belongs to one workgroup, one lane */
  } else if (RES) {
    double r = fma(-kr, x[row], s);
    if (xi) r = fma(ki, xi[row], r);
    acc = fma(r, r, acc);
  } else if (nt) {
    __builtin_nontemporal_store(s, &y[row]);
  } else {
    y[row] = s;
  }
}

/* The x entry a matrix entry multiplies: with HOT, column indices from nx on are slots of the
 * packed array xh (the most referenced columns of a power-law operator, dense so that they
 * stay in L2) */
template <int HOT>
__device__ __forceinline__ double x_entry(const double *__restrict__ x, const double *__restrict__ xh, int32_t nx, int32_t ci) {
  if (HOT) {
    const double *b = ci >= nx ? xh : x;
    return b[ci >= nx ? ci - nx : ci];
  }
  return x[ci];
}

/* VT = 1: the operator has at most 256 distinct values (stencils, unweighted graphs): the entry
 * stream carries a one-byte index per entry instead of the eight-byte value, and the value comes
 * from a table of at most 2 KiB that stays in the CU's L1 -- the same values, 7 bytes per entry
 * less to move.  VT = 2: the same for an operator whose off-diagonal entries have at most 255
 * distinct values while the diagonal is arbitrary (graph Laplacians: -1 off the diagonal, the
 * degree on it; shifted stencils): the index 255 marks a diagonal entry, whose value is read from
 * a dense vector of the diagonal at its column (= its row: neighbouring rows, neighbouring loads). */
template <int RES, int HOT, int VT>
__global__ __launch_bounds__(KSG_BLOCK) void k_spmv_tiles(const int32_t *__restrict__ tiles,
                                                           const int64_t *__restrict__ tile_p,
                                                           const uint16_t *__restrict__ rowoff,
                                                           const int32_t *__restrict__ colidx,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ x,
                                                           double *__restrict__ y, double kr, double ki,
                                                           const double *__restrict__ xi,
                                                           double *__restrict__ partial,
                                                           const double *__restrict__ xh, int32_t nx,
                                                           const uint8_t *__restrict__ vidx,
                                                           const double *__restrict__ vtab,
                                                           const int32_t *__restrict__ rowmap,
                                                           const double *__restrict__ dvals, int32_t dshift) {
  __shared__ double prod[KSG_TILE_NNZ];
  __shared__ double red[KSG_BLOCK / 64];
  double acc = 0.0;
  /* row and entry ranges of the tile come from two independent tables (no dependent
   * load of rowptr[tiles[b]] before the streaming can start) */
  const int r0 = tiles[blockIdx.x], r1 = tiles[blockIdx.x + 1];
  const int64_t p0 = tile_p[blockIdx.x], p1 = tile_p[blockIdx.x + 1];
  const int nrows = r1 - r0;
  const int64_t nnz = p1 - p0;

  if (nnz > KSG_TILE_NNZ) {
    /* long row (nrows == 1): the whole workgroup strides over it */
    double a = 0.0;
    int64_t p = p0 + threadIdx.x;
    /* eight independent entry/gather pairs in flight per thread */
    for (; p + 7 * KSG_BLOCK < p1; p += 8 * KSG_BLOCK) {
      int32_t ci[8];
      double v[8], xv[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        ci[i] = __builtin_nontemporal_load(colidx + p + i * KSG_BLOCK);
        if (VT) {
          const int t = __builtin_nontemporal_load(vidx + p + i * KSG_BLOCK);
          v[i] = vtab[t];
          if (VT == 2 && t == 255) v[i] = dvals[ci[i] - dshift];
        } else v[i] = __builtin_nontemporal_load(val + p + i * KSG_BLOCK);
      }
#pragma unroll
      for (int i = 0; i < 8; i++) fence_i(ci[i]);
#pragma unroll
      for (int i = 0; i < 8; i++) xv[i] = x_entry<HOT>(x, xh, nx, ci[i]);
#pragma unroll
      for (int i = 0; i < 8; i++) fence_d(xv[i]);
#pragma unroll
      for (int i = 0; i < 8; i++) a = fma(v[i], xv[i], a);
    }
    for (; p < p1; p += KSG_BLOCK) {
      const int32_t c1 = __builtin_nontemporal_load(colidx + p);
      double v1;
      if (VT) {
        const int t = __builtin_nontemporal_load(vidx + p);
        v1 = vtab[t];
        if (VT == 2 && t == 255) v1 = dvals[c1 - dshift];
      } else v1 = __builtin_nontemporal_load(val + p);
      a = fma(v1, x_entry<HOT>(x, xh, nx, c1), a);
    }
    a = wave_sum_width(a, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = red[0];
#pragma unroll
      for (int w = 1; w < KSG_BLOCK / 64; w++) s += red[w];
      spmv_emit<RES>(s, r0, x, y, kr, ki, xi, acc, false, rowmap);
      if (RES == 1) partial[blockIdx.x] = acc;
    }
    return;
  }

  /* this thread's row bounds, requested before the entry stream so that their latency
   * hides behind it */
  constexpr int RPT = KSG_TILE_ROWS / KSG_BLOCK;          /* rows per thread */
  const uint16_t *ro = rowoff + r0 + blockIdx.x;        /* this tile's row offsets, relative to p0 */
  int rb[RPT], re[RPT];
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const int lr = threadIdx.x + j * KSG_BLOCK;
    rb[j] = re[j] = 0;
    if (lr < nrows) { rb[j] = ro[lr]; re[j] = ro[lr + 1]; }
  }

  /* stream the tile's entries: val/colidx are read once (coalesced, non-temporal, all of
   * a thread's requests in flight together), x is gathered through L2 */
  {
    const int n_ = (int)nnz;
    int32_t ci[KSG_TILE_PER_THREAD], vi[VT ? KSG_TILE_PER_THREAD : 1];
    double v[KSG_TILE_PER_THREAD], xv[KSG_TILE_PER_THREAD];
#pragma unroll
    for (int i = 0; i < KSG_TILE_PER_THREAD; i++) {
      const int t = threadIdx.x + i * KSG_BLOCK;
      ci[i] = 0; v[i] = 0.0;
      if (VT) vi[i] = 0;
      if (t < n_) {
        ci[i] = __builtin_nontemporal_load(colidx + p0 + t);
        if (VT) vi[i] = __builtin_nontemporal_load(vidx + p0 + t);
        else v[i] = __builtin_nontemporal_load(val + p0 + t);
      }
    }
#pragma unroll
    for (int i = 0; i < KSG_TILE_PER_THREAD; i++) fence_i(ci[i]);
    if (VT) {
#pragma unroll
      for (int i = 0; i < KSG_TILE_PER_THREAD; i++) fence_i(vi[i]);
    }
#pragma unroll
    for (int i = 0; i < KSG_TILE_PER_THREAD; i++) {
      const int t = threadIdx.x + i * KSG_BLOCK;
      xv[i] = 0.0;
      if (t < n_) xv[i] = x_entry<HOT>(x, xh, nx, ci[i]);
    }
    if (VT) {
#pragma unroll
      for (int i = 0; i < KSG_TILE_PER_THREAD; i++) v[i] = vtab[vi[i]];
      if (VT == 2) {
#pragma unroll
        for (int i = 0; i < KSG_TILE_PER_THREAD; i++) if (vi[i] == 255) v[i] = dvals[ci[i] - dshift];
      }
#pragma unroll
      for (int i = 0; i < KSG_TILE_PER_THREAD; i++) fence_d(v[i]);
    }
#pragma unroll
    for (int i = 0; i < KSG_TILE_PER_THREAD; i++) fence_d(xv[i]);
#pragma unroll
    for (int i = 0; i < KSG_TILE_PER_THREAD; i++) {
      const int t = threadIdx.x + i * KSG_BLOCK;
      if (t < n_) prod[t] = v[i] * xv[i];
    }
  }
  __syncthreads();

  if (nrows > KSG_BLOCK / 2 || nnz <= 8 * (int64_t)nrows) {   /* many rows, or few but short ones */
    /* one lane per row, products added in storage order */
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int lr = threadIdx.x + j * KSG_BLOCK;
      if (lr < nrows) {
        const int b = rb[j], e = re[j];
        double s = 0.0;
        for (int t = b; t < e; t++) s += prod[t];
        spmv_emit<RES>(s, r0 + lr, x, y, kr, ki, xi, acc, true, rowmap);
      }
    }
  } else {
    /* G lanes per row, G = largest power of two with nrows*G <= 256 (at most 64) */
    int G = 2;
    while (G < 64 && nrows * (G << 1) <= KSG_BLOCK) G <<= 1;
    const int lr = threadIdx.x / G, g = threadIdx.x % G;
    double s = 0.0;
    if (lr < nrows) {
      const int b = ro[lr], e = ro[lr + 1];
      for (int t = b + g; t < e; t += G) s += prod[t];
    }
    s = wave_sum_width(s, G);
    if (lr < nrows && g == 0) spmv_emit<RES>(s, r0 + lr, x, y, kr, ki, xi, acc, false, rowmap);
  }
  if (RES == 1) {
    /* sum of the squared residuals of the tile, in a fixed order */
    __syncthreads();                                   /* prod is dead: red is free again */
    acc = wave_sum_width(acc, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = red[0];
#pragma unroll
      for (int w = 1; w < KSG_BLOCK / 64; w++) t += red[w];
      partial[blockIdx.x] = t;
    }
  }
}

/* sum over the rows of (u - kr x + ki xi)^2, one partial per workgroup (the residual norm
 * when A x had to be completed by a halo exchange first) */
__global__ __launch_bounds__(KSG_BLOCK) void k_resnorm(const double *__restrict__ u, const double *__restrict__ x,
                                                        const double *__restrict__ xi, double kr, double ki,
                                                        int64_t n, double *__restrict__ partial) {
  __shared__ double red[KSG_BLOCK / 64];
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; i < n; i += stride) {
    double r = fma(-kr, x[i], u[i]);
    if (xi) r = fma(ki, xi[i], r);
    acc = fma(r, r, acc);
  }
  acc = wave_sum_width(acc, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = red[0];
#pragma unroll
    for (int w = 1; w < KSG_BLOCK / 64; w++) t += red[w];
    partial[blockIdx.x] = t;
  }
}

/* Y(:,0:k) = A X(:,0:k) in one pass over the matrix (BVMatMult, bvops.c:578-600, as a sparse
 * matrix times a block of vectors).  The entries of a tile are parked in LDS once and reused
 * for every column, in chunks of KC columns per pass over the tile.  Every column gets exactly
 * the arithmetic of k_spmv_tiles (the same products, added in the same order), so the result
 * equals k separate products bit for bit. */
template <int KC>
__global__ __launch_bounds__(KSG_BLOCK) void k_spmm_tiles(const int32_t *__restrict__ tiles,
                                                           const int64_t *__restrict__ tile_p,
                                                           const int64_t *__restrict__ rowptr,
                                                           const int32_t *__restrict__ colidx,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ X, int64_t ldx,
                                                           double *__restrict__ Y, int64_t ldy, int k) {
  /* products and sums are rounded separately, as they are where k_spmv_tiles parks the
   * products in LDS: no contraction into fused multiply-adds here */
#pragma clang fp contract(off)
  __shared__ double sv[KSG_TILE_NNZ];
  __shared__ int32_t sc[KSG_TILE_NNZ];
  __shared__ double red[KSG_BLOCK / 64];
  const int r0 = tiles[blockIdx.x], r1 = tiles[blockIdx.x + 1];
  const int64_t p0 = tile_p[blockIdx.x], p1 = tile_p[blockIdx.x + 1];
  const int nrows = r1 - r0;
  const int64_t nnz = p1 - p0;

  if (nnz > KSG_TILE_NNZ) {
    /* long row: the workgroup strides over it once per column */
    for (int j = 0; j < k; j++) {
      const double *x = X + (int64_t)j * ldx;
      double a = 0.0;
      for (int64_t p = p0 + threadIdx.x; p < p1; p += KSG_BLOCK)
        a = fma(__builtin_nontemporal_load(val + p), x[__builtin_nontemporal_load(colidx + p)], a);
      a = wave_sum_width(a, 64);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = red[0];
#pragma unroll
        for (int w = 1; w < KSG_BLOCK / 64; w++) t += red[w];
        Y[r0 + (int64_t)j * ldy] = t;
      }
    }
    return;
  }

  for (int t = threadIdx.x; t < (int)nnz; t += KSG_BLOCK) {
    sv[t] = __builtin_nontemporal_load(val + p0 + t);
    sc[t] = __builtin_nontemporal_load(colidx + p0 + t);
  }
  __syncthreads();

  if (nrows > KSG_BLOCK / 2 || nnz <= 8 * (int64_t)nrows) {   /* many rows, or few but short ones */
    /* one lane per row, KC columns at a time */
    constexpr int RPT = KSG_TILE_ROWS / KSG_BLOCK;
    int b[RPT], e[RPT];
#pragma unroll
    for (int i = 0; i < RPT; i++) {
      const int lr = threadIdx.x + i * KSG_BLOCK;
      b[i] = e[i] = 0;
      if (lr < nrows) { b[i] = (int)(rowptr[r0 + lr] - p0); e[i] = (int)(rowptr[r0 + lr + 1] - p0); }
    }
    for (int j0 = 0; j0 < k; j0 += KC) {
      const int kc = min(KC, k - j0);
      const double *x = X + (int64_t)j0 * ldx;
#pragma unroll
      for (int i = 0; i < RPT; i++) {
        const int lr = threadIdx.x + i * KSG_BLOCK;
        if (lr >= nrows) continue;
        double acc[KC];
#pragma unroll
        for (int q = 0; q < KC; q++) acc[q] = 0.0;
        if (kc == KC) {
          for (int t = b[i]; t < e[i]; t++) {
            const double a = sv[t];
            const double *xc = x + sc[t];
#pragma unroll
            for (int q = 0; q < KC; q++) acc[q] = acc[q] + a * xc[(int64_t)q * ldx];
          }
        } else {
          for (int t = b[i]; t < e[i]; t++) {
            const double a = sv[t];
            const double *xc = x + sc[t];
#pragma unroll
            for (int q = 0; q < KC; q++) if (q < kc) acc[q] = acc[q] + a * xc[(int64_t)q * ldx];
          }
        }
#pragma unroll
        for (int q = 0; q < KC; q++) if (q < kc) __builtin_nontemporal_store(acc[q], &Y[r0 + lr + (int64_t)(j0 + q) * ldy]);
      }
    }
  } else {
    /* G lanes per row, one column at a time */
    int G = 2;
    while (G < 64 && nrows * (G << 1) <= KSG_BLOCK) G <<= 1;
    const int lr = threadIdx.x / G, g = threadIdx.x % G;
    int b = 0, e = 0;
    if (lr < nrows) { b = (int)(rowptr[r0 + lr] - p0); e = (int)(rowptr[r0 + lr + 1] - p0); }
    for (int j = 0; j < k; j++) {
      const double *x = X + (int64_t)j * ldx;
      double s = 0.0;
      for (int t = b + g; t < e; t += G) s = s + sv[t] * x[sc[t]];
      s = wave_sum_width(s, G);
      if (lr < nrows && g == 0) Y[r0 + lr + (int64_t)j * ldy] = s;
    }
  }
}

__global__ __launch_bounds__(KSG_BLOCK) void k_gather(const double *__restrict__ x, const int32_t *__restrict__ idx,
                                                       double *__restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; i < n; i += stride) out[i] = x[idx[i]];
}

/* ------------------------------------------------------------------------ */
/* row-code kernel: stencils and other banded operators                        */
/* ------------------------------------------------------------------------ */
/* An operator whose entries take at most 256 distinct (column - row, value) pairs, with short
 * rows, needs neither column indices nor values in its entry stream: one byte per entry names
 * the pair.  One lane owns a row; a row's codes and its length travel in one 8-byte word (two
 * for rows of 8 to 15 entries), the words of consecutive rows are consecutive, and entry k of 64
 * consecutive rows gathers 64 consecutive x entries -- coalesced loads instead of the scattered
 * gather of the tile kernel, no LDS staging.  The products are rounded one by one and added in storage order: the bits of
 * the sequential CSR loop, like the tile kernel's one-lane-per-row path.
 * Workgroup b takes the 256-row block xcd*C + j of its group of 8*C blocks (xcd = b % 8, the
 * XCD a workgroup lands on under round-robin dispatch; j = (b / 8) % C): C consecutive blocks
 * share an XCD, so the x lines that neighbouring grid lines have in common are fetched into one
 * L2 instead of several.  C is chosen at creation from the operator's offsets. */
/* The codes of a row travel in NW 8-byte words (NW = 1: up to 7 entries, NW = 2: up to 15):
 * byte 7 of the first word is the row's length, the other bytes are its codes in storage order;
 * word j of row r is at words[j*nrows + r], so a wave reads 512 consecutive bytes per word. */
template <int RES, int NW>
__global__ __launch_bounds__(KSG_PC_ROWS) void k_spmv_rowcode(const uint64_t *__restrict__ words,
                                                               const int32_t *__restrict__ otab, const double *__restrict__ vtab,
                                                               int64_t nrows, int64_t nblocks, int C, const double *__restrict__ x,
                                                               double *__restrict__ y, double kr, double ki,
                                                               const double *__restrict__ xi, double *__restrict__ partial) {
#pragma clang fp contract(off)
  constexpr int WT = 8 * NW - 1;
  __shared__ int32_t so[256];
  __shared__ double sv[256];
  __shared__ double red[KSG_PC_ROWS / 64];
  so[threadIdx.x] = otab[threadIdx.x];
  sv[threadIdx.x] = vtab[threadIdx.x];
  const int64_t per = 8 * (int64_t)C;
  __syncthreads();
  double acc = 0.0;
  /* a workgroup walks the blocks b, b + grid, ...: the grid is a multiple of 8 C, so all its blocks
   * have the same place in their group and the two tables are staged once */
  for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int64_t within = b % per;
    const int64_t rb = (b / per) * per + (within % 8) * C + within / 8;
    const int64_t row = rb * KSG_PC_ROWS + threadIdx.x;
    if (row < nrows) {
      uint64_t w[NW];
#pragma unroll
      for (int j = 0; j < NW; j++) w[j] = __builtin_nontemporal_load(words + (int64_t)j * nrows + row);
      const int n_ = (int)(w[0] >> 56);
      int c[WT];
#pragma unroll
      for (int k = 0; k < WT; k++) c[k] = k < 7 ? (int)((w[0] >> (8 * k)) & 255) : (int)((w[NW - 1] >> (8 * (k - 7))) & 255);
      double xv[WT];
#pragma unroll
      for (int k = 0; k < WT; k++) xv[k] = (k < n_) ? x[row + so[c[k]]] : 0.0;
#pragma unroll
      for (int k = 0; k < WT; k++) fence_d(xv[k]);
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < WT; k++)
        if (k < n_) { const double p = sv[c[k]] * xv[k]; s = s + p; }
      if (RES) {
        double r = fma(-kr, x[row], s);
        if (xi) r = fma(ki, xi[row], r);
        acc = fma(r, r, acc);
      } else {
        __builtin_nontemporal_store(s, &y[row]);
      }
    }
  }
  if (RES) {
    acc = wave_sum_width(acc, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = red[0];
#pragma unroll
      for (int w2 = 1; w2 < KSG_PC_ROWS / 64; w2++) t += red[w2];
      partial[blockIdx.x] = t;
    }
  }
}

static inline int64_t rowcode_blocks(const ksg_csr_dev &D) {
  const int64_t nb = (D.nrows + KSG_PC_ROWS - 1) / KSG_PC_ROWS, per = 8 * (int64_t)D.pc_chunk;
  return (nb + per - 1) / per * per;              /* whole groups: the blocks past the last row do nothing */
}
/* workgroups of a launch: whole groups of 8 C, at most KSG_RC_GRID (default 16384: 4096^2 0.107 -> 0.097 ms, 512^3 0.965 -> 0.924 ms per product) */
static inline int64_t rowcode_grid(const ksg_csr_dev &D) {
  static int64_t cap = 0;
  if (!cap) { const char *e = getenv("KSG_RC_GRID"); cap = e ? atoll(e) : 0; if (cap < 8) cap = 16384; }
  const int64_t nb = rowcode_blocks(D), per = 8 * (int64_t)D.pc_chunk;
  int64_t g = cap / per * per;
  if (g < per) g = per;
  return nb < g ? nb : g;
}

template <int RES>
static void launch_rowcode(hipStream_t st, const ksg_csr_dev &D, const double *x, double *y, double kr, double ki,
                           const double *xi, double *partial) {
  const dim3 grid((unsigned)rowcode_grid(D)), block(KSG_PC_ROWS);
#define KSG_RC(NW) hipLaunchKernelGGL((k_spmv_rowcode<RES, NW>), grid, block, 0, st, (const uint64_t *)D.pc_words, \
                   (const int32_t *)D.pc_off, (const double *)D.pc_val, D.nrows, rowcode_blocks(D), D.pc_chunk, x, y, kr, ki, xi, partial)
  if (D.pc_w <= 7) KSG_RC(1); else KSG_RC(2);
#undef KSG_RC
}

/* One launch of the tile kernel over the block D: xh != NULL selects the variant whose column
 * indices from nx on are slots of the packed array xh, D.nvt the one-byte value indices. */
template <int RES>
static void launch_tiles(hipStream_t st, const ksg_csr_dev &D, const double *x, double *y, double kr, double ki,
                         const double *xi, double *partial, const double *xh, int32_t nx, const int32_t *rowmap) {
  const dim3 grid((unsigned)D.ntiles), block(KSG_BLOCK);
#define KSG_TILES(HOT, VT) hipLaunchKernelGGL((k_spmv_tiles<RES, HOT, VT>), grid, block, 0, st, D.tiles, D.tile_p, D.rowoff, D.colidx, \
                           D.val, x, y, kr, ki, xi, partial, xh, nx, (const uint8_t *)D.vidx, (const double *)D.vtab, rowmap, \
                           (const double *)D.dvals, D.dshift)
  if (xh) { if (D.nvt == 2) KSG_TILES(1, 2); else if (D.nvt) KSG_TILES(1, 1); else KSG_TILES(1, 0); }
  else    { if (D.nvt == 2) KSG_TILES(0, 2); else if (D.nvt) KSG_TILES(0, 1); else KSG_TILES(0, 0); }
#undef KSG_TILES
}

extern "C" int ksg_mat_info(ksg_mat A, int64_t info[4]) {
  if (!A || !info) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_info: NULL argument");
  info[0] = A->diag.n_stream; info[1] = A->diag.n_group; info[2] = A->diag.n_long; info[3] = A->nghost;
  return KSG_OK;
}

/* y = A x without statistics brackets (shared with ksg_bv_matmult) */
static int mat_mult_raw(ksg_mat A, const double *x, double *y) {
  ksg_ctx c = A->ctx;
  if (A->n == 0 && c->nranks == 1) return KSG_OK;
  const bool halo = c->nranks > 1;
  if (halo) {
    /* gather + exchange on the communication stream, behind the producer of x */
    KSG_HIP(hipEventRecord(c->ev_x_ready, c->stream));
    KSG_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_x_ready, 0));
    if (A->nsend)
      hipLaunchKernelGGL(k_gather, dim3(ksg_grid_for(A->nsend)), dim3(KSG_BLOCK), 0, c->comm_stream, x, A->d_send_idx, A->d_send, A->nsend);
  }
  if (A->bins.nent) {
    /* the binned part and the diagonal store y; the remaining entries are added below */
    KSG_CHK(ksg_bins_mult(&A->bins, x, A->nc, y, c->stream, NULL));
  } else if (A->diag.pc_w) {
    launch_rowcode<0>(c->stream, A->diag, x, y, 0.0, 0.0, NULL, NULL);
  } else if (A->diag.ntiles) {
    if (A->nhot) hipLaunchKernelGGL(k_gather, dim3(ksg_grid_for(A->nhot)), dim3(KSG_BLOCK), 0, c->stream, x, A->d_hot_idx, A->d_xhot, A->nhot);
    launch_tiles<0>(c->stream, A->diag, x, y, 0.0, 0.0, NULL, NULL, A->nhot ? A->d_xhot : NULL, (int32_t)A->nc, NULL);
  } else if (A->n) {
    KSG_HIP(hipMemsetAsync(y, 0, (size_t)A->n * sizeof(double), c->stream));     /* no local columns at all */
  }
  /* what the binned part left, added into the rows it touches */
  for (const ksg_rest &cg : A->rest) {
    const double *xg = x;
    if (cg.nhot) hipLaunchKernelGGL(k_gather, dim3(ksg_grid_for(cg.nhot)), dim3(KSG_BLOCK), 0, c->stream, xg, cg.d_hot_idx, cg.d_xhot, cg.nhot);
    launch_tiles<2>(c->stream, cg.blk, xg, y, 0.0, 0.0, NULL, NULL, cg.nhot ? cg.d_xhot : NULL, (int32_t)cg.ncols, cg.d_rows);
  }
  KSG_HIP(hipGetLastError());
  if (halo && A->halo_rounds && A->gbins.nent) {
    /* a ghost block that outweighs the local one (power-law operators): the halo travels in rounds, a
     * piece of every source's share per round, and the expand tasks whose slices of the ghost buffer
     * are complete start behind their round while the later rounds are still under way; the sums wait
     * for all */
    KSG_CHK(ksg_comm_exchange_rounds(c, A));
    for (int st = 0; st < A->halo_rounds; st++) {
      KSG_HIP(hipStreamWaitEvent(c->stream, c->ev_peer[(size_t)st], 0));
      KSG_CHK(ksg_bins_expand(&A->gbins, A->d_ghost, A->nghost, c->stream, st));
    }
    KSG_CHK(ksg_bins_finish(&A->gbins, A->d_ghost, y, c->stream, A->o_rows));
    for (const ksg_rest &cg : A->grest) {
      if (cg.nhot) hipLaunchKernelGGL(k_gather, dim3(ksg_grid_for(cg.nhot)), dim3(KSG_BLOCK), 0, c->stream, (const double *)A->d_ghost, cg.d_hot_idx, cg.d_xhot, cg.nhot);
      launch_tiles<2>(c->stream, cg.blk, A->d_ghost, y, 0.0, 0.0, NULL, NULL, cg.nhot ? cg.d_xhot : NULL, (int32_t)cg.ncols, cg.d_rows);
    }
    KSG_HIP(hipGetLastError());
  } else if (halo) {
    KSG_CHK(ksg_comm_exchange(c, A));
    KSG_HIP(hipEventRecord(c->ev_halo_done, c->comm_stream));
    KSG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo_done, 0));
    if (A->gbins.nent) {
      /* the ghost block in the binned form: LDS slices of the ghost buffer, sums added into the rows the block touches */
      KSG_CHK(ksg_bins_mult(&A->gbins, A->d_ghost, A->nghost, y, c->stream, A->o_rows));
      for (const ksg_rest &cg : A->grest) {
        if (cg.nhot) hipLaunchKernelGGL(k_gather, dim3(ksg_grid_for(cg.nhot)), dim3(KSG_BLOCK), 0, c->stream, (const double *)A->d_ghost, cg.d_hot_idx, cg.d_xhot, cg.nhot);
        launch_tiles<2>(c->stream, cg.blk, A->d_ghost, y, 0.0, 0.0, NULL, NULL, cg.nhot ? cg.d_xhot : NULL, (int32_t)cg.ncols, cg.d_rows);
      }
    } else if (A->no_rows) {
      /* the ghost block through the same tile kernel (one lane per row, several lanes per row,
       * whole workgroup for a long row), adding into the rows it touches */
      if (A->g_nhot) hipLaunchKernelGGL(k_gather, dim3(ksg_grid_for(A->g_nhot)), dim3(KSG_BLOCK), 0, c->stream, (const double *)A->d_ghost, A->d_g_hot_idx, A->d_g_xhot, A->g_nhot);
      launch_tiles<2>(c->stream, A->ghost, A->d_ghost, y, 0.0, 0.0, NULL, NULL, A->g_nhot ? A->d_g_xhot : NULL, (int32_t)A->nghost, A->o_rows);
    }
    KSG_HIP(hipGetLastError());
  }
  return KSG_OK;
}

/* Bytes one product moves, by the model of the kernel variant that runs (what the statistics are
 * charged with): the entry stream is 12 bytes per entry, or 5 when the values travel as one-byte
 * indices into the value table; the rows cost 2 bytes of tile-relative offset each instead of the
 * 8 of a row pointer, plus 12 bytes of tile tables per tile; x is counted once, y is written once;
 * packed hot columns add the gather that refreshes them (index, read, write); the ghost block adds
 * its own entry stream, row tables and the read-modify-write of the rows it touches; the halo adds
 * the pack kernel (index, read, write) and the received entries. */
static double block_stream_bytes(const ksg_csr_dev &D) {
  if (D.pc_w) return (D.pc_w <= 7 ? 8.0 : 16.0) * (double)D.nrows + 3072.0;  /* the rows' code words */
  return (D.nvt ? 5.0 : 12.0) * (double)D.nnz + (D.nvt == 2 ? 8.0 * (double)D.nrows : 0.0) +
         2.0 * (double)(D.nrows + D.ntiles) + 12.0 * (double)(D.ntiles + 1);
}
static inline double spmv_bytes(ksg_mat A) {
  double b = block_stream_bytes(A->diag) + 8.0 * (double)A->nc + 8.0 * (double)A->n + 20.0 * (double)A->nhot;
  for (const ksg_rest &cg : A->rest)         /* entry stream, row list, read-modify-write of the rows touched, packing */
    b += block_stream_bytes(cg.blk) + 20.0 * (double)cg.nrows + 20.0 * (double)cg.nhot;
  b += ksg_bins_bytes(&A->bins);
  if (A->ctx->nranks > 1) {
    b += (A->no_rows && !A->gbins.nent ? block_stream_bytes(A->ghost) + 20.0 * (double)A->no_rows : 0.0) + 20.0 * (double)A->g_nhot +
         8.0 * (double)A->nghost + 20.0 * (double)A->nsend;
    b += ksg_bins_bytes(&A->gbins);
    for (const ksg_rest &cg : A->grest) b += block_stream_bytes(cg.blk) + 20.0 * (double)cg.nrows + 20.0 * (double)cg.nhot;
  }
  return b;
}
/* The same product by the plain CSR model (12 bytes per entry, 8 per row pointer, x and y once):
 * what a kernel reading rowptr/colidx/val as they stand would move. */
static inline double spmv_bytes_csr_model(ksg_mat A) {
  return 12.0 * (double)A->nnz + 8.0 * (double)(A->n + 1) + 8.0 * (double)(A->nc + A->nghost) + 8.0 * (double)A->n;
}

extern "C" int ksg_mat_traffic_model(ksg_mat A, double *executed, double *csr_model) {
  if (!A) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_traffic_model: NULL matrix");
  if (executed) *executed = spmv_bytes(A);
  if (csr_model) *csr_model = spmv_bytes_csr_model(A);
  return KSG_OK;
}

extern "C" int ksg_mat_describe(ksg_mat A, char *buf, size_t len) {
  if (!A || !buf || !len) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_describe: NULL argument");
  int k;
  if (A->diag.pc_w)
    k = snprintf(buf, len, "row codes: one lane per row, one byte per entry naming its (column - row, value) pair out of at most 256, "
                 "%d bytes of codes per row (rows of up to %d entries), %d consecutive 256-row blocks per XCD; values: in the pair table; "
                 "packed hot columns: 0; ghost entries received per product: %lld", A->diag.pc_w <= 7 ? 8 : 16, A->diag.pc_w,
                 A->diag.pc_chunk, (long long)A->nghost);
  else
  k = snprintf(buf, len, "row tiles of <= %d entries / %d rows: %lld one lane per row, %lld several lanes per row, %lld single long rows; "
           "2-byte row offsets; values: %s; packed hot columns: %lld; ghost entries received per product: %lld",
           KSG_TILE_NNZ, KSG_TILE_ROWS, (long long)A->diag.n_stream, (long long)A->diag.n_group, (long long)A->diag.n_long,
           A->diag.nvt == 2 ? "1-byte indices into a 255-entry table off the diagonal, the diagonal as a dense vector" :
           A->diag.nvt ? "1-byte indices into a 256-entry table" : "8-byte stream", (long long)A->nhot, (long long)A->nghost);
  if (A->bins.nent && k > 0 && (size_t)k < len) {
    long long rows = 0, ent = 0, hot = 0;
    for (const ksg_rest &cg : A->rest) { rows += cg.nrows; ent += cg.blk.nnz; hot += cg.nhot; }
    k = snprintf(buf, len, "binned part: %lld of %lld entries in cells of %d-column blocks x row blocks of <= %d rows (%lld runs, %lld sums of "
                 "neighbouring entries of a row, %lld row blocks, values: %s), expanded through LDS slices of x and summed per row block in LDS%s; remaining entries: %lld in %lld rows through "
                 "the tile kernel (row tiles of <= %d entries, %lld packed hot columns), added afterwards",
                 (long long)A->bins.nreal, (long long)A->nnz, A->bins.W, A->bins.H, (long long)A->bins.nseg, (long long)A->bins.nent, (long long)A->bins.nbins,
                 A->bins.vmode == 0 ? "one" : A->bins.vmode == 1 ? "1-byte indices" : "8-byte stream",
                 A->bins.dv ? ", the diagonal as a dense vector in the same pass" : "", ent, rows, KSG_TILE_NNZ, hot);
  }
  if (A->ctx->nranks > 1 && A->gbins.nent && k > 0 && (size_t)k < len) {
    long long rows = 0, ent = 0;
    for (const ksg_rest &cg : A->grest) { rows += cg.nrows; ent += cg.blk.nnz; }
    snprintf(buf + k, len - (size_t)k, "; ghost block: %lld entries in %lld rows, %lld of them binned (LDS slices of the ghost buffer, %lld runs, %lld row blocks), "
             "%lld remaining entries in %lld rows through the tile kernel%s", (long long)A->o_nnz, (long long)A->no_rows, (long long)A->gbins.nreal,
             (long long)A->gbins.nseg, (long long)A->gbins.nbins, ent, rows,
             A->halo_rounds ? "; halo in rounds, a piece of every source's share per round, the expand tasks of a piece start when it has arrived" : "");
  } else if (A->ctx->nranks > 1 && k > 0 && (size_t)k < len)
    snprintf(buf + k, len - (size_t)k, "; ghost block: %lld entries in %lld rows through the same tile kernel (%lld/%lld/%lld tiles, %lld packed)",
             (long long)A->o_nnz, (long long)A->no_rows, (long long)A->ghost.n_stream, (long long)A->ghost.n_group, (long long)A->ghost.n_long,
             (long long)A->g_nhot);
  return KSG_OK;
}

extern "C" int ksg_mat_mult(ksg_mat A, const double *x, double *y) {
  if (!A) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_mult: NULL matrix");
  if (A->n && (!x || !y)) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_mult: NULL vector");
  if (x == y) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_mult: x and y must be different vectors");
  ksg_ctx c = A->ctx;
  KSG_CHK(ksg_op_begin(c, KSG_EV_MATMULT));
  KSG_CHK(mat_mult_raw(A, x, y));
  return ksg_op_end(c, KSG_EV_MATMULT, spmv_bytes(A), 2.0 * (double)A->nnz);
}

/* ||A x - kr x + ki xi||_2 over all ranks (xi may be NULL).  One rank: a single pass over
 * the matrix, nothing stored.  Several ranks: A x into `work` (halo exchange included), then
 * one pass over work, x, xi. */
extern "C" int ksg_mat_residual_norm(ksg_mat A, double kr, double ki, const double *x, const double *xi,
                                     double *work, double *norm) {
  if (!A || !norm) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_residual_norm: NULL argument");
  if (A->n && !x) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_residual_norm: NULL vector");
  if (A->n != A->nc || A->N != A->NC) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_residual_norm: the operator must be square");
  ksg_ctx c = A->ctx;
  const bool halo = c->nranks > 1 || !A->rest.empty() || A->bins.nent;     /* the product cannot be fused with the norm */
  if (halo && A->n && (!work || work == x || work == xi)) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_residual_norm: a separate work vector is needed with several ranks, column groups or a binned part");
  KSG_CHK(ksg_op_begin(c, KSG_EV_MATMULT));
  int nb = 0;
  if (!halo) {
    nb = A->diag.pc_w ? (int)rowcode_grid(A->diag) : (int)A->diag.ntiles;
    if (nb) {
      KSG_CHK(ksg_ensure_partial(c, (size_t)nb));
      if (A->diag.pc_w) launch_rowcode<1>(c->stream, A->diag, x, NULL, kr, ki, xi, c->d_partial);
      else {
        if (A->nhot) hipLaunchKernelGGL(k_gather, dim3(ksg_grid_for(A->nhot)), dim3(KSG_BLOCK), 0, c->stream, x, A->d_hot_idx, A->d_xhot, A->nhot);
        launch_tiles<1>(c->stream, A->diag, x, NULL, kr, ki, xi, c->d_partial, A->nhot ? A->d_xhot : NULL, (int32_t)A->nc, NULL);
      }
    }
  } else {
    KSG_CHK(mat_mult_raw(A, x, work));
    nb = A->n ? ksg_grid_for(A->n) : 0;
    if (nb) {
      KSG_CHK(ksg_ensure_partial(c, (size_t)nb));
      hipLaunchKernelGGL(k_resnorm, dim3(nb), dim3(KSG_BLOCK), 0, c->stream, work, x, xi, kr, ki, A->n, c->d_partial);
    }
  }
  KSG_HIP(hipGetLastError());
  if (nb) { KSG_CHK(ksg_launch_reduce_sum(c, c->d_partial, nb, 1, c->d_red)); }
  else KSG_HIP(hipMemsetAsync(c->d_red, 0, sizeof(double), c->stream));
  /* one rank: the product's bytes without the store of y; several: the product, then a pass over it */
  const double bytes = spmv_bytes(A) + (xi ? 8.0 * (double)A->n : 0.0) + (halo ? 16.0 * (double)A->n : -8.0 * (double)A->n);
  KSG_CHK(ksg_op_end(c, KSG_EV_MATMULT, bytes, 2.0 * (double)A->nnz + 5.0 * (double)A->n));
  double s = 0.0;
  KSG_CHK(ksg_comm_allreduce(c, c->d_red, &s, 1, 0));
  *norm = sqrt(s);
  return KSG_OK;
}

extern "C" int ksg_bv_matmult(ksg_bv V, ksg_mat A, ksg_bv W) {
  if (!V || !W || !A) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_matmult: NULL argument");
  if (V == W) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_matmult: V and W must differ");
  if (V->n != A->nc || W->n != A->n || V->k - V->l != W->k - W->l || V->l < 0 || W->l < 0)
    KSG_FAIL(KSG_ERR_ARG, "ksg_bv_matmult: size mismatch");
  ksg_ctx c = V->ctx;
  const int nc = V->k - V->l;
  KSG_CHK(ksg_op_begin(c, KSG_EV_MATMULT));
  const double *X = V->d + (int64_t)(V->nc + V->l) * V->ld;
  double *Y = W->d + (int64_t)(W->nc + W->l) * W->ld;
  if (c->nranks == 1 && nc > 1 && A->diag.ntiles && !A->nhot && A->rest.empty() && !A->bins.nent) {
    /* one pass over the matrix for all the columns */
    hipLaunchKernelGGL(k_spmm_tiles<8>, dim3((unsigned)A->diag.ntiles), dim3(KSG_BLOCK), 0, c->stream, A->diag.tiles,
                       A->diag.tile_p, A->diag.rowptr, A->diag.colidx, A->diag.val, X, V->ld, Y, W->ld, nc);
    KSG_HIP(hipGetLastError());
    /* k_spmm_tiles reads the 8-byte values and the row pointers as they stand */
    const double bytes = 12.0 * (double)A->nnz + 8.0 * (double)(A->n + 1) + 12.0 * (double)(A->diag.ntiles + 1) + 8.0 * nc * (double)(A->nc + A->n);
    return ksg_op_end(c, KSG_EV_MATMULT, bytes, 2.0 * nc * (double)A->nnz);
  }
  /* several ranks, or packed x entries: column by column, each product with its halo exchange */
  for (int j = 0; j < nc; j++) KSG_CHK(mat_mult_raw(A, X + (int64_t)j * V->ld, Y + (int64_t)j * W->ld));
  return ksg_op_end(c, KSG_EV_MATMULT, nc * spmv_bytes(A), 2.0 * nc * (double)A->nnz);
}
