/* Binned product for power-law operators: y = B x + d .* x for the part B of a CSR block whose
 * entries fall into well filled (row block, column block) cells, and the block's diagonal d.
 *
 * A gather x[col] of a power-law operator whose x is far larger than the caches costs one
 * request beyond L2 per entry, and that path serves a fixed number of requests per second
 * whatever their size.  The binned form turns those gathers into two streaming passes:
 *
 *   expand : the columns are cut into blocks of W; a workgroup holds one block's slice of x in
 *            LDS and streams the entries of those columns (2 bytes of column within the block
 *            per entry) and looks x up in LDS.  The entries of one (row block, column block)
 *            cell form a run that is consecutive in both orders, and within a run the entries
 *            come sorted by row, so the entries a row has in a column block are neighbours: a
 *            stretch.  The wave adds up the products of a stretch in registers (a segmented
 *            scan over rows of 16 lanes) and writes one sum per stretch, a slot, where the
 *            second pass wants it -- on a power-law operator half of the products or fewer.
 *            The writes are runs, not scatters; every run's slots are filled up with zeros to
 *            whole 64-byte pieces and written with non-temporal stores (a partly written piece
 *            costs the memory system a read of the rest).
 *   reduce : the rows are cut into blocks of at most H rows and E entries; one wave holds a
 *            block's partial sums in LDS, streams the block's slots (8 bytes) and their row
 *            numbers within the block (2 bytes), adds them up with LDS atomics (one wave per
 *            block: the order of the additions is fixed; lanes that still share a row -- a
 *            stretch cut at a 16-lane boundary -- are added up in registers first), adds the
 *            diagonal term and stores y.
 *            A row longer than E is cut into pieces, each summed by a wave of its own in
 *            registers; the sums of the pieces are added to the row afterwards, in their order.
 *
 * Only cells with at least T entries are binned; what is left (the entries of sparse cells) is a
 * CSR over the rows it touches, which the tile kernel adds afterwards.  A row's sum is then: its
 * binned entries by column block, its diagonal entry, its remaining entries in storage order --
 * the same products as the reference's CSR loop (MatMult_SeqAIJ behind
 * src/sys/classes/st/impls/shift/shift.c:38) in another order, the same order in every product.
 *
 * Tuning variables, read when an operator is created: KSG_BINS_W (columns per column block, a
 * power of two up to 16384 = 128 KiB of LDS), KSG_BINS_H (rows per row block, up to 8192; 4096),
 * KSG_BINS_E (entries per row block; unless set, 1e-3 of the block's entries as a power of two
 * between 65536 and 1048576), KSG_BINS_T (fewest entries of a binned cell), KSG_BINS_TASK_GROUPS
 * (groups of 64 entries per expand workgroup; unless set, up to 8192), KSG_BINS_THREADS (threads
 * of an expand workgroup), KSG_BINS_ALIGN (slots a run is filled up to a multiple of; 8). */
#include <algorithm>
#include <numeric>
#include <stdlib.h>

#include "ksg_internal.h"

#define BINS_P1_THREADS 1024    /* most threads of an expand workgroup (launch bound); KSG_BINS_THREADS sets the actual number */
#ifndef BINS_P1_UNROLL
#define BINS_P1_UNROLL  8      /* groups of 64 entries a wave of the expand pass has in flight: 16 spill registers at 1024 threads */
#endif
#ifndef BINS_P2_UNROLL
#define BINS_P2_UNROLL  32     /* loads a reduce wave has in flight (16: 4.58 ms per product at R-MAT scale 26, 24: 4.51, 32: 4.50) */
#endif

/* data-parallel-primitive move inside a row of 16 lanes: the value of the lane N places below;
 * lanes whose source lies outside their row get `old` */
template <int N>
__device__ __forceinline__ double bins_row_shr(double old, double src) {
  const long long o = __double_as_longlong(old), v = __double_as_longlong(src);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)v, 0x110 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(v >> 32), 0x110 + N, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int N>
__device__ __forceinline__ int bins_row_shr(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x110 + N, 0xf, 0xf, false); }

/* Bit `lane` of a mask that is the same in every lane of the wave (it lives in scalar registers):
 * one select with the mask as the per-lane condition, instead of a 64-bit shift per lane. */
__device__ __forceinline__ int bins_lane_bit(uint64_t mask) {
  int r;
  asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(mask));
  return r;
}
/* number of set bits of a wave-uniform mask in the lanes below this one */
__device__ __forceinline__ uint32_t bins_count_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
/* One step of the segmented scan: the value of the lane N below within the lane's row of 16 lanes,
 * +0.0 where that lane lies outside the row or where the lane's bit of the wave-uniform mask `stop`
 * is set.  The move across lanes and the select are one instruction per half: the select takes its
 * first operand through the data-parallel-primitive path (lanes without a source get zero) and the
 * mask as its condition.  The operand may have been written by the instruction before: the two wait
 * states such a read needs stand in front. */
template <int N>
__device__ __forceinline__ double bins_below_unless(double v, uint64_t stop) {
  const long long b = __double_as_longlong(v);
  const int lo = (int)b, hi = (int)(b >> 32), zero = 0;
  int rlo, rhi;
  asm("s_nop 1\n\t"
      "s_mov_b64 vcc, %5\n\t"
      "v_cndmask_b32_dpp %0, %2, %4, vcc row_shr:%6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %1, %3, %4, vcc row_shr:%6 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi), "v"(zero), "s"(stop), "n"(N) : "vcc");
  return __longlong_as_double(((long long)rhi << 32) | (unsigned int)rlo);
}

/* What a wave holds of BINS_P1_UNROLL groups of 64 entries between its loads and its stores. */
struct bins_step {
  uint32_t u[BINS_P1_UNROLL];                           /* columns within the slice of the lane's entry and of its neighbour (one 4-byte load per pair of lanes) */
  uint8_t vi[BINS_P1_UNROLL];                           /* value index (VM = 1)                    */
  double v[BINS_P1_UNROLL];                             /* value (VM = 2)                          */
  int64_t o[BINS_P1_UNROLL];                            /* O of the lane's part */
  uint64_t sm[BINS_P1_UNROLL];                          /* the groups' stretch starts (scalar)      */
};

/* The loads of one step: BINS_P1_UNROLL groups from group g on (g is the same in every lane:
 * what a group shares comes through scalar loads).  A lane's part is the part of the group's
 * first entry plus the part starts among the lanes up to it (lane 0 never counts as one: the
 * starts at lanes 1..lane are the bits below the lane of the mask moved down by one). */
template <int VM, bool FULL>
__device__ __forceinline__ void bins_expand_load(bins_step &s, int64_t g, int64_t g1, int lane, const uint16_t *__restrict__ col16,
                                                 const uint8_t *__restrict__ vidx, const double *__restrict__ val,
                                                 const uint32_t *__restrict__ gfirst, const uint64_t *__restrict__ runmask,
                                                 const uint64_t *__restrict__ rowmask, const int64_t *__restrict__ O) {
#pragma unroll
  for (int j = 0; j < BINS_P1_UNROLL; j++) {
    s.u[j] = 0; s.vi[j] = 0; s.v[j] = 0.0; s.o[j] = 0; s.sm[j] = ~0ull;
    if (FULL || g + j < g1) {                                      /* uniform over the wave */
      const int64_t p = (g + j) * 64 + lane;
      s.u[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(col16) + (p >> 1));
      if (VM == 1) s.vi[j] = __builtin_nontemporal_load(vidx + p);
      if (VM == 2) s.v[j] = __builtin_nontemporal_load(val + p);
      s.sm[j] = rowmask[g + j];
      s.o[j] = (O + gfirst[g + j])[bins_count_below(runmask[g + j] >> 1)];
    }
  }
}

/* Products, sums of the stretches, stores.  A lane forms its entry's product; the products of a
 * stretch -- neighbouring entries of one row, within a row of 16 lanes -- are added up by a
 * segmented scan in registers: at step N a lane adds the value of the lane N below it when no
 * stretch starts in between (masks formed from the group's stretch starts by scalar operations),
 * and the last lane of a stretch writes the sum: slot O[part] + the stretch starts among the lanes
 * up to it (O has the starts that precede the part taken off; lane 0 always starts a stretch, so
 * the count is one plus the bits below the lane of the mask moved down by one -- the one is in xe1).
 * The products are rounded one by one (no contraction with the additions), the additions come in
 * the scan's fixed order. */
template <int VM, bool FULL>
__device__ __forceinline__ void bins_expand_finish(const bins_step &s, int64_t g, int64_t g1, int half, const double *xs,
                                                   const double *__restrict__ vtab, double vc, double *__restrict__ xe1) {
#pragma clang fp contract(off)
  double a[BINS_P1_UNROLL];
#pragma unroll
  for (int j = 0; j < BINS_P1_UNROLL; j++) {
    a[j] = xs[(s.u[j] >> half) & 0xffffu];
    if (VM == 0) a[j] *= vc;
    else if (VM == 1) a[j] *= vtab[s.vi[j]];
    else a[j] *= s.v[j];
  }
#pragma unroll
  for (int j = 0; j < BINS_P1_UNROLL; j++) {
    /* a step in which no lane would take anything is left out, and with it the steps after it (a
     * group of the cold columns has nothing but stretches of one entry: no step at all) */
    const uint64_t f1 = s.sm[j], f2 = f1 | (f1 << 1), f4 = f2 | (f2 << 2), f8 = f4 | (f4 << 4);
    if (~f1) {
      a[j] += bins_below_unless<1>(a[j], f1);
      if (~f2) {
        a[j] += bins_below_unless<2>(a[j], f2);
        if (~f4) {
          a[j] += bins_below_unless<4>(a[j], f4);
          if (~f8) a[j] += bins_below_unless<8>(a[j], f8);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < BINS_P1_UNROLL; j++) {
    /* the last lanes of the stretches store: the lanes whose next lane starts one, and lane 63 --
     * a mask in scalar registers, taken as the execution mask as it is */
    const uint64_t above = s.sm[j] >> 1;
    if ((FULL || g + j < g1) && __builtin_amdgcn_inverse_ballot_w64(above | (1ull << 63)))
      __builtin_nontemporal_store(a[j], xe1 + (s.o[j] + (int64_t)bins_count_below(above)));
  }
}

template <int VM>
__global__ __launch_bounds__(BINS_P1_THREADS) void k_bins_expand(const int64_t *__restrict__ tasks,
                                                                  const uint16_t *__restrict__ col16,
                                                                  const uint8_t *__restrict__ vidx,
                                                                  const double *__restrict__ vtab,
                                                                  const double *__restrict__ val, double vc,
                                                                  const uint32_t *__restrict__ gfirst,
                                                                  const uint64_t *__restrict__ runmask,
                                                                  const uint64_t *__restrict__ rowmask,
                                                                  const int64_t *__restrict__ O,
                                                                  const double *__restrict__ x, int64_t ncols,
                                                                  double *__restrict__ xe, int W) {
  extern __shared__ double xs[];
  const int64_t cb = tasks[3 * (int64_t)blockIdx.x];
  const int64_t g0 = tasks[3 * (int64_t)blockIdx.x + 1], g1 = tasks[3 * (int64_t)blockIdx.x + 2];
  const int64_t cbase = cb * W;
  const int nthreads = blockDim.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   /* the same in every lane: group numbers stay scalar */
  const int64_t stride = (int64_t)(nthreads / 64) * BINS_P1_UNROLL;
  const int half = (lane & 1) << 4;                        /* where the lane's column stands in the four bytes it loads */
  /* the wave's first step is requested before the slice of x: its latency hides behind the slice's */
  bins_step sa, sb;
  int64_t g = g0 + (int64_t)wave * BINS_P1_UNROLL;
  if (g + BINS_P1_UNROLL <= g1) bins_expand_load<VM, true>(sa, g, g1, lane, col16, vidx, val, gfirst, runmask, rowmask, O);
  else if (g < g1) bins_expand_load<VM, false>(sa, g, g1, lane, col16, vidx, val, gfirst, runmask, rowmask, O);
  /* the slice of x: all of a thread's loads in flight together (the last slice may be short:
   * the loads are clamped to the last column, what lies beyond is never looked up) */
  for (int i0 = 0; i0 < W; i0 += 8 * nthreads) {
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int64_t col = cbase + i0 + j * nthreads + threadIdx.x;
      t[j] = x[col < ncols ? col : ncols - 1];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = i0 + j * nthreads + threadIdx.x;
      if (i < W) xs[i] = t[j];
    }
  }
  if (threadIdx.x == 0) xs[W] = 0.0;                      /* the column of the entries that fill a run up to whole cache lines */
  __syncthreads();
  /* Every step's loads are requested before the stores of the step before it are issued: loads
   * and stores complete in the order of their issue, so a wave that stored first would wait for
   * its stores to be written before it could use the next loads.  The two steps in flight take
   * turns in two sets of registers (a copy from one set to the other would wait for the loads). */
  while (g + 2 * stride + BINS_P1_UNROLL <= g1) {          /* sa holds the groups from g on; these and the next two steps are whole */
    bins_expand_load<VM, true>(sb, g + stride, g1, lane, col16, vidx, val, gfirst, runmask, rowmask, O);
    __builtin_amdgcn_sched_barrier(0);                      /* nothing of the step before moves in front of these loads (it would wait for its own) */
    bins_expand_finish<VM, true>(sa, g, g1, half, xs, vtab, vc, xe + 1);
    __builtin_amdgcn_sched_barrier(0);
    bins_expand_load<VM, true>(sa, g + 2 * stride, g1, lane, col16, vidx, val, gfirst, runmask, rowmask, O);
    __builtin_amdgcn_sched_barrier(0);
    bins_expand_finish<VM, true>(sb, g + stride, g1, half, xs, vtab, vc, xe + 1);
    __builtin_amdgcn_sched_barrier(0);
    g += 2 * stride;
  }
  while (g < g1) {                                          /* the last steps, whole or not */
    const int64_t gn = g + stride;
    if (gn < g1) bins_expand_load<VM, false>(sb, gn, g1, lane, col16, vidx, val, gfirst, runmask, rowmask, O);
    bins_expand_finish<VM, false>(sa, g, g1, half, xs, vtab, vc, xe + 1);
    sa = sb;
    g = gn;
  }
}

/* one step of a segmented scan: f = a stretch starts at or below this lane within the window so far */
template <int N>
__device__ __forceinline__ void bins_scan_step(double &v, int &f) {
  const double vv = bins_row_shr<N>(0.0, v);
  const int ff = bins_row_shr<N>(1, f);
  if (!f) { v += vv; f = ff; }
}

/* The 64 products of one load of a reduce wave into the block's sums.  The products of a cell come
 * sorted by row, and a row with many entries in a column block of hub columns fills whole stretches
 * of lanes: additions to one LDS address are done one after the other (all 64 lanes on one row:
 * 2.7 times the time of distinct rows in tools/bins_stream_probe.hip; a quarter of the pass at
 * R-MAT scale 24).  When at least half of the lanes repeat the row of the lane before them, every
 * row of 16 lanes first adds up its stretches of equal rows with a segmented scan in registers
 * (four DPP steps, no LDS traffic, a fixed order) and only the last lane of a stretch adds to LDS:
 * the pass 0.71 -> 0.56 ms (scan from 8 / 16 / 32 repeats on, or always: 1.478 / 1.468 / 1.445 /
 * 1.488 ms per product; without it 1.601 ms).  Every lane of the wave must be active. */
__device__ __forceinline__ void bins_add_group(double *acc, uint16_t r, double v) {
  const int ri = (int)r;
  const int rprev = bins_row_shr<1>(-1, ri);                /* lane 0 of a row of 16: -1, a stretch starts */
  int f = rprev != ri;
  if (__popcll(__ballot(f)) > 32) {                         /* uniform over the wave */
    atomicAdd(&acc[r], v);
    return;
  }
  const int next_starts = __builtin_amdgcn_update_dpp(1, f, 0x101, 0xf, 0xf, false);   /* row_shl:1: the flag of the lane above */
  bins_scan_step<1>(v, f);
  bins_scan_step<2>(v, f);
  bins_scan_step<4>(v, f);
  bins_scan_step<8>(v, f);
  if (next_starts) atomicAdd(&acc[r], v);
}

/* One wave per row block: y[r] = sum of the block's products of row r + dv[r] * x[r + dshift].
 * A block whose first row lies at or beyond nrows is a piece of a long row: it has one row, and
 * its sum goes to part[r0 - nrows] (k_bins_long_rows adds the pieces to y afterwards).
 * rowmap (may be NULL): row i of the block is row rowmap[i] of y and the sums are added to what y
 * holds -- the ghost block of a distributed operator, whose rows are a list of the local rows. */
__global__ __launch_bounds__(64) void k_bins_reduce(const int64_t *__restrict__ binp0, const int64_t *__restrict__ binp1,
                                                    const int32_t *__restrict__ bin_r0, const int32_t *__restrict__ bin_nr,
                                                    const uint16_t *__restrict__ rowloc, const double *__restrict__ xe,
                                                    const double *__restrict__ dv, const double *__restrict__ x, int64_t dshift,
                                                    double *__restrict__ y, int64_t nrows, double *__restrict__ part,
                                                    const int32_t *__restrict__ rowmap) {
  extern __shared__ double acc[];
  const int64_t b = blockIdx.x;
  const int64_t p0 = binp0[b], p1 = binp1[b];
  const int r0 = bin_r0[b], nr = bin_nr[b];
  const int lane = threadIdx.x;
  if (r0 >= nrows) {
    /* one row: every lane sums its share in registers, the wave adds the 64 sums in a fixed order */
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    int64_t p = p0 + lane;
    for (; p + 64 * 15 < p1; p += 64 * 16) {               /* sixteen loads in flight; the same four sums as the loop below */
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(xe + p + 64 * j);
#pragma unroll
      for (int j = 0; j < 16; j++) t[j & 3] += v[j];
    }
    for (; p + 64 * 3 < p1; p += 64 * 4) {
      double v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = __builtin_nontemporal_load(xe + p + 64 * j);
#pragma unroll
      for (int j = 0; j < 4; j++) t[j] += v[j];
    }
    for (; p < p1; p += 64) t[0] += xe[p];
    double s = (t[0] + t[1]) + (t[2] + t[3]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) part[r0 - nrows] = s;
    return;
  }
  for (int i = lane; i < nr; i += 64) acc[i] = 0.0;
  int64_t p = p0 + lane;
  /* whole batches of loads (every lane active: bins_add_group), then one guarded batch for what is left */
  for (; p - lane + 64 * BINS_P2_UNROLL <= p1; p += 64 * BINS_P2_UNROLL) {
    uint16_t r[BINS_P2_UNROLL];
    double v[BINS_P2_UNROLL];
#pragma unroll
    for (int j = 0; j < BINS_P2_UNROLL; j++) {
      r[j] = __builtin_nontemporal_load(rowloc + p + 64 * j);
      v[j] = __builtin_nontemporal_load(xe + p + 64 * j);
    }
#pragma unroll
    for (int j = 0; j < BINS_P2_UNROLL; j++) bins_add_group(acc, r[j], v[j]);
  }
  {
    uint16_t r[BINS_P2_UNROLL];
    double v[BINS_P2_UNROLL];
#pragma unroll
    for (int j = 0; j < BINS_P2_UNROLL; j++) {
      r[j] = 0; v[j] = 0.0;
      if (p + 64 * j < p1) { r[j] = rowloc[p + 64 * j]; v[j] = xe[p + 64 * j]; }
    }
#pragma unroll
    for (int j = 0; j < BINS_P2_UNROLL; j++)
      if (p + 64 * j < p1) atomicAdd(&acc[r[j]], v[j]);
  }
  /* the wave's own LDS operations complete in order: no barrier between its adds and its reads */
  if (rowmap) {
    /* the block's rows stand for rows rowmap[.] of y, which already holds the other blocks' sums */
    for (int i0 = 0; i0 < nr; i0 += 64 * 8) {
      int32_t rr[8];
      double yy[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = i0 + 64 * j + lane;
        rr[j] = i < nr ? rowmap[r0 + i] : 0;
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = i0 + 64 * j + lane;
        yy[j] = i < nr ? y[rr[j]] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = i0 + 64 * j + lane;
        if (i < nr) y[rr[j]] = yy[j] + acc[i];
      }
    }
    return;
  }
  for (int i0 = 0; i0 < nr; i0 += 64 * 8) {
    double dd[8], xx[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = i0 + 64 * j + lane;
      dd[j] = 0.0; xx[j] = 0.0;
      if (dv && i < nr) { dd[j] = dv[r0 + i]; xx[j] = x[r0 + i + dshift]; }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = i0 + 64 * j + lane;
      if (i < nr) y[r0 + i] = dv ? fma(dd[j], xx[j], acc[i]) : acc[i];
    }
  }
}

/* y[row] += the sums of the row's pieces, in their order (one thread per long row) */
__global__ __launch_bounds__(KSG_BLOCK) void k_bins_long_rows(const int32_t *__restrict__ rows, const int32_t *__restrict__ first,
                                                              int nlong, const double *__restrict__ part, double *__restrict__ y) {
  const int j = blockIdx.x * KSG_BLOCK + threadIdx.x;
  if (j >= nlong) return;
  double s = 0.0;
  for (int k = first[j]; k < first[j + 1]; k++) s += part[k];
  y[rows[j]] += s;
}

template <class T>
static int bins_upload(T **dst, const T *src, size_t count) {
  *dst = NULL;
  hipError_t e = hipMalloc((void **)dst, (count ? count : 1) * sizeof(T));
  if (e != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "binned operator upload: %zu bytes: %s", count * sizeof(T), hipGetErrorString(e));
  if (count) KSG_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return KSG_OK;
}

static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && atoi(e) > 0 ? atoi(e) : dflt;
}

/* A run: the entries of one (unit, column block) cell; what the expand pass writes for it -- one
 * sum per stretch of entries of one row, a slot -- is filled up with zeros to a multiple of
 * BINS_RUN_ALIGN slots: every run then starts on a 64-byte boundary of the array of sums and
 * covers whole 64-byte pieces.  Measured (tools/bins_stream_probe.hip, runs of 64 products scattered
 * over 2 GiB): runs that start 24 bytes off a line 1.07 ms against 0.61 ms for aligned ones; with
 * non-temporal stores whole 64-byte pieces are enough (0.58-0.67 ms), with plain stores a half
 * written 128-byte line still costs (0.77-0.87 ms). */
#ifndef BINS_RUN_ALIGN
#define BINS_RUN_ALIGN 8
#endif
struct bins_seg { int32_t c, real, unit; int64_t j, praw; };   /* column block, entries, unit; number among the runs of its column block; first entry in the raw expand order */

void ksg_bins_free(ksg_bins *B) {
  if (B->col16) hipFree(B->col16);
  if (B->vidx) hipFree(B->vidx);
  if (B->vtab) hipFree(B->vtab);
  if (B->val) hipFree(B->val);
  if (B->gfirst) hipFree(B->gfirst);
  if (B->O) hipFree(B->O);
  if (B->runmask) hipFree(B->runmask);
  if (B->rowmask) hipFree(B->rowmask);
  if (B->tasks) hipFree(B->tasks);
  if (B->rowloc) hipFree(B->rowloc);
  if (B->binp0) hipFree(B->binp0);
  if (B->binp1) hipFree(B->binp1);
  if (B->bin_r0) hipFree(B->bin_r0);
  if (B->bin_nr) hipFree(B->bin_nr);
  if (B->dv) hipFree(B->dv);
  if (B->long_rows) hipFree(B->long_rows);
  if (B->long_first) hipFree(B->long_first);
  if (B->piece_row) hipFree(B->piece_row);
  if (B->part) hipFree(B->part);
  if (B->xe) hipFree(B->xe);
  free(B->task_stage);
  memset(B, 0, sizeof(*B));
}

/* The binned part on the host, array for array what the device holds. */
struct bins_host {
  int64_t nent = 0, nreal = 0, nent1 = 0, nseg = 0, nparts = 0, nbins = 0, ntasks = 0, nrows = 0, dshift = 0;
  int W = 0, H = 0, vmode = 2;
  bool fold_diag = false;
  double vconst = 0.0;
  ksg_hvec<uint16_t> col16, rowloc;                       /* the large arrays: filled by the build's threads, never cleared first */
  ksg_hvec<uint8_t> vidx;
  ksg_hvec<double> val, dv;
  ksg_hvec<uint32_t> gfirst;
  ksg_hvec<uint64_t> runmask, rowmask;
  ksg_hvec<int64_t> O;
  std::vector<double> vtab;
  std::vector<int64_t> tasks, binp0, binp1, task_stage;
  std::vector<int32_t> r0s, nrs;
  std::vector<int32_t> long_rows, long_first, piece_row;   /* rows cut into pieces; their pieces long_first[j] .. long_first[j+1]; row of every piece */
};

/* a unit of the reduce pass: rows r0 .. r0+nr of the block (those longer than E skipped), or, with
 * e0 >= 0, the entries e0 .. e1 of the long row r0 (piece number `piece`) */
struct bins_block { int32_t r0, nr, piece; int64_t e0, e1; };

/* Splits the block (nrows x ncols CSR) into its binned part *B (with the diagonal, when every row
 * has exactly one entry in column r + dshift and has_diag says that this is the operator's
 * diagonal) and the rest, returned as a CSR over the rows it touches: row i of lrp/lci/lva is row
 * lrows[i] of the block.  When fewer than half of the entries can be binned nothing is built
 * (B->nent == 0, the vectors stay empty). */
static void bins_build_host(bins_host *B, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *colidx,
                            const double *val, bool has_diag, int64_t dshift, std::vector<int32_t> &lrows, std::vector<int64_t> &lrp,
                            std::vector<int32_t> &lci, std::vector<double> &lva, int H, int64_t E, int maxW,
                            const ksg_colstages *colstage) {
  *B = bins_host();
  lrows.clear(); lrp.clear(); lci.clear(); lva.clear();
  /* R-MAT scale 26 (1.13e9 entries, E = 65536): 6.65 ms per product with H = 2048, 7.27 with 4096; before the reduce pass combined
   * equal rows: 7.05 / 7.11 / 7.21 ms with H = 2048 / 4096 / 8192 (E = 131072), T = 6: 7.26;
   * scale 24 then: W = 16384 1.61 ms, 8192 1.63, 4096 1.69 (profiles/r02_spmv_rmat_binned.txt) */
  int W = env_int("KSG_BINS_W", 16384);
  const int T = env_int("KSG_BINS_T", 4);
  int wshift = 0;
  while ((1 << wshift) < W) wshift++;
  W = 1 << wshift;
  if (W > 16384) W = 16384, wshift = 14;                   /* 128 KiB of x in LDS */
  while (W > maxW && W > 64) { W >>= 1; wshift--; }        /* what the device's LDS holds */
  if (W < 64) W = 64, wshift = 6;
  if (H > 8192) H = 8192;                                  /* 64 KiB of sums in LDS */
  if (H < 64) H = 64;
  if (E < 1) E = 1;
  const int64_t nnz = rowptr[nrows];
  const int64_t NC = (ncols + W - 1) >> wshift;
  if (nnz == 0 || nrows == 0 || NC > (int64_t)INT32_MAX || nrows > (int64_t)INT32_MAX / 2) return;

  ksg_phase_timer tm("binned build");
  /* the passes over the entries run on several host threads; every array comes out the same
   * whatever their number (parts are consecutive rows or units, joined in order) */
  const int NT = nnz >= ((int64_t)1 << 20) ? ksg_host_threads() : 1;

  /* the diagonal is folded into the reduce pass only when every row has it exactly once */
  bool fold = has_diag;
  if (fold) {
    std::vector<char> bad((size_t)NT, 0);
    ksg_parallel(NT, [&](int t, int nt) {
      const int64_t ra = ksg_split_rows(rowptr, nrows, t, nt), rb = ksg_split_rows(rowptr, nrows, t + 1, nt);
      for (int64_t r = ra; r < rb; r++) {
        int seen = 0;
        for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) if ((int64_t)colidx[p] == r + dshift) seen++;
        if (seen != 1) { bad[(size_t)t] = 1; break; }
      }
    });
    for (int t = 0; t < NT; t++) if (bad[(size_t)t]) fold = false;
  }
  tm.lap("diagonal check");

  /* the distinct values among the entries that may be binned (all but a folded diagonal), numbered
   * in storage order, if there are at most 256: the build then writes one-byte indices at once
   * instead of collecting 8-byte values and indexing them afterwards */
  ksg_value_set vset;
  std::vector<double> pretab;
  bool few = true;
  {
    std::vector<std::vector<uint64_t>> seen((size_t)NT);
    ksg_parallel(NT, [&](int t, int nt) {
      std::vector<uint64_t> &mine = seen[(size_t)t];
      ksg_value_set set;
      uint64_t lastbits = 0;
      bool have = false;
      const int64_t ra = ksg_split_rows(rowptr, nrows, t, nt), rb = ksg_split_rows(rowptr, nrows, t + 1, nt);
      for (int64_t r = ra; r < rb; r++)
        for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
          if (fold && (int64_t)colidx[p] == r + dshift) continue;
          uint64_t bits;
          memcpy(&bits, &val[p], 8);
          if (have && bits == lastbits) continue;
          lastbits = bits; have = true;
          if (set.find_or_add(bits, (int)mine.size()) == (int)mine.size()) mine.push_back(bits);
          if (mine.size() > 256) return;
        }
    });
    for (int t = 0; t < NT && few; t++) {
      if (seen[(size_t)t].size() > 256) few = false;
      for (size_t i = 0; i < seen[(size_t)t].size() && few; i++)
        if (vset.find_or_add(seen[(size_t)t][i], (int)pretab.size()) == (int)pretab.size()) {
          if (pretab.size() == 256) { few = false; break; }
          double v;
          memcpy(&v, &seen[(size_t)t][i], 8);
          pretab.push_back(v);
        }
    }
  }
  tm.lap("distinct values");

  /* row blocks: consecutive rows, at most H rows and E entries.  A row longer than E belongs to a
   * block (which stores its diagonal term) but its entries are cut into pieces of at most E, each
   * a unit of its own whose sum is added to the row afterwards. */
  std::vector<bins_block> blocks;
  for (int64_t r = 0; r < nrows;) {
    const int64_t start = r;
    int64_t ent = 0;
    while (r < nrows && r - start < H) {
      int64_t l = rowptr[r + 1] - rowptr[r];
      if (l > E) l = 0;
      if (ent + l > E && r > start) break;
      ent += l; r++;
    }
    blocks.push_back({(int32_t)start, (int32_t)(r - start), -1, -1, -1});
  }
  B->long_first.push_back(0);
  for (int64_t r = 0; r < nrows; r++) {
    const int64_t l = rowptr[r + 1] - rowptr[r];
    if (l <= E) continue;
    const int64_t np = (l + E - 1) / E;
    for (int64_t k = 0; k < np; k++) {
      blocks.push_back({(int32_t)r, 1, (int32_t)B->piece_row.size(), rowptr[r] + k * l / np, rowptr[r] + (k + 1) * l / np});
      B->piece_row.push_back((int32_t)r);
    }
    B->long_rows.push_back((int32_t)r);
    B->long_first.push_back((int32_t)B->piece_row.size());
  }
  const int64_t NB = (int64_t)blocks.size();
  /* the units in NT consecutive parts of about as many entries each: part t is ub[t] .. ub[t+1] */
  std::vector<int64_t> ub((size_t)NT + 1, NB);
  {
    std::vector<int64_t> w((size_t)NB + 1, 0);
    for (int64_t b = 0; b < NB; b++) {
      const bins_block &k = blocks[(size_t)b];
      w[(size_t)b + 1] = w[(size_t)b] + (k.e0 >= 0 ? k.e1 - k.e0 : rowptr[(int64_t)k.r0 + k.nr] - rowptr[k.r0]);
    }
    for (int t = 0; t < NT; t++)
      ub[(size_t)t] = t == 0 ? 0 : (int64_t)(std::lower_bound(w.begin(), w.end(), w[(size_t)NB] / NT * t) - w.begin());
    for (int t = 1; t <= NT; t++) ub[(size_t)t] = std::max(std::min(ub[(size_t)t], NB), ub[(size_t)t - 1]);
  }
  tm.lap("row blocks");

  /* cells: per unit the entries per column block; cells of at least T entries (any cell of a
   * piece) are binned: one run each, listed unit by unit, within a unit by column block */
  ksg_hvec<bins_seg> S;
  std::vector<int64_t> segb((size_t)NB + 1, 0);            /* runs of unit b: S[segb[b] .. segb[b+1]) */
  std::vector<int64_t> sfirst((size_t)NT + 1, 0);          /* part t's runs: S[sfirst[t] .. sfirst[t+1]) */
  {
    std::vector<std::vector<bins_seg>> St((size_t)NT);
    std::vector<int64_t> treal((size_t)NT, 0);
    ksg_parallel(NT, [&](int t, int) {
      std::vector<bins_seg> &Sl = St[(size_t)t];
      std::vector<int32_t> cnt((size_t)NC, 0), touched;
      int64_t nreal = 0;
      for (int64_t b = ub[(size_t)t]; b < ub[(size_t)t + 1]; b++) {
        const bins_block &k = blocks[(size_t)b];
        touched.clear();
        const size_t before = Sl.size();
        for (int64_t r = k.r0; r < (int64_t)k.r0 + k.nr; r++) {
          int64_t pa = rowptr[r], pb = rowptr[r + 1];
          if (k.e0 >= 0) { pa = k.e0; pb = k.e1; }
          else if (pb - pa > E) continue;
          for (int64_t p = pa; p < pb; p++) {
            if (fold && (int64_t)colidx[p] == r + dshift) continue;
            const int32_t cb = colidx[p] >> wshift;
            if (cnt[(size_t)cb]++ == 0) touched.push_back(cb);
          }
        }
        std::sort(touched.begin(), touched.end());
        for (int32_t cb : touched) {
          if (cnt[(size_t)cb] >= T || k.e0 >= 0) { Sl.push_back({cb, cnt[(size_t)cb], (int32_t)b, 0, 0}); nreal += cnt[(size_t)cb]; }
          cnt[(size_t)cb] = 0;
        }
        segb[(size_t)b + 1] = (int64_t)(Sl.size() - before);
      }
      treal[(size_t)t] = nreal;
    });
    int64_t nreal = 0;
    for (int t = 0; t < NT; t++) {
      sfirst[(size_t)t + 1] = sfirst[(size_t)t] + (int64_t)St[(size_t)t].size();
      nreal += treal[(size_t)t];
    }
    for (int64_t b = 0; b < NB; b++) segb[(size_t)b + 1] += segb[(size_t)b];
    B->nreal = nreal;
    if (B->nreal * 2 < nnz || (uint64_t)sfirst[(size_t)NT] >= (uint64_t)UINT32_MAX / 4 - (uint64_t)NC) { *B = bins_host(); return; }
    S.resize((size_t)sfirst[(size_t)NT]);
    ksg_parallel(NT, [&](int t, int) {
      if (!St[(size_t)t].empty()) memcpy(S.data() + sfirst[(size_t)t], St[(size_t)t].data(), St[(size_t)t].size() * sizeof(bins_seg));
      std::vector<bins_seg>().swap(St[(size_t)t]);
    });
  }
  const int64_t nruns = (int64_t)S.size();
  tm.lap("cells");
  B->W = W; B->H = H; B->nrows = nrows; B->fold_diag = fold; B->dshift = dshift;

  /* the expand order: by column block, within it by unit, within a cell by row and column.  First
   * the raw order (the entries alone): where every run starts in it, and the run's number j among
   * the runs of its column block */
  std::vector<int64_t> rawpos((size_t)NC + 1, 0), runpos((size_t)NC + 1, 0);
  {
    std::vector<std::vector<int64_t>> he((size_t)NT), hs((size_t)NT);   /* per part: entries and runs per column block, then where they start */
    ksg_parallel(NT, [&](int t, int) {
      he[(size_t)t].assign((size_t)NC, 0); hs[(size_t)t].assign((size_t)NC, 0);
      for (int64_t i = sfirst[(size_t)t]; i < sfirst[(size_t)t + 1]; i++) { he[(size_t)t][(size_t)S[(size_t)i].c] += S[(size_t)i].real; hs[(size_t)t][(size_t)S[(size_t)i].c]++; }
    });
    for (int64_t cb = 0; cb < NC; cb++) {
      int64_t fe = rawpos[(size_t)cb], fs = runpos[(size_t)cb];
      for (int t = 0; t < NT; t++) {
        const int64_t e = he[(size_t)t][(size_t)cb], s = hs[(size_t)t][(size_t)cb];
        he[(size_t)t][(size_t)cb] = fe; hs[(size_t)t][(size_t)cb] = fs;
        fe += e; fs += s;
      }
      rawpos[(size_t)cb + 1] = fe; runpos[(size_t)cb + 1] = fs;
    }
    ksg_parallel(NT, [&](int t, int) {
      std::vector<int64_t> &fpos = he[(size_t)t], &fseg = hs[(size_t)t];
      for (int64_t i = sfirst[(size_t)t]; i < sfirst[(size_t)t + 1]; i++) {
        bins_seg &s = S[(size_t)i];
        s.j = fseg[(size_t)s.c]++;
        s.praw = fpos[(size_t)s.c];
        fpos[(size_t)s.c] += s.real;
      }
    });
  }
  /* the runs by column block: run j is S[sof[j]] */
  ksg_hvec<int64_t> sof((size_t)nruns);
  ksg_parallel(NT, [&](int t, int) {
    for (int64_t i = sfirst[(size_t)t]; i < sfirst[(size_t)t + 1]; i++) sof[(size_t)S[(size_t)i].j] = i;
  });
  tm.lap("raw expand order");

  /* the entries in the raw order: column within the column block, row within the unit, value; what
   * is not binned goes to the CSR of the remaining entries (the row blocks come first and in row
   * order, so that CSR is in row order too); the diagonal as a dense vector */
  ksg_hvec<uint16_t> rcol((size_t)B->nreal), rrow((size_t)B->nreal);
  ksg_hvec<uint8_t> rvi;
  ksg_hvec<double> rva;
  uint8_t nused[256] = {0};                               /* values of the table that binned entries have */
  const bool one = few && pretab.size() == 1;             /* a single value: nothing to store per entry */
  if (one) nused[0] = 1;
  else if (few) rvi.resize((size_t)B->nreal);
  else rva.resize((size_t)B->nreal);
  if (fold) B->dv.resize((size_t)nrows);
  {
    std::vector<std::vector<int32_t>> tci((size_t)NT), trows((size_t)NT);
    std::vector<std::vector<double>> tva((size_t)NT);
    std::vector<std::vector<int64_t>> trp((size_t)NT);      /* ends of the part's rows, counted from the part's first entry */
    std::vector<std::vector<uint8_t>> tused((size_t)NT, std::vector<uint8_t>(256, 0));
    ksg_parallel(NT, [&](int t, int) {
      std::vector<int64_t> segid((size_t)NC, -1);
      std::vector<int32_t> fill((size_t)NC, 0);
      uint8_t *used = tused[(size_t)t].data();             /* values of the table that binned entries of this part have */
      uint64_t lastbits = 0;
      int lastcode = -1;
      std::vector<int32_t> &lc = tci[(size_t)t], &lr = trows[(size_t)t];
      std::vector<double> &lv = tva[(size_t)t];
      std::vector<int64_t> &lp = trp[(size_t)t];
      for (int64_t b = ub[(size_t)t]; b < ub[(size_t)t + 1]; b++) {
        const bins_block &k = blocks[(size_t)b];
        for (int64_t i = segb[(size_t)b]; i < segb[(size_t)b + 1]; i++) { segid[(size_t)S[(size_t)i].c] = i; fill[(size_t)S[(size_t)i].c] = 0; }
        for (int64_t r = k.r0; r < (int64_t)k.r0 + k.nr; r++) {
          int64_t pa = rowptr[r], pb = rowptr[r + 1];
          const bool longrow = k.e0 < 0 && pb - pa > E;     /* its entries are binned with its pieces */
          if (k.e0 >= 0) { pa = k.e0; pb = k.e1; }
          const size_t before = lc.size();
          for (int64_t p = pa; p < pb; p++) {
            if (fold && (int64_t)colidx[p] == r + dshift) { if (k.e0 < 0) B->dv[(size_t)r] = val[p]; continue; }
            if (longrow) continue;
            const int32_t cb = colidx[p] >> wshift;
            const int64_t i = segid[(size_t)cb];
            if (i < 0) { lc.push_back(colidx[p]); lv.push_back(val[p]); continue; }
            const int64_t q = S[(size_t)i].praw + fill[(size_t)cb]++;
            rcol[(size_t)q] = (uint16_t)(colidx[p] & (W - 1));
            rrow[(size_t)q] = (uint16_t)(r - k.r0);
            if (one) {
            } else if (few) {
              uint64_t bits;
              memcpy(&bits, &val[p], 8);
              if (lastcode < 0 || bits != lastbits) { lastcode = vset.find(bits); lastbits = bits; used[lastcode] = 1; }
              rvi[(size_t)q] = (uint8_t)lastcode;
            } else {
              rva[(size_t)q] = val[p];
            }
          }
          if (lc.size() > before) { lr.push_back((int32_t)r); lp.push_back((int64_t)lc.size()); }
        }
        for (int64_t i = segb[(size_t)b]; i < segb[(size_t)b + 1]; i++) segid[(size_t)S[(size_t)i].c] = -1;
      }
    });
    for (int t = 0; t < NT; t++) for (int i = 0; i < 256; i++) if (tused[(size_t)t][(size_t)i]) nused[i] = 1;
    /* the remaining entries of the parts, one behind the other */
    std::vector<int64_t> eoff((size_t)NT + 1, 0), roff((size_t)NT + 1, 0);
    for (int t = 0; t < NT; t++) { eoff[(size_t)t + 1] = eoff[(size_t)t] + (int64_t)tci[(size_t)t].size(); roff[(size_t)t + 1] = roff[(size_t)t] + (int64_t)trows[(size_t)t].size(); }
    lci.resize((size_t)eoff[(size_t)NT]); lva.resize((size_t)eoff[(size_t)NT]);
    lrows.resize((size_t)roff[(size_t)NT]); lrp.resize((size_t)roff[(size_t)NT] + 1);
    lrp[0] = 0;
    ksg_parallel(NT, [&](int t, int) {
      if (!tci[(size_t)t].empty()) {
        memcpy(lci.data() + eoff[(size_t)t], tci[(size_t)t].data(), tci[(size_t)t].size() * sizeof(int32_t));
        memcpy(lva.data() + eoff[(size_t)t], tva[(size_t)t].data(), tva[(size_t)t].size() * sizeof(double));
      }
      for (size_t i = 0; i < trows[(size_t)t].size(); i++) {
        lrows[(size_t)roff[(size_t)t] + i] = trows[(size_t)t][i];
        lrp[(size_t)roff[(size_t)t] + i + 1] = eoff[(size_t)t] + trp[(size_t)t][i];
      }
    });
  }
  tm.lap("entries, raw order");

  /* The expand pass adds up, before it writes, the products of neighbouring entries of one row (a
   * stretch: the entries of a cell come sorted by row, so the entries a row has in a column block
   * are neighbours), within the rows of 16 lanes its scan works in.  What it writes per run is one
   * sum per stretch, a slot; a run's slots are filled up to a multiple of A with the products of
   * fill entries (x = 0), so that a run starts and ends on whole 64-byte pieces of the array of
   * sums.  Where the 16-lane boundaries fall in a run depends on where the run starts in its
   * column block's stream, which depends on the fill of the runs before it: one walk per column
   * block, in the order of its runs, counts slots and fill; a second one, once the positions of
   * all runs in both orders are known, writes the arrays. */
  const int A = std::min(std::max(env_int("KSG_BINS_ALIGN", BINS_RUN_ALIGN), 1), 64);
  ksg_hvec<int32_t> rslots((size_t)nruns);                /* slots of run j, fill included */
  ksg_hvec<uint8_t> rfill((size_t)nruns);                 /* its fill entries */
  std::vector<int64_t> clen((size_t)NC, 0), cparts((size_t)NC, 0);
  /* the column blocks in CT consecutive parts of about as many entries each (the blocks of the hub
   * columns hold many times the entries of the others) */
  const int CT = (int)std::min<int64_t>(NT, NC);
  std::vector<int64_t> cbnd((size_t)CT + 1, NC);
  for (int t = 0; t <= CT; t++)
    cbnd[(size_t)t] = t == 0 ? 0 : t == CT ? NC : (int64_t)(std::lower_bound(rawpos.begin(), rawpos.end(), B->nreal / CT * t) - rawpos.begin());
  for (int t = 1; t <= CT; t++) cbnd[(size_t)t] = std::max(std::min(cbnd[(size_t)t], NC), cbnd[(size_t)t - 1]);
  ksg_parallel(CT, [&](int t, int) {
    for (int64_t cb = cbnd[(size_t)t]; cb < cbnd[(size_t)t + 1]; cb++) {
      int64_t q = 0, parts = 0;
      for (int64_t j = runpos[(size_t)cb]; j < runpos[(size_t)cb + 1]; j++) {
        const bins_seg &s = S[(size_t)sof[(size_t)j]];
        const uint16_t *rr = rrow.data() + s.praw;
        int64_t slots = 1;
        for (int32_t f = 1; f < s.real; f++) slots += (((q + f) & 15) == 0) || rr[f] != rr[f - 1];
        const int fl = (int)((A - slots % A) % A);
        rslots[(size_t)j] = (int32_t)(slots + fl); rfill[(size_t)j] = (uint8_t)fl;
        const int64_t cnt = s.real + fl;
        parts += (q + cnt - 1) / 64 - q / 64 + 1;
        q += cnt;
      }
      if (q % 64) parts++;                                 /* the entries that fill the last group form a run of their own */
      clen[(size_t)cb] = q; cparts[(size_t)cb] = parts;
    }
  });
  std::vector<int64_t> cpos((size_t)NC + 1, 0), partbase((size_t)NC + 1, 0);
  for (int64_t cb = 0; cb < NC; cb++) {
    cpos[(size_t)cb + 1] = cpos[(size_t)cb] + (clen[(size_t)cb] + 63) / 64 * 64;
    partbase[(size_t)cb + 1] = partbase[(size_t)cb] + cparts[(size_t)cb];
  }
  const int64_t nent1 = cpos[(size_t)NC], nparts = partbase[(size_t)NC], ngroups = nent1 / 64;
  /* the reduce order: unit by unit, within a unit by column block; pos2 of run S[i] */
  ksg_hvec<int64_t> rpos2((size_t)nruns);                 /* by run number j */
  std::vector<int64_t> bp0((size_t)NB), bp1((size_t)NB);
  {
    int64_t acc = 0;
    for (int64_t b = 0; b < NB; b++) {
      bp0[(size_t)b] = acc;
      for (int64_t i = segb[(size_t)b]; i < segb[(size_t)b + 1]; i++) { rpos2[(size_t)S[(size_t)i].j] = acc; acc += rslots[(size_t)S[(size_t)i].j]; }
      bp1[(size_t)b] = acc;
    }
    B->nent = acc;
  }
  if ((uint64_t)nparts >= (uint64_t)UINT32_MAX) { *B = bins_host(); lrows.clear(); lrp.clear(); lci.clear(); lva.clear(); return; }
  tm.lap("slots, both orders");

  ksg_hvec<uint16_t> &col16 = B->col16, &rowloc = B->rowloc;
  ksg_hvec<double> &val1 = B->val;
  ksg_hvec<uint8_t> &vidx1 = B->vidx;
  ksg_hvec<uint8_t> pad;                                  /* entries that stand for nothing: the fill of the runs and of the groups */
  col16.resize((size_t)nent1); rowloc.resize((size_t)B->nent);
  if (one) { }
  else if (few) vidx1.resize((size_t)nent1);
  else { val1.resize((size_t)nent1); pad.resize((size_t)nent1); }
  B->gfirst.resize((size_t)ngroups); B->runmask.resize((size_t)ngroups); B->rowmask.resize((size_t)ngroups);
  B->O.resize((size_t)nparts);
  ksg_parallel(CT, [&](int t, int) {
    for (int64_t cb = cbnd[(size_t)t]; cb < cbnd[(size_t)t + 1]; cb++) {
      int64_t p = cpos[(size_t)cb], part = partbase[(size_t)cb];
      uint64_t rm = 0, sm = 0;
      int gstarts = 0;                                     /* stretch starts of the current group so far */
      /* one entry of the stream: `start` opens a stretch (slot `at`), `newpart` a part whose first slot is `at`
       * (O holds that slot minus the stretch starts of the group up to and including the part's first:
       * a lane adds the starts up to itself) */
      auto put = [&](bool start, bool newpart, int64_t at, uint16_t c16, int vi, double vv, bool isfill) {
        const int lane = (int)(p & 63);
        const int64_t g = p >> 6;
        if (lane == 0) { rm = 0; sm = 0; gstarts = 0; newpart = true; start = true; }
        if ((lane & 15) == 0) start = true;
        if (start) gstarts++;
        if (newpart) {
          B->O[(size_t)part] = at - gstarts;
          if (lane == 0) B->gfirst[(size_t)g] = (uint32_t)part; else rm |= 1ull << lane;
          part++;
        }
        if (start) sm |= 1ull << lane;
        col16[(size_t)p] = c16;
        if (one) { }
        else if (few) vidx1[(size_t)p] = (uint8_t)vi;
        else { val1[(size_t)p] = vv; pad[(size_t)p] = isfill ? 1 : 0; }
        if (lane == 63) { B->runmask[(size_t)g] = rm; B->rowmask[(size_t)g] = sm; }
        p++;
      };
      for (int64_t j = runpos[(size_t)cb]; j < runpos[(size_t)cb + 1]; j++) {
        const bins_seg &s = S[(size_t)sof[(size_t)j]];
        const uint16_t *rr = rrow.data() + s.praw, *rc = rcol.data() + s.praw;
        const int32_t unr = blocks[(size_t)s.unit].nr;
        const int64_t base = rpos2[(size_t)j];
        int64_t slot = -1;
        for (int32_t f = 0; f < s.real; f++) {
          const bool start = f == 0 || (p & 15) == 0 || rr[f] != rr[f - 1];
          if (start) { slot++; rowloc[(size_t)(base + slot)] = rr[f]; }
          put(start, f == 0, base + slot, rc[f], one || !few ? 0 : rvi[(size_t)(s.praw + f)], few ? 0.0 : rva[(size_t)(s.praw + f)], false);
        }
        for (int f = 0; f < (int)rfill[(size_t)j]; f++) {     /* the fill: column W (x = 0 there), a slot each, any row of the unit */
          slot++;
          rowloc[(size_t)(base + slot)] = (uint16_t)(f % unr);
          put(true, false, base + slot, (uint16_t)W, 255, 0.0, true);
        }
      }
      /* the fill of the column block's last group: a run of its own whose sums land in the spare slots behind the last */
      for (int f = 0; p < cpos[(size_t)cb + 1]; f++) put(true, f == 0, B->nent + f, (uint16_t)W, 255, 0.0, true);
    }
  });
  { ksg_hvec<bins_seg>().swap(S); ksg_hvec<int64_t>().swap(sof); ksg_hvec<uint16_t>().swap(rcol); ksg_hvec<uint16_t>().swap(rrow);
    ksg_hvec<uint8_t>().swap(rvi); ksg_hvec<double>().swap(rva); }
  B->nparts = nparts;
  tm.lap("entries, final order");


  /* every unit stores its rows of y (or its piece's sum); the units with the most products are
   * started first */
  {
    std::vector<int64_t> order((size_t)NB);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return bp1[a] - bp0[a] > bp1[b] - bp0[b]; });
    B->binp0.resize((size_t)NB); B->binp1.resize((size_t)NB); B->r0s.resize((size_t)NB); B->nrs.resize((size_t)NB);
    for (int64_t i = 0; i < NB; i++) {
      const bins_block &k = blocks[(size_t)order[(size_t)i]];
      B->binp0[(size_t)i] = bp0[order[(size_t)i]]; B->binp1[(size_t)i] = bp1[order[(size_t)i]];
      B->r0s[(size_t)i] = k.e0 >= 0 ? (int32_t)(nrows + k.piece) : k.r0;
      B->nrs[(size_t)i] = k.nr;
    }
    B->nbins = NB;
  }

  tm.lap("unit order");
  /* values: one, at most 256 (one-byte indices into a table), or as they are; the padding entries
   * take no part */
  B->vmode = 2;
  if (few) {
    int count = 0, only = 0;
    for (int i = 0; i < 256; i++) if (nused[i]) { count++; only = i; }
    if (count == 1) { B->vmode = 0; B->vconst = pretab[(size_t)only]; ksg_hvec<uint8_t>().swap(B->vidx); }
    else { B->vmode = 1; B->vtab = pretab; B->vtab.resize(256, 0.0); }
  } else {
    /* more than 256 values among all entries, perhaps not among the binned ones */
    const int distinct = ksg_build_value_table(nent1, val1.data(), pad.data(), 256, B->vtab, B->vidx);
    if (distinct == 1) { B->vmode = 0; B->vconst = B->vtab[0]; }
    else if (distinct > 1) B->vmode = 1;
    if (B->vmode != 1) { ksg_hvec<uint8_t>().swap(B->vidx); B->vtab.clear(); }
    if (B->vmode != 2) ksg_hvec<double>().swap(val1);
  }

  tm.lap("value table");
  /* expand tasks: a column block's groups in pieces of at most TG groups, the longest first */
  /* A task loads its column block's slice of x (128 KiB) before it streams its entries: the larger
   * the tasks, the less of that, as long as there are enough of them to balance 256 CUs.  Unless set,
   * the largest of 8192, 4096, 2048 groups that leaves 2048 tasks (R-MAT scale 26, ~4250 groups per
   * column block: 6.54 ms with 2048 groups per task, 6.45 with 4096, 6.42 with 8192 or 16384; scale 24
   * the same for all; scale 22 has 256 column blocks in all). */
  int64_t TG = env_int("KSG_BINS_TASK_GROUPS", 0);
  if (TG <= 0)
    for (TG = 8192; TG > 2048; TG /= 2) {
      int64_t nt = 0;
      for (int64_t cb = 0; cb < NC; cb++) nt += (cpos[cb + 1] / 64 - cpos[cb] / 64 + TG - 1) / TG;
      if (nt >= 2048) break;
    }
  {
    std::vector<int64_t> t3;
    for (int64_t cb = 0; cb < NC; cb++)
      for (int64_t g = cpos[cb] / 64; g < cpos[cb + 1] / 64; g += TG) {
        t3.push_back(cb); t3.push_back(g); t3.push_back(std::min(g + TG, cpos[cb + 1] / 64));
      }
    const int64_t nt = (int64_t)t3.size() / 3;
    std::vector<int64_t> order((size_t)nt);
    std::iota(order.begin(), order.end(), (int64_t)0);
    /* with column stages: the tasks of a stage together, stage by stage -- a column block belongs to
     * the last stage any of its columns needs; within a stage the longest first */
    std::vector<int> tstage((size_t)nt, 0);
    if (colstage && colstage->nstages > 1) {
      for (int64_t i = 0; i < nt; i++) {
        const int64_t c0 = t3[(size_t)(3 * i)] * W, c1 = std::min<int64_t>(c0 + W, ncols);
        int st = 0;
        for (size_t k = 0; k + 1 < colstage->bound.size(); k++)
          if (colstage->bound[k] < c1 && colstage->bound[k + 1] > c0) st = std::max(st, colstage->stage[k]);
        tstage[(size_t)i] = st;
      }
    }
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
      if (tstage[(size_t)a] != tstage[(size_t)b]) return tstage[(size_t)a] < tstage[(size_t)b];
      return t3[3 * a + 2] - t3[3 * a + 1] > t3[3 * b + 2] - t3[3 * b + 1];
    });
    B->tasks.resize(t3.size());
    for (int64_t i = 0; i < nt; i++)
      for (int k = 0; k < 3; k++) B->tasks[(size_t)(3 * i + k)] = t3[(size_t)(3 * order[(size_t)i] + k)];
    B->ntasks = nt;
    if (colstage && colstage->nstages > 1) {
      B->task_stage.assign((size_t)colstage->nstages + 1, nt);
      B->task_stage[0] = 0;
      for (int64_t i = nt - 1; i >= 0; i--) B->task_stage[(size_t)tstage[(size_t)order[(size_t)i]]] = i;
      for (int st = colstage->nstages - 1; st >= 1; st--)      /* a stage without tasks starts where the next one does */
        if (B->task_stage[(size_t)st] > B->task_stage[(size_t)st + 1]) B->task_stage[(size_t)st] = B->task_stage[(size_t)st + 1];
      B->task_stage[0] = 0;
    }
  }
  tm.lap("tasks");
  B->nent1 = nent1; B->nseg = nruns;
}

/* y = B x + d .* x on the host, walking the arrays the way the two kernels do (the lanes of a
 * group find their run from the group's first run and the run flags below them; one sum per row
 * block in the order of the product stream). */
static void bins_emulate(const bins_host &B, const double *x, int64_t ncols, double *y) {
  if (!B.nent) return;
  std::vector<double> xe((size_t)B.nent + 64, 0.0), xs((size_t)B.W + 1, 0.0), acc((size_t)B.H);
  for (int64_t t = 0; t < B.ntasks; t++) {
    const int64_t cb = B.tasks[3 * t];
    for (int i = 0; i < B.W; i++) xs[i] = cb * B.W + i < ncols ? x[cb * B.W + i] : 0.0;
    for (int64_t g = B.tasks[3 * t + 1]; g < B.tasks[3 * t + 2]; g++) {
      const uint64_t rm = B.runmask[(size_t)g], sm = B.rowmask[(size_t)g];
      double a[64], a0[64];
      for (int l = 0; l < 64; l++) {
        const int64_t p = g * 64 + l;
        volatile double prod = xs[B.col16[(size_t)p] & 0x7fffu] * (B.vmode == 0 ? B.vconst : B.vmode == 1 ? B.vtab[B.vidx[(size_t)p]] : B.val[(size_t)p]);
        a[l] = prod;
      }
      uint64_t F = sm;                                       /* a stretch starts in the window of N lanes that ends at the lane */
      for (int N = 1; N < 16 && ~F; N *= 2) {                /* the segmented scan over rows of 16 lanes, step by step as the lanes do it; a step no lane takes part in is left out */
        memcpy(a0, a, sizeof(a));
        for (int l = 0; l < 64; l++) {
          const bool take = (l & 15) >= N && !((F >> l) & 1ull);
          volatile double sum = a0[l] + (take ? a0[l - N] : 0.0);
          a[l] = sum;
        }
        F |= F << N;
      }
      for (int l = 0; l < 64; l++) {
        if (l != 63 && !((sm >> (l + 1)) & 1ull)) continue;  /* not the last lane of its stretch */
        const uint64_t upto = l == 63 ? ~0ull : ((2ull << l) - 1ull);   /* lanes 0..l */
        const int64_t part = (int64_t)B.gfirst[(size_t)g] + __builtin_popcountll(rm & upto);   /* bit 0 of rm is never set */
        xe[(size_t)(B.O[(size_t)part] + __builtin_popcountll(sm & upto))] = a[l];
      }
    }
  }
  std::vector<double> part(B.piece_row.size() + 1, 0.0);
  for (int64_t b = 0; b < B.nbins; b++) {
    const int64_t r0 = B.r0s[(size_t)b];
    if (r0 >= B.nrows) {                                   /* a piece of a long row: 64 lane sums of 4 partial sums each, then the wave's tree */
      double t[64][4];
      for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) t[l][j] = 0.0;
      const int64_t p0 = B.binp0[(size_t)b], p1 = B.binp1[(size_t)b];
      for (int l = 0; l < 64; l++) {
        int64_t p = p0 + l;
        for (; p + 64 * 3 < p1; p += 64 * 4) for (int j = 0; j < 4; j++) t[l][j] += xe[(size_t)(p + 64 * j)];
        for (; p < p1; p += 64) t[l][0] += xe[(size_t)p];
      }
      double sl[128];
      for (int l = 0; l < 64; l++) { sl[l] = (t[l][0] + t[l][1]) + (t[l][2] + t[l][3]); sl[64 + l] = 0.0; }
      for (int off = 32; off > 0; off >>= 1) for (int l = 0; l < 64; l++) sl[l] += l + off < 64 ? sl[l + off] : sl[l];
      part[(size_t)(r0 - B.nrows)] = sl[0];
      continue;
    }
    for (int i = 0; i < B.nrs[(size_t)b]; i++) acc[i] = 0.0;
    for (int64_t p = B.binp0[(size_t)b]; p < B.binp1[(size_t)b]; p++) acc[B.rowloc[(size_t)p]] += xe[(size_t)p];
    for (int i = 0; i < B.nrs[(size_t)b]; i++)
      y[r0 + i] = B.fold_diag ? fma(B.dv[(size_t)(r0 + i)], x[r0 + i + B.dshift], acc[i]) : acc[i];
  }
  for (size_t j = 0; j < B.long_rows.size(); j++) {
    double sum = 0.0;
    for (int k = B.long_first[j]; k < B.long_first[j + 1]; k++) sum += part[(size_t)k];
    y[B.long_rows[j]] += sum;
  }
}

/* The binned part of a block with the sizes of the row blocks chosen: KSG_BINS_H rows (4096 unless
 * set) and E entries at most. */
static void bins_build_sized(bins_host *B, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *colidx,
                             const double *val, bool has_diag, int64_t dshift, std::vector<int32_t> &lrows, std::vector<int64_t> &lrp,
                             std::vector<int32_t> &lci, std::vector<double> &lva, int maxW = 16384,
                             const ksg_colstages *colstage = NULL) {
  /* Rows per row block: the reduce pass keeps a block's sums in LDS, 8 H bytes per wave.  With the
   * sums of neighbouring entries formed in the expand pass the reduce pass has little to add per
   * row, and larger blocks -- fewer, better filled cells, fewer runs -- pay although only five
   * waves fit a CU: R-MAT scale 26 4.93 ms per product with H = 2048, 4.79 with 4096, 5.51 with
   * 8192; scale 24 1.116 / 1.100 / 1.37 ms; scale 22 0.258 ms either way. */
  const int H = env_int("KSG_BINS_H", 4096);
  /* Most entries per row block.  Where rows are long (the hub rows of a power-law operator) it is E
   * and not H that ends a row block, and the fewer rows a block has, the thinner its cells: a larger
   * E bins more entries in longer runs.  But one wave sums a row block, so the largest blocks must
   * stay well below the time of the whole pass.  Unless set: 1e-3 of the block's entries, as a power
   * of two between 65536 and 1048576.  R-MAT scale 26 (1.13e9 entries, H = 4096): 4.87 ms per product
   * with E = 262144, 4.79 with 524288, 4.77 with 1048576; scale 24 (2.8e8 entries): 1.165 ms with
   * 65536, 1.127 with 131072, 1.100 with 262144. */
  int64_t E = env_int("KSG_BINS_E", 0);
  if (E <= 0) {
    E = 65536;
    while (E < 1048576 && (double)(2 * E) <= 1e-3 * (double)rowptr[nrows]) E *= 2;
  }
  bins_build_host(B, nrows, ncols, rowptr, colidx, val, has_diag, dshift, lrows, lrp, lci, lva, H, E, maxW, colstage);
}

/* Host-side check of the binned layout, no device needed: y = A x for the CSR block given, formed
 * as the library forms it (the binned part and the diagonal through the same walk of the same arrays
 * as the kernels, then the remaining entries row by row in storage order); info = {binned entries,
 * runs, row blocks, value mode (0 one value, 1 one-byte indices, 2 eight-byte values), entries left
 * in the CSR, 1 if the diagonal is folded into the reduce pass}.  The tuning variables
 * KSG_BINS_W/H/E/T apply as in ksg_mat_create_csr. */
extern "C" int ksg_bins_host_check(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *colidx, const double *val,
                                   int has_diag, int64_t dshift, const double *x, double *y, int64_t info[6]) {
  if (!rowptr || !x || !y || nrows < 0 || ncols < 0 || (rowptr[nrows] && (!colidx || !val)))
    KSG_FAIL(KSG_ERR_ARG, "ksg_bins_host_check: bad arguments");
  if (rowptr[0] != 0) KSG_FAIL(KSG_ERR_ARG, "ksg_bins_host_check: rowptr[0] must be 0");
  for (int64_t r = 0; r < nrows; r++) {
    if (rowptr[r + 1] < rowptr[r]) KSG_FAIL(KSG_ERR_ARG, "ksg_bins_host_check: rowptr decreases at row %lld", (long long)r);
    for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++)
      if (colidx[p] < 0 || (int64_t)colidx[p] >= ncols)
        KSG_FAIL(KSG_ERR_ARG, "ksg_bins_host_check: column %d of row %lld outside [0,%lld)", colidx[p], (long long)r, (long long)ncols);
  }
  bins_host B;
  std::vector<int32_t> lrows, lci;
  std::vector<int64_t> lrp;
  std::vector<double> lva;
  bins_build_sized(&B, nrows, ncols, rowptr, colidx, val, has_diag != 0, dshift, lrows, lrp, lci, lva);
  if (!B.nent) {
    for (int64_t r = 0; r < nrows; r++) {
      double s = 0.0;
      for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) s += val[p] * x[colidx[p]];
      y[r] = s;
    }
  } else {
    bins_emulate(B, x, ncols, y);
    for (size_t i = 0; i < lrows.size(); i++) {
      double s = 0.0;
      for (int64_t p = lrp[i]; p < lrp[i + 1]; p++) s += lva[(size_t)p] * x[lci[(size_t)p]];
      y[lrows[i]] += s;
    }
  }
  if (info) {
    info[0] = B.nreal; info[1] = B.nseg; info[2] = B.nbins; info[3] = B.vmode;
    info[4] = B.nent ? (int64_t)lci.size() : rowptr[nrows]; info[5] = B.fold_diag ? 1 : 0;
  }
  return KSG_OK;
}

/* The binned part on the device.  rowmap (may be NULL): the rows of y the block's rows stand for. */
static int bins_to_device(ksg_ctx c, ksg_bins *B, const bins_host &Hb, const int32_t *rowmap) {
  B->threads = env_int("KSG_BINS_THREADS", 1024) / 64 * 64;
  if (B->threads < 64) B->threads = 64;
  if (B->threads > BINS_P1_THREADS) B->threads = BINS_P1_THREADS;
  B->nent = Hb.nent; B->nreal = Hb.nreal; B->nent1 = Hb.nent1;
  if (!Hb.task_stage.empty()) {
    B->nstages = (int)Hb.task_stage.size() - 1;
    B->task_stage = (int64_t *)malloc(Hb.task_stage.size() * sizeof(int64_t));
    if (!B->task_stage) KSG_FAIL(KSG_ERR_MEM, "binned operator: out of host memory");
    memcpy(B->task_stage, Hb.task_stage.data(), Hb.task_stage.size() * sizeof(int64_t));
  }
  B->nseg = Hb.nseg; B->nparts = Hb.nparts; B->nbins = Hb.nbins; B->ntasks = Hb.ntasks;
  B->nrows = Hb.nrows; B->dshift = Hb.dshift; B->W = Hb.W; B->H = Hb.H; B->vmode = Hb.vmode; B->vconst = Hb.vconst;
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(bins_upload(&B->col16, Hb.col16.data(), Hb.col16.size()));
  if (B->vmode == 1) { KSG_CHK(bins_upload(&B->vidx, Hb.vidx.data(), Hb.vidx.size())); KSG_CHK(bins_upload(&B->vtab, Hb.vtab.data(), Hb.vtab.size())); }
  if (B->vmode == 2) KSG_CHK(bins_upload(&B->val, Hb.val.data(), Hb.val.size()));
  KSG_CHK(bins_upload(&B->gfirst, Hb.gfirst.data(), Hb.gfirst.size()));
  KSG_CHK(bins_upload(&B->O, Hb.O.data(), Hb.O.size()));
  KSG_CHK(bins_upload(&B->runmask, Hb.runmask.data(), Hb.runmask.size()));
  KSG_CHK(bins_upload(&B->rowmask, Hb.rowmask.data(), Hb.rowmask.size()));
  KSG_CHK(bins_upload(&B->tasks, Hb.tasks.data(), Hb.tasks.size()));
  KSG_CHK(bins_upload(&B->rowloc, Hb.rowloc.data(), Hb.rowloc.size()));
  KSG_CHK(bins_upload(&B->binp0, Hb.binp0.data(), Hb.binp0.size()));
  KSG_CHK(bins_upload(&B->binp1, Hb.binp1.data(), Hb.binp1.size()));
  KSG_CHK(bins_upload(&B->bin_r0, Hb.r0s.data(), Hb.r0s.size()));
  KSG_CHK(bins_upload(&B->bin_nr, Hb.nrs.data(), Hb.nrs.size()));
  if (Hb.fold_diag) KSG_CHK(bins_upload(&B->dv, Hb.dv.data(), Hb.dv.size()));
  B->nlong = (int64_t)Hb.long_rows.size(); B->npieces = (int64_t)Hb.piece_row.size();
  if (B->nlong) {
    /* k_bins_long_rows adds into y: with a row list, the rows of y the long rows stand for */
    std::vector<int32_t> ly(Hb.long_rows);
    if (rowmap) for (size_t j = 0; j < ly.size(); j++) ly[j] = rowmap[ly[j]];
    KSG_CHK(bins_upload(&B->long_rows, ly.data(), ly.size()));
    KSG_CHK(bins_upload(&B->long_first, Hb.long_first.data(), Hb.long_first.size()));
    KSG_CHK(bins_upload(&B->piece_row, Hb.piece_row.data(), Hb.piece_row.size()));
    hipError_t e2 = hipMalloc((void **)&B->part, (size_t)B->npieces * sizeof(double));
    if (e2 != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "binned operator: sums of the pieces of long rows: %s", hipGetErrorString(e2));
  }
  hipError_t e = hipMalloc((void **)&B->xe, (size_t)(B->nent + 64) * sizeof(double));
  if (e != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "binned operator: %zu bytes of products: %s", (size_t)(B->nent + 64) * sizeof(double), hipGetErrorString(e));
  return KSG_OK;
}

/* Columns per column block the device's LDS can hold (a power of two, at most 16384), and, once per
 * device, the expand kernels' permission to use that much: the slice of x may exceed the 64 KiB a
 * kernel gets without asking.  The attribute belongs to the kernel on the device, not to an
 * operator, so it is set to the largest slice any operator on this device may ask for. */
static int bins_device_max_w(ksg_ctx c, int *maxW) {
  static int known[64];                                   /* per device: 0 = not asked yet, -1 = no binned form here */
  *maxW = 0;
  if (c->device < 0 || c->device >= 64) return KSG_OK;
  if (!known[c->device]) {
    KSG_HIP(hipSetDevice(c->device));
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, c->device) != hipSuccess || optin <= 0)
      if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess) optin = 0;
    int w = 16384;
    while (w >= 1024 && (w + 1) * (int)sizeof(double) > optin) w >>= 1;
    known[c->device] = -1;
    if (w >= 1024) {
      const int lds = (w + 1) * (int)sizeof(double);
      if (hipFuncSetAttribute((const void *)k_bins_expand<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
          hipFuncSetAttribute((const void *)k_bins_expand<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
          hipFuncSetAttribute((const void *)k_bins_expand<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess)
        known[c->device] = w;
      else (void)hipGetLastError();
    }
  }
  if (known[c->device] > 0) *maxW = known[c->device];
  return KSG_OK;
}

/* The same split, with the binned part uploaded as *B.  rowmap (may be NULL): the rows of y the
 * block's rows stand for (ksg_bins_mult is then given the same list on the device). */
int ksg_bins_build(ksg_ctx c, ksg_bins *B, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *colidx,
                   const double *val, bool has_diag, int64_t dshift, std::vector<int32_t> &lrows, std::vector<int64_t> &lrp,
                   std::vector<int32_t> &lci, std::vector<double> &lva, const int32_t *rowmap, const ksg_colstages *colstage) {
  memset(B, 0, sizeof(*B));
  lrows.clear(); lrp.clear(); lci.clear(); lva.clear();
  int maxW = 0;
  KSG_CHK(bins_device_max_w(c, &maxW));
  if (!maxW) return KSG_OK;                               /* too little LDS for a slice worth having: the block stays a CSR */
  bins_host Hb;
  bins_build_sized(&Hb, nrows, ncols, rowptr, colidx, val, has_diag, dshift, lrows, lrp, lci, lva, maxW, colstage);
  if (!Hb.nent) return KSG_OK;
  const int rc = bins_to_device(c, B, Hb, rowmap);
  if (rc) {
    /* an upload failed: nothing of the binned form is kept, the caller holds the block as a CSR (and
     * reports the shortage itself if that does not fit either) */
    ksg_bins_free(B);
    lrows.clear(); lrp.clear(); lci.clear(); lva.clear();
  }
  return KSG_OK;
}

/* The expand tasks of one stage of the block's column stages (stage < 0, or a block built without
 * stages: all of them) on `st`. */
int ksg_bins_expand(const ksg_bins *B, const double *x, int64_t ncols, hipStream_t st, int stage) {
  if (!B->nent) return KSG_OK;
  int64_t t0 = 0, t1 = B->ntasks;
  if (stage >= 0 && B->task_stage) {
    if (stage >= B->nstages) return KSG_OK;
    t0 = B->task_stage[stage]; t1 = B->task_stage[stage + 1];
  } else if (stage > 0) return KSG_OK;
  if (t1 <= t0) return KSG_OK;
  const size_t lds1 = ((size_t)B->W + 1) * sizeof(double);
  const dim3 g1((unsigned)(t1 - t0)), b1((unsigned)B->threads);
  const int64_t *tasks = B->tasks + 3 * t0;
  if (B->vmode == 0)
    hipLaunchKernelGGL(k_bins_expand<0>, g1, b1, lds1, st, tasks, B->col16, B->vidx, B->vtab, B->val, B->vconst, B->gfirst, B->runmask, B->rowmask, B->O, x, ncols, B->xe, B->W);
  else if (B->vmode == 1)
    hipLaunchKernelGGL(k_bins_expand<1>, g1, b1, lds1, st, tasks, B->col16, B->vidx, B->vtab, B->val, 1.0, B->gfirst, B->runmask, B->rowmask, B->O, x, ncols, B->xe, B->W);
  else
    hipLaunchKernelGGL(k_bins_expand<2>, g1, b1, lds1, st, tasks, B->col16, B->vidx, B->vtab, B->val, 1.0, B->gfirst, B->runmask, B->rowmask, B->O, x, ncols, B->xe, B->W);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

/* The sums of the slots the expand tasks wrote, per row block, on `st`: every row of the block is
 * stored (y = B x + d .* x); with a row list, y[list] += B x. */
int ksg_bins_finish(const ksg_bins *B, const double *x, double *y, hipStream_t st, const int32_t *d_rowmap) {
  if (!B->nent) return KSG_OK;
  const size_t lds2 = (size_t)B->H * sizeof(double);
  hipLaunchKernelGGL(k_bins_reduce, dim3((unsigned)B->nbins), dim3(64), lds2, st, B->binp0, B->binp1, B->bin_r0, B->bin_nr, B->rowloc, B->xe,
                     B->dv, x, B->dshift, y, B->nrows, B->part, d_rowmap);
  if (B->nlong)
    hipLaunchKernelGGL(k_bins_long_rows, dim3((unsigned)((B->nlong + KSG_BLOCK - 1) / KSG_BLOCK)), dim3(KSG_BLOCK), 0, st, B->long_rows,
                       B->long_first, (int)B->nlong, B->part, y);
  KSG_HIP(hipGetLastError());
  return KSG_OK;
}

int ksg_bins_mult(const ksg_bins *B, const double *x, int64_t ncols, double *y, hipStream_t st, const int32_t *d_rowmap) {
  KSG_CHK(ksg_bins_expand(B, x, ncols, st, -1));
  return ksg_bins_finish(B, x, y, st, d_rowmap);
}

/* Bytes one binned product moves: expand reads the 2-byte columns (+ value bytes), 20 bytes per
 * group of 64 (first part, part starts, stretch starts), one offset per part and the slice of x
 * per task, and writes one sum per slot; reduce reads the sums and their 2-byte rows, the diagonal
 * and x at the rows, and writes y. */
double ksg_bins_bytes(const ksg_bins *B) {
  if (!B->nent) return 0.0;
  const double vb = B->vmode == 2 ? 8.0 : B->vmode == 1 ? 1.0 : 0.0;
  return (2.0 + vb) * (double)B->nent1 + 8.0 * (double)B->nparts + 20.0 * (double)(B->nent1 / 64) +
         (8.0 * B->W + 24.0) * (double)B->ntasks + 8.0 * (double)B->nent +
         10.0 * (double)B->nent + 24.0 * (double)B->nbins + (B->dv ? 24.0 : 8.0) * (double)B->nrows;
}

/* The binned entries and the folded diagonal back on the host as (row, column, value) triplets,
 * in no particular order. */
int ksg_bins_to_host(const ksg_bins *B, std::vector<int32_t> &rows, std::vector<int32_t> &cols, std::vector<double> &vals) {
  rows.clear(); cols.clear(); vals.clear();
  if (!B->nent) return KSG_OK;
  std::vector<uint16_t> col16((size_t)B->nent1), rowloc((size_t)B->nent);
  std::vector<uint32_t> gfirst((size_t)(B->nent1 / 64));
  std::vector<uint64_t> runmask((size_t)(B->nent1 / 64)), rowmask((size_t)(B->nent1 / 64));
  std::vector<int64_t> O((size_t)B->nparts), tasks((size_t)B->ntasks * 3), binp0((size_t)B->nbins);
  std::vector<int32_t> r0s((size_t)B->nbins);
  std::vector<uint8_t> vidx;
  std::vector<double> tab, val1, dv;
#define DL(dst, src) KSG_HIP(hipMemcpy((dst).data(), (src), (dst).size() * sizeof((dst)[0]), hipMemcpyDeviceToHost))
  DL(col16, B->col16); DL(rowloc, B->rowloc); DL(gfirst, B->gfirst); DL(runmask, B->runmask); DL(rowmask, B->rowmask); DL(O, B->O);
  DL(tasks, B->tasks); DL(binp0, B->binp0); DL(r0s, B->bin_r0);
  if (B->vmode == 1) { vidx.resize((size_t)B->nent1); tab.resize(256); DL(vidx, B->vidx); DL(tab, B->vtab); }
  if (B->vmode == 2) { val1.resize((size_t)B->nent1); DL(val1, B->val); }
  if (B->dv) { dv.resize((size_t)B->nrows); DL(dv, B->dv); }
  std::vector<int32_t> piece_row((size_t)B->npieces);
  if (B->npieces) DL(piece_row, B->piece_row);
#undef DL
  /* the row blocks that hold sums, by their first slot */
  std::vector<int64_t> binp1((size_t)B->nbins);
  KSG_HIP(hipMemcpy(binp1.data(), B->binp1, binp1.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
  std::vector<int64_t> order;
  for (int64_t b = 0; b < B->nbins; b++) if (binp1[(size_t)b] > binp0[(size_t)b]) order.push_back(b);
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return binp0[(size_t)a] < binp0[(size_t)b]; });
  std::vector<int64_t> starts(order.size());
  for (size_t i = 0; i < order.size(); i++) starts[i] = binp0[(size_t)order[i]];
  rows.reserve((size_t)(B->nreal + (B->dv ? B->nrows : 0))); cols.reserve(rows.capacity()); vals.reserve(rows.capacity());
  int64_t cur_lo = 0, cur_hi = 0, cur_r0 = 0;
  for (int64_t t = 0; t < B->ntasks; t++) {
    const int64_t cb = tasks[3 * t];
    for (int64_t g = tasks[3 * t + 1]; g < tasks[3 * t + 2]; g++) {
      int64_t part = gfirst[(size_t)g], nst = 0;
      for (int l = 0; l < 64; l++) {
        const int64_t p = g * 64 + l;
        if (l && ((runmask[(size_t)g] >> l) & 1ull)) part++;
        if ((rowmask[(size_t)g] >> l) & 1ull) nst++;            /* lane 0 and every first lane of a part start a stretch */
        const int64_t q = O[(size_t)part] + nst;
        if (q >= B->nent || (col16[(size_t)p] & 0x7fffu) == (unsigned)B->W) continue;   /* fill of a group or of a run */
        /* the block that holds slot q (consecutive slots mostly share it) */
        if (q < cur_lo || q >= cur_hi) {
          const int64_t i = (int64_t)(std::upper_bound(starts.begin(), starts.end(), q) - starts.begin()) - 1;
          cur_lo = starts[(size_t)i]; cur_hi = binp1[(size_t)order[(size_t)i]]; cur_r0 = r0s[(size_t)order[(size_t)i]];
        }
        const int64_t r0 = cur_r0;
        rows.push_back(r0 >= B->nrows ? piece_row[(size_t)(r0 - B->nrows)] : (int32_t)(r0 + rowloc[(size_t)q]));
        cols.push_back((int32_t)(cb * B->W + (col16[(size_t)p] & 0x7fffu)));
        vals.push_back(B->vmode == 0 ? B->vconst : B->vmode == 1 ? tab[vidx[(size_t)p]] : val1[(size_t)p]);
      }
    }
  }
  for (int64_t r = 0; r < (B->dv ? B->nrows : 0); r++) { rows.push_back((int32_t)r); cols.push_back((int32_t)(r + B->dshift)); vals.push_back(dv[(size_t)r]); }
  return KSG_OK;
}
