/* Set-up of a CSR operator: validation, the choice of the form every block is held in (row tiles,
 * value table, packed hot columns, row codes, binned part), the conversion on host threads, the
 * upload, and -- with several ranks -- the split into the diagonal and the ghost block and the
 * halo plan (what PETSc's MatAssemblyEnd does for an MPIAIJ matrix behind MatMult,
 * src/sys/classes/st/impls/shift/shift.c:38).  Also the way back: the rows of an operator on the
 * host again, and its explicit transpose.  The kernels are in ksg_spmv.hip. */
#include <algorithm>
#include <stdlib.h>

#include "ksg_internal.h"

/* ------------------------------------------------------------------------ */
/* host side: tiling, upload, halo plan                                       */
/* ------------------------------------------------------------------------ */
/* Lanes a row of `len` entries needs so that no lane adds more than 32 products in the
 * reduction phase of a tile (a power of two, at most a whole wave). */
static inline int lanes_for_row(int64_t len) {
  int g = 1;
  while (g < 64 && len > 32 * (int64_t)g) g <<= 1;
  return g;
}

/* Cuts the rows into tiles: consecutive rows whose entries fit the LDS tile, and few enough
 * of them that the longest row of the tile still gets its lanes (rows * lanes <= workgroup
 * size; up to KSG_TILE_ROWS rows when one lane per row is enough).  Power-law matrices thus
 * get short tiles around their hub rows instead of one lane walking thousands of products
 * while the rest of the workgroup waits. */
static void build_tiles(int64_t nrows, const int64_t *rowptr, std::vector<int32_t> &tiles,
                        int64_t *n_stream, int64_t *n_group, int64_t *n_long) {
  const char *env = getenv("KSG_SPMV_BALANCE");
  const bool balance = !(env && env[0] == '0');
  tiles.clear();
  tiles.push_back(0);
  *n_stream = *n_group = *n_long = 0;
  int64_t r = 0;
  while (r < nrows) {
    int64_t e = r, maxlen = 0;
    const int64_t base = rowptr[r];
    while (e < nrows && e - r < KSG_TILE_ROWS && rowptr[e + 1] - base <= KSG_TILE_NNZ) {
      if (balance) {
        const int64_t len = rowptr[e + 1] - rowptr[e];
        const int g = lanes_for_row(len > maxlen ? len : maxlen);
        const int64_t allowed = g == 1 ? KSG_TILE_ROWS : KSG_BLOCK / g;
        if (e - r + 1 > allowed && e > r) break;
        if (len > maxlen) maxlen = len;
      }
      e++;
    }
    if (e == r) { e = r + 1; (*n_long)++; }          /* a single row longer than the tile */
    else if (e - r > KSG_BLOCK / 2) (*n_stream)++;
    else (*n_group)++;
    tiles.push_back((int32_t)e);
    r = e;
  }
}

template <class T>
static int upload(T **dst, const T *src, size_t count) {
  *dst = NULL;
  hipError_t e = hipMalloc((void **)dst, (count ? count : 1) * sizeof(T));
  if (e != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "matrix upload: %zu bytes: %s", count * sizeof(T), hipGetErrorString(e));
  if (count) KSG_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return KSG_OK;
}

/* Host threads for a pass over nnz entries (one below a million). */
static int setup_threads(int64_t nnz) { return nnz >= ((int64_t)1 << 20) ? ksg_host_threads() : 1; }

/* Entries per column of one block, counted once for the decisions that need them (binned form,
 * packed columns). */
struct col_counts {
  const int32_t *of = NULL;        /* the index array counted */
  std::vector<int32_t> cnt;
};

/* cnt[j] = entries in column j. */
static void count_columns(int64_t ncols, int64_t nnz, const int32_t *colidx, std::vector<int32_t> &cnt) {
  cnt.assign((size_t)ncols, 0);
  int32_t *c = cnt.data();
  ksg_parallel(setup_threads(nnz), [&](int t, int nt) {
    if (nt == 1) { for (int64_t p = 0; p < nnz; p++) c[colidx[p]]++; return; }
    for (int64_t p = nnz * t / nt; p < nnz * (t + 1) / nt; p++) __atomic_fetch_add(&c[colidx[p]], 1, __ATOMIC_RELAXED);
  });
}

static const std::vector<int32_t> &counts_of(col_counts *cc, col_counts &own, int64_t ncols, int64_t nnz, const int32_t *colidx) {
  col_counts &k = cc ? *cc : own;
  if (k.of != colidx || (int64_t)k.cnt.size() != ncols) { count_columns(ncols, nnz, colidx, k.cnt); k.of = colidx; }
  return k.cnt;
}

/* Chooses the columns whose x entries are kept packed (see KSG_OPT_HOT_COLS): those referenced
 * at least four times as often as the average, the most referenced first, at most `maxhot`,
 * and only when they hold 15 % of the entries and x is much larger than the packed array.
 * On success `hot` lists them and `relabeled` is a copy of colidx in which their entries carry
 * ncols + slot; both stay empty when nothing is packed. */
static void choose_hot_columns(int64_t ncols, int64_t nnz, const int32_t *colidx, int64_t maxhot,
                               std::vector<int32_t> &hot, std::vector<int32_t> &relabeled, col_counts *cc = NULL) {
  hot.clear(); relabeled.clear();
  /* the packed array has to stay much smaller than x: operators too small for the full limit
   * get the 2 MiB (one half of an XCD's L2) that suits them */
  if (ncols <= 4 * maxhot && maxhot > 262144) maxhot = 262144;
  if (maxhot <= 0 || ncols <= 4 * maxhot || nnz == 0 || ncols + maxhot > (int64_t)INT32_MAX) return;
  col_counts own;
  const std::vector<int32_t> &cnt = counts_of(cc, own, ncols, nnz, colidx);
  const int64_t thresh = std::max<int64_t>(8, 4 * ((nnz + ncols - 1) / ncols));
  for (int64_t j = 0; j < ncols; j++) if (cnt[j] >= thresh) hot.push_back((int32_t)j);
  /* most referenced first; ties by column, so that every run packs alike */
  std::sort(hot.begin(), hot.end(), [&](int32_t a, int32_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
  if ((int64_t)hot.size() > maxhot) hot.resize((size_t)maxhot);
  int64_t covered = 0;
  for (size_t h = 0; h < hot.size(); h++) covered += cnt[hot[h]];
  if (hot.empty() || covered * 100 < nnz * 15) { hot.clear(); return; }
  std::vector<int32_t> slot((size_t)ncols, -1);
  for (size_t h = 0; h < hot.size(); h++) slot[hot[h]] = (int32_t)h;
  relabeled.resize((size_t)nnz);
  ksg_parallel(setup_threads(nnz), [&](int t, int nt) {
    for (int64_t p = nnz * t / nt; p < nnz * (t + 1) / nt; p++) {
      const int32_t sl = slot[colidx[p]];
      relabeled[p] = sl >= 0 ? (int32_t)(ncols + sl) : colidx[p];
    }
  });
}

/* Indices of the entries' values in a table of their distinct bit patterns, numbered in the order
 * in which the entries meet them; entries with skip[p] set are left out (their index becomes 255).
 * Returns the number of distinct values, or -1 when there are more than `limit` (at most 256).
 * Two passes on several host threads: every part lists the patterns it meets, the lists are
 * joined in order, then the indices are looked up. */
int ksg_build_value_table(int64_t nnz, const double *val, const uint8_t *skip, size_t limit, std::vector<double> &tab,
                          ksg_hvec<uint8_t> &idx) {
  tab.clear();
  idx.clear();
  const int NT = setup_threads(nnz);
  std::vector<std::vector<uint64_t>> seen((size_t)NT);
  ksg_parallel(NT, [&](int t, int nt) {
    std::vector<uint64_t> &mine = seen[(size_t)t];
    ksg_value_set set;
    uint64_t lastbits = 0;
    bool have = false;
    for (int64_t p = nnz * t / nt; p < nnz * (t + 1) / nt; p++) {
      if (skip && skip[p]) continue;
      uint64_t bits;
      memcpy(&bits, &val[p], 8);
      if (have && bits == lastbits) continue;
      lastbits = bits; have = true;
      if (set.find_or_add(bits, (int)mine.size()) == (int)mine.size()) mine.push_back(bits);
      if (mine.size() > limit) return;
    }
  });
  ksg_value_set all;
  std::vector<uint64_t> order;
  for (int t = 0; t < NT; t++) {
    if (seen[(size_t)t].size() > limit) return -1;
    for (size_t i = 0; i < seen[(size_t)t].size(); i++) {
      if (all.find_or_add(seen[(size_t)t][i], (int)order.size()) == (int)order.size()) order.push_back(seen[(size_t)t][i]);
      if (order.size() > limit) return -1;
    }
  }
  tab.assign(256, 0.0);
  for (size_t i = 0; i < order.size(); i++) memcpy(&tab[i], &order[i], 8);
  idx.resize((size_t)nnz);
  ksg_parallel(NT, [&](int t, int nt) {
    uint64_t lastbits = 0;
    int last = -1;
    for (int64_t p = nnz * t / nt; p < nnz * (t + 1) / nt; p++) {
      if (skip && skip[p]) { idx[(size_t)p] = 255; continue; }
      uint64_t bits;
      memcpy(&bits, &val[p], 8);
      if (last < 0 || bits != lastbits) { last = all.find(bits); lastbits = bits; }
      idx[(size_t)p] = (uint8_t)last;
    }
  });
  return (int)order.size();
}

/* diag_cols: for a block that holds the operator's diagonal, the column index every entry had
 * before the packed columns were relabelled (row r's diagonal entry is the one in column
 * r + dshift); NULL for a block without a diagonal (ghost blocks, what a binned part leaves). */
static int csr_to_device(ksg_csr_dev *D, int64_t nrows, const int64_t *rowptr, const int32_t *colidx,
                         const double *val, const int32_t *diag_cols = NULL, int64_t dshift = 0) {
  D->nrows = nrows;
  D->nnz = rowptr[nrows];
  ksg_phase_timer tm("csr block");
  std::vector<int32_t> tiles;
  build_tiles(nrows, rowptr, tiles, &D->n_stream, &D->n_group, &D->n_long);
  D->ntiles = (int64_t)tiles.size() - 1;
  tm.lap("tiles");
  KSG_CHK(upload(&D->rowptr, rowptr, (size_t)nrows + 1));
  KSG_CHK(upload(&D->colidx, colidx, (size_t)D->nnz));
  KSG_CHK(upload(&D->val, val, (size_t)D->nnz));
  KSG_CHK(upload(&D->tiles, tiles.data(), tiles.size()));
  std::vector<int64_t> tile_p(tiles.size());
  for (size_t i = 0; i < tiles.size(); i++) tile_p[i] = rowptr[tiles[i]];
  KSG_CHK(upload(&D->tile_p, tile_p.data(), tile_p.size()));
  tm.lap("upload");
  /* at most 256 distinct values: one-byte indices into a table for the product kernel; failing
   * that, at most 255 distinct values off the diagonal and the diagonal as a dense vector */
  D->nvt = 0;
  if (D->nnz >= 4096) {
    std::vector<double> tab;
    ksg_hvec<uint8_t> idx;
    if (ksg_build_value_table(D->nnz, val, NULL, 256, tab, idx) >= 0) D->nvt = 1;
    else if (diag_cols && dshift > -(int64_t)INT32_MAX && dshift < (int64_t)INT32_MAX) {
      std::vector<uint8_t> isdiag((size_t)D->nnz, 0);
      std::vector<double> dv((size_t)nrows, 0.0);
      bool one_per_row = true;
      for (int64_t r = 0; r < nrows && one_per_row; r++) {
        int seen = 0;
        const int64_t want = r + dshift;
        for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++)
          if ((int64_t)diag_cols[p] == want) { isdiag[(size_t)p] = 1; dv[(size_t)r] = val[p]; seen++; }
        if (seen > 1) one_per_row = false;              /* duplicate entries: leave the matrix as it is */
      }
      if (one_per_row && ksg_build_value_table(D->nnz, val, isdiag.data(), 255, tab, idx) >= 0) {
        KSG_CHK(upload(&D->dvals, dv.data(), dv.size()));
        D->dshift = (int32_t)dshift;                    /* the kernel reads dvals[column - dshift] */
        D->nvt = 2;
      }
    }
    if (D->nvt) {
      KSG_CHK(upload(&D->vidx, idx.data(), idx.size()));
      KSG_CHK(upload(&D->vtab, tab.data(), tab.size()));
    }
  }
  tm.lap("value table");
  /* row offsets inside the tiles (a tile that is one long row has no use for them) */
  std::vector<uint16_t> rowoff((size_t)nrows + (size_t)D->ntiles + 1, 0);
  for (int64_t i = 0; i < D->ntiles; i++) {
    const int64_t base = rowptr[tiles[i]];
    if (rowptr[tiles[i + 1]] - base > KSG_TILE_NNZ) continue;
    for (int64_t r = tiles[i]; r <= tiles[i + 1]; r++) rowoff[(size_t)(r + i)] = (uint16_t)(rowptr[r] - base);
  }
  tm.lap("row offsets");
  KSG_CHK(upload(&D->rowoff, rowoff.data(), rowoff.size()));
  return KSG_OK;
}

/* Fraction of (row, offset) pairs whose x entry lies in a 256-row block that another XCD works
 * on, when C consecutive blocks share an XCD (block i of a group of 8*C goes to XCD i / C). */
static double cross_xcd_cost(const std::vector<int32_t> &offs, int C) {
  const int64_t per = 8 * (int64_t)C;
  double cost = 0.0;
  for (size_t t = 0; t < offs.size(); t++) {
    const int64_t o = offs[t];
    if (o == 0) continue;
    const int64_t ao = o < 0 ? -o : o;
    /* a shift by ao rows moves a fraction f of a block's rows d+1 blocks on and the rest d blocks */
    const int64_t d = ao / KSG_PC_ROWS;
    const double f = (double)(ao % KSG_PC_ROWS) / KSG_PC_ROWS;
    for (int part = 0; part < 2; part++) {
      const int64_t dd = d + part;
      const double wgt = part ? f : 1.0 - f;
      if (wgt == 0.0) continue;
      int64_t diff = 0;
      for (int64_t i = 0; i < per; i++) if ((i / C) != (((i + dd) % per) / C)) diff++;
      cost += wgt * (double)diff / (double)per;
    }
  }
  return cost;
}

/* Builds the row-code form of a block if it has one: no packed columns, rows of at most 15
 * entries with little padding, at most 256 distinct (column - row - shift, value) pairs. */
static int rowcode_to_device(ksg_csr_dev *D, int64_t nrows, const int64_t *rowptr, const int32_t *colidx, const double *val,
                             int64_t ncols) {
  ksg_phase_timer tm("row codes");
  D->pc_w = 0;
  const char *env = getenv("KSG_SPMV_ROWCODE");
  if (env && env[0] == '0') return KSG_OK;
  const int64_t nnz = rowptr[nrows];
  if (nrows < 4096 || nnz < 4096) return KSG_OK;
  const int NT = setup_threads(nnz);
  int64_t w = 0;
  {
    std::vector<int64_t> wt((size_t)NT, 0);
    ksg_parallel(NT, [&](int t, int nt) {
      int64_t m = 0;
      for (int64_t r = nrows * t / nt; r < nrows * (t + 1) / nt; r++) m = std::max(m, rowptr[r + 1] - rowptr[r]);
      wt[(size_t)t] = m;
    });
    for (int t = 0; t < NT; t++) w = std::max(w, wt[(size_t)t]);
  }
  if (w > 15 || (double)w * (double)nrows > 1.25 * (double)nnz) return KSG_OK;
  const int nw = w <= 7 ? 1 : 2;
  /* the distinct (column - row, value) pairs, numbered in the order in which the rows meet them:
   * every part of the rows lists its own, the lists are joined in order, then the codes are looked up */
  struct pair_set {
    struct slot { int64_t off; uint64_t bits; int code; };
    std::vector<slot> ht;
    pair_set() : ht(2048) { for (auto &e : ht) e.code = -1; }
    static unsigned hash(int64_t off, uint64_t bits) {
      return (unsigned)((((uint64_t)off * 0x9E3779B97F4A7C15ull) ^ (bits * 0xC2B2AE3D27D4EB4Full)) >> 53);
    }
    int find_or_add(int64_t off, uint64_t bits, int next) {
      unsigned h = hash(off, bits);
      while (ht[h].code >= 0 && (ht[h].off != off || ht[h].bits != bits)) h = (h + 1) & 2047;
      if (ht[h].code < 0) { ht[h].off = off; ht[h].bits = bits; ht[h].code = next; }
      return ht[h].code;
    }
    int find(int64_t off, uint64_t bits) const {
      unsigned h = hash(off, bits);
      while (ht[h].code >= 0 && (ht[h].off != off || ht[h].bits != bits)) h = (h + 1) & 2047;
      return ht[h].code;
    }
  };
  std::vector<std::vector<std::pair<int64_t, uint64_t>>> seen((size_t)NT);
  ksg_parallel(NT, [&](int t, int nt) {
    std::vector<std::pair<int64_t, uint64_t>> &mine = seen[(size_t)t];
    pair_set set;
    for (int64_t r = nrows * t / nt; r < nrows * (t + 1) / nt; r++)
      for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
        const int64_t off = (int64_t)colidx[p] - r;
        uint64_t bits;
        memcpy(&bits, &val[p], 8);
        if (set.find_or_add(off, bits, (int)mine.size()) == (int)mine.size()) mine.push_back({off, bits});
        if (mine.size() > 256) return;
      }
  });
  pair_set all;
  std::vector<int32_t> otab;
  std::vector<double> vtab;
  for (int t = 0; t < NT; t++) {
    if (seen[(size_t)t].size() > 256) return KSG_OK;         /* too many pairs: the tile kernel keeps the block */
    for (size_t i = 0; i < seen[(size_t)t].size(); i++) {
      const int64_t off = seen[(size_t)t][i].first;
      const uint64_t bits = seen[(size_t)t][i].second;
      if (all.find_or_add(off, bits, (int)otab.size()) == (int)otab.size()) {
        if (otab.size() == 256) return KSG_OK;
        double v;
        memcpy(&v, &bits, 8);
        otab.push_back((int32_t)off);
        vtab.push_back(v);
      }
    }
  }
  ksg_hvec<uint64_t> words((size_t)nw * (size_t)nrows);
  ksg_parallel(NT, [&](int t, int nt) {
    for (int64_t r = nrows * t / nt; r < nrows * (t + 1) / nt; r++) {
      uint64_t w0 = (uint64_t)(rowptr[r + 1] - rowptr[r]) << 56, w1 = 0;
      for (int64_t p = rowptr[r], k = 0; p < rowptr[r + 1]; p++, k++) {
        uint64_t bits;
        memcpy(&bits, &val[p], 8);
        const uint64_t code = (uint64_t)all.find((int64_t)colidx[p] - r, bits);
        if (k < 7) w0 |= code << (8 * k);
        else w1 |= code << (8 * (k - 7));
      }
      words[(size_t)r] = w0;
      if (nw == 2) words[(size_t)nrows + (size_t)r] = w1;
    }
  });
  tm.lap("codes");
  (void)ncols;
  /* consecutive blocks per XCD: the smallest chunk with the least x shared across XCDs */
  int best = 1;
  double bestc = cross_xcd_cost(otab, 1);
  for (int C = 2; C <= 32; C *= 2) {
    const double cst = cross_xcd_cost(otab, C);
    if (cst < bestc - 1e-9) { bestc = cst; best = C; }
  }
  env = getenv("KSG_SPMV_ROWCODE_CHUNK");
  if (env && atoi(env) > 0) best = atoi(env);
  otab.resize(256, 0);
  vtab.resize(256, 0.0);
  KSG_CHK(upload(&D->pc_words, words.data(), words.size()));
  KSG_CHK(upload(&D->pc_off, otab.data(), otab.size()));
  KSG_CHK(upload(&D->pc_val, vtab.data(), vtab.size()));
  D->pc_chunk = best;
  D->pc_w = (int)w;
  return KSG_OK;
}

/* Uploads one block whose column indices address a vector of `ncols` entries, after deciding
 * on the packing of its most referenced columns. */
static int block_to_device(ksg_ctx c, ksg_csr_dev *D, int64_t ncols, int64_t nrows, const int64_t *rowptr, const int32_t *colidx,
                           const double *val, std::vector<int32_t> &hot_cols, int64_t *nhot, int32_t **d_hot_idx, double **d_xhot,
                           bool has_diag = false, int64_t dshift = 0, col_counts *cc = NULL) {
  std::vector<int32_t> relabeled;
  choose_hot_columns(ncols, rowptr[nrows], colidx, c->hot_cols, hot_cols, relabeled, cc);
  *nhot = (int64_t)hot_cols.size();
  if (*nhot) {
    /* a diagonal entry keeps its own column: the kernel finds its value by that column, and the
     * x entry it multiplies sits next to those of the neighbouring rows anyway */
    if (has_diag)
      for (int64_t r = 0; r < nrows; r++) {
        const int64_t want = r + dshift;
        for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++)
          if ((int64_t)colidx[p] == want) relabeled[(size_t)p] = colidx[p];
      }
    KSG_CHK(upload(d_hot_idx, hot_cols.data(), hot_cols.size()));
    hipError_t e = hipMalloc(d_xhot, (size_t)*nhot * sizeof(double));
    if (e != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "matrix upload: %zu bytes: %s", (size_t)*nhot * sizeof(double), hipGetErrorString(e));
  }
  KSG_CHK(csr_to_device(D, nrows, rowptr, *nhot ? relabeled.data() : colidx, val, has_diag ? colidx : NULL, dshift));
  if (has_diag && !*nhot && ncols == nrows) KSG_CHK(rowcode_to_device(D, nrows, rowptr, colidx, val, ncols));    /* x and y indexed alike */
  return KSG_OK;
}

/* Power-law test of the packing (choose_hot_columns): columns referenced at least four times as
 * often as the average hold 15 % or more of the entries. */
static bool is_power_law(int64_t ncols, int64_t nnz, const int32_t *colidx, col_counts *cc = NULL) {
  if (ncols <= 0 || nnz <= 0) return false;
  col_counts own;
  const std::vector<int32_t> &cnt = counts_of(cc, own, ncols, nnz, colidx);
  const int64_t thresh = std::max<int64_t>(8, 4 * ((nnz + ncols - 1) / ncols));
  int64_t covered = 0;
  for (int64_t j = 0; j < ncols; j++) if (cnt[j] >= thresh) covered += cnt[j];
  return covered * 100 >= nnz * 15;
}

/* Whether a block of a new operator (the local block, or the ghost block with the ghost buffer
 * as its x) gets a binned part (KSG_OPT_BINS): a
 * power-law operator (the test of the packing), and by default an x of 16 MiB or more (R-MAT scale 20,
 * 8 MiB: 0.128 ms per product as a CSR and binned alike -- too few column blocks to fill the device;
 * scale 21: 0.256 and 0.160 ms; scale 22: 0.504 and 0.258 ms; scale 24: 2.09 and 1.10 ms; scale 26:
 * 10.8 and 4.79 ms; profiles/r03_spmv_rmat_binned.txt). */
static bool bins_wanted(ksg_ctx c, int64_t ncols, int64_t nnz, const int32_t *colidx, col_counts *cc = NULL) {
  if (c->bins == 1 || ncols < 64 || ncols > (int64_t)INT32_MAX) return false;
  if (c->bins == 0 && ncols < ((int64_t)16 << 20) / 8) return false;
  /* from 2^27 entries on the test is first made on every eighth stretch of 4096 entries (the same
   * entries whatever the number of host threads): a block that passes it needs no exact count at
   * all -- at R-MAT scale 26 the count is 1.0 s of a 4.3 s set-up -- and for one that fails it the
   * other stretches are counted into the same array, which is then the exact count the decisions
   * that follow use */
  col_counts own;
  col_counts &kc = cc ? *cc : own;
  const char *smin = getenv("KSG_SAMPLE_MIN_NNZ");          /* for tests: the size from which the sample is taken */
  const int64_t sample_from = smin && atoll(smin) > 0 ? atoll(smin) : ((int64_t)1 << 27);
  if (nnz >= sample_from && nnz >= 8 * 4096 && !(kc.of == colidx && (int64_t)kc.cnt.size() == ncols)) {
    const int64_t nch = nnz / (8 * 4096), ns = nch * 4096;
    kc.of = NULL;
    kc.cnt.assign((size_t)ncols, 0);
    int32_t *k = kc.cnt.data();
    ksg_parallel(setup_threads(ns), [&](int t, int nt) {
      for (int64_t ch = nch * t / nt; ch < nch * (t + 1) / nt; ch++) {
        const int32_t *ci = colidx + ch * 8 * 4096;
        for (int i = 0; i < 4096; i++) __atomic_fetch_add(&k[ci[i]], 1, __ATOMIC_RELAXED);
      }
    });
    const int64_t thresh = std::max<int64_t>(8, 4 * ((ns + ncols - 1) / ncols));
    int64_t covered = 0;
    for (int64_t j = 0; j < ncols; j++) if (k[j] >= thresh) covered += k[j];
    if (covered * 100 >= ns * 15) return true;
    ksg_parallel(setup_threads(nnz), [&](int t, int nt) {
      for (int64_t ch = nch * t / nt; ch < nch * (t + 1) / nt; ch++) {
        const int32_t *ci = colidx + ch * 8 * 4096;
        for (int i = 4096; i < 8 * 4096; i++) __atomic_fetch_add(&k[ci[i]], 1, __ATOMIC_RELAXED);
      }
    });
    for (int64_t p = nch * 8 * 4096; p < nnz; p++) k[colidx[p]]++;
    kc.of = colidx;
  }
  return is_power_law(ncols, nnz, colidx, &kc);
}

static int rest_to_device(ksg_ctx c, std::vector<ksg_rest> &out, int64_t ncols, const std::vector<int32_t> &rows_y,
                          const std::vector<int64_t> &lrp, const std::vector<int32_t> &lci, const std::vector<double> &lva) {
  /* what a binned part leaves: a CSR over the rows it touches (rows_y: the row of y each stands
   * for), added by the tile kernel afterwards */
  out.clear();
  if (rows_y.empty()) return KSG_OK;
  out.emplace_back();
  ksg_rest &R = out.back();
  memset(&R.blk, 0, sizeof(R.blk));
  R.ncols = ncols; R.nrows = (int64_t)rows_y.size();
  R.d_rows = NULL; R.nhot = 0; R.d_hot_idx = NULL; R.d_xhot = NULL;
  KSG_CHK(upload(&R.d_rows, rows_y.data(), rows_y.size()));
  /* entries of a diagonal that could not be folded are ordinary entries here */
  return block_to_device(c, &R.blk, R.ncols, R.nrows, lrp.data(), lci.data(), lva.data(), R.hot_cols, &R.nhot, &R.d_hot_idx, &R.d_xhot);
}

static int diag_to_device(ksg_mat A, int64_t nrows, const int64_t *rowptr, const int32_t *colidx, const double *val) {
  /* the diagonal of a square operator lies in this block: row r (global row0 + r) has it in the
   * local column row0 + r - col0 */
  const bool square = A->N == A->NC;
  ksg_ctx c = A->ctx;
  ksg_phase_timer tm("local block");
  col_counts cc;                                         /* the block's entries per column, counted at most once */
  if (bins_wanted(c, A->nc, rowptr[nrows], colidx, &cc)) {
    { std::vector<int32_t>().swap(cc.cnt); cc.of = NULL; }
    /* well filled (row block, column block) cells and the diagonal go to the binned form, which
     * stores y; what is left is a CSR over the rows it touches, added afterwards.  A->diag stays empty. */
    std::vector<int32_t> lrows, lci;
    std::vector<int64_t> lrp;
    std::vector<double> lva;
    tm.lap("power-law test");
    KSG_CHK(ksg_bins_build(c, &A->bins, nrows, A->nc, rowptr, colidx, val, square, A->row0 - A->col0, lrows, lrp, lci, lva, NULL));
    tm.lap("binned part");
    if (A->bins.nent) return rest_to_device(c, A->rest, A->nc, lrows, lrp, lci, lva);
  }
  tm.lap("binned form not taken");
  return block_to_device(c, &A->diag, A->nc, nrows, rowptr, colidx, val, A->hot_cols, &A->nhot, &A->d_hot_idx, &A->d_xhot,
                         square, A->row0 - A->col0, &cc);
}

static void csr_free(ksg_csr_dev *D) {
  if (D->rowptr) hipFree(D->rowptr);
  if (D->colidx) hipFree(D->colidx);
  if (D->val) hipFree(D->val);
  if (D->tiles) hipFree(D->tiles);
  if (D->tile_p) hipFree(D->tile_p);
  if (D->rowoff) hipFree(D->rowoff);
  if (D->vidx) hipFree(D->vidx);
  if (D->vtab) hipFree(D->vtab);
  if (D->dvals) hipFree(D->dvals);
  if (D->pc_words) hipFree(D->pc_words);
  if (D->pc_off) hipFree(D->pc_off);
  if (D->pc_val) hipFree(D->pc_val);
  memset(D, 0, sizeof(*D));
}

static int validate_csr(int64_t n, int64_t N, const int64_t *rowptr, const int32_t *colidx) {
  /* N: number of columns */
  if (rowptr[0] != 0) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: rowptr[0] must be 0");
  for (int64_t i = 0; i < n; i++)
    if (rowptr[i + 1] < rowptr[i]) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: rowptr decreases at row %lld", (long long)i);
  const int64_t nnz = rowptr[n];
  const int NT = setup_threads(nnz);
  std::vector<int64_t> bad((size_t)NT, -1);                /* per part: its first entry out of range */
  ksg_parallel(NT, [&](int t, int nt) {
    for (int64_t p = nnz * t / nt; p < nnz * (t + 1) / nt; p++)
      if (colidx[p] < 0 || (int64_t)colidx[p] >= N) { bad[(size_t)t] = p; return; }
  });
  for (int t = 0; t < NT; t++)
    if (bad[(size_t)t] >= 0)
      KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: column index %d out of range at entry %lld", colidx[bad[(size_t)t]], (long long)bad[(size_t)t]);
  return KSG_OK;
}

/* all-gather of one int64 per rank plus the end of the last block, through a zero-padded sum */
static int gather_starts(ksg_ctx c, int64_t first, int64_t count, std::vector<int64_t> &out) {
  const int P = c->nranks;
  std::vector<double> starts((size_t)P + 1, 0.0);
  starts[c->rank] = (double)first;
  if (c->rank == P - 1) starts[P] = (double)(first + count);
  KSG_CHK(ksg_comm_allreduce_host(c, starts.data(), P + 1, 0));
  out.resize((size_t)P + 1);
  for (int p = 0; p <= P; p++) out[p] = (int64_t)starts[p];
  return KSG_OK;
}

/* Ranks agree on whether a set-up stage succeeded everywhere, so that a rank-local failure does
 * not leave the others waiting in the next collective: every rank returns an error or none does. */
static int agree(ksg_ctx c, int rc, const char *stage) {
  double flag = rc ? 1.0 : 0.0;
  const int rc2 = ksg_comm_allreduce_host(c, &flag, 1, 1);
  if (rc) return rc;                                   /* the local failure and its message */
  if (rc2) return rc2;
  if (flag != 0.0) KSG_FAIL(KSG_ERR_COMM, "ksg_mat_create_csr: another rank failed while %s", stage);
  return KSG_OK;
}

static int mat_create(ksg_ctx c, int64_t n, int64_t N, int64_t row0, int64_t nc, int64_t NC, int64_t col0,
                      const int64_t *rowptr, const int32_t *colidx, const double *val, ksg_mat *out) {
  if (!c || !out || !rowptr || n < 0 || N < n || row0 < 0 || row0 + n > N || nc < 0 || NC < nc || col0 < 0 || col0 + nc > NC)
    KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: bad arguments");
  if (NC > (int64_t)INT32_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_mat_create_csr: N exceeds 32-bit column indices");
  if (n > (int64_t)INT32_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_mat_create_csr: more than 2^31-1 local rows");
  if (rowptr[n] > 0 && (!colidx || !val)) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: NULL arrays");
  ksg_phase_timer tmc("operator");
  KSG_CHK(validate_csr(n, NC, rowptr, colidx));
  KSG_HIP(hipSetDevice(c->device));
  tmc.lap("validation");
  ksg_mat A = new (std::nothrow) ksg_mat_s();
  if (!A) KSG_FAIL(KSG_ERR_MEM, "ksg_mat_create_csr: out of host memory");
  A->ctx = c; A->n = n; A->N = N; A->row0 = row0; A->nnz = rowptr[n];
  A->nc = nc; A->NC = NC; A->col0 = col0;
  memset(&A->diag, 0, sizeof(A->diag));
  A->no_rows = 0; A->o_rows = NULL; A->o_nnz = 0;
  memset(&A->ghost, 0, sizeof(A->ghost));
  memset(&A->bins, 0, sizeof(A->bins));
  memset(&A->gbins, 0, sizeof(A->gbins));
  A->g_nhot = 0; A->d_g_hot_idx = NULL; A->d_g_xhot = NULL;
  A->nghost = A->nsend = 0;
  A->halo_rounds = 0;
  A->d_ghost = A->d_send = A->h_ghost = A->h_send = NULL; A->d_send_idx = NULL;
  A->nhot = 0; A->d_hot_idx = NULL; A->d_xhot = NULL;
  const int P = c->nranks;
  int rc = KSG_OK;

  if (P == 1) {
    if (row0 != 0 || n != N || col0 != 0 || nc != NC) { delete A; KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: one rank must own all rows and columns"); }
    rc = diag_to_device(A, n, rowptr, colidx, val);
    tmc.lap("local block");
    if (rc) { ksg_mat_destroy(A); return rc; }
    *out = A;
    return KSG_OK;
  }

  /* ---- several ranks: ownership ranges (allgather through a zero-padded sum) ---- */
  if ((size_t)P * P > KSG_RED_MAX) { delete A; KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_mat_create_csr: too many ranks"); }   /* the same on every rank */
  std::vector<int64_t> rs;
  rc = gather_starts(c, row0, n, rs);
  /* the vector x is distributed by column blocks (the same as the rows for a square operator) */
  if (!rc) rc = gather_starts(c, col0, nc, A->cstarts);
  if (rc) { delete A; return rc; }                     /* a failed collective: nothing to agree on */
  const std::vector<int64_t> &cs = A->cstarts;
  /* every rank holds the same two tables, so these checks fail on all ranks or on none */
  if (rs[0] != 0 || rs[P] != N) { delete A; KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: row blocks do not tile [0,N)"); }
  for (int p = 0; p < P; p++) if (rs[p + 1] < rs[p]) { delete A; KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: row blocks are not ordered by rank"); }
  if (cs[0] != 0 || cs[P] != NC) { delete A; KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: column blocks do not tile [0,NC)"); }
  for (int p = 0; p < P; p++) if (cs[p + 1] < cs[p]) { delete A; KSG_FAIL(KSG_ERR_ARG, "ksg_mat_create_csr: column blocks are not ordered by rank"); }

  /* ---- split into diagonal and ghost block ---- */
  const int64_t nnz = rowptr[n];
  const int64_t lo = col0, hi = col0 + nc;
  std::vector<int64_t> d_rp((size_t)n + 1, 0), o_rp;
  std::vector<int32_t> d_ci, o_ci, o_rows;
  std::vector<double> d_v, o_v;
  d_ci.reserve((size_t)nnz); d_v.reserve((size_t)nnz);
  std::vector<int32_t> ghosts;                    /* global indices of the columns the ghost block uses, in order */
  o_rp.push_back(0);
  for (int64_t i = 0; i < n; i++) {
    bool has = false;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
      const int64_t cg = colidx[p];
      if (cg >= lo && cg < hi) { d_ci.push_back((int32_t)(cg - lo)); d_v.push_back(val[p]); }
      else { o_ci.push_back((int32_t)cg); o_v.push_back(val[p]); has = true; }
    }
    d_rp[i + 1] = (int64_t)d_ci.size();
    if (has) { o_rows.push_back((int32_t)i); o_rp.push_back((int64_t)o_ci.size()); }
  }
  {
    const int64_t onz = (int64_t)o_ci.size();
    /* (KSG_GHOST_SORT=1 takes the other branch whatever the sizes, for tests) */
    const char *gs = getenv("KSG_GHOST_SORT");
    if (!(gs && gs[0] == '1') && NC <= 16 * onz) {
      /* a mark per column of the operator, then the marked columns numbered in order -- 4 NC bytes
       * and a pass over all columns, so only when the ghost entries are a fair share of them (a
       * stencil's slab has half a million ghost entries among 1.3e8 columns: sorted instead) */
      std::vector<int32_t> num((size_t)NC, -1);
      for (int64_t p = 0; p < onz; p++) num[(size_t)o_ci[(size_t)p]] = 0;
      for (int64_t j = 0; j < NC; j++) if (num[(size_t)j] == 0) { num[(size_t)j] = (int32_t)ghosts.size(); ghosts.push_back((int32_t)j); }
      ksg_parallel(setup_threads(onz), [&](int t, int nt) {
        for (int64_t p = onz * t / nt; p < onz * (t + 1) / nt; p++) o_ci[(size_t)p] = num[(size_t)o_ci[(size_t)p]];
      });
    } else {
      ghosts.assign(o_ci.begin(), o_ci.end());
      std::sort(ghosts.begin(), ghosts.end());
      ghosts.erase(std::unique(ghosts.begin(), ghosts.end()), ghosts.end());
      ksg_parallel(setup_threads(onz), [&](int t, int nt) {
        for (int64_t p = onz * t / nt; p < onz * (t + 1) / nt; p++)
          o_ci[(size_t)p] = (int32_t)(std::lower_bound(ghosts.begin(), ghosts.end(), o_ci[(size_t)p]) - ghosts.begin());
      });
    }
  }
  A->nghost = (int64_t)ghosts.size();
  A->ghosts = ghosts;

  tmc.lap("split, ghost columns");
  /* ---- who owns my ghosts: receive offsets (ghosts are sorted, so grouped by owner) ---- */
  A->recvoff.assign((size_t)P + 1, 0);
  {
    size_t g = 0;
    for (int p = 0; p < P; p++) {
      A->recvoff[p] = (int64_t)g;
      while (g < ghosts.size() && (int64_t)ghosts[g] < cs[p + 1]) g++;
    }
    A->recvoff[P] = (int64_t)ghosts.size();
  }
  /* ---- counts matrix (all-to-all of counts through a sum) ---- */
  std::vector<double> counts((size_t)P * P, 0.0);
  for (int p = 0; p < P; p++) counts[(size_t)c->rank * P + p] = (double)(A->recvoff[p + 1] - A->recvoff[p]);
  rc = ksg_comm_allreduce_host(c, counts.data(), P * P, 0);
  if (rc) { delete A; return rc; }                     /* a failed collective: nothing to agree on */
  A->sendoff.assign((size_t)P + 1, 0);
  for (int p = 0; p < P; p++) A->sendoff[p + 1] = A->sendoff[p] + (int64_t)counts[(size_t)p * P + c->rank];
  A->nsend = A->sendoff[P];

  /* ---- tell the owners which of their entries I need (indices travel as doubles) ---- */
  std::vector<double> want((size_t)std::max<int64_t>(A->nghost, 1)), asked((size_t)std::max<int64_t>(A->nsend, 1));
  for (int64_t g = 0; g < A->nghost; g++) want[g] = (double)ghosts[g];
  /* (a rank-local failure inside the exchange is agreed upon there; what is left is a failed collective) */
  rc = ksg_comm_exchange_host(c, want.data(), A->recvoff.data(), asked.data(), A->sendoff.data());
  std::vector<int32_t> send_idx((size_t)std::max<int64_t>(A->nsend, 1));
  for (int64_t i = 0; i < A->nsend && !rc; i++) {
    const int64_t gidx = (int64_t)asked[i];
    if (gidx < lo || gidx >= hi) { ksg_set_error("halo plan: rank asked for entry %lld that is not mine", (long long)gidx); rc = KSG_ERR_COMM; break; }
    send_idx[i] = (int32_t)(gidx - lo);
  }
  rc = agree(c, rc, "building the halo plan");
  if (rc) { delete A; return rc; }

  tmc.lap("halo plan");
  /* ---- upload ---- */
  rc = diag_to_device(A, n, d_rp.data(), d_ci.data(), d_v.data());
  A->no_rows = (int64_t)o_rows.size();
  A->o_nnz = (int64_t)o_ci.size();
  if (!rc) rc = upload(&A->o_rows, o_rows.data(), o_rows.size());
  if (!rc && A->no_rows && bins_wanted(c, A->nghost, A->o_nnz, o_ci.data())) {
    /* the ghost block in the binned form: its x is the ghost buffer, its rows the compressed list
     * o_rows (the reduce pass adds into y through that list), what is left a group over some of them */
    std::vector<int32_t> lrows, lci;
    std::vector<int64_t> lrp;
    std::vector<double> lva;
    /* the ghost buffer arrives in rounds (ksg_comm_exchange_rounds): piece s of every source's share in
     * round s, which is the stage of its columns (KSG_HALO_ROUNDS: the number of rounds, 4 unless set;
     * 0 or 1: one exchange, one stage) */
    ksg_colstages cst;
    const char *hr = getenv("KSG_HALO_ROUNDS");
    A->halo_rounds = hr && hr[0] ? atoi(hr) : 4;
    if (A->halo_rounds < 2 || P < 2) A->halo_rounds = 0;
    if (A->halo_rounds > 64) A->halo_rounds = 64;
    if (A->halo_rounds) {
      const int R = A->halo_rounds;
      cst.nstages = R;
      for (int p = 0; p < P; p++) {
        const int64_t lr = A->recvoff[(size_t)p + 1] - A->recvoff[(size_t)p];
        for (int st = 0; st < R; st++) {
          cst.bound.push_back(A->recvoff[(size_t)p] + lr * st / R);
          cst.stage.push_back(st);
        }
      }
      cst.bound.push_back(A->recvoff[(size_t)P]);
    }
    rc = ksg_bins_build(c, &A->gbins, A->no_rows, A->nghost, o_rp.data(), o_ci.data(), o_v.data(), false, 0, lrows, lrp, lci, lva, o_rows.data(),
                        A->halo_rounds ? &cst : NULL);
    if (!rc && A->gbins.nent && !lrows.empty()) {
      for (size_t i = 0; i < lrows.size(); i++) lrows[i] = o_rows[(size_t)lrows[i]];      /* local rows of y */
      rc = rest_to_device(c, A->grest, A->nghost, lrows, lrp, lci, lva);
    }
  }
  if (!rc && A->no_rows && !A->gbins.nent)
    rc = block_to_device(c, &A->ghost, A->nghost, A->no_rows, o_rp.data(), o_ci.data(), o_v.data(), A->g_hot_cols, &A->g_nhot,
                         &A->d_g_hot_idx, &A->d_g_xhot);
  if (!rc) rc = upload(&A->d_send_idx, send_idx.data(), (size_t)A->nsend);
  if (!rc) { hipError_t e = hipMalloc(&A->d_ghost, (size_t)std::max<int64_t>(A->nghost, 1) * sizeof(double)); if (e != hipSuccess) rc = KSG_ERR_MEM; }
  if (!rc) { hipError_t e = hipMalloc(&A->d_send, (size_t)std::max<int64_t>(A->nsend, 1) * sizeof(double)); if (e != hipSuccess) rc = KSG_ERR_MEM; }
  if (!rc && c->comm_kind == KSG_COMM_CALLBACKS && !c->cb_on_device) {
    if (hipHostMalloc(&A->h_ghost, (size_t)std::max<int64_t>(A->nghost, 1) * sizeof(double), hipHostMallocDefault) != hipSuccess) rc = KSG_ERR_MEM;
    if (!rc && hipHostMalloc(&A->h_send, (size_t)std::max<int64_t>(A->nsend, 1) * sizeof(double), hipHostMallocDefault) != hipSuccess) rc = KSG_ERR_MEM;
  }
  tmc.lap("blocks to the device");
  if (rc == KSG_ERR_MEM) ksg_set_error("ksg_mat_create_csr: out of device memory");
  rc = agree(c, rc, "uploading the matrix");
  if (rc) { ksg_mat_destroy(A); return rc; }
  *out = A;
  return KSG_OK;
}

extern "C" int ksg_mat_create_csr(ksg_ctx c, int64_t n, int64_t N, int64_t row0, const int64_t *rowptr,
                                  const int32_t *colidx, const double *val, ksg_mat *out) {
  return mat_create(c, n, N, row0, n, N, row0, rowptr, colidx, val, out);
}

extern "C" int ksg_mat_create_csr_rect(ksg_ctx c, int64_t n, int64_t N, int64_t row0, int64_t nc, int64_t NC,
                                       int64_t col0, const int64_t *rowptr, const int32_t *colidx,
                                       const double *val, ksg_mat *out) {
  return mat_create(c, n, N, row0, nc, NC, col0, rowptr, colidx, val, out);
}

extern "C" int ksg_mat_col_sizes(ksg_mat A, int64_t *nc, int64_t *NC, int64_t *col0) {
  if (!A) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_col_sizes: NULL matrix");
  if (nc) *nc = A->nc;
  if (NC) *NC = A->NC;
  if (col0) *col0 = A->col0;
  return KSG_OK;
}

template <class T>
static int download(std::vector<T> &dst, const T *src, size_t count) {
  dst.resize(count);
  if (count) KSG_HIP(hipMemcpy(dst.data(), src, count * sizeof(T), hipMemcpyDeviceToHost));
  return KSG_OK;
}

/* One block of an operator back on the host as a CSR over its `nrows` rows, columns as the block
 * numbers them: the base CSR `D` (over all rows; the columns of its packed entries named again),
 * what a binned part left (`cgs`: a CSR over some of the rows; `rows_index` gives, for an entry of
 * its d_rows, the row of this block it stands for -- NULL: d_rows holds those rows themselves) and the
 * binned part.  Every row in ascending column order when it was given so. */
static int block_to_host(const ksg_csr_dev &D, const std::vector<int32_t> &hot, int64_t base_ncols, const std::vector<ksg_rest> &cgs,
                         const ksg_bins &bins, int64_t nrows, const std::vector<int32_t> *rows_index, std::vector<int64_t> &rp,
                         std::vector<int32_t> &ci, std::vector<double> &va) {
  std::vector<int64_t> drp;
  std::vector<int32_t> dci;
  std::vector<double> dva;
  if (D.rowptr) {
    KSG_CHK(download(drp, D.rowptr, (size_t)nrows + 1));
    KSG_CHK(download(dci, D.colidx, (size_t)D.nnz));
    KSG_CHK(download(dva, D.val, (size_t)D.nnz));
    if (!hot.empty()) for (size_t p = 0; p < dci.size(); p++) if ((int64_t)dci[p] >= base_ncols) dci[p] = hot[(size_t)((int64_t)dci[p] - base_ncols)];
  } else {
    drp.assign((size_t)nrows + 1, 0);
  }
  if (cgs.empty() && !bins.nent) { rp.swap(drp); ci.swap(dci); va.swap(dva); return KSG_OK; }
  /* triplets of everything that is not in the base CSR */
  std::vector<int32_t> tr, tc;
  std::vector<double> tv;
  bool ordered = true;
  for (const ksg_rest &cg : cgs) {
    std::vector<int32_t> grows, gci;
    std::vector<int64_t> grp;
    std::vector<double> gva;
    KSG_CHK(download(grows, cg.d_rows, (size_t)cg.nrows));
    KSG_CHK(download(grp, cg.blk.rowptr, (size_t)cg.nrows + 1));
    KSG_CHK(download(gci, cg.blk.colidx, (size_t)cg.blk.nnz));
    KSG_CHK(download(gva, cg.blk.val, (size_t)cg.blk.nnz));
    ordered = false;                                       /* the remaining entries of a binned block: any column */
    for (int64_t i = 0; i < cg.nrows; i++) {
      int32_t r = grows[(size_t)i];
      if (rows_index) r = (int32_t)(std::lower_bound(rows_index->begin(), rows_index->end(), r) - rows_index->begin());
      for (int64_t p = grp[(size_t)i]; p < grp[(size_t)i + 1]; p++) {
        int64_t col = gci[(size_t)p];
        if (col >= cg.ncols) col = cg.hot_cols[(size_t)(col - cg.ncols)];
        tr.push_back(r); tc.push_back((int32_t)col); tv.push_back(gva[(size_t)p]);
      }
    }
  }
  if (bins.nent) {
    std::vector<int32_t> br, bc;
    std::vector<double> bv;
    KSG_CHK(ksg_bins_to_host(&bins, br, bc, bv));
    tr.insert(tr.end(), br.begin(), br.end()); tc.insert(tc.end(), bc.begin(), bc.end()); tv.insert(tv.end(), bv.begin(), bv.end());
    ordered = false;
  }
  rp.assign((size_t)nrows + 1, 0);
  for (int64_t i = 0; i < nrows; i++) rp[i + 1] = drp[i + 1] - drp[i];
  for (size_t e = 0; e < tr.size(); e++) rp[(size_t)tr[e] + 1]++;
  for (int64_t i = 0; i < nrows; i++) rp[i + 1] += rp[i];
  ci.resize((size_t)rp[nrows]); va.resize((size_t)rp[nrows]);
  std::vector<int64_t> pos(rp.begin(), rp.end() - 1);
  for (int64_t i = 0; i < nrows; i++)
    for (int64_t p = drp[i]; p < drp[i + 1]; p++) { ci[(size_t)pos[i]] = dci[p]; va[(size_t)pos[i]] = dva[p]; pos[i]++; }
  for (size_t e = 0; e < tr.size(); e++) { const int64_t q = pos[tr[e]]++; ci[(size_t)q] = tc[e]; va[(size_t)q] = tv[e]; }
  if (!ordered) {
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t i = 0; i < nrows; i++) {
      const int64_t a = rp[i], b = rp[i + 1];
      if (b - a == drp[i + 1] - drp[i]) continue;          /* nothing added to this row */
      row.clear();
      for (int64_t p = a; p < b; p++) row.push_back({ci[(size_t)p], va[(size_t)p]});
      std::stable_sort(row.begin(), row.end(), [](const std::pair<int32_t, double> &u, const std::pair<int32_t, double> &v) { return u.first < v.first; });
      for (int64_t p = a; p < b; p++) { ci[(size_t)p] = row[(size_t)(p - a)].first; va[(size_t)p] = row[(size_t)(p - a)].second; }
    }
  }
  return KSG_OK;
}

/* the local rows back on the host with global column indices (the two blocks of a
 * distributed matrix merged again, every row in ascending column order when it was given so) */
static int mat_to_host(ksg_mat A, std::vector<int64_t> &rp, std::vector<int32_t> &ci, std::vector<double> &va) {
  ksg_ctx c = A->ctx;
  KSG_HIP(hipSetDevice(c->device));
  KSG_HIP(hipStreamSynchronize(c->comm_stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  std::vector<int64_t> drp;
  std::vector<int32_t> dci;
  std::vector<double> dva;
  KSG_CHK(block_to_host(A->diag, A->hot_cols, A->nc, A->rest, A->bins, A->n, NULL, drp, dci, dva));
  if (c->nranks == 1) {
    rp.swap(drp); ci.swap(dci); va.swap(dva);
    return KSG_OK;
  }
  std::vector<int32_t> orows, oci;
  std::vector<int64_t> orp;
  std::vector<double> ova;
  KSG_CHK(download(orows, A->o_rows, (size_t)A->no_rows));
  if (A->no_rows) KSG_CHK(block_to_host(A->ghost, A->g_hot_cols, A->nghost, A->grest, A->gbins, A->no_rows, &orows, orp, oci, ova));
  else orp.assign(1, 0);
  rp.assign((size_t)A->n + 1, 0);
  ci.clear(); va.clear();
  ci.reserve((size_t)A->nnz); va.reserve((size_t)A->nnz);
  size_t o = 0;
  for (int64_t i = 0; i < A->n; i++) {
    int64_t a = drp[i], ae = drp[i + 1], b = 0, be = 0;
    if (o < orows.size() && orows[o] == i) { b = orp[o]; be = orp[o + 1]; o++; }
    while (a < ae || b < be) {
      const int64_t ga = a < ae ? (int64_t)dci[a] + A->col0 : INT64_MAX;
      const int64_t gb = b < be ? (int64_t)A->ghosts[oci[b]] : INT64_MAX;
      if (ga <= gb) { ci.push_back((int32_t)ga); va.push_back(dva[a]); a++; }
      else { ci.push_back((int32_t)gb); va.push_back(ova[b]); b++; }
    }
    rp[i + 1] = (int64_t)ci.size();
  }
  return KSG_OK;
}

extern "C" int ksg_mat_get_csr(ksg_mat A, int64_t *rowptr, int32_t *colidx, double *val) {
  if (!A || !rowptr) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_get_csr: NULL argument");
  if (A->nnz && (!colidx || !val)) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_get_csr: NULL arrays");
  std::vector<int64_t> rp;
  std::vector<int32_t> ci;
  std::vector<double> va;
  KSG_CHK(mat_to_host(A, rp, ci, va));
  memcpy(rowptr, rp.data(), rp.size() * sizeof(int64_t));
  if (!ci.empty()) { memcpy(colidx, ci.data(), ci.size() * sizeof(int32_t)); memcpy(val, va.data(), va.size() * sizeof(double)); }
  return KSG_OK;
}

/* AT = A^T as a new operator: rank p receives the entries of the columns it owns and holds
 * them as rows (what MatTranspose does for an MPIAIJ matrix).  Collective. */
extern "C" int ksg_mat_transpose(ksg_mat A, ksg_mat *out) {
  if (!A || !out) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_transpose: NULL argument");
  ksg_ctx c = A->ctx;
  const int P = c->nranks;
  std::vector<int64_t> rp;
  std::vector<int32_t> ci;
  std::vector<double> va;
  KSG_CHK(mat_to_host(A, rp, ci, va));
  /* entries for the local rows of the transpose as (row of AT = column of A minus col0, column of AT = global row of A, value),
   * ordered by sender rank and then by row of A */
  std::vector<int32_t> er, ec;
  std::vector<double> ev;
  if (P == 1) {
    er.reserve(ci.size()); ec.reserve(ci.size());
    for (int64_t i = 0; i < A->n; i++)
      for (int64_t p = rp[i]; p < rp[i + 1]; p++) { er.push_back(ci[p]); ec.push_back((int32_t)i); }
    ev = va;
  } else {
    if (A->N > (int64_t)INT32_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_mat_transpose: more than 2^31-1 rows");
    const std::vector<int64_t> &cs = A->cstarts;
    std::vector<int64_t> cnt((size_t)P, 0);
    auto owner = [&](int32_t col) { return (int)(std::upper_bound(cs.begin(), cs.end(), (int64_t)col) - cs.begin()) - 1; };
    for (size_t p = 0; p < ci.size(); p++) cnt[owner(ci[p])]++;
    if ((size_t)P * P > KSG_RED_MAX) KSG_FAIL(KSG_ERR_UNSUPPORTED, "ksg_mat_transpose: too many ranks");
    std::vector<double> counts((size_t)P * P, 0.0);
    for (int p = 0; p < P; p++) counts[(size_t)c->rank * P + p] = (double)cnt[p];
    KSG_CHK(ksg_comm_allreduce_host(c, counts.data(), P * P, 0));
    std::vector<int64_t> sendoff((size_t)P + 1, 0), recvoff((size_t)P + 1, 0), fill((size_t)P);
    for (int p = 0; p < P; p++) {
      sendoff[p + 1] = sendoff[p] + 3 * cnt[p];
      recvoff[p + 1] = recvoff[p] + 3 * (int64_t)counts[(size_t)p * P + c->rank];
      fill[p] = sendoff[p];
    }
    std::vector<double> send((size_t)std::max<int64_t>(sendoff[P], 1)), recv((size_t)std::max<int64_t>(recvoff[P], 1));
    for (int64_t i = 0; i < A->n; i++)
      for (int64_t p = rp[i]; p < rp[i + 1]; p++) {
        int64_t &f = fill[owner(ci[p])];
        send[f] = (double)(A->row0 + i); send[f + 1] = (double)ci[p]; send[f + 2] = va[p];
        f += 3;
      }
    KSG_CHK(ksg_comm_exchange_host(c, send.data(), sendoff.data(), recv.data(), recvoff.data()));
    const int64_t ne = recvoff[P] / 3;
    er.resize((size_t)ne); ec.resize((size_t)ne); ev.resize((size_t)ne);
    for (int64_t e = 0; e < ne; e++) {
      const int64_t col = (int64_t)recv[3 * e + 1];
      if (col < A->col0 || col >= A->col0 + A->nc) KSG_FAIL(KSG_ERR_COMM, "ksg_mat_transpose: received column %lld that is not mine", (long long)col);
      er[e] = (int32_t)(col - A->col0); ec[e] = (int32_t)recv[3 * e]; ev[e] = recv[3 * e + 2];
    }
  }
  /* stable counting sort by row: columns stay ascending */
  const int64_t nrow = A->nc, ne = (int64_t)er.size();
  std::vector<int64_t> trp((size_t)nrow + 1, 0);
  for (int64_t e = 0; e < ne; e++) trp[er[e] + 1]++;
  for (int64_t r = 0; r < nrow; r++) trp[r + 1] += trp[r];
  std::vector<int64_t> pos(trp.begin(), trp.end() - 1);
  std::vector<int32_t> tci((size_t)std::max<int64_t>(ne, 1));
  std::vector<double> tva((size_t)std::max<int64_t>(ne, 1));
  for (int64_t e = 0; e < ne; e++) { const int64_t q = pos[er[e]]++; tci[q] = ec[e]; tva[q] = ev[e]; }
  return mat_create(c, A->nc, A->NC, A->col0, A->n, A->N, A->row0, trp.data(), tci.data(), tva.data(), out);
}

extern "C" int ksg_mat_destroy(ksg_mat A) {
  if (!A) return KSG_OK;
  hipSetDevice(A->ctx->device);
  hipStreamSynchronize(A->ctx->comm_stream);
  hipStreamSynchronize(A->ctx->stream);
  csr_free(&A->diag);
  csr_free(&A->ghost);
  for (ksg_rest &cg : A->rest) {
    csr_free(&cg.blk);
    if (cg.d_rows) hipFree(cg.d_rows);
    if (cg.d_hot_idx) hipFree(cg.d_hot_idx);
    if (cg.d_xhot) hipFree(cg.d_xhot);
  }
  A->rest.clear();
  ksg_bins_free(&A->bins);
  ksg_bins_free(&A->gbins);
  for (ksg_rest &cg : A->grest) {
    csr_free(&cg.blk);
    if (cg.d_rows) hipFree(cg.d_rows);
    if (cg.d_hot_idx) hipFree(cg.d_hot_idx);
    if (cg.d_xhot) hipFree(cg.d_xhot);
  }
  A->grest.clear();
  if (A->o_rows) hipFree(A->o_rows);
  if (A->d_g_hot_idx) hipFree(A->d_g_hot_idx);
  if (A->d_g_xhot) hipFree(A->d_g_xhot);
  if (A->d_send_idx) hipFree(A->d_send_idx);
  if (A->d_hot_idx) hipFree(A->d_hot_idx);
  if (A->d_xhot) hipFree(A->d_xhot);
  if (A->d_ghost) hipFree(A->d_ghost);
  if (A->d_send) hipFree(A->d_send);
  if (A->h_ghost) hipHostFree(A->h_ghost);
  if (A->h_send) hipHostFree(A->h_send);
  delete A;
  return KSG_OK;
}

extern "C" int ksg_mat_sizes(ksg_mat A, int64_t *n, int64_t *N, int64_t *row0, int64_t *nnz) {
  if (!A) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_sizes: NULL matrix");
  if (n) *n = A->n;
  if (N) *N = A->N;
  if (row0) *row0 = A->row0;
  if (nnz) *nnz = A->nnz;
  return KSG_OK;
}

extern "C" int ksg_mat_hot_columns(ksg_mat A, int64_t *nhot) {
  if (!A || !nhot) KSG_FAIL(KSG_ERR_ARG, "ksg_mat_hot_columns: NULL argument");
  *nhot = A->nhot;
  for (const ksg_rest &cg : A->rest) *nhot += cg.nhot;
  return KSG_OK;
}

