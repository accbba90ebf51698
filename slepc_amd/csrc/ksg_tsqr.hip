/* Tall-skinny QR of the first k columns of a basis: V = Q R, Q^T Q = I, R upper
 * triangular with a positive diagonal -- the block orthogonalisation the reference
 * offers as BV_ORTHOG_BLOCK_TSQR (BVOrthogonalize_TSQR, bvorthog.c:770-782, on
 * BVOrthogonalize_LAPACK_Private, bvblas.c:345-432: a Householder QR of the local
 * rows with geqrf/orgqr, and for two processes a second QR of the stacked
 * triangles; more than two processes are refused there).
 *
 * Here R comes from a reduction tree of Householder factorisations and Q from
 * R, never from a Gram matrix:
 *   leaf    : a workgroup walks its blocks of 64 rows, keeps a k x k triangle in
 *             LDS and folds every block into it with k structured Householder
 *             reflections (the reflector of column j touches the diagonal entry
 *             j and the 64 rows of the block: one lane per row, DPP lane swaps
 *             for the inner products, one barrier per reflection).  The block
 *             after the one being folded is already on its way from HBM.  Every
 *             basis element is read once; nothing is written but one triangle
 *             per workgroup.
 *   merge   : the triangles of the workgroups, then those of the ranks (any
 *             number of them), are folded pairwise on the host, again by
 *             Householder reflections that use their structure.
 *   apply   : V <- V R^-1 with the restart kernel (k_multinplace).
 * The loss of orthogonality of V R^-1 grows with the condition number of V, so the
 * three steps run twice and the two triangles are multiplied: after the second
 * sweep Q is orthogonal to working precision for any V that is numerically of
 * full rank, which is the property the Householder method gives the reference.  A
 * basis that is numerically rank deficient (a diagonal entry of R below 1000 eps
 * times the largest) is refused with an error instead of being completed by
 * arbitrary directions; when that is found in the first sweep V is unchanged. */
#include <math.h>
#include <stdlib.h>

#include "ksg_internal.h"

#define TSQR_ROWS 64                 /* rows of a block = lanes of a wave */
#define TSQR_KMAX 64                 /* most columns */
#define TSQR_TRI(k) ((k) * ((k) + 1) / 2)

/* Sum over the 64 lanes, the same bits in every lane.  Inside a row of 16 lanes the
 * partners are reached with DPP lane swaps (xor 1, xor 2, then the mirrors of the half row and
 * of the row, which pair groups that already hold equal values); the four row sums are read
 * as scalars and added in a fixed order.  No trip through the LDS crossbar as with
 * __shfl_xor, whose six dependent rounds dominated the time of a reflection.  All 64 lanes
 * must be active. */
template <int CTRL>
__device__ __forceinline__ double dpp_swap(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_allsum(double v) {
  v += dpp_swap<0xB1>(v);        /* quad_perm [1,0,3,2] */
  v += dpp_swap<0x4E>(v);        /* quad_perm [2,3,0,1] */
  v += dpp_swap<0x141>(v);       /* row_half_mirror     */
  v += dpp_swap<0x140>(v);       /* row_mirror          */
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

/* packed upper triangle, column by column: (i, j), i <= j */
__host__ __device__ __forceinline__ int tri_idx(int i, int j) { return j * (j + 1) / 2 + i; }

__global__ __launch_bounds__(KSG_BLOCK) void k_tsqr_leaf(const double *__restrict__ X, int64_t ld, int k, int64_t n,
                                                          double *__restrict__ Rout, int64_t ldo) {
  __shared__ double blk[TSQR_KMAX * TSQR_ROWS];       /* the block, column c at blk + 64 c */
  __shared__ double Rp[TSQR_TRI(TSQR_KMAX)];
  constexpr int NW = KSG_BLOCK / 64;
  constexpr int CPW = TSQR_KMAX / NW;                  /* columns a wave parks */
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tri = TSQR_TRI(k);
  for (int i = threadIdx.x; i < tri; i += KSG_BLOCK) Rp[i] = 0.0;
  const int64_t nblk = (n + TSQR_ROWS - 1) / TSQR_ROWS;
  double pre[CPW];
  {
    const int64_t r = (int64_t)blockIdx.x * TSQR_ROWS + lane;
#pragma unroll
    for (int i = 0; i < CPW; i++) {
      const int c = w + NW * i;
      pre[i] = (c < k && r < n) ? __builtin_nontemporal_load(X + (int64_t)c * ld + r) : 0.0;
    }
  }
  /* the trip counts of both loops are the same for every thread of the workgroup */
  for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    __syncthreads();                                   /* the previous block is folded, Rp is initialised */
#pragma unroll
    for (int i = 0; i < CPW; i++) {
      const int c = w + NW * i;
      if (c < k) blk[c * TSQR_ROWS + lane] = pre[i];
    }
    {
      const int64_t r = (b + gridDim.x) * TSQR_ROWS + lane;
#pragma unroll
      for (int i = 0; i < CPW; i++) {
        const int c = w + NW * i;
        pre[i] = (c < k && r < n) ? __builtin_nontemporal_load(X + (int64_t)c * ld + r) : 0.0;
      }
    }
    __syncthreads();
    int pend = -1;                                     /* diagonal entry wave 0 still has to write */
    double pend_beta = 0.0;
    for (int j = 0; j < k; j++) {
      /* every wave has read the diagonal entry of step `pend` before the barrier that ended it */
      if (w == 0 && lane == 0 && pend >= 0) Rp[tri_idx(pend, pend)] = pend_beta;
      pend = -1;
      const double xr = blk[j * TSQR_ROWS + lane];
      const double x0 = Rp[tri_idx(j, j)];
      const double sigma = wave_allsum(xr * xr);
      if (sigma != 0.0) {                              /* wave-uniform, the same in every wave */
        const double mu = sqrt(fma(x0, x0, sigma));
        const double beta = x0 >= 0.0 ? -mu : mu;
        const double tau = (beta - x0) / beta;
        const double vr = xr / (x0 - beta);
        /* four columns at a time: their reductions are independent and overlap */
        for (int c0 = j + 1 + w; c0 < k; c0 += 4 * NW) {
          double a[4], d[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int c = c0 + u * NW;
            a[u] = c < k ? blk[c * TSQR_ROWS + lane] : 0.0;
            d[u] = vr * a[u];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) d[u] = wave_allsum(d[u]);
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int c = c0 + u * NW;
            if (c < k) {
              const double rjc = Rp[tri_idx(j, c)];
              const double wv = tau * (rjc + d[u]);
              blk[c * TSQR_ROWS + lane] = fma(-vr, wv, a[u]);
              if (lane == 0) Rp[tri_idx(j, c)] = rjc - wv;
            }
          }
        }
        pend = j;
        pend_beta = beta;
      }
      __syncthreads();
    }
    if (w == 0 && lane == 0 && pend >= 0) Rp[tri_idx(pend, pend)] = pend_beta;
  }
  __syncthreads();
  if (ldo == 0) {                                       /* packed triangles, one after the other */
    for (int i = threadIdx.x; i < tri; i += KSG_BLOCK) Rout[(int64_t)blockIdx.x * tri + i] = Rp[i];
  } else {                                              /* k dense rows of a matrix with leading dimension ldo */
    for (int e = threadIdx.x; e < k * k; e += KSG_BLOCK) {
      const int i = e % k, j = e / k;
      Rout[(int64_t)blockIdx.x * k + i + (int64_t)j * ldo] = i <= j ? Rp[tri_idx(i, j)] : 0.0;
    }
  }
}

/* [R; T] = Q' [R'; 0] for two packed k x k upper triangles; R is overwritten by R', T is
 * destroyed.  The reflector of column j spans the diagonal entry (j, j) of R and the rows
 * 0..j of T, the only ones where column j of T is not zero. */
static void fold_triangle(int k, double *R, double *T) {
  double v[TSQR_KMAX];
  for (int j = 0; j < k; j++) {
    double *tj = T + tri_idx(0, j);
    double sigma = 0.0;
    for (int i = 0; i <= j; i++) sigma += tj[i] * tj[i];
    if (sigma == 0.0) continue;
    const double x0 = R[tri_idx(j, j)];
    const double mu = sqrt(x0 * x0 + sigma);
    const double beta = x0 >= 0.0 ? -mu : mu;
    const double tau = (beta - x0) / beta;
    const double sc = 1.0 / (x0 - beta);
    for (int i = 0; i <= j; i++) v[i] = tj[i] * sc;
    R[tri_idx(j, j)] = beta;
    for (int c = j + 1; c < k; c++) {
      double *tc = T + tri_idx(0, c);
      double d = R[tri_idx(j, c)];
      for (int i = 0; i <= j; i++) d += v[i] * tc[i];
      const double wv = tau * d;
      R[tri_idx(j, c)] -= wv;
      for (int i = 0; i <= j; i++) tc[i] -= v[i] * wv;
    }
  }
}

/* One sweep: the triangle of the first k columns over all ranks, as a dense k x k upper
 * triangular matrix with a positive diagonal (leading dimension k). */
static int tsqr_triangle(ksg_bv V, int k, double *Rd) {
  ksg_ctx c = V->ctx;
  const int tri = TSQR_TRI(k);
  const int64_t nblk = (V->n + TSQR_ROWS - 1) / TSQR_ROWS;
  const int grid = (int)(nblk < 768 ? (nblk > 0 ? nblk : 1) : 768);   /* three workgroups per CU */
  /* many workgroups: their triangles, stacked as a (grid k) x k matrix, go through the same
   * kernel once more and only a few triangles are folded on the host */
  const int grid2 = grid > 256 ? 64 : (grid > 16 ? 8 : 0);   /* at most 12 blocks per workgroup there */
  const int nt = grid2 ? grid2 : grid;
  std::vector<double> T((size_t)nt * tri, 0.0), R((size_t)tri, 0.0);
  if (V->n > 0) {
    const int64_t ldo = (int64_t)grid * k;
    KSG_CHK(ksg_ensure_partial(c, grid2 ? (size_t)ldo * k + (size_t)grid2 * tri : (size_t)grid * tri));
    double *d_tri = c->d_partial + (grid2 ? (size_t)ldo * k : 0);
    hipLaunchKernelGGL(k_tsqr_leaf, dim3(grid), dim3(KSG_BLOCK), 0, c->stream, V->d + (int64_t)V->nc * V->ld, V->ld, k,
                       V->n, c->d_partial, grid2 ? ldo : (int64_t)0);
    if (grid2)
      hipLaunchKernelGGL(k_tsqr_leaf, dim3(grid2), dim3(KSG_BLOCK), 0, c->stream, (const double *)c->d_partial, ldo, k,
                         ldo, d_tri, (int64_t)0);
    KSG_HIP(hipGetLastError());
    KSG_HIP(hipMemcpyAsync(T.data(), d_tri, (size_t)nt * tri * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    KSG_HIP(hipStreamSynchronize(c->stream));
    memcpy(R.data(), T.data(), (size_t)tri * sizeof(double));
    for (int g = 1; g < nt; g++) fold_triangle(k, R.data(), T.data() + (size_t)g * tri);
  }
  if (c->nranks > 1) {
    /* every rank receives the triangles of the others and folds all of them in rank order:
     * the same R everywhere.  Nothing is sent to the rank itself. */
    const int P = c->nranks;
    std::vector<int64_t> off(P + 1);
    off[0] = 0;
    for (int p = 0; p < P; p++) off[p + 1] = off[p] + (p == c->rank ? 0 : tri);
    std::vector<double> send((size_t)off[P] + 1), recv((size_t)off[P] + 1), own(R);
    for (int p = 0; p < P; p++)
      if (p != c->rank) memcpy(send.data() + off[p], own.data(), (size_t)tri * sizeof(double));
    KSG_CHK(ksg_comm_exchange_host(c, send.data(), off.data(), recv.data(), off.data()));
    for (int p = 0; p < P; p++) {
      double *Tp = p == c->rank ? own.data() : recv.data() + off[p];
      if (p == 0) memcpy(R.data(), Tp, (size_t)tri * sizeof(double));
      else fold_triangle(k, R.data(), Tp);
    }
  }
  for (int j = 0; j < k; j++)
    for (int i = 0; i < k; i++) Rd[i + (size_t)j * k] = i <= j ? R[tri_idx(i, j)] : 0.0;
  for (int i = 0; i < k; i++)
    if (Rd[i + (size_t)i * k] < 0.0)
      for (int j = i; j < k; j++) Rd[i + (size_t)j * k] = -Rd[i + (size_t)j * k];
  return KSG_OK;
}

/* S = R^-1 for an upper triangular k x k matrix (column by column back substitution) */
static void tri_inverse(int k, const double *R, double *S) {
  memset(S, 0, (size_t)k * k * sizeof(double));
  for (int j = 0; j < k; j++) {
    S[j + (size_t)j * k] = 1.0 / R[j + (size_t)j * k];
    for (int i = j - 1; i >= 0; i--) {
      double s = 0.0;
      for (int t = i + 1; t <= j; t++) s += R[i + (size_t)t * k] * S[t + (size_t)j * k];
      S[i + (size_t)j * k] = -s / R[i + (size_t)i * k];
    }
  }
}

extern "C" int ksg_bv_tsqr(ksg_bv V, double *R_host, int ldr) {
  if (!V) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_tsqr: NULL basis");
  const int k = V->k;
  if (V->nc) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_tsqr: not available for a basis with constraints");
  if (k < 1 || k > TSQR_KMAX) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_tsqr: %d active columns, 1 to %d supported", k, TSQR_KMAX);
  if (R_host && ldr < k) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_tsqr: ldr=%d smaller than the %d columns", ldr, k);
  if (V->N < k) KSG_FAIL(KSG_ERR_ARG, "ksg_bv_tsqr: %d columns of length %lld", k, (long long)V->N);
  ksg_ctx c = V->ctx;
  KSG_HIP(hipSetDevice(c->device));
  const int l0 = V->l;
  std::vector<double> R1((size_t)k * k), R2((size_t)k * k), S((size_t)k * k);
  V->l = 0;
  int rc = KSG_OK;
  for (int sweep = 0; sweep < 2 && !rc; sweep++) {
    double *Rs = sweep ? R2.data() : R1.data();
    rc = ksg_op_begin(c, KSG_EV_DOT);
    if (rc) break;
    rc = tsqr_triangle(V, k, Rs);
    if (rc) break;
    rc = ksg_op_end(c, KSG_EV_DOT, 8.0 * (double)V->n * k, 2.0 * (double)V->n * k * k);
    if (rc) break;
    double dmax = 0.0, dmin = HUGE_VAL;
    for (int i = 0; i < k; i++) {
      const double d = Rs[i + (size_t)i * k];
      dmax = fmax(dmax, d);
      dmin = fmin(dmin, d);
    }
    if (!(dmin > 0.0) || dmin <= 1000.0 * 2.2204460492503131e-16 * dmax) {
      ksg_set_error("ksg_bv_tsqr: the columns are numerically linearly dependent (diagonal of R from %g to %g)", dmin, dmax);
      rc = KSG_ERR_ARG;
      break;
    }
    tri_inverse(k, Rs, S.data());
    rc = ksg_bv_multinplace(V, S.data(), k, 0, k, 0);
  }
  V->l = l0;
  if (rc) return rc;
  if (R_host) {
    /* R = R2 R1 */
    for (int j = 0; j < k; j++)
      for (int i = 0; i < k; i++) {
        double s = 0.0;
        for (int t = i; t <= j; t++) s += R2[i + (size_t)t * k] * R1[t + (size_t)j * k];
        R_host[i + (size_t)j * ldr] = i <= j ? s : 0.0;
      }
  }
  return KSG_OK;
}
