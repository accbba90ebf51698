/* Device-vector operations: what the host logic calls on BV column views through
 * PETSc's Vec interface in the reference (VecCopy, VecAXPY, VecScale, VecSet,
 * VecDot, VecNorm) plus the deterministic start-vector fill. */
#include "ksg_internal.h"

__global__ __launch_bounds__(KSG_BLOCK) void k_vec_set(double *__restrict__ x, double a, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride) x[p] = a;
}
__global__ __launch_bounds__(KSG_BLOCK) void k_vec_scale(double *__restrict__ x, double a, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride) x[p] *= a;
}
__global__ __launch_bounds__(KSG_BLOCK) void k_vec_axpy(double *__restrict__ y, double a, const double *__restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride) y[p] = fma(a, x[p], y[p]);
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(KSG_BLOCK) void k_vec_fill_hash(double *__restrict__ x, int64_t n, int64_t row0, uint64_t seed) {
  const int64_t stride = (int64_t)gridDim.x * KSG_BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * KSG_BLOCK + threadIdx.x; p < n; p += stride) {
    const uint64_t z = splitmix64(((uint64_t)(row0 + p)) ^ seed);
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    x[p] = 2.0 * u - 1.0;
  }
}

#define VEC_ARGS(name) do { if (!c) KSG_FAIL(KSG_ERR_ARG, name ": NULL context"); if (n < 0) KSG_FAIL(KSG_ERR_ARG, name ": n<0"); } while (0)

extern "C" int ksg_vec_alloc(ksg_ctx c, int64_t n, double **x) {
  VEC_ARGS("ksg_vec_alloc");
  if (!x) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_alloc: NULL output");
  KSG_HIP(hipSetDevice(c->device));
  const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(double);
  hipError_t e = hipMalloc(x, bytes);
  if (e != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "ksg_vec_alloc: %zu bytes: %s", bytes, hipGetErrorString(e));
  KSG_HIP(hipMemsetAsync(*x, 0, bytes, c->stream));
  return KSG_OK;
}

extern "C" int ksg_vec_free(ksg_ctx c, double *x) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_free: NULL context");
  if (!x) return KSG_OK;
  KSG_HIP(hipSetDevice(c->device));
  KSG_HIP(hipStreamSynchronize(c->stream));
  KSG_HIP(hipFree(x));
  return KSG_OK;
}

extern "C" int ksg_vec_upload(ksg_ctx c, double *xd, const double *xh, int64_t n) {
  VEC_ARGS("ksg_vec_upload");
  if (!n) return KSG_OK;
  if (!xd || !xh) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_upload: NULL pointer");
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(ksg_flush_pending(c));
  c->gen++;
  KSG_HIP(hipMemcpyAsync(xd, xh, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  return KSG_OK;
}

extern "C" int ksg_vec_download(ksg_ctx c, double *xh, const double *xd, int64_t n) {
  VEC_ARGS("ksg_vec_download");
  if (!n) return KSG_OK;
  if (!xd || !xh) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_download: NULL pointer");
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(ksg_flush_pending(c));
  KSG_HIP(hipMemcpyAsync(xh, xd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  return KSG_OK;
}

extern "C" int ksg_vec_copy(ksg_ctx c, const double *x, double *y, int64_t n) {
  VEC_ARGS("ksg_vec_copy");
  if (!n || x == y) return KSG_OK;
  if (!x || !y) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_copy: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  KSG_HIP(hipMemcpyAsync(y, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return ksg_op_end(c, KSG_EV_VECOPS, 16.0 * (double)n, 0.0);
}

extern "C" int ksg_vec_set(ksg_ctx c, double *x, double a, int64_t n) {
  VEC_ARGS("ksg_vec_set");
  if (!n) return KSG_OK;
  if (!x) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_set: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  hipLaunchKernelGGL(k_vec_set, dim3(ksg_grid_for(n)), dim3(KSG_BLOCK), 0, c->stream, x, a, n);
  KSG_HIP(hipGetLastError());
  return ksg_op_end(c, KSG_EV_VECOPS, 8.0 * (double)n, 0.0);
}

extern "C" int ksg_vec_scale(ksg_ctx c, double *x, double a, int64_t n) {
  VEC_ARGS("ksg_vec_scale");
  if (!n || a == 1.0) return KSG_OK;
  if (!x) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_scale: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  hipLaunchKernelGGL(k_vec_scale, dim3(ksg_grid_for(n)), dim3(KSG_BLOCK), 0, c->stream, x, a, n);
  KSG_HIP(hipGetLastError());
  return ksg_op_end(c, KSG_EV_VECOPS, 16.0 * (double)n, (double)n);
}

extern "C" int ksg_vec_axpy(ksg_ctx c, double *y, double a, const double *x, int64_t n) {
  VEC_ARGS("ksg_vec_axpy");
  if (!n) return KSG_OK;
  if (!x || !y) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_axpy: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  hipLaunchKernelGGL(k_vec_axpy, dim3(ksg_grid_for(n)), dim3(KSG_BLOCK), 0, c->stream, y, a, x, n);
  KSG_HIP(hipGetLastError());
  return ksg_op_end(c, KSG_EV_VECOPS, 24.0 * (double)n, 2.0 * (double)n);
}

extern "C" int ksg_vec_dot(ksg_ctx c, const double *x, const double *y, int64_t n, double *result) {
  VEC_ARGS("ksg_vec_dot");
  if (!result || (n && (!x || !y))) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_dot: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  KSG_CHK(ksg_launch_dotvec(c, x, 2, 1, y, n, c->d_red));
  KSG_CHK(ksg_op_end(c, KSG_EV_VECOPS, 16.0 * (double)n, 2.0 * (double)n));
  return ksg_comm_allreduce(c, c->d_red, result, 1, 0);
}

/* the same product over the local rows only (the dot_local slot of a Vec type, which PETSc's
 * split-phase VecDotBegin/VecNormBegin sum over the ranks themselves) */
extern "C" int ksg_vec_dot_local(ksg_ctx c, const double *x, const double *y, int64_t n, double *result) {
  VEC_ARGS("ksg_vec_dot_local");
  if (!result || (n && (!x || !y))) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_dot_local: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  KSG_CHK(ksg_launch_dotvec(c, x, 2, 1, y, n, c->d_red));
  KSG_CHK(ksg_op_end(c, KSG_EV_VECOPS, (x == y ? 8.0 : 16.0) * (double)n, 2.0 * (double)n));
  KSG_HIP(hipMemcpyAsync(c->h_red, c->d_red, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  *result = c->h_red[0];
  return KSG_OK;
}

extern "C" int ksg_vec_norm2(ksg_ctx c, const double *x, int64_t n, double *result) {
  VEC_ARGS("ksg_vec_norm2");
  if (!result || (n && !x)) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_norm2: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  if ((((uintptr_t)x) & 15) == 0) { KSG_CHK(ksg_launch_sumsq(c, x, 2, 1, n, c->d_red)); }
  else { KSG_CHK(ksg_launch_dotvec(c, x, 2, 1, x, n, c->d_red)); }
  KSG_CHK(ksg_op_end(c, KSG_EV_VECOPS, 8.0 * (double)n, 2.0 * (double)n));
  double s = 0.0;
  KSG_CHK(ksg_comm_allreduce(c, c->d_red, &s, 1, 0));
  *result = sqrt(s);
  return KSG_OK;
}

extern "C" int ksg_vec_fill_hash(ksg_ctx c, double *x, int64_t n, int64_t row0, uint64_t seed) {
  VEC_ARGS("ksg_vec_fill_hash");
  if (!n) return KSG_OK;
  if (!x) KSG_FAIL(KSG_ERR_ARG, "ksg_vec_fill_hash: NULL pointer");
  KSG_CHK(ksg_op_begin(c, KSG_EV_VECOPS));
  hipLaunchKernelGGL(k_vec_fill_hash, dim3(ksg_grid_for(n)), dim3(KSG_BLOCK), 0, c->stream, x, n, row0, seed);
  KSG_HIP(hipGetLastError());
  return ksg_op_end(c, KSG_EV_VECOPS, 8.0 * (double)n, 0.0);
}
