/* Communicator of libksgpu: RCCL (loaded at run time, so that single-GPU users
 * need no RCCL at all) or caller-supplied collectives.
 *
 * The reference does every reduction with a host MPI_Allreduce after staging the
 * data through the host (sveccuda.cu:336-341, bvblas.c:252-260); here the
 * coefficients stay in device memory, are reduced by RCCL on the compute stream
 * and only then copied to the pinned host mirror the host logic reads. */
#include <dlfcn.h>

#include "ksg_internal.h"

/* the few RCCL declarations needed (ABI of rccl.h, NCCL 2.x) */
typedef struct { char internal[128]; } k_ncclUniqueId;
typedef void *k_ncclComm_t;
enum { k_ncclSum = 0, k_ncclMax = 2, k_ncclDouble = 8 };

struct rccl_api {
  void *handle;
  int (*GetUniqueId)(k_ncclUniqueId *);
  int (*CommInitRank)(k_ncclComm_t *, int, k_ncclUniqueId, int);
  int (*CommDestroy)(k_ncclComm_t);
  int (*AllReduce)(const void *, void *, size_t, int, int, k_ncclComm_t, hipStream_t);
  int (*Send)(const void *, size_t, int, int, k_ncclComm_t, hipStream_t);
  int (*Recv)(void *, size_t, int, int, k_ncclComm_t, hipStream_t);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
  const char *(*GetErrorString)(int);
};
static rccl_api g_rccl = {0};

static int load_rccl(void) {
  if (g_rccl.handle) return KSG_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", NULL};
  void *h = NULL;
  for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) KSG_FAIL(KSG_ERR_COMM, "cannot load RCCL: %s", dlerror());
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); \
    if (!g_rccl.field) KSG_FAIL(KSG_ERR_COMM, "RCCL symbol %s missing", name); } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.handle = h;
  return KSG_OK;
}

#define KSG_NCCL(call) do { int r_ = (call); if (r_ != 0) { \
    ksg_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
    return KSG_ERR_COMM; } } while (0)

extern "C" int ksg_comm_unique_id(void *id128) {
  if (!id128) KSG_FAIL(KSG_ERR_ARG, "NULL id buffer");
  KSG_CHK(load_rccl());
  k_ncclUniqueId id;
  KSG_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, 128);
  return KSG_OK;
}

extern "C" int ksg_ctx_comm_init_rccl(ksg_ctx c, int rank, int nranks, const void *id128) {
  if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
    KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_comm_init_rccl: bad arguments");
  if (c->comm_kind != KSG_COMM_NONE) KSG_FAIL(KSG_ERR_ARG, "communicator already set");
  KSG_CHK(load_rccl());
  KSG_HIP(hipSetDevice(c->device));
  k_ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  k_ncclComm_t comm = NULL;
  KSG_NCCL(g_rccl.CommInitRank(&comm, nranks, id, rank));
  c->rccl_comm = comm;
  c->comm_kind = KSG_COMM_RCCL;
  c->rank = rank;
  c->nranks = nranks;
  return KSG_OK;
}

extern "C" int ksg_ctx_comm_init_callbacks(ksg_ctx c, int rank, int nranks, int on_device,
                                           ksg_allreduce_fn allreduce, ksg_exchange_fn exchange,
                                           void *user) {
  if (!c || nranks < 1 || rank < 0 || rank >= nranks || !allreduce || !exchange)
    KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_comm_init_callbacks: bad arguments");
  if (c->comm_kind != KSG_COMM_NONE) KSG_FAIL(KSG_ERR_ARG, "communicator already set");
  c->comm_kind = KSG_COMM_CALLBACKS;
  c->rank = rank;
  c->nranks = nranks;
  c->cb_on_device = on_device ? 1 : 0;
  c->cb_allreduce = allreduce;
  c->cb_exchange = exchange;
  c->cb_user = user;
  return KSG_OK;
}

int ksg_comm_destroy(ksg_ctx c);

/* Leaves the communicator (collective); the context is single-rank again and may join
 * another one. */
extern "C" int ksg_ctx_comm_finalize(ksg_ctx c) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "NULL context");
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(ksg_flush_pending(c));
  KSG_HIP(hipStreamSynchronize(c->comm_stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  ksg_comm_destroy(c);
  c->rank = 0;
  c->nranks = 1;
  return KSG_OK;
}

extern "C" int ksg_ctx_comm_rank(ksg_ctx c, int *rank, int *nranks) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "NULL context");
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  return KSG_OK;
}

int ksg_comm_destroy(ksg_ctx c) {
  if (c->comm_kind == KSG_COMM_RCCL && c->rccl_comm) g_rccl.CommDestroy(c->rccl_comm);
  c->rccl_comm = NULL;
  c->comm_kind = KSG_COMM_NONE;
  return KSG_OK;
}

/* Reduce `count` doubles that a kernel left in d_buf (on the compute stream)
 * over all ranks and deliver them to the host array h_out.  Completed on
 * return. */
int ksg_comm_allreduce(ksg_ctx c, double *d_buf, double *h_out, int count, int op) {
  if (count > KSG_RED_MAX) KSG_FAIL(KSG_ERR_ARG, "reduction of %d values exceeds %d", count, KSG_RED_MAX);
  if (c->comm_kind == KSG_COMM_RCCL) {
    KSG_NCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, k_ncclDouble,
                              op ? k_ncclMax : k_ncclSum, c->rccl_comm, c->stream));
  } else if (c->comm_kind == KSG_COMM_CALLBACKS && c->nranks > 1 && c->cb_on_device) {
    KSG_HIP(hipStreamSynchronize(c->stream));
    if (c->cb_allreduce(c->cb_user, d_buf, count, op))
      KSG_FAIL(KSG_ERR_COMM, "allreduce callback failed");
  }
  KSG_HIP(hipMemcpyAsync(c->h_red, d_buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  if (c->comm_kind == KSG_COMM_CALLBACKS && c->nranks > 1 && !c->cb_on_device) {
    if (c->cb_allreduce(c->cb_user, c->h_red, count, op))
      KSG_FAIL(KSG_ERR_COMM, "allreduce callback failed");
  }
  memcpy(h_out, c->h_red, (size_t)count * sizeof(double));
  return KSG_OK;
}

/* Host-side reduction used while a matrix is being set up. */
int ksg_comm_allreduce_host(ksg_ctx c, double *h_buf, int count, int op) {
  if (c->nranks <= 1) return KSG_OK;
  if (count > KSG_RED_MAX) KSG_FAIL(KSG_ERR_ARG, "reduction of %d values exceeds %d", count, KSG_RED_MAX);
  if (c->comm_kind == KSG_COMM_CALLBACKS && !c->cb_on_device) {
    if (c->cb_allreduce(c->cb_user, h_buf, count, op)) KSG_FAIL(KSG_ERR_COMM, "allreduce callback failed");
    return KSG_OK;
  }
  KSG_HIP(hipMemcpyAsync(c->d_red, h_buf, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  KSG_CHK(ksg_comm_allreduce(c, c->d_red, h_buf, count, op));
  return KSG_OK;
}

/* All-to-all-v of doubles between device buffers, on `stream`. */
static int exchange_device(ksg_ctx c, const double *send, const int64_t *sendoff, double *recv,
                           const int64_t *recvoff, hipStream_t stream) {
  if (c->comm_kind == KSG_COMM_RCCL) {
    KSG_NCCL(g_rccl.GroupStart());
    for (int p = 0; p < c->nranks; p++) {
      int64_t ns = sendoff[p + 1] - sendoff[p], nr = recvoff[p + 1] - recvoff[p];
      if (ns > 0) KSG_NCCL(g_rccl.Send(send + sendoff[p], (size_t)ns, k_ncclDouble, p, c->rccl_comm, stream));
      if (nr > 0) KSG_NCCL(g_rccl.Recv(recv + recvoff[p], (size_t)nr, k_ncclDouble, p, c->rccl_comm, stream));
    }
    KSG_NCCL(g_rccl.GroupEnd());
    return KSG_OK;
  }
  /* device-pointer callbacks: make the packed data visible, then hand over */
  KSG_HIP(hipStreamSynchronize(stream));
  if (c->cb_exchange(c->cb_user, send, sendoff, recv, recvoff))
    KSG_FAIL(KSG_ERR_COMM, "exchange callback failed");
  return KSG_OK;
}

/* Every rank reports whether its staging buffers are ready (rc == 0); unless all are, all
 * return an error and none enters the collective that would follow. */
static int staging_agreed(ksg_ctx c, int rc, const char *what) {
  double flag = rc ? 1.0 : 0.0;
  const int rc2 = ksg_comm_allreduce_host(c, &flag, 1, 1);
  if (rc) return rc;
  if (rc2) return rc2;
  if (flag != 0.0) KSG_FAIL(KSG_ERR_COMM, "%s: another rank could not stage its buffers", what);
  return KSG_OK;
}

/* Self-test of the communicator (collective): one sum over the ranks with a known answer, and one
 * ring exchange -- every rank sends a block to the next rank and receives the previous rank's (its
 * own, on a single rank) -- through the same all-to-all-v path, on the communication stream, that
 * the halo exchange of a product uses. */
extern "C" int ksg_ctx_comm_selftest(ksg_ctx c) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "NULL context");
  if (c->comm_kind == KSG_COMM_NONE) return KSG_OK;
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(ksg_flush_pending(c));
  const int P = c->nranks, me = c->rank;
  double h[4] = {(double)(me + 1), 1.0, (double)me * (double)me, -1.0};
  KSG_CHK(ksg_comm_allreduce_host(c, h, 4, 0));
  const double want[4] = {0.5 * P * (P + 1.0), (double)P, (P - 1.0) * P * (2.0 * P - 1.0) / 6.0, -(double)P};
  for (int i = 0; i < 4; i++)
    if (h[i] != want[i])
      KSG_FAIL(KSG_ERR_COMM, "communicator self-test: sum %d over %d ranks is %.17g, expected %.17g", i, P, h[i], want[i]);
  const int64_t nblk = 4096;
  const int next = (me + 1) % P, prev = (me + P - 1) % P;
  std::vector<int64_t> soff((size_t)P + 1, 0), roff((size_t)P + 1, 0);
  for (int p = 0; p < P; p++) { soff[p + 1] = soff[p] + (p == next ? nblk : 0); roff[p + 1] = roff[p] + (p == prev ? nblk : 0); }
  std::vector<double> hs((size_t)nblk), hr((size_t)nblk, -1.0);
  for (int64_t i = 0; i < nblk; i++) hs[(size_t)i] = 1000.0 * me + (double)(i % 997);
  int rc = KSG_OK;
  if (c->comm_kind == KSG_COMM_CALLBACKS && !c->cb_on_device) {
    rc = ksg_comm_exchange_host(c, hs.data(), soff.data(), hr.data(), roff.data());
  } else {
    double *ds = NULL, *dr = NULL;
    hipError_t e = hipMalloc(&ds, (size_t)nblk * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&dr, (size_t)nblk * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(ds, hs.data(), (size_t)nblk * sizeof(double), hipMemcpyHostToDevice, c->comm_stream);
    if (e == hipSuccess) e = hipMemsetAsync(dr, 0xff, (size_t)nblk * sizeof(double), c->comm_stream);
    if (e != hipSuccess) { ksg_set_error("communicator self-test: staging failed: %s", hipGetErrorString(e)); rc = KSG_ERR_HIP; }
    /* the exchange is a collective: the ranks enter it only when the staging of all of them succeeded */
    rc = staging_agreed(c, rc, "communicator self-test");
    if (!rc) rc = exchange_device(c, ds, soff.data(), dr, roff.data(), c->comm_stream);
    if (!rc) {
      e = hipMemcpyAsync(hr.data(), dr, (size_t)nblk * sizeof(double), hipMemcpyDeviceToHost, c->comm_stream);
      if (e != hipSuccess) { ksg_set_error("communicator self-test: copy failed: %s", hipGetErrorString(e)); rc = KSG_ERR_HIP; }
    }
    hipStreamSynchronize(c->comm_stream);
    if (ds) hipFree(ds);
    if (dr) hipFree(dr);
  }
  if (rc) return rc;
  for (int64_t i = 0; i < nblk; i++)
    if (hr[(size_t)i] != 1000.0 * prev + (double)(i % 997))
      KSG_FAIL(KSG_ERR_COMM, "communicator self-test: entry %lld received from rank %d is %.17g, expected %.17g",
               (long long)i, prev, hr[(size_t)i], 1000.0 * prev + (double)(i % 997));
  return KSG_OK;
}

/* Halo exchange of one multiply: A->d_send has been packed on the comm stream. */
int ksg_comm_exchange(ksg_ctx c, ksg_mat A) {
  if (c->nranks <= 1) return KSG_OK;
  if (c->comm_kind == KSG_COMM_CALLBACKS && !c->cb_on_device) {
    if (A->nsend)
      KSG_HIP(hipMemcpyAsync(A->h_send, A->d_send, (size_t)A->nsend * sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
    KSG_HIP(hipStreamSynchronize(c->comm_stream));
    if (c->cb_exchange(c->cb_user, A->h_send, A->sendoff.data(), A->h_ghost, A->recvoff.data()))
      KSG_FAIL(KSG_ERR_COMM, "exchange callback failed");
    if (A->nghost)
      KSG_HIP(hipMemcpyAsync(A->d_ghost, A->h_ghost, (size_t)A->nghost * sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
    return KSG_OK;
  }
  return exchange_device(c, A->d_send, A->sendoff.data(), A->d_ghost, A->recvoff.data(), c->comm_stream);
}

/* The halo exchange of one multiply in rounds: every source's share of the ghost buffer is cut into
 * A->halo_rounds pieces, and round s moves piece s between every pair of ranks in one group -- on a
 * node whose GPUs are linked pairwise every link is busy in every round, and the whole exchange takes
 * what a single grouped one takes -- with an event recorded behind every round, so that the work on the
 * pieces that have arrived runs while the later ones travel (MatMult_MPIAIJ's scatter-begin / diagonal
 * / scatter-end / off-diagonal pattern behind src/sys/classes/st/impls/shift/shift.c:38, made finer).
 * (One source per round instead would use one link at a time.)  Piece s of a share of len entries is
 * [len*s/R, len*(s+1)/R): the same cut on the sending and on the receiving side.  Transports that
 * deliver everything in one call record all events after it. */
int ksg_comm_exchange_rounds(ksg_ctx c, ksg_mat A) {
  const int P = c->nranks, me = c->rank, R = A->halo_rounds;
  if (P <= 1) return KSG_OK;
  while ((int)c->ev_peer.size() < R) {
    hipEvent_t e;
    KSG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_peer.push_back(e);
  }
  if (c->comm_kind != KSG_COMM_RCCL) {
    KSG_CHK(ksg_comm_exchange(c, A));
    for (int s = 0; s < R; s++) KSG_HIP(hipEventRecord(c->ev_peer[(size_t)s], c->comm_stream));
    return KSG_OK;
  }
  for (int s = 0; s < R; s++) {
    KSG_NCCL(g_rccl.GroupStart());
    for (int p = 0; p < P; p++) {
      if (p == me) continue;
      const int64_t ls = A->sendoff[(size_t)p + 1] - A->sendoff[(size_t)p], lr = A->recvoff[(size_t)p + 1] - A->recvoff[(size_t)p];
      const int64_t s0 = ls * s / R, s1 = ls * (s + 1) / R, r0 = lr * s / R, r1 = lr * (s + 1) / R;
      if (s1 > s0) KSG_NCCL(g_rccl.Send(A->d_send + A->sendoff[(size_t)p] + s0, (size_t)(s1 - s0), k_ncclDouble, p, c->rccl_comm, c->comm_stream));
      if (r1 > r0) KSG_NCCL(g_rccl.Recv(A->d_ghost + A->recvoff[(size_t)p] + r0, (size_t)(r1 - r0), k_ncclDouble, p, c->rccl_comm, c->comm_stream));
    }
    KSG_NCCL(g_rccl.GroupEnd());
    KSG_HIP(hipEventRecord(c->ev_peer[(size_t)s], c->comm_stream));
  }
  return KSG_OK;
}

/* Set-up time exchange of host arrays (index lists encoded as doubles). */
int ksg_comm_exchange_host(ksg_ctx c, const double *send, const int64_t *sendoff, double *recv,
                           const int64_t *recvoff) {
  int P = c->nranks;
  if (P <= 1) return KSG_OK;
  if (c->comm_kind == KSG_COMM_CALLBACKS && !c->cb_on_device) {
    if (c->cb_exchange(c->cb_user, send, sendoff, recv, recvoff))
      KSG_FAIL(KSG_ERR_COMM, "exchange callback failed");
    return KSG_OK;
  }
  int64_t ns = sendoff[P], nr = recvoff[P];
  double *ds = NULL, *dr = NULL;
  int rc = KSG_OK;
  hipError_t e = hipMalloc(&ds, (size_t)(ns > 0 ? ns : 1) * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dr, (size_t)(nr > 0 ? nr : 1) * sizeof(double));
  if (e == hipSuccess && ns) e = hipMemcpyAsync(ds, send, (size_t)ns * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) { ksg_set_error("halo set-up staging failed: %s", hipGetErrorString(e)); rc = KSG_ERR_HIP; }
  /* the exchange is a collective: the ranks enter it only when the staging of all of them
   * succeeded (the flag travels through the reduction buffer the context already owns) */
  rc = staging_agreed(c, rc, "halo set-up");
  if (!rc) rc = exchange_device(c, ds, sendoff, dr, recvoff, c->stream);
  if (!rc && nr) {
    e = hipMemcpyAsync(recv, dr, (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) { ksg_set_error("halo set-up copy failed: %s", hipGetErrorString(e)); rc = KSG_ERR_HIP; }
  }
  hipStreamSynchronize(c->stream);
  if (ds) hipFree(ds);
  if (dr) hipFree(dr);
  return rc;
}
