/* Context, error reporting, workspaces and per-event statistics of libksgpu. */
#include <stdarg.h>
#include <stdlib.h>

#include "ksg_internal.h"

static thread_local char g_err[512] = "";

void ksg_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *ksg_last_error(void) { return g_err; }

int ksg_host_threads(void) {
  int n = 0;
  const char *e = getenv("KSG_HOST_THREADS");
  if (e && atoi(e) > 0) n = atoi(e);
  if (!n) { e = getenv("OMP_NUM_THREADS"); if (e && atoi(e) > 0) n = atoi(e); }
  if (!n) n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  return n > 16 ? 16 : n;
}

int ksg_comm_destroy(ksg_ctx c);
extern "C" int ksg_ctx_destroy(ksg_ctx c);

/* Streams, events and the small fixed buffers of a context whose plain fields are set. */
static int ctx_init_device(ksg_ctx c) {
  KSG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  KSG_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  KSG_HIP(hipEventCreateWithFlags(&c->ev_x_ready, hipEventDisableTiming));
  KSG_HIP(hipEventCreateWithFlags(&c->ev_halo_done, hipEventDisableTiming));
  KSG_HIP(hipMalloc(&c->d_red, KSG_RED_MAX * sizeof(double)));
  KSG_HIP(hipHostMalloc(&c->h_red, KSG_RED_MAX * sizeof(double), hipHostMallocDefault));
  KSG_HIP(hipMalloc(&c->d_spec, (KSG_QMAX + 1) * sizeof(double)));
  KSG_HIP(hipMalloc(&c->d_batch, KSG_RED_MAX * sizeof(double)));
  KSG_CHK(ksg_scratch_bind(c));
  return KSG_OK;
}

extern "C" int ksg_ctx_create(int device, ksg_ctx *out) {
  if (!out) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_create: NULL output");
  int ndev = 0;
  KSG_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev)
    KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_create: device %d out of range (%d visible)", device, ndev);
  KSG_HIP(hipSetDevice(device));
  ksg_ctx c = new (std::nothrow) ksg_ctx_s();       /* value-initialised: every pointer NULL */
  if (!c) KSG_FAIL(KSG_ERR_MEM, "ksg_ctx_create: out of host memory");
  c->device = device;
  c->comm_kind = KSG_COMM_NONE;
  c->rank = 0;
  c->nranks = 1;
  c->spec.valid = false;
  c->fuse_dots = 1;
  c->lazy_update = 1;
  c->hot_cols = 1048576;
  { const char *e = getenv("KSG_SPMV_BINS"); c->bins = e ? atoi(e) : 0; if (c->bins < 0 || c->bins > 2) c->bins = 0; }
  c->pend.active = false;
  c->dot_grid = KSG_DOT_GRID;
  c->mv_grid = KSG_MV_GRID;
  memset(&c->stats, 0, sizeof(c->stats));
  const int rc = ctx_init_device(c);
  if (rc) { ksg_ctx_destroy(c); return rc; }        /* frees whatever was created (the message is kept) */
  *out = c;
  return KSG_OK;
}

extern "C" int ksg_ctx_sync(ksg_ctx c) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_sync: NULL context");
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(ksg_flush_pending(c));
  KSG_HIP(hipStreamSynchronize(c->comm_stream));
  KSG_HIP(hipStreamSynchronize(c->stream));
  return KSG_OK;
}

extern "C" int ksg_ctx_flush(ksg_ctx c) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_flush: NULL context");
  KSG_HIP(hipSetDevice(c->device));
  return ksg_flush_pending(c);
}

extern "C" int ksg_ctx_touch(ksg_ctx c) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_touch: NULL context");
  KSG_HIP(hipSetDevice(c->device));
  KSG_CHK(ksg_flush_pending(c));
  c->gen++;                       /* reductions formed ahead of their request are stale now */
  return KSG_OK;
}

extern "C" void *ksg_ctx_stream(ksg_ctx c) { return c ? (void *)c->stream : NULL; }

extern "C" int ksg_ctx_set_option(ksg_ctx c, int option, int value) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_set_option: NULL context");
  switch (option) {
    case KSG_OPT_FUSE_DOTS: KSG_CHK(ksg_flush_pending(c)); c->fuse_dots = value ? 1 : 0; c->spec.valid = false; return KSG_OK;
    case KSG_OPT_LAZY_UPDATE: KSG_CHK(ksg_flush_pending(c)); c->lazy_update = value ? 1 : 0; return KSG_OK;
    case KSG_OPT_HOT_COLS:
      if (value < 0) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_set_option: negative size");
      c->hot_cols = value;
      return KSG_OK;
    case KSG_OPT_BINS:
      if (value < 0 || value > 2) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_set_option: binned form %d (0: by size, 1: never, 2: always)", value);
      c->bins = value;
      return KSG_OK;
    case KSG_OPT_DOT_GRID:
    case KSG_OPT_MV_GRID:
      if (value < 1 || value > 65535) KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_set_option: grid %d out of range", value);
      (option == KSG_OPT_DOT_GRID ? c->dot_grid : c->mv_grid) = value;
      return KSG_OK;
    default: KSG_FAIL(KSG_ERR_ARG, "ksg_ctx_set_option: unknown option %d", option);
  }
}

extern "C" int ksg_ctx_destroy(ksg_ctx c) {
  if (!c) return KSG_OK;
  hipSetDevice(c->device);
  if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
  if (c->stream) hipStreamSynchronize(c->stream);
  ksg_comm_destroy(c);
  for (auto &t : c->timers) { hipEventDestroy(t.e0); hipEventDestroy(t.e1); }
  for (auto &e : c->event_pool) hipEventDestroy(e);
  if (c->d_partial) hipFree(c->d_partial);
  if (c->d_q) hipFree(c->d_q);
  if (c->d_work) hipFree(c->d_work);
  if (c->h_q) hipHostFree(c->h_q);
  if (c->d_red) hipFree(c->d_red);
  if (c->d_spec) hipFree(c->d_spec);
  if (c->d_batch) hipFree(c->d_batch);
  if (c->h_red) hipHostFree(c->h_red);
  if (c->ev_x_ready) hipEventDestroy(c->ev_x_ready);
  if (c->ev_halo_done) hipEventDestroy(c->ev_halo_done);
  for (hipEvent_t e : c->ev_peer) hipEventDestroy(e);
  c->ev_peer.clear();
  if (c->comm_stream) hipStreamDestroy(c->comm_stream);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return KSG_OK;
}

static int ensure(ksg_ctx c, double **p, size_t *cap, size_t doubles) {
  if (*cap >= doubles) return KSG_OK;
  KSG_HIP(hipStreamSynchronize(c->stream));
  if (*p) KSG_HIP(hipFree(*p));
  *p = NULL; *cap = 0;
  size_t want = doubles + doubles / 4 + 1024;
  hipError_t e = hipMalloc(p, want * sizeof(double));
  if (e != hipSuccess) KSG_FAIL(KSG_ERR_MEM, "device workspace of %zu bytes: %s", want * sizeof(double), hipGetErrorString(e));
  *cap = want;
  return KSG_OK;
}
int ksg_ensure_partial(ksg_ctx c, size_t d) { return ensure(c, &c->d_partial, &c->partial_cap, d); }
int ksg_ensure_q(ksg_ctx c, size_t d) { return ensure(c, &c->d_q, &c->q_cap, d); }
int ksg_ensure_work(ksg_ctx c, size_t d) { return ensure(c, &c->d_work, &c->work_cap, d); }

/* ---- statistics ------------------------------------------------------- */
static int get_event(ksg_ctx c, hipEvent_t *e) {
  if (!c->event_pool.empty()) { *e = c->event_pool.back(); c->event_pool.pop_back(); return KSG_OK; }
  KSG_HIP(hipEventCreate(e));
  return KSG_OK;
}

static int drain_timers(ksg_ctx c) {
  for (auto &t : c->timers) {
    float ms = 0.f;
    KSG_HIP(hipEventSynchronize(t.e1));
    KSG_HIP(hipEventElapsedTime(&ms, t.e0, t.e1));
    c->stats.ms[t.ev] += ms;
    c->event_pool.push_back(t.e0);
    c->event_pool.push_back(t.e1);
  }
  c->timers.clear();
  return KSG_OK;
}

int ksg_op_begin(ksg_ctx c, int ev) { return ksg_op_begin_ex(c, ev, 0); }

int ksg_op_begin_ex(ksg_ctx c, int ev, int flags) {
  KSG_HIP(hipSetDevice(c->device));
  if (!(flags & KSG_OPB_NOFLUSH) && c->pend.active) KSG_CHK(ksg_flush_pending(c));
  /* everything but the pure reductions may write vectors */
  if (!(flags & KSG_OPB_NOBUMP) && ev != KSG_EV_DOTVEC && ev != KSG_EV_NORM && ev != KSG_EV_DOT) c->gen++;
  if (c->timing) {
    ksg_timer t; t.ev = ev;
    KSG_CHK(get_event(c, &t.e0));
    KSG_CHK(get_event(c, &t.e1));
    KSG_HIP(hipEventRecord(t.e0, c->stream));
    c->timers.push_back(t);
  }
  return KSG_OK;
}

int ksg_op_end(ksg_ctx c, int ev, double bytes, double flops) {
  c->stats.calls[ev] += 1;
  c->stats.bytes[ev] += bytes;
  c->stats.flops[ev] += flops;
  if (c->timing) {
    KSG_HIP(hipEventRecord(c->timers.back().e1, c->stream));
    if (c->timers.size() > 4096) KSG_CHK(drain_timers(c));
  }
  return KSG_OK;
}

extern "C" int ksg_stats_enable_timing(ksg_ctx c, int on) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "NULL context");
  if (!on) KSG_CHK(drain_timers(c));
  c->timing = on ? 1 : 0;
  return KSG_OK;
}

extern "C" int ksg_stats_get(ksg_ctx c, ksg_stats *out) {
  if (!c || !out) KSG_FAIL(KSG_ERR_ARG, "NULL argument");
  KSG_CHK(drain_timers(c));
  *out = c->stats;
  return KSG_OK;
}

extern "C" int ksg_stats_reset(ksg_ctx c) {
  if (!c) KSG_FAIL(KSG_ERR_ARG, "NULL context");
  KSG_CHK(drain_timers(c));
  memset(&c->stats, 0, sizeof(c->stats));
  return KSG_OK;
}
