// Data-parallel gradient exchange: a thin RCCL communicator behind the C-ABI (include/awt.h "awt_comm").
// Build-defined (the reference has no distributed code: SURVEY.md §5, §8a a17, §8e): one process per GPU, one flat fp32
// buffer of adapter gradients, ncclAllReduce in place over xGMI.  librccl is dlopen'ed on first use so that libawt.so
// loads (and its exports can be checked) on hosts without RCCL or without a GPU.
#include <dlfcn.h>
#include <mutex>

#include "common.h"
#include "comm.h"

namespace {

// the five RCCL entry points used, with their rccl.h signatures (ncclUniqueId is a 128-byte struct passed by value)
struct UniqueId { char internal[AWT_COMM_ID_BYTES]; };
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s);
typedef const char* (*fn_error_string)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0, kNcclAvg = 4;   // rccl.h: ncclDataType_t / ncclRedOp_t

struct Rccl {
  void* so = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_error_string error_string = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.so) break;
    }
    if (!r.so) { r.why = std::string("librccl.so could not be loaded: ") + dlerror(); return; }
    r.get_unique_id = (fn_get_unique_id)dlsym(r.so, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.so, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.so, "ncclCommDestroy");
    r.all_reduce = (fn_all_reduce)dlsym(r.so, "ncclAllReduce");
    r.error_string = (fn_error_string)dlsym(r.so, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.error_string) {
      r.why = "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetErrorString";
      r.so = nullptr;
    }
  });
  return r.so ? &r : nullptr;
}

int rccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  return awt_fail(AWT_ERR_HIP, std::string(what) + ": " + (r ? r->error_string(rc) : "rccl unavailable"));
}

int reduce(awt_comm* m, float* buf, size_t n, int op, hipStream_t s, const char* what) {
  AWT_REQUIRE(m && m->nccl, AWT_ERR_INVALID, std::string(what) + ": null communicator");
  AWT_REQUIRE(buf && n > 0, AWT_ERR_INVALID, std::string(what) + ": empty buffer");
  AWT_REQUIRE(((uintptr_t)buf & 3) == 0, AWT_ERR_INVALID, std::string(what) + ": buffer must be 4-byte aligned");
  const int rc = rccl()->all_reduce(buf, buf, n, kNcclFloat32, op, m->nccl, s);
  if (rc) return rccl_fail(what, rc);
  return AWT_OK;
}

}  // namespace

extern "C" int awt_comm_unique_id(void* id_out) {
  AWT_REQUIRE(id_out, AWT_ERR_INVALID, "comm_unique_id: null output");
  Rccl* r = rccl();
  if (!r) return awt_fail(AWT_ERR_STATE, "comm_unique_id: RCCL is not available on this host");
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, AWT_COMM_ID_BYTES);
  return AWT_OK;
}

extern "C" int awt_comm_create(awt_ctx* c, const void* id, int rank, int world, awt_comm** out) {
  AWT_REQUIRE(c && id && out, AWT_ERR_INVALID, "comm_create: null argument");
  AWT_REQUIRE(world >= 1 && rank >= 0 && rank < world, AWT_ERR_INVALID, "comm_create: rank must be in [0, world)");
  Rccl* r = rccl();
  if (!r) return awt_fail(AWT_ERR_STATE, "comm_create: RCCL is not available on this host");
  AWT_HIP_CHECK(hipSetDevice(c->device));
  awt_comm* m = new awt_comm();
  m->ctx = c; m->rank = rank; m->world = world;
  UniqueId uid;
  memcpy(uid.internal, id, AWT_COMM_ID_BYTES);
  int rc = r->comm_init_rank(&m->nccl, world, uid, rank);
  if (rc) { delete m; return rccl_fail("ncclCommInitRank", rc); }
  if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess) {
    r->comm_destroy(m->nccl); delete m;
    return awt_fail(AWT_ERR_HIP, "comm_create: could not create the side stream");
  }
  for (hipEvent_t& ev : m->ev)
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { awt_comm_destroy(m); return awt_fail(AWT_ERR_HIP, "comm_create: could not create an event"); }
  *out = m;
  return AWT_OK;
}

extern "C" void awt_comm_destroy(awt_comm* m) {
  if (!m) return;
  if (m->side) { (void)hipStreamSynchronize(m->side); (void)hipStreamDestroy(m->side); }
  for (hipEvent_t ev : m->ev) if (ev) (void)hipEventDestroy(ev);
  for (awt_comm::Span& sp : m->spans) { if (sp.a) (void)hipEventDestroy(sp.a); if (sp.b) (void)hipEventDestroy(sp.b); }
  if (m->nccl && rccl()) rccl()->comm_destroy(m->nccl);
  delete m;
}

extern "C" int awt_comm_world(const awt_comm* m) { return m ? m->world : 0; }

extern "C" int awt_allreduce_sum_f32(awt_comm* m, float* buf, size_t n, void* stream) {
  return reduce(m, buf, n, kNcclSum, (hipStream_t)stream, "allreduce_sum_f32");
}
extern "C" int awt_allreduce_mean_f32(awt_comm* m, float* buf, size_t n, void* stream) {
  return reduce(m, buf, n, kNcclAvg, (hipStream_t)stream, "allreduce_mean_f32");
}

// `buf[0 : n)` is final on `producer` as of now: average it over the ranks on the side stream, behind that point
int comm_reduce_async(awt_comm* m, float* buf, size_t n, hipStream_t producer) {
  hipEvent_t ev = m->ev[m->next_ev];
  m->next_ev = (m->next_ev + 1) % (int)(sizeof(m->ev) / sizeof(m->ev[0]) - 1);
  AWT_HIP_CHECK(hipEventRecord(ev, producer));
  AWT_HIP_CHECK(hipStreamWaitEvent(m->side, ev, 0));
  awt_comm::Span* sp = (m->timing && m->n_spans < (int)(sizeof(m->spans) / sizeof(m->spans[0]))) ? &m->spans[m->n_spans] : nullptr;
  if (sp) {
    if (!sp->a) { AWT_HIP_CHECK(hipEventCreate(&sp->a)); AWT_HIP_CHECK(hipEventCreate(&sp->b)); }
    AWT_HIP_CHECK(hipEventRecord(sp->a, m->side));
  }
  const int rc = reduce(m, buf, n, kNcclAvg, m->side, "allreduce_mean_f32 (side stream)");
  if (sp && !rc) { AWT_HIP_CHECK(hipEventRecord(sp->b, m->side)); sp->bytes = n * 4; ++m->n_spans; }
  return rc;
}

// Bucket timing of the in-backward exchange: enable != 0 starts recording an event pair around every side-stream reduction (at most 8 per read-out);
// the call returns, and clears, what was recorded since the previous call: milliseconds on the side stream and bytes per bucket, in issue order.
extern "C" int awt_comm_bucket_stats(awt_comm* m, int enable, double* ms, int64_t* bytes, int max, int* n_out) {
  AWT_REQUIRE(m && n_out, AWT_ERR_INVALID, "comm_bucket_stats: null argument");
  int n = 0;
  for (int i = 0; i < m->n_spans; ++i) {
    AWT_HIP_CHECK(hipEventSynchronize(m->spans[i].b));
    float t = 0.f;
    AWT_HIP_CHECK(hipEventElapsedTime(&t, m->spans[i].a, m->spans[i].b));
    if (ms && bytes && n < max) { ms[n] = t; bytes[n] = (int64_t)m->spans[i].bytes; ++n; }
  }
  m->n_spans = 0;
  m->timing = enable != 0;
  *n_out = n;
  return AWT_OK;
}
// everything enqueued on `consumer` after this call sees the side stream's completed reductions
int comm_join(awt_comm* m, hipStream_t consumer) {
  hipEvent_t ev = m->ev[sizeof(m->ev) / sizeof(m->ev[0]) - 1];
  AWT_HIP_CHECK(hipEventRecord(ev, m->side));
  AWT_HIP_CHECK(hipStreamWaitEvent(consumer, ev, 0));
  return AWT_OK;
}
