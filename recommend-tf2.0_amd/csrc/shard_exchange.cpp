// C2 — the row-sharded lookup's exchange behind the C ABI (include/recamd.h, rec_comm_* / rec_shard_*).
//
// The reference has no sharded lookup (its only distribution is tf.distribute.MirroredStrategy, e.g.
// src/ctr/fm/train.py:43-45); this is the north-star extension: tables row-sharded cyclically over the G GPUs of a
// node, one process per GPU, RCCL over xGMI.  One lookup =
//     plan_ids      device: exact de-duplication + stable bucketing by owner; the G send counts are all-gathered and
//                   copied to pinned host memory behind an event (issued one step ahead by a pipelined caller, so the
//                   host never waits for it)
//     plan_finish   host: wait for that event, derive the all-to-all(v) split sizes
//     exchange_ids  all-to-all #1: int32 local rows of the unique ids          (~4 B per unique lookup)
//     serve         owner-side gather from this rank's shard                   (the K1 kernel)
//     exchange_rows all-to-all #2: fp32 rows back                              (D*4 B per unique lookup)
// after which the consumer kernels read the returned rows through `uidx` (the fused gather + pairwise-dot kernel with
// the receive buffer as its table: no un-permute pass).  The same plan carries the backward: gradients of the
// unique rows travel owner-wards by the reverse all-to-all (exchange_rows with reverse = 1) and are scatter-added
// by the owner; dense-parameter gradients merge with rec_comm_allreduce_sum_f32 (MirroredStrategy's all-reduce).
//
// Transport: RCCL, resolved at run time with dlopen (librecamd.so itself does not link RCCL, and a process that has
// torch's RCCL loaded shares that copy); grouped ncclSend/ncclRecv form the all-to-all(v) — on the xGMI full mesh
// every peer pair has its own link.  A caller may also inject a transport (rec_comm_create_with_transport): the
// in-process test transport (rec_comm_create_local) runs G simulated ranks on one GPU with the real kernels.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace rec {
int shard_plan_device(const int32_t* vids, int64_t n, int32_t G, int32_t me, int32_t* rep, const int32_t* cache_slot,
                      int32_t* hot_count, int32_t cache_base, int32_t recv_base, uint64_t* stat, int32_t* first,
                      int32_t* uniq, int32_t* perm, int32_t* uidx, int32_t* send_local, int32_t* counts, void* ws_hist,
                      hipStream_t st);
int shard_vec_add(float* a, const float* b, int64_t n, hipStream_t st);
}  // namespace rec

using namespace rec;

// ---- RCCL, by name -----------------------------------------------------------------------------------
namespace {
typedef struct ncclComm* ncclComm_t;
struct NcclUid {
  char b[128];
};
struct Rccl {
  int (*GetUniqueId)(NcclUid*);
  int (*CommInitRank)(ncclComm_t*, int, NcclUid, int);
  int (*CommDestroy)(ncclComm_t);
  const char* (*GetErrorString)(int);
  int (*GroupStart)();
  int (*GroupEnd)();
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  bool ok = false;
  char err[256] = "";
};
constexpr int kNcclInt8 = 0, kNcclInt32 = 2, kNcclFloat32 = 7, kNcclSum = 0;

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      snprintf(r.err, sizeof(r.err), "cannot load librccl.so: %s", dlerror());
      return;
    }
#define REC_SYM(field, sym)                                                    \
  *reinterpret_cast<void**>(&r.field) = dlsym(h, sym);                         \
  if (!r.field) {                                                              \
    snprintf(r.err, sizeof(r.err), "librccl.so has no symbol %s", sym);        \
    return;                                                                    \
  }
    REC_SYM(GetUniqueId, "ncclGetUniqueId")
    REC_SYM(CommInitRank, "ncclCommInitRank")
    REC_SYM(CommDestroy, "ncclCommDestroy")
    REC_SYM(GetErrorString, "ncclGetErrorString")
    REC_SYM(GroupStart, "ncclGroupStart")
    REC_SYM(GroupEnd, "ncclGroupEnd")
    REC_SYM(Send, "ncclSend")
    REC_SYM(Recv, "ncclRecv")
    REC_SYM(AllGather, "ncclAllGather")
    REC_SYM(AllReduce, "ncclAllReduce")
#undef REC_SYM
    r.ok = true;
  });
  return r;
}

#define REC_NCCL(call, who)                                                             \
  do {                                                                                  \
    int rc__ = (call);                                                                  \
    if (rc__ != 0) {                                                                    \
      set_error("%s: RCCL error %d: %s", who, rc__, rccl().GetErrorString(rc__));      \
      return REC_EHIP;                                                                  \
    }                                                                                   \
  } while (0)

struct RcclCtx {
  ncclComm_t comm = nullptr;
  bool owned = false;
  int world = 1, rank = 0;
};

int rccl_allgather_counts(void* ctx, const int32_t* counts, int32_t* matrix, void* stream) {
  RcclCtx* c = static_cast<RcclCtx*>(ctx);
  REC_NCCL(rccl().AllGather(counts, matrix, (size_t)c->world, kNcclInt32, c->comm, (hipStream_t)stream),
           "rec_shard_plan_ids");
  return REC_OK;
}

int rccl_alltoallv(void* ctx, int32_t rank, const void* send, const int64_t* sc, const int64_t* sd, void* recv,
                   const int64_t* rc, const int64_t* rd, int32_t elem, void* stream) {
  RcclCtx* c = static_cast<RcclCtx*>(ctx);
  (void)rank;
  // bytes as ncclInt8 would cap a message at 2^31 elements only on very old RCCL; element counts here are
  // rows * D * 4 <= a few hundred MB.  fp32 / int32 payloads travel as 4-byte words.
  const int dt = (elem % 4 == 0) ? kNcclInt32 : kNcclInt8;
  const int64_t per = (elem % 4 == 0) ? elem / 4 : elem;
  hipStream_t st = (hipStream_t)stream;
  REC_NCCL(rccl().GroupStart(), "rec_shard_exchange");
  for (int p = 0; p < c->world; ++p) {
    if (sc[p] > 0)
      REC_NCCL(rccl().Send(static_cast<const char*>(send) + sd[p] * elem, (size_t)(sc[p] * per), dt, p, c->comm, st),
               "rec_shard_exchange");
    if (rc[p] > 0)
      REC_NCCL(rccl().Recv(static_cast<char*>(recv) + rd[p] * elem, (size_t)(rc[p] * per), dt, p, c->comm, st),
               "rec_shard_exchange");
  }
  REC_NCCL(rccl().GroupEnd(), "rec_shard_exchange");
  return REC_OK;
}

int rccl_allreduce(void* ctx, int32_t rank, float* buf, int64_t n, void* stream) {
  RcclCtx* c = static_cast<RcclCtx*>(ctx);
  (void)rank;
  REC_NCCL(rccl().AllReduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream),
           "rec_comm_allreduce_sum_f32");
  return REC_OK;
}

// ---- in-process transport: G simulated ranks on one device and one stream (tests) -------------------------
// Collectives are deferred: every rank registers its buffers, the LAST rank to arrive enqueues all the copies.  The
// caller must therefore run each phase for every rank before starting the next phase (tests/test_shard_cabi_gpu.py).
struct LocalOp {
  const void* send = nullptr;
  void* recv = nullptr;
  std::vector<int64_t> sc, sd, rc, rd;
  int32_t elem = 0;
  float* buf = nullptr;
  int64_t n = 0;
  const int32_t* counts = nullptr;
  int32_t* matrix = nullptr;
  bool set = false;
};
struct LocalGroup {
  int world = 0;
  int refs = 0;
  std::vector<LocalOp> ops;
  int arrived = 0;
};
struct LocalCtx {
  LocalGroup* g = nullptr;
  int rank = 0;
};

int local_arrive(LocalGroup* g, int rank, const char* who) {
  if (g->ops[rank].set) {
    set_error("%s (local transport): rank %d entered a collective twice before the other ranks entered it", who, rank);
    return REC_EINVAL;
  }
  g->ops[rank].set = true;
  g->arrived++;
  return REC_OK;
}
void local_reset(LocalGroup* g) {
  for (auto& o : g->ops) o = LocalOp();
  g->arrived = 0;
}

int local_allgather_counts(void* ctx, const int32_t* counts, int32_t* matrix, void* stream) {
  LocalCtx* c = static_cast<LocalCtx*>(ctx);
  LocalGroup* g = c->g;
  int rc = local_arrive(g, c->rank, "rec_shard_plan_ids");
  if (rc != REC_OK) return rc;
  g->ops[c->rank].counts = counts;
  g->ops[c->rank].matrix = matrix;
  if (g->arrived < g->world) return REC_OK;
  for (int dst = 0; dst < g->world; ++dst)
    for (int src = 0; src < g->world; ++src) {
      hipError_t e = hipMemcpyAsync(g->ops[dst].matrix + (int64_t)src * g->world, g->ops[src].counts,
                                    sizeof(int32_t) * g->world, hipMemcpyDeviceToDevice, (hipStream_t)stream);
      REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "local transport: memcpy: %s", hipGetErrorString(e));
    }
  local_reset(g);
  return REC_OK;
}

int local_alltoallv(void* ctx, int32_t rank, const void* send, const int64_t* sc, const int64_t* sd, void* recv,
                    const int64_t* rc, const int64_t* rd, int32_t elem, void* stream) {
  LocalCtx* c = static_cast<LocalCtx*>(ctx);
  LocalGroup* g = c->g;
  int rcode = local_arrive(g, rank, "rec_shard_exchange");
  if (rcode != REC_OK) return rcode;
  LocalOp& o = g->ops[rank];
  o.send = send;
  o.recv = recv;
  o.elem = elem;
  o.sc.assign(sc, sc + g->world);
  o.sd.assign(sd, sd + g->world);
  o.rc.assign(rc, rc + g->world);
  o.rd.assign(rd, rd + g->world);
  if (g->arrived < g->world) return REC_OK;
  for (int src = 0; src < g->world; ++src)
    for (int dst = 0; dst < g->world; ++dst) {
      const LocalOp& s = g->ops[src];
      const LocalOp& d = g->ops[dst];
      if (s.sc[dst] != d.rc[src] || s.elem != d.elem) {
        set_error("local transport: rank %d sends %lld x %d B to rank %d, which expects %lld x %d B", src,
                  (long long)s.sc[dst], s.elem, dst, (long long)d.rc[src], d.elem);
        local_reset(g);
        return REC_ESHAPE;
      }
      if (s.sc[dst] == 0) continue;
      hipError_t e = hipMemcpyAsync(static_cast<char*>(d.recv) + d.rd[src] * d.elem,
                                    static_cast<const char*>(s.send) + s.sd[dst] * s.elem, (size_t)(s.sc[dst] * s.elem),
                                    hipMemcpyDeviceToDevice, (hipStream_t)stream);
      REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "local transport: memcpy: %s", hipGetErrorString(e));
    }
  local_reset(g);
  return REC_OK;
}

int local_allreduce(void* ctx, int32_t rank, float* buf, int64_t n, void* stream) {
  LocalCtx* c = static_cast<LocalCtx*>(ctx);
  LocalGroup* g = c->g;
  int rc = local_arrive(g, rank, "rec_comm_allreduce_sum_f32");
  if (rc != REC_OK) return rc;
  g->ops[rank].buf = buf;
  g->ops[rank].n = n;
  if (g->arrived < g->world) return REC_OK;
  hipStream_t st = (hipStream_t)stream;
  for (int r = 1; r < g->world; ++r) {  // fixed order: rank 0 + rank 1 + ... (deterministic)
    if (g->ops[r].n != g->ops[0].n) {
      set_error("local transport: all-reduce sizes differ between ranks");
      local_reset(g);
      return REC_ESHAPE;
    }
    rc = shard_vec_add(g->ops[0].buf, g->ops[r].buf, n, st);
    if (rc != REC_OK) return rc;
  }
  for (int r = 1; r < g->world; ++r) {
    hipError_t e = hipMemcpyAsync(g->ops[r].buf, g->ops[0].buf, sizeof(float) * n, hipMemcpyDeviceToDevice, st);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "local transport: memcpy: %s", hipGetErrorString(e));
  }
  local_reset(g);
  return REC_OK;
}
}  // namespace

// ---- rec_comm ------------------------------------------------------------------------------------------
struct rec_comm {
  rec_transport t{};
  int world = 1, rank = 0;
  RcclCtx* rccl_ctx = nullptr;
  LocalCtx* local_ctx = nullptr;
};

extern "C" int rec_comm_unique_id(void* id128) {
  REC_CHECK_ARG(id128, REC_EINVAL, "rec_comm_unique_id: NULL");
  REC_CHECK_ARG(rccl().ok, REC_EHIP, "rec_comm_unique_id: %s", rccl().err);
  NcclUid u;
  REC_NCCL(rccl().GetUniqueId(&u), "rec_comm_unique_id");
  memcpy(id128, u.b, 128);
  return REC_OK;
}

static rec_comm* make_rccl_comm(ncclComm_t nc, bool owned, int world, int rank) {
  rec_comm* c = new rec_comm();
  c->world = world;
  c->rank = rank;
  c->rccl_ctx = new RcclCtx();
  c->rccl_ctx->comm = nc;
  c->rccl_ctx->owned = owned;
  c->rccl_ctx->world = world;
  c->rccl_ctx->rank = rank;
  c->t.ctx = c->rccl_ctx;
  c->t.allgather_counts = rccl_allgather_counts;
  c->t.alltoallv = rccl_alltoallv;
  c->t.allreduce_sum_f32 = rccl_allreduce;
  c->t.deferred = 0;
  return c;
}

extern "C" int rec_comm_init_rank(rec_comm** out, const void* id128, int32_t world, int32_t rank) {
  const char* who = "rec_comm_init_rank";
  REC_CHECK_ARG(out && id128, REC_EINVAL, "%s: NULL", who);
  REC_CHECK_ARG(world >= 1 && world <= 64 && rank >= 0 && rank < world, REC_ESHAPE, "%s: world=%d rank=%d", who, world,
                rank);
  REC_CHECK_ARG(rccl().ok, REC_EHIP, "%s: %s", who, rccl().err);
  NcclUid u;
  memcpy(u.b, id128, 128);
  ncclComm_t nc = nullptr;
  REC_NCCL(rccl().CommInitRank(&nc, world, u, rank), who);
  *out = make_rccl_comm(nc, true, world, rank);
  return REC_OK;
}

extern "C" int rec_comm_from_nccl(rec_comm** out, void* nccl_comm, int32_t world, int32_t rank) {
  const char* who = "rec_comm_from_nccl";
  REC_CHECK_ARG(out && nccl_comm, REC_EINVAL, "%s: NULL", who);
  REC_CHECK_ARG(world >= 1 && world <= 64 && rank >= 0 && rank < world, REC_ESHAPE, "%s: world=%d rank=%d", who, world,
                rank);
  REC_CHECK_ARG(rccl().ok, REC_EHIP, "%s: %s", who, rccl().err);
  *out = make_rccl_comm(static_cast<ncclComm_t>(nccl_comm), false, world, rank);
  return REC_OK;
}

extern "C" int rec_comm_create_with_transport(rec_comm** out, const rec_transport* t, int32_t world, int32_t rank) {
  const char* who = "rec_comm_create_with_transport";
  REC_CHECK_ARG(out && t && t->allgather_counts && t->alltoallv, REC_EINVAL, "%s: NULL transport entry", who);
  REC_CHECK_ARG(world >= 1 && world <= 64 && rank >= 0 && rank < world, REC_ESHAPE, "%s: world=%d rank=%d", who, world,
                rank);
  rec_comm* c = new rec_comm();
  c->t = *t;
  c->world = world;
  c->rank = rank;
  *out = c;
  return REC_OK;
}

extern "C" int rec_comm_create_local(int32_t world, rec_comm** comms_out) {
  const char* who = "rec_comm_create_local";
  REC_CHECK_ARG(comms_out && world >= 1 && world <= 64, REC_EINVAL, "%s: world=%d", who, world);
  LocalGroup* g = new LocalGroup();
  g->world = world;
  g->refs = world;
  g->ops.resize(world);
  for (int r = 0; r < world; ++r) {
    rec_comm* c = new rec_comm();
    c->world = world;
    c->rank = r;
    c->local_ctx = new LocalCtx();
    c->local_ctx->g = g;
    c->local_ctx->rank = r;
    c->t.ctx = c->local_ctx;
    c->t.allgather_counts = local_allgather_counts;
    c->t.alltoallv = local_alltoallv;
    c->t.allreduce_sum_f32 = local_allreduce;
    c->t.deferred = 1;
    comms_out[r] = c;
  }
  return REC_OK;
}

extern "C" int rec_comm_destroy(rec_comm* c) {
  if (!c) return REC_OK;
  if (c->rccl_ctx) {
    if (c->rccl_ctx->owned && c->rccl_ctx->comm) (void)rccl().CommDestroy(c->rccl_ctx->comm);
    delete c->rccl_ctx;
  }
  if (c->local_ctx) {
    if (--c->local_ctx->g->refs == 0) delete c->local_ctx->g;
    delete c->local_ctx;
  }
  delete c;
  return REC_OK;
}

extern "C" int32_t rec_comm_world(const rec_comm* c) { return c ? c->world : 0; }
extern "C" const char* rec_comm_transport_name(const rec_comm* c) {
  if (!c) return "none";
  if (c->rccl_ctx) return c->rccl_ctx->owned ? "rccl" : "rccl (borrowed ncclComm_t)";
  if (c->local_ctx) return "in-process";
  return "caller-supplied";
}
extern "C" int32_t rec_comm_rank(const rec_comm* c) { return c ? c->rank : -1; }

extern "C" int rec_comm_allreduce_sum_f32(rec_comm* c, float* buf, int64_t n, void* stream) {
  const char* who = "rec_comm_allreduce_sum_f32";
  REC_CHECK_ARG(c && (buf || n == 0) && n >= 0, REC_EINVAL, "%s: bad arguments", who);
  if (n == 0) return REC_OK;
  if (c->world == 1 && !c->t.deferred) return REC_OK;  // a single replica: the sum is the value
  REC_CHECK_ARG(c->t.allreduce_sum_f32, REC_ENOTIMPL, "%s: transport has no all-reduce", who);
  return c->t.allreduce_sum_f32(c->t.ctx, c->rank, buf, n, stream);
}

// ---- plan ----------------------------------------------------------------------------------------------
struct rec_shard_plan {
  rec_comm* comm = nullptr;
  int64_t max_ids = 0;
  int32_t* h_matrix = nullptr;  // pinned (G x G): row p = rank p's send counts
  hipEvent_t ev = nullptr;
  // state of the current lookup
  int64_t n = 0;
  char* ws = nullptr;
  bool planned = false, finished = false;
  hipStream_t plan_stream = nullptr;
  std::vector<int64_t> sc, sd, rc, rd;  // in rows
  int64_t n_unique = 0, n_recv = 0;
};

namespace {
struct WsLayout {
  int64_t counts, matrix, hist, first, uniq, perm, uidx, send_local, total;
};
WsLayout ws_layout(int64_t max_ids, int G) {
  auto al = [](int64_t x) { return (x + 255) / 256 * 256; };
  WsLayout L{};
  int64_t o = 0;
  L.counts = o, o += al(4 * G);
  L.matrix = o, o += al(4 * (int64_t)G * G);
  L.hist = o, o += al(rec_shard_bucket_workspace_bytes(max_ids, G));
  L.first = o, o += al(4 * max_ids);
  L.uniq = o, o += al(4 * max_ids);
  L.perm = o, o += al(4 * max_ids);
  L.uidx = o, o += al(4 * max_ids);
  L.send_local = o, o += al(4 * max_ids);
  L.total = o;
  return L;
}
}  // namespace

extern "C" int64_t rec_shard_plan_workspace_bytes(int64_t max_ids, int32_t world) {
  if (max_ids < 0 || world < 1 || world > 64) return 0;
  return ws_layout(max_ids, world).total;
}

extern "C" int rec_shard_plan_create(rec_comm* comm, int64_t max_ids, rec_shard_plan** out) {
  const char* who = "rec_shard_plan_create";
  REC_CHECK_ARG(comm && out && max_ids >= 0 && max_ids <= 0x7fffffffLL, REC_EINVAL, "%s: bad arguments", who);
  rec_shard_plan* p = new rec_shard_plan();
  p->comm = comm;
  p->max_ids = max_ids;
  const int G = comm->world;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p->h_matrix), sizeof(int32_t) * G * G, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error("%s: %s", who, hipGetErrorString(e));
    if (p->h_matrix) (void)hipHostFree(p->h_matrix);
    delete p;
    return REC_EHIP;
  }
  p->sc.resize(G), p->sd.resize(G), p->rc.resize(G), p->rd.resize(G);
  *out = p;
  return REC_OK;
}

extern "C" int rec_shard_plan_destroy(rec_shard_plan* p) {
  if (!p) return REC_OK;
  if (p->ev) (void)hipEventDestroy(p->ev);
  if (p->h_matrix) (void)hipHostFree(p->h_matrix);
  delete p;
  return REC_OK;
}

extern "C" int rec_shard_plan_ids_ex(rec_shard_plan* p, const int32_t* vids, int64_t n, int32_t* rep_table,
                                     const rec_shard_resolve_opts* opts, void* workspace, void* stream) {
  const char* who = "rec_shard_plan_ids";
  REC_CHECK_ARG(p && workspace && (vids || n == 0), REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(n >= 0 && n <= p->max_ids, REC_ESHAPE, "%s: n=%lld exceeds the plan's max_ids=%lld", who, (long long)n,
                (long long)p->max_ids);
  const int G = p->comm->world;
  const WsLayout L = ws_layout(p->max_ids, G);
  char* ws = static_cast<char*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  auto I = [&](int64_t off) { return reinterpret_cast<int32_t*>(ws + off); };
  rec_shard_resolve_opts o{};
  if (opts) o = *opts;
  REC_CHECK_ARG(o.cache_base >= 0 && o.recv_base >= 0, REC_ESHAPE, "%s: negative row base", who);
  int rc = shard_plan_device(vids, n, G, o.bypass_local ? p->comm->rank : -1, rep_table, o.cache_slot, o.hot_count,
                             o.cache_base, o.recv_base, o.stat, I(L.first), I(L.uniq), I(L.perm), I(L.uidx),
                             I(L.send_local), I(L.counts), ws + L.hist, st);
  if (rc != REC_OK) return rc;
  rc = p->comm->t.allgather_counts(p->comm->t.ctx, I(L.counts), I(L.matrix), stream);
  if (rc != REC_OK) return rc;
  p->n = n;
  p->ws = ws;
  p->plan_stream = st;
  p->planned = true;
  p->finished = false;
  if (!p->comm->t.deferred) {
    hipError_t e = hipMemcpyAsync(p->h_matrix, I(L.matrix), sizeof(int32_t) * G * G, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(p->ev, st);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: %s", who, hipGetErrorString(e));
  }
  return REC_OK;
}

extern "C" int rec_shard_plan_ids(rec_shard_plan* p, const int32_t* vids, int64_t n, int32_t* rep_table, void* workspace,
                                  void* stream) {
  return rec_shard_plan_ids_ex(p, vids, n, rep_table, nullptr, workspace, stream);
}

extern "C" int rec_shard_plan_finish(rec_shard_plan* p, int64_t* n_unique, int64_t* n_recv) {
  const char* who = "rec_shard_plan_finish";
  REC_CHECK_ARG(p && p->planned, REC_EINVAL, "%s: no rec_shard_plan_ids call to finish", who);
  const int G = p->comm->world, me = p->comm->rank;
  if (!p->finished) {
    hipError_t e;
    if (p->comm->t.deferred) {  // in-process transport: the gather was enqueued by the last rank to arrive
      const WsLayout L = ws_layout(p->max_ids, G);
      e = hipMemcpyAsync(p->h_matrix, p->ws + L.matrix, sizeof(int32_t) * G * G, hipMemcpyDeviceToHost, p->plan_stream);
      if (e == hipSuccess) e = hipStreamSynchronize(p->plan_stream);
    } else {
      e = hipEventSynchronize(p->ev);
    }
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: %s", who, hipGetErrorString(e));
    int64_t so = 0, ro = 0;
    for (int q = 0; q < G; ++q) {
      p->sc[q] = p->h_matrix[(int64_t)me * G + q];
      p->rc[q] = p->h_matrix[(int64_t)q * G + me];
      REC_CHECK_ARG(p->sc[q] >= 0 && p->rc[q] >= 0, REC_EHIP, "%s: negative count from the exchange", who);
      p->sd[q] = so, so += p->sc[q];
      p->rd[q] = ro, ro += p->rc[q];
    }
    p->n_unique = so;
    p->n_recv = ro;
    p->finished = true;
  }
  if (n_unique) *n_unique = p->n_unique;
  if (n_recv) *n_recv = p->n_recv;
  return REC_OK;
}

extern "C" const int32_t* rec_shard_plan_uidx(const rec_shard_plan* p) {
  if (!p || !p->planned) return nullptr;
  return reinterpret_cast<const int32_t*>(p->ws + ws_layout(p->max_ids, p->comm->world).uidx);
}

extern "C" int rec_shard_exchange_ids(rec_shard_plan* p, int32_t* recv_local, void* stream) {
  const char* who = "rec_shard_exchange_ids";
  REC_CHECK_ARG(p && p->finished, REC_EINVAL, "%s: call rec_shard_plan_finish first", who);
  REC_CHECK_ARG(recv_local || p->n_recv == 0, REC_EINVAL, "%s: NULL recv_local", who);
  const WsLayout L = ws_layout(p->max_ids, p->comm->world);
  return p->comm->t.alltoallv(p->comm->t.ctx, p->comm->rank, p->ws + L.send_local, p->sc.data(), p->sd.data(), recv_local,
                              p->rc.data(), p->rd.data(), 4, stream);
}

extern "C" int rec_shard_serve_f32(rec_shard_plan* p, const float* arena, int64_t arena_rows, int32_t D,
                                   const int32_t* recv_local, float* served, int32_t* oob_flag, void* stream) {
  const char* who = "rec_shard_serve_f32";
  REC_CHECK_ARG(p && p->finished, REC_EINVAL, "%s: call rec_shard_plan_finish first", who);
  if (p->n_recv == 0) return REC_OK;
  REC_CHECK_ARG(arena && recv_local && served, REC_EINVAL, "%s: NULL pointer", who);
  rec_table_desc d{arena, arena_rows, D, 0};
  return rec_gather_concat_f32(&d, 1, recv_local, REC_IDS_I32, 1, p->n_recv, served, D, oob_flag, stream);
}

extern "C" int rec_shard_exchange_rows_f32(rec_shard_plan* p, const float* src, int32_t D, float* dst, int32_t reverse,
                                           void* stream) {
  const char* who = "rec_shard_exchange_rows_f32";
  REC_CHECK_ARG(p && p->finished, REC_EINVAL, "%s: call rec_shard_plan_finish first", who);
  REC_CHECK_ARG(D >= 1, REC_ESHAPE, "%s: D=%d", who, D);
  // forward: owners return the served rows (their recv layout) to the requesters (their send layout);
  // reverse: requesters send one gradient row per unique lookup back to the owners
  const std::vector<int64_t>&sc = reverse ? p->sc : p->rc, &sd = reverse ? p->sd : p->rd;
  const std::vector<int64_t>&rc = reverse ? p->rc : p->sc, &rd = reverse ? p->rd : p->sd;
  return p->comm->t.alltoallv(p->comm->t.ctx, p->comm->rank, src, sc.data(), sd.data(), dst, rc.data(), rd.data(), D * 4,
                              stream);
}

extern "C" int rec_shard_lookup_f32(rec_shard_plan* p, const float* arena, int64_t arena_rows, int32_t D,
                                    int32_t* recv_local, int64_t recv_cap, float* served, float* rows_out,
                                    int64_t rows_cap, int32_t* oob_flag, void* stream) {
  const char* who = "rec_shard_lookup_f32";
  int64_t nu = 0, nr = 0;
  int rc = rec_shard_plan_finish(p, &nu, &nr);
  if (rc != REC_OK) return rc;
  REC_CHECK_ARG(nr <= recv_cap && nu <= rows_cap, REC_ESHAPE,
                "%s: buffers too small: %lld rows to serve (capacity %lld), %lld unique rows to receive (capacity %lld)",
                who, (long long)nr, (long long)recv_cap, (long long)nu, (long long)rows_cap);
  rc = rec_shard_exchange_ids(p, recv_local, stream);
  if (rc != REC_OK) return rc;
  rc = rec_shard_serve_f32(p, arena, arena_rows, D, recv_local, served, oob_flag, stream);
  if (rc != REC_OK) return rc;
  return rec_shard_exchange_rows_f32(p, served, D, rows_out, 0, stream);
}
