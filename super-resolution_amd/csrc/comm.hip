// comm.hip -- the exchanges of the sharded MAP evaluation / solve (SURVEY.md
// section 8e), inside the library and on the evaluation's HIP stream:
//   all-reduce of device buffers (gradient, CG scalars), and neighbour
//   send/receive of strided row blocks / channel planes (halos of x).
// Two backends behind one small interface:
//   * RCCL over xGMI (production): ncclAllReduce / grouped ncclSend + ncclRecv.
//     librccl is dlopen'ed on first use, so libsrmap.so loads (and every
//     single-GPU entry point works) on hosts without it, and a process that
//     already carries an RCCL (PyTorch ships its own copy under the same
//     soname) shares that one instead of loading a second.
//   * host callbacks (MPI / gloo harnesses, one-GPU tests): device buffers are
//     staged through pinned host memory around the caller's functions.
// The reference has no counterpart: it is a single-process program.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include <mutex>

#include "comm.hpp"

namespace srmap {

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;                       // optional
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;     // optional (RCCL >= 2.18)
  ncclResult_t (*GetVersion)(int*) = nullptr;                                          // optional
};

RcclApi* rccl_api(srmap_ctx* ctx) {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {  // one thread loads the library; every later caller sees the finished table
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      bool ok = true;
      auto sym = [&](const char* name) { void* s = dlsym(api.lib, name); if (!s) ok = false; return s; };
      api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
      api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
      api.Send = (decltype(api.Send))sym("ncclSend");
      api.Recv = (decltype(api.Recv))sym("ncclRecv");
      api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
      api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
      api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
      api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
      api.CommSplit = (decltype(api.CommSplit))dlsym(api.lib, "ncclCommSplit");
      api.GetVersion = (decltype(api.GetVersion))dlsym(api.lib, "ncclGetVersion");
      if (!ok) { dlclose(api.lib); api.lib = nullptr; }
    }
  });
  if (!api.lib) { set_error(ctx, SRMAP_EUNSUPPORTED, "librccl.so.1 could not be loaded"); return nullptr; }
  return &api;
}

#define SRMAP_NCCL(c, call)                                                                             \
  do {                                                                                                  \
    ncclResult_t r_ = (call);                                                                           \
    if (r_ != ncclSuccess)                                                                              \
      return set_error((c)->ctx, SRMAP_EHIP, "%s failed: %s", #call, (c)->api->GetErrorString(r_));     \
  } while (0)

}  // namespace

}  // namespace srmap

using namespace srmap;

struct srmap_comm {
  srmap_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  int kind = 0;  // 0 = host callbacks, 1 = RCCL
  RcclApi* api = nullptr;
  ncclComm_t nccl = nullptr;
  srmap_host_allreduce_fn ar = nullptr;
  srmap_host_sendrecv_fn sr = nullptr;
  void* user = nullptr;
  void* h_send = nullptr; void* h_recv = nullptr; size_t h_cap = 0;  // pinned staging (host backend)
  // row shards: the halo exchange runs on this side stream, under the tiles that read no halo row (solver.hip)
  hipStream_t side = nullptr;
  hipEvent_t ev_x = nullptr, ev_halo = nullptr;
  int overlap = -1;  // row shards: halo exchange under the interior tile rows; -1 = backend default (srmap_comm_set_overlap)
};

namespace srmap {

int comm_side(srmap_comm* c, hipStream_t* side, hipEvent_t* ev_x, hipEvent_t* ev_halo) {
  if (!c->side) {
    SRMAP_HIP(c->ctx, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    SRMAP_HIP(c->ctx, hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming));
    SRMAP_HIP(c->ctx, hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  }
  *side = c->side; *ev_x = c->ev_x; *ev_halo = c->ev_halo;
  return SRMAP_OK;
}

// Overlap of the row-shard halo exchange with the interior tile rows: on by default for the host-callback backend (the
// hook blocks there, so the two-phase launch is exercised without any concurrency), OFF by default for RCCL until it has
// been seen running on two GPUs (the same ncclComm_t is then used from two streams) -- srmap_comm_set_overlap opts in.
bool comm_overlap(const srmap_comm* c) { return c && (c->overlap < 0 ? c->kind == 0 : c->overlap != 0); }
int comm_rank(const srmap_comm* c) { return c ? c->rank : 0; }
int comm_world(const srmap_comm* c) { return c ? c->world : 1; }

static int ensure_host(srmap_comm* c, size_t bytes) {
  if (c->h_cap >= bytes) return SRMAP_OK;
  if (c->h_send) (void)hipHostFree(c->h_send);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  c->h_send = c->h_recv = nullptr; c->h_cap = 0;
  const size_t cap = bytes + (bytes >> 2) + 4096;
  SRMAP_HIP(c->ctx, hipHostMalloc(&c->h_send, cap, hipHostMallocDefault));
  SRMAP_HIP(c->ctx, hipHostMalloc(&c->h_recv, cap, hipHostMallocDefault));
  c->h_cap = cap;
  return SRMAP_OK;
}

// In-place all-reduce of `count` elements of a device buffer; op 0 = sum, 1 = max.
int comm_allreduce(srmap_comm* c, void* dev, size_t count, int dtype, int op, hipStream_t st) {
  if (!c || c->world <= 1 || count == 0) return SRMAP_OK;
  const size_t esz = dtype == SRMAP_F32 ? 4 : 8;
  if (c->kind == 1) {
    SRMAP_NCCL(c, c->api->AllReduce(dev, dev, count, dtype == SRMAP_F32 ? ncclFloat32 : ncclFloat64,
                                    op == 1 ? ncclMax : ncclSum, c->nccl, st));
    return SRMAP_OK;
  }
  int rc = ensure_host(c, count * esz);
  if (rc) return rc;
  SRMAP_HIP(c->ctx, hipMemcpyAsync(c->h_send, dev, count * esz, hipMemcpyDeviceToHost, st));
  SRMAP_HIP(c->ctx, hipStreamSynchronize(st));
  if (c->ar(c->h_send, count, dtype, op, c->user) != 0)
    return set_error(c->ctx, SRMAP_EHIP, "host all-reduce callback failed");
  SRMAP_HIP(c->ctx, hipMemcpyAsync(dev, c->h_send, count * esz, hipMemcpyHostToDevice, st));
  SRMAP_HIP(c->ctx, hipStreamSynchronize(st));  // the pinned buffer is reused by the next call
  return SRMAP_OK;
}

// Neighbour exchange of `nseg` equally sized segments (one per channel): send[i] (seg elements each) to rank
// dst, recv[i] from rank src; either side may be absent (rank < 0).  Both directions of a halo exchange are issued
// by every rank in the same order, so the RCCL group / the caller's sendrecv cannot deadlock.
int comm_exchange(srmap_comm* c, const void* const* send, int dst, void* const* recv, int src, int nseg,
                  size_t send_seg, size_t recv_seg, int dtype, hipStream_t st) {
  if (!c || c->world <= 1 || nseg == 0) return SRMAP_OK;
  if (send_seg == 0) dst = -1;
  if (recv_seg == 0) src = -1;
  if (dst < 0 && src < 0) return SRMAP_OK;
  const size_t esz = dtype == SRMAP_F32 ? 4 : 8;
  if (c->kind == 1) {
    const ncclDataType_t t = dtype == SRMAP_F32 ? ncclFloat32 : ncclFloat64;
    SRMAP_NCCL(c, c->api->GroupStart());
    ncclResult_t first = ncclSuccess;  // the group is ALWAYS closed: a dangling group swallows every later collective
    for (int i = 0; i < nseg && first == ncclSuccess; ++i) {
      if (dst >= 0) first = c->api->Send(send[i], send_seg, t, dst, c->nccl, st);
      if (src >= 0 && first == ncclSuccess) first = c->api->Recv(recv[i], recv_seg, t, src, c->nccl, st);
    }
    const ncclResult_t end = c->api->GroupEnd();
    if (first != ncclSuccess) return set_error(c->ctx, SRMAP_EHIP, "ncclSend / ncclRecv failed: %s", c->api->GetErrorString(first));
    if (end != ncclSuccess) return set_error(c->ctx, SRMAP_EHIP, "ncclGroupEnd failed: %s", c->api->GetErrorString(end));
    return SRMAP_OK;
  }
  const size_t sbytes = (size_t)nseg * send_seg * esz, rbytes = (size_t)nseg * recv_seg * esz;
  int rc = ensure_host(c, sbytes > rbytes ? sbytes : rbytes);
  if (rc) return rc;
  if (dst >= 0)
    for (int i = 0; i < nseg; ++i)
      SRMAP_HIP(c->ctx, hipMemcpyAsync((char*)c->h_send + (size_t)i * send_seg * esz, send[i], send_seg * esz, hipMemcpyDeviceToHost, st));
  SRMAP_HIP(c->ctx, hipStreamSynchronize(st));
  if (c->sr(c->h_send, dst >= 0 ? sbytes : 0, dst, c->h_recv, src >= 0 ? rbytes : 0, src, c->user) != 0)
    return set_error(c->ctx, SRMAP_EHIP, "host send/receive callback failed");
  if (src >= 0) {
    for (int i = 0; i < nseg; ++i)
      SRMAP_HIP(c->ctx, hipMemcpyAsync(recv[i], (char*)c->h_recv + (size_t)i * recv_seg * esz, recv_seg * esz, hipMemcpyHostToDevice, st));
    SRMAP_HIP(c->ctx, hipStreamSynchronize(st));
  }
  return SRMAP_OK;
}

// Both directions of a halo exchange in ONE RCCL group (one launch on the stream instead of two): `a` travels towards
// rank `down` / arrives from `up`, `b` the other way.  Host backend: the two one-directional exchanges in turn.
int comm_exchange2(srmap_comm* c, const void* const* send_a, void* const* recv_a, size_t send_a_seg, size_t recv_a_seg,
                   const void* const* send_b, void* const* recv_b, size_t send_b_seg, size_t recv_b_seg, int up, int down,
                   int nseg, int dtype, hipStream_t st) {
  if (!c || c->world <= 1 || nseg == 0) return SRMAP_OK;
  if (c->kind != 1) {
    int rc = comm_exchange(c, send_a, down, recv_a, up, nseg, down >= 0 ? send_a_seg : 0, up >= 0 ? recv_a_seg : 0, dtype, st);
    if (rc) return rc;
    return comm_exchange(c, send_b, up, recv_b, down, nseg, up >= 0 ? send_b_seg : 0, down >= 0 ? recv_b_seg : 0, dtype, st);
  }
  const ncclDataType_t t = dtype == SRMAP_F32 ? ncclFloat32 : ncclFloat64;
  SRMAP_NCCL(c, c->api->GroupStart());
  ncclResult_t first = ncclSuccess;
  auto chk = [&](ncclResult_t r) { if (first == ncclSuccess) first = r; };
  for (int i = 0; i < nseg; ++i) {
    if (down >= 0 && send_a_seg) chk(c->api->Send(send_a[i], send_a_seg, t, down, c->nccl, st));
    if (up >= 0 && recv_a_seg) chk(c->api->Recv(recv_a[i], recv_a_seg, t, up, c->nccl, st));
    if (up >= 0 && send_b_seg) chk(c->api->Send(send_b[i], send_b_seg, t, up, c->nccl, st));
    if (down >= 0 && recv_b_seg) chk(c->api->Recv(recv_b[i], recv_b_seg, t, down, c->nccl, st));
  }
  const ncclResult_t end = c->api->GroupEnd();
  if (first != ncclSuccess) return set_error(c->ctx, SRMAP_EHIP, "ncclSend / ncclRecv failed: %s", c->api->GetErrorString(first));
  if (end != ncclSuccess) return set_error(c->ctx, SRMAP_EHIP, "ncclGroupEnd failed: %s", c->api->GetErrorString(end));
  return SRMAP_OK;
}

// Gradient (count elements of dtype) and cost (one double) summed over the ranks of `c` as ONE RCCL group.
int comm_allreduce_grad_cost(srmap_comm* c, void* g, size_t count, int dtype, double* cost, hipStream_t st) {
  if (!c || c->world <= 1) return SRMAP_OK;
  if (c->kind != 1) {
    if (g && count) { int rc = comm_allreduce(c, g, count, dtype, 0, st); if (rc) return rc; }
    return cost ? comm_allreduce(c, cost, 1, SRMAP_F64, 0, st) : SRMAP_OK;
  }
  SRMAP_NCCL(c, c->api->GroupStart());
  ncclResult_t first = ncclSuccess;
  if (g && count) first = c->api->AllReduce(g, g, count, dtype == SRMAP_F32 ? ncclFloat32 : ncclFloat64, ncclSum, c->nccl, st);
  if (cost && first == ncclSuccess) first = c->api->AllReduce(cost, cost, 1, ncclFloat64, ncclSum, c->nccl, st);
  const ncclResult_t end = c->api->GroupEnd();
  if (first != ncclSuccess) return set_error(c->ctx, SRMAP_EHIP, "ncclAllReduce failed: %s", c->api->GetErrorString(first));
  if (end != ncclSuccess) return set_error(c->ctx, SRMAP_EHIP, "ncclGroupEnd failed: %s", c->api->GetErrorString(end));
  return SRMAP_OK;
}

}  // namespace srmap

extern "C" {

int srmap_comm_get_unique_id(srmap_ctx* ctx, char* id128) {
  if (!ctx || !id128) return SRMAP_EINVAL;
  RcclApi* api = rccl_api(ctx);
  if (!api) return SRMAP_EUNSUPPORTED;
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  const ncclResult_t r = api->GetUniqueId(&id);
  if (r != ncclSuccess) return set_error(ctx, SRMAP_EHIP, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
  static_assert(sizeof(id.internal) == SRMAP_UNIQUE_ID_BYTES, "unique id size");
  std::memcpy(id128, id.internal, SRMAP_UNIQUE_ID_BYTES);
  return SRMAP_OK;
}

int srmap_comm_create_rccl(srmap_ctx* ctx, const char* id128, int rank, int world, srmap_comm** out) {
  if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return SRMAP_EINVAL;
  *out = nullptr;
  RcclApi* api = rccl_api(ctx);
  if (!api) return SRMAP_EUNSUPPORTED;
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  srmap_comm* c = new (std::nothrow) srmap_comm();
  if (!c) return SRMAP_ENOMEM;
  c->ctx = ctx; c->rank = rank; c->world = world; c->kind = 1; c->api = api;
  ncclUniqueId id;
  std::memcpy(id.internal, id128, SRMAP_UNIQUE_ID_BYTES);
  const ncclResult_t r = api->CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return set_error(ctx, SRMAP_EHIP, "ncclCommInitRank failed: %s", api->GetErrorString(r));
  }
  *out = c;
  return SRMAP_OK;
}

int srmap_comm_create_host(srmap_ctx* ctx, int rank, int world, srmap_host_allreduce_fn allreduce,
                           srmap_host_sendrecv_fn sendrecv, void* user, srmap_comm** out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return SRMAP_EINVAL;
  *out = nullptr;
  if (world > 1 && (!allreduce || !sendrecv)) return set_error(ctx, SRMAP_EINVAL, "host communicator needs both callbacks");
  srmap_comm* c = new (std::nothrow) srmap_comm();
  if (!c) return SRMAP_ENOMEM;
  c->ctx = ctx; c->rank = rank; c->world = world; c->kind = 0;
  c->ar = allreduce; c->sr = sendrecv; c->user = user;
  *out = c;
  return SRMAP_OK;
}

/* In-place all-reduce of a device buffer through the communicator (harness / diagnostics entry point: the solver
 * and the sharded evaluation call the same code internally).  Always goes through the backend, world 1 included. */
int srmap_comm_allreduce(srmap_comm* c, void* dev_buf, size_t count, int dtype, int op, void* hip_stream) {
  if (!c || !dev_buf) return SRMAP_EINVAL;
  SRMAP_HIP(c->ctx, hipSetDevice(c->ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->ctx->stream;
  if (c->kind == 1) {
    SRMAP_NCCL(c, c->api->AllReduce(dev_buf, dev_buf, count, dtype == SRMAP_F32 ? ncclFloat32 : ncclFloat64,
                                    op == 1 ? ncclMax : ncclSum, c->nccl, st));
    return SRMAP_OK;
  }
  return comm_allreduce(c, dev_buf, count, dtype, op, st);
}

/* What the communicator itself reports: its rank, its size (ncclCommCount for RCCL) and its backend (1 = RCCL, 0 = host
 * callbacks).  Any out pointer may be NULL. */
// Which collective library the communicator runs on: "rccl <ncclGetVersion> <file the symbols were resolved from>" or
// "host callbacks".  A process may carry more than one librccl (PyTorch ships its own): this names the one in use.
int srmap_comm_describe(srmap_comm* c, char* buf, size_t cap) {
  if (!c || !buf || cap == 0) return SRMAP_EINVAL;
  if (c->kind != 1) { snprintf(buf, cap, "host callbacks"); return SRMAP_OK; }
  int ver = 0;
  if (c->api->GetVersion) (void)c->api->GetVersion(&ver);
  Dl_info di;
  const char* path = (dladdr((void*)c->api->AllReduce, &di) && di.dli_fname) ? di.dli_fname : "?";
  snprintf(buf, cap, "rccl %d %s", ver, path);
  return SRMAP_OK;
}

int srmap_comm_set_overlap(srmap_comm* c, int on) {
  if (!c) return SRMAP_EINVAL;
  c->overlap = on ? 1 : 0;
  return SRMAP_OK;
}

int srmap_comm_info(srmap_comm* c, int* rank, int* world, int* backend) {
  if (!c) return SRMAP_EINVAL;
  int w = c->world;
  if (c->kind == 1 && c->api->CommCount) {
    int n = 0;
    SRMAP_NCCL(c, c->api->CommCount(c->nccl, &n));
    w = n;
  }
  if (rank) *rank = c->rank;
  if (world) *world = w;
  if (backend) *backend = c->kind;
  return SRMAP_OK;
}

/* ncclCommSplit: the ranks that pass the same `color` form a new communicator, ordered by `key` (RCCL backend only;
 * host-callback harnesses create the sub-communicator themselves).  Collective over `c`. */
int srmap_comm_split(srmap_comm* c, int color, int key, int new_rank, int new_world, srmap_comm** out) {
  if (!c || !out || new_world < 1 || new_rank < 0 || new_rank >= new_world) return SRMAP_EINVAL;
  *out = nullptr;
  if (c->kind != 1) return set_error(c->ctx, SRMAP_EUNSUPPORTED, "srmap_comm_split needs the RCCL backend");
  if (!c->api->CommSplit) return set_error(c->ctx, SRMAP_EUNSUPPORTED, "this librccl has no ncclCommSplit");
  SRMAP_HIP(c->ctx, hipSetDevice(c->ctx->device));
  srmap_comm* n = new (std::nothrow) srmap_comm();
  if (!n) return SRMAP_ENOMEM;
  n->ctx = c->ctx; n->rank = new_rank; n->world = new_world; n->kind = 1; n->api = c->api;
  const ncclResult_t r = c->api->CommSplit(c->nccl, color, key, &n->nccl, nullptr);
  if (r != ncclSuccess || n->nccl == nullptr) {
    delete n;
    return set_error(c->ctx, SRMAP_EHIP, "ncclCommSplit failed: %s", c->api->GetErrorString(r));
  }
  *out = n;
  return SRMAP_OK;
}

void srmap_comm_destroy(srmap_comm* c) {
  if (!c) return;
  if (c->kind == 1 && c->nccl && c->api) (void)c->api->CommDestroy(c->nccl);
  if (c->h_send) (void)hipHostFree(c->h_send);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->ev_x) (void)hipEventDestroy(c->ev_x);
  if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
  delete c;
}

}  // extern "C"
