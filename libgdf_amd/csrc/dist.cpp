// dist.cpp -- the wire of gdf_amd_dist_inner_join (include/gdf/gdf_amd_ext.h): the RCCL implementation of gdf_amd_transport.
//
// No counterpart in the reference (single-GPU: SURVEY.md section 2 rows 34-35, 8e).  RCCL API: /opt/rocm/include/rccl/rccl.h.
// librccl is resolved with dlopen when the first transport is made: libgdf.so itself has no link-time dependency on it, and a
// process that already carries an RCCL (PyTorch bundles one under the same soname) shares that copy.
//
// xGMI is point-to-point: an all-to-all of equal blocks is one group of ncclSend / ncclRecv pairs, which drives all of a GPU's
// links at once.  Messages are cut at 2^29 bytes: RCCL 2.26.6 delivered only half of a >= 2.0e9-byte send / recv
// (tools/rccl_message_size_check.py).  The collectives run on a NON-BLOCKING stream of the transport's own, ordered against the
// library's (legacy default) stream with events, so that a slice travels while the next one is regrouped.
#include "common.h"
#include "gdf/gdf_amd_ext.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>

namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int *);
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
  bool ok = false;
};

const RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    bool all = true;
    auto sym = [&](const char *name) { void *p = dlsym(h, name); all = all && p != nullptr; return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
    api.ok = all;
  });
  return api;
}

constexpr size_t MAX_MESSAGE = (size_t)1 << 29;

struct RcclCtx {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;        // non-blocking: never implicitly ordered against the library's legacy default stream
  int64_t *scratch = nullptr;          // [16] device words of the small all-reduces
  int rank = 0, world = 1;
};

int rccl_all_to_all(void *vctx, const void *send, void *recv, size_t bytes_per_rank, void **ticket) {
  RcclCtx *c = static_cast<RcclCtx *>(vctx);
  const RcclApi &r = rccl();
  hipEvent_t ready = nullptr, done = nullptr;
  if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return 1;
  if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(ready); return 1; }
  // behind everything the library has issued so far (the regroup kernel that filled `send`)
  bool ok = hipEventRecord(ready, gdf_amd::stream0()) == hipSuccess && hipStreamWaitEvent(c->stream, ready, 0) == hipSuccess;
  const char *s = static_cast<const char *>(send);
  char *d = static_cast<char *>(recv);
  // a rank's own block is a device copy; the others one send / recv pair per piece, all pieces of one offset in one group
  if (ok && bytes_per_rank)
    ok = hipMemcpyAsync(d + (size_t)c->rank * bytes_per_rank, s + (size_t)c->rank * bytes_per_rank, bytes_per_rank, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
  for (size_t off = 0; ok && off < bytes_per_rank && c->world > 1; off += MAX_MESSAGE) {
    const size_t len = bytes_per_rank - off < MAX_MESSAGE ? bytes_per_rank - off : MAX_MESSAGE;
    ok = r.GroupStart() == ncclSuccess;
    for (int p = 0; ok && p < c->world; ++p) {
      if (p == c->rank) continue;
      ok = r.Send(s + (size_t)p * bytes_per_rank + off, len, ncclChar, p, c->comm, c->stream) == ncclSuccess &&
           r.Recv(d + (size_t)p * bytes_per_rank + off, len, ncclChar, p, c->comm, c->stream) == ncclSuccess;
    }
    ok = (r.GroupEnd() == ncclSuccess) && ok;
  }
  ok = ok && hipEventRecord(done, c->stream) == hipSuccess;
  (void)hipEventDestroy(ready);
  if (!ok) { (void)hipEventDestroy(done); return 1; }
  *ticket = done;
  return 0;
}

// the all-to-all-v: one ncclSend / ncclRecv pair per peer and 2^29-byte piece, every piece of one offset in one group (xGMI is
// point-to-point: the group drives all links at once); a rank's own share is a device copy
int rccl_all_to_all_v(void *vctx, const void *send, const size_t *send_off, void *recv, const size_t *recv_off, void **ticket) {
  RcclCtx *c = static_cast<RcclCtx *>(vctx);
  const RcclApi &r = rccl();
  if (!send_off || !recv_off) return 1;
  hipEvent_t ready = nullptr, done = nullptr;
  if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return 1;
  if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(ready); return 1; }
  bool ok = hipEventRecord(ready, gdf_amd::stream0()) == hipSuccess && hipStreamWaitEvent(c->stream, ready, 0) == hipSuccess;
  const char *s = static_cast<const char *>(send);
  char *d = static_cast<char *>(recv);
  const size_t mine = send_off[c->rank + 1] - send_off[c->rank];
  if (ok && mine != recv_off[c->rank + 1] - recv_off[c->rank]) ok = false;
  if (ok && mine) ok = hipMemcpyAsync(d + recv_off[c->rank], s + send_off[c->rank], mine, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
  size_t longest = 0;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    longest = std::max(longest, std::max(send_off[p + 1] - send_off[p], recv_off[p + 1] - recv_off[p]));
  }
  for (size_t off = 0; ok && off < longest; off += MAX_MESSAGE) {
    ok = r.GroupStart() == ncclSuccess;
    for (int p = 0; ok && p < c->world; ++p) {
      if (p == c->rank) continue;
      const size_t sl = send_off[p + 1] - send_off[p], rl = recv_off[p + 1] - recv_off[p];
      if (off < sl) ok = r.Send(s + send_off[p] + off, std::min(sl - off, MAX_MESSAGE), ncclChar, p, c->comm, c->stream) == ncclSuccess;
      if (ok && off < rl) ok = r.Recv(d + recv_off[p] + off, std::min(rl - off, MAX_MESSAGE), ncclChar, p, c->comm, c->stream) == ncclSuccess;
    }
    ok = (r.GroupEnd() == ncclSuccess) && ok;
  }
  ok = ok && hipEventRecord(done, c->stream) == hipSuccess;
  (void)hipEventDestroy(ready);
  if (!ok) { (void)hipEventDestroy(done); return 1; }
  *ticket = done;
  return 0;
}

int rccl_wait(void *, void *ticket) {
  hipEvent_t done = static_cast<hipEvent_t>(ticket);
  if (!done) return 0;
  const bool ok = hipStreamWaitEvent(gdf_amd::stream0(), done, 0) == hipSuccess;
  (void)hipEventDestroy(done);         // (the runtime keeps a recorded event alive until the waits on it are through)
  return ok ? 0 : 1;
}

int rccl_all_reduce_i64(void *vctx, int64_t *values, int count, int op) {
  RcclCtx *c = static_cast<RcclCtx *>(vctx);
  if (count < 0 || count > 16 || op < 0 || op > 2) return 1;
  if (count == 0 || c->world == 1) return 0;
  const ncclRedOp_t ops[3] = {ncclMin, ncclMax, ncclSum};
  if (hipMemcpyAsync(c->scratch, values, sizeof(int64_t) * count, hipMemcpyHostToDevice, c->stream) != hipSuccess) return 1;
  if (rccl().AllReduce(c->scratch, c->scratch, (size_t)count, ncclInt64, ops[op], c->comm, c->stream) != ncclSuccess) return 1;
  if (hipMemcpyAsync(values, c->scratch, sizeof(int64_t) * count, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 1;
  return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : 1;
}

void rccl_destroy(void *vctx) {
  RcclCtx *c = static_cast<RcclCtx *>(vctx);
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) gdf_error gdf_amd_rccl_unique_id(char id[128]) {
  return gdf_amd::guarded([&]() -> gdf_error {
    GDF_REQUIRE(id, GDF_DATASET_EMPTY);
    GDF_REQUIRE(rccl().ok, GDF_UNSUPPORTED_METHOD);                  // no librccl in this process / on this machine
    ncclUniqueId u;
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    GDF_REQUIRE(rccl().GetUniqueId(&u) == ncclSuccess, GDF_C_ERROR);
    std::memcpy(id, &u, sizeof(u));
    return GDF_SUCCESS;
  });
}

__attribute__((visibility("default"))) gdf_error gdf_amd_rccl_transport_create(const char id[128], int world, int rank, gdf_amd_transport **out) {
  return gdf_amd::guarded([&]() -> gdf_error {
    GDF_REQUIRE(id && out, GDF_DATASET_EMPTY);
    GDF_REQUIRE(world >= 1 && rank >= 0 && rank < world, GDF_INVALID_API_CALL);
    GDF_REQUIRE(rccl().ok, GDF_UNSUPPORTED_METHOD);
    RcclCtx *c = new RcclCtx();
    c->rank = rank;
    c->world = world;
    gdf_amd_transport *t = new gdf_amd_transport{c, rank, world, rccl_all_to_all, rccl_wait, rccl_all_reduce_i64, rccl_destroy, rccl_all_to_all_v};
    auto fail = [&](gdf_error e) { rccl_destroy(c); delete t; return e; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(GDF_CUDA_ERROR);
    if (hipMalloc((void **)&c->scratch, sizeof(int64_t) * 16) != hipSuccess) return fail(GDF_MEMORYMANAGER_ERROR);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    if (rccl().CommInitRank(&c->comm, world, u, rank) != ncclSuccess) { c->comm = nullptr; return fail(GDF_C_ERROR); }
    *out = t;
    return GDF_SUCCESS;
  });
}

// what the RCCL COMMUNICATOR of a transport made by gdf_amd_rccl_transport_create says about itself (ncclCommCount / ncclCommUserRank):
// bench.py reports these next to WORLD_SIZE, so that a multi-GPU line shows how many ranks the collectives really spanned
__attribute__((visibility("default"))) gdf_error gdf_amd_rccl_transport_ranks(gdf_amd_transport *transport, int *nranks, int *rank) {
  return gdf_amd::guarded([&]() -> gdf_error {
    GDF_REQUIRE(transport && nranks && rank, GDF_DATASET_EMPTY);
    GDF_REQUIRE(transport->all_to_all == rccl_all_to_all && transport->ctx, GDF_INVALID_API_CALL);      // one of ours
    RcclCtx *c = static_cast<RcclCtx *>(transport->ctx);
    GDF_REQUIRE(rccl().ok && c->comm, GDF_UNSUPPORTED_METHOD);
    GDF_REQUIRE(rccl().CommCount(c->comm, nranks) == ncclSuccess && rccl().CommUserRank(c->comm, rank) == ncclSuccess, GDF_C_ERROR);
    return GDF_SUCCESS;
  });
}

__attribute__((visibility("default"))) void gdf_amd_transport_free(gdf_amd_transport *transport) {
  if (!transport) return;
  if (transport->destroy) transport->destroy(transport->ctx);
  delete transport;
}

__attribute__((visibility("default"))) gdf_error gdf_amd_copy(void *dst, const void *src, size_t bytes, int direction) {
  return gdf_amd::guarded([&]() -> gdf_error {
    GDF_REQUIRE(direction >= 0 && direction <= 2, GDF_INVALID_API_CALL);
    if (bytes == 0) return GDF_SUCCESS;
    GDF_REQUIRE(dst && src, GDF_DATASET_EMPTY);
    const hipMemcpyKind kind = direction == 0 ? hipMemcpyDeviceToHost : (direction == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, gdf_amd::stream0()));
    HIP_TRY(hipStreamSynchronize(gdf_amd::stream0()));
    return GDF_SUCCESS;
  });
}

}  // extern "C"
