// Data-parallel collectives behind the C ABI: RCCL (= the "nccl" backend of torch.distributed on ROCm) over xGMI, one communicator per
// process / GPU.  Replaces the reference's single-process nn.DataParallel gradient gather (/root/reference/src/train_ldm.py:190-192,
// /root/reference/src/train_autoencoderkl.py:230-233): every rank all-reduces the model's FLAT fp32 gradient buffer in a few large
// buckets on a dedicated communication stream, ordered after the backward by an event, so the transfer of the finished (deep) half of
// the gradients overlaps the rest of the backward (eegldm_unet_set_grad_hook).  xGMI is point-to-point (7 links per GPU), so ring
// collectives are per-link bound: few large buckets (32 MB default) rather than many small ones.
//
// RCCL is resolved with dlopen at communicator creation -- libeegldm.so has no link-time dependency on it, so the single-GPU paths
// and the CPU-side ABI checks never touch it, and a process that already holds torch's RCCL binds to that same copy.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

struct eegldm_comm {
  eegldm_ctx* ctx = nullptr;
  void* dl = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;                 // communication stream (non-blocking): collectives never sit in front of compute
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

namespace {
void* open_rccl() {
  // EEGLDM_RCCL_LIB=<path>: bind to a specific build of the library (a newer RCCL than the process already holds; the call-recording
  // stand-in of tests/fake_rccl, which lets the bucket arithmetic below run for world sizes no single-GPU box can provide)
  if (const char* over = getenv("EEGLDM_RCCL_LIB")) { if (*over) return dlopen(over, RTLD_NOW | RTLD_LOCAL); }
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) return h; }
  return nullptr;
}
#define NCCL_TRY(c, expr)                                                                                     \
  do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) EEG_FAIL(EEGLDM_ERR_HIP, "RCCL: %s -> %s", #expr, (c)->GetErrorString ? (c)->GetErrorString(r_) : "?"); } while (0)
}  // namespace

extern "C" int eegldm_comm_unique_id(char* out128) {
  EEG_CHECK(out128, "null pointer");
  void* h = open_rccl();
  EEG_CHECK(h, "librccl.so not found (%s)", dlerror());
  auto get = (ncclResult_t(*)(ncclUniqueId*))dlsym(h, "ncclGetUniqueId");
  EEG_CHECK(get, "ncclGetUniqueId missing");
  ncclUniqueId id;
  const ncclResult_t r = get(&id);
  EEG_CHECK(r == ncclSuccess, "ncclGetUniqueId failed (%d)", (int)r);
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int eegldm_comm_destroy(eegldm_comm* c);
extern "C" int eegldm_comm_create(eegldm_ctx* ctx, const char* id128, int rank, int world, eegldm_comm** out) {
  EEG_CHECK(ctx && id128 && out && world >= 1 && rank >= 0 && rank < world, "bad argument (rank %d of %d)", rank, world);
  eegldm_comm* c = new eegldm_comm();
  c->ctx = ctx; c->rank = rank; c->world = world;
  c->dl = open_rccl();
  if (!c->dl) { delete c; EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "librccl.so not found (%s)", dlerror()); }
#define SYM(field, name) do { *(void**)(&c->field) = dlsym(c->dl, name); if (!c->field) { delete c; EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "RCCL symbol %s missing", name); } } while (0)
  SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(AllReduce, "ncclAllReduce"); SYM(Broadcast, "ncclBroadcast");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  // from here on a failure must not leak the struct (nor a half-made communicator / stream / events): the *_TRY macros return, so the
  // fallible part runs in a lambda and eegldm_comm_destroy() -- which tolerates every partially initialised state -- cleans up
  auto init = [&]() -> int {
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id; memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    NCCL_TRY(c, c->CommInitRank(&c->comm, world, id, rank));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    return 0;
  };
  const int rc = init();
  if (rc != 0) { (void)eegldm_comm_destroy(c); return rc; }
  *out = c;
  return 0;
}

extern "C" int eegldm_comm_destroy(eegldm_comm* c) {
  if (!c) return 0;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && c->CommDestroy) (void)c->CommDestroy(c->comm);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int eegldm_comm_rank(const eegldm_comm* c) { return c ? c->rank : -1; }
extern "C" int eegldm_comm_world(const eegldm_comm* c) { return c ? c->world : 0; }

// buf[0 .. n) <- mean over ranks, in place, in buckets of bucket_elems (<= 0: one collective).  Ordered AFTER everything enqueued on the
// context's stream so far; runs on the communication stream; eegldm_comm_wait() makes the context's stream wait for it.  ncclAvg: the
// 1 / world scaling is part of the collective (no separate pass over the buffer).
extern "C" int eegldm_comm_allreduce_mean_f32(eegldm_comm* c, float* buf, long n, long bucket_elems) {
  EEG_CHECK(c && n >= 0 && (buf || n == 0), "bad argument");
  if (n == 0) return 0;
  HIP_TRY(hipEventRecord(c->ev_ready, c->ctx->stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
  const long step = bucket_elems > 0 ? bucket_elems : n;
  NCCL_TRY(c, c->GroupStart());
  for (long s = 0; s < n; s += step) {
    const long e = s + step < n ? s + step : n;
    const ncclResult_t r = c->AllReduce(buf + s, buf + s, (size_t)(e - s), ncclFloat32, ncclAvg, c->comm, c->stream);
    if (r != ncclSuccess) { (void)c->GroupEnd(); EEG_FAIL(EEGLDM_ERR_HIP, "ncclAllReduce -> %s", c->GetErrorString(r)); }
  }
  NCCL_TRY(c, c->GroupEnd());
  return 0;
}

// parameters of rank `root` to everybody (start of training), same ordering rules
extern "C" int eegldm_comm_broadcast_f32(eegldm_comm* c, float* buf, long n, int root) {
  EEG_CHECK(c && buf && n >= 0 && root >= 0 && root < c->world, "bad argument");
  if (n == 0) return 0;
  HIP_TRY(hipEventRecord(c->ev_ready, c->ctx->stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
  NCCL_TRY(c, c->Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm, c->stream));
  return 0;
}

// the context's stream waits (on the device) for every collective issued so far: call before the optimizer reads the gradients
extern "C" int eegldm_comm_wait(eegldm_comm* c) {
  EEG_CHECK(c, "null communicator");
  HIP_TRY(hipEventRecord(c->ev_done, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
  return 0;
}
