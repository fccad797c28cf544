// Test infrastructure, NOT product: a call-recording stand-in for librccl, loaded by csrc/comm.hip through EEGLDM_RCCL_LIB.
// Purpose: run eegldm_comm_allreduce_mean_f32 / _broadcast_f32 / _wait -- the bucket arithmetic, the group bracket, the stream ordering
// against the context's stream -- for world sizes 2 / 4 / 8 on a box with ONE GPU, where a real communicator of that size cannot exist.
// The "collective" it performs is deterministic and checkable: ncclAllReduce(ncclAvg) adds 1.0 to every element it was handed (on the
// stream it was handed), so after one eegldm_comm_allreduce_mean_f32 every element of the buffer must have grown by exactly 1 -- an
// element covered by no bucket stays, one covered twice grows by 2.  Every call is logged (kind, element offset relative to the first
// call's pointer, count, dtype, op, inside-a-group flag) and can be read back with fake_rccl_log().
// Reference for what the real calls do: /root/reference/src/train_ldm.py:190-192 (nn.DataParallel gradient reduction) is what the
// communicator replaces; the API is RCCL's (rccl/rccl.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

namespace {
struct Rec { int kind; long off; long count; int dtype, op, in_group, world, rank; };
std::vector<Rec> g_log;
const char* g_base = nullptr;
int g_group = 0, g_world = 0, g_rank = 0, g_live = 0;
__global__ void add_one(float* p, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id->internal, 0x5a, NCCL_UNIQUE_ID_BYTES); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; i++) if (id.internal[i] != 0x5a) return ncclInvalidArgument;      // the id must arrive intact
  if (rank < 0 || rank >= nranks) return ncclInvalidArgument;
  g_world = nranks; g_rank = rank; g_live++;
  *comm = (ncclComm_t)(&g_live);
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t) { g_live--; return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_group++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (g_group <= 0) return ncclInvalidUsage; g_group--; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake_rccl error"; }
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t, hipStream_t stream) {
  if (send != recv) return ncclInvalidArgument;                 // the library reduces in place
  if (!g_base) g_base = (const char*)recv;
  g_log.push_back({0, (long)(((const char*)recv - g_base) / 4), (long)count, (int)dt, (int)op, g_group, g_world, g_rank});
  if (count) hipLaunchKernelGGL(add_one, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, (float*)recv, count);
  return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t, hipStream_t) {
  if (send != recv) return ncclInvalidArgument;
  g_log.push_back({1, 0, (long)count, (int)dt, root, g_group, g_world, g_rank});
  return ncclSuccess;
}
// ---- test accessors
int fake_rccl_log_size() { return (int)g_log.size(); }
int fake_rccl_log(int i, long out[8]) {
  if (i < 0 || i >= (int)g_log.size()) return -1;
  const Rec& r = g_log[i];
  out[0] = r.kind; out[1] = r.off; out[2] = r.count; out[3] = r.dtype; out[4] = r.op; out[5] = r.in_group; out[6] = r.world; out[7] = r.rank;
  return 0;
}
void fake_rccl_reset() { g_log.clear(); g_base = nullptr; }
int fake_rccl_live() { return g_live; }
int fake_rccl_avg_op() { return (int)ncclAvg; }
int fake_rccl_f32() { return (int)ncclFloat32; }
}
