// comm.hip — RCCL-backed gradient exchange behind the C ABI (`pm_allreduce(buf, count, comm, stream)`).
//
// The reference is single-device (train.py:120-122); data parallelism is this build's addition.  The Python trainer
// exchanges gradients through torch.distributed (backend "nccl" = RCCL), whose communicator is private to PyTorch; a host
// that is NOT PyTorch (or native code that wants a collective between two kernels of the step, e.g. synchronised
// BatchNorm statistics) uses these entry points instead: one communicator per process / GPU, created from a 128-byte
// unique id that rank 0 generates and the host distributes by whatever means it has (torch.distributed store, MPI, a file).
//
// RCCL is resolved at run time (dlopen of librccl.so.1, the copy already loaded by the host process if any), so the
// library itself carries no link-time dependency on it and every other entry point works without RCCL installed.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
typedef struct { char internal[128]; } UniqueId;        // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                     // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
constexpr int kFloat32 = 7, kSum = 0;                   // ncclFloat32, ncclSum (rccl.h)

struct Api { void* h; GetUniqueIdFn uid; CommInitRankFn init; CommDestroyFn destroy; AllReduceFn allreduce; bool ok; };
Api& api() {
  static Api a = [] {
    Api r;
    memset(&r, 0, sizeof(r));
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) return r;
    r.uid = (GetUniqueIdFn)dlsym(r.h, "ncclGetUniqueId");
    r.init = (CommInitRankFn)dlsym(r.h, "ncclCommInitRank");
    r.destroy = (CommDestroyFn)dlsym(r.h, "ncclCommDestroy");
    r.allreduce = (AllReduceFn)dlsym(r.h, "ncclAllReduce");
    r.ok = r.uid && r.init && r.destroy && r.allreduce;
    return r;
  }();
  return a;
}
}  // namespace

extern "C" int pm_comm_unique_id(uint8_t* id128) {
  if (!id128) return PM_E_INVALID;
  if (!api().ok) return PM_E_UNSUPPORTED;
  UniqueId u;
  if (api().uid(&u) != 0) return PM_E_LAUNCH;
  memcpy(id128, u.internal, 128);
  return PM_OK;
}
extern "C" int pm_comm_init(const uint8_t* id128, int32_t rank, int32_t world, void** comm) {
  if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return PM_E_INVALID;
  if (!api().ok) return PM_E_UNSUPPORTED;
  UniqueId u;
  memcpy(u.internal, id128, 128);
  Comm c = nullptr;
  if (api().init(&c, world, u, rank) != 0 || !c) return PM_E_LAUNCH;
  *comm = c;
  return PM_OK;
}
extern "C" int pm_comm_destroy(void* comm) {
  if (!comm) return PM_E_INVALID;
  if (!api().ok) return PM_E_UNSUPPORTED;
  return api().destroy((Comm)comm) == 0 ? PM_OK : PM_E_LAUNCH;
}
// In-place sum over the ranks of `comm`, enqueued on `stream` (ordered after the kernels that produced `buf`).
extern "C" int pm_allreduce(float* buf, int64_t count, void* comm, pm_stream_t stream) {
  if (!buf || count <= 0 || !comm) return PM_E_INVALID;
  if (!api().ok) return PM_E_UNSUPPORTED;
  return api().allreduce(buf, buf, (size_t)count, kFloat32, kSum, (Comm)comm, (hipStream_t)stream) == 0 ? PM_OK : PM_E_LAUNCH;
}
