// comm.hip.h -- RCCL over xGMI for the one exchange step of the path: the gather of
// per-member output at diagnostic time.  librccl.so (570 MB) is dlopen()ed lazily.
#pragma once
#include <dlfcn.h>
#include "common.hip.h"

namespace pm {

struct NcclId { char internal[PM_COMM_ID_BYTES]; };
typedef void *nccl_comm;
enum { NCCL_FLOAT64 = 8, NCCL_MAX = 2 };

struct NcclApi {
  int (*GetUniqueId)(NcclId *);
  int (*CommInitRank)(nccl_comm *, int, NcclId, int);
  int (*CommDestroy)(nccl_comm);
  int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t);
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t);
  int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t);
  int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
  const char *(*GetErrorString)(int);
  bool ok = false;
};

inline NcclApi &nccl() {
  static NcclApi api;
  if (api.ok) return api;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return api;
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
  api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
  api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
  api.Send = (decltype(api.Send))dlsym(h, "ncclSend");
  api.Recv = (decltype(api.Recv))dlsym(h, "ncclRecv");
  api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather &&
           api.AllReduce && api.GetErrorString && api.Send && api.Recv && api.GroupStart &&
           api.GroupEnd;
  return api;
}

struct Comm {
  nccl_comm comm = nullptr;
  int nranks = 0, rank = 0;
  double *scratch = nullptr;  // 2 doubles for the barrier all-reduce
};

#define PM_NCCL(call)                                                              \
  do {                                                                             \
    int r_ = (call);                                                               \
    if (r_ != 0)                                                                   \
      return pm::fail(PM_ENCCL, "%s failed: %s", #call, pm::nccl().GetErrorString(r_)); \
  } while (0)

}  // namespace pm
