// Gradient exchange of the data-parallel step on a caller-supplied RCCL communicator (scripts/train.py:1127: the
// DistributedDataParallel wrap all-reduces the trainable parameters' gradients during backward).  The Python host uses
// torch.distributed (backend "nccl" = RCCL) for the same exchange; this entry point is what a C / C++ host that owns its
// own ncclComm_t binds.  librccl is opened on first use, so single-GPU users of the library carry no RCCL dependency.
#include <dlfcn.h>

#include "common.hpp"

namespace tcavt {
namespace {
// ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
using allreduce_fn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
using errstr_fn = const char* (*)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;  // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Rccl {
  allreduce_fn allreduce = nullptr;
  errstr_fn errstr = nullptr;
  const char* why = nullptr;
};

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      x.why = "librccl.so not found (dlopen)";
      return x;
    }
    x.allreduce = reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce"));
    x.errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
    if (!x.allreduce) x.why = "ncclAllReduce not found in librccl";
    return x;
  }();
  return r;
}
}  // namespace
}  // namespace tcavt

using namespace tcavt;

extern "C" int tcavt_allreduce_flat(float* buf, int64_t n, void* nccl_comm, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(buf && n > 0 && nccl_comm, "allreduce_flat: null buffer / communicator or n <= 0");
  const Rccl& r = rccl();
  if (!r.allreduce) {
    set_error("allreduce_flat: %s", r.why ? r.why : "RCCL unavailable");
    return TCAVT_ERR_HIP;
  }
  const int rc = r.allreduce(buf, buf, static_cast<size_t>(n), kNcclFloat32, kNcclSum, nccl_comm, static_cast<hipStream_t>(stream));
  if (rc != 0) {
    set_error("allreduce_flat: ncclAllReduce failed: %s", r.errstr ? r.errstr(rc) : "(no error string)");
    return TCAVT_ERR_HIP;
  }
  return TCAVT_OK;
}
