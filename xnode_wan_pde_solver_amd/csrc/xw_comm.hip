// xw_comm.hip -- the one exchange step of the path: in-place float64 sum over the ranks of a node (RCCL over xGMI).
//
// Replaces what nn.DataParallel does around both nets in the reference (src/training.py:93-97: scatter the batch,
// gather the outputs, reduce-add the gradients): here every rank keeps its shard of the Monte-Carlo paths and only the
// packed partial sums / gradients of a sub-step cross the links -- ONE all-reduce per generator sub-step, two per
// discriminator sub-step (dist.py).  The call only enqueues on the caller's stream, so it can sit inside the captured
// HIP graph of a sub-step (no host round trip between the kernels before and after the exchange).
//
// RCCL is bound at run time (dlopen): the library stays loadable on a host without a GPU, and a process that has
// already loaded PyTorch's own copy of librccl shares it instead of mapping a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstring>
#include "xnwan.h"

namespace {
typedef struct { char internal[128]; } XwNcclUniqueId;     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_unique_id)(XwNcclUniqueId*);
typedef int (*fn_comm_init_rank)(void**, int, XwNcclUniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_comm_destroy)(void*);
struct Rccl {
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  bool ok = false;
};
const Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // PyTorch's copy, if the process has it mapped
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return x;
    x.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    x.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    x.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    x.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    x.ok = x.get_unique_id && x.comm_init_rank && x.all_reduce && x.comm_destroy;
    return x;
  }();
  return r;
}
const int kNcclFloat64 = 8, kNcclSum = 0;                  // ncclDataType_t::ncclFloat64, ncclRedOp_t::ncclSum (rccl.h)
}  // namespace

// 0 when RCCL could be bound in this process (all four symbols), XW_E_COMM otherwise.  Not collective: the host side asks
// every rank BEFORE anyone enters xw_comm_init -- ncclCommInitRank is collective, a rank that returned early from it would
// leave the others blocked in the bootstrap.
extern "C" int xw_comm_available(void) { return rccl().ok ? 0 : XW_E_COMM; }

extern "C" int xw_comm_unique_id(unsigned char* id128) {
  if (!id128) return XW_E_ARG;
  if (!rccl().ok) return XW_E_COMM;
  XwNcclUniqueId id;
  const int rc = rccl().get_unique_id(&id);
  if (rc != 0) return XW_E_COMM;
  memcpy(id128, id.internal, 128);
  return 0;
}

extern "C" int xw_comm_init(const unsigned char* id128, int nranks, int rank, void** comm) {
  if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return XW_E_ARG;
  if (!rccl().ok) return XW_E_COMM;
  XwNcclUniqueId id;
  memcpy(id.internal, id128, 128);
  return rccl().comm_init_rank(comm, nranks, id, rank) == 0 ? 0 : XW_E_COMM;
}

extern "C" int xw_allreduce(double* buf, int count, void* comm, void* stream) {
  if (!buf || count <= 0 || !comm) return XW_E_ARG;
  if (!rccl().ok) return XW_E_COMM;
  return rccl().all_reduce(buf, buf, (size_t)count, kNcclFloat64, kNcclSum, comm, (hipStream_t)stream) == 0 ? 0 : XW_E_COMM;
}

extern "C" int xw_comm_destroy(void* comm) {
  if (!comm) return XW_E_ARG;
  if (!rccl().ok) return XW_E_COMM;
  return rccl().comm_destroy(comm) == 0 ? 0 : XW_E_COMM;
}
