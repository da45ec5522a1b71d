// collective.hip -- the one collective of the multi-GPU path:
// an in-place sum all-reduce of the THM_N_COUNTERS u64 counter vector over RCCL
// (xGMI inside a node).  Reads shard with no data-path exchange, so this is the
// only place ranks talk; 128 bytes, latency-bound.
//
// RCCL is resolved with dlopen at first use: a single-GPU host (or a box without
// librccl) can load and use the rest of the library.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "aligner_internal.h"

struct thm_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
};

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
    if (!r.h) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
  });
  return r;
}

int no_rccl() {
  thm::set_global_error("librccl.so could not be loaded: the counter all-reduce needs RCCL");
  return THM_ERR_UNSUPPORTED;
}

int nccl_fail(thm_aligner* a, const char* what, ncclResult_t e) {
  Rccl& r = rccl();
  return fail(a, THM_ERR_HIP, "%s failed: %s", what, r.GetErrorString ? r.GetErrorString(e) : "RCCL error");
}

}  // namespace

extern "C" {

int32_t thm_comm_unique_id(uint8_t out[THM_COMM_ID_BYTES]) {
  static_assert(THM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!out) return THM_ERR_INVALID_ARG;
  Rccl& r = rccl();
  if (!r.ok) return no_rccl();
  ncclUniqueId id;
  const ncclResult_t e = r.GetUniqueId(&id);
  if (e != ncclSuccess) return nccl_fail(nullptr, "ncclGetUniqueId", e);
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return THM_OK;
}

int32_t thm_comm_create(const uint8_t id[THM_COMM_ID_BYTES], int32_t nranks, int32_t rank, int32_t device_id, thm_comm** out) {
  if (!out) return THM_ERR_INVALID_ARG;
  *out = nullptr;
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) return THM_ERR_INVALID_ARG;
  Rccl& r = rccl();
  if (!r.ok) return no_rccl();
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev) {
    thm::set_global_error("thm_comm_create: no such HIP device");
    return THM_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device_id) != hipSuccess) return THM_ERR_HIP;
  ncclUniqueId uid;
  memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  thm_comm* c = new thm_comm();
  c->nranks = nranks;
  c->rank = rank;
  c->device = device_id;
  const ncclResult_t e = r.CommInitRank(&c->comm, nranks, uid, rank);
  if (e != ncclSuccess) {
    delete c;
    return nccl_fail(nullptr, "ncclCommInitRank", e);
  }
  *out = c;
  return THM_OK;
}

void thm_comm_free(thm_comm* c) {
  if (!c) return;
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)rccl().CommDestroy(c->comm);
  }
  delete c;
}

int32_t thm_counters_allreduce(thm_aligner* a, thm_comm* c) {
  if (!a || !c) return THM_ERR_INVALID_ARG;
  if (c->device != a->device) return fail(a, THM_ERR_INVALID_ARG, "communicator and aligner are on different devices");
  Rccl& r = rccl();
  if (!r.ok) return no_rccl();
  HIPCHK(a, hipSetDevice(a->device));
  const ncclResult_t e = r.AllReduce(a->d_counters.p, a->d_counters.p, THM_N_COUNTERS, ncclUint64, ncclSum, c->comm, a->stream);
  if (e != ncclSuccess) return nccl_fail(a, "ncclAllReduce", e);
  HIPCHK(a, hipStreamSynchronize(a->stream));
  return THM_OK;
}

}  // extern "C"
