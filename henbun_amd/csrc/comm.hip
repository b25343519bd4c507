// Data-parallel exchange step: ONE RCCL all-reduce (sum) of the flat gradient buffer per Adam step, issued on the
// caller's stream so that it sits inside the captured step graph between the backward kernels and the Adam kernel.
//
// The reference has no counterpart (single tf.Session, reference Henbun/model.py:57,255-269).
//
// RCCL is resolved at run time: first among the symbols already loaded into the process (PyTorch loads its own
// librccl.so.1 -- the process must not end up with two RCCL instances), then by dlopen("librccl.so.1").  The
// library itself carries no link-time dependency on RCCL, so it loads (and the ABI test runs) on a box without it.
#include "common.cuh"
#include "../../include/henbun_hip.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    void* h = RTLD_DEFAULT;
    if (dlsym(h, "ncclAllReduce") == nullptr) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) return a;
    }
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}
}  // namespace

#define HB_RCCL(call)                                                                     \
  do {                                                                                    \
    ncclResult_t r__ = (call);                                                            \
    if (r__ != ncclSuccess) {                                                             \
      hb_set_error("%s:%d: RCCL: %s", __FILE__, __LINE__, rccl().GetErrorString(r__));    \
      return 1000 + (int)r__;                                                             \
    }                                                                                     \
  } while (0)

extern "C" int hb_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int hb_comm_unique_id(char* id128) {
  HB_REQUIRE(id128 != nullptr, "hb_comm_unique_id: NULL buffer");
  HB_REQUIRE(rccl().ok, "hb_comm_unique_id: RCCL is not available in this process");
  static_assert(sizeof(ncclUniqueId) == HB_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  HB_RCCL(rccl().GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int hb_comm_init(const char* id128, int rank, int world, void** comm_out) {
  HB_REQUIRE(id128 && comm_out, "hb_comm_init: NULL pointer");
  HB_REQUIRE(world >= 1 && rank >= 0 && rank < world, "hb_comm_init: rank %d of %d", rank, world);
  HB_REQUIRE(rccl().ok, "hb_comm_init: RCCL is not available in this process");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  HB_RCCL(rccl().CommInitRank(&comm, world, id, rank));
  *comm_out = (void*)comm;
  return 0;
}

extern "C" int hb_comm_destroy(void* comm) {
  if (comm != nullptr && rccl().ok) HB_RCCL(rccl().CommDestroy((ncclComm_t)comm));
  return 0;
}

template <typename T>
static int allreduce_sum(T* buf, long n, void* comm, hipStream_t stream) {
  HB_REQUIRE(comm != nullptr, "hb_allreduce_sum: NULL communicator");
  HB_REQUIRE(n >= 0 && (n == 0 || buf != nullptr), "hb_allreduce_sum: bad buffer");
  if (n == 0) return 0;
  HB_RCCL(rccl().AllReduce(buf, buf, (size_t)n, sizeof(T) == 4 ? ncclFloat32 : ncclFloat64, ncclSum, (ncclComm_t)comm,
                           stream));
  return 0;
}
extern "C" int hb_allreduce_sum_f32(float* buf, long n, void* comm, void* stream) {
  return allreduce_sum<float>(buf, n, comm, (hipStream_t)stream);
}
extern "C" int hb_allreduce_sum_f64(double* buf, long n, void* comm, void* stream) {
  return allreduce_sum<double>(buf, n, comm, (hipStream_t)stream);
}

// The two extra words that travel with the gradient: tail[0] = this rank's objective value (the ranks' mean is
// what a data-parallel run reports), tail[1] = 1 if any factorisation of this rank's step failed.  After the
// all-reduce tail[1] != 0 on EVERY rank as soon as one rank failed, and hb_adam_step (dpflag) blocks the update
// everywhere: all ranks stay at the last good step and raise together.
template <typename T>
__global__ void __launch_bounds__(64) dp_pack_kernel(T* __restrict__ tail, const T* __restrict__ objective,
                                                     const int* __restrict__ info, long n_info) {
  int bad = 0;
  for (long i = threadIdx.x; i < n_info; i += 64) bad |= (info[i] != 0);
  bad = __any(bad);
  if (threadIdx.x == 0) {
    tail[0] = objective != nullptr ? objective[0] : T(0);
    tail[1] = bad ? T(1) : T(0);
  }
}
template <typename T>
static int dp_pack(T* tail, const T* objective, const int* info, long n_info, hipStream_t stream) {
  HB_REQUIRE(tail != nullptr, "hb_dp_pack: NULL tail");
  HB_REQUIRE(n_info >= 0 && (n_info == 0 || info != nullptr), "hb_dp_pack: info/n_info");
  hipLaunchKernelGGL(dp_pack_kernel<T>, dim3(1), dim3(64), 0, stream, tail, objective, info, n_info);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_dp_pack_f32(float* tail, const float* objective, const int* info, long n_info, void* stream) {
  return dp_pack<float>(tail, objective, info, n_info, (hipStream_t)stream);
}
extern "C" int hb_dp_pack_f64(double* tail, const double* objective, const int* info, long n_info, void* stream) {
  return dp_pack<double>(tail, objective, info, n_info, (hipStream_t)stream);
}
