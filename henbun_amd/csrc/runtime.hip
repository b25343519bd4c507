// Runtime plumbing of libhenbun_hip.so: error string, device info, hipGraph
// capture/replay of a launch sequence.
#include "common.cuh"
#include "side_jobs.cuh"
#include "../../include/henbun_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void hb_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hb_last_error_string(void) { return g_err; }

// ---- diagnostic switches: see common.cuh.  A handful of keys, set by tests and tools before the calls they steer.
#include <mutex>
namespace {
struct HbDebugEntry { char key[40]; long value; };
HbDebugEntry g_dbg[32];
int g_ndbg = 0;
std::mutex g_dbg_mu;
}  // namespace
long hb_debug_get(const char* key, long dflt) {
  std::lock_guard<std::mutex> lk(g_dbg_mu);
  for (int i = 0; i < g_ndbg; ++i)
    if (strcmp(g_dbg[i].key, key) == 0) return g_dbg[i].value;
  return dflt;
}
extern "C" int hb_debug_set(const char* key, long value) {
  HB_REQUIRE(key && key[0] && strlen(key) < sizeof(g_dbg[0].key), "hb_debug_set: bad key");
  std::lock_guard<std::mutex> lk(g_dbg_mu);
  for (int i = 0; i < g_ndbg; ++i)
    if (strcmp(g_dbg[i].key, key) == 0) {
      g_dbg[i].value = value;
      return 0;
    }
  HB_REQUIRE(g_ndbg < 32, "hb_debug_set: table full");
  strcpy(g_dbg[g_ndbg].key, key);
  g_dbg[g_ndbg++].value = value;
  return 0;
}
extern "C" int hb_debug_clear(void) {
  std::lock_guard<std::mutex> lk(g_dbg_mu);
  g_ndbg = 0;
  return 0;
}
extern "C" int hb_version(void) { return HB_ABI_VERSION; }

extern "C" int hb_device_info(char* buf, int buflen, int* cu_count) {
  int dev = 0;
  HB_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HB_HIP(hipGetDeviceProperties(&prop, dev));
  if (buf && buflen > 0) snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
  if (cu_count) *cu_count = prop.multiProcessorCount;
  return 0;
}

extern "C" int hb_graph_begin_capture(void* stream) {
  HB_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}

extern "C" int hb_graph_end_capture(void* stream, void** exec_out) {
  HB_REQUIRE(exec_out != nullptr, "hb_graph_end_capture: exec_out is NULL");
  hipGraph_t graph = nullptr;
  HB_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) {
    hb_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return (int)e;
  }
  *exec_out = (void*)exec;
  return 0;
}

extern "C" int hb_graph_launch(void* exec, void* stream) {
  HB_REQUIRE(exec != nullptr, "hb_graph_launch: exec is NULL");
  HB_HIP(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
  return 0;
}

extern "C" int hb_graph_destroy(void* exec) {
  if (exec) HB_HIP(hipGraphExecDestroy((hipGraphExec_t)exec));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Side jobs (side_jobs.cuh): the pending list of the calling thread, and the launch of their own that flushes it.
// ---------------------------------------------------------------------------------------------------------------
static thread_local HbSideJobs hb_side_list = {};

__global__ void __launch_bounds__(256) hb_side_jobs_kernel(HbSideJobs J) { hb_side_run(J, (int)blockIdx.x); }

static int hb_side_launch(const HbSideJobs& J, hipStream_t stream) {
  if (J.n == 0 || J.total == 0) return 0;
  hipLaunchKernelGGL(hb_side_jobs_kernel, dim3((unsigned)J.total), dim3(256), 0, stream, J);
  HB_LAUNCH_CHECK();
  return 0;
}

HbSideJobs hb_side_take() {
  HbSideJobs J = hb_side_list;
  hb_side_list.n = 0;
  hb_side_list.total = 0;
  return J;
}

int hb_side_push(const HbSideJob& job, hipStream_t stream) {
  if (hb_side_list.n == HB_SIDE_MAX) {   // full: what is pending runs now, as a launch of its own
    const HbSideJobs J = hb_side_take();
    const int rc = hb_side_launch(J, stream);
    if (rc) return rc;
  }
  hb_side_list.job[hb_side_list.n++] = job;
  hb_side_list.total += job.nblocks;
  return 0;
}

extern "C" int hb_side_pending(void) { return hb_side_list.n; }
// Drop the recorded jobs WITHOUT running them: the caller's launch sequence was abandoned between a push and its host
// (an argument check failed, a Python exception, a failed capture), and the raw device pointers they hold must not ride
// on some unrelated later launch.  Returns the number of jobs dropped.
extern "C" int hb_side_discard(void) {
  const int n = hb_side_list.n;
  hb_side_list.n = 0;
  hb_side_list.total = 0;
  return n;
}
extern "C" int hb_side_flush(void* stream) {
  const HbSideJobs J = hb_side_take();
  return hb_side_launch(J, (hipStream_t)stream);
}
