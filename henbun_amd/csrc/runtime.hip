// Runtime plumbing of libhenbun_hip.so: error string, device info, hipGraph
// capture/replay of a launch sequence.
#include "common.cuh"
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
