// Per-launch cost of a chain of dependent tiny kernels: stream launches from C++ vs one captured hipGraph.
// Build: hipcc -O3 --offload-arch=gfx950 tools/launch_overhead.hip -o gpurun_out/launch_overhead
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void tiny(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
__global__ void empty_k() {}

int main() {
  const int N = 2000;
  float* buf;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(buf, 0, 1 << 20));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int blocks : {1, 16, 256}) {
    // (a) stream launches
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::high_resolution_clock::now();
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, buf, blocks * 256);
      CK(hipEventRecord(e1, st));
      auto t1 = std::chrono::high_resolution_clock::now();
      CK(hipStreamSynchronize(st));
      auto t2 = std::chrono::high_resolution_clock::now();
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1)
        printf("stream  blocks=%3d: GPU %.2f us/launch, CPU issue %.2f us/launch, wall %.2f us/launch\n", blocks,
               ms * 1e3 / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    // (b) graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, buf, blocks * 256);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("graph   blocks=%3d: GPU %.2f us/launch\n", blocks, ms * 1e3 / N);
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  // empty kernels
  {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream  empty     : GPU %.2f us/launch\n", ms * 1e3 / N);
  }
  return 0;
}
