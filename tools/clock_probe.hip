// Diagnostic: effective shader clock seen by a tiny kernel vs a chip-filling one.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(float* out, long long* cyc, long long* wall, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  for (int i = 0; i < iters; ++i) x = fmaf(x, y, 0.5f);   // dependent chain
  long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; wall[0] = w1 - w0; }
}
int main() {
  float* o; long long *c, *w; long long hc, hw;
  (void)hipMalloc(&o, 1 << 24); (void)hipMalloc(&c, 8); (void)hipMalloc(&w, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int wrate = 0; (void)hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", wrate);
  for (int blocks : {1, 16, 256, 2048}) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      for (int l = 0; l < 200; ++l) hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, o, c, w, 2000);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hw, w, 8, hipMemcpyDeviceToHost);
      double ns = hw * 1e6 / wrate;
      printf("blocks=%4d: %.2f us/launch; in-kernel: %lld shader cycles in %.0f ns -> %.0f MHz; %.2f cycles/fma\n", blocks,
             ms * 1e3 / 200, hc, ns, hc / ns * 1e3, hc / 2000.0);
    }
  }
  return 0;
}
