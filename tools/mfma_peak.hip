// Measures the achieved f32 / f64 MFMA rate and a trivial-kernel launch time (diagnostic only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_f32(float* out, int iters) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void __launch_bounds__(256) k_f64(double* out, int iters) {
  f64x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void k_empty(float* out) { if (threadIdx.x == 0) out[0] = 1.f; }
int main() {
  float* o; double* od;
  hipMalloc(&o, 1 << 22); hipMalloc(&od, 1 << 23);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    for (int iters : {256, 4096}) {
      int launches = iters == 256 ? 200 : 50;
      hipEventRecord(e0);
      for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_f32, dim3(256), dim3(256), 0, 0, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = 256.0 * 4 * iters * 4 * 4096.0 * launches;  // blocks*waves*iters*4 mfma*flops
      printf("f32 mfma iters=%d: %.2f us/launch  %.1f TF/s\n", iters, ms * 1e3 / launches, fl / (ms * 1e-3) * 1e-12);
      hipEventRecord(e0);
      for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_f64, dim3(256), dim3(256), 0, 0, od, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      double fl64 = 256.0 * 4 * iters * 4 * (16 * 16 * 4 * 2.0) * launches;
      printf("f64 mfma iters=%d: %.2f us/launch  %.1f TF/s\n", iters, ms * 1e3 / launches, fl64 / (ms * 1e-3) * 1e-12);
    }
    hipEventRecord(e0);
    for (int l = 0; l < 1000; ++l) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel: %.2f us/launch\n", ms);
  }
  return 0;
}
