// Diagnostic: cycles per v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 as a function of the number of independent accumulator chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4v __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, long long* cyc, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8 / (NACC > 8 ? 8 : NACC) * 1; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
template <int NACC>
__global__ void __launch_bounds__(256) k64(double* out, long long* cyc, int iters) {
  d4v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.;
  double a = threadIdx.x * 1e-3, b = 1.0;
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8 / (NACC > 8 ? 8 : NACC) * 1; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
template <int NACC>
void run() {
  float* out; double* outd; long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&outd, 256 * 256 * 8); (void)hipMalloc(&cyc, 8);
  const int iters = 2000;
  const int per_iter = (8 / (NACC > 8 ? 8 : NACC)) * NACC;
  long long h;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k32<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("f32 32x32x2, %d independent accumulators: %.1f cycles / MFMA (pipe-limited ideal 64)\n", NACC, (double)h / iters / per_iter);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k64<NACC>, dim3(256), dim3(256), 0, 0, outd, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("f64 16x16x4, %d independent accumulators: %.1f cycles / MFMA\n", NACC, (double)h / iters / per_iter);
}
int main() {
  run<1>(); run<2>(); run<4>(); run<8>();
  return 0;
}
