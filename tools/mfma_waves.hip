// Diagnostic: how the fp32 matrix pipe of ONE SIMD is shared by 1, 2 or 4 waves that each run a DEPENDENT chain of MFMAs
// (what the strip kernels' waves do).  One workgroup of 256 / 512 / 1024 threads on one CU; every wave times its own
// chain; cycles per MFMA per SIMD = slowest wave's cycles / (chain length x waves per SIMD).  64 (32x32x2) / 32 (16x16x4)
// would be a pipe that never idles.   hipcc -O3 --offload-arch=gfx950 tools/mfma_waves.hip -o tools/_bin/mfma_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int KIND, int NACC>   // KIND 0: 32x32x2, 1: 16x16x4;  NACC independent accumulators per wave
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, int iters) {
  f16v acc[NACC];
  f4v acc4[NACC];
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  }
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  __syncthreads();
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        else
          acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
      }
  }
  const long long c1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 4; ++r) s += acc4[i][r];
  }
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = c1 - c0;
}

template <int KIND, int NACC>
void run(int threads) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&cyc, 16 * 8);
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, NACC>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[16];
  (void)hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  long long mx = 0;
  for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  const int wps = threads / 256;
  printf("%s  %d accumulator chain(s) per wave  %d wave(s) per SIMD: %.1f cycles per MFMA per SIMD\n", KIND ? "16x16x4" : "32x32x2", NACC, wps,
         (double)mx / (iters * 16.0 * wps));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int t : {256, 512, 1024}) run<0, 1>(t);
  for (int t : {256, 512, 1024}) run<0, 2>(t);
  for (int t : {256, 512, 1024}) run<1, 1>(t);
  for (int t : {256, 512, 1024}) run<1, 2>(t);
  return 0;
}
