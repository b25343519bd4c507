// STREAM-like HBM bandwidth of this MI355X ("measure the achievable peak on the box"): float4 copy,
// read-only sum and write-only fill over buffers far larger than the 256 MiB Infinity Cache.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_peak.hip -o tools/_bin/stream_peak && tools/_bin/stream_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ a, f4* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_copy_nt(const f4* __restrict__ a, f4* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]), &b[i]);
}
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ a, float* __restrict__ out, long n) {
  f4 s = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += a[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;   // never true: keeps the loads
}
__global__ void __launch_bounds__(256) k_fill(f4* __restrict__ b, long n) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = v;
}

template <typename F>
static double time_us(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / iters;
}

int main() {
  const long bytes = 2L << 30;   // 2 GiB per buffer
  const long n = bytes / 16;
  f4 *a, *b;
  float* out;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 0, bytes);
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs; buffers of %.1f GiB\n", p.name, p.multiProcessorCount, bytes / 1073741824.0);
  for (int wg_per_cu : {4, 8, 16, 32}) {
    const int grid = p.multiProcessorCount * wg_per_cu;
    const double c = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 10);
    const double cn = time_us([&] { hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(256), 0, 0, a, b, n); }, 10);
    const double r = time_us([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n); }, 10);
    const double w = time_us([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n); }, 10);
    printf("grid %5d (%2d WG/CU): copy %.2f TB/s (read+write bytes)  copy-nt %.2f TB/s  read %.2f TB/s  fill %.2f TB/s\n", grid,
           wg_per_cu, 2.0 * bytes / c * 1e-6, 2.0 * bytes / cn * 1e-6, bytes / r * 1e-6, bytes / w * 1e-6);
  }
  return 0;
}
