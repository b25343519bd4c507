// Diagnostic build of the column-strip contraction with in-kernel phase stamps (not part of the product).
#include <hip/hip_runtime.h>
__device__ long long hb_sstamps[8 * 8];
#define HB_SSTAMP(i)                                                                              \
  do {                                                                                            \
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) hb_sstamps[(threadIdx.x >> 6) * 8 + (i)] = clock64(); \
  } while (0)
#include "../henbun_amd/csrc/runtime.hip"
#include "../henbun_amd/csrc/elementwise.hip"
#include "../henbun_amd/csrc/linalg.hip"
#include "../henbun_amd/csrc/sgp.hip"
#include <stdio.h>
#include <vector>
int main() {
  const int M = 512, n = 8192;
  float *W, *z, *x, *A, *ell;
  (void)hipMalloc(&W, M * M * 4); (void)hipMalloc(&z, M * 4); (void)hipMalloc(&x, n * 4); (void)hipMalloc(&A, (size_t)M * n * 4); (void)hipMalloc(&ell, 4);
  std::vector<float> h(M * M, 0.01f), hz(M), hx(n);
  for (int i = 0; i < M; ++i) hz[i] = i * 0.5f;
  for (int i = 0; i < n; ++i) hx[i] = (i % 997) * 0.25f;
  float one = 1.f;
  (void)hipMemcpy(W, h.data(), M * M * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(z, hz.data(), M * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(ell, &one, 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hb_sgp_A_f32(0, x, 0, z, ell, 1, W, A, 1, n, M, 1, 0);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hb_sgp_A_f32(0, x, 0, z, ell, 1, W, A, 1, n, M, 1, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("hb_sgp_A_f32 (strip): %.2f us per launch (back-to-back stream launches)\n", ms * 1e3 / 50);
  long long st[64];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(hb_sstamps), sizeof(st));
  for (int w = 0; w < 8; ++w) {
    long long* s = st + w * 8;
    printf("wave %d: synth %6lld  phase0 (both tiles) %6lld  phase1 (deep tile) %6lld  epilogue %6lld   total %6lld cycles = %.2f us\n", w,
           s[1] - s[0], s[2] - s[1], s[3] - s[2], s[6] - s[5], s[6] - s[0], (s[6] - s[0]) / 2400.0);
  }
  return 0;
}
