// Diagnostic build of the Cholesky panel kernel with phase stamps (not part of the product).
#include <hip/hip_runtime.h>
__device__ long long g_stamps[8];
#define HB_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_stamps[i] = clock64(); } while (0)
#include "../henbun_amd/csrc/linalg.hip"
#include <stdio.h>
#include <vector>
#include <cmath>
void hb_set_error(const char*, ...) {}
int main() {
  const int M = 512;
  std::vector<float> h((size_t)M * M);
  for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) { float d = 0.5f * (i - j); h[(size_t)i * M + j] = expf(-0.5f * d * d) + (i == j ? 1e-2f : 0.f); }
  float *A, *L; int* info;
  (void)hipMalloc(&A, (size_t)M * M * 4); (void)hipMalloc(&L, (size_t)M * M * 4); (void)hipMalloc(&info, 4);
  (void)hipMemcpy(A, h.data(), (size_t)M * M * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) hb_cholesky_f32(A, L, 1, M, info, 0);
  (void)hipDeviceSynchronize();
  for (long j0 : {0L, 128L, 256L, 448L}) {
    // replay the factorisation up to j0, then stamp panel j0
    const long below = M - j0 - CH_NB;
    const int gx = below > 0 ? (int)((below + CH_RB - 1) / CH_RB) : 1;
    hipLaunchKernelGGL(chol_panel_kernel<float, true>, dim3(gx, 1), dim3(256), 0, 0, A, L, (long)M, j0, info);
    (void)hipDeviceSynchronize();
    long long st[8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    printf("j0=%3ld gx=%d  gemm %6lld  epilogue %6lld  potrf %6lld  solve %6lld  store %6lld  cycles (total %lld = %.1f us @2.4GHz)\n", j0, gx,
           st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[5] - st[0], (st[5] - st[0]) / 2400.0);
  }
  int hi; (void)hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost); printf("info %d\n", hi);
  return 0;
}
