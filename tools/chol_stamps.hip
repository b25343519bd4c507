// Diagnostic build of the Cholesky kernel with in-kernel phase stamps (not part of the product).
// hipcc -O3 --offload-arch=gfx950 tools/chol_stamps.hip -o /tmp/chol_stamps && /tmp/chol_stamps
#include <hip/hip_runtime.h>
__device__ long long hb_stamps[64 * 8];
__device__ int hb_stamp_k;
#define HB_STAMP(i)                                                                       \
  do {                                                                                    \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hb_stamps[k * 8 + (i)] = clock64(); \
  } while (0)
#include "../henbun_amd/csrc/runtime.hip"
#include "../henbun_amd/csrc/linalg.hip"
#include <stdio.h>
#include <vector>
#include <cmath>

int main() {
  const int M = 512;
  std::vector<float> h(M * M);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) h[i * M + j] = expf(-0.5f * (i - j) * (i - j) * 0.25f) + (i == j ? 0.01f : 0.f);
  float *A, *L, *W, *ws;
  int* info;
  (void)hipMalloc(&A, M * M * 4); (void)hipMalloc(&L, M * M * 4); (void)hipMalloc(&W, M * M * 4); (void)hipMalloc(&ws, M * M * 4);
  (void)hipMalloc(&info, 4);
  (void)hipMemcpy(A, h.data(), M * M * 4, hipMemcpyHostToDevice);
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 100; ++rep) hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hb_cholesky_inverse_f32, M = %d: %.1f us per call (back-to-back stream launches)\n", M, ms * 10.0);
  }
  for (int inv = 0; inv < 2; ++inv) {
    for (int rep = 0; rep < 3; ++rep) {
      if (inv) hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
      else hb_cholesky_f32(A, L, 1, M, info, 0);
    }
    (void)hipDeviceSynchronize();
    long long st[64 * 8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(hb_stamps), sizeof(st));
    int hinfo;
    (void)hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
    printf("%s (info %d): cycles of factor workgroup 0, per launch k\n", inv ? "cholesky+inverse" : "cholesky", hinfo);
    for (int k : {0, 1, 3, 5, 7}) {
      long long* s = st + k * 8;
      printf("  k=%2d  load+rank32 %6lld  in-panel(4x publish/potrf8/solve/rank8) %6lld  store %6lld   total %6lld = %.2f us\n", k,
             s[1] - s[0], s[2] - s[1], s[3] - s[2], s[3] - s[0], (s[3] - s[0]) / 2400.0);

    }
  }
  return 0;
}
