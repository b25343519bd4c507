// Diagnostic build of the Cholesky kernel with in-kernel phase stamps (not part of the product).
// hipcc -O3 --offload-arch=gfx950 tools/chol_stamps.hip -o /tmp/chol_stamps && /tmp/chol_stamps
#include <hip/hip_runtime.h>
__device__ long long hb_stamps[64 * 8];
__device__ int hb_stamp_k;
#define HB_STAMP(i)                                                                       \
  do {                                                                                    \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hb_stamps[k * 8 + (i)] = clock64(); \
  } while (0)
__device__ long long hb_wstamps[8 * 4 * 20];
#define HB_WSTAMP(w, lane, i)                                                                     \
  do {                                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (lane) == 0) hb_wstamps[(k * 4 + (w)) * 20 + (i)] = clock64(); \
  } while (0)
__device__ long long hb_pstamps[8 * 8];
#define HB_PSTAMP(i)                                                                                   \
  do {                                                                                                 \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hb_pstamps[k * 8 + (i)] = clock64(); \
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
      printf("  k=%2d  load+rank64 %6lld  in-panel(8 steps, independent waves) %6lld  store %6lld   total %6lld = %.2f us\n", k,
             s[1] - s[0], s[2] - s[1], s[3] - s[2], s[3] - s[0], (s[3] - s[0]) / 2400.0);

    }
    if (inv) {
      long long ws_[8 * 4 * 20];
      (void)hipMemcpyFromSymbol(ws_, HIP_SYMBOL(hb_wstamps), sizeof(ws_));
      long long ps_[64];
      (void)hipMemcpyFromSymbol(ps_, HIP_SYMBOL(hb_pstamps), sizeof(ps_));
      for (int k : {0, 1, 5})
        printf("  launch k=%d, wave 0: before loads %lld, tile loads issued %lld, B loads issued %lld |", k, ps_[k * 8 + 4] - st[k * 8], ps_[k * 8 + 5] - st[k * 8], ps_[k * 8 + 6] - st[k * 8]),
        printf("  launch k=%d, wave 0 of factor workgroup 0, cycles since kernel entry: loads issued %lld, tile in registers %lld, staged in LDS %lld, past the barrier %lld, update done %lld\n",
               k, ps_[k * 8 + 0] - st[k * 8], ps_[k * 8 + 1] - st[k * 8], ps_[k * 8 + 2] - st[k * 8], ps_[k * 8 + 3] - st[k * 8], st[k * 8 + 1] - st[k * 8]);
      for (int k : {1, 5}) {
        const long long t0 = st[k * 8 + 1];   // HB_STAMP(1): end of load + rank-64 update of wave 0
        printf("  launch k=%d, factor workgroup 0: per wave, cycles since wave 0 left the update phase: start | per step (solve done, update done)\n", k);
        for (int w = 0; w < 4; ++w) {
          long long* q = ws_ + (k * 4 + w) * 20;
          printf("    wave %d: %6lld |", w, q[0] - t0);
          for (int kb = 0; kb < (w == 0 ? 4 : 8); ++kb) printf(" %5lld,%5lld", q[2 * kb + 1] - t0, q[2 * kb + 2] - t0);
          printf("\n");
        }
      }
    }
  }
  return 0;
}
