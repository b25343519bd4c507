// Diagnostic build of the persistent Cholesky + inverse (csrc/chol_persist.cuh) with per-wave s_memtime stamps (not part
// of the product): where a column block's workgroup spends its time -- waiting for chunks, updating, in-panel, stores.
// build: tools/build_chol_persist_stamps.sh (linalg.hip recompiled with -DHB_CP_STAMPS, the other objects of the library as built)
#include <hip/hip_runtime.h>
#include "../include/henbun_hip.h"
unsigned long long* hb_cp_stamps_buffer = nullptr;
#include <stdio.h>
#include <vector>
#include <cmath>
#include <algorithm>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 512;
  const int nb = M / 64, nwg = nb * nb;
  std::vector<float> h((size_t)M * M);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) h[(size_t)i * M + j] = expf(-0.5f * (i - j) * (i - j) * 0.25f) + (i == j ? 0.01f : 0.f);
  float *A, *L, *W, *ws;
  int* info;
  const long wse = hb_cholesky_inverse_ws_elems(1, M, 4);
  (void)hipMalloc(&A, (size_t)M * M * 4); (void)hipMalloc(&L, (size_t)M * M * 4); (void)hipMalloc(&W, (size_t)M * M * 4);
  (void)hipMalloc(&ws, wse * 4); (void)hipMemset(ws, 0, wse * 4);
  (void)hipMalloc(&info, 4);
  (void)hipMalloc(&hb_cp_stamps_buffer, (size_t)nwg * 8 * 64 * 8);
  (void)hipMemcpy(A, h.data(), (size_t)M * M * 4, hipMemcpyHostToDevice);
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 100; ++rep) hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hb_cholesky_inverse_f32 (stamped build), M = %d: %.1f us per call incl. the tril pass\n", M, ms * 10.0);
  }
  (void)hipMemset(hb_cp_stamps_buffer, 0, (size_t)nwg * 8 * 64 * 8);
  hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
  (void)hipDeviceSynchronize();
  int hinfo; (void)hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> st((size_t)nwg * 8 * 64);
  (void)hipMemcpy(st.data(), hb_cp_stamps_buffer, st.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (auto v : st) if (v) { t0 = std::min(t0, v); t1 = std::max(t1, v); }
  printf("info %d; first stamp -> last stamp %llu cycles (s_memtime ticks)\n", hinfo, t1 - t0);
  auto S = [&](int ticket, int w, int slot) -> long long {   // s_memtime differs between XCDs: relative to the workgroup's own start
    const unsigned long long v = st[((size_t)ticket * 8 + w) * 64 + slot], base = st[((size_t)ticket * 8 + 0) * 64 + 0];
    return v ? (long long)(v - base) : -1;
  };
  // wave ids: diag (0,q): w = q; strip (1,q): w = 4 + (3 - q)
  printf("critical workgroups (column block j, strip 0): cycles since the first stamp of the launch\n");
  for (int j = 0; j < nb; ++j) {
    const int t = j * nb;   // B = 1: ticket = j * nb + s when workgroups start in order
    printf(" j=%d start %6lld |", j, S(t, 0, 0));
    if (j > 0) {
      printf(" last panel chunks seen (diag wave q=0):");
      for (int c = 0; c < 4; ++c) printf(" %6lld", S(t, 0, 8 + 4 * ((j - 1) & 7) + c));
      printf(" | their MFMAs done:");
      for (int c = 0; c < 4; ++c) printf(" %6lld", S(t, 0, 40 + c));
    }
    printf("\n      diag waves: updates done / followed / own group done:");
    for (int q = 0; q < 4; ++q) printf("  q%d %6lld %6lld %6lld", q, S(t, q, 1), S(t, q, 2), S(t, q, 3));
    printf("\n      strip waves:                                        ");
    for (int q = 0; q < 4; ++q) { const int w = 4 + (3 - q); printf("  q%d %6lld %6lld %6lld", q, S(t, w, 1), S(t, w, 2), S(t, w, 3)); }
    printf("\n      outputs stored %6lld\n", S(t, 0, 4));
    if (j == 1) {
      for (int q = 0; q < 4; ++q) {
        const int w = 4 + (3 - q);
        printf("      pivot q%d publishes sub-groups at:", q);
        for (int sg = 0; sg < 4; ++sg) printf(" %6lld", S(t, q, 56 + sg));
        printf("   strip q%d per sub-group (before wait, after wait, published):", q);
        for (int sg = 0; sg < 4; ++sg) printf("  %6lld %6lld %6lld", S(t, w, 44 + 3 * sg), S(t, w, 45 + 3 * sg), S(t, w, 46 + 3 * sg));
        printf("\n");
      }
    }
  }
  return 0;
}
