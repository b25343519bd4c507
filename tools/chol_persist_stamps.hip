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
  (void)hipMalloc(&hb_cp_stamps_buffer, ((size_t)nwg * 8 * 16 + (size_t)nwg * 8) * 8);
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
  const size_t nst = (size_t)nwg * 8 * 16 + (size_t)nwg * 8;
  (void)hipMemset(hb_cp_stamps_buffer, 0, nst * 8);
  hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
  (void)hipDeviceSynchronize();
  int hinfo; (void)hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> st(nst);
  (void)hipMemcpy(st.data(), hb_cp_stamps_buffer, st.size() * 8, hipMemcpyDeviceToHost);
  printf("info %d.  s_memtime differs between XCDs: every number is cycles since the workgroup's own first stamp.\n", hinfo);
  auto S = [&](int ticket, int w, int slot) -> long long {
    const unsigned long long v = st[((size_t)ticket * 8 + w) * 16 + slot], base = st[((size_t)ticket * 8 + 0) * 16 + 0];
    return v ? (long long)(v - base) : -1;
  };
  // wave ids: diag (0,q): w = q; strip (1,q): w = 4 + (3 - q)
  printf("critical workgroups (column block j, strip 0; B = 1: ticket j * nb when workgroups start in order)\n");
  for (int j = 0; j < nb; ++j) {
    const int t = j * nb;
    printf(" j=%d: last chunk of panel j-1 seen %6lld, its MFMAs done %6lld, updates done %6lld (diag wave q0)\n", j, S(t, 0, 5), S(t, 0, 6), S(t, 0, 1));
    for (int q = 0; q < 4; ++q)
      printf("      diag  q%d: followed %6lld  row-per-lane %6lld  sub-groups published %6lld %6lld %6lld %6lld\n", q, S(t, q, 2), S(t, q, 11),
             S(t, q, 7), S(t, q, 8), S(t, q, 9), S(t, q, 10));
    for (int q = 0; q < 4; ++q) {
      const int w = 4 + (3 - q);
      printf("      strip q%d: followed %6lld  row-per-lane %6lld  sub-groups published %6lld %6lld %6lld %6lld  chunk drained %6lld  flag %6lld\n", q,
             S(t, w, 2), S(t, w, 11), S(t, w, 7), S(t, w, 8), S(t, w, 9), S(t, w, 10), S(t, w, 12), S(t, w, 3));
    }
    printf("      outputs stored %6lld\n", S(t, 0, 4));
  }
  return 0;
}
