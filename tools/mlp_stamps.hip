// Diagnostic build of the fused encoder's backward kernel (csrc/mlp.hip) with per-wave s_memtime stamps (not part of the
// product).  build: tools/build_mlp_stamps.sh
#include <hip/hip_runtime.h>
#include "../include/henbun_hip.h"
#include <stdio.h>
#include <vector>
unsigned long long* hb_mlp_stamps_buffer = nullptr;
int main() {
  const long n = 32768, din = 64, hid = 256;
  float *y, *w0, *b0, *w1, *o, *u, *x, *xbar, *klbar, *dw0, *db0, *dw1, *db1, *ws;
  auto al = [&](float** p, size_t e) { (void)hipMalloc(p, e * 4); (void)hipMemset(*p, 0, e * 4); };
  al(&y, n * din); al(&w0, din * hid); al(&b0, hid); al(&w1, hid * 32); al(&o, n * 32); al(&u, n * 16); al(&x, n * 16); al(&xbar, n * 16);
  al(&klbar, 1); al(&dw0, din * hid); al(&db0, hid); al(&dw1, hid * 32); al(&db1, 32); al(&ws, hb_mlp2_sample_ws_elems(n, din, hid));
  std::vector<float> hy(n * din);
  for (size_t i = 0; i < hy.size(); ++i) hy[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  (void)hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w0, hy.data(), din * hid * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w1, hy.data(), hid * 32 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(xbar, hy.data(), n * 16 * 4, hipMemcpyHostToDevice);
  const int nwg = 4 * 64;
  (void)hipMalloc(&hb_mlp_stamps_buffer, (size_t)nwg * 4 * 24 * 8);
  (void)hipMemset(hb_mlp_stamps_buffer, 0, (size_t)nwg * 4 * 24 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hb_mlp2_sample_bwd_f32(y, w0, b0, w1, HB_ACT_SIGMOID, o, u, x, xbar, klbar, dw0, db0, dw1, db1, n, din, hid, ws, 0);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hb_mlp2_sample_bwd_f32(y, w0, b0, w1, HB_ACT_SIGMOID, o, u, x, xbar, klbar, dw0, db0, dw1, db1, n, din, hid, ws, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("hb_mlp2_sample_bwd_f32 (stamped build, kernel + finish): %.1f us per call\n", ms * 20.0);
  std::vector<unsigned long long> st((size_t)nwg * 4 * 24);
  (void)hipMemcpy(st.data(), hb_mlp_stamps_buffer, st.size() * 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"start", "weights staged", "tile 1: top", "tile 1: staged (global loads + LDS writes)", "tile 1: (a) h done", "tile 1: (b) dW1 done",
                      "tile 1: (c) dh done", "tile 1: end (both T)", "all tiles done", "reduced + written"};
  for (int wg : {0, 100, 255}) {
    printf("workgroup %d, wave 0 / wave 3: cycles since the wave's first stamp\n", wg);
    for (int i = 0; i < 10; ++i)
      printf("   %-46s %8lld %8lld\n", nm[i], (long long)(st[((size_t)wg * 4 + 0) * 24 + i] - st[((size_t)wg * 4 + 0) * 24]),
             (long long)(st[((size_t)wg * 4 + 3) * 24 + i] - st[((size_t)wg * 4 + 3) * 24]));
  }
  return 0;
}
