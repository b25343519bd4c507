// Diagnostic build of the column-strip contraction (second form, fragment-major W) with in-kernel phase stamps for
// EVERY workgroup and wave (not part of the product): hipcc -O3 --offload-arch=gfx950 tools/strip_stamps.hip -o /tmp/ss
#include <hip/hip_runtime.h>
__device__ long long hb_sstamps[256 * 8 * 4];
__device__ long long hb_srt[256 * 2];
#define HB_SSTAMP(i)                                                                                         \
  do {                                                                                                       \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) {                                                       \
      hb_sstamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + (i)] = clock64();                               \
      if (threadIdx.x == 0 && ((i) == 0 || (i) == 3)) hb_srt[blockIdx.x * 2 + ((i) == 3)] = wall_clock64(); \
    }                                                                                                        \
  } while (0)
#include "../henbun_amd/csrc/runtime.hip"
#include "../henbun_amd/csrc/elementwise.hip"
#include "../henbun_amd/csrc/gram.hip"
#include "../henbun_amd/csrc/linalg.hip"
#include "../henbun_amd/csrc/sgp.hip"
// (the serial-chain recorder lives in csrc/jit.hip, which this diagnostic build leaves out: nothing is ever recording)
bool hb_chain_recording() { return false; }
int hb_chain_push(const HbChainJob&, hipStream_t) { return 0; }
int hb_chain_flush(hipStream_t) { return 0; }
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
// the in-step form: hb_sgp_fwd with the fragment-major A only (no row-major A), statistics and finish
static float *g_u, *g_eps, *g_Af, *g_f, *g_v, *g_ws;
#define RUN() hb_sgp_fwd_f32(0, 1, x, 0, z, ell, 1, W, Wf, 0, g_u, g_eps, nullptr, 0, nullptr, nullptr, g_Af, g_f, g_v, 1, n, M, 1, 1, g_ws, 0)
int main(int argc, char** argv) {
  const int form2 = argc > 1 && atoi(argv[1]) == 2;   // default: the third strip form (sgp_A_strip2t_kernel, finishing pass inside)
  hb_debug_set("sgp_strip_form2", form2);
  const int M = 512, n = 8192;
  float *K, *L, *W, *ws, *Wf, *z, *x, *A, *ell;
  int* info;
  const long wse = hb_cholesky_inverse_ws_elems(1, M, 4);   // exchange area + sync words of the persistent factorisation, zero at entry
  (void)hipMalloc(&K, M * M * 4); (void)hipMalloc(&L, M * M * 4); (void)hipMalloc(&W, M * M * 4); (void)hipMalloc(&ws, wse * 4);
  (void)hipMemset(ws, 0, wse * 4);
  (void)hipMalloc(&Wf, 2 * M * M * 4); (void)hipMalloc(&info, 4);
  (void)hipMalloc(&z, M * 4); (void)hipMalloc(&x, n * 4); (void)hipMalloc(&A, (size_t)M * n * 4); (void)hipMalloc(&ell, 4);
  (void)hipMalloc(&g_u, M * 4); (void)hipMalloc(&g_eps, n * 4); (void)hipMalloc(&g_Af, (size_t)M * n * 4); (void)hipMalloc(&g_f, n * 4);
  (void)hipMalloc(&g_v, n * 4); (void)hipMalloc(&g_ws, (size_t)(n + M + 32L * M * M) * 4);
  (void)hipMemset(g_u, 0, M * 4); (void)hipMemset(g_eps, 0, n * 4);
  std::vector<float> hz(M), hx(n);
  for (int i = 0; i < M; ++i) hz[i] = i * 0.5f;
  for (int i = 0; i < n; ++i) hx[i] = (i % 997) * 0.25f;
  float one = 1.f;
  (void)hipMemcpy(z, hz.data(), M * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(ell, &one, 4, hipMemcpyHostToDevice);
  if (hb_gram_fwd_f32(0, z, 0, z, 0, ell, 0, 1, K, 1, M, M, 1, 1e-3, 0)) { printf("gram: %s\n", hb_last_error_string()); return 1; }
  if (hb_cholesky_inverse_f32(K, L, W, 1, M, info, ws, Wf, 0, 0)) { printf("chol: %s\n", hb_last_error_string()); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) RUN();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) RUN();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  if (RUN()) { printf("fwd: %s\n", hb_last_error_string()); return 1; }
  printf("hb_sgp_fwd_f32 (%s strip form, fragment-major W and A): %.2f us per launch (back-to-back stream launches)\n", form2 ? "second" : "third", ms * 1e3 / 50);
  std::vector<long long> st(256 * 8 * 4), rt(512);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(hb_sstamps), st.size() * 8);
  (void)hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(hb_srt), rt.size() * 8);
  long long t0 = rt[0], t1 = rt[1];
  for (int b = 0; b < 256; ++b) { t0 = std::min(t0, rt[2 * b]); t1 = std::max(t1, rt[2 * b + 1]); }
  printf("last launch: first workgroup start -> last workgroup end %.2f us (100 MHz wall clock)\n", (t1 - t0) / 100.0);
  std::vector<double> starts, durs;
  for (int b = 0; b < 256; ++b) { starts.push_back((rt[2 * b] - t0) / 100.0); durs.push_back((rt[2 * b + 1] - rt[2 * b]) / 100.0); }
  std::sort(starts.begin(), starts.end()); std::sort(durs.begin(), durs.end());
  printf("workgroup start offsets (us): min %.2f median %.2f p90 %.2f max %.2f ; durations: min %.2f median %.2f p90 %.2f max %.2f\n",
         starts[0], starts[128], starts[230], starts[255], durs[0], durs[128], durs[230], durs[255]);
  for (int b : {0, 7, 128, 255})
    for (int w = 0; w < 8; ++w) {
      long long* s = &st[(b * 8 + w) * 4];
      printf("wg %3d wave %d: synth %6lld  loop %6lld  epilogue %6lld  total %6lld cycles\n", b, w, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[3] - s[0]);
    }
  return 0;
}
