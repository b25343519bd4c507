// Diagnostic build of the Cholesky chain with forward riders: in-kernel phase stamps of the rider workgroups of
// every launch (not part of the product).  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/rider_stamps.hip -o tools/_bin/rs -ldl
#include <hip/hip_runtime.h>
__device__ long long hb_rst[9 * 256 * 8 * 6];   // [row block][strip][wave][stamp]
__device__ long long hb_rwall[9 * 256 * 2];
#define HB_RSTAMP(i)                                                                                   \
  do {                                                                                                 \
    if ((threadIdx.x & 63) == 0 && strip < 256) {                                                      \
      hb_rst[((rb * 256 + strip) * 8 + (threadIdx.x >> 6)) * 6 + (i)] = clock64();                     \
      if (threadIdx.x == 0 && ((i) == 0 || (i) == 5)) hb_rwall[(rb * 256 + strip) * 2 + ((i) == 5)] = wall_clock64(); \
    }                                                                                                  \
  } while (0)
#include "../henbun_amd/csrc/runtime.hip"
#include "../henbun_amd/csrc/elementwise.hip"
#include "../henbun_amd/csrc/gram.hip"
#include "../henbun_amd/csrc/linalg.hip"
#include <algorithm>
#include <stdio.h>
#include <vector>
int main() {
  const int M = 512, n = 8192;
  float *K, *L, *W, *ws, *Wf, *z, *x, *ell, *u, *Af, *sws;
  int* info;
  (void)hipMalloc(&K, M * M * 4); (void)hipMalloc(&L, M * M * 4); (void)hipMalloc(&W, M * M * 4); (void)hipMalloc(&ws, M * M * 4);
  (void)hipMalloc(&Wf, 2 * M * M * 4); (void)hipMalloc(&info, 4);
  (void)hipMalloc(&z, M * 4); (void)hipMalloc(&x, n * 4); (void)hipMalloc(&ell, 4); (void)hipMalloc(&u, M * 4);
  (void)hipMalloc(&Af, (size_t)M * n * 4); (void)hipMalloc(&sws, (size_t)(n + M + 5 * 8 * n) * 4);
  std::vector<float> hz(M), hx(n);
  for (int i = 0; i < M; ++i) hz[i] = i * 0.5f;
  for (int i = 0; i < n; ++i) hx[i] = (i % 997) * 0.25f;
  float one = 1.f;
  (void)hipMemcpy(z, hz.data(), M * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(ell, &one, 4, hipMemcpyHostToDevice);
  (void)hipMemset(u, 0, M * 4);
  if (hb_gram_fwd_f32(0, z, 0, z, 0, ell, 0, 1, K, 1, M, M, 1, 1e-3, 0)) { printf("gram: %s\n", hb_last_error_string()); return 1; }
  for (int rep = 0; rep < 3; ++rep)
    if (hb_cholesky_inverse_sgp_f32(K, L, W, M, info, ws, Wf, 0, x, z, ell, 1, u, n, 1, 1, Af, sws, 0)) { printf("chol: %s\n", hb_last_error_string()); return 1; }
  (void)hipDeviceSynchronize();
  std::vector<long long> st(9 * 256 * 8 * 6), wl(9 * 256 * 2);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(hb_rst), st.size() * 8);
  (void)hipMemcpyFromSymbol(wl.data(), HIP_SYMBOL(hb_rwall), wl.size() * 8);
  for (int rb = 0; rb < 8; ++rb) {
    long long t0 = wl[(rb * 256) * 2], t1 = 0;
    std::vector<double> starts, durs;
    for (int s = 0; s < 256; ++s) { t0 = std::min(t0, wl[(rb * 256 + s) * 2]); t1 = std::max(t1, wl[(rb * 256 + s) * 2 + 1]); }
    for (int s = 0; s < 256; ++s) { starts.push_back((wl[(rb * 256 + s) * 2] - t0) / 100.0); durs.push_back((wl[(rb * 256 + s) * 2 + 1] - wl[(rb * 256 + s) * 2]) / 100.0); }
    std::sort(starts.begin(), starts.end()); std::sort(durs.begin(), durs.end());
    printf("row block %d: riders span %.2f us; start offsets median %.2f p90 %.2f max %.2f; durations min %.2f median %.2f max %.2f us\n", rb,
           (t1 - t0) / 100.0, starts[128], starts[230], starts[255], durs[0], durs[128], durs[255]);
    for (int s : {0, 200}) {
      for (int w : {0, 7}) {
        long long* q = &st[((rb * 256 + s) * 8 + w) * 6];
        printf("   strip %3d wave %d: load issue %6lld  synth %6lld  mfma %6lld  exchange+retire %6lld  tail %6lld cycles\n", s, w, q[1] - q[0], q[2] - q[1],
               q[3] - q[2], q[4] - q[3], q[5] - q[4]);
      }
    }
  }
  return 0;
}
