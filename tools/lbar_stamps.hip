// Diagnostic build of the fragment-major Lbar contraction with per-wave stamps (not part of the product).
#include <hip/hip_runtime.h>
__device__ long long hb_lst[1024 * 4 * 4];
__device__ long long hb_lrt[1024 * 2];
__device__ int hb_lcu[1024];
#define HB_LSTAMP(i)                                                                                          \
  do {                                                                                                        \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {                                                       \
      hb_lst[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (i)] = clock64();                                    \
      if (threadIdx.x == 0 && ((i) == 0 || (i) == 2)) hb_lrt[blockIdx.x * 2 + ((i) == 2)] = wall_clock64();  \
      if (threadIdx.x == 0 && (i) == 0) hb_lcu[blockIdx.x] = __smid();                                        \
    }                                                                                                         \
  } while (0)
#define HB_SSTAMP(i)
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
#include <map>
#include <stdio.h>
#include <vector>
int main() {
  const int M = 512, n = 8192, nS = n / 32;
  float *Kf, *Af, *slabs, *Lbar;
  (void)hipMalloc(&Kf, (size_t)M * n * 4); (void)hipMalloc(&Af, (size_t)M * n * 4);
  (void)hipMalloc(&slabs, (size_t)32 * M * M * 4); (void)hipMalloc(&Lbar, M * M * 4);
  std::vector<float> h((size_t)M * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  (void)hipMemcpy(Kf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(Af, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int S_used = 0;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) sgp_lbar_frag_launch(Kf, Af, slabs, 32, 1, M, nS, 0, &S_used, 0);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) sgp_lbar_frag_launch(Kf, Af, slabs, 32, 1, M, nS, 0, &S_used, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("lbar (kernel + finish): %.2f us per call\n", ms * 1e3 / 50);
  std::vector<long long> st(1024 * 16), rt(2048);
  std::vector<int> cu(1024);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(hb_lst), st.size() * 8);
  (void)hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(hb_lrt), rt.size() * 8);
  (void)hipMemcpyFromSymbol(cu.data(), HIP_SYMBOL(hb_lcu), cu.size() * 4);
  const int nwg = 504;
  long long t0 = rt[0], t1 = rt[1];
  for (int b = 0; b < nwg; ++b) { t0 = std::min(t0, rt[2 * b]); t1 = std::max(t1, rt[2 * b + 1]); }
  printf("first workgroup start -> last end: %.2f us\n", (t1 - t0) / 100.0);
  std::vector<double> starts, durs;
  std::map<int, int> percu;
  for (int b = 0; b < nwg; ++b) { starts.push_back((rt[2 * b] - t0) / 100.0); durs.push_back((rt[2 * b + 1] - rt[2 * b]) / 100.0); percu[cu[b]]++; }
  std::sort(starts.begin(), starts.end()); std::sort(durs.begin(), durs.end());
  printf("start offsets us: min %.2f med %.2f p90 %.2f max %.2f ; durations us: min %.2f med %.2f p90 %.2f max %.2f\n", starts[0],
         starts[nwg / 2], starts[nwg * 9 / 10], starts[nwg - 1], durs[0], durs[nwg / 2], durs[nwg * 9 / 10], durs[nwg - 1]);
  std::map<int, int> hist;
  for (auto& kv : percu) hist[kv.second]++;
  printf("distinct SM ids %zu; workgroups per id histogram:", percu.size());
  for (auto& kv : hist) printf(" %d wg x %d ids;", kv.first, kv.second);
  printf("\n");
  for (int b : {0, 100, 300, 503})
    for (int w = 0; w < 4; ++w) {
      long long* s = &st[(b * 4 + w) * 4];
      printf("wg %3d wave %d: loop %6lld  reduce+store %6lld cycles\n", b, w, s[1] - s[0], s[2] - s[1]);
    }
  return 0;
}
