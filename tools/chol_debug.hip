// Diagnostic: hb_cholesky_f32 / hb_cholesky_inverse_f32 against a host fp64 factorisation, with the position of the
// first mismatch (not part of the product).  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/chol_debug.hip -o tools/_bin/chol_debug -ldl
#include <hip/hip_runtime.h>
#include "../henbun_amd/csrc/runtime.hip"
#include "../henbun_amd/csrc/linalg.hip"
#include <stdio.h>
#include <vector>
#include <cmath>

__global__ void swap_probe(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned lo = lane, hi = lane;
  asm volatile("" : "+v"(hi));
  const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
}

int main() {
  {
    unsigned* d;
    (void)hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(swap_probe, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("permlane32_swap(v, v) with v = lane: r[0] lanes 0,1,31,32,33,63 = %u %u %u %u %u %u;  r[1] = %u %u %u %u %u %u\n", h[0], h[1], h[31],
           h[32], h[33], h[63], h[64], h[65], h[95], h[96], h[97], h[127]);
  }
  for (int M : {64, 128, 512}) {
    std::vector<float> h(M * M);
    std::vector<double> a(M * M), l(M * M, 0.0);
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < M; ++j) {
        h[i * M + j] = expf(-0.5f * (i - j) * (i - j) * 0.25f) + (i == j ? 0.05f : 0.f);
        a[i * M + j] = h[i * M + j];
      }
    for (int j = 0; j < M; ++j) {
      double s = a[j * M + j];
      for (int p = 0; p < j; ++p) s -= l[j * M + p] * l[j * M + p];
      l[j * M + j] = sqrt(s);
      for (int i = j + 1; i < M; ++i) {
        double t = a[i * M + j];
        for (int p = 0; p < j; ++p) t -= l[i * M + p] * l[j * M + p];
        l[i * M + j] = t / l[j * M + j];
      }
    }
    float *A, *L, *W, *ws;
    int* info;
    (void)hipMalloc(&A, M * M * 4); (void)hipMalloc(&L, M * M * 4); (void)hipMalloc(&W, M * M * 4); (void)hipMalloc(&ws, M * M * 4);
    (void)hipMalloc(&info, 4);
    (void)hipMemcpy(A, h.data(), M * M * 4, hipMemcpyHostToDevice);
    for (int inv = 0; inv < 2; ++inv)
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(L, 0xff, M * M * 4);
        if (inv) hb_cholesky_inverse_f32(A, L, W, 1, M, info, ws, nullptr, 0, 0);
        else hb_cholesky_f32(A, L, 1, M, info, 0);
        (void)hipDeviceSynchronize();
        std::vector<float> g(M * M);
        int hinfo;
        (void)hipMemcpy(g.data(), L, M * M * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
        double worst = 0;
        int wi = -1, wj = -1, fi = -1, fj = -1;
        for (int j = 0; j < M && fi < 0; ++j)
          for (int i = j; i < M; ++i) {
            const double e = fabs(g[i * M + j] - l[i * M + j]);
            if (!(e < 1e-3)) { fi = i; fj = j; break; }
          }
        for (int i = 0; i < M; ++i)
          for (int j = 0; j <= i; ++j) {
            const double e = fabs(g[i * M + j] - l[i * M + j]);
            if (!(e <= worst)) worst = e, wi = i, wj = j;
          }
        printf("M=%d inv=%d rep=%d info=%d worst |L - ref| = %.3e at (%d,%d); first bad entry in column order: (%d,%d) got %g want %g\n", M, inv, rep,
               hinfo, worst, wi, wj, fi, fj, fi >= 0 ? g[fi * M + fj] : 0.f, fi >= 0 ? l[fi * M + fj] : 0.0);
      }
  }
  return 0;
}
