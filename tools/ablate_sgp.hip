// Diagnostic: time ablated variants of the A = W K(z,x) tile loop (not part of the product).
#include "../henbun_amd/csrc/gemm_tile.cuh"
#include <stdio.h>
#include <vector>
void hb_set_error(const char*, ...) {}
#define BMc 64
#define BNc 128
template <int ABL, int DBG>
__global__ void __launch_bounds__(256) kern(const float* __restrict__ W, const float* __restrict__ z,
                                            const float* __restrict__ x, float* __restrict__ A, int M, int n) {
  typedef TileGemm<float, BMc, BNc, 16, 2, 2> G;
  __shared__ float lds[G::LDS_ELEMS + 4096];
  float* zs = lds + G::LDS_ELEMS;
  const int col0 = blockIdx.x * BNc;
  const int nRB = (M + BMc - 1) / BMc;
  const int jc = col0 + (threadIdx.x % BNc);
  const float xs = x[jc];
  for (int t = threadIdx.x; t < M; t += blockDim.x) zs[t] = z[t];
  __syncthreads();
  for (int half = 0; half < 2; ++half) {
    const int rb = half == 0 ? (int)blockIdx.y : nRB - 1 - (int)blockIdx.y;
    if (half == 1 && rb <= (int)blockIdx.y) break;
    const int row0 = rb * BMc;
    int kend = row0 + BMc;
    G g;
    g.zero();
    auto la = [&](int m, int k) -> float { return (ABL & 2) ? 1.0f : W[(row0 + m) * M + k]; };
    auto fa = [&](float raw, int m, int k) -> float { return (k <= row0 + m) ? raw : 0.f; };
    auto lb = [&](int k, int nn) -> float { return 0.f; };
    auto fb = [&](float raw, int k, int nn) -> float {
      if (ABL & 1) return zs[k] - xs;
      const float t = zs[k] - xs;
      return __expf(-0.5f * t * t);
    };
    g.template run<true, false, decltype(la), decltype(fa), decltype(lb), decltype(fb), DBG>(0, kend, la, fa, lb, fb, lds);
    g.for_each([&](int row, int col, float v) { A[(long)(row0 + row) * n + col0 + col] = v; });
  }
}
template <int ABL, int DBG>
void bench(const char* name, const float* W, const float* z, const float* x, float* A, int M, int n) {
  dim3 grid(n / BNc, (M / BMc + 1) / 2);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((kern<ABL, DBG>), grid, dim3(256), 0, 0, W, z, x, A, M, n);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((kern<ABL, DBG>), grid, dim3(256), 0, 0, W, z, x, A, M, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.1f us\n", name, ms * 1e3 / 50);
}
int main() {
  const int M = 512, n = 8192;
  float *W, *z, *x, *A;
  (void)hipMalloc(&W, M * M * 4); (void)hipMalloc(&z, M * 4); (void)hipMalloc(&x, n * 4); (void)hipMalloc(&A, (size_t)M * n * 4);
  std::vector<float> h(M * M, 0.01f), hz(M), hx(n);
  for (int i = 0; i < M; ++i) hz[i] = i * 0.5f;
  for (int i = 0; i < n; ++i) hx[i] = (i % 997) * 0.25f;
  (void)hipMemcpy(W, h.data(), M * M * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(z, hz.data(), M * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  bench<0, 0>("full", W, z, x, A, M, n);
  bench<1, 0>("no exp", W, z, x, A, M, n);
  bench<2, 0>("no W loads", W, z, x, A, M, n);
  bench<3, 0>("no exp, no W loads", W, z, x, A, M, n);
  bench<0, 1>("no LDS stash", W, z, x, A, M, n);
  bench<0, 2>("no MFMA", W, z, x, A, M, n);
  bench<0, 4>("no fragment reads", W, z, x, A, M, n);
  bench<3, 1>("no exp/W/stash (reads+MFMA only)", W, z, x, A, M, n);
  bench<3, 5>("MFMA only (no reads, no stash, no loads)", W, z, x, A, M, n);
  bench<3, 7>("nothing (loop skeleton + barriers)", W, z, x, A, M, n);
  return 0;
}
