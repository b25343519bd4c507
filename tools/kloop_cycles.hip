// Diagnostic: cycles per k-step of TileGemm::run_vec on cache-resident operands (not part of the product).
// hipcc -O3 --offload-arch=gfx950 -I. tools/kloop_cycles.hip -o /tmp/kloop && /tmp/kloop
#include "../henbun_amd/csrc/gemm_tile.cuh"
#include <stdio.h>
#include <vector>
void hb_set_error(const char*, ...) {}

template <int BM, int BN, int BK, int AMODE, int BMODE>
__global__ void __launch_bounds__(256) kern(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                            int lda, int ldb, int K, long long* __restrict__ cyc) {
  typedef TileGemm<float, BM, BN, BK, 2, 2> G;
  __shared__ float lds[G::LDS_ELEMS];
  typedef typename G::VT VT;
  G g;
  g.zero();
  const int row0 = (blockIdx.x % 4) * BM, col0 = (blockIdx.x / 4 % 4) * BN;
  auto la = [&](int m, int k) -> VT {
    return AMODE == HB_KC ? *reinterpret_cast<const VT*>(&A[(row0 + m) * lda + k]) : *reinterpret_cast<const VT*>(&A[k * lda + row0 + m]);
  };
  auto fa = [&](VT raw, int m, int k) -> VT { return raw; };
  auto lb = [&](int k, int n) -> VT {
    return BMODE == HB_KC ? *reinterpret_cast<const VT*>(&B[(col0 + n) * ldb + k]) : *reinterpret_cast<const VT*>(&B[k * ldb + col0 + n]);
  };
  auto fb = [&](VT raw, int k, int n) -> VT { return raw; };
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  g.template run_vec<AMODE, BMODE>(0, K, la, fa, lb, fb, lds);
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  float s = 0.f;
  g.for_each([&](int row, int col, float v) { s += v; });
  if (s == 123.456f) C[threadIdx.x] = s;
  if (threadIdx.x == 0) {
    cyc[2 * blockIdx.x] = c1 - c0;
    cyc[2 * blockIdx.x + 1] = t1 - t0;
  }
}

template <int BM, int BN, int BK, int AMODE, int BMODE>
void bench(const char* name, const float* A, const float* B, float* C, long long* cyc, int grid) {
  const int K = 512;  // operands [512 x 512]: 1 MB each, L2 resident after the first pass
  typedef TileGemm<float, BM, BN, BK, 2, 2> G;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kern<BM, BN, BK, AMODE, BMODE>), dim3(grid), dim3(256), 0, 0, A, B, C, 512, 512, K, cyc);
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((kern<BM, BN, BK, AMODE, BMODE>), dim3(grid), dim3(256), 0, 0, A, B, C, 512, 512, K, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double kernel_us = ms * 1e3 / 10;
  std::vector<long long> h(2 * grid);
  (void)hipMemcpy(h.data(), cyc, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost);
  double sc = 0, st = 0;
  for (int i = 0; i < grid; ++i) { sc += h[2 * i]; st += h[2 * i + 1]; }
  const int steps = K / BK;
  const double mfma = (double)G::NS * G::RM * G::RN * 64.0;
  const double chip_util = (double)grid * 4 * steps * mfma / (1024.0 * kernel_us * 2400.0);
  printf("%-34s grid %4d: %8.0f clk/k-step per WG (ideal %5.0f -> %4.1f%%), kernel %6.2f us, chip MFMA utilisation %4.1f%%\n", name, grid,
         sc / grid / steps, mfma, 100.0 * mfma / (sc / grid / steps), kernel_us, 100.0 * chip_util);
}

int main() {
  float *A, *B, *C;
  long long* cyc;
  (void)hipMalloc(&A, 512 * 512 * 4); (void)hipMalloc(&B, 512 * 512 * 4); (void)hipMalloc(&C, 1 << 20); (void)hipMalloc(&cyc, 16 * 4096);
  (void)hipMemset(A, 0, 512 * 512 * 4); (void)hipMemset(B, 0, 512 * 512 * 4);
  for (int grid : {256, 512, 1024}) {
    bench<64, 128, 16, HB_KC, HB_KC>("64x128 BK16 KC/KC (sgp_A shape)", A, B, C, cyc, grid);
    bench<64, 128, 16, HB_KC, HB_MC>("64x128 BK16 KC/MC", A, B, C, cyc, grid);
    bench<64, 128, 16, HB_MC, HB_MC>("64x128 BK16 MC/MC (kbar shape)", A, B, C, cyc, grid);
    bench<128, 128, 16, HB_KC, HB_KC>("128x128 BK16 KC/KC (Lbar shape)", A, B, C, cyc, grid);
    bench<128, 128, 32, HB_KC, HB_KC>("128x128 BK32 KC/KC", A, B, C, cyc, grid);
    bench<64, 64, 16, HB_KC, HB_KC>("64x64 BK16 KC/KC", A, B, C, cyc, grid);
    bench<128, 128, 16, HB_MC, HB_MC>("128x128 BK16 MC/MC", A, B, C, cyc, grid);
  }
  return 0;
}
