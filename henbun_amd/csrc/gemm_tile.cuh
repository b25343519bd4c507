// Block-level MFMA tile GEMM engine (gfx950).
//
// A workgroup of WM x WN waves owns a BM x BN output tile; each wave owns a
// (BM/WM) x (BN/WN) sub-tile made of Mma<T> fragments (32x32x2 f32 or
// 16x16x4 f64 MFMA).  Operands are staged through LDS "k-major":
//   As[kk][m]  (m contiguous)   Bs[kk][n]  (n contiguous)
// so that a wave's fragment read (lane -> row/col = lane % T{M,N}, k = lane / T{M,N})
// is a conflict-free ds_read of consecutive addresses.  Global values for the
// next k-step are prefetched into registers while the current step's MFMAs
// run (f32 MFMA issues at 64 cycles per 32x32x2, so a 16-deep k-step gives
// ~2k cycles of cover per wave).
//
// Operand values come from caller-supplied functors fa(m, k) / fb(k, n)
// (tile-local m,n; global k) which do their own bounds handling; this lets
// kernels synthesise an operand on the fly (the RBF cross-covariance block in
// sgp.hip is never materialised).
#pragma once
#include "common.cuh"

template <typename T, int BM_, int BN_, int BK_, int WM_, int WN_>
struct TileGemm {
  typedef Mma<T> MM;
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int WTM = BM / WM, WTN = BN / WN;
  static constexpr int RM = WTM / MM::TM, RN = WTN / MM::TN;
  static constexpr int PAD = (sizeof(T) == 8) ? 16 : 4;
  static constexpr int LDA = BM + PAD, LDB = BN + PAD;
  static constexpr int EA = (BM * BK) / NT, EB = (BN * BK) / NT;
  static constexpr int LDS_ELEMS = BK * (LDA + LDB);
  static_assert(WTM % MM::TM == 0 && WTN % MM::TN == 0, "wave tile must be a multiple of the MFMA tile");
  static_assert((BM * BK) % NT == 0 && (BN * BK) % NT == 0, "fill must divide evenly");
  static_assert(BK % MM::TK == 0, "BK must be a multiple of the MFMA k");

  typename MM::Acc acc[RM][RN];

  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < MM::NACC; ++r) acc[i][j][r] = T(0);
  }

  __device__ __forceinline__ void mma_block(const T* __restrict__ As, const T* __restrict__ Bs) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w / WN, wn = w % WN;
    const int am = wm * WTM + (lane % MM::TM), ak = lane / MM::TM;
    const int bn = wn * WTN + (lane % MM::TN), bk = lane / MM::TN;
#pragma unroll
    for (int kk = 0; kk < BK; kk += MM::TK) {
      T a[RM], b[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = As[(kk + ak) * LDA + am + i * MM::TM];
#pragma unroll
      for (int j = 0; j < RN; ++j) b[j] = Bs[(kk + bk) * LDB + bn + j * MM::TN];
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = MM::mma(a[i], b[j], acc[i][j]);
    }
  }

  // acc += sum_{k in [kbeg,kend)} fa(m,k) * fb(k,n).  AKF/BKF: consecutive
  // threads walk k (true) or m/n (false) when filling -- pick whichever is
  // contiguous in the operand's memory.  Every thread of the block must call
  // this with the same [kbeg,kend).
  template <bool AKF, bool BKF, class FA, class FB>
  __device__ __forceinline__ void run(long kbeg, long kend, FA fa, FB fb, T* __restrict__ As, T* __restrict__ Bs) {
    if (kbeg >= kend) return;
    const int tid = threadIdx.x;
    T ra[EA], rb[EB];
    auto fetch = [&](long k0) {
#pragma unroll
      for (int e = 0; e < EA; ++e) {
        const int idx = e * NT + tid;
        const int m = AKF ? idx / BK : idx % BM;
        const int kk = AKF ? idx % BK : idx / BM;
        const long k = k0 + kk;
        ra[e] = k < kend ? fa(m, k) : T(0);
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int idx = e * NT + tid;
        const int n = BKF ? idx / BK : idx % BN;
        const int kk = BKF ? idx % BK : idx / BN;
        const long k = k0 + kk;
        rb[e] = k < kend ? fb(k, n) : T(0);
      }
    };
    fetch(kbeg);
    for (long k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
      for (int e = 0; e < EA; ++e) {
        const int idx = e * NT + tid;
        const int m = AKF ? idx / BK : idx % BM;
        const int kk = AKF ? idx % BK : idx / BM;
        As[kk * LDA + m] = ra[e];
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int idx = e * NT + tid;
        const int n = BKF ? idx / BK : idx % BN;
        const int kk = BKF ? idx % BK : idx / BN;
        Bs[kk * LDB + n] = rb[e];
      }
      __syncthreads();
      if (k0 + BK < kend) fetch(k0 + BK);
      mma_block(As, Bs);
      __syncthreads();
    }
  }

  // f(row, col, value) over this thread's accumulator elements (tile-local)
  template <class F>
  __device__ __forceinline__ void for_each(F f) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w / WN, wn = w % WN;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < MM::NACC; ++r) {
          const int row = wm * WTM + i * MM::TM + MM::acc_row(lane, r);
          const int col = wn * WTN + j * MM::TN + MM::acc_col(lane);
          const T val = acc[i][j][r];
          f(row, col, val);
        }
  }
};
