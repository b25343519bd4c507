// Block-level MFMA tile GEMM engine (gfx950).
//
// A workgroup of WM x WN waves owns a BM x BN output tile; each wave owns a
// (BM/WM) x (BN/WN) sub-tile made of Mma<T> fragments (32x32x2 f32 or
// 16x16x4 f64 MFMA).  Operands are staged through LDS "k-major":
//   As[kk][m]  (m contiguous)   Bs[kk][n]  (n contiguous)
// so that a wave's fragment read (lane -> row/col = lane % T{M,N}, k = lane / T{M,N})
// is a conflict-free ds_read of consecutive addresses.
//
// Pipeline (one wave per SIMD is the normal residency for these kernels, so
// nothing but the wave's own instruction stream can hide latency):
//   * LDS is double-buffered; ONE raw s_barrier per k-step;
//   * the global load for an operand element of k-step t+2 is issued right
//     after the same element of step t+1 has been stashed, and is consumed a
//     full iteration later (>= 1k cycles of MFMA cover);
//   * the mask/transform + ds_write of step t+1's operands is cut into slices
//     that are placed BETWEEN the MFMA groups of step t, so the VALU/LDS work
//     issues in the shadow of the 64-cycle f32 MFMAs.
//
// Operand values come from caller-supplied functor pairs (see run()): a raw
// loader that only issues address-clamped, UNCONDITIONAL global loads, and a
// finisher that masks / transforms the raw value when it is stashed to LDS.
// (A predicated load costs a branch plus an s_waitcnt vmcnt(0) per element and
// serialises the whole fill.)  This also lets kernels synthesise an operand on
// the fly (the RBF cross-covariance block in sgp.hip is never materialised).
// All indices are 32-bit: callers guarantee every operand matrix has fewer
// than 2^31 elements.
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
  static constexpr int NE = EA + EB;
  static constexpr int NS = BK / MM::TK;              // MFMA sub-steps per k-step
  static constexpr int BUF_ELEMS = BK * (LDA + LDB);  // one LDS buffer
  static constexpr int LDS_ELEMS = 2 * BUF_ELEMS;     // double buffered
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

  // acc += sum_{k in [kbeg,kend)} a(m,k) * b(k,n), operands produced in two phases:
  //   raw = la(m, k) / lb(k, n)           address-clamped global loads only;
  //   val = fa(raw, m, k) / fb(raw, k, n) masking / arithmetic at stash time.
  // k passed to the functors is always inside [kbeg,kend); out-of-range k is
  // zero-filled here.  AKF/BKF: consecutive threads walk k (true) or m/n
  // (false) when filling -- pick whichever is contiguous in memory.  Every
  // thread of the block must call this with the same [kbeg,kend).  `lds`
  // needs LDS_ELEMS elements.
  // DBG (diagnostic builds only, tools/ablate_sgp.hip): 1 = skip LDS stash writes,
  // 2 = skip MFMAs, 4 = skip fragment reads.
  template <bool AKF, bool BKF, class LA, class FA, class LB, class FB, int DBG = 0>
  __device__ __forceinline__ void run(int kbeg, int kend, LA la, FA fa, LB lb, FB fb, T* __restrict__ lds) {
    if (kbeg >= kend) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    const int am = wm * WTM + (lane % MM::TM), ak = lane / MM::TM;
    const int bn = wn * WTN + (lane % MM::TN), bk = lane / MM::TN;
    typedef decltype(la(0, 0)) RawA;
    typedef decltype(lb(0, 0)) RawB;
    RawA ra[EA];
    RawB rb[EB];

    // element e of this thread -> (m or n, kk) inside the tile
    auto a_m = [&](int e) { const int idx = e * NT + tid; return AKF ? idx / BK : idx % BM; };
    auto a_kk = [&](int e) { const int idx = e * NT + tid; return AKF ? idx % BK : idx / BM; };
    auto b_n = [&](int e) { const int idx = e * NT + tid; return BKF ? idx / BK : idx % BN; };
    auto b_kk = [&](int e) { const int idx = e * NT + tid; return BKF ? idx % BK : idx / BN; };

    // issue the (clamped, unconditional) global load of element e of the tile at k0
    auto fetch_one = [&](int e, int k0) {
      if (e < EA) {
        const int k = k0 + a_kk(e);
        ra[e < EA ? e : 0] = la(a_m(e), k < kend ? k : kend - 1);
      } else {
        const int eb = e - EA;
        const int k = k0 + b_kk(eb);
        rb[eb >= 0 ? eb : 0] = lb(k < kend ? k : kend - 1, b_n(eb));
      }
    };
    // finish + stash element `e` (A elements first, then B) of the tile at k0 into buffer `buf`
    auto stash_one = [&](int e, int k0, T* buf) {
      if (e < EA) {
        const int m = a_m(e), kk = a_kk(e), k = k0 + kk;
        const T val = fa(ra[e < EA ? e : 0], m, k < kend ? k : kend - 1);
        if (DBG & 1) {
          asm volatile("" ::"v"(val));
        } else {
          buf[kk * LDA + m] = k < kend ? val : T(0);
        }
      } else {
        const int eb = e - EA;
        const int n = b_n(eb), kk = b_kk(eb), k = k0 + kk;
        const T val = fb(rb[eb >= 0 ? eb : 0], k < kend ? k : kend - 1, n);
        if (DBG & 1) {
          asm volatile("" ::"v"(val));
        } else {
          buf[BK * LDA + kk * LDB + n] = k < kend ? val : T(0);
        }
      }
    };

    // prologue: tile kbeg -> buffer 0; raw of tile kbeg+BK in flight
#pragma unroll
    for (int e = 0; e < NE; ++e) fetch_one(e, kbeg);
#pragma unroll
    for (int e = 0; e < NE; ++e) stash_one(e, kbeg, lds);
#pragma unroll
    for (int e = 0; e < NE; ++e) fetch_one(e, kbeg + BK);
    __syncthreads();

    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const T* __restrict__ As = lds + cur * BUF_ELEMS;
      const T* __restrict__ Bs = As + BK * LDA;
      T* __restrict__ nxt = lds + (cur ^ 1) * BUF_ELEMS;
      // fragments of the whole k-step
      T a[NS][RM], b[NS][RN];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < RM; ++i) a[s][i] = (DBG & 4) ? T(s + i) : As[(s * MM::TK + ak) * LDA + am + i * MM::TM];
#pragma unroll
        for (int j = 0; j < RN; ++j) b[s][j] = (DBG & 4) ? T(s - j) : Bs[(s * MM::TK + bk) * LDB + bn + j * MM::TN];
      }
      // MFMA groups; in their shadow, slice by slice: finish+stash the next
      // tile's element, then immediately refill its raw register with the
      // load for the tile after next (consumed one full iteration later).
      // Past the end both are harmless: stash writes zeros, loads are clamped.
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j) {
            if (DBG & 2) {
              asm volatile("" ::"v"(a[s][i]), "v"(b[s][j]));
            } else {
              acc[i][j] = MM::mma(a[s][i], b[s][j], acc[i][j]);
            }
          }
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if (e * NS / NE == s) {
            stash_one(e, k0 + BK, nxt);
            fetch_one(e, k0 + 2 * BK);
          }
      }
      // LDS writes of `nxt` visible + everyone done reading `cur`; global
      // loads stay in flight across the barrier (no vmcnt wait).
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
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
