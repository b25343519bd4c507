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

#define HB_KC 0  // operand source contiguous along k
#define HB_MC 1  // operand source contiguous along m (A) / n (B)

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
  // A k-contiguous operand (HB_KC, vector path) is staged row-major [m][LDK] instead of k-major [kk][LDA]:
  // its 16-byte global groups go to LDS with ONE ds_write_b128 and a lane's fragments of a whole k-step
  // are NS consecutive elements = NS/VEC ds_read_b128 (the k-major form needs NS ds_read_b32).
#ifndef HB_LDK_PAD
#define HB_LDK_PAD 4  // (8 and 12 measured no better: tools/kloop_cycles.hip with -DHB_LDK_PAD=...)
#endif
  static constexpr int LDK = BK + HB_LDK_PAD;
  static constexpr int A_ELEMS = (BM * LDK > BK * LDA) ? BM * LDK : BK * LDA;
  static constexpr int B_ELEMS = (BN * LDK > BK * LDB) ? BN * LDK : BK * LDB;
  static constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS;  // one LDS buffer
  static constexpr int LDS_ELEMS = 2 * BUF_ELEMS;      // double buffered
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
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // ---------------------------------------------------------------------------
  // Vectorised operand path.  Same pipeline as run(), but every thread moves
  // 16-byte groups (VEC = 4 floats / 2 doubles): one global load, one address
  // computation and one mask per group instead of per element -- on gfx950 the
  // fp32 MFMA shares the vector ALU, so every VALU instruction saved in the
  // loaders is MFMA issue time won back (profiles/r01_ablate_sgpA_loop.txt).
  //
  // Operand modes:
  //   HB_KC  source is k-contiguous (row-major A(m,:) / B^T(n,:)): the group is
  //          VEC consecutive k of one row/column; stashed transposed as VEC
  //          ds_write_b32 into the k-major LDS tile.
  //   HB_MC  source is contiguous along m (or n): the group is VEC consecutive
  //          rows/columns of one k; stashed with ONE ds_write_b128.
  // Functors (VT = ext_vector(VEC)):
  //   la(m, k) -> RawA (any POD; usually VT)   fa(RawA, m, k) -> VT
  //   lb(k, n) -> RawB                         fb(RawB, k, n) -> VT
  // with (m,k) / (k,n) the group's FIRST element.  Requirements (checked by the
  // host launchers, which fall back to run() otherwise): (kend - kbeg) % BK == 0,
  // 16-byte aligned group addresses.
  // ---------------------------------------------------------------------------
  static constexpr int VEC = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(VEC)));
  static constexpr int GA = (BM * BK / VEC) / NT, GB = (BN * BK / VEC) / NT;
  static constexpr int NG = GA + GB;

  template <int AMODE, int BMODE, class LA, class FA, class LB, class FB>
  __device__ __forceinline__ void run_vec(int kbeg, int kend, LA la, FA fa, LB lb, FB fb, T* __restrict__ lds) {
    static_assert((BM * BK / VEC) % NT == 0 && (BN * BK / VEC) % NT == 0, "vector fill must divide evenly");
    static_assert(BM % VEC == 0 && BN % VEC == 0 && BK % VEC == 0 && LDA % VEC == 0 && LDB % VEC == 0, "alignment");
    if (kbeg >= kend) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / WN, wn = w % WN;
    const int am = wm * WTM + (lane % MM::TM), ak = lane / MM::TM;
    const int bn = wn * WTN + (lane % MM::TN), bk = lane / MM::TN;
    typedef decltype(la(0, 0)) RawA;
    typedef decltype(lb(0, 0)) RawB;
    // two raw register sets: the tile stashed in step k was fetched two steps earlier, so operands that
    // stream from HBM (contraction lengths of 1e4..1e5) have two k-steps to arrive, not one
    RawA ra[2][GA];
    RawB rb[2][GB];

    // group g of this thread -> first (m, kk) / (kk, n) inside the tile
    auto a_m = [&](int g) { const int i = g * NT + tid; return AMODE == HB_KC ? i / (BK / VEC) : (i % (BM / VEC)) * VEC; };
    auto a_kk = [&](int g) { const int i = g * NT + tid; return AMODE == HB_KC ? (i % (BK / VEC)) * VEC : i / (BM / VEC); };
    auto b_n = [&](int g) { const int i = g * NT + tid; return BMODE == HB_KC ? i / (BK / VEC) : (i % (BN / VEC)) * VEC; };
    auto b_kk = [&](int g) { const int i = g * NT + tid; return BMODE == HB_KC ? (i % (BK / VEC)) * VEC : i / (BN / VEC); };

    auto fetch_one = [&](int g, int k0, RawA* ra, RawB* rb) {
      // past the end the tile is never consumed: re-read the last tile instead of running off the operand
      const int kc = k0 < kend ? k0 : kend - BK;
      if (g < GA) {
        ra[g < GA ? g : 0] = la(a_m(g), kc + a_kk(g));
      } else {
        const int gb = g - GA;
        rb[gb >= 0 ? gb : 0] = lb(kc + b_kk(gb), b_n(gb));
      }
    };
    auto stash_one = [&](int g, int k0, T* buf, const RawA* ra, const RawB* rb) {
      const int kc = k0 < kend ? k0 : kend - BK;
      if (g < GA) {
        const int m = a_m(g), kk = a_kk(g);
        const VT v = fa(ra[g < GA ? g : 0], m, kc + kk);
        if (AMODE == HB_KC) {
          *reinterpret_cast<VT*>(&buf[m * LDK + kk]) = v;
        } else {
          *reinterpret_cast<VT*>(&buf[kk * LDA + m]) = v;
        }
      } else {
        const int gb = g - GA;
        const int n = b_n(gb), kk = b_kk(gb);
        const VT v = fb(rb[gb >= 0 ? gb : 0], kc + kk, n);
        if (BMODE == HB_KC) {
          *reinterpret_cast<VT*>(&buf[A_ELEMS + n * LDK + kk]) = v;
        } else {
          *reinterpret_cast<VT*>(&buf[A_ELEMS + kk * LDB + n]) = v;
        }
      }
    };

#pragma unroll
    for (int g = 0; g < NG; ++g) fetch_one(g, kbeg, ra[0], rb[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) fetch_one(g, kbeg + BK, ra[1], rb[1]);
#pragma unroll
    for (int g = 0; g < NG; ++g) stash_one(g, kbeg, lds, ra[0], rb[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) fetch_one(g, kbeg + 2 * BK, ra[0], rb[0]);
    __syncthreads();

    // MFMA sub-step s contracts, for the lane's k-group h = lane / T{M,N}, over k = h*NS + s (the
    // contraction order is free as long as both operands agree): a lane's NS values are consecutive in k
    static_assert(NS % VEC == 0 && MM::TK * NS == BK && NS % 2 == 0, "fragment vectors");
    typedef T FragA[NS][RM];
    typedef T FragB[NS][RN];
    auto read_frags = [&](const T* __restrict__ As, FragA& a, FragB& b) {
      const T* __restrict__ Bs = As + A_ELEMS;
      if (AMODE == HB_KC) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int q = 0; q < NS; q += VEC) {
            const VT v = *reinterpret_cast<const VT*>(&As[(am + i * MM::TM) * LDK + ak * NS + q]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[q + e][i] = v[e];
          }
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int i = 0; i < RM; ++i) a[s][i] = As[(ak * NS + s) * LDA + am + i * MM::TM];
      }
      if (BMODE == HB_KC) {
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int q = 0; q < NS; q += VEC) {
            const VT v = *reinterpret_cast<const VT*>(&Bs[(bn + j * MM::TN) * LDK + bk * NS + q]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) b[q + e][j] = v[e];
          }
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int j = 0; j < RN; ++j) b[s][j] = Bs[(bk * NS + s) * LDB + bn + j * MM::TN];
      }
    };
    // One k-step, software-pipelined across the single barrier it contains:
    //   first half of the MFMAs  | finish + stash tile k+1 into `nxt`, refill the raw registers with tile k+2
    //   lgkmcnt(0) + s_barrier   (stash visible; `cur` was last read half a step ago)
    //   second half of the MFMAs | ds_read the fragments of tile k+1 from `nxt` into the other register set
    // so the LDS read latency sits under MFMA issue instead of at the top of every step.
    // the stash is packed into the first QS sub-steps so that its LDS writes have landed by the time the
    // mid-step wait is reached (a wave cannot issue MFMAs past an s_waitcnt)
    constexpr int QS = NS / 4 > 0 ? NS / 4 : 1;
    auto step = [&](const FragA& a, const FragB& b, FragA& an, FragB& bn_, int k0, T* __restrict__ nxt, RawA* ra, RawB* rb) {
#pragma unroll
      for (int s = 0; s < NS / 2; ++s) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j) acc[i][j] = MM::mma(a[s][i], b[s][j], acc[i][j]);
#pragma unroll
        for (int g = 0; g < NG; ++g)
          if (s < QS && g * QS / NG == s) {
            stash_one(g, k0 + BK, nxt, ra, rb);
            fetch_one(g, k0 + 3 * BK, ra, rb);
          }
        __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise hoists the stash (and the wait) to the top
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: global loads stay in flight across the barrier
      __builtin_amdgcn_s_barrier();
      read_frags(nxt, an, bn_);
      // keep the fragment reads issued here, ahead of the second half's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = NS / 2; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j) acc[i][j] = MM::mma(a[s][i], b[s][j], acc[i][j]);
    };

    FragA a0, a1;
    FragB b0, b1;
    T* __restrict__ buf0 = lds;
    T* __restrict__ buf1 = lds + BUF_ELEMS;
    read_frags(buf0, a0, b0);
    // two steps per trip, in ONE basic block (fragment reads must not be sunk into a successor block)
    int k0 = kbeg;
    for (; k0 + 2 * BK <= kend; k0 += 2 * BK) {
      step(a0, b0, a1, b1, k0, buf1, ra[1], rb[1]);
      step(a1, b1, a0, b0, k0 + BK, buf0, ra[0], rb[0]);
    }
    if (k0 < kend) step(a0, b0, a1, b1, k0, buf1, ra[1], rb[1]);
    // the last step's look-ahead reads are never used, but the LDS buffers are: callers reuse them right away
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  }

  // tile-local column of this thread's accumulator fragment j (its columns do not depend on i or r)
  __device__ __forceinline__ int frag_col(int j) const {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    return (w % WN) * WTN + j * MM::TN + MM::acc_col(lane);
  }
  // f(j, row, col, value): as for_each, with the column-fragment index (per-column data can be preloaded per j)
  template <class F>
  __device__ __forceinline__ void for_each_j(F f) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / WN, wn = w % WN;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < MM::NACC; ++r) {
          const int row = wm * WTM + i * MM::TM + MM::acc_row(lane, r);
          const int col = wn * WTN + j * MM::TN + MM::acc_col(lane);
          f(j, row, col, acc[i][j][r]);
        }
  }

  // f(row, col, value&): in-place variant of for_each
  template <class F>
  __device__ __forceinline__ void for_each_ref(F f) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / WN, wn = w % WN;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < MM::NACC; ++r) {
          const int row = wm * WTM + i * MM::TM + MM::acc_row(lane, r);
          const int col = wn * WTN + j * MM::TN + MM::acc_col(lane);
          T val = acc[i][j][r];  // (an accumulator is a vector register: its elements cannot be bound by reference)
          f(row, col, val);
          acc[i][j][r] = val;
        }
  }

  // f(row, col, value) over this thread's accumulator elements (tile-local)
  template <class F>
  __device__ __forceinline__ void for_each(F f) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
