// Dense linear algebra on the MFMA tile engine: batched matmul (+bias/act
// epilogue, split-K, lower-only output), blocked Cholesky, triangular inverse.
//
// Reference call sites: tf.matmul (gp/gp.py:50,122,125,171; nn.py:32),
// tf.cholesky (gp/kernels.py:101; gp/gp.py:135), tf.matrix_triangular_solve
// (gp/gp.py:162,169).  TensorFlow supplied these through Eigen/cuSOLVER; here
// they are hand-written for gfx950.
#include "common.cuh"
#include "sgp_strip.cuh"
#include "side_jobs.cuh"
#include "gemm_tile.cuh"
#include "../../include/henbun_hip.h"

// ===========================================================================
// matmul
// ===========================================================================
template <typename T>
struct MmArgs {
  const T* A;
  const T* B;
  T* C;
  long M, N, K, lda, ldb, ldc, sA, sB, sC, batch;
  T alpha, beta;
  const T* bias;
  long sBias;
  int act, flags, S, tile;
  int to_ws;  // results go to the workspace slabs and the finish kernel writes C (split-K, or SYM_OUT)
  int bfast;  // batch index fastest in blockIdx.x (batch % 8 == 0), see matmul_kernel
  T* ws;
  // HB_MM_COLSUM_B (hb_matmul_colsum: the bias gradient of a MatBias layer next to its weight gradient): the column
  // sums of op(B) = B[K][N] are accumulated by the workgroups of the first tile row while B streams through them;
  // colsum_ws: per-slab partials [S][N] (split-K), colsum: the result [N]
  T* colsum;
  T* colsum_ws;
};

template <typename T>
__device__ __forceinline__ T apply_act(int act, T v) {
  switch (act) {
    case HB_ACT_SIGMOID: return hb_sigmoid(v);
    case HB_ACT_RELU: return v > T(0) ? v : T(0);
    case HB_ACT_TANH: return hb_tanh(v);
    default: return v;
  }
}

// derivative of the activation expressed through its OUTPUT y (HB_MM_ACTGRAD)
template <typename T>
__device__ __forceinline__ T act_grad(int act, T y) {
  switch (act) {
    case HB_ACT_SIGMOID: return y * (T(1) - y);
    case HB_ACT_RELU: return y > T(0) ? T(1) : T(0);
    case HB_ACT_TANH: return T(1) - y * y;
    default: return T(1);
  }
}

// k-step depth.  (32 was tried for the k-contiguous/k-contiguous 128x128 case -- whole 128-byte lines per row
// and step -- and measured slower: 62.8 vs 56.3 us on the Lbar contraction, profiles/r01_kloop_cycles.txt.)
template <typename T, bool TA, bool TB, int BT>
struct MmBK {
  static constexpr int value = 16;
};

template <typename T, bool TA, bool TB, bool FAST, int BT>
__global__ void __launch_bounds__(256) matmul_kernel(MmArgs<T> a) {
  typedef TileGemm<T, BT, BT, MmBK<T, TA, TB, BT>::value, 2, 2> G;
  __shared__ __attribute__((aligned(16))) T lds[G::LDS_ELEMS];
  const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
  const int lda = (int)a.lda, ldb = (int)a.ldb;
  const int tiles_n = (N + BT - 1) / BT;
  // blockIdx.x = tile * S + slab, slab fastest: workgroups are dealt to the 8 XCDs round-robin, so with S a
  // multiple of 8 every XCD works on its own contraction slabs (all tiles of them) and an operand slab is fetched
  // into ONE L2 instead of all eight (the Lbar contraction moved 106 MB for 33 MB of operands before this)
  // With a batch that is a multiple of 8 (experts) the batch index is fastest instead: an XCD then works on "its"
  // matrices only, and consecutive workgroups of one matrix walk the tiles of one slab (tile fastest), which share
  // operand rows through that XCD's L2.
  int tile, s;
  long b;
  if (a.bfast) {
    b = blockIdx.x % a.batch;
    const int rest = (int)(blockIdx.x / a.batch);
    const int ntile = tiles_n * ((M + BT - 1) / BT);
    s = rest / ntile;
    tile = rest - s * ntile;
  } else {
    tile = blockIdx.x / a.S;
    s = blockIdx.x - tile * a.S;
    b = blockIdx.y;
  }
  const int row0 = (tile / tiles_n) * BT;
  const int col0 = (tile % tiles_n) * BT;
  if ((a.flags & (HB_MM_LOWER_OUT | HB_MM_TRIL_OUT | HB_MM_PHI_OUT | HB_MM_SYMLOW_OUT)) && col0 > row0 + BT - 1) {
    if ((a.flags & (HB_MM_TRIL_OUT | HB_MM_PHI_OUT)) && !a.to_ws) {
      // a tile wholly above the diagonal: all zero (with split-K the finish kernel writes them)
      T* Cb = a.C + b * a.sC;
      for (int idx = threadIdx.x; idx < BT * BT; idx += 256) {
        const int r = row0 + idx / BT, c = col0 + idx % BT;
        if (r < M && c < N) Cb[(long)r * a.ldc + c] = T(0);
      }
    }
    return;
  }
  int kchunk = (K + a.S - 1) / a.S;
  kchunk = ((kchunk + G::BK - 1) / G::BK) * G::BK;
  const int kbeg = s * kchunk;
  int kend = kbeg + kchunk;
  if (kend > K) kend = K;
  const T* Ab = a.A + b * a.sA;
  const T* Bb = a.B + b * a.sB;
  G g;
  g.zero();
  const int Mm1 = M - 1, Nm1 = N - 1;
  auto la = [&](int m, int k) -> T {
    const int r = row0 + m;
    const int rr = r < M ? r : Mm1;
    return TA ? Ab[k * lda + rr] : Ab[rr * lda + k];
  };
  auto fa = [&](T raw, int m, int k) -> T { return row0 + m < M ? raw : T(0); };
  auto lb = [&](int k, int n) -> T {
    const int c = col0 + n;
    const int cc = c < N ? c : Nm1;
    return TB ? Bb[cc * ldb + k] : Bb[k * ldb + cc];
  };
  auto fb = [&](T raw, int k, int n) -> T { return col0 + n < N ? raw : T(0); };
  if constexpr (FAST) {
    // vector path: 16-byte groups; M (TA) / N (!TB) multiples of VEC so a group is wholly in or out
    typedef typename G::VT VT;
    constexpr int VEC = G::VEC;
    const VT zero = {};
    auto la4 = [&](int m, int k) -> VT {
      const int r = row0 + m;
      if (TA) return *reinterpret_cast<const VT*>(&Ab[k * lda + (r < M ? r : M - VEC)]);
      return *reinterpret_cast<const VT*>(&Ab[(r < M ? r : Mm1) * lda + k]);
    };
    auto fa4 = [&](VT raw, int m, int k) -> VT { return row0 + m < M ? raw : zero; };
    auto lb4 = [&](int k, int n) -> VT {
      const int c = col0 + n;
      if (TB) return *reinterpret_cast<const VT*>(&Bb[(c < N ? c : Nm1) * ldb + k]);
      return *reinterpret_cast<const VT*>(&Bb[k * ldb + (c < N ? c : N - VEC)]);
    };
    // column sums of B on the way (HB_MM_COLSUM_B; TA, !TB only): a thread's B groups always sit on the same VEC columns
    // (group index = g * 256 + tid, BN / VEC groups per k row, 256 % (BN / VEC) == 0), so it keeps ONE running vector; the
    // engine stashes every tile of [kbeg, kend) exactly once, in order, and then re-stashes the last one as a look-ahead
    // that is never consumed: only the first (kend - kbeg) / BK * groups-per-thread stashes count.
    VT colacc = zero;
    int cs_left = 0;
    if (TA && !TB && (a.flags & HB_MM_COLSUM_B) && row0 == 0) cs_left = ((kend - kbeg) / G::BK) * ((BT * G::BK / VEC) / 256);
    auto fb4 = [&](VT raw, int k, int n) -> VT {
      const VT v = col0 + n < N ? raw : zero;
      if (TA && !TB) {
        if (cs_left > 0) {
          colacc += v;
          --cs_left;
        }
      }
      return v;
    };
    g.template run_vec<(TA ? HB_MC : HB_KC), (TB ? HB_KC : HB_MC)>(kbeg, kend, la4, fa4, lb4, fb4, lds);
    if (TA && !TB && (a.flags & HB_MM_COLSUM_B) && row0 == 0) {
      // 256 / (BT / VEC) threads hold partial sums of the same VEC columns: fold them through LDS (free after run_vec)
      constexpr int GPR = BT / VEC, ROWS = 256 / GPR;
      T* red = lds;
#pragma unroll
      for (int e = 0; e < VEC; ++e) red[(threadIdx.x / GPR) * BT + (threadIdx.x % GPR) * VEC + e] = colacc[e];
      __syncthreads();
      if ((int)threadIdx.x < BT && col0 + (int)threadIdx.x < N) {
        T sum = T(0);
#pragma unroll 4
        for (int r = 0; r < ROWS; ++r) sum += red[r * BT + threadIdx.x];   // fixed order
        T* dst = a.to_ws ? a.colsum_ws + (long)s * N : a.colsum;
        dst[col0 + threadIdx.x] = sum;
      }
      __syncthreads();
    }
  } else {
    g.template run<!TA, TB>(kbeg, kend, la, fa, lb, fb, lds);
  }
  if (a.to_ws) {
    T* wsb = a.ws + ((long)s * a.batch + b) * a.M * a.N;
    g.for_each([&](int row, int col, T v) {
      const long r = row0 + row, c = col0 + col;
      if (r < a.M && c < a.N) wsb[r * a.N + c] = a.alpha * v;
    });
  } else {
    T* Cb = a.C + b * a.sC;
    if (a.flags & HB_MM_ACTGRAD) {
      // two passes: every Y load is issued before the first store (a load inside the store loop waits for the
      // store before it -- C may alias Y as far as the compiler knows)
      const T* __restrict__ Yb = a.bias + b * a.sBias;
      g.for_each_ref([&](int row, int col, T& v) {
        const long r = row0 + row, c = col0 + col;
        const T y = Yb[(r < a.M ? r : a.M - 1) * a.N + (c < a.N ? c : a.N - 1)];
        v *= a.alpha * act_grad<T>(a.act, y);
      });
      g.for_each([&](int row, int col, T v) {
        const long r = row0 + row, c = col0 + col;
        if (r < a.M && c < a.N) Cb[r * a.ldc + c] = v;
      });
      return;
    }
    const T* biasb = a.bias ? a.bias + b * a.sBias : nullptr;
    // the bias of this thread's columns, loaded once: a `biasb[c]` inside the store loop is re-loaded after every
    // store (it may alias C as far as the compiler knows): one dependent round trip per output element
    T breg[G::RN];
#pragma unroll
    for (int j = 0; j < G::RN; ++j) {
      const int c = col0 + g.frag_col(j);
      breg[j] = biasb ? biasb[c < N ? c : N - 1] : T(0);
    }
    g.for_each_j([&](int j, int row, int col, T v) {
      const long r = row0 + row, c = col0 + col;
      if (r < a.M && c < a.N) {
        T o = a.alpha * v;
        o += breg[j];
        o = apply_act<T>(a.act, o);
        if (a.beta != T(0)) o += a.beta * Cb[r * a.ldc + c];
        if ((a.flags & (HB_MM_TRIL_OUT | HB_MM_PHI_OUT)) && c > r) o = T(0);
        if ((a.flags & HB_MM_PHI_OUT) && c == r) o *= T(0.5);
        if (a.flags & HB_MM_SYMLOW_OUT) {
          // half the lower triangle, mirrored: elements above the diagonal come from their mirror images
          if (c > r) return;
          o *= T(0.5);
          Cb[c * a.ldc + r] = o;
        }
        Cb[r * a.ldc + c] = o;
      }
    });
  }
}

template <typename T>
__global__ void __launch_bounds__(256) matmul_splitk_finish_kernel(MmArgs<T> a) {
  const long total = a.batch * a.M * a.N;
  const long stride = (long)gridDim.x * blockDim.x;
  if ((a.flags & HB_MM_COLSUM_B) && a.colsum) {
    // the column sums of B ride along: S <= 128 slab partials per column.  16 lanes per column, every lane's (at most
    // four) loads independent, then a fixed-order fold across the 16 lanes: one memory round trip (a serial loop over the
    // slabs is one dependent round trip PER SLAB -- it tripled the duration of this launch)
    const int sl = threadIdx.x & 15;
    for (long cb = blockIdx.x; cb * 16 < a.N; cb += gridDim.x) {
      const long c = cb * 16 + (threadIdx.x >> 4);
      const long cc = c < a.N ? c : a.N - 1;
      T v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int s = sl + 16 * q;
        v[q] = s < a.S ? a.colsum_ws[(long)s * a.N + cc] : T(0);
      }
      T acc = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
      if (sl == 0 && c < a.N) a.colsum[c] = acc;
    }
  }
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long b = t / (a.M * a.N);
    const long rem = t - b * a.M * a.N;
    const long r = rem / a.N, c = rem - r * a.N;
    T* cp = a.C + b * a.sC + r * a.ldc + c;
    if ((a.flags & (HB_MM_TRIL_OUT | HB_MM_PHI_OUT)) && c > r) {
      cp[0] = T(0);
      continue;
    }
    if ((a.flags & HB_MM_LOWER_OUT) && (c / a.tile) * a.tile > (r / a.tile) * a.tile + a.tile - 1) continue;
    // slab sums, four independent loads in flight per step (a plain `acc += ws[s]` loop with a run-time trip count
    // is one dependent round trip per slab); the order of the additions is fixed, so the result is deterministic
    auto slab_sum = [&](long at) -> T {
      T acc0 = T(0), acc1 = T(0), acc2 = T(0), acc3 = T(0);
      int s = 0;
      // sixteen loads in flight per round while they last (64 slabs of a minibatch-deep weight gradient were sixteen
      // dependent round trips at four per round: 7.5 us for 4 MB); the order of the additions stays fixed
      for (; s + 16 <= a.S; s += 16) {
        T v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = a.ws[(long)(s + q) * total + at];
#pragma unroll
        for (int q = 0; q < 16; q += 4) acc0 += v[q], acc1 += v[q + 1], acc2 += v[q + 2], acc3 += v[q + 3];
      }
      for (; s + 4 <= a.S; s += 4) {
        const T v0 = a.ws[(long)s * total + at], v1 = a.ws[(long)(s + 1) * total + at];
        const T v2 = a.ws[(long)(s + 2) * total + at], v3 = a.ws[(long)(s + 3) * total + at];
        acc0 += v0, acc1 += v1, acc2 += v2, acc3 += v3;
      }
      for (; s < a.S; ++s) acc0 += a.ws[(long)s * total + at];
      return (acc0 + acc1) + (acc2 + acc3);
    };
    if (a.flags & HB_MM_SYMLOW_OUT) {
      cp[0] = T(0.5) * slab_sum(c > r ? b * a.M * a.N + c * a.N + r : t);
      continue;
    }
    T acc = slab_sum(t);
    if (a.flags & HB_MM_SYM_OUT) {
      const long tt = b * a.M * a.N + c * a.N + r;  // the mirrored element
      const T acc2 = slab_sum(tt);
      cp[0] = T(0.5) * (acc + acc2);
      continue;
    }
    if ((a.flags & HB_MM_PHI_OUT) && c == r) acc *= T(0.5);
    if (a.flags & HB_MM_ACTGRAD) {
      cp[0] = acc * act_grad<T>(a.act, a.bias[b * a.sBias + r * a.N + c]);
      continue;
    }
    if (a.bias) acc += a.bias[b * a.sBias + c];
    acc = apply_act<T>(a.act, acc);
    if (a.beta != T(0)) acc += a.beta * cp[0];
    cp[0] = acc;
  }
}

// ===========================================================================
// Small GEMMs (the three M^3 products of the Cholesky VJP, 512^3 at cfg 2): split-K INSIDE the workgroup.
//
// A 512x512 result is 64 tiles of 64x64: to fill 256 CUs the tile engine above splits the contraction over
// workgroups, writes S slabs and needs a second launch to fold them (8-10 us + 5.9 us per product).  Here one
// 256-thread workgroup owns ONE 32x32 output tile (256 of them: one per CU) and its four waves each contract a
// quarter of K into a private accumulator; the four partial tiles meet in LDS (16 KB) and every thread finishes
// four output elements -- epilogues (alpha, bias, activation, beta, tril / Phi / symmetrise) included, no slabs in
// HBM, no finish launch.  Operands go straight from L2 into MFMA fragments, two register sets used alternately
// (see sgp_A_strip2_kernel for why the loop is written that way): a "k-major" operand (stored [K][M]) loads whole
// 128-byte rows (lane (li, h) takes k = k0 + 16 h + j, j = 0..15), an "m-major" one 64 contiguous bytes per lane.
// SYM_OUT: the workgroup of tile (i, j >= ... i >= j) carries a second accumulator for the mirrored tile (j, i) and
// writes both halves of (X + X^T)/2.
// ===========================================================================
#define WGK_LD 33
// Gram VJP in the epilogue (round 4): when the product IS Kbar -- the symmetric result S = L^-T Phi L^-1 of the Cholesky VJP,
// whose only reader is the VJP of K(X, X) -- the workgroup of tile (ti, tj) turns its 32 x 32 block of S into the tile's
// share of the row gradients (reference gp/kernels.py:54-101 under TF autodiff; the formulas of gram_bwd_side_kernel, side 3),
// leaves it in `part`, and the LAST workgroup of a row of tiles to arrive (counter per (batch, ti): zero at entry, left
// zero) folds the tiles in a fixed order into Xbar and the lengthscale partials.  One launch less per step.
struct MmGramVjp {
  const float* X = nullptr;     // [B][M][d] (sX: batch stride, 0 = shared)
  long sX = 0;
  const float* ell = nullptr;   // [B?][dl]
  long sEll = 0, dl = 1, d = 1;
  float* Xbar = nullptr;        // [B][M][d]
  float* ell_partial = nullptr; // [B * M][d]
  float* part = nullptr;        // [B][tiles_m][tiles_n][32][2 d]
  unsigned* counters = nullptr; // [B][tiles_m]
};
#define HB_MMG_MAXD 4
template <bool TA, bool TB, bool SYM>
__global__ void __launch_bounds__(256) matmul_wgk_kernel(MmArgs<float> a, int nown, HbSideJobs side, MmGramVjp gv) {
  if ((int)blockIdx.x >= nown) {   // side jobs riding on this launch (side_jobs.cuh)
    if (blockIdx.y == 0) hb_side_run(side, (int)blockIdx.x - nown);
    return;
  }
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  __shared__ float red[SYM ? 8 : 4][32][WGK_LD];
  const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
  const int lda = (int)a.lda, ldb = (int)a.ldb;
  const int tiles_n = N / 32;
  const long b = blockIdx.y;
  int ti, tj;
  if (SYM) {
    // lower tiles only, linear index -> (ti >= tj)
    int t = blockIdx.x;
    ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.x / tiles_n;
    tj = blockIdx.x - ti * tiles_n;
  }
  const int row0 = ti * 32, col0 = tj * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  float* Cb = a.C + b * a.sC;
  const bool lower_only = (a.flags & (HB_MM_LOWER_OUT | HB_MM_TRIL_OUT | HB_MM_PHI_OUT | HB_MM_SYMLOW_OUT)) != 0;
  if (!SYM && lower_only && col0 > row0 + 31) {
    if (a.flags & (HB_MM_TRIL_OUT | HB_MM_PHI_OUT)) {
      for (int idx = tid; idx < 1024; idx += 256) Cb[(long)(row0 + (idx >> 5)) * a.ldc + col0 + (idx & 31)] = 0.f;
    }
    return;
  }
  const float* __restrict__ Ab = a.A + b * a.sA;
  const float* __restrict__ Bb = a.B + b * a.sB;
  const int Kw = K / 4, kbeg = w * Kw, nch = Kw / 32;
  typename MM::Acc acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f, acc2[r] = 0.f;

  // CH chunks of 32 per operand register set: a whole wave-slice of K = 512 (4 chunks) is in flight at once -- these
  // products are a single dependent round trip to L2 / the Infinity Cache plus ~2 us of MFMAs, so the loads of every
  // chunk are issued before the first MFMA (1 wave per SIMD: 512 registers to spend); deeper contractions
  // alternate two such sets.
  constexpr int CH = SYM ? 2 : 4;
  struct Frag {
    float a[CH][16], b[CH][16], a2[SYM ? CH : 1][16], b2[SYM ? CH : 1][16];
  };
  // operand element (m, k): A_op[m][k] = TA ? A[k][m] : A[m][k];  B_op[k][n] = TB ? B[n][k] : B[k][n]
  auto load_side = [&](float (&f)[16], const float* __restrict__ base, int ld, bool kmajor, int idx0, int k0) {
    if (kmajor) {
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = base[(long)(k0 + 16 * h + j) * ld + idx0 + li];
    } else {
      const float* p = base + (long)(idx0 + li) * ld + k0 + 16 * h;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const V4 q = *reinterpret_cast<const V4*>(p + 4 * v);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) f[4 * v + s2] = q[s2];
      }
    }
  };
  auto load = [&](Frag& f, int c0) {
    if (c0 >= nch) return;  // (uniform) nothing left for this set
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int cc = c0 + q < nch ? c0 + q : nch - 1;   // a ragged last set re-reads its last chunk (never used)
      const int k0 = kbeg + 32 * cc;
      load_side(f.a[q], Ab, lda, TA, row0, k0);
      load_side(f.b[q], Bb, ldb, !TB, col0, k0);
      if (SYM) {
        load_side(f.a2[q], Ab, lda, TA, col0, k0);   // mirrored tile (tj, ti): rows of tile tj ...
        load_side(f.b2[q], Bb, ldb, !TB, row0, k0);  // ... against the columns of tile ti
      }
    }
  };
  auto compute = [&](const Frag& f, int c0) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      if (c0 + q >= nch) break;  // uniform
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc = MM::mma(f.a[q][j], f.b[q][j], acc);
        if (SYM) acc2 = MM::mma(f.a2[q][j], f.b2[q][j], acc2);
      }
    }
  };
  {
    Frag fa, fb;
    load(fa, 0);
#pragma nounroll
    for (int c = 0; c < nch; c += 2 * CH) {
      load(fb, c + CH);
      __builtin_amdgcn_sched_barrier(0);
      compute(fa, c);
      __builtin_amdgcn_sched_barrier(0);
      load(fa, c + 2 * CH);
      __builtin_amdgcn_sched_barrier(0);
      compute(fb, c + CH);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the four partial tiles meet in LDS
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[w][MM::acc_row(lane, r)][li] = acc[r];
    if (SYM) red[4 + w][MM::acc_row(lane, r)][li] = acc2[r];
  }
  __syncthreads();
  const float* biasb = a.bias ? a.bias + b * a.sBias : nullptr;
  float gval[4];   // (the Gram VJP epilogue reads the four values this thread stores)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e, r = idx >> 5, c = idx & 31;
    float v = ((red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c])) * a.alpha;
    gval[e] = v;
    const long gr = row0 + r, gcn = col0 + c;
    if (SYM) {
      // (X + X^T)/2: element (r, c) of tile (ti, tj) pairs with element (c, r) of the mirrored tile (tj, ti)
      const float vm = ((red[4][c][r] + red[5][c][r]) + (red[6][c][r] + red[7][c][r])) * a.alpha;
      const float sy = 0.5f * (v + vm);
      Cb[gr * a.ldc + gcn] = sy;
      continue;
    }
    if (a.flags & HB_MM_SYMLOW_OUT) {
      // half the lower triangle, mirrored.  On a diagonal tile the element above the diagonal takes its mirror
      // image's value; the mirrored TILE of an off-diagonal one is written below (transposed read: coalesced store)
      const int rr = (ti == tj && c > r) ? c : r, cc = (ti == tj && c > r) ? r : c;
      Cb[gr * a.ldc + gcn] = 0.5f * a.alpha * ((red[0][rr][cc] + red[1][rr][cc]) + (red[2][rr][cc] + red[3][rr][cc]));
      continue;
    }
    if (biasb) v += biasb[gcn];
    v = apply_act<float>(a.act, v);
    if (a.beta != 0.f) v += a.beta * Cb[gr * a.ldc + gcn];
    if ((a.flags & (HB_MM_TRIL_OUT | HB_MM_PHI_OUT)) && gcn > gr) v = 0.f;
    if ((a.flags & HB_MM_PHI_OUT) && gcn == gr) v *= 0.5f;
    Cb[gr * a.ldc + gcn] = v;
  }
  if (!SYM && gv.X) {
    // ---- the tile's share of the Gram VJP: rows (tid >> 5) + 8 e, column tid & 31; 32-lane sums over the columns
    const int d = (int)gv.d, c = tid & 31;
    const float* Xb = gv.X + b * gv.sX;
    const float* eb = gv.ell + b * gv.sEll;
    float il[HB_MMG_MAXD], xj[HB_MMG_MAXD];
#pragma unroll
    for (int k = 0; k < HB_MMG_MAXD; ++k) {
      il[k] = k < d ? 1.f / eb[gv.dl == 1 ? 0 : k] : 0.f;
      xj[k] = k < d ? Xb[(long)(col0 + c) * d + k] : 0.f;
    }
    const int tiles_m = M / 32;
    float* pt = gv.part + ((((long)b * tiles_m + ti) * tiles_n + tj) * 32) * 2 * d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = (tid >> 5) + 8 * e;
      float dm[HB_MMG_MAXD], r2 = 0.f;
#pragma unroll
      for (int k = 0; k < HB_MMG_MAXD; ++k) {
        dm[k] = k < d ? (Xb[(long)(row0 + r) * d + k] * il[k] - xj[k] * il[k]) : 0.f;
        r2 += dm[k] * dm[k];
      }
      const float km = hb_exp(-0.5f * r2), kb = gval[e];
      const float em = kb * km, gm = (kb + kb) * km;
#pragma unroll
      for (int k = 0; k < HB_MMG_MAXD; ++k) {
        if (k < d) {
          float ga = (-dm[k] * gm) * il[k], la = (dm[k] * dm[k] * em) * il[k];
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) {
            ga += __shfl_xor(ga, off, 32);
            la += __shfl_xor(la, off, 32);
          }
          if (c == 0) {
            __hip_atomic_store(pt + r * 2 * d + k, ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pt + r * 2 * d + d + k, la, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    // write-through partials, drained, then the row's counter (cdna_hip_programming.md Guideline 16 R1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ unsigned s_lastg;
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(gv.counters + b * tiles_m + ti, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_lastg = old == (unsigned)tiles_n - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_lastg) {
      const float* p0 = gv.part + (((long)b * tiles_m + ti) * tiles_n) * 32 * 2 * d;
      for (int idx = tid; idx < 32 * 2 * d; idx += 256) {
        float sum = 0.f;
        for (int t = 0; t < tiles_n; ++t) sum += __hip_atomic_load(const_cast<float*>(p0) + (long)t * 32 * 2 * d + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int r = idx / (2 * d), q = idx - r * 2 * d;
        if (q < d)
          gv.Xbar[((long)b * M + row0 + r) * d + q] = sum;
        else
          gv.ell_partial[((long)b * M + row0 + r) * d + q - d] = sum;
      }
      if (tid == 0) __hip_atomic_store(gv.counters + b * tiles_m + ti, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!SYM && (a.flags & HB_MM_SYMLOW_OUT) && ti != tj) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e, r = idx >> 5, c = idx & 31;
      Cb[(long)(col0 + r) * a.ldc + row0 + c] = 0.5f * a.alpha * ((red[0][c][r] + red[1][c][r]) + (red[2][c][r] + red[3][c][r]));
    }
  }
  if (SYM && ti != tj) {
    // the mirrored tile (tj, ti): element (r, c) there = the symmetrised element (c, r) here
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e, r = idx >> 5, c = idx & 31;
      const float v = ((red[0][c][r] + red[1][c][r]) + (red[2][c][r] + red[3][c][r])) * a.alpha;
      const float vm = ((red[4][r][c] + red[5][r][c]) + (red[6][r][c] + red[7][r][c])) * a.alpha;
      Cb[(long)(col0 + r) * a.ldc + row0 + c] = 0.5f * (v + vm);
    }
  }
}

// eligibility of the in-workgroup split-K kernel (fp32 only)
template <typename T>
static bool matmul_wgk_ok(const MmArgs<T>&, long, long, long, long, int, bool) { return false; }
template <>
bool matmul_wgk_ok<float>(const MmArgs<float>& a, long M, long N, long K, long batch, int flags, bool aligned) {
  const bool off = hb_debug_get("mm_no_wgk", 0) != 0;  // diagnostic A/B switch (hb_debug_set)
  if (off || !aligned) return false;
  if (M % 32 || N % 32 || K % 128 || K < 128 || K > 4096) return false;
  if (flags & HB_MM_ACTGRAD) return false;
  if ((flags & HB_MM_SYM_OUT) && M != N) return false;
  const long tiles = (M / 32) * (N / 32) * batch;
  return tiles <= 1024 && batch <= 65535;
}
template <typename T>
static int matmul_wgk_launch(const MmArgs<T>&, int, int, hipStream_t) { return -1; }
int matmul_wgk_launch_g(const MmArgs<float>& a, int transA, int transB, hipStream_t stream, const MmGramVjp& gv);
template <>
int matmul_wgk_launch<float>(const MmArgs<float>& a, int transA, int transB, hipStream_t stream) { return matmul_wgk_launch_g(a, transA, transB, stream, MmGramVjp()); }
int matmul_wgk_launch_g(const MmArgs<float>& a, int transA, int transB, hipStream_t stream, const MmGramVjp& gv) {
  const bool sym = (a.flags & HB_MM_SYM_OUT) != 0;
  const long nt = a.M / 32;
  const int nown = (int)(sym ? nt * (nt + 1) / 2 : (a.M / 32) * (a.N / 32));
  const HbSideJobs sj = hb_side_take();   // pending side jobs of this thread ride along as extra workgroups
  dim3 grid((unsigned)(nown + sj.total), (unsigned)a.batch, 1);
#define HB_WGK(TA_, TB_)                                                                                   \
  do {                                                                                                     \
    if (sym)                                                                                               \
      hipLaunchKernelGGL((matmul_wgk_kernel<TA_, TB_, true>), grid, dim3(256), 0, stream, a, nown, sj, gv);    \
    else                                                                                                   \
      hipLaunchKernelGGL((matmul_wgk_kernel<TA_, TB_, false>), grid, dim3(256), 0, stream, a, nown, sj, gv);   \
  } while (0)
  if (!transA && !transB)
    HB_WGK(false, false);
  else if (!transA && transB)
    HB_WGK(false, true);
  else if (transA && !transB)
    HB_WGK(true, false);
  else
    HB_WGK(true, true);
#undef HB_WGK
  HB_LAUNCH_CHECK();
  return 0;
}

// ===========================================================================
// Row-streaming GEMM (round 3): C[M, N] = epilogue(A[M, K] op(B)[K, N]) for a TALL A (a minibatch of rows) and a small
// op(B) (a layer's weights: K, N <= 256) -- the MatBias layers of the amortised encoder and their input gradients
// (reference nn.py:31-32, 73-84; cfg 4: [32768, 64] x [64, 256], [32768, 256] x [256, 32], ...).
//
// The tile engine above gives every 64 x 64 output tile its own workgroup with a double-buffered LDS pipeline and one
// barrier per 16-deep k-step: for K = 16 .. 256 that is all prologue and epilogue, and every such launch took ~15 us at
// n = 32768 whatever its size (2.2-2.6 TB/s on operands that stream once).  Here the WEIGHTS are the LDS-resident
// operand: a workgroup stages op(B) once as Ws[column][k] (k contiguous: the B-fragment of an MFMA is one 16-byte LDS
// read per four k), then each of its four waves takes 32 rows of A at a time, reads them straight from global memory as
// MFMA A-fragments (the contraction index is permuted so a lane reads G contiguous floats of its row: whole cache
// lines, no LDS staging, no barrier in the loop), keeps the 32 x N result in registers (N/32 accumulator tiles) and
// writes it once with the epilogue applied (bias + activation, or the activation gradient from the layer's output).
// Two workgroups per CU (72 KB of LDS each).
// ===========================================================================
#define RS_LDS_FLOATS 18432
// NT accumulator tiles (32 columns each) per wave; WC waves side by side along the columns (N <= 32 NT WC), NW / WC wave
// rows of 32 matrix rows each.  n = 32768 rows are 1024 wave tiles for 2048 resident waves: with a whole 32 x N result
// per wave every wave runs ONE tile -- staging, operand latency, MFMAs and a long epilogue in sequence, nothing to cover
// them (N = 256 as NT = 8: 39 us; as NT = 4 x 2 wave columns: 31.6 us; the tile engine 25.7).  For N > 128 a workgroup is
// therefore EIGHT waves side by side, one 32 x 32 tile each (16 accumulators, <= 128 registers: four waves per SIMD
// with two workgroups per CU), and it walks several row tiles so the staged weights are used more than once.
template <bool TB, int NT, int WC, int G, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) matmul_rows_kernel(MmArgs<float> a) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  __shared__ __attribute__((aligned(16))) float Ws[RS_LDS_FLOATS];   // [32 NT WC columns][K + 4]
  const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
  const int lda = (int)a.lda, ldb = (int)a.ldb, ldc = (int)a.ldc;
  const int KLD = K + 4;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  constexpr int NP = 32 * NT * WC, ROWS = 32 * (NW / WC), NTHR = 64 * NW;
  const int wr = w / WC, wc = w % WC;
  // ---- stage op(B): element (column c, k) = op(B)[k][c]; columns past N are zero
  {
    const int kq = K >> 2;
    if (TB) {             // stored [N][ldb]: k contiguous
      for (int idx = tid; idx < NP * kq; idx += NTHR) {
        const int c = idx / kq, k4 = idx - c * kq;
        V4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < N) v = *reinterpret_cast<const V4*>(a.B + (long)c * ldb + 4 * k4);
        *reinterpret_cast<V4*>(&Ws[c * KLD + 4 * k4]) = v;
      }
    } else {
      // stored [K][ldb]: consecutive threads take consecutive COLUMNS of four k rows -- four coalesced scalar loads, one
      // 16-byte LDS store whose addresses are K + 4 floats apart (all bank groups).  (A 4 x 4 register transpose with
      // 16-byte loads was tried in both thread orders: column-major order puts a wave's stores on two bank groups, a
      // 32-way conflict; k-major order turns every load into 64 cache lines, and 512 workgroups asking the same 512
      // lines of the weights cost ~7 us of L2 hot-spotting.  Either way the launch never got under 26 us.)
      for (int idx = tid; idx < NP * kq; idx += NTHR) {
        const int k4 = idx / NP, c = idx - k4 * NP;
        V4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < N) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = a.B[(long)(4 * k4 + i) * ldb + c];
        }
        *reinterpret_cast<V4*>(&Ws[c * KLD + 4 * k4]) = v;
      }
    }
  }
  const int cbase = 32 * NT * wc;   // this wave's first column
  float breg[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = cbase + 32 * t + li;
    breg[t] = (a.bias && !(a.flags & HB_MM_ACTGRAD) && c < N) ? a.bias[c] : 0.f;
  }
  __syncthreads();
  const int nchunk = K / (2 * G);
  const int ntile = (M + ROWS - 1) / ROWS;
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int r0 = tile * ROWS + 32 * wr;
    if (r0 >= M || cbase >= N) continue;   // (whole waves; no barrier below)
    const int arow = r0 + li < M ? r0 + li : M - 1;
    const float* __restrict__ ap = a.A + (long)arow * lda + G * h;
    typename MM::Acc acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto load = [&](V4 (&f)[G / 4], int c) {
      const int cc = c < nchunk ? c : nchunk - 1;
#pragma unroll
      for (int v = 0; v < G / 4; ++v) f[v] = *reinterpret_cast<const V4*>(ap + 2 * G * cc + 4 * v);
    };
    auto compute = [&](const V4 (&f)[G / 4], int c) {
      if (c >= nchunk) return;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* wp = &Ws[(cbase + 32 * t + li) * KLD + 2 * G * c + G * h];
        V4 bv[G / 4];
#pragma unroll
        for (int v = 0; v < G / 4; ++v) bv[v] = *reinterpret_cast<const V4*>(wp + 4 * v);
#pragma unroll
        for (int v = 0; v < G / 4; ++v)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) acc[t] = MM::mma(f[v][s2], bv[v][s2], acc[t]);
      }
    };
    V4 fa[G / 4], fb[G / 4];
    load(fa, 0);
#pragma nounroll
    for (int c = 0; c < nchunk; c += 2) {
      load(fb, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(fa, c);
      __builtin_amdgcn_sched_barrier(0);
      load(fa, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(fb, c + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: accumulator layout = column on the lane, rows in the registers (two 128-byte row segments per store)
    if (a.flags & HB_MM_ACTGRAD) {
      const float* __restrict__ Y = a.bias;   // the layer's output, [M][N]
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = cbase + 32 * t + li;
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + MM::acc_row(lane, r);
          y[r] = Y[(long)(row < M ? row : M - 1) * N + (c < N ? c : N - 1)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + MM::acc_row(lane, r);
          if (row < M && c < N) a.C[(long)row * ldc + c] = a.alpha * acc[t][r] * act_grad<float>(a.act, y[r]);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = cbase + 32 * t + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + MM::acc_row(lane, r);
          if (row < M && c < N) a.C[(long)row * ldc + c] = apply_act<float>(a.act, a.alpha * acc[t][r] + breg[t]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-streaming GEMM with REGISTER-resident weights (K <= 128): a wave owns one 32-column tile of the result and keeps
// that tile's K x 32 block of op(B) in registers (K / 2 per lane, loaded once, straight from global memory in MFMA
// B-fragment order), then walks its share of the 32-row tiles of A: next tile's A fragments requested before the
// current tile's MFMAs, 32 x 32 result written with the epilogue applied.  No LDS, no barrier, no staging phase: the
// LDS form above runs every workgroup through stage -> barrier -> load -> MFMA -> store in lock step (all 512 of them
// at once, two row tiles each), so memory and matrix phases never overlap -- 24.6 us for [32768, 64] x [64, 256] whose
// MFMAs are 7.5 us and whose bytes are 8 us.  Here the waves are independent and drift apart: 23.7 us for that product,
// 18.0 (LDS form 21.7, tile engine 27.8) for the input gradient [32768, 32] x [256, 32]^T with the activation-gradient
// epilogue, 7.2 (8.9) for [32768, 16] x [16, 64].  What still holds the first one at 24 us was narrowed down with two
// timing variants: without its stores it takes the same time, with its MFMAs replaced by one FMA each it takes LONGER
// (41 us) -- the operand side is the limit: a lane's A fragment is 16 bytes of its own row, so every load instruction
// touches 64 different cache lines (the access pattern the fragment-major images remove for the sparse-GP operands);
// a coalesced load + LDS transpose of the row tile is the form that remains to be built.
// ---------------------------------------------------------------------------------------------------------------
// EPI: the epilogue, fixed at compile time -- 0 bias only, 1 bias + sigmoid, 2 bias + the activation in a.act (run-time
// switch), 3 the sigmoid's gradient from the layer's output, 4 the gradient of the activation in a.act.  With the
// activation switched per ELEMENT at run time the kernel was 13 000 lines of ISA in 2 500 basic blocks (tanhf inlined
// sixteen times per epilogue copy).
// EPI 5 (round 4): the Gaussian likelihood head of the product -- f = alpha A B + bias never leaves the registers: with y
// of the same shape the epilogue writes dmu = (y - f s) / var into C (and fbar = s (post dmu) into hd.fbar) and keeps the
// lane's partial sums of (ll, dscale, dvar); one partial triple per wave in hd.part[3][units], folded by hb_gauss_ll_fold.
// The per-point arithmetic is hb_gauss_point (chain_bodies.cuh): dmu / fbar agree with hb_gauss_ll to fp32 rounding.
struct MmHead {
  const float* y = nullptr;
  const float* scale = nullptr;
  const float* var = nullptr;
  float post = 0.f;
  float* fbar = nullptr;
  float* part = nullptr;
  long units = 0;
};
template <bool TB, int G, int KC, int EPI>
__global__ void __launch_bounds__(256) matmul_rowsreg_kernel(MmArgs<float> a, MmHead hd) {

  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  const int M = (int)a.M, N = (int)a.N;
  const int lda = (int)a.lda, ldb = (int)a.ldb, ldc = (int)a.ldc;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, h = lane >> 5;
  const int nct = (N + 31) / 32, nrt = (M + 31) / 32;
  const int gw = blockIdx.x * 4 + w, nw = gridDim.x * 4;
  const int ct = gw % nct;
  const int rstride = nw / nct;            // (the launcher makes nw a multiple of nct)
  const int col = 32 * ct + li, colc = col < N ? col : N - 1;
  // ---- this tile's weights: bf[c][4 v + s] = op(B)[2G c + 8 v + 4 h + s][col].  The contraction index is interleaved
  // between the half-waves at 16-byte granularity: the A fragments of lanes (li, 0) and (li, 1) are then ADJACENT 16-byte
  // pieces of row li, and a load instruction touches 32 cache lines instead of 64
  float bf[KC][G];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (TB) {
      const float* bp = a.B + (long)colc * ldb + 2 * G * c + 4 * h;
#pragma unroll
      for (int v = 0; v < G / 4; ++v) {
        const V4 q = *reinterpret_cast<const V4*>(bp + 8 * v);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) bf[c][4 * v + s2] = col < N ? q[s2] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const float q = a.B[(long)(2 * G * c + 8 * (j >> 2) + 4 * h + (j & 3)) * ldb + colc];
        bf[c][j] = col < N ? q : 0.f;
      }
    }
  }
  const float bias = (a.bias && !(a.flags & HB_MM_ACTGRAD) && col < N) ? a.bias[col] : 0.f;
  float h_ll = 0.f, h_sc = 0.f, h_vr = 0.f;     // EPI 5
  const float hs = (EPI == 5 && hd.scale) ? hd.scale[0] : 1.f;
  const float hv = EPI == 5 ? hd.var[0] : 1.f;
  const float hiv = 1.f / hv, hlc = -0.91893853320467274178f - 0.5f * hb_log(hv);
  auto load = [&](V4 (&f)[KC][G / 4], int rt) {
    const int rtc = rt < nrt ? rt : nrt - 1;
    const int row = 32 * rtc + li;
    const float* ap = a.A + (long)(row < M ? row : M - 1) * lda + 4 * h;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int v = 0; v < G / 4; ++v) f[c][v] = *reinterpret_cast<const V4*>(ap + 2 * G * c + 8 * v);
  };
  auto compute = [&](const V4 (&f)[KC][G / 4], int rt) {
    if (rt >= nrt) return;
    typename MM::Acc acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int v = 0; v < G / 4; ++v)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) acc = MM::mma(f[c][v][s2], bf[c][4 * v + s2], acc);
    const int r0 = 32 * rt;
    if (EPI == 5) {
      float y[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + MM::acc_row(lane, r);
        y[r] = hd.y[(long)(row < M ? row : M - 1) * N + colc];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + MM::acc_row(lane, r);
        if (row < M && col < N) {
          float gg;
          hb_gauss_point<float>(y[r], a.alpha * acc[r] + bias, hs, hiv, hlc, gg, h_ll, h_sc, h_vr);
          a.C[(long)row * ldc + col] = gg;
          if (hd.fbar) hd.fbar[(long)row * ldc + col] = hs * (hd.post * gg);
        }
      }
    } else if (EPI >= 3 && r0 + 32 <= M && col < N) {
      const float* __restrict__ yp = a.bias + (long)(r0 + 4 * h) * N + col;
      float* __restrict__ cp = a.C + (long)(r0 + 4 * h) * ldc + col;
      float y[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) y[r] = yp[((r & 3) + 8 * (r >> 2)) * N];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float gy = EPI == 3 ? y[r] * (1.f - y[r]) : act_grad<float>(a.act, y[r]);
        cp[((r & 3) + 8 * (r >> 2)) * ldc] = a.alpha * acc[r] * gy;
      }
    } else if (EPI >= 3) {
      const float* __restrict__ Y = a.bias;
      float y[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + MM::acc_row(lane, r);
        y[r] = Y[(long)(row < M ? row : M - 1) * N + colc];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + MM::acc_row(lane, r);
        const float gy = EPI == 3 ? y[r] * (1.f - y[r]) : act_grad<float>(a.act, y[r]);
        if (row < M && col < N) a.C[(long)row * ldc + col] = a.alpha * acc[r] * gy;
      }
    } else if (r0 + 32 <= M && col < N) {
      // whole tile inside the result: one 64-bit address per tile, 32-bit row offsets, no per-element masks (the epilogue
      // runs on the same pipe as the fp32 MFMAs and costs about as much as they do at K = 64: every instruction counts)
      float* __restrict__ cp = a.C + (long)(r0 + 4 * h) * ldc + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float o = a.alpha * acc[r] + bias;
        cp[((r & 3) + 8 * (r >> 2)) * ldc] = EPI == 0 ? o : (EPI == 1 ? hb_sigmoid(o) : apply_act<float>(a.act, o));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + MM::acc_row(lane, r);
        const float o = a.alpha * acc[r] + bias;
        const float v = EPI == 0 ? o : (EPI == 1 ? hb_sigmoid(o) : apply_act<float>(a.act, o));
        if (row < M && col < N) a.C[(long)row * ldc + col] = v;
      }
    }
  };
  int rt = gw / nct;
  if (rt < nrt) {
    V4 fa[KC][G / 4], fb[KC][G / 4];
    load(fa, rt);
#pragma nounroll
    for (; rt < nrt; rt += 2 * rstride) {
      load(fb, rt + rstride);
      __builtin_amdgcn_sched_barrier(0);
      compute(fa, rt);
      __builtin_amdgcn_sched_barrier(0);
      load(fa, rt + 2 * rstride);
      __builtin_amdgcn_sched_barrier(0);
      compute(fb, rt + rstride);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (EPI == 5) {
    // this wave's partial sums (fixed order inside the wave; a wave without tiles contributes zeros)
    h_ll = wave_sum(h_ll), h_sc = wave_sum(h_sc), h_vr = wave_sum(h_vr);
    if (lane == 0 && gw < hd.units) {
      hd.part[gw] = h_ll;
      hd.part[hd.units + gw] = h_sc;
      hd.part[2 * hd.units + gw] = h_vr;
    }
  }
}

static inline bool matmul_rowsreg_ok(const MmArgs<float>& a, int& G, int& KC) {
  const bool off = hb_debug_get("mm_no_rowsreg", 0) != 0;   // diagnostic: the LDS form
  // one column tile (N <= 32): the LDS form is the faster one ([32768, 64] x [64, 16]: 5.7 against 8.7 us) -- every wave
  // would hold the same weights
  if (off || a.N <= 32) return false;
  if (a.K == 16) {
    G = 8, KC = 1;
    return true;
  }
  G = 16;
  KC = (int)(a.K / 32);
  return a.K % 32 == 0 && (KC == 1 || KC == 2 || KC == 4);
}
static inline long matmul_rowsreg_wgs(long M, long N, int KC) {
  const long nct = hb_cdiv(N, 32), nrt = hb_cdiv(M, 32);
  // enough waves to fill the chip at this register count, at least ~2 row tiles each; a multiple of nct waves
  const long occ = KC >= 4 ? 2 : (KC == 2 ? 3 : 4);
  long wgs = hb_cdiv(nrt * nct, 8);
  if (wgs > 256 * occ) wgs = 256 * occ;
  long unit = nct;                 // 4 wgs must be a multiple of nct (nct <= 8)
  while (unit % 4 != 0) unit *= 2;
  unit /= 4;
  wgs = (wgs / unit) * unit;
  if (wgs < unit) wgs = unit;
  return wgs;
}
static int matmul_rowsreg_launch(const MmArgs<float>& a, int transB, int G, int KC, hipStream_t stream, const MmHead* head = nullptr) {
  dim3 grid((unsigned)matmul_rowsreg_wgs(a.M, a.N, KC), 1, 1);
  const bool ag = (a.flags & HB_MM_ACTGRAD) != 0;
  const int epi = head ? 5 : (ag ? (a.act == HB_ACT_SIGMOID ? 3 : 4) : (a.act == HB_ACT_NONE ? 0 : (a.act == HB_ACT_SIGMOID ? 1 : 2)));
  const MmHead hd = head ? *head : MmHead();
#define HB_RR3(TB_, G_, KC_, E_) hipLaunchKernelGGL((matmul_rowsreg_kernel<TB_, G_, KC_, E_>), grid, dim3(256), 0, stream, a, hd)
#define HB_RR2(TB_, G_, KC_)     \
  do {                           \
    if (epi == 0)                \
      HB_RR3(TB_, G_, KC_, 0);   \
    else if (epi == 1)           \
      HB_RR3(TB_, G_, KC_, 1);   \
    else if (epi == 2)           \
      HB_RR3(TB_, G_, KC_, 2);   \
    else if (epi == 3)           \
      HB_RR3(TB_, G_, KC_, 3);   \
    else if (epi == 4)           \
      HB_RR3(TB_, G_, KC_, 4);   \
    else                         \
      HB_RR3(TB_, G_, KC_, 5);   \
  } while (0)
#define HB_RR1(TB_)           \
  do {                        \
    if (G == 8)               \
      HB_RR2(TB_, 8, 1);      \
    else if (KC == 1)         \
      HB_RR2(TB_, 16, 1);     \
    else if (KC == 2)         \
      HB_RR2(TB_, 16, 2);     \
    else                      \
      HB_RR2(TB_, 16, 4);     \
  } while (0)
  if (transB)
    HB_RR1(true);
  else
    HB_RR1(false);
#undef HB_RR1
#undef HB_RR2
#undef HB_RR3
  HB_LAUNCH_CHECK();
  return 0;
}

static inline void matmul_rows_shape(long N, int& NT, int& WC, int& NW) {
  const long nt = (N + 31) / 32;
  if (nt > 4) {
    NT = 1, WC = 8, NW = 8;
    return;
  }
  WC = 1, NW = 4;
  NT = nt <= 1 ? 1 : nt <= 2 ? 2 : 4;
}
template <typename T>
static bool matmul_rows_ok(const MmArgs<T>&, int, int) { return false; }
template <>
bool matmul_rows_ok<float>(const MmArgs<float>& a, int transA, int transB) {
  const bool off = hb_debug_get("mm_no_rows", 0) != 0;   // diagnostic: the tile engine for everything
  if (off || transA || a.batch != 1 || a.M < 2048 || a.K < 8 || a.K > 256 || a.K % 8 != 0 || a.N < 1 || a.N > 256) return false;
  if (!(a.flags == 0 || a.flags == HB_MM_ACTGRAD) || a.beta != 0.f) return false;
  int NT, WC, NW;
  matmul_rows_shape(a.N, NT, WC, NW);
  if (32L * NT * WC * (a.K + 4) > RS_LDS_FLOATS) return false;
  if ((uintptr_t)a.A % 16 != 0 || a.lda % 4 != 0) return false;
  if (transB && ((uintptr_t)a.B % 16 != 0 || a.ldb % 4 != 0)) return false;
  return a.M * a.lda < 2147483647L && a.M * a.ldc < 2147483647L;
}
template <typename T>
static int matmul_rows_launch(const MmArgs<T>&, int, hipStream_t) { return -1; }
template <>
int matmul_rows_launch<float>(const MmArgs<float>& a, int transB, hipStream_t stream) {
  {
    int G2, KC2;
    if (matmul_rowsreg_ok(a, G2, KC2)) return matmul_rowsreg_launch(a, transB, G2, KC2, stream);
  }
  int NT, WC, NW;
  matmul_rows_shape(a.N, NT, WC, NW);
  const int G = a.K % 32 == 0 ? 16 : (a.K % 16 == 0 ? 8 : 4);
  const long ntile = hb_cdiv(a.M, 32 * (NW / WC));
  dim3 grid((unsigned)(ntile < 512 ? ntile : 512), 1, 1);
#define HB_RS3(TB_, NT_, WC_, G_, NW_) \
  hipLaunchKernelGGL((matmul_rows_kernel<TB_, NT_, WC_, G_, NW_>), grid, dim3(64 * NW_), 0, stream, a)
#define HB_RS2(TB_, NT_, WC_, NW_)     \
  do {                                 \
    if (G == 16)                       \
      HB_RS3(TB_, NT_, WC_, 16, NW_);  \
    else if (G == 8)                   \
      HB_RS3(TB_, NT_, WC_, 8, NW_);   \
    else                               \
      HB_RS3(TB_, NT_, WC_, 4, NW_);   \
  } while (0)
#define HB_RS1(TB_)          \
  do {                       \
    if (WC == 8)             \
      HB_RS2(TB_, 1, 8, 8);  \
    else if (NT == 1)        \
      HB_RS2(TB_, 1, 1, 4);  \
    else if (NT == 2)        \
      HB_RS2(TB_, 2, 1, 4);  \
    else                     \
      HB_RS2(TB_, 4, 1, 4);  \
  } while (0)
  if (transB)
    HB_RS1(true);
  else
    HB_RS1(false);
#undef HB_RS1
#undef HB_RS2
#undef HB_RS3
  HB_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int matmul_launch(const T* A, const T* B, T* C, long batch, long M, long N, long K, long lda, long ldb,
                         long ldc, long sA, long sB, long sC, int transA, int transB, double alpha, double beta,
                         const T* bias, long sBias, int act, int flags, T* ws, long ws_elems, hipStream_t stream,
                         T* colsum = nullptr) {
  HB_REQUIRE(batch >= 0 && M >= 0 && N >= 0 && K >= 0, "hb_matmul: negative extent");
  HB_REQUIRE(A && B && C, "hb_matmul: NULL pointer");
  HB_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "hb_matmul: leading dimension too small");
  HB_REQUIRE(batch <= 65535, "hb_matmul: batch too large");
  HB_REQUIRE((transA ? K : M) * lda < 2147483647L && (transB ? N : K) * ldb < 2147483647L,
             "hb_matmul: operand too large for 32-bit indexing");
  HB_REQUIRE(act >= HB_ACT_NONE && act <= HB_ACT_TANH, "hb_matmul: unknown activation %d", act);
  if (batch * M * N == 0) return 0;
  MmArgs<T> a;
  a.A = A; a.B = B; a.C = C;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.sA = sA; a.sB = sB; a.sC = sC;
  a.batch = batch;
  a.alpha = (T)alpha; a.beta = (T)beta;
  a.bias = bias; a.sBias = sBias;
  a.act = act; a.flags = flags;
  a.ws = ws;
  a.colsum = nullptr;
  a.colsum_ws = nullptr;
  if (colsum) {
    // the slab partials of the column sums sit at the end of the workspace (128 slabs at most)
    HB_REQUIRE(ws && ws_elems > 128 * N, "hb_matmul_colsum: workspace too small");
    ws_elems -= 128 * N;
    a.colsum_ws = ws + ws_elems;
  }
  // Tile / split choice.  These are latency-and-occupancy problems more often than throughput ones: a
  // 128x128 tile (64x64 per wave: 4 MFMAs per fragment pair) has the lowest staging cost per MFMA but
  // only pays when there are enough of them to cover the 256 CUs; otherwise 64x64 tiles, and the
  // contraction is split until ~2.5 workgroups per CU are in flight (each slice >= 64 deep).
  HB_REQUIRE(!(flags & HB_MM_SYM_OUT) || (M == N && ws && ws_elems >= batch * M * N && !bias && act == HB_ACT_NONE &&
                                          beta == 0.0 && !(flags & ~HB_MM_SYM_OUT)),
             "hb_matmul: SYM_OUT needs a square result, a workspace, and no other epilogue");
  HB_REQUIRE(!(flags & HB_MM_ACTGRAD) || (bias && beta == 0.0 && flags == HB_MM_ACTGRAD),
             "hb_matmul: ACTGRAD needs Y in `bias`, beta = 0 and no other flag");
  HB_REQUIRE(!(flags & HB_MM_SYMLOW_OUT) || (M == N && !bias && act == HB_ACT_NONE && beta == 0.0 && flags == HB_MM_SYMLOW_OUT),
             "hb_matmul: SYMLOW_OUT needs a square result and no other epilogue");
  if (!colsum && matmul_rows_ok<T>(a, transA, transB)) return matmul_rows_launch<T>(a, transB, stream);   // tall A, small op(B)
  // (A^T B over a minibatch into a small result -- the weight gradients of cfg 4 -- stays on the tile engine below.  A
  //  form with both operands read straight from global memory as MFMA fragments (64 x 64 block and one slab of rows per
  //  workgroup, four waves taking the slab's 32-row chunks in turn) was built twice: with `in range ? v : 0` on every
  //  loaded element the compiler issued a chunk's 96 loads two at a time with a full wait behind each pair (54 / 47 / 16 us
  //  per product); with a select-free common path 19.5 / 13.9 / 17.3 us -- against 15.2 each for the tile engine.  Removed;
  //  what it left behind is the split-K finish with sixteen slab loads in flight (7.5 -> 5.0 us).)
  const bool lower = (flags & (HB_MM_LOWER_OUT | HB_MM_TRIL_OUT | HB_MM_PHI_OUT | HB_MM_SYMLOW_OUT)) != 0;
  auto active_tiles = [&](int bt) -> long {
    const long tr = hb_cdiv(M, bt), tc = hb_cdiv(N, bt);
    return (lower && M == N) ? tr * (tr + 1) / 2 : tr * tc;
  };
  int BT = 64, S = 1;
  // (short contractions are store-bound: more, smaller workgroups hide the output traffic better -- the MLP
  //  layers of cfg 4, K = 32 / 64: 0.319 ms/step with 128-tiles, 0.272 with 64-tiles)
  if (M >= 256 && N >= 256 && K >= 256) {
    const long a128 = active_tiles(128) * batch;
    if (a128 >= 200) {
      BT = 128;  // enough big tiles to cover the chip
    }
  }
  a.tile = BT;
  const long tiles = (long)hb_cdiv(M, BT) * hb_cdiv(N, BT);
  const long active = active_tiles(BT);
  if (BT == 64 && ws && (active * batch < 320 || K >= 32768) && K >= 128) {
    // few tiles: split the contraction, each slice >= 64 deep.  Deep problems (K >= 2048) want ~3.5 workgroups per
    // CU; shallow ones are dominated by the slab traffic (S slabs written, then read by the finish kernel), so they
    // stop at ~1.25 workgroups per CU
    long s0 = (K >= 2048 ? 896 : 320) / (active * batch);
    // very deep contractions: slabs of ~2048 even when there are plenty of tiles (8 experts x K = 65536: 2.47 ms
    // with 2 slabs, 1.74 ms with 32 -- tools/lbar_probe.py)
    if (K >= 32768 && K / 2048 > s0) s0 = K / 2048;
    const long s1 = K / 64;
    const long s3 = ws_elems / (batch * M * N);
    if (s0 > s1) s0 = s1;
    if (s0 > s3) s0 = s3;
    // (128 slabs for the weight gradients of cfg 4 -- K = 32768 into 64 x 256 -- were tried: the product 16.3 -> 13.5 us,
    //  its finish launch 6.9 -> 11.9 us)
    if (s0 > 64) s0 = 64;
    if (s0 >= 2) S = (int)s0;
  }
  {
    // diagnostic overrides (tools/stride_probe.py): hb_debug_set("mm_force_bt", 64|128), ("mm_force_s", <slabs>)
    const long fbt = hb_debug_get("mm_force_bt", 0), fs = hb_debug_get("mm_force_s", 0);
    if (fbt) {
      BT = fbt == 128 ? 128 : 64;
      a.tile = BT;
    }
    if (fs && ws) {
      long s = fs;
      const long s3 = ws_elems / (batch * M * N);
      if (s > s3) s = s3;
      if (s > K / 16) s = K / 16;
      S = s < 1 ? 1 : (int)s;
    }
  }
  if (S > 8) S -= S % 8;  // slab <-> XCD affinity (see matmul_kernel)
  a.S = S;
  a.to_ws = (S > 1 || (flags & HB_MM_SYM_OUT)) ? 1 : 0;
  {
    // results that would go through slabs + a finish launch: one 32x32 tile per workgroup with the contraction
    // split over its four waves instead (matmul_wgk_kernel)
    constexpr long VEC0 = 16 / sizeof(T);
    const bool aligned0 = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && lda % VEC0 == 0 && ldb % VEC0 == 0 &&
                          sA % VEC0 == 0 && sB % VEC0 == 0;
    if (!colsum && a.to_ws && matmul_wgk_ok<T>(a, M, N, K, batch, flags, aligned0))
      return matmul_wgk_launch<T>(a, transA, transB, stream);
  }
  const long tiles_final = (long)hb_cdiv(M, BT) * hb_cdiv(N, BT);
  HB_REQUIRE(tiles_final * S * batch < 2147483647L, "hb_matmul: grid too large");
  a.bfast = (batch > 1 && batch % 8 == 0) ? 1 : 0;
  dim3 grid = a.bfast ? dim3((unsigned)(tiles_final * S * batch), 1, 1) : dim3((unsigned)(tiles_final * S), (unsigned)batch, 1);
  constexpr long VEC = 16 / sizeof(T);
  const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && lda % VEC == 0 && ldb % VEC == 0 &&
                       sA % VEC == 0 && sB % VEC == 0;
  const bool fast = aligned && K % 16 == 0 && K > 0 && (!transA || M % VEC == 0) && (transB || N % VEC == 0);
  // column sums inside the GEMM: the vector path of the A^T B form, 64-tiles, every slab a whole number of k-steps
  bool colsum_fused = false;
  if (colsum) {
    const long kchunk = (((K + S - 1) / S + 15) / 16) * 16;
    colsum_fused = fast && transA && !transB && batch == 1 && BT == 64 && K % 16 == 0 && (K % kchunk) % 16 == 0;
    if (colsum_fused) {
      a.flags |= HB_MM_COLSUM_B;
      a.colsum = colsum;
    }
  }
#define HB_MM_LAUNCH2(TA_, TB_, F_)                                                                 \
  do {                                                                                              \
    if (BT == 128)                                                                                  \
      hipLaunchKernelGGL((matmul_kernel<T, TA_, TB_, F_, 128>), grid, dim3(256), 0, stream, a);     \
    else                                                                                            \
      hipLaunchKernelGGL((matmul_kernel<T, TA_, TB_, F_, 64>), grid, dim3(256), 0, stream, a);      \
  } while (0)
#define HB_MM_LAUNCH(TA_, TB_)            \
  do {                                    \
    if (fast)                             \
      HB_MM_LAUNCH2(TA_, TB_, true);      \
    else                                  \
      HB_MM_LAUNCH2(TA_, TB_, false);     \
  } while (0)
  if (!transA && !transB)
    HB_MM_LAUNCH(false, false);
  else if (!transA && transB)
    HB_MM_LAUNCH(false, true);
  else if (transA && !transB)
    HB_MM_LAUNCH(true, false);
  else
    HB_MM_LAUNCH(true, true);
#undef HB_MM_LAUNCH
#undef HB_MM_LAUNCH2
  HB_LAUNCH_CHECK();
  if (a.to_ws) {
    hipLaunchKernelGGL(matmul_splitk_finish_kernel<T>, dim3(hb_stream_grid(batch * M * N, 256)), dim3(256), 0, stream,
                       a);
    HB_LAUNCH_CHECK();
  }
  if (colsum && !colsum_fused) {
    // shapes the fused form does not take: a reduction launch of its own (sequential on the stream: ws is free again)
    if (sizeof(T) == 4)
      return hb_reduce_f32(HB_RED_SUM, (const float*)B, (float*)colsum, 1, K, N, (float*)ws, ws_elems, stream);
    return hb_reduce_f64(HB_RED_SUM, (const double*)B, (double*)colsum, 1, K, N, (double*)ws, ws_elems, stream);
  }
  return 0;
}

extern "C" int hb_matmul_colsum_f32(const float* A, const float* B, float* C, float* colsum, long M, long N, long K, long lda,
                                    long ldb, long ldc, float* ws, long ws_elems, void* stream) {
  HB_REQUIRE(colsum && ldb == N, "hb_matmul_colsum: colsum is NULL or B is not contiguous (ldb != N)");
  return matmul_launch<float>(A, B, C, 1, M, N, K, lda, ldb, ldc, 0, 0, 0, 1, 0, 1.0, 0.0, nullptr, 0, HB_ACT_NONE, 0, ws,
                              ws_elems, (hipStream_t)stream, colsum);
}
extern "C" int hb_matmul_colsum_f64(const double* A, const double* B, double* C, double* colsum, long M, long N, long K,
                                    long lda, long ldb, long ldc, double* ws, long ws_elems, void* stream) {
  HB_REQUIRE(colsum && ldb == N, "hb_matmul_colsum: colsum is NULL or B is not contiguous (ldb != N)");
  return matmul_launch<double>(A, B, C, 1, M, N, K, lda, ldb, ldc, 0, 0, 0, 1, 0, 1.0, 0.0, nullptr, 0, HB_ACT_NONE, 0, ws,
                               ws_elems, (hipStream_t)stream, colsum);
}

extern "C" int hb_matmul_f32(const float* A, const float* B, float* C, long batch, long M, long N, long K, long lda,
                             long ldb, long ldc, long sA, long sB, long sC, int transA, int transB, double alpha,
                             double beta, const float* bias, long sBias, int act, int flags, float* ws, long ws_elems,
                             void* stream) {
  return matmul_launch<float>(A, B, C, batch, M, N, K, lda, ldb, ldc, sA, sB, sC, transA, transB, alpha, beta, bias,
                              sBias, act, flags, ws, ws_elems, (hipStream_t)stream);
}
// The square product whose result is Kbar of a Gram matrix K(X, X), with that Gram matrix's VJP in its epilogue (MmGramVjp).
extern "C" int hb_matmul_gram_vjp_ok(long M, long K, long batch, long d) {
  MmArgs<float> t = {};
  if (hb_debug_get("mm_no_gram_vjp", 0) != 0 || d < 1 || d > HB_MMG_MAXD) return 0;
  return matmul_wgk_ok<float>(t, M, M, K, batch, 0, true) ? 1 : 0;
}
extern "C" long hb_matmul_gram_vjp_ws_elems(long batch, long M, long d) { return batch * (M / 32) * (M / 32) * 32 * 2 * d; }
extern "C" int hb_matmul_gram_vjp_f32(const float* A, const float* B, float* C, long batch, long M, long K, long lda, long ldb,
                                      long ldc, long sA, long sB, long sC, int transA, int transB, const float* X, long sX,
                                      const float* ell, long sEll, long dl, long d, float* Xbar, float* ell_partial,
                                      float* part, unsigned* counters, void* stream) {
  HB_REQUIRE(A && B && C && X && ell && Xbar && ell_partial && part && counters, "hb_matmul_gram_vjp: NULL pointer");
  HB_REQUIRE(hb_matmul_gram_vjp_ok(M, K, batch, d), "hb_matmul_gram_vjp: shape outside the in-workgroup split-K form (hb_matmul_gram_vjp_ok)");
  HB_REQUIRE(dl == 1 || dl == d, "hb_matmul_gram_vjp: lengthscales must have 1 or d entries");
  const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0;
  HB_REQUIRE(aligned, "hb_matmul_gram_vjp: operands must be 16-byte aligned with leading dimensions %% 4 == 0");
  HB_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : M) && ldc >= M, "hb_matmul_gram_vjp: leading dimension too small");
  MmArgs<float> a = {};
  a.A = A, a.B = B, a.C = C;
  a.M = M, a.N = M, a.K = K, a.lda = lda, a.ldb = ldb, a.ldc = ldc, a.sA = sA, a.sB = sB, a.sC = sC, a.batch = batch;
  a.alpha = 1.f, a.beta = 0.f, a.act = HB_ACT_NONE, a.flags = 0;
  MmGramVjp gv;
  gv.X = X, gv.sX = sX, gv.ell = ell, gv.sEll = sEll, gv.dl = dl, gv.d = d;
  gv.Xbar = Xbar, gv.ell_partial = ell_partial, gv.part = part, gv.counters = counters;
  return matmul_wgk_launch_g(a, transA, transB, (hipStream_t)stream, gv);
}

// The Gaussian likelihood head of a MatBias layer (reference nn.py:31-32 feeding densities.py:25-27 under tf.reduce_sum):
// f = A B + bias is consumed in the epilogue of the row-streaming product, never written (EPI 5 above).
static inline bool matmul_gauss_shape(long n, long K, long N, int& G, int& KC) {
  if (hb_debug_get("mm_no_gauss_head", 0) != 0) return false;
  if (n < 2048 || N <= 32 || N > 256 || n * N >= 2147483647L) return false;
  MmArgs<float> t = {};
  t.N = N, t.K = K;
  return matmul_rowsreg_ok(t, G, KC);
}
extern "C" long hb_matmul_gauss_units(long n, long K, long N) {
  int G, KC;
  if (!matmul_gauss_shape(n, K, N, G, KC)) return 0;
  return 4 * matmul_rowsreg_wgs(n, N, KC);
}
extern "C" int hb_matmul_gauss_f32(const float* A, long lda, const float* B, long ldb, const float* bias, const float* y,
                                   const float* scale, const float* var, double post, float* dmu, float* fbar, float* part,
                                   long units, long n, long K, long N, void* stream) {
  int G, KC;
  HB_REQUIRE(A && B && y && var && dmu && part, "hb_matmul_gauss: NULL pointer");
  HB_REQUIRE(matmul_gauss_shape(n, K, N, G, KC) && units == 4 * matmul_rowsreg_wgs(n, N, KC),
             "hb_matmul_gauss: units must be hb_matmul_gauss_units(n, K, N) > 0 (n = %ld, K = %ld, N = %ld)", n, K, N);
  HB_REQUIRE((uintptr_t)A % 16 == 0 && lda % 4 == 0 && lda >= K && ldb >= N, "hb_matmul_gauss: A must be 16-byte aligned with lda %% 4 == 0");
  MmArgs<float> a = {};
  a.A = A, a.B = B, a.C = dmu, a.bias = bias;
  a.M = n, a.N = N, a.K = K, a.lda = lda, a.ldb = ldb, a.ldc = N;
  a.batch = 1, a.alpha = 1.f, a.beta = 0.f, a.act = HB_ACT_NONE, a.flags = 0;
  MmHead hd;
  hd.y = y, hd.scale = scale, hd.var = var, hd.post = (float)post, hd.fbar = fbar, hd.part = part, hd.units = units;
  return matmul_rowsreg_launch(a, 0, G, KC, (hipStream_t)stream, &hd);
}

extern "C" int hb_matmul_f64(const double* A, const double* B, double* C, long batch, long M, long N, long K,
                             long lda, long ldb, long ldc, long sA, long sB, long sC, int transA, int transB,
                             double alpha, double beta, const double* bias, long sBias, int act, int flags, double* ws,
                             long ws_elems, void* stream) {
  return matmul_launch<double>(A, B, C, batch, M, N, K, lda, ldb, ldc, sA, sB, sC, transA, transB, alpha, beta, bias,
                               sBias, act, flags, ws, ws_elems, (hipStream_t)stream);
}

// Finishing pass of the factorisation: clears the strict upper triangles of L (and W = L^-1), and -- when `Wf` is
// given (M % 32 == 0) -- also leaves two FRAGMENT-MAJOR copies of W for the M^2 n contractions (csrc/sgp.hip):
//   Wf  [B][M/32 row tiles][M/32 k chunks][4][64 lanes][4]:  element (t, Q, v, lane = (li, h), s) = W [32t+li][32Q+16h+4v+s]
//   WTf (B*M*M elements further on): the same with W^T,                                        = W [32Q+16h+4v+s][32t+li]
// i.e. exactly the order in which the MFMA A-operand loads of a 32-row tile consume a 32-deep k chunk: load
// instruction v of a wave reads 64 lanes x 16 B = ONE contiguous kilobyte (8 whole cache lines).  Read from the
// row-major matrix the same instruction touches 32 different lines for 32 bytes each, and the CU's texture
// addresser / L1 -- 64 B per clock -- was as busy as the matrix pipes (profiles/r01_strip_ablation.txt: loads alone
// 12.4 us against MFMAs alone 16.3 us).

template <typename T>
__global__ void __launch_bounds__(256) tril_inplace_kernel(T* __restrict__ L, T* __restrict__ W, T* __restrict__ Wf, int bf16x3,
                                                           long B, long M) {
  const int Mi = (int)M;
  const long mm = M * M, total = B * mm;
  const long stride = (long)gridDim.x * blockDim.x;
  const int nT = Mi / 32;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long b = t / mm;
    const int rem = (int)(t - b * mm);
    const int i = rem / Mi, j = rem - i * Mi;
    if (j > i) {
      L[t] = T(0);
      if (W) W[t] = T(0);
    }
    if (Wf) {
      // output-major: this thread writes element `rem` of both fragment images of matrix b
      const T* Wb = W + b * mm;
      {
        const int s = rem & 3, lane = (rem >> 2) & 63, v = (rem >> 8) & 3, blk = rem >> 10;
        const int Q = blk % nT, tt = blk / nT, li = lane & 31, h = lane >> 5;
        const int r = 32 * tt + li, k = 32 * Q + 16 * h + 4 * v + s;
        Wf[t] = k <= r ? Wb[(long)r * Mi + k] : T(0);           // W  [r][k]   (lower triangular)
        Wf[total + t] = r <= k ? Wb[(long)k * Mi + r] : T(0);   // W^T[r][k] = W[k][r]
      }
      if (bf16x3) {
        // bf16 images, one per split term p: [p][B][t][Q][q = 0..1][64 lanes][8] with the element
        // X[32t + li][32Q + 16q + 8h + j] -- the A-operand fragment of v_mfma_f32_32x32x16_bf16 for k16-step q
        const int jj = rem & 7, lane = (rem >> 3) & 63, q = (rem >> 9) & 1, blk = rem >> 10;
        const int Q = blk % nT, tt = blk / nT, li = lane & 31, h = lane >> 5;
        const int r = 32 * tt + li, k = 32 * Q + 16 * q + 8 * h + jj;
        __bf16* W3 = reinterpret_cast<__bf16*>(Wf + 2 * total);   // 3 planes of W, then 3 planes of W^T
        const float x = k <= r ? (float)Wb[(long)r * Mi + k] : 0.f;
        const float xt = r <= k ? (float)Wb[(long)k * Mi + r] : 0.f;
        __bf16 a0, a1, a2;
        hb_split_bf16x3(x, a0, a1, a2);
        W3[t] = a0, W3[total + t] = a1, W3[2 * total + t] = a2;
        hb_split_bf16x3(xt, a0, a1, a2);
        W3[3 * total + t] = a0, W3[4 * total + t] = a1, W3[5 * total + t] = a2;
      }
    }
  }
}

#define CH_NB 32
#define CH_LD (CH_NB + 4)  // LDS row stride: keeps rows 16-byte aligned for vector reads

// x L^T = t for one row per thread (x returned in t): right-looking forward
// substitution, so the 31-k updates of step k are independent FMAs; column k
// of L comes from row k of the transposed LDS copy by 16-byte uniform reads.
template <typename T>
__device__ __forceinline__ void trsolve_row32(T (&t)[CH_NB], const T (*LsT)[CH_LD], const T* invd) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(VEC)));
  T iv[CH_NB];
#pragma unroll
  for (int j = 0; j < CH_NB; j += VEC) {
    const VT v = *reinterpret_cast<const VT*>(&invd[j]);
#pragma unroll
    for (int q = 0; q < VEC; ++q) iv[j + q] = v[q];
  }
#pragma unroll
  for (int k = 0; k < CH_NB; ++k) {
    const T xk = t[k] * iv[k];
    t[k] = xk;
    T lcol[CH_NB];
#pragma unroll
    for (int c = ((k + 1) / VEC) * VEC; c < CH_NB; c += VEC) {
      const VT v = *reinterpret_cast<const VT*>(&LsT[k][c]);
#pragma unroll
      for (int q = 0; q < VEC; ++q) lcol[c + q] = v[q];
    }
#pragma unroll
    for (int c = k + 1; c < CH_NB; ++c) t[c] -= xk * lcol[c];
  }
}

// ===========================================================================
// Cholesky, right-looking: one launch per 32-column block k; every launch is
// a single global-load round trip deep.
//
//   launch k:  every remaining tile (i, j >= k) takes the rank-32 update by
//              panel k-1,  T_ij -= L[i, k-1] L[j, k-1]^T   (16 f32 MFMAs / wave,
//              operands straight from global memory into MFMA fragments -- the
//              contraction index is permuted so that each lane reads 64
//              contiguous bytes);  the workgroups of block column k then factor
//              it: the 32x32 diagonal block (recomputed by each of them, cheaper
//              than a cross-workgroup hand-off) and 96 rows below it, stacked
//              in LDS and processed in four 8-column steps:
//                potrf8   8x8 diagonal block factored redundantly by every lane
//                         in registers (no cross-lane traffic, no barriers);
//                solve    one thread per row, 8 columns (28 FMAs);
//                update   the remaining columns by rank-8 MFMA, accumulators
//                         staying in registers across the four steps.
//
// The whole trailing matrix is rewritten by every launch (11 MB in total at
// M = 512: noise), which spreads the O(M^3) work over all CUs and leaves
// load -> 16 MFMAs -> potrf/solve -> store on the critical path of a launch
// instead of a K = j0 deep contraction on four CUs (the left-looking form this
// replaces: 17.4 us average per panel, profiles/r01_chol_panel_phase_stamps.txt).
//
// Storage: results go to L.  A tile is read from A on its first touch (k <= 1)
// and from L afterwards.  Until its column is factored, a DIAGONAL tile (j, j)
// lives in the unused upper tile (j-1, j): the factor workgroups of launch j all
// read it while one of them writes the factored block to (j, j).  The upper
// triangle is cleared by tril_inplace_kernel after the last launch.
// ===========================================================================
// In-kernel phase stamps: compiled in only by tools/chol_stamps.hip (diagnostic build).
#ifndef HB_STAMP
#define HB_STAMP(i)
#endif
#ifndef HB_WSTAMP
#define HB_WSTAMP(w, lane, i)   // per-wave stamps of the in-panel chain (tools/chol_stamps.hip)
#define HB_PSTAMP(i)            // wave 0's stamps inside the load / update phase
#endif
#define CR_B 32         // block size
#define CR_ROWS 128     // stacked panel: diagonal block + 96 rows
#define CR_LD 36        // LDS row stride (rows stay 16-byte aligned)
#define CR_FROWS 96     // rows below the diagonal block per factor workgroup

// Tile lists of launch k (inv: the inverse rides along, see below).
//   factor column k : wave 0 of every workgroup takes the diagonal block; the other three
//                     waves walk  [A blocks k+1 .. nblk-1] ++ [Y blocks 0 .. k]
//   update column j : four waves walk  [A blocks j .. nblk-1] ++ [Y blocks 0 .. k-1]
static inline int chol_rl_factor_strips(int nblk, int k, int inv) {
  const int n = (nblk - k - 1) + (inv ? k + 1 : 0);
  return n > 0 ? (n + 2) / 3 : 1;
}
static inline int chol_rl_grid(int nblk, int k, int inv) {
  int g = chol_rl_factor_strips(nblk, k, inv);
  if (k > 0)
    for (int j = k + 1; j < nblk; ++j) g += (nblk - j + (inv ? k : 0) + 3) / 4;
  return g;
}

// The inverse for free: run the same elimination on the stacked matrix [A; I].
// Its lower half converges to Y = L^-T (block recurrence
//   Y_ck = (I_ck - sum_{j<k} Y_cj L_kj^T) L_kk^-T ),
// i.e. the rows of the identity are just more rows "below the diagonal block":
// they take the same rank-32 updates (a-operand from Y instead of L) and the
// same in-LDS solves.  Y is upper block-triangular: block row c joins at launch
// c (tile (c,c) starts as I) and tile (c,j) is first touched -- from zero -- at
// launch c+1.  The working copy Y lives in a caller workspace (row-major, so the
// update operands stay 64-byte contiguous per lane); every finished panel is also
// written transposed into W = L^-1.  All of it is extra width per launch, none of
// it is on the critical path, and hb_trinv's ten dependent launches disappear.
template <typename T, bool FAST>
__global__ void __launch_bounds__(256) chol_rl_kernel(const T* __restrict__ Ain, T* __restrict__ L, T* __restrict__ Y,
                                                      T* __restrict__ W, int M, int k, int* __restrict__ info) {
  typedef Mma<T> MM;
  constexpr int TM = MM::TM;        // 32 (f32) | 16 (f64)
  constexpr int KH = 64 / TM;       // lane groups along the contraction index: 2 | 4
  constexpr int RT = CR_B / TM;     // MFMA tiles per 32: 1 | 2
  constexpr int CK = CR_B / KH;     // contraction entries per lane, rank-32 update: 16 | 8   (64 bytes)
  constexpr int PK = 8 / KH;        // contraction entries per lane, rank-8 update:   4 | 2   (16 bytes)
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef T VT __attribute__((ext_vector_type(VEC)));
  __shared__ __attribute__((aligned(16))) T Cs[CR_ROWS][CR_LD];
  __shared__ __attribute__((aligned(16))) T Dg[8][8];  // snapshot of the step's 8x8 diagonal sub-block (see chol_rl64_kernel)

  const long boff = (long)blockIdx.y * M * M;
  Ain += boff;
  L += boff;
  const bool inv = Y != nullptr;
  if (inv) {
    Y += boff;
    W += boff;
  }
  info += blockIdx.y;
  const int nblk = (M + CR_B - 1) / CR_B;
  // workgroup -> (block column j, strip s)
  int j = k, s = blockIdx.x;
  {
    const int nf = (nblk - k - 1) + (inv ? k + 1 : 0);
    int cnt = nf > 0 ? (nf + 2) / 3 : 1;
    while (s >= cnt) {
      s -= cnt;
      ++j;
      cnt = (nblk - j + (inv ? k : 0) + 3) / 4;
    }
  }
  const bool factor = (j == k);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane % TM, h = lane / TM;
  // entry g of this workgroup's tile list -> (live, is a Y tile, block row)
  auto describe = [&](int g, bool& live_, bool& yt_, int& rb_) {
    const int nA = factor ? nblk - k - 1 : nblk - j;
    const int nY = inv ? (factor ? k + 1 : k) : 0;
    const int e = factor ? 3 * s + g - 1 : 4 * s + g;
    if (factor && g == 0) {
      live_ = true, yt_ = false, rb_ = k;
    } else if (e < nA) {
      live_ = true, yt_ = false, rb_ = (factor ? k + 1 : j) + e;
    } else if (e < nA + nY) {
      live_ = true, yt_ = true, rb_ = e - nA;
    } else {
      live_ = false, yt_ = false, rb_ = nblk - 1;  // idle wave: recompute a valid tile, store nothing
    }
  };
  bool live, yt;
  int bi;
  describe(w, live, yt, bi);
  const int Mm1 = M - 1;
  const int row0 = bi * CR_B, col0 = j * CR_B;
  const bool ydiag = yt && bi == j;  // tile (k,k) of Y: starts as the identity, takes no update

  HB_STAMP(0);
  typename MM::Acc acc[RT][RT];
  {
    // where the tile currently lives: A on first touch (k <= 1), then L -- a diagonal tile one block up -- or Y
    const T* src = yt ? Y : ((k <= 1) ? Ain : L);
    const int hrow0 = (!yt && k >= 2 && bi == j) ? (j - 1) * CR_B : row0;
    const bool fresh = yt && (ydiag || k == bi + 1);
#pragma unroll
    for (int si = 0; si < RT; ++si)
#pragma unroll
      for (int sj = 0; sj < RT; ++sj)
#pragma unroll
        for (int r = 0; r < MM::NACC; ++r) {
          const int tr = si * TM + MM::acc_row(lane, r), tc = sj * TM + MM::acc_col(lane);
          int gr = hrow0 + tr, gc = col0 + tc;
          if (!FAST) {
            // clamp by the TRUE row (the home of a diagonal tile is shifted up by one block)
            gr = (row0 + tr) < M ? gr : hrow0 + (Mm1 - row0);
            gc = gc < M ? gc : Mm1;
          }
          const T init = (ydiag && tr == tc) ? T(1) : T(0);
          // unconditional load + select (a fresh Y tile reads whatever its workspace holds and discards it): a
          // conditional load compiles to a branch and an s_waitcnt vmcnt(0) per element -- serialised round trips
          const T ld = src[gr * M + gc];
          acc[si][sj][r] = fresh ? init : ld;
        }
  }

  if (k > 0 && !ydiag) {
    // rank-32 update by panel k-1; lane (li, h) contracts over entries [h*CK, (h+1)*CK) of the panel row
    const int pc = (k - 1) * CR_B + h * CK;
    const T* asrc = yt ? Y : L;
    T a[RT][CK], bq[RT][CK];
#pragma unroll
    for (int si = 0; si < RT; ++si) {
      int ra = row0 + si * TM + li, rb = col0 + si * TM + li;
      if (!FAST) {
        ra = ra < M ? ra : Mm1;
        rb = rb < M ? rb : Mm1;
      }
      const T* pa = asrc + ra * M + pc;
      const T* pb = L + rb * M + pc;
      if (FAST) {
#pragma unroll
        for (int q = 0; q < CK; q += VEC) {
          const VT va = *reinterpret_cast<const VT*>(pa + q);
          const VT vb = *reinterpret_cast<const VT*>(pb + q);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            a[si][q + e] = -va[e];
            bq[si][q + e] = vb[e];
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < CK; ++q) {
          a[si][q] = -pa[q];
          bq[si][q] = pb[q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CK; ++q)
#pragma unroll
      for (int si = 0; si < RT; ++si)
#pragma unroll
        for (int sj = 0; sj < RT; ++sj) acc[si][sj] = MM::mma(a[si][q], bq[sj][q], acc[si][sj]);
  }

  HB_STAMP(1);
  if (!factor) {
    if (live) {
      T* dst = yt ? Y : L;
      const int wrow0 = (!yt && bi == j) ? (j - 1) * CR_B : row0;
#pragma unroll
      for (int si = 0; si < RT; ++si)
#pragma unroll
        for (int sj = 0; sj < RT; ++sj)
#pragma unroll
          for (int r = 0; r < MM::NACC; ++r) {
            const int tr = si * TM + MM::acc_row(lane, r), tc = sj * TM + MM::acc_col(lane);
            if (FAST || ((row0 + tr) < M && (col0 + tc) < M)) dst[(wrow0 + tr) * M + col0 + tc] = acc[si][sj][r];
          }
    }
    return;
  }

  // ---- factor block column k: stacked panel rows [32w, 32w+32) belong to wave w ----
  const int nb = (M - k * CR_B) < CR_B ? (M - k * CR_B) : CR_B;
  int fail = 0;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    // publish columns [8kb, 8kb+8) of the accumulators
#pragma unroll
    for (int si = 0; si < RT; ++si)
#pragma unroll
      for (int sj = 0; sj < RT; ++sj) {
        const int tc = sj * TM + MM::acc_col(lane);
        if ((tc >> 3) == kb) {
#pragma unroll
          for (int r = 0; r < MM::NACC; ++r) {
            const int tr = si * TM + MM::acc_row(lane, r);
            T v = acc[si][sj][r];
            if (!FAST && w == 0 && (tr >= nb || tc >= nb)) v = (tr == tc) ? T(1) : T(0);  // identity padding
            Cs[w * CR_B + tr][tc] = v;
            if (((w * CR_B + tr) >> 3) == kb) Dg[(w * CR_B + tr) & 7][tc & 7] = v;
          }
        }
      }
    __syncthreads();
    if (kb == 1) HB_STAMP(4);
    if (tid < CR_ROWS) {
      // potrf8: every lane (of the two waves that own rows) factors the 8x8 diagonal block
      // Cs[8kb.., 8kb..] in registers: no cross-lane traffic, no barriers.  The block is held by
      // COLUMNS, two rows per register pair, so that the rank-1 updates (and the row solve below,
      // which walks the same columns) are packed v_pk_fma_f32 on fp32.
      typedef T T2 __attribute__((ext_vector_type(2)));
      T2 col[8][4];  // col[c][h] = (L[2h][c], L[2h+1][c])
      T pinv[8];
      {
        T p[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int q = 0; q < 8; q += VEC) {
            const VT v = *reinterpret_cast<const VT*>(&Dg[i][q]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) p[i][q + e] = v[e];
          }
        // only the lower triangle is meaningful; mirror it so the pairs hold finite values
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            col[c][h][0] = (2 * h >= c) ? p[2 * h][c] : p[c][2 * h];
            col[c][h][1] = (2 * h + 1 >= c) ? p[2 * h + 1][c] : p[c][2 * h + 1];
          }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const T d = col[c][c / 2][c % 2];
        if (fail == 0 && !(d > T(0))) fail = 8 * kb + c + 1;
        T lcc, pi;
        pivot_sqrt(d, lcc, pi);
        pinv[c] = pi;
        const T2 pi2 = {pi, pi};
#pragma unroll
        for (int h = c / 2; h < 4; ++h) col[c][h] *= pi2;
        col[c][c / 2][c % 2] = lcc;
#pragma unroll
        for (int c2 = c + 1; c2 < 8; ++c2) {
          const T s = col[c][c2 / 2][c2 % 2];
          const T2 ns = {-s, -s};
#pragma unroll
          for (int h = c2 / 2; h < 4; ++h) col[c2][h] = __builtin_elementwise_fma(col[c][h], ns, col[c2][h]);
        }
      }
      if (kb == 1) HB_STAMP(5);
      // solve: thread t owns stacked row t;  x L_kk^T = c  over the 8 columns of this step, right-looking:
      // once x_c is final, x_{c2 > c} -= x_c L[c2][c] -- the same column pairs as above
      T2 xv[4];
      static_assert(VEC % 2 == 0, "pairs");
#pragma unroll
      for (int q = 0; q < 8; q += VEC) {
        const VT v = *reinterpret_cast<const VT*>(&Cs[tid][8 * kb + q]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) xv[(q + e) / 2][(q + e) % 2] = v[e];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const T xc = xv[c / 2][c % 2] * pinv[c];
        const T2 nx = {-xc, -xc};
#pragma unroll
        for (int h = c / 2; h < 4; ++h) xv[h] = __builtin_elementwise_fma(col[c][h], nx, xv[h]);
        xv[c / 2][c % 2] = xc;
      }
#pragma unroll
      for (int q = 0; q < 8; q += VEC) {
        VT v;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = xv[(q + e) / 2][(q + e) % 2];
        *reinterpret_cast<VT*>(&Cs[tid][8 * kb + q]) = v;
      }
    }
    __syncthreads();
    if (kb == 1) HB_STAMP(6);
    if (kb < 3) {
      // rank-8 update of the columns to the right: acc -= X[:, 8kb:8kb+8] X_D[cols, 8kb:8kb+8]^T
      T a2[RT][PK], b2[RT][PK];
#pragma unroll
      for (int si = 0; si < RT; ++si) {
        const int dr = si * TM + li;  // row of the diagonal block = column of the panel
        const bool right = dr >= 8 * (kb + 1);
        static_assert(PK == VEC, "one 16-byte LDS read per fragment");
        const VT va = *reinterpret_cast<const VT*>(&Cs[w * CR_B + si * TM + li][8 * kb + h * PK]);
        const VT vb = *reinterpret_cast<const VT*>(&Cs[dr][8 * kb + h * PK]);
#pragma unroll
        for (int e = 0; e < PK; ++e) {
          a2[si][e] = -va[e];
          b2[si][e] = right ? vb[e] : T(0);
        }
      }
#pragma unroll
      for (int e = 0; e < PK; ++e)
#pragma unroll
        for (int si = 0; si < RT; ++si)
#pragma unroll
          for (int sj = 0; sj < RT; ++sj) acc[si][sj] = MM::mma(a2[si][e], b2[sj][e], acc[si][sj]);
    }
    if (kb == 1) HB_STAMP(7);
  }
  HB_STAMP(2);
  if (s == 0 && tid == 0) {
    // block 0 of the factor column is the only writer of info; launch 0 resets it
    const int bad = (fail != 0 && fail <= nb) ? k * CR_B + fail : 0;
    if (k == 0)
      *info = bad;
    else if (bad != 0 && *info == 0)
      *info = bad;
  }
  // store the stacked panel, 32 rows (one wave's tile) at a time
  constexpr int VPR = CR_B / VEC;  // 16-byte groups per row
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {
    bool glive, gy;
    int gb;
    describe(g, glive, gy, gb);
    if (!glive || (g == 0 && s != 0)) continue;  // the diagonal block is written by strip 0 only
    for (int idx = tid; idx < CR_B * VPR; idx += 256) {
      const int pr = idx / VPR, c = (idx % VPR) * VEC;
      const int gr = gb * CR_B + pr;
      if (gr >= M) continue;
      VT v = *reinterpret_cast<const VT*>(&Cs[g * CR_B + pr][c]);
      if (g == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (c + e > pr) v[e] = T(0);  // strict upper part of the diagonal block
      }
      T* dst = (gy ? Y : L) + gr * M + k * CR_B + c;
      if (FAST) {
        *reinterpret_cast<VT*>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (c + e < nb) dst[e] = v[e];
      }
    }
    if (gy) {
      // the finished panel of Y = L^-T, transposed into W = L^-1:  W[32k + c][32 gb + r] = Y[32 gb + r][32k + c]
      for (int idx = tid; idx < CR_B * CR_B; idx += 256) {
        const int r = idx % CR_B, c = idx / CR_B;
        if (c < nb && gb * CR_B + r < M) W[(k * CR_B + c) * M + gb * CR_B + r] = Cs[g * CR_B + r][c];
      }
    }
  }
  HB_STAMP(3);
}

// ===========================================================================
// 64-column launches (fp32, M % 64 == 0): the same right-looking algorithm with two 32-column blocks per launch
// (half the kernel boundaries, cold-start load round trips and panel stores on the critical path).
// A wave tile is 32 rows x 64 columns (two 32x32 accumulators); a factor workgroup stacks the 64x64 diagonal block
// (waves 0, 1) and 64 more rows (waves 2, 3).  Row tiles are 32 rows (index rt), column blocks 64 wide (index j);
// the diagonal block of column j (row tiles 2j, 2j+1) lives one block up, in the unused upper block (j-1, j),
// until it is factored.
//
// Round 3: the in-panel phase of a factor workgroup is a chain of INDEPENDENT waves, not a barrier-stepped loop.
//   * The accumulators of a factor workgroup are TRANSPOSED tiles (the update computes T^T = B X^T instead of X B^T):
//     the 32x32 accumulator layout then puts the stacked ROW on the lane (l & 31) and the panel columns in the
//     registers -- lane (i, h) holds columns {0-3, 8-11, 16-19, 24-27} + 4h of row i -- so the 8 entries of a row that
//     an 8-column elimination step needs are the lane's own registers plus four v_permlane32_swap with its
//     half-wave partner.  No LDS publish, no barrier, in front of a step.
//   * The 64 x 64 diagonal block is factored by its two waves alone: wave 0 owns rows 0..31 and is the PIVOT wave of
//     steps 0..3, wave 1 owns rows 32..63 (pivot wave of steps 4..7).  A pivot wave eliminates through v_readlane as
//     before (its pivot rows are its own lanes), writes its 32 solved rows and the 8 reciprocal pivots to LDS, raises
//     a step counter -- and continues with the rank-8 update of ITS OWN tile straight from registers (the A and B
//     operands of that update are both the wave's own solved entries).  Its loop never reads LDS and never waits:
//     ~1000 cycles per 8 columns (elimination ~700 + 4 dependent MFMAs) against ~1940 for the barrier-stepped form.
//   * Every other wave FOLLOWS: it waits (LDS counter, s_sleep poll) until the step it needs is published, solves
//     its rows by substitution with the published 8 x 8 sub-block (uniform LDS reads, the same operation order as
//     the pivot wave's, so results do not depend on which wave solved a row), writes them to its own LDS rows for the
//     final store and updates its tiles with the published diagonal rows as the MFMA A operand.  Followers lag the
//     pivot wave by a step or two and finish ~1000 cycles after it; they never hold it back: the counters are
//     monotonic, a published entry is never rewritten, and nobody a wave waits for ever waits for that wave
//     (diag wave 1 waits for wave 0 only; row waves wait for diag waves only), so the polls cannot deadlock.
// ===========================================================================
#define C64_NB 64
#define C64_ROWS 128
#define C64_LD 68

static inline int chol64_factor_strips(int nrt, int k, int inv) {
  const int n = (nrt - 2 * (k + 1)) + (inv ? 2 * (k + 1) : 0);
  return n > 0 ? (n + 1) / 2 : 1;
}
static inline int chol64_grid(int nrt, int k, int inv) {
  int g = chol64_factor_strips(nrt, k, inv);
  if (k > 0)
    for (int j = k + 1; j < nrt / 2; ++j) g += (nrt - 2 * j + (inv ? 2 * k : 0) + 3) / 4;
  return g;
}

// steps published by a diagonal wave: monotonic counter in LDS, release / acquire at workgroup scope
__device__ __forceinline__ void c64_publish(int* flag, int steps) {
  __hip_atomic_store(flag, steps, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// `seen`: the largest count this wave has read so far.  The counters only grow and a follower usually runs a step or two
// behind, so most waits are answered by the cached value and cost no LDS round trip.
__device__ __forceinline__ void c64_wait(int* flag, int steps, int& seen) {
  if (seen >= steps) return;
  while ((seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < steps) __builtin_amdgcn_s_sleep(1);
}

// One 8-column in-panel step (columns [8 KB, 8 KB + 8) of the 64-column block) of one wave of a factor workgroup; every
// index below is a compile-time constant, so the accumulators and the row entries stay in registers.
template <int KB>
__device__ __forceinline__ void c64_step(Mma<float>::Acc (&acc)[2], float (*Cs)[C64_LD], float (*L8s)[8][8], int* steps_done,
                                         const int w, const int lane, int& fail, int (&seen)[2], const int k) {
  (void)k;   // (the launch index: only the diagnostic stamps use it)
  typedef float T;
  typedef Mma<float> MM;
  typedef float VT __attribute__((ext_vector_type(4)));
  typedef float V3 __attribute__((ext_vector_type(3)));
  typedef float V2 __attribute__((ext_vector_type(2)));
  constexpr int J = KB >> 2, sub = KB & 3, c0 = 8 * KB;
  const int li = lane & 31, h = lane >> 5;
  auto bcast = [&](T v, int l) { return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
  // the 8 entries of this lane's row in columns [c0, c0+8): registers 4 sub .. 4 sub + 3 of both half-waves
  T x[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // (a float temporary first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 with this
    // compiler; and the swap as inline asm with both operands tied: given the same value twice, the builtin form
    // returned its first result for both -- tools/chol_debug.hip probes the instruction itself)
    const T tq = acc[J][4 * sub + q];
    unsigned lo = __builtin_bit_cast(unsigned, tq), hi = lo;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
    x[q] = __builtin_bit_cast(T, lo);      // [lower half's register | lower half's register]: columns c0 + q
    x[4 + q] = __builtin_bit_cast(T, hi);  // [upper half's          | upper half's         ]: columns c0 + 4 + q
  }
  const bool pivot = (w == J);
  if (pivot) {
    // PIVOT wave: rows 8 sub .. 8 sub + 7 of this wave are the step's diagonal rows; the pivot and the multipliers
    // L[c2][c] (= diagonal row c2's finished x[c]) reach every lane through v_readlane (uniform values in SGPRs) --
    // the diagonal rows' own solves ARE the factorisation of the 8 x 8 sub-block
    constexpr int dl0 = 8 * sub;
    T pis[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const T d = bcast(x[c], dl0 + c);
      T lcc, pi;
      pivot_sqrt(d, lcc, pi);
      (void)lcc;
      pis[c] = pi;
      x[c] *= pi;
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2) x[c2] = __builtin_fmaf(-x[c], bcast(x[c], dl0 + c2), x[c2]);
    }
    // A pivot d <= 0 (or NaN) makes l_cc = d * rsq(d) a NaN, and a NaN column poisons every later pivot: the last
    // diagonal entry of a wave's last pivot step is NaN exactly when some pivot of the wave (or an earlier one)
    // failed.  One check per wave; the failing column is then the first NaN on the diagonal (LAPACK's info), which
    // the wave reads back from its own rows of the LDS panel.
    if (sub == 3) {
      const T l77 = bcast(x[7], dl0 + 7);
      if (!(l77 == l77)) fail = -1;   // located after the wave's last step (c64_locate_failure)
    }
    if (h == 0) {
      const VT v0 = {x[0], x[1], x[2], x[3]}, v1 = {x[4], x[5], x[6], x[7]};
      *reinterpret_cast<VT*>(&Cs[w * 32 + li][c0]) = v0;
      *reinterpret_cast<VT*>(&Cs[w * 32 + li][c0 + 4]) = v1;
      if (li >= dl0 && li < dl0 + 8) {
        // the 8 x 8 sub-block, compact, for the followers: row c2 = (L[c2][0 .. c2-1], 1 / L[c2][c2], ...)
        const int c2 = li - dl0;
        VT s0 = v0, s1 = v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (c2 == e) s0[e] = pis[e];
          if (c2 == 4 + e) s1[e] = pis[4 + e];
        }
        *reinterpret_cast<VT*>(&L8s[KB][c2][0]) = s0;
        *reinterpret_cast<VT*>(&L8s[KB][c2][4]) = s1;
      }
    }
  } else {
    // FOLLOWER: substitution with the published 8 x 8 sub-block, same operation order as the pivot wave's.  Row c2 of
    // the compact block is read with a load of exactly c2 + 1 elements (uniform addresses): every loaded register is
    // used, so the twelve reads are in flight together (partly used 16-byte reads made the register allocator overlap
    // their destinations, one LDS round trip per read).
    c64_wait(&steps_done[J], KB + 1, seen[J]);
    T l8[8][8];
    {
      const float(*B)[8] = L8s[KB];
      l8[0][0] = B[0][0];
      const V2 r1 = *reinterpret_cast<const V2*>(&B[1][0]);
      l8[1][0] = r1[0], l8[1][1] = r1[1];
      const V3 r2 = *reinterpret_cast<const V3*>(&B[2][0]);
      l8[2][0] = r2[0], l8[2][1] = r2[1], l8[2][2] = r2[2];
#pragma unroll
      for (int c2 = 3; c2 < 8; ++c2) {
        const VT a0 = *reinterpret_cast<const VT*>(&B[c2][0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) l8[c2][e] = a0[e];
      }
      l8[4][4] = B[4][4];
      const V2 r5 = *reinterpret_cast<const V2*>(&B[5][4]);
      l8[5][4] = r5[0], l8[5][5] = r5[1];
      const V3 r6 = *reinterpret_cast<const V3*>(&B[6][4]);
      l8[6][4] = r6[0], l8[6][5] = r6[1], l8[6][6] = r6[2];
      const VT r7 = *reinterpret_cast<const VT*>(&B[7][4]);
      l8[7][4] = r7[0], l8[7][5] = r7[1], l8[7][6] = r7[2], l8[7][7] = r7[3];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      x[c] *= l8[c][c];   // the diagonal of the compact block holds the reciprocal pivots
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2) x[c2] = __builtin_fmaf(-x[c], l8[c2][c], x[c2]);
    }
    if (h == 0) {
      const VT v0 = {x[0], x[1], x[2], x[3]}, v1 = {x[4], x[5], x[6], x[7]};
      *reinterpret_cast<VT*>(&Cs[w * 32 + li][c0]) = v0;
      *reinterpret_cast<VT*>(&Cs[w * 32 + li][c0 + 4]) = v1;
    }
  }
  HB_WSTAMP(w, lane, 2 * KB + 1);
  // A diagonal wave publishes the step (its rows are the A operand of everybody's update of its column block) AFTER
  // issuing the first MFMA of its own update: the LDS writes above complete under that MFMA instead of stalling the
  // wave in front of it.
  const bool publisher = w < 2;
  bool published = false;
  if (KB < 7) {
    // rank-8 update of the columns to the right, transposed: T^T[c][i] -= sum_j D[c][c0 + j] X[i][c0 + j] with D the
    // solved rows of the diagonal block (A operand) and X this wave's solved rows (B operand, its own registers)
    const T xb[4] = {h ? x[1] : x[0], h ? x[3] : x[2], h ? x[5] : x[4], h ? x[7] : x[6]};
#pragma unroll
    for (int Jp = J; Jp < 2; ++Jp) {
      if (w < 2 && Jp > w) continue;       // strict upper part of the diagonal block
      const bool right = 32 * Jp + li >= 8 * (KB + 1);   // column 32 Jp + li is still to be factored
      T a[4];
      if (w < 2 && Jp == w) {
        // the diagonal tile of a diagonal wave: both operands are its own solved entries
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = right ? xb[t] : T(0);
      } else {
        if (Jp != J) c64_wait(&steps_done[Jp], KB + 1, seen[Jp]);   // rows 32 Jp .. of the diagonal block, solved by diagonal wave Jp
        const VT d0 = *reinterpret_cast<const VT*>(&Cs[32 * Jp + li][c0]);
        const VT d1 = *reinterpret_cast<const VT*>(&Cs[32 * Jp + li][c0 + 4]);
        a[0] = h ? d0[1] : d0[0];
        a[1] = h ? d0[3] : d0[2];
        a[2] = h ? d1[1] : d1[0];
        a[3] = h ? d1[3] : d1[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = right ? a[t] : T(0);
      }
      acc[Jp] = MM::mma(-a[0], xb[0], acc[Jp]);
      if (publisher && !published) {
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0) c64_publish(&steps_done[w], KB + 1);
        __builtin_amdgcn_sched_barrier(0);
        published = true;
      }
#pragma unroll
      for (int t = 1; t < 4; ++t) acc[Jp] = MM::mma(-a[t], xb[t], acc[Jp]);
    }
  }
  if (publisher && !published && lane == 0) c64_publish(&steps_done[w], KB + 1);
  HB_WSTAMP(w, lane, 2 * KB + 2);
}

// First failed pivot of a diagonal wave whose last pivot step produced a NaN: the wave's own rows of the LDS panel
// hold its diagonal (columns 32 w .. 32 w + 31 of rows 32 w ..); the first NaN on it, 1-based within the 64-column block.
__device__ __forceinline__ int c64_locate_failure(float (*Cs)[C64_LD], int w, int lane) {
  const int i = lane & 31;
  const float dgl = Cs[32 * w + i][32 * w + i];
  const unsigned long long nanmask = __ballot(!(dgl == dgl)) & 0xffffffffull;
  return nanmask ? 32 * w + __builtin_ctzll(nanmask) + 1 : 32 * w + 32;
}

__global__ void __launch_bounds__(256) chol_rl64_kernel(const float* __restrict__ Ain, float* __restrict__ L, float* __restrict__ Y,
                                                        float* __restrict__ W, int M, int k, int* __restrict__ info, int nown,
                                                        HbSideJobs side) {
  if ((int)blockIdx.x >= nown) {
    // launch 0 of the chain keeps 8 workgroups busy: small independent launches of the step (minibatch draw +
    // gather, the sample of q(u)) ride here as extra workgroups (side_jobs.cuh)
    if (k == 0 && blockIdx.y == 0) hb_side_run(side, (int)blockIdx.x - nown);
    return;
  }
  typedef float T;
  typedef Mma<float> MM;
  typedef float VT __attribute__((ext_vector_type(4)));
  constexpr int CK = 32;  // contraction entries per lane in the rank-64 update (128 bytes)
  __shared__ __attribute__((aligned(16))) T Cs[C64_ROWS][C64_LD];
  __shared__ __attribute__((aligned(16))) T Bs[C64_NB][C64_LD];  // panel rows of the column block (rank-64 update)
  __shared__ __attribute__((aligned(16))) T L8s[8][8][8];        // in-panel step kb: its 8 x 8 sub-block, compact (c64_step)
  __shared__ int steps_done[2];                                   // in-panel steps published by diagonal wave 0 / 1
  __shared__ int wave_fail[2];

  const long boff = (long)blockIdx.y * M * M;
  Ain += boff;
  L += boff;
  const bool inv = Y != nullptr;
  if (inv) {
    Y += boff;
    W += boff;
  }
  info += blockIdx.y;
  const int nrt = M / 32;
  // workgroup -> (column block j, strip s)
  int j = k, s = blockIdx.x;
  {
    const int nf = (nrt - 2 * (k + 1)) + (inv ? 2 * (k + 1) : 0);
    int cnt = nf > 0 ? (nf + 1) / 2 : 1;
    while (s >= cnt) {
      s -= cnt;
      ++j;
      cnt = (nrt - 2 * j + (inv ? 2 * k : 0) + 3) / 4;
    }
  }
  const bool factor = (j == k);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  // entry g of this workgroup's tile list -> (live, is a Y tile, row tile)
  auto describe = [&](int g, bool& live_, bool& yt_, int& rt_) {
    const int nA = factor ? nrt - 2 * (k + 1) : nrt - 2 * j;
    const int nY = inv ? (factor ? 2 * (k + 1) : 2 * k) : 0;
    const int e = factor ? 2 * s + g - 2 : 4 * s + g;
    if (factor && g < 2) {
      live_ = true, yt_ = false, rt_ = 2 * k + g;  // the diagonal block
    } else if (e < nA) {
      live_ = true, yt_ = false, rt_ = (factor ? 2 * (k + 1) : 2 * j) + e;
    } else if (e < nA + nY) {
      live_ = true, yt_ = true, rt_ = e - nA;
    } else {
      live_ = false, yt_ = false, rt_ = nrt - 1;  // idle wave: recompute a valid tile, store nothing
    }
  };
  bool live, yt;
  int rt;
  describe(w, live, yt, rt);
  const int row0 = rt * 32, col0 = j * C64_NB;
  const bool adiag = !yt && (rt >> 1) == j;  // a row tile of column j's diagonal block
  const bool ydiag = yt && (rt >> 1) == j;   // rows of the identity that start in this column: no update yet
  if (factor && tid < 2) steps_done[tid] = 0, wave_fail[tid] = 0;

  HB_STAMP(0);
  // acc[sj]: update workgroups hold tile (rows, columns 32 sj ..) in the natural layout (column on the lane);
  // factor workgroups hold its TRANSPOSE (row on the lane: lane (li, h) register r = column 32 sj + acc_row(lane, r)).
  // Every global load of the launch -- the tile itself and the panel rows of the rank-64 update -- is ISSUED before
  // any is consumed: tile, B rows and A rows used to be three dependent round trips (load -> select / LDS store ->
  // next loads), ~2000 cycles each on data the previous launch has just written.
  typename MM::Acc acc[2];
  HB_PSTAMP(4);
  const int pc = (k - 1) * C64_NB;
  VT tileF[2][4];   // factor form: 16 bytes per lane and register group
  T tileU[2][16];   // update form: one element per register
  VT bstage[4], astage[8];
  {
    const T* src = yt ? Y : ((k <= 1) ? Ain : L);
    const int hrow0 = (adiag && k >= 2) ? row0 - C64_NB : row0;  // home of a diagonal-block tile: one block up
    if (factor) {
      // lane (li, h) reads columns 32 sj + 8 g + 4 h .. + 3 of row li
#pragma unroll
      for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          tileF[sj][g] = *reinterpret_cast<const VT*>(src + (hrow0 + li) * M + col0 + sj * 32 + 8 * g + 4 * h);
    } else {
      // unconditional loads (a fresh Y tile reads whatever its workspace holds and discards it): a conditional load
      // compiles to a branch and an s_waitcnt vmcnt(0) PER ELEMENT, i.e. 32 serialised round trips
#pragma unroll
      for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int r = 0; r < 16; ++r) tileU[sj][r] = src[(hrow0 + MM::acc_row(lane, r)) * M + col0 + sj * 32 + li];
    }
    HB_PSTAMP(5);
    if (k > 0) {
      // the panel rows come in with coalesced 16-byte loads (16 lanes per 256-byte row) and are re-read from LDS as MFMA
      // fragments: a lane loading its own fragment row directly makes every load instruction touch 64 different cache
      // lines, and the address unit -- not the memory -- becomes the bound (tools/chol_stamps.hip: ~190 cycles each).
      // B rows: the 64 rows of column block j, shared by the four waves; A rows: this wave's tile (ydiag tiles take no
      // update: their loads are harmless reads of valid rows)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = p * 16 + (tid >> 4), c4 = (tid & 15) * 4;
        bstage[p] = *reinterpret_cast<const VT*>(L + (col0 + r) * M + pc + c4);
      }
      HB_PSTAMP(6);
      const T* ap = ((yt && !ydiag) ? Y : L) + (ydiag ? col0 : row0) * M + pc;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int r = p * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        astage[p] = *reinterpret_cast<const VT*>(ap + r * M + c4);
      }
    }
  }
  HB_PSTAMP(0);
  {
    const bool fresh = yt && (ydiag || k == (rt >> 1) + 1);
    const int doff = 32 * (rt - 2 * j);  // ydiag: the 1s sit at column (row + doff)
    if (factor) {
#pragma unroll
      for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int tc = sj * 32 + 8 * g + 4 * h + e;
            const T init = (ydiag && tc == li + doff) ? T(1) : T(0);
            acc[sj][4 * g + e] = fresh ? init : tileF[sj][g][e];
          }
    } else {
#pragma unroll
      for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tr = MM::acc_row(lane, r), tc = sj * 32 + li;
          const T init = (ydiag && tc == tr + doff) ? T(1) : T(0);
          acc[sj][r] = fresh ? init : tileU[sj][r];
        }
    }
  }

  HB_PSTAMP(1);
  if (k > 0) {
    // rank-64 update by panel k-1
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = p * 16 + (tid >> 4), c4 = (tid & 15) * 4;
      *reinterpret_cast<VT*>(&Bs[r][c4]) = bstage[p];
    }
    if (!ydiag) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int r = p * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        *reinterpret_cast<VT*>(&Cs[w * 32 + r][c4]) = astage[p];
      }
    }
    HB_PSTAMP(2);
    __syncthreads();
    HB_PSTAMP(3);
    if (!ydiag) {
      // lane (li, h) contracts over entries [32h, 32h+32) of the panel rows.  All 24 fragment reads of the wave are
      // issued before the first MFMA (a read -> wait -> 8 MFMAs loop left the matrix pipe idle for an LDS round trip
      // per iteration: 134 cycles per MFMA instead of 64), and the three forms of the loop are separate straight-line
      // code (branches inside it made the compiler shuffle accumulators between register sets).
      VT va[CK / 4], vb0[CK / 4], vb1[CK / 4];
#pragma unroll
      for (int q = 0; q < CK / 4; ++q) {
        va[q] = *reinterpret_cast<const VT*>(&Cs[w * 32 + li][32 * h + 4 * q]);
        vb0[q] = *reinterpret_cast<const VT*>(&Bs[li][32 * h + 4 * q]);
        vb1[q] = *reinterpret_cast<const VT*>(&Bs[32 + li][32 * h + 4 * q]);
      }
      if (!factor) {
#pragma unroll
        for (int q = 0; q < CK / 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0] = MM::mma(-va[q][e], vb0[q][e], acc[0]);
            acc[1] = MM::mma(-va[q][e], vb1[q][e], acc[1]);
          }
      } else if (w == 0) {
        // wave 0 of a factor workgroup owns rows 0..31 of the diagonal block: its second accumulator is the block's
        // strict upper part, which nobody reads -- it is left alone (32 MFMAs fewer in front of the pivot wave's first step)
#pragma unroll
        for (int q = 0; q < CK / 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[0] = MM::mma(-vb0[q][e], va[q][e], acc[0]);
      } else {
        // transposed tiles: the column block's rows are the A operand.  Columns 0..31 first: a follower's first four
        // in-panel steps read (and update) only that accumulator
#pragma unroll
        for (int q = 0; q < CK / 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[0] = MM::mma(-vb0[q][e], va[q][e], acc[0]);
#pragma unroll
        for (int q = 0; q < CK / 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[1] = MM::mma(-vb1[q][e], va[q][e], acc[1]);
      }
    }
  } else if (factor) {
    __syncthreads();  // steps_done / wave_fail are initialised before anybody polls them
  }

  HB_STAMP(1);
  if (!factor) {
    if (live) {
      T* dst = yt ? Y : L;
      const int wrow0 = adiag ? row0 - C64_NB : row0;
#pragma unroll
      for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(wrow0 + MM::acc_row(lane, r)) * M + col0 + sj * 32 + li] = acc[sj][r];
    }
    return;
  }

  // ---- factor column block k.  Stacked panel rows [32w, 32w+32) belong to wave w (rows 0..63: the diagonal block);
  // lane (li, h) of wave w holds row 32w + li, both halves of a wave work on the same 32 rows.
  if (live) {
    int fail = 0;
    int seen[2] = {0, 0};
    HB_WSTAMP(w, lane, 0);
    c64_step<0>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
    c64_step<1>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
    c64_step<2>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
    c64_step<3>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
    if (w != 0) {   // diagonal wave 0 has finished its 32 rows
      c64_step<4>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
      c64_step<5>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
      c64_step<6>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
      c64_step<7>(acc, Cs, L8s, steps_done, w, lane, fail, seen, k);
    }
    if (w < 2) {
      // (a diagonal wave reads its own, completed, LDS rows: its LDS writes are ordered before its reads)
      if (fail != 0) fail = c64_locate_failure(Cs, w, lane);
      if (lane == 0) wave_fail[w] = fail;
    }
  }
  __syncthreads();
  HB_STAMP(2);
  if (s == 0 && tid == 0) {
    const int fail = wave_fail[0] != 0 ? wave_fail[0] : wave_fail[1];
    const int bad = fail != 0 ? k * C64_NB + fail : 0;
    if (k == 0)
      *info = bad;
    else if (bad != 0 && *info == 0)
      *info = bad;
  }
  // store the stacked panel, 32 rows (one wave's tile) at a time
  constexpr int VPR = C64_NB / 4;  // 16-byte groups per row
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {
    bool glive, gy;
    int grt;
    describe(g, glive, gy, grt);
    if (!glive || (g < 2 && s != 0)) continue;  // the diagonal block is written by strip 0 only
    for (int idx = tid; idx < 32 * VPR; idx += 256) {
      const int pr = idx / VPR, c = (idx % VPR) * 4;
      VT v = *reinterpret_cast<const VT*>(&Cs[g * 32 + pr][c]);
      if (g < 2) {
        const int dpr = g * 32 + pr;  // row inside the diagonal block
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e > dpr) v[e] = T(0);  // strict upper part of the diagonal block
      }
      *reinterpret_cast<VT*>((gy ? Y : L) + (grt * 32 + pr) * M + k * C64_NB + c) = v;
    }
    if (gy) {
      // the finished panel of Y = L^-T, transposed into W = L^-1
      for (int idx = tid; idx < 32 * C64_NB; idx += 256) {
        const int r = idx % 32, c = idx / 32;
        W[(k * C64_NB + c) * M + grt * 32 + r] = Cs[g * 32 + r][c];
      }
    }
  }
  HB_STAMP(3);
}

#include "chol_persist.cuh"
#ifdef HB_CP_STAMPS
extern unsigned long long* hb_cp_stamps_buffer;   // diagnostic build (tools/chol_persist_stamps.hip)
#endif

// One persistent launch (chol_persist.cuh) for fp32 factor + inverse at M % 64 == 0; `ws` then carries the exchange area
// and, behind it, the launch's sync words (zero at entry, left zero at exit: hb_cholesky_inverse_ws_elems).
struct CpGram {   // K(X, X) + diag I to be synthesised by the launch (A == nullptr); see hb_gram_cholesky_inverse_f32
  const float* X = nullptr;
  const float* ell = nullptr;
  long sX = 0, sEll = 0, dl = 0, d = 0;
  int kind = 0;
  float diag = 0.f;
};
__global__ void __launch_bounds__(512, 2) chol_persist_kernel(CpArgs a, HbSideJobs side) {
  if ((int)blockIdx.x >= a.total) {
    // small independent launches of the step ride here as extra workgroups (side_jobs.cuh); their bodies are written
    // for 256-thread blocks: the upper half of this block leaves
    if (threadIdx.x < 256) hb_side_run(side, (int)blockIdx.x - a.total);
    return;
  }
  __shared__ __attribute__((aligned(16))) char lds[sizeof(CpLds)];
  __shared__ unsigned s_ticket;
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&a.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  chol_persist_body(a, *reinterpret_cast<CpLds*>(lds), s_ticket);
}

// csrc/sgp.hip: a recorded forward contraction (hb_sgp_rider_begin) that this factorisation feeds starts inside the same
// launch.  1: launched (factorisation included); 0: nothing recorded or not compatible; < 0: error.
int hb_sgp_rider_launch_with(CpArgs& a, const HbSideJobs& sj, hipStream_t stream);
int hb_sgp_rider_launch_alone(hipStream_t stream);   // a recorded forward that could not ride: launched on its own now

static int chol_persist_launch(const float* A, const CpGram& g, float* L, float* W, float* ws, float* Wf, int bf16x3, long B, long M,
                               int* info, hipStream_t stream) {
  CpArgs a;
  a.A = A, a.L = L, a.W = W, a.X = ws;
  a.gX = g.X, a.gell = g.ell, a.gsX = g.sX, a.gsEll = g.sEll, a.gdl = g.dl, a.gd = g.d, a.gkind = g.kind, a.gdiag = g.diag;
  a.Wf = Wf, a.bf16x3 = bf16x3;
  a.sync = reinterpret_cast<unsigned*>(ws + B * M * M);
  a.info = info;
  a.M = (int)M, a.B = (int)B, a.nb = (int)(M / CP_NB);
  a.total = (int)cp_total(B, a.nb, 1);
  a.early = (hb_debug_get("chol_chunk_stores_late", 0) ? 16 : 0) | (hb_debug_get("chol_pair_prev", 0) ? 32 : 0), a.nside = 0, a.arrive = a.total, a.poll_naps = 1;
#ifdef HB_CP_STAMPS
  a.stamps = hb_cp_stamps_buffer;   // diagnostic build (tools/chol_persist_stamps.hip)
#else
  a.stamps = nullptr;
#endif
  const HbSideJobs sj = hb_side_take();   // pending side jobs of this thread ride on this launch
  const int rr = hb_sgp_rider_launch_with(a, sj, stream);
  if (rr != 0) return rr < 0 ? rr : 0;
  hipLaunchKernelGGL(chol_persist_kernel, dim3((unsigned)(a.total + sj.total)), dim3(512), 0, stream, a, sj);
  HB_LAUNCH_CHECK();
  return hb_sgp_rider_launch_alone(stream);
}
static inline bool chol_persist_shape(long B, long M, size_t elem) {
  return elem == 4 && M >= CP_NB && M % CP_NB == 0 && M <= 8192 && B >= 1 && B * (M / CP_NB) * (M / CP_NB) <= 65536;
}
extern "C" int hb_cholesky_persistent_shape(long B, long M, int elem_bytes) {
  return chol_persist_shape(B, M, (size_t)elem_bytes) && hb_debug_get("chol_persist", 1) != 0 ? 1 : 0;
}
// L = chol(K(X, X) + diag_add I) and W = L^-1 without K ever being written: the persistent launch synthesises its tiles from
// the points (same function and bits as hb_gram_fwd).  Shapes for which hb_cholesky_persistent_shape() is 0 are refused.
extern "C" int hb_gram_cholesky_inverse_f32(int kind, const float* X, long sX, const float* ell, long sEll, long dl, long d,
                                            double diag_add, float* L, float* W, long B, long M, int* info, float* ws,
                                            float* Wfrag, int frag_bf16x3, void* stream) {
  HB_REQUIRE(kind >= HB_KERN_RBF && kind < HB_KERN_SQDIST, "hb_gram_cholesky_inverse: unknown kernel kind %d", kind);
  HB_REQUIRE(X && ell && L && W && info && ws, "hb_gram_cholesky_inverse: NULL pointer");
  HB_REQUIRE(B >= 1 && d >= 1 && (dl == 1 || dl == d), "hb_gram_cholesky_inverse: bad extents");
  HB_REQUIRE(sEll == 0 || sEll == dl, "hb_gram_cholesky_inverse: lengthscale batch stride must be 0 or dl");
  HB_REQUIRE(hb_cholesky_persistent_shape(B, M, 4), "hb_gram_cholesky_inverse: fp32 with M %% 64 == 0 only (M = %ld, B = %ld)", M, B);
  HB_REQUIRE(((uintptr_t)L % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)ws % 16 == 0) && (!Wfrag || (uintptr_t)Wfrag % 16 == 0),
             "hb_gram_cholesky_inverse: outputs and workspace must be 16-byte aligned");
  HB_REQUIRE(!frag_bf16x3 || Wfrag, "hb_gram_cholesky_inverse: bf16x3 images need Wfrag");
  CpGram g;
  g.X = X, g.ell = ell, g.sX = sX, g.sEll = sEll, g.dl = dl, g.d = d, g.kind = kind, g.diag = (float)diag_add;
  return chol_persist_launch(nullptr, g, L, W, ws, Wfrag, frag_bf16x3, B, M, info, (hipStream_t)stream);
}
extern "C" long hb_cholesky_inverse_ws_elems(long B, long M, int elem_bytes) {
  if (B <= 0 || M <= 0) return 1;
  return B * M * M + (chol_persist_shape(B, M, (size_t)elem_bytes) ? cp_sync_words(B, M) : 0);
}

template <typename T>
static int cholesky_launch(const T* A, T* L, T* W, T* ws, T* Wf, int bf16x3, long B, long M, int* info, hipStream_t stream) {
  HB_REQUIRE(!Wf || (W && M % 32 == 0), "hb_cholesky_inverse: the fragment-major copies need W and M %% 32 == 0");
  HB_REQUIRE(!bf16x3 || (Wf && sizeof(T) == 4), "hb_cholesky_inverse: bf16x3 images need Wfrag and fp32");
  HB_REQUIRE(B >= 0 && M >= 0, "hb_cholesky: negative extent");
  HB_REQUIRE(A && L && info, "hb_cholesky: NULL pointer");
  HB_REQUIRE(!W || ws, "hb_cholesky_inverse: workspace of B*M*M elements required");
  HB_REQUIRE(B <= 65535, "hb_cholesky: batch too large");
  HB_REQUIRE(M * M < 2147483647L, "hb_cholesky: matrix too large for 32-bit indexing");
  if (B == 0) return 0;
  if (M == 0) {
    HB_HIP(hb_zero_async(info, sizeof(int) * B, stream));
    return 0;
  }
  HB_REQUIRE((const void*)A != (const void*)L && (const void*)A != (const void*)W && (const void*)L != (const void*)W,
             "hb_cholesky: A, L (and W) must not alias");
  const int inv = W != nullptr;
  const int nblk = hb_cdiv(M, CR_B);
  const bool fast = ((uintptr_t)L % 16 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)ws % 16 == 0) && M % CR_B == 0;
  const bool no64 = hb_debug_get("chol_no64", 0) != 0;  // diagnostic A/B switches (hb_debug_set)
  if (sizeof(T) == 4 && fast && inv && chol_persist_shape(B, M, sizeof(T)) && hb_debug_get("chol_persist", 1) != 0) {
    // (every block of L, W and of the fragment-major images is written by the launch itself: no finishing pass)
    return chol_persist_launch((const float*)A, CpGram(), (float*)L, (float*)W, (float*)ws, (float*)Wf, bf16x3, B, M, info, stream);
  } else if (sizeof(T) == 4 && fast && M % C64_NB == 0 && !no64) {
    const int nrt = (int)(M / 32);
    HbSideJobs noside = {};
    for (int k = 0; k < nrt / 2; ++k) {
      const int nown = chol64_grid(nrt, k, inv);
      const HbSideJobs sj = k == 0 ? hb_side_take() : noside;   // pending side jobs of this thread ride on launch 0
      dim3 grid((unsigned)(nown + (k == 0 ? sj.total : 0)), (unsigned)B);
      hipLaunchKernelGGL(chol_rl64_kernel, grid, dim3(256), 0, stream, (const float*)A, (float*)L,
                         inv ? (float*)ws : (float*)nullptr, (float*)W, (int)M, k, info, nown, sj);
      HB_LAUNCH_CHECK();
    }
  } else {
    for (int k = 0; k < nblk; ++k) {
      dim3 grid((unsigned)chol_rl_grid(nblk, k, inv), (unsigned)B);
      if (fast)
        hipLaunchKernelGGL((chol_rl_kernel<T, true>), grid, dim3(256), 0, stream, A, L, inv ? ws : (T*)nullptr, W, (int)M, k,
                           info);
      else
        hipLaunchKernelGGL((chol_rl_kernel<T, false>), grid, dim3(256), 0, stream, A, L, inv ? ws : (T*)nullptr, W, (int)M,
                           k, info);
      HB_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL(tril_inplace_kernel<T>, dim3(hb_stream_grid(B * M * M, 256)), dim3(256), 0, stream, L, W, Wf, bf16x3, B, M);
  HB_LAUNCH_CHECK();
  return 0;
}

extern "C" int hb_cholesky_f32(const float* A, float* L, long B, long M, int* info, void* stream) {
  return cholesky_launch<float>(A, L, nullptr, nullptr, nullptr, 0, B, M, info, (hipStream_t)stream);
}
extern "C" int hb_cholesky_f64(const double* A, double* L, long B, long M, int* info, void* stream) {
  return cholesky_launch<double>(A, L, nullptr, nullptr, nullptr, 0, B, M, info, (hipStream_t)stream);
}
extern "C" int hb_cholesky_inverse_f32(const float* A, float* L, float* W, long B, long M, int* info, float* ws,
                                       float* Wfrag, int frag_bf16x3, void* stream) {
  HB_REQUIRE(W, "hb_cholesky_inverse: NULL pointer");
  return cholesky_launch<float>(A, L, W, ws, Wfrag, frag_bf16x3, B, M, info, (hipStream_t)stream);
}
extern "C" int hb_cholesky_inverse_f64(const double* A, double* L, double* W, long B, long M, int* info, double* ws,
                                       double* Wfrag, int frag_bf16x3, void* stream) {
  HB_REQUIRE(W, "hb_cholesky_inverse: NULL pointer");
  return cholesky_launch<double>(A, L, W, ws, Wfrag, frag_bf16x3, B, M, info, (hipStream_t)stream);
}

// ===========================================================================
// W = L^{-1} by recursive doubling over diagonal blocks:
//   inv([[L11,0],[L21,L22]]) = [[W11,0],[-W22 L21 W11, W22]]
// 32x32 diagonal inverses by forward substitution (one wave each), then
// log2(M/32) levels of two batched GEMMs, both triangular-aware.
// ===========================================================================
template <typename T>
__global__ void __launch_bounds__(64) trinv_diag_kernel(const T* __restrict__ L, T* __restrict__ W, long M) {
  // One wave per 32x32 diagonal block: lane c solves  x L_ii^T = e_c^T, i.e. x = column c of L_ii^{-1},
  // with the same right-looking row solve the Cholesky panel uses (L_ii^T staged in LDS).
  __shared__ __attribute__((aligned(16))) T LsT[CH_NB][CH_LD];
  __shared__ __attribute__((aligned(16))) T invd[CH_NB];
  const long b = blockIdx.y;
  L += b * M * M;
  W += b * M * M;
  const long i0 = (long)blockIdx.x * CH_NB;
  const int nb = (int)((M - i0) < CH_NB ? (M - i0) : CH_NB);
  const int lane = threadIdx.x;
  for (int idx = lane; idx < CH_NB * CH_NB; idx += 64) {
    const int i = idx / CH_NB, j = idx % CH_NB;  // element L[i][j], stored transposed
    const bool ok = (i < nb) & (j < nb) & (j <= i);
    const T v = L[(i0 + (i < nb ? i : 0)) * M + i0 + (j < nb ? j : 0)];
    const T val = ok ? v : (i == j ? T(1) : T(0));  // identity padding of a ragged last block
    LsT[j][i] = val;
    if (i == j) invd[i] = T(1) / val;
  }
  __syncthreads();
  const int c = lane & 31;
  T t[CH_NB];
#pragma unroll
  for (int i = 0; i < CH_NB; ++i) t[i] = (i == c) ? T(1) : T(0);
  trsolve_row32<T>(t, LsT, invd);
  if (lane < nb) {
#pragma unroll
    for (int i = 0; i < CH_NB; ++i)
      if (i < nb) W[(i0 + i) * M + i0 + lane] = t[i];
  }
}

// phase 0: Tm[r0+i, c0+j] = sum_k L[r0+i, c0+k] W[c0+k, c0+j]      (k >= j: W11 lower)
// phase 1: W [r0+i, c0+j] = -sum_k W[r0+i, r0+k] Tm[r0+k, c0+j]    (k <= i: W22 lower)
template <typename T, int PHASE, bool FAST>
__global__ void __launch_bounds__(256) trinv_level_kernel(const T* __restrict__ L, T* __restrict__ W,
                                                          T* __restrict__ Tm, long Ml, long sl) {
  typedef TileGemm<T, 64, 64, 16, 2, 2> G;
  __shared__ __attribute__((aligned(16))) T lds[G::LDS_ELEMS];
  const long b = blockIdx.z;
  L += b * Ml * Ml;
  W += b * Ml * Ml;
  Tm += b * Ml * Ml;
  const int M = (int)Ml, s = (int)sl;
  const int pair = blockIdx.y;
  const int c0 = 2 * pair * s, r0 = c0 + s;
  const int tps = (s + 63) / 64;  // tiles per side
  const int ti = (blockIdx.x / tps) * 64, tj = (blockIdx.x % tps) * 64;
  if (r0 + ti >= M) return;  // tile entirely below the matrix
  G g;
  g.zero();
  if (PHASE == 0) {
    auto la = [&](int m, int k) -> T {
      const int r = r0 + ti + m, c = c0 + k;
      return L[(r < M ? r : M - 1) * M + c];
    };
    auto fa = [&](T raw, int m, int k) -> T { return ((ti + m < s) & (r0 + ti + m < M)) ? raw : T(0); };
    auto lb = [&](int k, int n) -> T {
      const int cj = c0 + tj + n;
      return W[(c0 + k) * M + (cj < M ? cj : M - 1)];
    };
    auto fb = [&](T raw, int k, int n) -> T {
      const int j = tj + n;
      return ((j < s) & (k >= j)) ? raw : T(0);
    };
    if constexpr (FAST) {
      typedef typename G::VT VT;
      constexpr int VEC = G::VEC;
      auto la4 = [&](int m, int k) -> VT {
        const int r = r0 + ti + m;
        return *reinterpret_cast<const VT*>(&L[(r < M ? r : M - 1) * M + c0 + k]);
      };
      auto fa4 = [&](VT raw, int m, int k) -> VT {
        const VT zero = {};
        return ((ti + m < s) & (r0 + ti + m < M)) ? raw : zero;
      };
      auto lb4 = [&](int k, int n) -> VT {
        const int cj = c0 + tj + n;
        return *reinterpret_cast<const VT*>(&W[(c0 + k) * M + (cj < M ? cj : M - VEC)]);
      };
      auto fb4 = [&](VT raw, int k, int n) -> VT {
        VT v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const int j = tj + n + q;
          v[q] = ((j < s) & (k >= j)) ? raw[q] : T(0);
        }
        return v;
      };
      g.template run_vec<HB_KC, HB_MC>(tj, s, la4, fa4, lb4, fb4, lds);
    } else {
      g.template run<true, false>(tj, s, la, fa, lb, fb, lds);
    }
    g.for_each([&](int row, int col, T v) {
      const int i = ti + row, j = tj + col;
      if (i < s && j < s && r0 + i < M) Tm[(r0 + i) * M + c0 + j] = v;
    });
  } else {
    int kend = ti + 64;
    if (kend > s) kend = s;
    auto la = [&](int m, int k) -> T {
      const int ri = r0 + ti + m, rk = r0 + k;
      return W[(ri < M ? ri : M - 1) * M + (rk < M ? rk : M - 1)];
    };
    auto fa = [&](T raw, int m, int k) -> T {
      const int i = ti + m;
      return ((i < s) & (k <= i) & (r0 + i < M)) ? raw : T(0);
    };
    auto lb = [&](int k, int n) -> T {
      const int rk = r0 + k, cj = c0 + tj + n;
      return Tm[(rk < M ? rk : M - 1) * M + (cj < M ? cj : M - 1)];
    };
    auto fb = [&](T raw, int k, int n) -> T { return ((tj + n < s) & (r0 + k < M)) ? raw : T(0); };
    if constexpr (FAST) {
      typedef typename G::VT VT;
      constexpr int VEC = G::VEC;
      auto la4 = [&](int m, int k) -> VT {
        const int ri = r0 + ti + m, rk = r0 + k;
        return *reinterpret_cast<const VT*>(&W[(ri < M ? ri : M - 1) * M + (rk < M ? rk : M - VEC)]);
      };
      auto fa4 = [&](VT raw, int m, int k) -> VT {
        const int i = ti + m;
        VT v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = ((i < s) & (k + q <= i) & (r0 + i < M)) ? raw[q] : T(0);
        return v;
      };
      auto lb4 = [&](int k, int n) -> VT {
        const int rk = r0 + k, cj = c0 + tj + n;
        return *reinterpret_cast<const VT*>(&Tm[(rk < M ? rk : M - 1) * M + (cj < M ? cj : M - VEC)]);
      };
      auto fb4 = [&](VT raw, int k, int n) -> VT {
        VT v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = ((tj + n + q < s) & (r0 + k < M)) ? raw[q] : T(0);
        return v;
      };
      g.template run_vec<HB_KC, HB_MC>(0, kend, la4, fa4, lb4, fb4, lds);
    } else {
      g.template run<true, false>(0, kend, la, fa, lb, fb, lds);
    }
    g.for_each([&](int row, int col, T v) {
      const int i = ti + row, j = tj + col;
      if (i < s && j < s && r0 + i < M) W[(r0 + i) * M + c0 + j] = -v;
    });
  }
}

template <typename T>
static int trinv_launch(const T* L, T* W, long B, long M, T* ws, hipStream_t stream) {
  HB_REQUIRE(B >= 0 && M >= 0, "hb_trinv: negative extent");
  HB_REQUIRE(L && W, "hb_trinv: NULL pointer");
  HB_REQUIRE(L != W, "hb_trinv: in-place not supported");
  HB_REQUIRE(B <= 65535, "hb_trinv: batch too large");
  HB_REQUIRE(M * M < 2147483647L, "hb_trinv: matrix too large for 32-bit indexing");
  if (B * M == 0) return 0;
  HB_REQUIRE(M <= CH_NB || ws, "hb_trinv: workspace of B*M*M elements required");
  HB_HIP(hb_zero_async(W, sizeof(T) * B * M * M, stream));
  // vector path: every k-range of the level kernels is a multiple of 16 once M is
  const bool fast = ((uintptr_t)L % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)ws % 16 == 0) && M % 16 == 0;
  const int nblk = hb_cdiv(M, CH_NB);
  hipLaunchKernelGGL(trinv_diag_kernel<T>, dim3(nblk, (unsigned)B), dim3(64), 0, stream, L, W, M);
  HB_LAUNCH_CHECK();
  for (long s = CH_NB; s < M; s *= 2) {
    const int pairs = hb_cdiv(M, 2 * s);
    const int tps = hb_cdiv(s, 64);
    dim3 grid(tps * tps, pairs, (unsigned)B);
    if (fast) {
      hipLaunchKernelGGL((trinv_level_kernel<T, 0, true>), grid, dim3(256), 0, stream, L, W, ws, M, s);
      HB_LAUNCH_CHECK();
      hipLaunchKernelGGL((trinv_level_kernel<T, 1, true>), grid, dim3(256), 0, stream, L, W, ws, M, s);
    } else {
      hipLaunchKernelGGL((trinv_level_kernel<T, 0, false>), grid, dim3(256), 0, stream, L, W, ws, M, s);
      HB_LAUNCH_CHECK();
      hipLaunchKernelGGL((trinv_level_kernel<T, 1, false>), grid, dim3(256), 0, stream, L, W, ws, M, s);
    }
    HB_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int hb_trinv_f32(const float* L, float* W, long B, long M, float* ws, void* stream) {
  return trinv_launch<float>(L, W, B, M, ws, (hipStream_t)stream);
}
extern "C" int hb_trinv_f64(const double* L, double* W, long B, long M, double* ws, void* stream) {
  return trinv_launch<double>(L, W, B, M, ws, (hipStream_t)stream);
}
