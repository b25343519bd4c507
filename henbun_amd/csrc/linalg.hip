// Dense linear algebra on the MFMA tile engine: batched matmul (+bias/act
// epilogue, split-K, lower-only output), blocked Cholesky, triangular inverse.
//
// Reference call sites: tf.matmul (gp/gp.py:50,122,125,171; nn.py:32),
// tf.cholesky (gp/kernels.py:101; gp/gp.py:135), tf.matrix_triangular_solve
// (gp/gp.py:162,169).  TensorFlow supplied these through Eigen/cuSOLVER; here
// they are hand-written for gfx950.
#include "common.cuh"
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
  T* ws;
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

template <typename T, bool TA, bool TB, bool FAST, int BT>
__global__ void __launch_bounds__(256) matmul_kernel(MmArgs<T> a) {
  typedef TileGemm<T, BT, BT, 16, 2, 2> G;
  __shared__ T lds[G::LDS_ELEMS];
  const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
  const int lda = (int)a.lda, ldb = (int)a.ldb;
  const int tiles_n = (N + BT - 1) / BT;
  const int row0 = (blockIdx.x / tiles_n) * BT;
  const int col0 = (blockIdx.x % tiles_n) * BT;
  const long b = blockIdx.y;
  const int s = blockIdx.z;
  if ((a.flags & HB_MM_LOWER_OUT) && col0 > row0 + BT - 1) return;
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
    auto fb4 = [&](VT raw, int k, int n) -> VT { return col0 + n < N ? raw : zero; };
    g.template run_vec<(TA ? HB_MC : HB_KC), (TB ? HB_KC : HB_MC)>(kbeg, kend, la4, fa4, lb4, fb4, lds);
  } else {
    g.template run<!TA, TB>(kbeg, kend, la, fa, lb, fb, lds);
  }
  if (a.S > 1) {
    T* wsb = a.ws + ((long)s * a.batch + b) * a.M * a.N;
    g.for_each([&](int row, int col, T v) {
      const long r = row0 + row, c = col0 + col;
      if (r < a.M && c < a.N) wsb[r * a.N + c] = a.alpha * v;
    });
  } else {
    T* Cb = a.C + b * a.sC;
    const T* biasb = a.bias ? a.bias + b * a.sBias : nullptr;
    g.for_each([&](int row, int col, T v) {
      const long r = row0 + row, c = col0 + col;
      if (r < a.M && c < a.N) {
        T o = a.alpha * v;
        if (biasb) o += biasb[c];
        o = apply_act<T>(a.act, o);
        if (a.beta != T(0)) o += a.beta * Cb[r * a.ldc + c];
        Cb[r * a.ldc + c] = o;
      }
    });
  }
}

template <typename T>
__global__ void __launch_bounds__(256) matmul_splitk_finish_kernel(MmArgs<T> a) {
  const long total = a.batch * a.M * a.N;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long b = t / (a.M * a.N);
    const long rem = t - b * a.M * a.N;
    const long r = rem / a.N, c = rem - r * a.N;
    if ((a.flags & HB_MM_LOWER_OUT) && (c / a.tile) * a.tile > (r / a.tile) * a.tile + a.tile - 1) continue;
    T acc = T(0);
    for (int s = 0; s < a.S; ++s) acc += a.ws[(long)s * total + t];
    if (a.bias) acc += a.bias[b * a.sBias + c];
    acc = apply_act<T>(a.act, acc);
    T* cp = a.C + b * a.sC + r * a.ldc + c;
    if (a.beta != T(0)) acc += a.beta * cp[0];
    cp[0] = acc;
  }
}

template <typename T>
static int matmul_launch(const T* A, const T* B, T* C, long batch, long M, long N, long K, long lda, long ldb,
                         long ldc, long sA, long sB, long sC, int transA, int transB, double alpha, double beta,
                         const T* bias, long sBias, int act, int flags, T* ws, long ws_elems, hipStream_t stream) {
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
  // large outputs: 128x128 tiles (64x64 per wave: 4 MFMAs per fragment pair) cut the per-MFMA staging cost
  const int BT = (M >= 256 && N >= 256) ? 128 : 64;
  a.tile = BT;
  const long tiles = (long)hb_cdiv(M, BT) * hb_cdiv(N, BT);
  int S = 1;
  if (ws && tiles * batch < 192 && K >= 256) {
    // few output tiles: spread the contraction over the idle CUs (each slice >= 128 deep)
    long s1 = K / 128;
    long s2 = 512 / (tiles * batch);
    long s3 = ws_elems / (batch * M * N);
    S = (int)(s1 < s2 ? s1 : s2);
    if (S > s3) S = (int)s3;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
  }
  a.S = S;
  dim3 grid((unsigned)tiles, (unsigned)batch, (unsigned)S);
  constexpr long VEC = 16 / sizeof(T);
  const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && lda % VEC == 0 && ldb % VEC == 0 &&
                       sA % VEC == 0 && sB % VEC == 0;
  const bool fast = aligned && K % 16 == 0 && K > 0 && (!transA || M % VEC == 0) && (transB || N % VEC == 0);
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
  if (S > 1) {
    hipLaunchKernelGGL(matmul_splitk_finish_kernel<T>, dim3(hb_stream_grid(batch * M * N, 256)), dim3(256), 0, stream,
                       a);
    HB_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int hb_matmul_f32(const float* A, const float* B, float* C, long batch, long M, long N, long K, long lda,
                             long ldb, long ldc, long sA, long sB, long sC, int transA, int transB, double alpha,
                             double beta, const float* bias, long sBias, int act, int flags, float* ws, long ws_elems,
                             void* stream) {
  return matmul_launch<float>(A, B, C, batch, M, N, K, lda, ldb, ldc, sA, sB, sC, transA, transB, alpha, beta, bias,
                              sBias, act, flags, ws, ws_elems, (hipStream_t)stream);
}
extern "C" int hb_matmul_f64(const double* A, const double* B, double* C, long batch, long M, long N, long K,
                             long lda, long ldb, long ldc, long sA, long sB, long sC, int transA, int transB,
                             double alpha, double beta, const double* bias, long sBias, int act, int flags, double* ws,
                             long ws_elems, void* stream) {
  return matmul_launch<double>(A, B, C, batch, M, N, K, lda, ldb, ldc, sA, sB, sC, transA, transB, alpha, beta, bias,
                               sBias, act, flags, ws, ws_elems, (hipStream_t)stream);
}

// ===========================================================================
// Cholesky: left-looking, one launch per 32-column panel.
//
// Launch j (panel columns [j0, j0+32)): workgroup g owns the 32 diagonal rows
// (every workgroup recomputes the diagonal block -- cheaper than a cross-
// workgroup hand-off) plus 96 rows below; it forms
//   C = A[rows, panel] - L[rows, 0:j0] L[panel, 0:j0]^T        (MFMA)
// factors the 32x32 diagonal block with ONE wave working in LDS (no workgroup
// barriers in the 32-step loop), then solves its rows against it.
// ===========================================================================
// In-kernel phase stamps: compiled in only by tools/chol_stamps.hip (diagnostic build).
#ifndef HB_STAMP
#define HB_STAMP(i)
#endif

#define CH_NB 32
#define CH_RB 96
#define CH_LD (CH_NB + 4)  // LDS row stride of the panel tile: keeps rows 16-byte aligned for vector reads

__device__ __forceinline__ float bcast_lane(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  const long long r = ((long long)hi << 32) | (unsigned int)lo;
  return __builtin_bit_cast(double, r);
}

// l = sqrt(d), inv = 1/l.  fp32: one v_rsq_f32 (1 ulp) + one multiply instead of the IEEE sqrt and
// divide sequences (~25 instructions on the critical path of every column); fp64 keeps the exact forms.
__device__ __forceinline__ void pivot_sqrt(float d, float& l, float& inv) {
  inv = __builtin_amdgcn_rsqf(d);
  l = d * inv;
}
__device__ __forceinline__ void pivot_sqrt(double d, double& l, double& inv) {
  l = sqrt(d);
  inv = 1.0 / l;
}

// Factor the 32x32 SPD block whose lower triangle sits in Ls[0..31][0..31] (row
// stride CH_LD, 16-byte aligned rows), in place, with ONE wave; also leaves
// L^T in LsT and 1/l_kk in invd for the row solves.
//
// Left-looking, fully unrolled.  Lane r owns row r: `a[k]` is its input entry,
// `l[k]` its finished entries.  Step k forms column k,
//     c_r = a_rk - sum_{j<k} l_rj * l_kj ,
// where row k of L (the l_kj) is the same for every lane: entries 0..k-2 were
// written to LDS at least two steps earlier and are fetched -- one step AHEAD,
// so their latency is off the critical path -- with 16-byte uniform-address
// reads (LDS broadcast -> VGPRs, no SGPR pressure); the newest entry l_{k,k-1}
// comes by a single v_readlane from lane k.  The dot product runs on four
// partial sums (a lone wave retires a dependent FMA only every ~8 cycles).
// Returns k+1 of the first non-positive pivot (0 = ok).
template <typename T>
__device__ __forceinline__ int potrf32_lds(T (*Ls)[CH_LD], T (*LsT)[CH_LD], T* invd, int lane) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int NV = CH_NB / VEC;
  typedef T VT __attribute__((ext_vector_type(VEC)));
  const int r = lane & 31;
  T a[CH_NB], l[CH_NB];
#pragma unroll
  for (int j = 0; j < CH_NB; j += VEC) {
    const VT v = *reinterpret_cast<const VT*>(&Ls[r][j]);
#pragma unroll
    for (int q = 0; q < VEC; ++q) a[j + q] = v[q];
  }
  int fail = 0;
  VT cur[NV], nxt[NV];
#pragma unroll
  for (int k = 0; k < CH_NB; ++k) {
    T acc[4] = {a[k], T(0), T(0), T(0)};
    if (k > 0) {
      T lrow[CH_NB];
#pragma unroll
      for (int j = 0; j + 1 < k; j += VEC)
#pragma unroll
        for (int q = 0; q < VEC; ++q) lrow[j + q] = cur[j / VEC][q];
      lrow[k - 1] = bcast_lane(l[k - 1], k);
#pragma unroll
      for (int j = 0; j < k; ++j) acc[j & 3] -= l[j] * lrow[j];
    }
    // prefetch row k+1, entries 0..k-1 (all written by step k-1 at the latest; same-wave DS ops are ordered)
    if (k + 1 < CH_NB) {
#pragma unroll
      for (int j = 0; j < k; j += VEC) nxt[j / VEC] = *reinterpret_cast<const VT*>(&Ls[k + 1][j]);
    }
    const T c = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    const T d = bcast_lane(c, k);
    if (fail == 0 && !(d > T(0))) fail = k + 1;
    T lkk, inv;
    pivot_sqrt(d, lkk, inv);
    l[k] = (r == k) ? lkk : c * inv;
    if (lane < CH_NB) {
      Ls[r][k] = l[k];
      LsT[k][r] = l[k];
      if (r == k) invd[k] = inv;
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) cur[q] = nxt[q];
  }
  return fail;
}

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

template <typename T, bool FAST>
__global__ void __launch_bounds__(256) chol_panel_kernel(const T* __restrict__ Ain, T* __restrict__ L, long M, long j0,
                                                         int* __restrict__ info) {
  typedef TileGemm<T, 128, 32, 32, 4, 1> G;
  __shared__ T lds[G::LDS_ELEMS];
  __shared__ __attribute__((aligned(16))) T Cs[128][CH_LD];
  __shared__ __attribute__((aligned(16))) T LsT[CH_NB][CH_LD];
  __shared__ __attribute__((aligned(16))) T invd[CH_NB];
  const long b = blockIdx.y;
  Ain += b * M * M;
  L += b * M * M;
  info += b;
  const int Mi = (int)M, j0i = (int)j0;
  const int nb = (Mi - j0i) < CH_NB ? (Mi - j0i) : CH_NB;
  const int r0 = j0i + CH_NB + blockIdx.x * CH_RB;
  auto rvalid = [&](int m) -> bool { return m < CH_NB ? m < nb : (r0 + (m - CH_NB)) < Mi; };
  // global row of tile row m, clamped to a safe row when invalid
  auto grow = [&](int m) -> int { return rvalid(m) ? (m < CH_NB ? j0i + m : r0 + (m - CH_NB)) : j0i; };
  HB_STAMP(0);
  G g;
  g.zero();
  auto la = [&](int m, int k) -> T { return L[grow(m) * Mi + k]; };
  auto fa = [&](T raw, int m, int k) -> T { return rvalid(m) ? raw : T(0); };
  auto lb = [&](int k, int n) -> T { return L[(j0i + (n < nb ? n : nb - 1)) * Mi + k]; };
  auto fb = [&](T raw, int k, int n) -> T { return n < nb ? raw : T(0); };
  if constexpr (FAST) {
    typedef typename G::VT VT;
    const VT zero = {};
    auto la4 = [&](int m, int k) -> VT { return *reinterpret_cast<const VT*>(&L[grow(m) * Mi + k]); };
    auto fa4 = [&](VT raw, int m, int k) -> VT { return rvalid(m) ? raw : zero; };
    auto lb4 = [&](int k, int n) -> VT { return *reinterpret_cast<const VT*>(&L[(j0i + (n < nb ? n : nb - 1)) * Mi + k]); };
    auto fb4 = [&](VT raw, int k, int n) -> VT { return n < nb ? raw : zero; };
    g.template run_vec<HB_KC, HB_KC>(0, j0i, la4, fa4, lb4, fb4, lds);
  } else {
    g.template run<true, true>(0, j0i, la, fa, lb, fb, lds);
  }
  HB_STAMP(1);
  g.for_each([&](int row, int col, T v) {
    const bool ok = rvalid(row) && col < nb;
    const T aval = Ain[(long)grow(row) * M + j0 + (col < nb ? col : nb - 1)];
    T c = ok ? aval - v : T(0);
    if (row < CH_NB && row >= nb && col == row) c = T(1);  // identity padding of a ragged last panel
    Cs[row][col] = c;
  });
  __syncthreads();
  HB_STAMP(2);
  if (threadIdx.x < 64) {
    const int fail = potrf32_lds<T>(Cs, LsT, invd, threadIdx.x);
    // info: block 0 of every panel launch is its only writer; the first panel resets it
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const int bad = (fail != 0 && fail <= nb) ? (int)(j0 + fail) : 0;
      if (j0 == 0)
        *info = bad;
      else if (bad != 0 && *info == 0)
        *info = bad;
    }
  }
  __syncthreads();
  HB_STAMP(3);
  // panel rows: x L_jj^T = c
  if (threadIdx.x < CH_RB) {
    const int m = CH_NB + threadIdx.x;
    const long r = (long)r0 + threadIdx.x;
    if (r < M) {
      constexpr int VEC = 16 / sizeof(T);
      typedef T VT __attribute__((ext_vector_type(VEC)));
      T xr[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; c += VEC) {
        const VT v = *reinterpret_cast<const VT*>(&Cs[m][c]);
#pragma unroll
        for (int q = 0; q < VEC; ++q) xr[c + q] = v[q];
      }
      trsolve_row32<T>(xr, LsT, invd);
      T* dst = L + r * M + j0;
      if (nb == CH_NB) {
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) dst[c] = xr[c];
      } else {
#pragma unroll
        for (int c = 0; c < CH_NB; ++c)
          if (c < nb) dst[c] = xr[c];
      }
    }
  }
  HB_STAMP(4);
  if (blockIdx.x == 0) {
    // diagonal block, strict upper part zeroed (the rest of the upper triangle is cleared once, after the last panel)
    for (int idx = threadIdx.x; idx < CH_NB * CH_NB; idx += blockDim.x) {
      const int i = idx / CH_NB, j = idx % CH_NB;
      if (i < nb && j < nb) L[(j0 + i) * M + j0 + j] = j <= i ? Cs[i][j] : T(0);
    }
  }
  HB_STAMP(5);
}

template <typename T>
__global__ void __launch_bounds__(256) tril_inplace_kernel(T* __restrict__ L, long B, long M) {
  const int Mi = (int)M;
  const long total = B * M * M;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int rem = (int)(t % (M * M));
    const int i = rem / Mi, j = rem - i * Mi;
    if (j > i) L[t] = T(0);
  }
}

template <typename T>
static int cholesky_launch(const T* A, T* L, long B, long M, int* info, hipStream_t stream) {
  HB_REQUIRE(B >= 0 && M >= 0, "hb_cholesky: negative extent");
  HB_REQUIRE(A && L && info, "hb_cholesky: NULL pointer");
  HB_REQUIRE(B <= 65535, "hb_cholesky: batch too large");
  HB_REQUIRE(M * M < 2147483647L, "hb_cholesky: matrix too large for 32-bit indexing");
  if (B == 0) return 0;
  if (M == 0) {
    HB_HIP(hb_zero_async(info, sizeof(int) * B, stream));
    return 0;
  }
  const bool fast = ((uintptr_t)L % 16 == 0) && M % (16 / (long)sizeof(T)) == 0;
  for (long j0 = 0; j0 < M; j0 += CH_NB) {
    const long below = M - j0 - CH_NB;
    const int gx = below > 0 ? hb_cdiv(below, CH_RB) : 1;
    if (fast)
      hipLaunchKernelGGL((chol_panel_kernel<T, true>), dim3(gx, (unsigned)B), dim3(256), 0, stream, A, L, M, j0, info);
    else
      hipLaunchKernelGGL((chol_panel_kernel<T, false>), dim3(gx, (unsigned)B), dim3(256), 0, stream, A, L, M, j0, info);
    HB_LAUNCH_CHECK();
  }
  if (M > CH_NB) {
    hipLaunchKernelGGL(tril_inplace_kernel<T>, dim3(hb_stream_grid(B * M * M, 256)), dim3(256), 0, stream, L, B, M);
    HB_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int hb_cholesky_f32(const float* A, float* L, long B, long M, int* info, void* stream) {
  return cholesky_launch<float>(A, L, B, M, info, (hipStream_t)stream);
}
extern "C" int hb_cholesky_f64(const double* A, double* L, long B, long M, int* info, void* stream) {
  return cholesky_launch<double>(A, L, B, M, info, (hipStream_t)stream);
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
  __shared__ T lds[G::LDS_ELEMS];
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
