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
  int act, flags, S;
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

template <typename T, bool TA, bool TB>
__global__ void __launch_bounds__(256) matmul_kernel(MmArgs<T> a) {
  typedef TileGemm<T, 64, 64, 16, 2, 2> G;
  __shared__ T lds[G::LDS_ELEMS];
  const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
  const int lda = (int)a.lda, ldb = (int)a.ldb;
  const int tiles_n = (N + 63) / 64;
  const int row0 = (blockIdx.x / tiles_n) * 64;
  const int col0 = (blockIdx.x % tiles_n) * 64;
  const long b = blockIdx.y;
  const int s = blockIdx.z;
  if ((a.flags & HB_MM_LOWER_OUT) && col0 > row0 + 63) return;
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
  g.template run<!TA, TB>(kbeg, kend, la, fa, lb, fb, lds);
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
    if ((a.flags & HB_MM_LOWER_OUT) && (c / 64) * 64 > (r / 64) * 64 + 63) continue;
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
  const long tiles = (long)hb_cdiv(M, 64) * hb_cdiv(N, 64);
  int S = 1;
  if (ws && tiles * batch < 128 && K >= 1024) {
    long s1 = K / 256;
    long s2 = 512 / (tiles * batch);
    long s3 = ws_elems / (batch * M * N);
    S = (int)(s1 < s2 ? s1 : s2);
    if (S > s3) S = (int)s3;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
  }
  a.S = S;
  dim3 grid((unsigned)tiles, (unsigned)batch, (unsigned)S);
  if (!transA && !transB)
    hipLaunchKernelGGL((matmul_kernel<T, false, false>), grid, dim3(256), 0, stream, a);
  else if (!transA && transB)
    hipLaunchKernelGGL((matmul_kernel<T, false, true>), grid, dim3(256), 0, stream, a);
  else if (transA && !transB)
    hipLaunchKernelGGL((matmul_kernel<T, true, false>), grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((matmul_kernel<T, true, true>), grid, dim3(256), 0, stream, a);
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
#define CH_NB 32
#define CH_RB 96

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

// Factor a 32x32 SPD block held one ROW per lane (a[j] = row `lane&31`, column
// j; lanes 32..63 mirror 0..31).  Right-looking, fully unrolled: column k is
// scaled in place, then every later column j gets a[j] -= l_rk * l_jk with
// l_jk broadcast from lane j by v_readlane (no LDS, no barriers).  Returns
// k+1 of the first non-positive pivot (0 = ok).  Upper-triangle entries end
// up holding garbage and must be ignored by the caller.
template <typename T>
__device__ __forceinline__ int potrf32_regs(T (&a)[CH_NB], int lane) {
  const int r = lane & 31;
  int fail = 0;
#pragma unroll
  for (int k = 0; k < CH_NB; ++k) {
    const T d = bcast_lane(a[k], k);
    if (fail == 0 && !(d > T(0))) fail = k + 1;
    const T lkk = hb_sqrt(d);
    const T inv = T(1) / lkk;
    a[k] = (r == k) ? lkk : a[k] * inv;
#pragma unroll
    for (int j = k + 1; j < CH_NB; ++j) {
      const T ljk = bcast_lane(a[k], j);
      a[j] -= a[k] * ljk;
    }
  }
  return fail;
}

template <typename T>
__global__ void __launch_bounds__(256) chol_panel_kernel(const T* __restrict__ Ain, T* __restrict__ L, long M, long j0,
                                                         int* __restrict__ info) {
  typedef TileGemm<T, 128, 32, 16, 4, 1> G;
  __shared__ T lds[G::LDS_ELEMS];
  __shared__ T Cs[128][CH_NB + 1];
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
  G g;
  g.zero();
  auto la = [&](int m, int k) -> T { return L[grow(m) * Mi + k]; };
  auto fa = [&](T raw, int m, int k) -> T { return rvalid(m) ? raw : T(0); };
  auto lb = [&](int k, int n) -> T { return L[(j0i + (n < nb ? n : nb - 1)) * Mi + k]; };
  auto fb = [&](T raw, int k, int n) -> T { return n < nb ? raw : T(0); };
  g.template run<true, true>(0, j0i, la, fa, lb, fb, lds);
  g.for_each([&](int row, int col, T v) {
    const bool ok = rvalid(row) && col < nb;
    const T aval = Ain[(long)grow(row) * M + j0 + (col < nb ? col : nb - 1)];
    T c = ok ? aval - v : T(0);
    if (row < CH_NB && row >= nb && col == row) c = T(1);  // identity padding of a ragged last panel
    Cs[row][col] = c;
  });
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    T a[CH_NB];
#pragma unroll
    for (int j = 0; j < CH_NB; ++j) a[j] = Cs[lane & 31][j];
    const int fail = potrf32_regs<T>(a, lane);
    if (lane < CH_NB) {
#pragma unroll
      for (int j = 0; j < CH_NB; ++j) Cs[lane][j] = a[j];
    }
    if (blockIdx.x == 0 && lane == 0 && fail != 0 && fail <= nb) {
      if (*info == 0) *info = (int)(j0 + fail);
    }
  }
  __syncthreads();
  // panel rows: x L_jj^T = c, forward substitution along the row
  if (threadIdx.x < CH_RB) {
    const int m = CH_NB + threadIdx.x;
    const long r = (long)r0 + threadIdx.x;
    if (r < M) {
      T xr[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) xr[c] = Cs[m][c];
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) {
        T sacc = xr[c];
#pragma unroll
        for (int k = 0; k < c; ++k) sacc -= xr[k] * Cs[c][k];
        xr[c] = sacc / Cs[c][c];
        if (c < nb) L[r * M + j0 + c] = xr[c];
      }
    }
  }
  if (blockIdx.x == 0) {
    // diagonal block (strict upper zeroed) and the rest of these rows' upper part
    for (int idx = threadIdx.x; idx < CH_NB * CH_NB; idx += blockDim.x) {
      const int i = idx / CH_NB, j = idx % CH_NB;
      if (i < nb && j < nb) L[(j0 + i) * M + j0 + j] = j <= i ? Cs[i][j] : T(0);
    }
    const long ncols = M - (j0 + nb);
    for (long idx = threadIdx.x; idx < (long)nb * ncols; idx += blockDim.x) {
      const long i = idx / ncols, c = idx - i * ncols;
      L[(j0 + i) * M + j0 + nb + c] = T(0);
    }
  }
}

template <typename T>
static int cholesky_launch(const T* A, T* L, long B, long M, int* info, hipStream_t stream) {
  HB_REQUIRE(B >= 0 && M >= 0, "hb_cholesky: negative extent");
  HB_REQUIRE(A && L && info, "hb_cholesky: NULL pointer");
  HB_REQUIRE(B <= 65535, "hb_cholesky: batch too large");
  HB_REQUIRE(M * M < 2147483647L, "hb_cholesky: matrix too large for 32-bit indexing");
  if (B == 0) return 0;
  HB_HIP(hipMemsetAsync(info, 0, sizeof(int) * B, stream));
  for (long j0 = 0; j0 < M; j0 += CH_NB) {
    const long below = M - j0 - CH_NB;
    const int gx = below > 0 ? hb_cdiv(below, CH_RB) : 1;
    hipLaunchKernelGGL(chol_panel_kernel<T>, dim3(gx, (unsigned)B), dim3(256), 0, stream, A, L, M, j0, info);
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
  __shared__ T Ls[CH_NB][CH_NB + 1];
  __shared__ T Ws[CH_NB][CH_NB + 1];
  const long b = blockIdx.y;
  L += b * M * M;
  W += b * M * M;
  const long i0 = (long)blockIdx.x * CH_NB;
  const int nb = (int)((M - i0) < CH_NB ? (M - i0) : CH_NB);
  for (int idx = threadIdx.x; idx < CH_NB * CH_NB; idx += 64) {
    const int i = idx / CH_NB, j = idx % CH_NB;
    Ls[i][j] = (i < nb && j < nb && j <= i) ? L[(i0 + i) * M + i0 + j] : (i == j ? T(1) : T(0));
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < nb) {
    for (int i = 0; i < c; ++i) Ws[i][c] = T(0);
    Ws[c][c] = T(1) / Ls[c][c];
    for (int i = c + 1; i < nb; ++i) {
      T s = T(0);
      for (int k = c; k < i; ++k) s += Ls[i][k] * Ws[k][c];
      Ws[i][c] = -s / Ls[i][i];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < CH_NB * CH_NB; idx += 64) {
    const int i = idx / CH_NB, j = idx % CH_NB;
    if (i < nb && j < nb) W[(i0 + i) * M + i0 + j] = Ws[i][j];
  }
}

// phase 0: Tm[r0+i, c0+j] = sum_k L[r0+i, c0+k] W[c0+k, c0+j]      (k >= j: W11 lower)
// phase 1: W [r0+i, c0+j] = -sum_k W[r0+i, r0+k] Tm[r0+k, c0+j]    (k <= i: W22 lower)
template <typename T, int PHASE>
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
    g.template run<true, false>(tj, s, la, fa, lb, fb, lds);
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
    g.template run<true, false>(0, kend, la, fa, lb, fb, lds);
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
  HB_HIP(hipMemsetAsync(W, 0, sizeof(T) * B * M * M, stream));
  const int nblk = hb_cdiv(M, CH_NB);
  hipLaunchKernelGGL(trinv_diag_kernel<T>, dim3(nblk, (unsigned)B), dim3(64), 0, stream, L, W, M);
  HB_LAUNCH_CHECK();
  for (long s = CH_NB; s < M; s *= 2) {
    const int pairs = hb_cdiv(M, 2 * s);
    const int tps = hb_cdiv(s, 64);
    dim3 grid(tps * tps, pairs, (unsigned)B);
    hipLaunchKernelGGL((trinv_level_kernel<T, 0>), grid, dim3(256), 0, stream, L, W, ws, M, s);
    HB_LAUNCH_CHECK();
    hipLaunchKernelGGL((trinv_level_kernel<T, 1>), grid, dim3(256), 0, stream, L, W, ws, M, s);
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
