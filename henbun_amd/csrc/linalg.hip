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
  const int tiles_n = (int)((a.N + 63) / 64);
  const long row0 = (long)(blockIdx.x / tiles_n) * 64;
  const long col0 = (long)(blockIdx.x % tiles_n) * 64;
  const long b = blockIdx.y;
  const int s = blockIdx.z;
  if ((a.flags & HB_MM_LOWER_OUT) && col0 > row0 + 63) return;
  long kchunk = (a.K + a.S - 1) / a.S;
  kchunk = ((kchunk + G::BK - 1) / G::BK) * G::BK;
  const long kbeg = s * kchunk;
  long kend = kbeg + kchunk;
  if (kend > a.K) kend = a.K;
  const T* Ab = a.A + b * a.sA;
  const T* Bb = a.B + b * a.sB;
  G g;
  g.zero();
  auto fa = [&](int m, long k) -> T {
    const long r = row0 + m;
    if (r >= a.M) return T(0);
    return TA ? Ab[k * a.lda + r] : Ab[r * a.lda + k];
  };
  auto fb = [&](long k, int n) -> T {
    const long c = col0 + n;
    if (c >= a.N) return T(0);
    return TB ? Bb[c * a.ldb + k] : Bb[k * a.ldb + c];
  };
  g.template run<!TA, TB>(kbeg, kend, fa, fb, lds, lds + G::BK * G::LDA);
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

template <typename T>
__device__ __forceinline__ void potrf32_wave(volatile T (*Cs)[CH_NB + 1], int nb, int* info, long j0, bool report) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  for (int k = 0; k < nb; ++k) {
    const T p = Cs[k][k];
    if (report && lane == 0 && !(p > T(0))) {
      if (*info == 0) *info = (int)(j0 + k + 1);
    }
    const T lkk = hb_sqrt(p);
    const T inv = T(1) / lkk;
    __builtin_amdgcn_wave_barrier();
    if (h == 0 && r > k && r < nb) Cs[r][k] = Cs[r][k] * inv;
    if (lane == 0) Cs[k][k] = lkk;
    __builtin_amdgcn_wave_barrier();
    if (r > k && r < nb) {
      const T lrk = Cs[r][k];
      for (int j = k + 1 + h; j <= r; j += 2) Cs[r][j] = Cs[r][j] - lrk * Cs[j][k];
    }
    __builtin_amdgcn_wave_barrier();
  }
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
  const int nb = (int)((M - j0) < CH_NB ? (M - j0) : CH_NB);
  const long r0 = j0 + CH_NB + (long)blockIdx.x * CH_RB;
  auto grow = [&](int m) -> long { return m < CH_NB ? j0 + m : r0 + (m - CH_NB); };
  auto rvalid = [&](int m) -> bool { return m < CH_NB ? m < nb : (r0 + (m - CH_NB)) < M; };
  G g;
  g.zero();
  auto fa = [&](int m, long k) -> T { return rvalid(m) ? L[grow(m) * M + k] : T(0); };
  auto fb = [&](long k, int n) -> T { return n < nb ? L[(j0 + n) * M + k] : T(0); };
  g.template run<true, true>(0, j0, fa, fb, lds, lds + G::BK * G::LDA);
  g.for_each([&](int row, int col, T v) {
    T aval = T(0);
    if (rvalid(row) && col < nb) aval = Ain[grow(row) * M + j0 + col];
    Cs[row][col] = aval - v;
  });
  __syncthreads();
  if (threadIdx.x < 64) potrf32_wave<T>((volatile T(*)[CH_NB + 1])Cs, nb, info, j0, blockIdx.x == 0);
  __syncthreads();
  // panel rows: x L_jj^T = c, forward substitution along the row
  if (threadIdx.x < CH_RB) {
    const int m = CH_NB + threadIdx.x;
    const long r = r0 + threadIdx.x;
    if (r < M) {
      T xr[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) xr[c] = Cs[m][c];
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) {
        if (c < nb) {
          T sacc = xr[c];
#pragma unroll
          for (int k = 0; k < c; ++k) sacc -= xr[k] * Cs[c][k];
          xr[c] = sacc / Cs[c][c];
          L[r * M + j0 + c] = xr[c];
        }
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
                                                          T* __restrict__ Tm, long M, long s) {
  typedef TileGemm<T, 64, 64, 16, 2, 2> G;
  __shared__ T lds[G::LDS_ELEMS];
  const long b = blockIdx.z;
  L += b * M * M;
  W += b * M * M;
  Tm += b * M * M;
  const long pair = blockIdx.y;
  const long c0 = 2 * pair * s, r0 = c0 + s;
  const int tps = (int)((s + 63) / 64);  // tiles per side
  const long ti = (long)(blockIdx.x / tps) * 64, tj = (long)(blockIdx.x % tps) * 64;
  if (r0 + ti >= M) return;  // tile entirely below the matrix
  G g;
  g.zero();
  if (PHASE == 0) {
    auto fa = [&](int m, long k) -> T {
      const long r = r0 + ti + m, c = c0 + k;
      return (ti + m < s && r < M) ? L[r * M + c] : T(0);
    };
    auto fb = [&](long k, int n) -> T {
      const long j = tj + n;
      return (j < s && k >= j) ? W[(c0 + k) * M + c0 + j] : T(0);
    };
    g.template run<true, false>(tj, s, fa, fb, lds, lds + G::BK * G::LDA);
    g.for_each([&](int row, int col, T v) {
      const long i = ti + row, j = tj + col;
      if (i < s && j < s && r0 + i < M) Tm[(r0 + i) * M + c0 + j] = v;
    });
  } else {
    long kend = ti + 64;
    if (kend > s) kend = s;
    auto fa = [&](int m, long k) -> T {
      const long i = ti + m;
      return (i < s && k <= i && r0 + i < M) ? W[(r0 + i) * M + r0 + k] : T(0);
    };
    auto fb = [&](long k, int n) -> T {
      const long j = tj + n;
      return (j < s && r0 + k < M) ? Tm[(r0 + k) * M + c0 + j] : T(0);
    };
    g.template run<true, false>(0, kend, fa, fb, lds, lds + G::BK * G::LDA);
    g.for_each([&](int row, int col, T v) {
      const long i = ti + row, j = tj + col;
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
