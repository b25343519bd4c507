// K3: stationary Gram matrices and their VJP.
//
// Reference: Henbun/gp/kernels.py:54-84 (square_dist: |a|^2+|b|^2-2ab^T on
// X/ell), :110-111 (UnitRBF.K = exp(-r2/2)), :122-131 (UnitCsymRBF).  The
// squared distance is formed directly as sum_d((x_id-x2_jd)/ell_d)^2: same
// value, better conditioned, inside the reference's own atol.
// These kernels serve the small M x M (and test-sized) Grams; the M x n block
// on the hot path is built inside sgp.hip and never written to memory.
#include "common.cuh"
#include "../../include/henbun_hip.h"
#include "chain.cuh"   // serial chains: the lengthscale fold may be recorded instead of launched
#include "gram_value.cuh"

template <typename T>
__global__ void __launch_bounds__(256) gram_fwd_kernel(int kind, const T* __restrict__ X, long sX,
                                                       const T* __restrict__ X2, long sX2, const T* __restrict__ ell,
                                                       long sEll, long dl, T* __restrict__ K, long B, long n, long n2,
                                                       long d, T diag_add) {
  const long total = B * n * n2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long b = t / (n * n2);
    const long rem = t - b * n * n2;
    const long i = rem / n2, j = rem - i * n2;
    const T kv = gram_value<T>(kind, X + b * sX + i * d, X2 + b * sX2 + j * d, ell + b * sEll, dl, d);
    K[t] = (i == j) ? kv + diag_add : kv;  // jitter (gp/kernels.py:101) folded in
  }
}

template <typename T>
static int gram_fwd(int kind, const T* X, long sX, const T* X2, long sX2, const T* ell, long sEll, long dl, T* K,
                    long B, long n, long n2, long d, double diag_add, hipStream_t stream) {
  HB_REQUIRE(kind >= HB_KERN_RBF && kind <= HB_KERN_SQDIST, "hb_gram_fwd: unknown kernel kind %d", kind);
  HB_REQUIRE(B >= 0 && n >= 0 && n2 >= 0 && d >= 1, "hb_gram_fwd: bad extents");
  HB_REQUIRE(dl == 1 || dl == d, "hb_gram_fwd: lengthscales must have 1 or d=%ld entries, got %ld", d, dl);
  HB_REQUIRE(X && X2 && ell && K, "hb_gram_fwd: NULL pointer");
  const long total = B * n * n2;
  if (total == 0) return 0;
  HB_REQUIRE(sEll == 0 || sEll == dl, "hb_gram_fwd: lengthscale batch stride must be 0 or dl");
  hipLaunchKernelGGL(gram_fwd_kernel<T>, dim3(hb_stream_grid(total, 256)), dim3(256), 0, stream, kind, X, sX, X2, sX2,
                     ell, sEll, dl, K, B, n, n2, d, (T)diag_add);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_gram_fwd_f32(int kind, const float* X, long sX, const float* X2, long sX2, const float* ell,
                               long sEll, long dl, float* K, long B, long n, long n2, long d, double diag_add,
                               void* stream) {
  return gram_fwd<float>(kind, X, sX, X2, sX2, ell, sEll, dl, K, B, n, n2, d, diag_add, (hipStream_t)stream);
}
extern "C" int hb_gram_fwd_f64(int kind, const double* X, long sX, const double* X2, long sX2, const double* ell,
                               long sEll, long dl, double* K, long B, long n, long n2, long d, double diag_add,
                               void* stream) {
  return gram_fwd<double>(kind, X, sX, X2, sX2, ell, sEll, dl, K, B, n, n2, d, diag_add, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// VJP.  With a = x/ell, b = x2/ell, E- = exp(-|a-b|^2/2), E+ = exp(-|a+b|^2/2):
//   dK/dx_k  = [-(a_k-b_k) E-  - (a_k+b_k) E+] / ell_k
//   dK/dx2_k = [+(a_k-b_k) E-  - (a_k+b_k) E+] / ell_k
//   dK/dell_k= [ (a_k-b_k)^2 E- + (a_k+b_k)^2 E+] / ell_k   (summed over k when dl == 1)
// (E+ only for the cylindrically symmetric kernel.)
// One block per (b, row) of the side being differentiated; threads stride the
// other side; deterministic block reductions, no atomics.
// ---------------------------------------------------------------------------
#define HB_GRAM_MAXD 8

template <typename T>
__global__ void __launch_bounds__(256)
gram_bwd_side_kernel(int kind, int side, const T* __restrict__ X, long sX, const T* __restrict__ X2, long sX2,
                     const T* __restrict__ ell, long sEll, long dl, const T* __restrict__ Kbar, T* __restrict__ out,
                     T* __restrict__ ell_partial, long n, long n2, long d) {
  // side 0: block row i of X, loop j over X2; side 1: block row j of X2, loop i over X;
  // side 2: X2 IS X (same points): side 0 whose point gradient also takes the transposed entry Kbar[j,i],
  // i.e. the sum of both sides in one pass (the lengthscale partial still counts every pair once).
  // side 3: side 2 for a SYMMETRIC Kbar (HB_KERN_KBAR_SYMMETRIC).
  __shared__ T smem[16];
  const long b = blockIdx.y;
  const long row = blockIdx.x;
  ell += b * sEll;
  const long nother = side != 1 ? n2 : n;
  // reciprocal lengthscales once per thread (they sat as IEEE divisions inside both inner loops)
  T ilv[HB_GRAM_MAXD];
  for (long k0 = 0; k0 < d; k0 += HB_GRAM_MAXD) {
#pragma unroll
    for (int k = 0; k < HB_GRAM_MAXD; ++k) ilv[k] = k0 + k < d ? T(1) / ell[dl == 1 ? 0 : k0 + k] : T(0);
    T gacc[HB_GRAM_MAXD], lacc[HB_GRAM_MAXD];
#pragma unroll
    for (int k = 0; k < HB_GRAM_MAXD; ++k) gacc[k] = lacc[k] = T(0);
#pragma unroll 4
    for (long o = threadIdx.x; o < nother; o += blockDim.x) {
      const long i = side != 1 ? row : o;
      const long j = side != 1 ? o : row;
      const T* xi = X + b * sX + i * d;
      const T* xj = X2 + b * sX2 + j * d;
      T r2 = T(0), r2m = T(0);
      for (long k = 0; k < d; ++k) {
        const T il = (d <= HB_GRAM_MAXD) ? ilv[k] : T(1) / ell[dl == 1 ? 0 : k];   // (d <= 8: k0 == 0, ilv covers every k)
        const T a = xi[k] * il, bb = xj[k] * il;
        r2 += (a - bb) * (a - bb);
        r2m += (a + bb) * (a + bb);
      }
      const T kb = Kbar[(b * n + i) * n2 + j];
      // squared distance: d r2 = 2 (a-b) d(a-b), i.e. the RBF formulas with E- := -2
      const T km = kind == HB_KERN_SQDIST ? T(-2) : hb_exp(T(-0.5) * r2);
      const T kp = kind == HB_KERN_CSYM_RBF ? hb_exp(T(-0.5) * r2m) : T(0);
      const T em = kb * km, ep = kb * kp;
      // point gradient: both orientations (side 3: Kbar is symmetric -- the Cholesky VJP's output -- so the
      // transposed entry, a strided read of one cache line per thread, is the entry itself)
      const T kbg = side == 2 ? kb + Kbar[(b * n + j) * n2 + i] : (side == 3 ? kb + kb : kb);
      const T gm = kbg * km, gp = kbg * kp;
#pragma unroll
      for (int k = 0; k < HB_GRAM_MAXD; ++k) {
        if (k0 + k < d) {
          const T il = ilv[k];
          const T a = xi[k0 + k] * il, bb = xj[k0 + k] * il;
          const T dm = a - bb, dp = a + bb;
          if (side != 1)
            gacc[k] += (-dm * gm - dp * gp) * il;
          else
            gacc[k] += (dm * gm - dp * gp) * il;
          lacc[k] += (dm * dm * em + dp * dp * ep) * il;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < HB_GRAM_MAXD; ++k) {
      if (k0 + k < d) {
        const T g = block_sum(gacc[k], smem);
        if (out && threadIdx.x == 0) out[(b * (side != 1 ? n : n2) + row) * d + k0 + k] = g;
        if (ell_partial) {
          const T l = block_sum(lacc[k], smem);
          // partial layout [B*rows, d]; reduced (and folded to dl) afterwards
          if (threadIdx.x == 0) ell_partial[(b * (side != 1 ? n : n2) + row) * d + k0 + k] = l;
        }
      }
    }
  }
}

// ellbar[c] = sum_r partial[r, c (or all columns when dl == 1)]
template <typename T>
__global__ void __launch_bounds__(256) gram_ell_finish_kernel(const T* __restrict__ partial, long rows, long d, long dl,
                                                              T* __restrict__ ellbar) {
  __shared__ T smem[16];
  // one group of rows per batch entry when the lengthscales are per batch (blockIdx.y)
  hb_gram_ell_body<T>(partial + (long)blockIdx.y * rows * d, rows, d, dl, (long)blockIdx.x, ellbar + (long)blockIdx.y * dl, smem);
}

template <typename T>
static int gram_bwd(int kind_flags, const T* X, long sX, const T* X2, long sX2, const T* ell, long sEll, long dl,
                    const T* Kbar, T* Xbar, T* X2bar, T* ellbar, long B, long n, long n2, long d, T* ws,
                    hipStream_t stream) {
  const int kind = kind_flags & ~HB_KERN_KBAR_SYMMETRIC;
  HB_REQUIRE(kind >= HB_KERN_RBF && kind <= HB_KERN_SQDIST, "hb_gram_bwd: unknown kernel kind %d", kind);
  HB_REQUIRE(B >= 0 && n >= 0 && n2 >= 0 && d >= 1, "hb_gram_bwd: bad extents");
  HB_REQUIRE(dl == 1 || dl == d, "hb_gram_bwd: lengthscales must have 1 or d entries");
  HB_REQUIRE(sEll == 0 || sEll == dl, "hb_gram_bwd: lengthscale batch stride must be 0 or dl");
  HB_REQUIRE(X && X2 && ell && Kbar, "hb_gram_bwd: NULL pointer");
  HB_REQUIRE(!ellbar || ws, "hb_gram_bwd: ellbar needs workspace");
  HB_REQUIRE(B <= 65535, "hb_gram_bwd: batch too large");
  if (B == 0) return 0;
  if (hb_chain_recording()) {
    const int crc = hb_chain_flush(stream);   // whatever was recorded before this call runs before its first launch
    if (crc) return crc;
  }
  if (n == 0 || n2 == 0) {
    if (Xbar && n > 0) HB_HIP(hb_zero_async(Xbar, sizeof(T) * B * n * d, stream));
    if (X2bar && n2 > 0) HB_HIP(hb_zero_async(X2bar, sizeof(T) * B * n2 * d, stream));
    if (ellbar) HB_HIP(hb_zero_async(ellbar, sizeof(T) * dl * (sEll != 0 ? B : 1), stream));
    return 0;
  }
  // X2bar == Xbar: X2 is X and the caller wants the total point gradient in one array
  const bool sym = Xbar && X2bar == Xbar;
  const bool kbar_sym = (kind_flags & HB_KERN_KBAR_SYMMETRIC) != 0;
  HB_REQUIRE(!kbar_sym || sym, "hb_gram_bwd: HB_KERN_KBAR_SYMMETRIC only applies to the one-pass form (X2 == X, X2bar == Xbar)");
  HB_REQUIRE(!sym || (X == X2 && sX == sX2 && n == n2), "hb_gram_bwd: Xbar == X2bar requires X2 == X");
  // side 0 pass also produces the lengthscale partials
  if (Xbar || ellbar) {
    hipLaunchKernelGGL(gram_bwd_side_kernel<T>, dim3(n, B), dim3(256), 0, stream, kind, sym ? (kbar_sym ? 3 : 2) : 0, X, sX, X2, sX2, ell, sEll,
                       dl, Kbar, Xbar, ellbar ? ws : (T*)nullptr, n, n2, d);
    HB_LAUNCH_CHECK();
    if (ellbar) {
      // d(K)/d(ell) = sum lacc (positive sign; see header comment)
      const long rows_ = sEll != 0 ? n : B * n, groups_ = sEll != 0 ? B : 1;
      if (hb_chain_recording() && rows_ * d * groups_ <= HB_CHAIN_ELL_MAX_N && groups_ * dl <= 16 && !(X2bar && !sym)) {
        // the fold opens a serial chain (the gradient cluster and Adam follow in the same launch)
        HbChainJob j;
        j.kind = HB_CHAIN_GRAM_ELL;
        j.is64 = sizeof(T) == 8;
        j.p[0] = ws, j.p[1] = ellbar;
        j.l[0] = rows_, j.l[1] = d, j.l[2] = dl, j.l[3] = groups_;
        return hb_chain_push(j, stream);
      }
      if (sEll != 0)
        hipLaunchKernelGGL(gram_ell_finish_kernel<T>, dim3(dl, B), dim3(256), 0, stream, ws, n, d, dl, ellbar);
      else
        hipLaunchKernelGGL(gram_ell_finish_kernel<T>, dim3(dl, 1), dim3(256), 0, stream, ws, B * n, d, dl, ellbar);
      HB_LAUNCH_CHECK();
    }
  }
  if (X2bar && !sym) {
    hipLaunchKernelGGL(gram_bwd_side_kernel<T>, dim3(n2, B), dim3(256), 0, stream, kind, 1, X, sX, X2, sX2, ell, sEll,
                       dl, Kbar, X2bar, (T*)nullptr, n, n2, d);
    HB_LAUNCH_CHECK();
  }
  return 0;
}
// The lengthscale fold alone: ellbar[c] = sum over the row partials [groups][rows][d] that a Gram VJP left behind (the last
// launch of hb_gram_bwd; stand-alone for the Gram VJP that ran in a product's epilogue, hb_matmul_gram_vjp).  Chain-aware.
template <typename T>
static int gram_ell_fold(const T* partial, long rows, long d, long dl, long groups, T* ellbar, hipStream_t stream) {
  HB_REQUIRE(partial && ellbar && rows >= 1 && d >= 1 && groups >= 1 && (dl == 1 || dl == d), "hb_gram_ell_fold: bad arguments");
  if (hb_chain_recording() && rows * d * groups <= HB_CHAIN_ELL_MAX_N && groups * dl <= 16) {
    HbChainJob j;
    j.kind = HB_CHAIN_GRAM_ELL;
    j.is64 = sizeof(T) == 8;
    j.p[0] = partial, j.p[1] = ellbar;
    j.l[0] = rows, j.l[1] = d, j.l[2] = dl, j.l[3] = groups;
    return hb_chain_push(j, stream);
  }
  if (hb_chain_recording()) {
    const int crc = hb_chain_flush(stream);
    if (crc) return crc;
  }
  hipLaunchKernelGGL(gram_ell_finish_kernel<T>, dim3(dl, groups), dim3(256), 0, stream, partial, rows, d, dl, ellbar);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_gram_ell_fold_f32(const float* partial, long rows, long d, long dl, long groups, float* ellbar, void* stream) {
  return gram_ell_fold<float>(partial, rows, d, dl, groups, ellbar, (hipStream_t)stream);
}
extern "C" int hb_gram_ell_fold_f64(const double* partial, long rows, long d, long dl, long groups, double* ellbar, void* stream) {
  return gram_ell_fold<double>(partial, rows, d, dl, groups, ellbar, (hipStream_t)stream);
}

extern "C" int hb_gram_bwd_f32(int kind, const float* X, long sX, const float* X2, long sX2, const float* ell,
                               long sEll, long dl, const float* Kbar, float* Xbar, float* X2bar, float* ellbar,
                               long B, long n, long n2, long d, float* ws, void* stream) {
  return gram_bwd<float>(kind, X, sX, X2, sX2, ell, sEll, dl, Kbar, Xbar, X2bar, ellbar, B, n, n2, d, ws,
                         (hipStream_t)stream);
}
extern "C" int hb_gram_bwd_f64(int kind, const double* X, long sX, const double* X2, long sX2, const double* ell,
                               long sEll, long dl, const double* Kbar, double* Xbar, double* X2bar, double* ellbar,
                               long B, long n, long n2, long d, double* ws, void* stream) {
  return gram_bwd<double>(kind, X, sX, X2, sX2, ell, sEll, dl, Kbar, Xbar, X2bar, ellbar, B, n, n2, d, ws,
                          (hipStream_t)stream);
}
