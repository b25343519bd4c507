// K9: fused flat-buffer Adam with the TensorFlow-1 update rule.
//
// Reference: Henbun/model.py:206,220 -- `optimizer.minimize(-objective)` with
// the default tf.train.AdamOptimizer (third party; formula from the TF docs,
// SURVEY.md A.9): epsilon is added to the UN-bias-corrected sqrt(v).  The
// reference pays one op dispatch per variable; here every optimised leaf lives
// in one flat buffer and the update is a single streaming pass
// (4*n*B read + 3*n*B written: HBM-bound).
#include "common.cuh"
#include "../../include/henbun_hip.h"

// Failure containment (reference behaviour: tf.cholesky raises inside session.run BEFORE apply_gradients, so the
// parameters stay at the last good step).  Calls are asynchronous here, so the update itself looks at this step's
// factorisation status words (`info[n_info]`, LAPACK convention, written earlier in the same stream / graph), at
// the all-reduced failure flag of the other ranks (`dpflag`, nullable) and at the sticky record `fail[2]`
// (nullable): if any is non-zero the launch is a no-op -- theta, m, v and the step counter keep their values --
// and `fail` records the first failing step {t+1, first non-zero status seen}.  Every later step is then a no-op
// too, until the host clears `fail`.
template <typename T>
__device__ __forceinline__ int adam_step_blocked(const long* t, const int* info, long n_info, const T* dpflag,
                                                 long* fail, bool record) {
  int bad = 0, what = 0;
  if (fail != nullptr && fail[0] != 0) bad = 1;
  if (dpflag != nullptr && dpflag[0] != (T)0) { bad = 1; what = -1; }
  for (long i = threadIdx.x; i < n_info; i += blockDim.x) {
    const int w = info[i];
    if (w != 0) { bad = 1; what = w; }
  }
  const int any = __syncthreads_or(bad);
  if (any && record && fail != nullptr) {
    // one writer: the lowest thread that saw a status word (or thread 0 for the flag-only case)
    __shared__ int who;
    if (threadIdx.x == 0) who = blockDim.x;
    __syncthreads();
    if (what != 0) atomicMin(&who, (int)threadIdx.x);
    __syncthreads();
    const int writer = who == (int)blockDim.x ? 0 : who;
    if ((int)threadIdx.x == writer && fail[0] == 0) {
      fail[1] = (long)what;
      fail[0] = t[0] + 1;
    }
  }
  return any;
}

template <typename T>
__global__ void __launch_bounds__(256) adam_kernel(T* __restrict__ theta, const T* __restrict__ g, T* __restrict__ m,
                                                   T* __restrict__ v, long n, double lr, double b1, double b2,
                                                   double eps, double gscale, long* t, int tick, const int* info,
                                                   long n_info, const T* dpflag, long* fail) {
  // The first batch of operands (8 elements per thread: a 2048-parameter model in one go) and the step counter are
  // requested BEFORE the status check: the check, the counter and the update were three dependent memory round
  // trips in a kernel whose arithmetic is a few hundred cycles.
  constexpr int U = 8;
  const long stride = (long)gridDim.x * blockDim.x;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  T g0[U], m0[U], v0[U], th0[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const long i = i0 + k * stride, ic = i < n ? i : (n > 0 ? n - 1 : 0);
    g0[k] = g[ic], m0[k] = m[ic], v0[k] = v[ic], th0[k] = theta[ic];
  }
  const long tnow = t[0];
  if (adam_step_blocked<T>(t, info, n_info, dpflag, fail, blockIdx.x == 0)) return;
  const double tt = (double)(tnow + 1);
  const T lr_t = (T)(lr * sqrt(1.0 - pow(b2, tt)) / (1.0 - pow(b1, tt)));
  const T c1 = (T)b1, c2 = (T)b2, d1 = (T)(1.0 - b1), d2 = (T)(1.0 - b2), e = (T)eps, gs = (T)gscale;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const long i = i0 + k * stride;
    if (i < n) {
      const T gi = g0[k] * gs;
      const T mi = c1 * m0[k] + d1 * gi;
      const T vi = c2 * v0[k] + d2 * gi * gi;
      m[i] = mi;
      v[i] = vi;
      theta[i] = th0[k] - lr_t * mi / (hb_sqrt(vi) + e);
    }
  }
  for (long i = i0 + U * stride; i < n; i += stride) {
    const T gi = g[i] * gs;
    const T mi = c1 * m[i] + d1 * gi;
    const T vi = c2 * v[i] + d2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    theta[i] -= lr_t * mi / (hb_sqrt(vi) + e);
  }
  if (gridDim.x == 1 && tick) {
    // a single block owns the whole update: it advances the step counter itself (every thread has read
    // t[0] by the barrier), saving the separate tick launch
    __syncthreads();
    if (threadIdx.x == 0) t[0] += 1;
  }
}

template <typename T>
__global__ void __launch_bounds__(64) adam_tick_kernel(long* t, const int* info, long n_info, const T* dpflag,
                                                       const long* fail) {
  // runs after a multi-block update: `fail` was set by that update if this step was blocked
  if (adam_step_blocked<T>(t, info, n_info, dpflag, const_cast<long*>(fail), false)) return;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] += 1;
}

template <typename T>
static int adam_launch(T* theta, const T* g, T* m, T* v, long n, double lr, double b1, double b2, double eps,
                       double gscale, long* t, int tick, const int* info, long n_info, const T* dpflag, long* fail,
                       hipStream_t stream) {
  HB_REQUIRE(n >= 0, "hb_adam_step: n < 0");
  HB_REQUIRE(theta && g && m && v && t, "hb_adam_step: NULL pointer");
  HB_REQUIRE(n_info >= 0 && (n_info == 0 || info != nullptr), "hb_adam_step: info/n_info");
  int grid = 0;
  if (n > 0) {
    grid = n <= 16384 ? 1 : hb_stream_grid(n, 256);  // small parameter sets: one block, tick included
    hipLaunchKernelGGL(adam_kernel<T>, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, lr, b1, b2, eps, gscale, t,
                       tick, info, n_info, dpflag, fail);
    HB_LAUNCH_CHECK();
  }
  if (grid != 1 && tick) {
    hipLaunchKernelGGL(adam_tick_kernel<T>, dim3(1), dim3(64), 0, stream, t, info, n_info, dpflag, fail);
    HB_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int hb_adam_step_f32(float* theta, const float* g, float* m, float* v, long n, double lr, double b1,
                                double b2, double eps, double gscale, long* t, int tick, const int* info, long n_info,
                                const float* dpflag, long* fail, void* stream) {
  return adam_launch<float>(theta, g, m, v, n, lr, b1, b2, eps, gscale, t, tick, info, n_info, dpflag, fail,
                            (hipStream_t)stream);
}
extern "C" int hb_adam_step_f64(double* theta, const double* g, double* m, double* v, long n, double lr, double b1,
                                double b2, double eps, double gscale, long* t, int tick, const int* info, long n_info,
                                const double* dpflag, long* fail, void* stream) {
  return adam_launch<double>(theta, g, m, v, n, lr, b1, b2, eps, gscale, t, tick, info, n_info, dpflag, fail,
                             (hipStream_t)stream);
}
