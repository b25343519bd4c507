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
#include "chain.cuh"   // serial chains: a one-workgroup update may be recorded instead of launched

// The update itself (failure containment included) is hb_adam_body in chain_bodies.cuh: the same body runs inside the
// run-time generated serial chains (csrc/jit.hip) where the plan folds the last gradient cluster and Adam into one launch.
template <typename T>
__global__ void __launch_bounds__(256) adam_kernel(T* __restrict__ theta, const T* __restrict__ g, T* __restrict__ m,
                                                   T* __restrict__ v, long n, double lr, double b1, double b2,
                                                   double eps, double gscale, long* t, int tick, const int* info,
                                                   long n_info, const T* dpflag, long* fail) {
  hb_adam_body<T>(theta, g, m, v, n, lr, b1, b2, eps, gscale, t, tick, info, n_info, dpflag, fail,
                  (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, blockIdx.x == 0, gridDim.x == 1);
}

template <typename T>
__global__ void __launch_bounds__(256) adam_vec_kernel(T* __restrict__ theta, const T* __restrict__ g, T* __restrict__ m,
                                                       T* __restrict__ v, long n, double lr, double b1, double b2,
                                                       double eps, double gscale, long* t, const int* info, long n_info,
                                                       const T* dpflag, long* fail) {
  hb_adam_body_vec<T>(theta, g, m, v, n, lr, b1, b2, eps, gscale, t, info, n_info, dpflag, fail,
                      (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, blockIdx.x == 0);
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
  if (hb_chain_recording()) {
    if (n > 0 && n <= HB_CHAIN_ADAM_MAX_N) {
      // the last job of a serial chain: the update of a small parameter set is one workgroup anyway
      HbChainJob j;
      j.kind = HB_CHAIN_ADAM;
      j.is64 = sizeof(T) == 8;
      j.p[0] = theta, j.p[1] = g, j.p[2] = m, j.p[3] = v, j.p[4] = t, j.p[5] = info, j.p[6] = dpflag, j.p[7] = fail;
      j.l[0] = n, j.l[1] = n_info, j.l[2] = tick;
      j.d[0] = lr, j.d[1] = b1, j.d[2] = b2, j.d[3] = eps, j.d[4] = gscale;
      return hb_chain_push(j, stream);
    }
    const int crc = hb_chain_flush(stream);
    if (crc) return crc;
  }
  if (n > 0) {
    // small parameter sets: one block, tick included -- as long as the block's up-front batch (8 elements per thread)
    // covers them: past it the body walks the rest one dependent read-modify-write round trip per 256 elements
    // (12 K parameters at cfg 5: 28.8 us in one block against 4.7 + 4.2 us for the vector kernel + tick)
    grid = n <= 2048 ? 1 : hb_stream_grid(n, 256);
    constexpr long VEC = 16 / (long)sizeof(T);
    const bool vec = grid > 1 && (((uintptr_t)theta | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) == 0;
    if (vec) {
      // 16 bytes per lane and array (scalar accesses ran at 2.0 TB/s on 1.05 M parameters: instruction bound)
      grid = hb_stream_grid(n / VEC, 256);
      hipLaunchKernelGGL(adam_vec_kernel<T>, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, lr, b1, b2, eps, gscale, t,
                         info, n_info, dpflag, fail);
      if (grid == 1) grid = 2;   // the vector kernel never advances the counter itself
    } else {
      hipLaunchKernelGGL(adam_kernel<T>, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, lr, b1, b2, eps, gscale, t,
                         tick, info, n_info, dpflag, fail);
    }
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
