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

template <typename T>
__global__ void __launch_bounds__(256) adam_kernel(T* __restrict__ theta, const T* __restrict__ g, T* __restrict__ m,
                                                   T* __restrict__ v, long n, double lr, double b1, double b2,
                                                   double eps, double gscale, const long* t) {
  const double tt = (double)(t[0] + 1);
  const T lr_t = (T)(lr * sqrt(1.0 - pow(b2, tt)) / (1.0 - pow(b1, tt)));
  const T c1 = (T)b1, c2 = (T)b2, d1 = (T)(1.0 - b1), d2 = (T)(1.0 - b2), e = (T)eps, gs = (T)gscale;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T gi = g[i] * gs;
    const T mi = c1 * m[i] + d1 * gi;
    const T vi = c2 * v[i] + d2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    theta[i] -= lr_t * mi / (hb_sqrt(vi) + e);
  }
  if (gridDim.x == 1) {
    // a single block owns the whole update: it advances the step counter itself (every thread has read
    // t[0] by the barrier), saving the separate tick launch
    __syncthreads();
    if (threadIdx.x == 0) const_cast<long*>(t)[0] += 1;
  }
}

__global__ void adam_tick_kernel(long* t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] += 1;
}

template <typename T>
static int adam_launch(T* theta, const T* g, T* m, T* v, long n, double lr, double b1, double b2, double eps,
                       double gscale, long* t, hipStream_t stream) {
  HB_REQUIRE(n >= 0, "hb_adam_step: n < 0");
  HB_REQUIRE(theta && g && m && v && t, "hb_adam_step: NULL pointer");
  int grid = 0;
  if (n > 0) {
    grid = n <= 16384 ? 1 : hb_stream_grid(n, 256);  // small parameter sets: one block, tick included
    hipLaunchKernelGGL(adam_kernel<T>, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, lr, b1, b2, eps, gscale, t);
    HB_LAUNCH_CHECK();
  }
  if (grid != 1) {
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, t);
    HB_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int hb_adam_step_f32(float* theta, const float* g, float* m, float* v, long n, double lr, double b1,
                                double b2, double eps, double gscale, long* t, void* stream) {
  return adam_launch<float>(theta, g, m, v, n, lr, b1, b2, eps, gscale, t, (hipStream_t)stream);
}
extern "C" int hb_adam_step_f64(double* theta, const double* g, double* m, double* v, long n, double lr, double b1,
                                double b2, double eps, double gscale, long* t, void* stream) {
  return adam_launch<double>(theta, g, m, v, n, lr, b1, b2, eps, gscale, t, (hipStream_t)stream);
}
