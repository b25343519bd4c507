// xoroshiro128+ per-lane noise source (north star: "xoroshiro-per-lane noise").
// Stands in for tf.random_normal (reference variationals.py:107,127;
// gp/gp.py:132,138,142) and np.random.randint (model.py:147-153).
//
// Stream layout: lane t of `nlanes` owns pairs p = t, t+nlanes, t+2*nlanes ...
// of the output (elements 2p, 2p+1), so a draw of n values is a pure function
// of (state, n) whatever the launch geometry -- the fused samplers in
// variational.hip / sgp.hip follow the same assignment (rng_pairs.cuh).
#include "common.cuh"
#include "rng_pairs.cuh"
#include "../../include/henbun_hip.h"

__global__ void __launch_bounds__(256) rng_init_kernel(uint64_t* state, long nlanes, uint64_t seed,
                                                       uint64_t stream_id) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nlanes) return;
  uint64_t x = seed ^ (stream_id * 0xD1342543DE82EF95ull) ^ ((uint64_t)t * 0x9E3779B97F4A7C15ull);
  x = hb_splitmix64(x) ^ (uint64_t)t;
  uint64_t a = hb_splitmix64(x);
  uint64_t b = hb_splitmix64(x);
  if (a == 0 && b == 0) b = 0x9E3779B97F4A7C15ull;
  state[t] = a;
  state[nlanes + t] = b;
}

extern "C" int hb_rng_init(uint64_t* state, long nlanes, uint64_t seed, uint64_t stream_id, void* stream) {
  HB_REQUIRE(state != nullptr && nlanes > 0, "hb_rng_init: bad state/nlanes");
  hipLaunchKernelGGL(rng_init_kernel, dim3(hb_cdiv(nlanes, 256)), dim3(256), 0, (hipStream_t)stream, state, nlanes,
                     seed, stream_id);
  HB_LAUNCH_CHECK();
  return 0;
}

template <typename T>
__global__ void __launch_bounds__(256) rng_normal_kernel(uint64_t* state, long nlanes, T* out, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nlanes) return;
  const long npairs = (n + 1) / 2;
  if (t >= npairs) return;  // lane draws nothing: state untouched
  HbRng g = rng_load(state, nlanes, t);
  for (long p = t; p < npairs; p += nlanes) {
    T z0, z1;
    g.normal2(z0, z1);
    out[2 * p] = (T)z0;
    if (2 * p + 1 < n) out[2 * p + 1] = (T)z1;
  }
  rng_store(state, nlanes, t, g);
}

extern "C" int hb_rng_normal_f32(uint64_t* state, long nlanes, float* out, long n, void* stream) {
  HB_REQUIRE(state != nullptr && nlanes > 0 && n >= 0, "hb_rng_normal: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(rng_normal_kernel<float>, dim3(hb_cdiv(nlanes, 256)), dim3(256), 0, (hipStream_t)stream, state,
                     nlanes, out, n);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_rng_normal_f64(uint64_t* state, long nlanes, double* out, long n, void* stream) {
  HB_REQUIRE(state != nullptr && nlanes > 0 && n >= 0, "hb_rng_normal: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(rng_normal_kernel<double>, dim3(hb_cdiv(nlanes, 256)), dim3(256), 0, (hipStream_t)stream, state,
                     nlanes, out, n);
  HB_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) rng_randint_kernel(uint64_t* state, long nlanes, long* out, long n, long lo,
                                                          uint64_t range) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nlanes || t >= n) return;
  HbRng g = rng_load(state, nlanes, t);
  for (long i = t; i < n; i += nlanes) out[i] = lo + (long)__umul64hi(g.next(), range);
  rng_store(state, nlanes, t, g);
}

extern "C" int hb_rng_randint(uint64_t* state, long nlanes, long* out, long n, long lo, long hi, void* stream) {
  HB_REQUIRE(state != nullptr && nlanes > 0 && n >= 0, "hb_rng_randint: bad arguments");
  HB_REQUIRE(hi > lo, "hb_rng_randint: empty range [%ld,%ld)", lo, hi);
  if (n == 0) return 0;
  hipLaunchKernelGGL(rng_randint_kernel, dim3(hb_cdiv(nlanes, 256)), dim3(256), 0, (hipStream_t)stream, state, nlanes,
                     out, n, lo, (uint64_t)(hi - lo));
  HB_LAUNCH_CHECK();
  return 0;
}
