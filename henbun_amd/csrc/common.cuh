// Shared device/host helpers for the henbun_amd HIP kernels (gfx950 / CDNA4 only).
//
// Everything here is wave64: reductions use 64-lane shuffles, MFMA fragments
// follow the gfx950 lane maps (32x32x2 f32, 16x16x4 f64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define HB_WAVE 64
#define HB_MAX_DIMS 6

// ---------------------------------------------------------------------------
// error plumbing (host side).  Entry points return 0 on success, <0 for a bad
// argument, >0 for a HIP runtime failure (hipError_t value).
// ---------------------------------------------------------------------------
void hb_set_error(const char* fmt, ...);
// Diagnostic switches (A/B forms of a dispatch rule, forced tile sizes): a process-wide key -> value table written only
// through hb_debug_set() (include/henbun_hip.h).  Nothing in the library reads the environment.
long hb_debug_get(const char* key, long dflt);

#define HB_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) {                                              \
      hb_set_error(__VA_ARGS__);                                \
      return -1;                                                \
    }                                                           \
  } while (0)

#define HB_LAUNCH_CHECK()                                                   \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      hb_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

#define HB_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e__ = (call);                                                \
    if (e__ != hipSuccess) {                                                \
      hb_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

static inline int hb_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// grid size for grid-stride memory-bound kernels: cap at 8 blocks per CU.
static inline int hb_stream_grid(long n, int block) {
  long g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------
// Device-side zero fill / copy.  hipMemsetAsync / hipMemcpyAsync are NOT used on
// any path that may run inside a hipGraph capture: measured on ROCm 7.2 / gfx950,
// a captured 32-byte hipMemsetAsync replayed an 8-byte pattern of stale host
// stack contents instead of zero (the Cholesky `info` words of a batch of 8).
// Plain kernels capture as kernel nodes with their arguments by value.
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) hb_zero_words_kernel(uint32_t* __restrict__ p, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) p[t] = 0u;
}
static __global__ void __launch_bounds__(256) hb_zero_quads_kernel(uint4* __restrict__ p, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  const uint4 z = {0u, 0u, 0u, 0u};
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) p[t] = z;
}
static __global__ void __launch_bounds__(256) hb_copy_words_kernel(const uint32_t* __restrict__ s,
                                                                  uint32_t* __restrict__ d, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) d[t] = s[t];
}
// bytes must be a multiple of 4 (all callers pass whole float / double / int arrays)
static inline hipError_t hb_zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if (bytes % 16 == 0 && ((uintptr_t)p % 16) == 0) {
    const long n = (long)(bytes / 16);
    hipLaunchKernelGGL(hb_zero_quads_kernel, dim3(hb_stream_grid(n, 256)), dim3(256), 0, stream, (uint4*)p, n);
  } else {
    const long n = (long)(bytes / 4);
    hipLaunchKernelGGL(hb_zero_words_kernel, dim3(hb_stream_grid(n, 256)), dim3(256), 0, stream, (uint32_t*)p, n);
  }
  return hipGetLastError();
}
static inline hipError_t hb_copy_async(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  const long n = (long)(bytes / 4);
  hipLaunchKernelGGL(hb_copy_words_kernel, dim3(hb_stream_grid(n, 256)), dim3(256), 0, stream, (const uint32_t*)src,
                     (uint32_t*)dst, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// wave / block reductions (sum, max).  Deterministic (no atomics).
// ---------------------------------------------------------------------------
#include "ew_math.cuh"  // wave_sum, block_sum, hb_exp ... hb_sigmoid

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    T o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T block_max(T v, T* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  T r = smem[lane < nw ? lane : 0];
  r = wave_max(r);
  return r;
}

// ---------------------------------------------------------------------------
// MFMA abstraction.  One wave computes a TM x TN tile, consuming TK of the
// contraction per instruction.  Lane l supplies A[row = l % TM][k = l / TM]
// and B[k = l / TN][col = l % TN]; accumulator register r of lane l holds
// C[acc_row(l, r)][acc_col(l)].
// ---------------------------------------------------------------------------
template <typename T>
struct Mma;

template <>
struct Mma<float> {
  static constexpr int TM = 32, TN = 32, TK = 2, NACC = 16;
  typedef float Acc __attribute__((ext_vector_type(16)));
  __device__ static __forceinline__ Acc mma(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ int acc_row(int lane, int r) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  }
  __device__ static __forceinline__ int acc_col(int lane) { return lane & 31; }
};

template <>
struct Mma<double> {
  static constexpr int TM = 16, TN = 16, TK = 4, NACC = 4;
  typedef double Acc __attribute__((ext_vector_type(4)));
  __device__ static __forceinline__ Acc mma(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ int acc_row(int lane, int r) {
    return (lane >> 4) + 4 * r;
  }
  __device__ static __forceinline__ int acc_col(int lane) { return lane & 15; }
};

#include "rng_core.cuh"      // HbRng, rng_load / rng_store
#include "chain_bodies.cuh"  // bodies of the small kernels (shared with the run-time generated serial chains)

__host__ __device__ static inline uint64_t hb_splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
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

// fp32 -> three bf16 terms hi + mid + lo (each rounded to nearest even of what is left): the operand form of the
// "bf16x3" contractions, whose six significant cross products reproduce the fp32-operand result to fp32 accuracy
// (profiles/r01_bf16_split_study.txt) at the bf16 MFMA rate.
__device__ __forceinline__ void hb_split_bf16x3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}
