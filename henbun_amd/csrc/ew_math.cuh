// Scalar math helpers, wave / block sums: shared by every kernel file (through common.cuh) AND compiled at run time
// by hiprtc as the prelude of the generated elementwise kernels (csrc/jit.hip) -- keep this file self-contained
// (no #include; only HIP built-ins and the device math library).
#ifndef HB_EW_MATH_CUH
#define HB_EW_MATH_CUH

// Floating-point contraction is switched OFF for everything in this file and back to the HIP default at its end.
// Under the default (fast) a multiply and an add fuse whenever both carry the `contract` flag -- including inside the
// backend's expansion of logf & co., whose flags depend on which identical calls the optimiser happened to merge:
// the same op then rounds differently from one compilation context to the next (seen: 1 ulp in log, 2e-5 in the
// digamma series between the ahead-of-time interpreter and a run-time compiled program).  With contraction off an
// elementwise op returns the same bits wherever it is compiled; fused multiply-adds that are WANTED are written
// hb_fma().
#pragma clang fp contract(off)

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64).  `smem` needs
// 16 elements.  Result valid in every thread.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();  // protect smem reuse across consecutive calls
  if (lane == 0) smem[w] = v;
  __syncthreads();
  T r = (lane < nw) ? smem[lane] : T(0);
  r = wave_sum(r);
  return r;
}

// ---------------------------------------------------------------------------
// math helpers that pick the right precision overload
// ---------------------------------------------------------------------------
__device__ __forceinline__ float hb_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double hb_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T hb_exp(T x);
template <> __device__ __forceinline__ float hb_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double hb_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T hb_log(T x);
template <> __device__ __forceinline__ float hb_log<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double hb_log<double>(double x) { return log(x); }
template <typename T> __device__ __forceinline__ T hb_sqrt(T x);
template <> __device__ __forceinline__ float hb_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double hb_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T hb_abs(T x) { return x < T(0) ? -x : x; }
template <typename T> __device__ __forceinline__ T hb_sign(T x) {
  return x > T(0) ? T(1) : (x < T(0) ? T(-1) : T(0));
}
template <typename T> __device__ __forceinline__ T hb_log1p(T x);
template <> __device__ __forceinline__ float hb_log1p<float>(float x) { return log1pf(x); }
template <> __device__ __forceinline__ double hb_log1p<double>(double x) { return log1p(x); }
template <typename T> __device__ __forceinline__ T hb_tanh(T x);
template <> __device__ __forceinline__ float hb_tanh<float>(float x) { return tanhf(x); }
template <> __device__ __forceinline__ double hb_tanh<double>(double x) { return tanh(x); }
template <typename T> __device__ __forceinline__ T hb_lgamma(T x);
template <> __device__ __forceinline__ float hb_lgamma<float>(float x) { return lgammaf(x); }
template <> __device__ __forceinline__ double hb_lgamma<double>(double x) { return lgamma(x); }
template <typename T> __device__ __forceinline__ T hb_pow(T x, T y);
template <> __device__ __forceinline__ float hb_pow<float>(float x, float y) { return powf(x, y); }
template <> __device__ __forceinline__ double hb_pow<double>(double x, double y) { return pow(x, y); }

// numerically stable softplus log(1+e^x) (tf.nn.softplus)
template <typename T>
__device__ __forceinline__ T hb_softplus(T x) {
  return (x > T(0) ? x : T(0)) + hb_log1p(hb_exp(-hb_abs(x)));
}
template <typename T>
__device__ __forceinline__ T hb_sigmoid(T x) {
  if (x >= T(0)) {
    return T(1) / (T(1) + hb_exp(-x));
  } else {
    T e = hb_exp(x);
    return e / (T(1) + e);
  }
}
// fp32: branch-free on v_exp_f32 / v_rcp_f32 (1 ulp each) -- the IEEE exp + divide sequences are ~40 instructions
// per element and dominated the epilogue of the bias+sigmoid GEMMs
template <>
__device__ __forceinline__ float hb_sigmoid<float>(float x) {
  // 1 / (1 + e^-x) as it stands: for x < -88 the exponential overflows to +inf and v_rcp_f32 returns 0, which is the
  // value (6e-39 and below) to fp32 precision; everywhere else both factors are good to 1 ulp.  Four vector instructions
  // instead of seven (the |x| form needed a compare, a select and a second multiply): in the epilogues of the fp32 MFMA
  // kernels every vector instruction is paid in full, the matrix instructions share that pipe.
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

#pragma clang fp contract(fast)
#endif  // HB_EW_MATH_CUH
