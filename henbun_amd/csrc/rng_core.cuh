// xoroshiro128+ per-lane generator and its state load / store: shared by every kernel file (through common.cuh) AND part
// of the prelude hiprtc compiles in front of run-time generated kernels (csrc/jit.hip) -- keep this file self-contained
// (no #include; only HIP built-ins and the device math library).
#ifndef HB_RNG_CORE_CUH
#define HB_RNG_CORE_CUH
#ifdef __HIPCC_RTC__
typedef unsigned long uint64_t;
typedef unsigned int uint32_t;
#endif

// ---------------------------------------------------------------------------
// xoroshiro128+ per-lane generator.  The state array holds two uint64 per
// lane, structure-of-arrays: s0[nlanes] then s1[nlanes] (coalesced).
// ---------------------------------------------------------------------------
struct HbRng {
  uint64_t s0, s1;
  __device__ __forceinline__ uint64_t next() {
    const uint64_t a = s0;
    uint64_t b = s1;
    const uint64_t r = a + b;
    b ^= a;
    s0 = ((a << 24) | (a >> 40)) ^ b ^ (b << 16);
    s1 = (b << 37) | (b >> 27);
    return r;
  }
  // uniform in (0, 1]
  __device__ __forceinline__ double uniform_pos() {
    return ((double)(next() >> 11) + 1.0) * (1.0 / 9007199254740992.0);
  }
  // uniform in [0, 1)
  __device__ __forceinline__ double uniform() {
    return (double)(next() >> 11) * (1.0 / 9007199254740992.0);
  }
  // A pair of independent standard normals, fp32 runs: Box-Muller on the hardware transcendentals from ONE 64-bit draw
  // (upper 32 bits -> radius, lower 32 bits -> angle).  v_log_f32 is log2 and v_sin_f32 / v_cos_f32 take their argument
  // in REVOLUTIONS -- exactly Box-Muller's 2 pi u -- so a pair costs one generator step, one log, one sqrt, one sin, one
  // cos and four multiplies; the double form below (two draws, fp64 log / sqrt / sincospi: ~150 instructions) made
  // the fp32 samplers RNG-bound at 0.25-0.55 TB/s (profiles/r02_bw_rows.txt).  u1 has 32 bits: |z| <= 6.66.
  // fp32 and fp64 runs therefore draw DIFFERENT variates from the same state (parity tests inject their noise).
  __device__ __forceinline__ void normal2(float& z0, float& z1) {
    const uint64_t b = next();
    const float u1 = ((float)(uint32_t)(b >> 32) + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]  (2^-32 steps, rounded to fp32)
    const float u2 = (float)(uint32_t)b * 2.3283064365386963e-10f;                     // [0, 1]
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), ln = ln2 * log2
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
  }
  // fp64 runs: Box-Muller evaluated in double (two draws per pair).
  __device__ __forceinline__ void normal2(double& z0, double& z1) {
    const double u1 = uniform_pos();
    const double u2 = uniform();
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    z0 = r * cs;
    z1 = r * sn;
  }
};

// Load / store of one lane's state (structure-of-arrays layout: s0[nlanes] then s1[nlanes], coalesced).
__device__ __forceinline__ HbRng rng_load(const uint64_t* state, long nlanes, long t) {
  HbRng g;
  g.s0 = state[t];
  g.s1 = state[nlanes + t];
  return g;
}
__device__ __forceinline__ void rng_store(uint64_t* state, long nlanes, long t, const HbRng& g) {
  state[t] = g.s0;
  state[nlanes + t] = g.s1;
}

#endif  // HB_RNG_CORE_CUH
