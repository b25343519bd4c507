// Pieces of the column-strip sparse-GP kernels (csrc/sgp.hip) that other translation units share: the strip constants,
// the exp2 form of the RBF value and the fragment-major store of a finished 32 x 32 tile.
#ifndef HB_SGP_STRIP_CUH
#define HB_SGP_STRIP_CUH
#include "common.cuh"

// exp(-r2/2) = 2^(-(s*r)^2) with s = sqrt(log2(e)/2): the coordinate DIFFERENCE times s/ell makes the RBF value a
// single v_exp_f32 of the negated square.  The difference is taken on the raw coordinates and scaled afterwards:
// pre-scaled coordinates (round 1/2) carry |z| s/ell * 2^-24 of rounding each, which at cfg 2 (z up to 256
// lengthscales) is 1.5e-5 in a scaled difference of ~3, i.e. ~1e-4 relative in K -- two orders above the
// fp32 rounding of everything else in the step (tests/test_fp32_parity_gpu.py, profiles/r03_observed_errors.txt).
#define SGP_EXP2_SCALE 0.84932180028801904272
template <typename T> __device__ __forceinline__ T hb_exp2_neg(T x);
template <> __device__ __forceinline__ float hb_exp2_neg<float>(float x) { return __builtin_amdgcn_exp2f(-x); }
template <> __device__ __forceinline__ double hb_exp2_neg<double>(double x) { return exp2(-x); }

#define SGP_SN 32
#define SGP_SM_MAX 512
#define SGP_SLD (SGP_SM_MAX + 4)

// Fragment-major copy of a finished 32 x 32 tile (accumulator layout: column on the lane, rows in the registers) of
// an [M, n] operand of the Lbar contraction: block (row tile t, strip s) holds, for v = 0..3, lane (li, h), s' = 0..3,
//     X[32 t + li][32 s + 16 h + 4 v + s']
// i.e. the MFMA operand fragments of a contraction over the DATA axis, in load order: each of the four stores of a
// wave -- and each of the consumer's loads -- is one contiguous kilobyte.  The tile is turned row-per-lane through the
// wave's own LDS buffer (no barrier: a wave's LDS operations execute in order).  Columns past n are written as zeros.
#define SGP_TLD 36
__device__ __forceinline__ void sgp_store_frag_tile(float* __restrict__ Xf, float (*T)[SGP_TLD],
                                                    const Mma<float>::Acc& acc, long e, int nT, int nS, int tile, int strip,
                                                    int col0, int n, int lane) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  const int li = lane & 31, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) T[Mma<float>::acc_row(lane, r)][li] = acc[r];
  float* blk = Xf + ((((long)e * nT + tile) * nS + strip) << 10) + 4 * lane;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    V4 q = *reinterpret_cast<const V4*>(&T[li][16 * h + 4 * v]);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
      if (col0 + 16 * h + 4 * v + s2 >= n) q[s2] = 0.f;
    *reinterpret_cast<V4*>(blk + 256 * v) = q;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed accumulators (round 3).  With the operands of the tile product swapped -- mma(K fragment, W fragment)
// instead of mma(W fragment, K fragment): the same products summed in the same order, so every element keeps its
// bits -- the 32 x 32 accumulator holds the tile TRANSPOSED: lane (li, h) owns tile ROW li and its 16 registers are
// the columns (r & 3) + 8 (r >> 2) + 4 h.  The row-per-lane view the fragment-major image and the row gradients want
// (lane (li, h): columns 16h .. 16h + 15) is then eight v_permlane32_swap with the half-wave partner -- no LDS
// transpose, no per-wave tile buffer (36.8 KB of LDS per workgroup in the first form, which also kept a second
// workgroup off the CU).
//
// sgp_acc_t_settle: the accumulator was written by MFMAs and is about to be read by inline asm, which the
// compiler's hazard recogniser does not look into: 19 wait states cover a 16-pass MFMA result (the count the compiler
// itself inserts in front of a VALU read).  The accumulator is an operand so that the MFMAs stay in front of it.
__device__ __forceinline__ void sgp_acc_t_settle(Mma<float>::Acc& acc) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));
}
// After the call row16[4 v + s] = X[row li][16 h + 4 v + s] (acc is consumed).
__device__ __forceinline__ void sgp_acc_t_rows(Mma<float>::Acc& acc, float (&row16)[16]) {
  float lo[8], hi[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    lo[r] = acc[r], hi[r] = acc[8 + r];
    // lo's upper half-wave <-> hi's lower half-wave
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo[r]), "+v"(hi[r]));
  }
  // lane (li, h): lo[0..3] -> columns 16h + 0..3, hi[0..3] -> 16h + 4..7, lo[4..7] -> 16h + 8..11, hi[4..7] -> 16h + 12..15
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    row16[s] = lo[s];
    row16[4 + s] = hi[s];
    row16[8 + s] = lo[4 + s];
    row16[12 + s] = hi[4 + s];
  }
}
// fragment-major store of a row-per-lane tile (see sgp_store_frag_tile for the layout); columns past n as zeros
__device__ __forceinline__ void sgp_store_frag_rows(float* __restrict__ Xf, float (&row16)[16], long e, int nT, int nS,
                                                    int tile, int strip, int col0, int n, int lane) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  const int h = lane >> 5;
  float* blk = Xf + ((((long)e * nT + tile) * nS + strip) << 10) + 4 * lane;
  // (only the last strip can hold columns past n: the per-element masks -- 32 vector instructions that compete with the
  // partner wave's fp32 MFMAs for the SIMD -- stay out of every other strip's way)
  if (col0 + 32 > n) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (col0 + 16 * h + i >= n) row16[i] = 0.f;
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const V4 q = {row16[4 * v], row16[4 * v + 1], row16[4 * v + 2], row16[4 * v + 3]};
    *reinterpret_cast<V4*>(blk + 256 * v) = q;
  }
}

#endif  // HB_SGP_STRIP_CUH
