// Pieces of the column-strip sparse-GP kernels shared between sgp.hip and linalg.hip, and the FORWARD RIDER: the
// contraction A = W K(z, x) computed row block by row block INSIDE the launches of the Cholesky + inverse chain.
//
// The right-looking factorisation is a chain of M/64 dependent launches whose critical path runs on a handful of
// CUs (~20 workgroups per launch at M = 512: ~230 of the 256 CUs idle for ~13 us per launch, 110 us per step),
// and row block k of W = L^-1 is final as soon as launch k has stored its panel.  The rows [64k, 64k+64) of
// A = W K(z, x) need nothing else.  So launch k+1 carries, besides its own workgroups, one extra workgroup per
// 32-column strip of the minibatch that synthesises K(z[0 .. 64(k+1)), x[strip]) into LDS and contracts it with
// W's row block k (two 32-row tiles; four waves: two per tile, splitting the contraction), stores the tiles
// fragment-major for the backward pass and leaves the column statistics of its rows (sum A^2, sum u A) as slice k
// of the partials that sgp_finish_part_kernel folds.  The last row block rides on the finishing (tril) launch.
// The M^2 n forward contraction -- 27 us as a kernel of its own at cfg 2 -- disappears into the chain's shadow.
// The riders take > 80 KB of LDS on purpose: a rider workgroup can then never share a CU with a factor workgroup
// (fp32 MFMAs of one wave stall the VALU work of its SIMD neighbours, which would stretch the critical path).
#ifndef HB_SGP_RIDER_CUH
#define HB_SGP_RIDER_CUH
#include "common.cuh"

// exp(-r2/2) = 2^(-(s*r)^2) with s = sqrt(log2(e)/2): coordinates staged pre-multiplied by s/ell make the
// RBF value a single v_exp_f32 of the negated squared difference.
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


#define SGP_DREG_R 4  // input dimensions a rider handles (= SGP_DREG)

#ifndef HB_RSTAMP
#define HB_RSTAMP(i)
#endif

struct SgpRider {
  const float* x;     // [n][d] minibatch inputs
  const float* z;     // [M][d] inducing inputs
  const float* ell;   // [dl] lengthscales
  const float* u;     // [P][M] (nullable: no statistics)
  float* Af;          // fragment-major A, [M/32][nS][4][64][4]  (NULL: no rider)
  float* part;        // column partials, slice rb at part + rb * 5 * n: [5][n] (q = 0: sum A^2, 1 + p: sum u_p A)
  long n;
  int d, dl, P, nS;   // nS = ceil(n / 32) strips
};

// LDS of a rider workgroup (dynamic shared memory of the carrying kernel; with the 52 KB the factor workgroups of the
// same kernel declare statically this stays under the CU's 160 KB -- and above 80 KB: one workgroup per CU)
#define SGP_RIDER_THREADS 512
#define SGP_RIDER_WAVES (SGP_RIDER_THREADS / 64)
struct SgpRiderLds {
  float Ks[SGP_SN][SGP_SLD];                  // K(z, x[strip]) block, [column][row]
  union {
    float zs[SGP_SM_MAX * SGP_DREG_R];        // scaled inducing inputs: only while the K block is synthesised
    float Tw[SGP_RIDER_WAVES][32][SGP_TLD];   // per-wave staging tile / partial-accumulator exchange / tile transpose
  };
  float us[4][64];
  float cred[2][5][32];
};
static_assert(sizeof(SgpRiderLds) > 80 * 1024 && sizeof(SgpRiderLds) + 53 * 1024 <= 160 * 1024, "rider LDS budget");

// One rider job: rows [64 rb, 64 rb + 64) of A for the 32 columns of `strip`.  W: row-major L^-1 of this matrix
// (rows of block rb final; entries above the diagonal are masked here, whatever the buffer holds).  512 threads:
// eight waves, four per 32-row tile, each taking every fourth 32-deep chunk of the contraction.
__device__ __forceinline__ void sgp_rider_job(const SgpRider& r, const float* __restrict__ W, int M, int rb, int strip,
                                              SgpRiderLds& S) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, h = lane >> 5;
  const int n = (int)r.n, d = r.d, nT = M / 32;
  const int col0 = strip * SGP_SN;
  const int R = 64 * (rb + 1);   // contraction depth: rows of K needed
  HB_RSTAMP(0);
  static_assert(SGP_SM_MAX % 64 == 0, "whole 64-row blocks");
  // ---- wave w: tile 2 rb + (w >> 2), chunks Q = (w & 3), (w & 3) + 4, ... <= tile (at most 4 of them at M = 512).
  // ALL of the wave's W chunks are requested before anything else (the row block was written by the previous launch:
  // cold in this XCD's L2), with COALESCED loads -- lane l takes 16 bytes of row 8v + l/8, so an instruction covers
  // 8 rows x 128 B = 16 cache lines; a lane loading its own MFMA fragment row touches 64 lines per instruction and
  // the CU's address unit, shared by the four waves, then takes ~190 cycles per load: riders of 20 us).  The chunks
  // are turned into fragments through the wave's private LDS tile, one at a time, right before their 16 MFMAs.
  const int tile = 2 * rb + (w >> 2), par = w & 3;
  const float* __restrict__ wsrc = W + (long)(32 * tile + (lane >> 3)) * M + 4 * (lane & 7);
  constexpr int NCH = SGP_SM_MAX / 128;   // chunks per wave
  V4 fr[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int Q = par + 4 * i;
    if (Q <= tile) {   // uniform per wave
#pragma unroll
      for (int v = 0; v < 4; ++v) fr[i][v] = *reinterpret_cast<const V4*>(wsrc + (long)(8 * v) * M + 32 * Q);
    }
  }
  HB_RSTAMP(1);
  // ---- scaled coordinates, u of this row block, K block -> LDS
  {
    const int c = tid & 31, kq = tid >> 5;   // 16 row groups
    const int cc = col0 + c < n ? col0 + c : n - 1;
    float sc[SGP_DREG_R], xs[SGP_DREG_R];
#pragma unroll
    for (int dd = 0; dd < SGP_DREG_R; ++dd) {
      sc[dd] = dd < d ? float(SGP_EXP2_SCALE) / r.ell[r.dl == 1 ? 0 : dd] : 0.f;
      xs[dd] = dd < d ? r.x[(long)cc * d + dd] * sc[dd] : 0.f;
    }
    for (int i = tid; i < R * d; i += SGP_RIDER_THREADS) S.zs[i] = r.z[i] * sc[i % d];
    if (r.u)
      for (int i = tid; i < 64 * r.P; i += SGP_RIDER_THREADS) S.us[i >> 6][i & 63] = r.u[(long)(i >> 6) * M + 64 * rb + (i & 63)];
    __syncthreads();
    if (d == 1) {
      // one input dimension (the common case): four rows per 16-byte LDS read, no inner loop
      const float x0 = xs[0];
#pragma unroll 4
      for (int k4 = kq * 4; k4 < R; k4 += 64) {
        const V4 zz = *reinterpret_cast<const V4*>(&S.zs[k4]);
        V4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float tt = zz[q] - x0;
          v[q] = hb_exp2_neg<float>(tt * tt);
        }
        *reinterpret_cast<V4*>(&S.Ks[c][k4]) = v;
      }
    } else {
      for (int k4 = kq * 4; k4 < R; k4 += 64) {
        V4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float r2 = 0.f;
#pragma unroll
          for (int dd = 0; dd < SGP_DREG_R; ++dd)
            if (dd < d) {
              const float tt = S.zs[(k4 + q) * d + dd] - xs[dd];
              r2 += tt * tt;
            }
          v[q] = hb_exp2_neg<float>(r2);
        }
        *reinterpret_cast<V4*>(&S.Ks[c][k4]) = v;
      }
    }
  }
  __syncthreads();
  HB_RSTAMP(2);
  typename MM::Acc acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  // Chunk i+1 goes through the wave's staging tile (coalesced layout in, fragment layout out) and its K fragments are
  // read BEFORE the 16 MFMAs of chunk i are issued, so the LDS round trip hides under them (LDS operations of one
  // wave execute in order: the tile can be rewritten as soon as the previous chunk's reads have been issued).
  float (*stg)[SGP_TLD] = S.Tw[w];
  auto stage = [&](int i, V4 (&av)[4], V4 (&bv)[4]) {
    const int Q = par + 4 * i;
    if (i >= NCH || Q > tile) return;   // uniform per wave
#pragma unroll
    for (int v = 0; v < 4; ++v) *reinterpret_cast<V4*>(&stg[8 * v + (lane >> 3)][4 * (lane & 7)]) = fr[i < NCH ? i : 0][v];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      av[v] = *reinterpret_cast<const V4*>(&stg[li][16 * h + 4 * v]);
      bv[v] = *reinterpret_cast<const V4*>(&S.Ks[li][32 * Q + 16 * h + 4 * v]);
    }
    if (Q == tile) {
      // the diagonal chunk: entries above the diagonal of W do not belong to the product
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
          if (16 * h + 4 * v + s2 > li) av[v][s2] = 0.f;
    }
  };
  auto mfmas = [&](int i, const V4 (&av)[4], const V4 (&bv)[4]) {
    if (i >= NCH || par + 4 * i > tile) return;   // uniform per wave
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) acc = MM::mma(av[v][s2], bv[v][s2], acc);
  };
  {
    V4 a0[4], b0[4], a1[4], b1[4];
    stage(0, a0, b0);
#pragma unroll
    for (int i = 0; i < NCH; i += 2) {
      stage(i + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(i, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      stage(i + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(i + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  HB_RSTAMP(3);
  // ---- the four quarters of a tile's contraction meet: waves 1..3 of a tile hand their accumulators over through THEIR
  // OWN LDS tiles (wave 0 may still be staging chunks through its tile); fixed order of the additions
  if (par) {
    float* xch = &S.Tw[w][0][0];   // 32 * 36 floats >= 16 * 64
#pragma unroll
    for (int q = 0; q < 16; ++q) xch[q * 64 + lane] = acc[q];
  }
  __syncthreads();
  float cs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (!par) {
    const float *x1 = &S.Tw[w + 1][0][0], *x2 = &S.Tw[w + 2][0][0], *x3 = &S.Tw[w + 3][0][0];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = (acc[q] + x1[q * 64 + lane]) + (x2[q * 64 + lane] + x3[q * 64 + lane]);
  }
  if (!par) {
    sgp_store_frag_tile(r.Af, S.Tw[w], acc, 0, nT, r.nS, tile, strip, col0, n, lane);
    if (r.u) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rl = 32 * (w >> 2) + MM::acc_row(lane, q);   // row inside the 64-row block
        const float v = acc[q];
        cs[0] += v * v;
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p < r.P) cs[1 + p] += S.us[p][rl] * v;
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) cs[q] += __shfl_xor(cs[q], 32);
      if (lane < 32) {
#pragma unroll
        for (int q = 0; q < 5; ++q) S.cred[w >> 2][q][lane] = cs[q];
      }
    }
  }
  __syncthreads();
  HB_RSTAMP(4);
  if (r.u && tid < 32 && col0 + tid < n) {
    float* pp = r.part + (long)rb * 5 * n + col0 + tid;
    for (int q = 0; q < 1 + r.P; ++q) pp[(long)q * n] = S.cred[0][q][tid] + S.cred[1][q][tid];
  }
  HB_RSTAMP(5);
}

#endif  // HB_SGP_RIDER_CUH
