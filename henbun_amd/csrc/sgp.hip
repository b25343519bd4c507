// K5/K6: fused sparse-GP conditional  A = L^{-1} K(z,x),  f = u A + sqrt|1 - colsum(A^2)| eps
// and its VJP -- the M^2 n contraction that dominates the ELBO step.
//
// Reference: Henbun/gp/gp.py:99-143 (SparseGP.samples), :146-162
// (_effective_LT = matrix_triangular_solve(Lm, K(z,x))), :177-189
// (_additional_cov 'diagonal'), with K from gp/kernels.py:54-111.
//
// Formulation.  W = L^{-1} is formed once per step (linalg.hip, the reference's
// own batched branch does the same, gp/gp.py:169), so the solve becomes the
// triangular GEMM A = W Kmn.  The RBF block Kmn is synthesised in the B-operand
// loader from x and z (exp on the VALU while the MFMA pipe runs) and is never
// written to memory.  W is lower triangular, so output row-block rb only
// contracts over k < (rb+1)*64; row-blocks rb and nRB-1-rb are paired in one
// workgroup so every workgroup does the same (nRB+1)*64-deep work:
// the kernel executes M^2 n flops, not 2 M^2 n.
//
// Algorithmic work per launch (E experts): flops = E * M^2 * n (fwd A),
// E * M^2 * n (Kbar = W^T Abar), E * M^2 * n (Lbar = -tril(Kbar A^T)).
// The roofline that bounds them is the f32 MFMA peak (v_mfma_f32_32x32x2_f32).
#include "common.cuh"
#include "sgp_strip.cuh"  // strip constants, sgp_store_frag_tile
#include "chain.cuh"      // serial chains: the finishing pass may be recorded instead of launched
#include "chol_persist.cuh"  // the early-start form runs inside the persistent factorisation's launch
#include <type_traits>
#include "gemm_tile.cuh"
#include "rng_pairs.cuh"
#include "../../include/henbun_hip.h"

#define SGP_BM 64
#define SGP_BN 128
#define SGP_DREG 4  // input dims held in registers by the operand loaders

// provided by linalg.hip / elementwise.hip (same shared object)
extern "C" int hb_matmul_f32(const float*, const float*, float*, long, long, long, long, long, long, long, long, long,
                             long, int, int, double, double, const float*, long, int, int, float*, long, void*);
extern "C" int hb_matmul_f64(const double*, const double*, double*, long, long, long, long, long, long, long, long,
                             long, long, int, int, double, double, const double*, long, int, int, double*, long,
                             void*);

template <typename T>
struct SgpArgs {
  const T* x;   // [E?, n, d]
  long sx;      // expert stride of x (0 = shared)
  const T* z;   // [E, M, d]
  const T* ell; // [E, dl]
  long dl;
  const T* W;   // [E, M, M]
  const T* Wf;  // fragment-major copy of W written by hb_cholesky_inverse (see tril_inplace_kernel), or nullptr
  const void* W3;  // bf16x3 fragment images of W (3 planes of E*M*M bf16), or nullptr
  long plane3;     // elements per bf16 plane (E*M*M)
  T* Af;           // fragment-major A: [E][M/32 row tiles][nS strips][4 v][64 lanes][4] (see hb_sgp_fwd), or nullptr
  const T* u;   // [E, P, M]
  T* A;         // [E, M, n]
  long n, M, d, P;
  T* part;      // [E, gridDim.y, 5, n] column partial sums (sum A^2, sum_m u_pm A_mj for p < 4) or nullptr
  long efast;   // > 0: the grid is (columns * E, rows, 1) with the expert index fastest (see sgp_block)
  // finishing pass inside the third strip form (fin != 0; see sgp_A_strip2t_kernel): v = 1 - sum A^2, f = u A + sqrt|v| eps
  int fin = 0, fin_diag = 0;
  const T* fin_eps_in = nullptr;
  uint64_t* fin_rng = nullptr;
  long fin_lanes = 0;
  T* fin_eps_out = nullptr;
  T* fin_f = nullptr;
  T* fin_v = nullptr;
  // ... and the per-point part of the Gaussian likelihood head behind it (head != 0; P == 1; hb_sgp_fwd_gauss):
  // dmu_j = (y_j - f_j s) / var, fbar_j = s (post dmu_j), and this strip's partial sums of (ll, dscale, dvar)
  int head = 0;
  const T* hy = nullptr;
  const T* hscale = nullptr;
  const T* hvar = nullptr;
  T* hdmu = nullptr;
  T* hfbar = nullptr;
  T hpost = T(0);
  T* hpart = nullptr;      // [3][hunits], unit = e * nS + strip
  long hunits = 0;
};

// Expert <-> XCD affinity.  Workgroups are dealt to the 8 XCDs round-robin by linear id, i.e. by blockIdx.x.  With
// the expert in blockIdx.z every XCD sees every expert and their E triangular factors (1 MB each at M = 512) compete
// for one 4 MB L2; when E is a multiple of 8 the expert is made the fastest grid coordinate instead, so an XCD only
// ever works on "its" experts.
__device__ __forceinline__ void sgp_block(long efast, long& e, int& bx) {
  if (efast > 0) {
    e = blockIdx.x % efast;
    bx = (int)(blockIdx.x / efast);
  } else {
    e = blockIdx.z;
    bx = (int)blockIdx.x;
  }
}
static inline dim3 sgp_grid(long gx, long gy, long E, long& efast) {
  efast = (E > 1 && E % 8 == 0) ? E : 0;
  return efast ? dim3((unsigned)(gx * E), (unsigned)gy, 1) : dim3((unsigned)gx, (unsigned)gy, (unsigned)E);
}

// ---------------------------------------------------------------------------
// forward: A = W K(z,x)
//
// D = compile-time input dimension (1..SGP_DREG) with z/ell staged in LDS and
// x/ell in registers: the B-operand loader is then branch-free LDS reads + one
// v_exp per element.  D = 0 is the generic (any d, any M) fallback reading
// global memory.
// ---------------------------------------------------------------------------
#define SGP_ZS_MAX 4096  // scaled inducing points kept in LDS (elements)

template <typename T> __device__ __forceinline__ T hb_exp_fast(T x);
template <> __device__ __forceinline__ float hb_exp_fast<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double hb_exp_fast<double>(double x) { return exp(x); }
template <typename T, int D>
struct SgpZ {
  T z[D];
};

// Row-block schedule of the triangular contractions.  gridDim.y == ceil(nRB/2): a workgroup takes the pair
// (y, nRB-1-y) in turn, so every workgroup does the same work.  gridDim.y == nRB ("unpaired", chosen by the host
// when the paired grid would leave one workgroup per CU): one row block per workgroup -- two workgroups are then
// co-resident per CU and each fills the other's non-MFMA issue slots (tools/kloop_cycles.hip: 61 -> 72 % MFMA
// utilisation) -- ordered so that workgroups y and y + nRB/2, which an in-order dispatch places together, are a
// shallow and a deep block.
__device__ __forceinline__ bool sgp_row_block(int half, int nRB, int& rb) {
  const int y = (int)blockIdx.y;
  if ((int)gridDim.y == nRB && nRB > 1) {
    const int h = nRB / 2;
    rb = y < h ? y : nRB - 1 - (y - h);
    return half == 0;
  }
  rb = half == 0 ? y : nRB - 1 - y;
  return !(half == 1 && rb <= y);  // odd count: the middle block is handled once
}

template <typename T, int D, bool FAST>
__global__ void __launch_bounds__(256) sgp_A_kernel(SgpArgs<T> a) {
  typedef TileGemm<T, SGP_BM, SGP_BN, 16, 2, 2> G;
  __shared__ __attribute__((aligned(16))) T lds[G::LDS_ELEMS + (D > 0 ? SGP_ZS_MAX : 1)];
  T* zs = lds + G::LDS_ELEMS;
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const T* x = a.x + e * a.sx;
  const T* z = a.z + e * a.M * a.d;
  const T* ell = a.ell + e * a.dl;
  const T* W = a.W + e * a.M * a.M;
  T* A = a.A + e * a.M * a.n;
  const int M = (int)a.M, n = (int)a.n, d = (int)a.d;
  const int col0 = bx * SGP_BN;
  const int nRB = (M + SGP_BM - 1) / SGP_BM;
  const int Mm1 = M - 1;

  // this thread's operand column is fixed across k-steps (256 % SGP_BN == 0)
  const int jcol = col0 + (threadIdx.x % SGP_BN);
  const bool jok = jcol < n;
  const int jc = jok ? jcol : n - 1;
  // coordinates are staged RAW and the difference is scaled (see hb_exp2_neg in sgp_strip.cuh)
  T xs[D > 0 ? D : 1], scl[D > 0 ? D : 1];
  if (D > 0) {
#pragma unroll
    for (int q = 0; q < D; ++q) {
      xs[q] = x[jc * D + q];
      scl[q] = T(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : q];
    }
    for (int t = threadIdx.x; t < M * D; t += blockDim.x) zs[t] = z[t];
    __syncthreads();
  }

  typedef typename TileGemm<T, SGP_BM, SGP_BN, 16, 2, 2>::VT VT;
  constexpr int VEC = TileGemm<T, SGP_BM, SGP_BN, 16, 2, 2>::VEC;
  constexpr int DD = D > 0 ? D : 1;
  // vector path: the RBF block is synthesised "k-contiguous" -- a group is VEC consecutive inducing points
  // for ONE data column, so it lands in LDS with one 16-byte write and is read back as 16-byte fragments.
  // A thread's groups always sit on the same GB columns: column of group g = (g*256 + tid) / (BK/VEC).
  constexpr int GBc = G::GB, GPR = 16 / VEC;  // groups per thread, groups per tile row
  T xg[GBc][DD];
  if (FAST) {
#pragma unroll
    for (int gq = 0; gq < GBc; ++gq) {
      const int cl = (gq * 256 + (int)threadIdx.x) / GPR;
      const int cc = col0 + cl < n ? col0 + cl : n - 1;  // out-of-range columns compute garbage that is never stored
#pragma unroll
      for (int dd = 0; dd < DD; ++dd) xg[gq][dd] = x[cc * DD + dd];
    }
  }

  // fused column statistics (the `finish` pass never re-reads A): per accumulator column of this lane,
  // [0] = sum_m A_mj^2, [1+p] = sum_m u_pm A_mj, carried across the two row blocks of the workgroup
  typedef Mma<T> MMc;
  constexpr int RMc = G::RM, RNc = G::RN;
  T cs[RNc][5];
#pragma unroll
  for (int jj = 0; jj < RNc; ++jj)
#pragma unroll
    for (int q = 0; q < 5; ++q) cs[jj][q] = T(0);
  const int npart = a.part ? (int)a.P : 0;

  for (int half = 0; half < 2; ++half) {
    int rb;
    if (!sgp_row_block(half, nRB, rb)) break;
    const int row0 = rb * SGP_BM;
    int kend = row0 + SGP_BM;
    if (kend > M) kend = M;
    G g;
    g.zero();
    auto la = [&](int m, int k) -> T {
      const int r = row0 + m;
      return W[(r < M ? r : Mm1) * M + k];
    };
    auto fa = [&](T raw, int m, int k) -> T {
      const int r = row0 + m;
      return ((r < M) & (k <= r)) ? raw : T(0);
    };
    auto lb = [&](int k, int nn) -> T { return T(0); };
    auto fb = [&](T raw, int k, int nn) -> T {
      T r2 = T(0);
      if (D > 0) {
#pragma unroll
        for (int q = 0; q < D; ++q) {
          const T t = (zs[k * D + q] - xs[q]) * scl[q];
          r2 += t * t;
        }
      } else {
        for (int q = 0; q < d; ++q) {
          const T t = (z[k * d + q] - x[jc * d + q]) / ell[a.dl == 1 ? 0 : q];
          r2 += t * t;
        }
      }
      const T val = D > 0 ? hb_exp2_neg<T>(r2) : hb_exp_fast<T>(T(-0.5) * r2);
      return jok ? val : T(0);
    };
    if constexpr (FAST) {
      // vector path (D >= 1, M % 16 == 0): W rows by 16-byte loads with the triangular mask per
      // component; the RBF block is synthesised 4 columns at a time (this thread's 4 columns are fixed)
      auto la4 = [&](int m, int k) -> VT {
        const int r = row0 + m;
        return *reinterpret_cast<const VT*>(&W[(r < M ? r : Mm1) * M + k]);
      };
      auto fa4 = [&](VT raw, int m, int k) -> VT {
        const int r = row0 + m;
        VT v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = ((r < M) & (k + q <= r)) ? raw[q] : T(0);
        return v;
      };
      // the scaled inducing coordinates of rows k..k+VEC-1 are the B operand's "raw" value: read from LDS
      // one iteration ahead, like a global load, so the latency is off the critical path
      auto lb4 = [&](int k, int nn) -> SgpZ<T, VEC * DD> {
        SgpZ<T, VEC * DD> r;
        if (DD == 1) {
          const VT v = *reinterpret_cast<const VT*>(&zs[k]);
#pragma unroll
          for (int q = 0; q < VEC; ++q) r.z[q] = v[q];
        } else {
#pragma unroll
          for (int q = 0; q < VEC * DD; ++q) r.z[q] = zs[k * DD + q];
        }
        return r;
      };
      auto fb4 = [&](SgpZ<T, VEC * DD> raw, int k, int nn) -> VT {
        // which of this thread's columns: groups are laid out GPR per column, 256/GPR columns per group index
        T xc[DD];
#pragma unroll
        for (int dd = 0; dd < DD; ++dd) {
          xc[dd] = xg[0][dd];
#pragma unroll
          for (int gq = 1; gq < GBc; ++gq) xc[dd] = nn >= gq * (256 / GPR) ? xg[gq][dd] : xc[dd];
        }
        VT v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          T r2 = T(0);
#pragma unroll
          for (int dd = 0; dd < DD; ++dd) {
            const T t = (raw.z[q * DD + dd] - xc[dd]) * scl[dd];
            r2 += t * t;
          }
          v[q] = hb_exp2_neg<T>(r2);
        }
        return v;
      };
      g.template run_vec<HB_KC, HB_KC>(0, kend, la4, fa4, lb4, fb4, lds);
    } else {
      g.template run<true, false>(0, kend, la, fa, lb, fb, lds);
    }
    g.for_each([&](int row, int col, T v) {
      const int r = row0 + row, c = col0 + col;
      if (r < M && c < n) A[(long)r * n + c] = v;
    });
    if (a.part) {
      const int lane = threadIdx.x & 63, wm = (threadIdx.x >> 6) / G::WN;
#pragma unroll
      for (int i = 0; i < RMc; ++i)
#pragma unroll
        for (int r = 0; r < MMc::NACC; ++r) {
          const int row = row0 + wm * G::WTM + i * MMc::TM + MMc::acc_row(lane, r);
          const int rc = row < M ? row : Mm1;  // rows past M hold zeros
          T up[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) up[p] = p < npart ? a.u[e * a.P * a.M + (long)p * M + rc] : T(0);
#pragma unroll
          for (int jj = 0; jj < RNc; ++jj) {
            const T v = g.acc[i][jj][r];
            cs[jj][0] += v * v;
#pragma unroll
            for (int p = 0; p < 4; ++p) cs[jj][1 + p] += up[p] * v;
          }
        }
    }
  }
  if (a.part) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = w / G::WN, wn = w % G::WN;
    // lanes that share an accumulator column differ in the bits above log2(TN)
#pragma unroll
    for (int jj = 0; jj < RNc; ++jj)
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        T s = cs[jj][q];
#pragma unroll
        for (int off = MMc::TN; off < 64; off <<= 1) s += __shfl_xor(s, off);
        cs[jj][q] = s;
      }
    __syncthreads();  // every wave is done with the operand buffers
    T* red = lds;     // [2 wave rows][5][SGP_BN]
    if (lane < MMc::TN) {
#pragma unroll
      for (int jj = 0; jj < RNc; ++jj)
#pragma unroll
        for (int q = 0; q < 5; ++q) red[(wm * 5 + q) * SGP_BN + wn * G::WTN + jj * MMc::TN + lane] = cs[jj][q];
    }
    __syncthreads();
    if (threadIdx.x < SGP_BN) {
      const int c = col0 + threadIdx.x;
      if (c < n) {
        T* pp = a.part + (e * gridDim.y + blockIdx.y) * 5 * a.n + c;
        for (int q = 0; q < 1 + npart; ++q) pp[(long)q * n] = red[q * SGP_BN + threadIdx.x] + red[(5 + q) * SGP_BN + threadIdx.x];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// forward, column-strip form (fp32, M <= 512, M % 32 == 0, d <= SGP_DREG)
//
// One workgroup owns a strip of 32 data columns and ALL M rows.  The strip's RBF block
// K(z, x[strip]) -- M x 32 -- is synthesised ONCE into LDS (one exp per element, where the tiled
// kernel above re-synthesises it per row block: 4.5x the exps at M = 512), and then every wave
// runs its own row tiles through the contraction with NO barriers and no LDS staging of W:
//   * B fragments: two 16-byte LDS reads per 16-deep step, shared by all the wave's row tiles;
//   * A fragments: W rows straight from global/L2 into registers (the contraction index is
//     permuted so a lane reads 32 contiguous bytes of its row), prefetched one step ahead;
//   * wave w takes the 32-row tiles {w, 7-w, 8+w, 15-w}: every wave contracts the same depth.
// That is ~0.5 non-MFMA instructions per MFMA against ~3 in the tiled kernel, which matters
// because on gfx950 the fp32 MFMA shares the wave's issue slot with everything else.
// The epilogue leaves the same column statistics (one partial row: gy = 1).
// ---------------------------------------------------------------------------
// In-kernel phase stamps: compiled in only by tools/strip_stamps.hip (diagnostic build).
#ifndef HB_SSTAMP
#define HB_SSTAMP(i)
#endif
// Ablation switches of the same diagnostic build (0 in the product): 1 = no MFMAs, 2 = no W loads inside the loop
#ifndef HB_STRIP_ABLATE
#define HB_STRIP_ABLATE 0
#endif

#define SGP_STRIP_THREADS 512  // 8 waves: two per SIMD, so one wave's W loads are in flight under the other's MFMAs

template <int D, bool FRAG>
__global__ void __launch_bounds__(SGP_STRIP_THREADS) sgp_A_strip_kernel(SgpArgs<float> a) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  __shared__ __attribute__((aligned(16))) float Ks[SGP_SN][SGP_SLD];
  __shared__ __attribute__((aligned(16))) float zs[SGP_SM_MAX * D];
  __shared__ float us[4][SGP_SM_MAX];  // u rows for the column statistics (staged once: the epilogue reads them per accumulator row)
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const float* __restrict__ x = a.x + e * a.sx;
  const float* __restrict__ z = a.z + e * a.M * D;
  const float* __restrict__ ell = a.ell + e * a.dl;
  const float* __restrict__ W = (FRAG ? a.Wf : a.W) + e * a.M * a.M;
  float* __restrict__ A = a.A ? a.A + e * a.M * a.n : nullptr;
  const int M = (int)a.M, n = (int)a.n;
  const int col0 = bx * SGP_SN;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;

  HB_SSTAMP(0);
  // ---- K(z, x[strip]) -> LDS, [column][k]: thread (c = tid % 32, kq = tid / 32) takes the 16-byte groups kq, kq+16, ...
  // z is staged (pre-scaled) in LDS first: read straight from global, every group would be a dependent
  // load round trip (measured: the prologue alone cost ~10 us).
  {
    const int c = tid & 31, kq = tid >> 5;
    const int cc = col0 + c < n ? col0 + c : n - 1;  // columns past n compute garbage that is never stored
    float sc[D], xs[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      sc[dd] = float(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : dd];
      xs[dd] = x[cc * D + dd];   // raw: the difference is taken first, then scaled (see hb_exp2_neg in sgp_strip.cuh)
    }
    // all staging loads are issued before any is consumed (a load -> LDS store loop would pay one dependent
    // round trip per iteration: five of them, ~5 us, in the first version of this prologue)
    constexpr int NTH = SGP_STRIP_THREADS;
    constexpr int ZIT = (SGP_SM_MAX * D) / NTH, UIT = SGP_SM_MAX / NTH;
    static_assert(ZIT >= 1 && UIT >= 1, "staging loops");
    float zt[ZIT], ut[4][UIT];
    const int npu = a.part ? ((int)a.P < 4 ? (int)a.P : 4) : 0;
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      zt[it] = z[i < M * D ? i : 0];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        ut[p][it] = p < npu ? a.u[e * a.P * a.M + (long)p * M + (i < M ? i : 0)] : 0.f;
      }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      if (i < M * D) zs[i] = zt[it];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        if (p < npu && i < M) us[p][i] = ut[p][it];
      }
    __syncthreads();
#pragma unroll 4
    for (int k4 = kq * 4; k4 < M; k4 += NTH / 8) {
      float zq[4 * D];
#pragma unroll
      for (int q = 0; q < 4 * D; q += 4) {
        const V4 zz = *reinterpret_cast<const V4*>(&zs[k4 * D + q]);
#pragma unroll
        for (int s = 0; s < 4; ++s) zq[q + s] = zz[s];
      }
      V4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float r2 = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const float tt = (zq[q * D + dd] - xs[dd]) * sc[dd];
          r2 += tt * tt;
        }
        v[q] = hb_exp2_neg<float>(r2);
      }
      *reinterpret_cast<V4*>(&Ks[c][k4]) = v;
    }
  }
  __syncthreads();

  HB_SSTAMP(1);
  // Wave w takes the 32-row tiles (w, nT-1-w): one shallow and one deep, the same total depth (nT + 1 chunks of 32)
  // for every wave.  A single tile (the middle one of an odd count) sits in the LAST slot, so that "slots >= P" is
  // always a set of real tiles in order of increasing depth; waves beyond ceil(nT/2) have nothing to do.
  const int nT = M / 32;
  int tile[2];
  bool tv[2];
  {
    const int deep = nT - 1 - w;
    tv[1] = w <= deep;
    tv[0] = w < deep;
    tile[1] = tv[1] ? deep : 0;  // empty slots point at a valid row block (prefetch stays in bounds)
    tile[0] = tv[0] ? w : 0;
  }
  typename MM::Acc acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // The wave's tiles are in order of increasing depth (tile j is active while Q < dep[j] = tile_j + 1 steps of 32),
  // so the loop splits into two phases with tiles j >= P active: no per-tile branch inside a phase (the
  // accumulators stay in AGPRs), and only tile P's last step crosses the diagonal and needs the k <= row mask.
  // A step is 32 deep: lane (row, h) reads the 64 contiguous bytes k = 32Q + 16h .. +15 of its W row, so the
  // two lanes of a row consume one whole 128-byte line per step (16-deep steps fetched every line twice: the
  // 64 KB of lines a workgroup touches per step do not survive in L1 until the next one).
  int dep[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) dep[j] = tv[j] ? tile[j] + 1 : 0;
  V4 an[2][4];
  auto load_a = [&](int Q, auto pc) {
    constexpr int P = decltype(pc)::value;
#pragma unroll
    for (int j = P; j < 2; ++j) {
      int qq = Q < dep[j] ? Q : dep[j] - 1;  // finished tile: re-read its last step (never used)
      qq = qq < 0 ? 0 : qq;                   // empty slot (depth 0): stay inside W
      if (FRAG) {
        // fragment-major image: load instruction v of the wave reads one contiguous kilobyte
        const float* p = W + ((long)(tile[j] * nT + qq) << 10) + 4 * lane;
#pragma unroll
        for (int v = 0; v < 4; ++v) an[j][v] = *reinterpret_cast<const V4*>(p + 256 * v);
      } else {
        const float* p = W + (32 * tile[j] + li) * M + 32 * qq + 16 * h;
#pragma unroll
        for (int v = 0; v < 4; ++v) an[j][v] = *reinterpret_cast<const V4*>(p + 4 * v);
      }
    }
  };
  auto step = [&](int Q, auto pc, auto mc) {
    constexpr int P = decltype(pc)::value;
    constexpr bool MASK = decltype(mc)::value;
    V4 ac[2][4];
    // The copy out of the prefetch registers is where the compiler waits for the loads issued one step ago.  Left to
    // itself the scheduler rotates it into the PREVIOUS step, in between that step's MFMAs (as soon as the operand
    // registers die), i.e. a few hundred cycles after the loads were issued: the wave then sits on s_waitcnt vmcnt
    // with its matrix pipe idle (48 % issue-wait + 30 % parked in profiles/r01_pmc_cfg2_kernels.txt).  Pinned here,
    // the loads have the whole previous step (32 MFMAs, >= 2048 cycles) to land.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = P; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) ac[j][v] = an[j][v];
    __builtin_amdgcn_sched_barrier(0);
    if (!(HB_STRIP_ABLATE & 2) && Q + 1 < dep[1]) load_a(Q + 1, pc);  // prefetch (a tile that finishes now re-reads its last step: harmless)
    V4 bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = *reinterpret_cast<const V4*>(&Ks[li][32 * Q + 16 * h + 4 * v]);
    if (MASK && !FRAG) {  // (the fragment image holds explicit zeros above the diagonal)
      // W is lower triangular: entries k > row are not part of it (only tile P crosses the diagonal here)
      const int r = 32 * tile[P] + li, kb = 32 * Q + 16 * h;
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int s = 0; s < 4; ++s) ac[P][v][s] = (kb + 4 * v + s <= r) ? ac[P][v][s] : 0.f;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = P; j < 2; ++j) {
          if (HB_STRIP_ABLATE & 1)
            asm volatile("" ::"v"(ac[j][v][s]), "v"(bv[v][s]));
          else
            acc[j] = MM::mma(ac[j][v][s], bv[v][s], acc[j]);
        }
  };
  auto phase = [&](int q0, auto pc) {
    constexpr int P = decltype(pc)::value;
    const int q1 = dep[P];
    int Q = q0;
    for (; Q < q1 - 1; ++Q) step(Q, pc, std::false_type());
    for (; Q < q1; ++Q) step(Q, pc, std::true_type());
    return q1 > q0 ? q1 : q0;
  };
  // A tile is stored (and folded into the column statistics) the moment its phase ends: the stores of three of the
  // four tiles then drain under the MFMAs of the later phases instead of forming a serial epilogue.
  const int npart = a.part ? (int)a.P : 0;
  float cs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int gc = col0 + li;
  auto retire = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    if (!tv[P]) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * tile[P] + MM::acc_row(lane, r);
      const float v = acc[P][r];
      if (gc < n) A[(long)row * n + gc] = v;
      if (a.part) {
        cs[0] += v * v;
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p < npart) cs[1 + p] += us[p][row] * v;
      }
    }
  };
  if (dep[1] > 0) {
    load_a(0, std::integral_constant<int, 0>());
    int Q = 0;
    Q = phase(Q, std::integral_constant<int, 0>());
    retire(std::integral_constant<int, 0>());
    HB_SSTAMP(2);
    Q = phase(Q, std::integral_constant<int, 1>());
    retire(std::integral_constant<int, 1>());
    HB_SSTAMP(3);
  }
  HB_SSTAMP(5);

  // ---- epilogue: column statistics across lanes and waves
  if (a.part) {
#pragma unroll
    for (int q = 0; q < 5; ++q) cs[q] += __shfl_xor(cs[q], 32);
    __syncthreads();  // every wave is done reading the K block
    float* red = &Ks[0][0];  // [8 waves][5][32]
    if (lane < 32) {
#pragma unroll
      for (int q = 0; q < 5; ++q) red[(w * 5 + q) * 32 + lane] = cs[q];
    }
    __syncthreads();
    if (tid < 32 && col0 + tid < n) {
      float* pp = a.part + e * 5 * a.n + col0 + tid;  // gy = 1
      for (int q = 0; q < 1 + npart; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < SGP_STRIP_THREADS / 64; ++ww) sum += red[(ww * 5 + q) * 32 + tid];  // fixed order
        pp[(long)q * n] = sum;
      }
    }
  }
  HB_SSTAMP(6);
}

// bf16x3 form of the fragment-major exchange (HB_PREC_BF16X3): three planes (hi, mid, lo terms) of bf16, each
// [E][M/32][nS][2 q][64 lanes][8] with the element X[32t + li][32s + 16h + 8q + j] -- the operand fragment of
// v_mfma_f32_32x32x16_bf16 for the k16-step q of a strip (the order of the 32 columns inside a strip is the same for
// both operands of the Lbar contraction, which is all that matters).  `row16` = this lane's 16 columns (row li,
// columns 16h .. 16h+15) in registers.
__device__ __forceinline__ void sgp_store_frag3_row(__bf16* __restrict__ X3, long plane, const float (&row16)[16], long e,
                                                    int nT, int nS, int tile, int strip, int lane) {
  typedef __bf16 B8 __attribute__((ext_vector_type(8)));
  __bf16* blk = X3 + ((((long)e * nT + tile) * nS + strip) << 10) + 8 * lane;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    B8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = row16[8 * q + j];
      const __bf16 b0 = (__bf16)x;
      const float r1 = x - (float)b0;
      const __bf16 b1 = (__bf16)r1;
      p0[j] = b0, p1[j] = b1, p2[j] = (__bf16)(r1 - (float)b1);
    }
    *reinterpret_cast<B8*>(blk + 512 * q) = p0;
    *reinterpret_cast<B8*>(blk + plane + 512 * q) = p1;
    *reinterpret_cast<B8*>(blk + 2 * plane + 512 * q) = p2;
  }
}
// accumulator-layout tile -> row-per-lane registers through the wave's LDS buffer (columns past n zeroed)
__device__ __forceinline__ void sgp_tile_rows(float (*T)[SGP_TLD], const Mma<float>::Acc& acc, float (&row16)[16], int col0,
                                              int n, int lane) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  const int li = lane & 31, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) T[Mma<float>::acc_row(lane, r)][li] = acc[r];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const V4 q = *reinterpret_cast<const V4*>(&T[li][16 * h + 4 * v]);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) row16[4 * v + s2] = (col0 + 16 * h + 4 * v + s2 < n) ? q[s2] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Column-strip contraction, second form: used when the fragment-major image of W is available (hb_cholesky_inverse's
// Wfrag).  Same decomposition (one workgroup = 32 data columns x all M rows, 8 waves, wave w owns the row tiles
// w and nT-1-w), but the wave walks its nT+1 "tile-steps" (16 MFMAs each: one 32-row tile x one 32-deep chunk of the
// contraction) as ONE flat sequence -- first all of the shallow tile, then all of the deep one -- with a single
// accumulator and two operand register sets used alternately (the loop is unrolled by two):
//     load B-set (step t+1) ; 16 MFMAs from A-set (step t) ; load A-set (step t+2) ; 16 MFMAs from B-set (step t+1)
// Every load has a whole step of MFMAs (>= 1024 cycles) between issue and first use, and no register copy carries
// an operand set around the loop.  In the first form the copy out of the prefetch registers was rotated by the
// compiler into the middle of the previous step, so the wave waited (s_waitcnt vmcnt) ~450 cycles after issuing the
// loads with its matrix pipe idle: loads and MFMAs did not overlap (profiles/r01_strip_ablation.txt: 16.3 + 12.4 us
// of work took 21.7 us).
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(SGP_STRIP_THREADS) sgp_A_strip2_kernel(SgpArgs<float> a) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  __shared__ __attribute__((aligned(16))) float Ks[SGP_SN][SGP_SLD];
  __shared__ __attribute__((aligned(16))) float zs[SGP_SM_MAX * D];
  __shared__ float us[4][SGP_SM_MAX];
  __shared__ __attribute__((aligned(16))) float Tw[SGP_STRIP_THREADS / 64][32][SGP_TLD];  // per-wave tile transpose (Af)
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const float* __restrict__ x = a.x + e * a.sx;
  const float* __restrict__ z = a.z + e * a.M * D;
  const float* __restrict__ ell = a.ell + e * a.dl;
  const float* __restrict__ Wf = a.Wf + e * a.M * a.M;
  float* __restrict__ A = a.A ? a.A + e * a.M * a.n : nullptr;
  const int M = (int)a.M, n = (int)a.n;
  const int col0 = bx * SGP_SN;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;

  HB_SSTAMP(0);
  // ---- K(z, x[strip]) -> LDS (as in the first form)
  {
    const int c = tid & 31, kq = tid >> 5;
    const int cc = col0 + c < n ? col0 + c : n - 1;
    float sc[D], xs[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      sc[dd] = float(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : dd];
      xs[dd] = x[cc * D + dd];   // raw: the difference is taken first, then scaled (see hb_exp2_neg in sgp_strip.cuh)
    }
    constexpr int NTH = SGP_STRIP_THREADS;
    constexpr int ZIT = (SGP_SM_MAX * D) / NTH, UIT = SGP_SM_MAX / NTH;
    float zt[ZIT], ut[4][UIT];
    const int npu = a.part ? ((int)a.P < 4 ? (int)a.P : 4) : 0;
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      zt[it] = z[i < M * D ? i : 0];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        ut[p][it] = p < npu ? a.u[e * a.P * a.M + (long)p * M + (i < M ? i : 0)] : 0.f;
      }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      if (i < M * D) zs[i] = zt[it];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        if (p < npu && i < M) us[p][i] = ut[p][it];
      }
    __syncthreads();
#pragma unroll 4
    for (int k4 = kq * 4; k4 < M; k4 += NTH / 8) {
      float zq[4 * D];
#pragma unroll
      for (int q = 0; q < 4 * D; q += 4) {
        const V4 zz = *reinterpret_cast<const V4*>(&zs[k4 * D + q]);
#pragma unroll
        for (int s = 0; s < 4; ++s) zq[q + s] = zz[s];
      }
      V4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float r2 = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const float tt = (zq[q * D + dd] - xs[dd]) * sc[dd];
          r2 += tt * tt;
        }
        v[q] = hb_exp2_neg<float>(r2);
      }
      *reinterpret_cast<V4*>(&Ks[c][k4]) = v;
    }
  }
  __syncthreads();
  HB_SSTAMP(1);

  // ---- this wave's tile-steps: [tile t0 = w: chunks 0..w] then [tile t1 = nT-1-w: chunks 0..t1]
  const int nT = M / 32;
  const int t1 = nT - 1 - w, t0 = w;
  const int d0 = w < t1 ? w + 1 : 0;      // the middle tile of an odd count is taken once, as t1
  const int d1 = w <= t1 ? t1 + 1 : 0;
  const int nts = d0 + d1;
  const int npart = a.part ? (int)a.P : 0;
  float cs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int gc = col0 + li;
  typename MM::Acc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  auto load = [&](V4 (&f)[4], int ts) {
    const int tc = ts < nts ? ts : nts - 1;            // past the end: re-read the last step (never used)
    const int tile = tc < d0 ? t0 : t1, Q = tc < d0 ? tc : tc - d0;
    const float* p = Wf + ((long)(tile * nT + Q) << 10) + 4 * lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) f[v] = *reinterpret_cast<const V4*>(p + 256 * v);
  };
  auto compute = [&](const V4 (&f)[4], int ts) {
    if (ts >= nts) return;                               // (uniform) the odd tail of the two-step loop
    const int tile = ts < d0 ? t0 : t1, Q = ts < d0 ? ts : ts - d0;
    V4 bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = *reinterpret_cast<const V4*>(&Ks[li][32 * Q + 16 * h + 4 * v]);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = MM::mma(f[v][s], bv[v][s], acc);
    if (ts == d0 - 1 || ts == nts - 1) {
      // the tile is complete (its last chunk holds the diagonal block; the image has explicit zeros above it):
      // store it and fold it into the column statistics; the stores drain under the next tile's MFMAs
      if (a.Af) sgp_store_frag_tile(a.Af, Tw[w], acc, e, nT, (n + SGP_SN - 1) / SGP_SN, tile, bx, col0, n, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * tile + MM::acc_row(lane, r);
        const float v = acc[r];
        if (a.A && gc < n) A[(long)row * n + gc] = v;
        if (a.part) {
          cs[0] += v * v;
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (p < npart) cs[1 + p] += us[p][row] * v;
        }
        acc[r] = 0.f;
      }
    }
  };
  if (nts > 0) {
    V4 fa[4], fb[4];
    load(fa, 0);
    // sched_barrier(0): nothing crosses.  Without them the scheduler hoists a set's reload above the last MFMAs
    // that still read it (renaming the registers), which turns the loop-carried set into a copy at the back edge
    // and moves the vmcnt wait to a few hundred cycles after the load.
#pragma nounroll
    for (int ts = 0; ts < nts; ts += 2) {
      load(fb, ts + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(fa, ts);
      __builtin_amdgcn_sched_barrier(0);
      load(fa, ts + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(fb, ts + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  HB_SSTAMP(2);

  // ---- epilogue: column statistics across lanes and waves
  if (a.part) {
#pragma unroll
    for (int q = 0; q < 5; ++q) cs[q] += __shfl_xor(cs[q], 32);
    __syncthreads();  // every wave is done reading the K block
    float* red = &Ks[0][0];  // [8 waves][5][32]
    if (lane < 32) {
#pragma unroll
      for (int q = 0; q < 5; ++q) red[(w * 5 + q) * 32 + lane] = cs[q];
    }
    __syncthreads();
    if (tid < 32 && col0 + tid < n) {
      float* pp = a.part + e * 5 * a.n + col0 + tid;  // gy = 1
      for (int q = 0; q < 1 + npart; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < SGP_STRIP_THREADS / 64; ++ww) sum += red[(ww * 5 + q) * 32 + tid];  // fixed order
        pp[(long)q * n] = sum;
      }
    }
  }
  HB_SSTAMP(3);
}

__device__ __forceinline__ float sgp_ag_load(const float* p) {   // agent-scope load (data written earlier in the SAME launch)
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define SGP_RED_LD 260   // floats per (quantity, column) in the end-of-kernel fold: 8 waves x 32 lanes + 4 (bank skew)
// Epilogue of the third strip form (also the early-start form below): fold the per-lane statistics over the 256 (wave,
// row-lane) contributions of every column, then -- a.fin -- the finishing pass and the likelihood head of the strip.
// AG: y was written by a side job of the SAME launch (early-start form): read it with agent-scope loads.
// L16: the sixteen-wave form (sgp_A_strip16_kernel): a lane holds row (lane & 15) of its 16-row sub-tiles and the columns
// 16 jb + 4 (lane >> 4) + r in csq / cu [4 jb + r]; its upper eight waves leave after handing their statistics over.
template <bool AG, bool L16 = false>
__device__ __forceinline__ void sgp_2t_epilogue(const SgpArgs<float>& a, float* lds_raw, const float (&csq)[16], const float (&cu)[16],
                                                const long e, const int bx, const int col0, const int n, const bool means) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  if (a.part) {
    __syncthreads();  // every wave is done reading the K block
    float* red = lds_raw;   // [2 quantities][32 columns][SGP_RED_LD]
    if (L16) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
        red[(0 * SGP_SN + c) * SGP_RED_LD + 16 * w + (lane & 15)] = csq[r];
        if (means) red[(1 * SGP_SN + c) * SGP_RED_LD + 16 * w + (lane & 15)] = cu[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[(0 * SGP_SN + c) * SGP_RED_LD + 32 * w + li] = csq[r];
        if (means) red[(1 * SGP_SN + c) * SGP_RED_LD + 32 * w + li] = cu[r];
      }
    }
    __syncthreads();
    if (L16 && tid >= 512) return;   // (the fold, the finishing pass and the head are written for 512 threads)
    // 8 threads per (quantity, column): 32 contributions each in a fixed order, then a fixed 3-level tree
    const int pair = tid >> 3, g = tid & 7;           // pair = quantity * 32 + column
    const float* rp = red + pair * SGP_RED_LD + 32 * g;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const V4 q = *reinterpret_cast<const V4*>(rp + 4 * i);
      sum += (q[0] + q[1]) + (q[2] + q[3]);
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    const int q = pair >> 5, c = pair & 31;
    if (!a.fin) {
      if (g == 0 && col0 + c < n && (q == 0 || means)) a.part[e * 5 * a.n + (long)q * n + col0 + c] = sum;
      return;
    }
    // The finishing pass, here: one workgroup owns ALL rows of its 32 columns, so the totals just folded are final --
    // v = 1 - sum A^2, f = u A + sqrt|v| eps for the strip (the residual noise drawn from the same per-pair RNG lanes
    // and in the same pair order as the stand-alone pass, sgp_finish_part_kernel, whose launch -- 4.9 us of the cfg-2
    // step for 64 KB of work -- disappears).  Same operations in the same order: same bits.
    __syncthreads();                     // every thread is done reading the fold buffer
    float* tot = lds_raw;                // [2][32]
    if (g == 0) tot[q * 32 + c] = (q == 0 || means) ? sum : 0.f;
    __syncthreads();
    float h_ll = 0.f, h_sc = 0.f, h_vr = 0.f;
    if (tid < 16) {
      const long j0 = col0 + 2 * tid;    // (n is even: a pair is wholly inside or outside)
      if (j0 < n) {
        const long idx0 = e * (long)a.n + j0, pidx = idx0 >> 1;
        float z0 = 0.f, z1 = 0.f;
        if (a.fin_rng) {
          HbRng gen = rng_load(a.fin_rng, a.fin_lanes, pidx);
          gen.normal2(z0, z1);
          rng_store(a.fin_rng, a.fin_lanes, pidx, gen);
        } else if (a.fin_eps_in) {
          z0 = AG ? sgp_ag_load(a.fin_eps_in + idx0) : a.fin_eps_in[idx0];
          z1 = AG ? sgp_ag_load(a.fin_eps_in + idx0 + 1) : a.fin_eps_in[idx0 + 1];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float s2 = 0.f + tot[2 * tid + i];
          const float vv = 1.f - s2;
          const float zi = i ? z1 : z0;
          a.fin_v[idx0 + i] = vv;
          if (a.fin_eps_out && a.fin_eps_out != a.fin_eps_in) a.fin_eps_out[idx0 + i] = zi;
          const float scale = a.fin_diag ? hb_sqrt(hb_abs(vv)) * zi : 0.f;
          const float fj = (0.f + tot[32 + 2 * tid + i]) + scale;
          if (a.P > 0) a.fin_f[(e * a.P) * (long)a.n + j0 + i] = fj;
          if (a.head) {
            // the likelihood head of this point (same operations as hb_gauss_ll's kernels)
            const float hs = a.hscale ? a.hscale[0] : 1.f, hv = a.hvar[0];
            const float iv = 1.f / hv, lc = -0.91893853320467274178f - 0.5f * hb_log(hv);
            float gg;
            hb_gauss_point<float>(AG ? sgp_ag_load(a.hy + idx0 + i) : a.hy[idx0 + i], fj, hs, iv, lc, gg, h_ll, h_sc, h_vr);
            a.hdmu[idx0 + i] = gg;
            if (a.hfbar) a.hfbar[idx0 + i] = hs * (a.hpost * gg);
          }
        }
      }
    }
    if (a.head && tid < 64) {
      // the strip's partial sums: the sixteen pair threads sit in lanes 0..15 of wave 0 (fixed order)
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        h_ll += __shfl_xor(h_ll, off, 16);
        h_sc += __shfl_xor(h_sc, off, 16);
        h_vr += __shfl_xor(h_vr, off, 16);
      }
      if (tid == 0) {
        const long unit = e * (long)((n + SGP_SN - 1) / SGP_SN) + bx;
        a.hpart[unit] = h_ll;
        a.hpart[a.hunits + unit] = h_sc;
        a.hpart[2 * a.hunits + unit] = h_vr;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Column-strip contraction, third form (round 3): transposed accumulators, two workgroups per CU.
//
// The second form keeps a 4.6 KB transpose buffer per wave (fragment-major store of a finished tile) and the staged u
// rows: 113 KB of LDS and 169 registers, i.e. ONE workgroup per CU -- while a workgroup synthesises its K block or
// retires tiles the matrix pipe of its CU idles, and a stalled wave has one partner on its SIMD.  Measured on the
// second form with a (racy) shared buffer, two resident workgroups alone are worth 12 % at cfg-5 size (1521 -> 1342 us).
// Here the tile product is computed transposed (sgp_strip.cuh: the row-per-lane view is eight permlane swaps, no LDS),
// the column statistics are kept per lane in accumulator order (row li of the tile on the lane: u_m is one register per
// tile) and folded across lanes and waves ONCE, through the K block's LDS after the last MFMA: 68 KB of LDS and
// <= 128 registers, so two workgroups (four waves per SIMD) share a CU.  Every A element keeps the bits of the second
// form; the column sums are taken in another (fixed) order.  Handles P <= 1 column means (the registers of more would
// not fit): the launcher keeps the second form for the rest.
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(SGP_STRIP_THREADS, 4) sgp_A_strip2t_kernel(SgpArgs<float> a) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  constexpr int KS_FLOATS = SGP_SN * SGP_SLD, RED_FLOATS = 2 * SGP_SN * SGP_RED_LD;
  __shared__ __attribute__((aligned(16))) float lds_raw[KS_FLOATS > RED_FLOATS ? KS_FLOATS : RED_FLOATS];
  __shared__ __attribute__((aligned(16))) float zs[SGP_SM_MAX * D];
  float (*Ks)[SGP_SLD] = reinterpret_cast<float (*)[SGP_SLD]>(lds_raw);
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const float* __restrict__ x = a.x + e * a.sx;
  const float* __restrict__ z = a.z + e * a.M * D;
  const float* __restrict__ ell = a.ell + e * a.dl;
  const float* __restrict__ Wf = a.Wf + e * a.M * a.M;
  float* __restrict__ A = a.A ? a.A + e * a.M * a.n : nullptr;
  const int M = (int)a.M, n = (int)a.n;
  const int col0 = bx * SGP_SN;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  const int nT = M / 32;
  const int t1 = nT - 1 - w, t0 = w;
  const int d0 = w < t1 ? w + 1 : 0;      // the middle tile of an odd count is taken once, as t1
  const int d1 = w <= t1 ? t1 + 1 : 0;
  const int nts = d0 + d1;
  const bool means = a.part && a.P > 0;
  HB_SSTAMP(0);
  // this lane's u_m for its two tiles (row li of each), requested before anything else
  float u0 = 0.f, u1 = 0.f;
  if (means) {
    const float* up = a.u + e * a.P * a.M;
    if (d0) u0 = up[32 * t0 + li];
    if (d1) u1 = up[32 * t1 + li];
  }
  // ---- K(z, x[strip]) -> LDS (as in the other forms: difference first, then the exp2 scale)
  {
    const int c = tid & 31, kq = tid >> 5;
    const int cc = col0 + c < n ? col0 + c : n - 1;
    float sc[D], xs[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      sc[dd] = float(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : dd];
      xs[dd] = x[cc * D + dd];
    }
    constexpr int NTH = SGP_STRIP_THREADS;
    constexpr int ZIT = (SGP_SM_MAX * D) / NTH;
    float zt[ZIT];
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      zt[it] = z[i < M * D ? i : 0];
    }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      if (i < M * D) zs[i] = zt[it];
    }
    __syncthreads();
#pragma unroll 4
    for (int k4 = kq * 4; k4 < M; k4 += NTH / 8) {
      float zq[4 * D];
#pragma unroll
      for (int q = 0; q < 4 * D; q += 4) {
        const V4 zz = *reinterpret_cast<const V4*>(&zs[k4 * D + q]);
#pragma unroll
        for (int s = 0; s < 4; ++s) zq[q + s] = zz[s];
      }
      V4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float r2 = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const float tt = (zq[q * D + dd] - xs[dd]) * sc[dd];
          r2 += tt * tt;
        }
        v[q] = hb_exp2_neg<float>(r2);
      }
      *reinterpret_cast<V4*>(&Ks[c][k4]) = v;
    }
  }
  __syncthreads();
  HB_SSTAMP(1);

  // per-lane column statistics in ACCUMULATOR order: register r of lane (li, h) is column (r & 3) + 8 (r >> 2) + 4 h
  float csq[16], cu[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) csq[r] = 0.f, cu[r] = 0.f;
  typename MM::Acc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nS = (n + SGP_SN - 1) / SGP_SN;

  auto load = [&](V4 (&f)[4], int ts) {
    const int tc = ts < nts ? ts : nts - 1;            // past the end: re-read the last step (never used)
    const int tile = tc < d0 ? t0 : t1, Q = tc < d0 ? tc : tc - d0;
    const float* p = Wf + ((long)(tile * nT + Q) << 10) + 4 * lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) f[v] = *reinterpret_cast<const V4*>(p + 256 * v);
  };
  auto compute = [&](const V4 (&f)[4], int ts) {
    if (ts >= nts) return;                               // (uniform) the odd tail of the two-step loop
    const int tile = ts < d0 ? t0 : t1, Q = ts < d0 ? ts : ts - d0;
    V4 bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = *reinterpret_cast<const V4*>(&Ks[li][32 * Q + 16 * h + 4 * v]);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = MM::mma(bv[v][s], f[v][s], acc);   // transposed tile: rows on the lanes
    if (ts == d0 - 1 || ts == nts - 1) {
      // the tile is complete: statistics from the registers as they stand, then the row-per-lane view
      if (a.part) {
        // (two columns per instruction, v_pk_fma_f32: these run on the pipe the partner wave's fp32 MFMAs use)
        typedef float V2 __attribute__((ext_vector_type(2)));
        const float um = ts < d0 ? u0 : u1;
        const V2 um2 = {um, um};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const V2 av = {acc[r], acc[r + 1]};
          const V2 q2 = __builtin_elementwise_fma(av, av, V2{csq[r], csq[r + 1]});
          const V2 m2 = __builtin_elementwise_fma(um2, av, V2{cu[r], cu[r + 1]});
          csq[r] = q2[0], csq[r + 1] = q2[1];
          cu[r] = m2[0], cu[r + 1] = m2[1];
        }
      }
      sgp_acc_t_settle(acc);
      float row16[16];
      sgp_acc_t_rows(acc, row16);
      if (a.Af) sgp_store_frag_rows(a.Af, row16, e, nT, nS, tile, bx, col0, n, lane);
      if (A) {
        float* ap = A + (long)(32 * tile + li) * n + col0 + 16 * h;
        if ((n & 3) == 0 && col0 + SGP_SN <= n) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            V4 q;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) q[s2] = row16[4 * v + s2];
            *reinterpret_cast<V4*>(ap + 4 * v) = q;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (col0 + 16 * h + i < n) ap[i] = row16[i];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
  };
  if (nts > 0) {
    V4 fa[4], fb[4];
    load(fa, 0);
#pragma nounroll
    for (int ts = 0; ts < nts; ts += 2) {
      load(fb, ts + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(fa, ts);
      __builtin_amdgcn_sched_barrier(0);
      load(fa, ts + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(fb, ts + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  HB_SSTAMP(2);
  sgp_2t_epilogue<false>(a, lds_raw, csq, cu, e, bx, col0, n, means);
  HB_SSTAMP(3);
}

// ---------------------------------------------------------------------------------------------------------------
// Sixteen-wave form (round 4): four waves per SIMD where the grid gives every CU exactly ONE strip (cfg 2: 256 strips).
//
// At n = 8192 the third form runs one 8-wave workgroup per CU: two waves per SIMD, whose tile-steps (global loads ->
// LDS reads -> 16 dependent MFMAs) leave the matrix pipe idle for a fifth of the loop (tools/strip_stamps.hip: 43 k
// cycles for 34.8 k of MFMAs) -- there is no second workgroup to fill the gaps as at cfg-5 size.  Here the SAME strip is
// cut into 16-row sub-tiles on v_mfma_f32_16x16x4_f32 (same rate per CU) and the workgroup has 16 waves, wave w owning
// sub-tiles w and 31 - w (17 k-steps each way: balanced).  No new image is needed: a 16x16x4 operand wants lane (n, g)
// to bring 4 contraction indices of row n, and MFMA e of a batch uses element e of everybody's 16-byte group -- so lane
// (n, g) of batch b loads group (v = g, lane (n + 16 half, h = b)) of the existing fragment-major block of W, and the
// K block is read as Ks[16 jb + i][32 Q + 16 b + 4 g .. + 3]: both sides name k = 32 Q + 16 b + 4 g + e.  The
// accumulator comes out with the ROW on the lane and four consecutive columns in its registers, which IS a 16-byte
// group of the fragment-major A: no lane swaps at all.  A differs from the other forms in the last place (four products
// per MFMA instead of two); statistics, finishing pass and head are the shared epilogue.
// ---------------------------------------------------------------------------------------------------------------
#define SGP_S16_THREADS 1024
template <int D>
__global__ void __launch_bounds__(SGP_S16_THREADS, 4) sgp_A_strip16_kernel(SgpArgs<float> a) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  constexpr int KS_FLOATS = SGP_SN * SGP_SLD, RED_FLOATS = 2 * SGP_SN * SGP_RED_LD;
  __shared__ __attribute__((aligned(16))) float lds_raw[KS_FLOATS > RED_FLOATS ? KS_FLOATS : RED_FLOATS];
  __shared__ __attribute__((aligned(16))) float zs[SGP_SM_MAX * D];
  float (*Ks)[SGP_SLD] = reinterpret_cast<float (*)[SGP_SLD]>(lds_raw);
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const float* __restrict__ x = a.x + e * a.sx;
  const float* __restrict__ z = a.z + e * a.M * D;
  const float* __restrict__ ell = a.ell + e * a.dl;
  const float* __restrict__ Wf = a.Wf + e * a.M * a.M;
  float* __restrict__ A = a.A ? a.A + e * a.M * a.n : nullptr;
  const int M = (int)a.M, n = (int)a.n;
  const int col0 = bx * SGP_SN;
  const int tid = threadIdx.x, lane = tid & 63, m16 = lane & 15, g4 = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nT = M / 32, nS16 = M / 16;
  const bool means = a.part && a.P > 0;
  const int nS = (n + SGP_SN - 1) / SGP_SN;

  // ---- K(z, x[strip]) -> LDS (as in the other forms: difference first, then the exp2 scale)
  {
    const int c = tid & 31, kq = tid >> 5;
    const int cc = col0 + c < n ? col0 + c : n - 1;
    float sc[D], xs[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      sc[dd] = float(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : dd];
      xs[dd] = x[cc * D + dd];
    }
    constexpr int NTH = SGP_S16_THREADS;
    for (int i = tid; i < M * D; i += NTH) zs[i] = z[i];
    __syncthreads();
#pragma unroll 4
    for (int k4 = kq * 4; k4 < M; k4 += NTH / 8) {
      float zq[4 * D];
#pragma unroll
      for (int q = 0; q < 4 * D; q += 4) {
        const V4 zz = *reinterpret_cast<const V4*>(&zs[k4 * D + q]);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) zq[q + s2] = zz[s2];
      }
      V4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float r2 = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const float tt = (zq[q * D + dd] - xs[dd]) * sc[dd];
          r2 += tt * tt;
        }
        v[q] = hb_exp2_neg<float>(r2);
      }
      *reinterpret_cast<V4*>(&Ks[c][k4]) = v;
    }
  }
  __syncthreads();

  float csq[16], cu[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) csq[r] = 0.f, cu[r] = 0.f;
  V4 acc[2];
  const V4 zero4 = {0.f, 0.f, 0.f, 0.f};
  acc[0] = zero4, acc[1] = zero4;

  // sub-tile s (rows 16 s .. 16 s + 15) contracts the 32-wide chunks Q = 0 .. s >> 1
  auto load = [&](V4 (&f)[2], int s, int Q) {
    const float* p = Wf + ((long)((s >> 1) * nT + Q) << 10) + (g4 * 64 + 16 * (s & 1) + m16) * 4;
    f[0] = *reinterpret_cast<const V4*>(p);
    f[1] = *reinterpret_cast<const V4*>(p + 128);     // batch b = 1: the other half-wave's groups (h = 1)
  };
  auto mma_step = [&](const V4 (&f)[2], int Q) {
    V4 kf[2][2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int b = 0; b < 2; ++b) kf[jb][b] = *reinterpret_cast<const V4*>(&Ks[16 * jb + m16][32 * Q + 16 * b + 4 * g4]);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0][b][ee], f[b][ee], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[1][b][ee], f[b][ee], acc[1], 0, 0, 0);
      }
  };
  auto retire = [&](int s) {
    const int T = s >> 1, li = 16 * (s & 1) + m16;
    if (a.part) {
      const float um = means ? (a.u + e * a.P * a.M)[16 * s + m16] : 0.f;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          csq[4 * jb + r] += acc[jb][r] * acc[jb][r];
          cu[4 * jb + r] += um * acc[jb][r];
        }
    }
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      V4 q = acc[jb];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + 16 * jb + 4 * g4 + r >= n) q[r] = 0.f;
      if (a.Af) *reinterpret_cast<V4*>(a.Af + ((((long)e * nT + T) * nS + bx) << 10) + (g4 * 64 + 32 * jb + li) * 4) = q;
      if (A) {
        float* ap = A + (long)(32 * T + li) * n + col0 + 16 * jb + 4 * g4;
        if ((n & 3) == 0 && col0 + SGP_SN <= n) {
          *reinterpret_cast<V4*>(ap) = q;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col0 + 16 * jb + 4 * g4 + r < n) ap[r] = q[r];
        }
      }
    }
    acc[0] = zero4, acc[1] = zero4;
  };
  // the wave's two sub-tiles as one flat sequence of k-steps, operands in two register sets used alternately
#pragma unroll 1
  for (int pr = w; pr < nS16 / 2; pr += SGP_S16_THREADS / 64) {
    const int s0 = pr, s1 = nS16 - 1 - pr;
    const int d0 = (s0 >> 1) + 1, nts = d0 + (s1 >> 1) + 1;
    auto at = [&](int ts, int& s, int& Q) {
      const int tc = ts < nts ? ts : nts - 1;          // past the end: re-read the last step (never used)
      s = tc < d0 ? s0 : s1;
      Q = tc < d0 ? tc : tc - d0;
    };
    V4 fa[2], fb[2];
    int s, Q;
    at(0, s, Q);
    load(fa, s, Q);
#pragma nounroll
    for (int ts = 0; ts < nts; ts += 2) {
      at(ts + 1, s, Q);
      load(fb, s, Q);
      __builtin_amdgcn_sched_barrier(0);
      at(ts, s, Q);
      mma_step(fa, Q);
      if (ts == d0 - 1 || ts == nts - 1) retire(s);
      __builtin_amdgcn_sched_barrier(0);
      at(ts + 2, s, Q);
      load(fa, s, Q);
      __builtin_amdgcn_sched_barrier(0);
      if (ts + 1 < nts) {
        at(ts + 1, s, Q);
        mma_step(fb, Q);
        if (ts + 1 == d0 - 1 || ts + 1 == nts - 1) retire(s);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  sgp_2t_epilogue<false, true>(a, lds_raw, csq, cu, e, bx, col0, n, means);
}

// ---------------------------------------------------------------------------------------------------------------
// Early-start form (round 4): the forward contraction inside the persistent Cholesky's launch.
//
// The persistent factorisation (chol_persist.cuh) is a chain of 512 dependent pivot columns: 60 us at M = 512 on nb^2 = 64
// workgroups, with three quarters of the chip idle, and A = W K(z, x) used to start only when its launch had ended.  But
// row block j of W = L^-1 (64 rows) is FINAL as soon as panel j is factored (W(j, c) = Y(c, j)^T; the elimination of
// [A; I] finishes column block j of Y in panel j).  So the strips of the forward contraction ride in the same grid: a
// strip workgroup synthesises its K block at once, then takes the row tiles of A in the order in which their rows of
// the W image become final -- a per-(matrix, row block) counter raised by the factorisation's workgroups behind
// write-through stores of the image, polled here, the image read with agent-scope loads (cdna_hip_programming.md
// Guideline 16 R1, as for the panels) -- and only the LAST row block's work remains when the factorisation ends.  That
// remainder (2 tiles x nT k-steps) is split three ways over six waves (k-steps p, p + 3, ...; partial accumulators
// meet through one 4 KB LDS slot per tile, owner + helper 1 + helper 2 in that order: fixed summation order).
//   Roles come from the launch's ticket counter in START order: factorisation first, then the side jobs of the launch
// (minibatch gather, sample of q(u): this form reads their outputs, so they release them and raise sync[3]), then the
// strips -- every wait is for a lower ticket, i.e. for a workgroup that is already running (no residency assumption),
// and every wait is bounded (CpWait).  All roles take part in the arrival count that ends in info[] and zeroed sync words.
//   Every tile of A but the last two keeps the bits of sgp_A_strip2t_kernel; those two differ by the order of three partial sums.
// ---------------------------------------------------------------------------------------------------------------
#define SGP_E_DMAX 2
#define SGP_E_RAW_BYTES (2 * SGP_SN * SGP_RED_LD * 4)                    // K block / fold buffer (the larger of the two)
#define SGP_E_ZS_OFF SGP_E_RAW_BYTES
#define SGP_E_RED_OFF (SGP_E_ZS_OFF + SGP_SM_MAX * SGP_E_DMAX * 4)
#define SGP_E_CTL_OFF (SGP_E_RED_OFF + 2 * 4096)
#define SGP_E_LDS_BYTES (SGP_E_CTL_OFF + 64)     // control words: [0], [1] slot states, [3] arrival, [4] poll lock, [8 + j] row block j known ready
static_assert(SGP_SN * SGP_SLD * 4 <= SGP_E_RAW_BYTES, "K block must fit the raw area");
static_assert(sizeof(CpLds) <= SGP_E_LDS_BYTES, "the factorisation's LDS is overlaid on the strip form's");
static_assert(2 * SGP_E_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

template <int D>
__device__ __forceinline__ void sgp_early_body(const SgpArgs<float>& a, const CpArgs& ca, char* lds, const int unit) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  float* lds_raw = reinterpret_cast<float*>(lds);
  float (*Ks)[SGP_SLD] = reinterpret_cast<float (*)[SGP_SLD]>(lds_raw);
  float* zs = reinterpret_cast<float*>(lds + SGP_E_ZS_OFF);
  float* red3 = reinterpret_cast<float*>(lds + SGP_E_RED_OFF);     // [tile of the last row block][4 v][64 lanes][4]
  int* ctl = reinterpret_cast<int*>(lds + SGP_E_CTL_OFF);           // (see SGP_E_LDS_BYTES)
  const int M = (int)a.M, n = (int)a.n, nT = M / 32, nb = ca.nb;
  const int nS = (n + SGP_SN - 1) / SGP_SN;
  const long e = unit / nS;
  const int bx = unit - (int)e * nS;
  const float* __restrict__ x = a.x + e * a.sx;
  const float* __restrict__ z = a.z + e * a.M * D;
  const float* __restrict__ ell = a.ell + e * a.dl;
  float* __restrict__ A = a.A ? a.A + e * a.M * a.n : nullptr;
  const int col0 = bx * SGP_SN;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool means = a.part && a.P > 0;
  CpWait wt = {ca.sync + 2, __builtin_amdgcn_s_memrealtime() + CP_TIMEOUT_TICKS, false};
  if (tid < 16) ctl[tid] = 0;
  // the side jobs of this launch have released x, u, y (ONE polling wave per workgroup: two thousand waves polling one
  // word starve the factorisation's own flag traffic)
  if (ca.nside > 0 && w == 0) wt.wait(ca.sync + 3, (unsigned)ca.nside);
  __syncthreads();

  // ---- K(z, x[strip]) -> LDS (as in the other forms: difference first, then the exp2 scale)
  {
    const int c = tid & 31, kq = tid >> 5;
    const int cc = col0 + c < n ? col0 + c : n - 1;
    float sc[D], xs[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      sc[dd] = float(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : dd];
      xs[dd] = sgp_ag_load(x + cc * D + dd);
    }
    constexpr int NTH = SGP_STRIP_THREADS;
    constexpr int ZIT = (SGP_SM_MAX * D) / NTH;
    float zt[ZIT];
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      zt[it] = z[i < M * D ? i : 0];
    }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      if (i < M * D) zs[i] = zt[it];
    }
    __syncthreads();
#pragma unroll 4
    for (int k4 = kq * 4; k4 < M; k4 += NTH / 8) {
      float zq[4 * D];
#pragma unroll
      for (int q = 0; q < 4 * D; q += 4) {
        const V4 zz = *reinterpret_cast<const V4*>(&zs[k4 * D + q]);
#pragma unroll
        for (int s = 0; s < 4; ++s) zq[q + s] = zz[s];
      }
      V4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float r2 = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const float tt = (zq[q * D + dd] - xs[dd]) * sc[dd];
          r2 += tt * tt;
        }
        v[q] = hb_exp2_neg<float>(r2);
      }
      *reinterpret_cast<V4*>(&Ks[c][k4]) = v;
    }
  }
  __syncthreads();

  float csq[16], cu[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) csq[r] = 0.f, cu[r] = 0.f;
  typename MM::Acc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wf) + e * a.M * a.M, 0,
                                                                      (int)(a.M * a.M * sizeof(float)), 0x00020000);
  const unsigned* wready = ca.sync + cp_wready_off(ca.B, nb) + e * nb;

  auto load = [&](V4 (&f)[4], int tile, int Q) {
    const int off = ((tile * nT + Q) << 12) + 16 * lane;      // bytes
#pragma unroll
    // (agent-scope loads, as for everything another workgroup of the same launch has written; measured against plain
    // L2-cached loads -- which would be safe here, the image being written through before its counter is raised and every
    // L2 having been invalidated when the launch started -- the time is the same: profiles/r04_early_forward_check_*.txt)
    for (int v = 0; v < 4; ++v) f[v] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(wr, off + 1024 * v, 0, 16));
  };
  auto mma_step = [&](const V4 (&f)[4], int Q) {
    V4 bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = *reinterpret_cast<const V4*>(&Ks[li][32 * Q + 16 * h + 4 * v]);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = MM::mma(bv[v][s], f[v][s], acc);   // transposed tile: rows on the lanes
  };
  // k-steps q0, q0 + dq, ... <= tile of one row tile, operands in two register sets used alternately
  auto run = [&](int tile, int q0, int dq) {
    if (q0 > tile || (ca.early & 4)) return;
    const int ns = (tile - q0) / dq + 1;
    V4 fa[4], fb[4];
    load(fa, tile, q0);
#pragma nounroll
    for (int i = 0; i < ns; i += 2) {
      const int Q0 = q0 + i * dq, Q1 = Q0 + dq, Q2 = Q1 + dq;
      load(fb, tile, Q1 <= tile ? Q1 : Q0);             // past the end: re-read (never used)
      __builtin_amdgcn_sched_barrier(0);
      mma_step(fa, Q0);
      __builtin_amdgcn_sched_barrier(0);
      load(fa, tile, Q2 <= tile ? Q2 : Q0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < ns) mma_step(fb, Q1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // a complete tile: statistics from the registers as they stand, then the row-per-lane view and the stores
  auto retire = [&](int tile, float um) {
    if (a.part) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        csq[r] += acc[r] * acc[r];
        cu[r] += um * acc[r];
      }
    }
    sgp_acc_t_settle(acc);
    float row16[16];
    sgp_acc_t_rows(acc, row16);
    if (a.Af) sgp_store_frag_rows(a.Af, row16, e, nT, nS, tile, bx, col0, n, lane);
    if (A) {
      float* ap = A + (long)(32 * tile + li) * n + col0 + 16 * h;
      if ((n & 3) == 0 && col0 + SGP_SN <= n) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          V4 q;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) q[s2] = row16[4 * v + s2];
          *reinterpret_cast<V4*>(ap + 4 * v) = q;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (col0 + 16 * h + i < n) ap[i] = row16[i];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };
  // row block j of the W image is final.  At most one wave of the workgroup polls the global counter at a time (LDS
  // lock), the others watch the LDS copy of the answer.
  auto wait_ready = [&](int j) {
    if (wt.dead) return;
    unsigned spins = 0;
    for (;;) {
      if (__hip_atomic_load(ctl + 8 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
      if (__hip_atomic_exchange(ctl + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
        const unsigned c = __hip_atomic_load(wready + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = c >= (unsigned)j + 1u;
        if (ok) __hip_atomic_store(ctl + 8 + j, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool give_up = false;
        if (!ok && (++spins & 31u) == 0u &&
            (__hip_atomic_load(wt.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || __builtin_amdgcn_s_memrealtime() > wt.deadline)) {
          __hip_atomic_store(wt.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(ctl + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          give_up = true;
        }
        __hip_atomic_store(ctl + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ok) break;
        if (give_up) { wt.dead = true; return; }
        for (int nap = 0; nap < ca.poll_naps; ++nap) __builtin_amdgcn_s_sleep(8);
      } else {
        if (__hip_atomic_load(ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) { wt.dead = true; return; }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the image loads below the poll
  };
  auto lds_wait = [&](const int* p, int v) {
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  const float* up = a.u + e * a.P * a.M;

  // ---- row blocks 0 .. nb - 2: tile t belongs to wave t & 7, whole
  const int last0 = nT - 2;
#pragma unroll 1
  for (int tile = w; tile < last0; tile += SGP_STRIP_THREADS / 64) {
    wait_ready(tile >> 1);
    const float um = means ? sgp_ag_load(up + 32 * tile + li) : 0.f;
    run(tile, 0, 1);
    retire(tile, um);
  }
  // ---- the last row block: tile last0 + th is split over the waves (last0 + th + 2 p) & 7, p = 0 (owner), 1, 2
  {
    const int th = w & 1, tile = last0 + th;
    const int p = ((w - th - last0) & 7) >> 1;
    if (p < 3 && p <= tile) {
      wait_ready(nb - 1);
      const float um = (means && p == 0) ? sgp_ag_load(up + 32 * tile + li) : 0.f;
      run(tile, p, 3);
      float* slot = red3 + th * 1024;
      int* st = ctl + th;
      if (p == 0) {
#pragma unroll 1
        for (int hp = 1; hp <= 2 && hp <= tile; ++hp) {
          lds_wait(st, 2 * hp - 1);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const V4 q = *reinterpret_cast<const V4*>(&slot[(v * 64 + lane) * 4]);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[4 * v + s2] += q[s2];
          }
          asm volatile("" ::: "memory");
          __hip_atomic_store(st, 2 * hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS runs a wave's operations in order)
          asm volatile("" ::: "memory");
        }
        retire(tile, um);
      } else {
        lds_wait(st, 2 * (p - 1));
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const V4 q = {acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]};
          *reinterpret_cast<V4*>(&slot[(v * 64 + lane) * 4]) = q;
        }
        asm volatile("" ::: "memory");
        __hip_atomic_store(st, 2 * p - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
      }
    }
  }

  sgp_2t_epilogue<true>(a, lds_raw, csq, cu, e, bx, col0, n, means);
  cp_arrive(ca, *reinterpret_cast<unsigned*>(ctl + 3));
}

// One launch, three roles by ticket (start order): the persistent factorisation, the launch's side jobs, the strips.
template <int D>
__global__ void __launch_bounds__(512, 4) chol_sgp_fwd_kernel(CpArgs ca, HbSideJobs side, SgpArgs<float> a) {
  __shared__ __attribute__((aligned(16))) char lds[SGP_E_LDS_BYTES];
  __shared__ unsigned s_ticket;
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&ca.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned t = s_ticket;
  if (t < (unsigned)ca.total) {
    chol_persist_body(ca, *reinterpret_cast<CpLds*>(lds), t);
    return;
  }
  if (t < (unsigned)(ca.total + ca.nside)) {
    // (the job bodies are written for 256-thread blocks: the upper half of this block leaves)
    if (threadIdx.x >= 256) return;
    hb_side_run(side, (int)t - ca.total);
    __syncthreads();   // every wave's stores have left for L2
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&ca.sync[3], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // L2 write-back, then the count
    return;
  }
  if (ca.early & 2) {   // (diagnostic: the strips leave at once)
    cp_arrive(ca, s_ticket);
    return;
  }
  sgp_early_body<D>(a, ca, lds, (int)t - ca.total - ca.nside);
}

// ---- the recorded forward ("rider"): hb_sgp_rider_begin() ... hb_sgp_fwd_* ... hb_cholesky_inverse_f32 / hb_gram_cholesky_inverse_f32
struct SgpRider {
  bool armed = false, valid = false;
  SgpArgs<float> a;
  long E = 0;
};
static thread_local SgpRider g_rider;
static int sgp_A_strip_launch(SgpArgs<float> a, long E, hipStream_t stream);
static inline bool sgp_fused_finish_ok(long E, long n, long M, long d, long P, int prec, bool has_wfrag, bool draw, long rng_lanes);

extern "C" int hb_sgp_rider_supported(long E, long n, long M, long d, long P, int prec, int has_wfrag, int draw, long rng_lanes) {
  if (hb_debug_get("sgp_early", 1) == 0) return 0;
  return sgp_fused_finish_ok(E, n, M, d, P, prec, has_wfrag != 0, draw != 0, rng_lanes) && d <= SGP_E_DMAX && M % CP_NB == 0 &&
                 hb_cholesky_persistent_shape(E, M, 4)
             ? 1
             : 0;
}
extern "C" int hb_sgp_rider_begin(void) {
  HB_REQUIRE(!g_rider.valid, "hb_sgp_rider_begin: a recorded forward is still pending (hb_sgp_rider_flush)");
  g_rider.armed = true;
  return 0;
}
extern "C" int hb_sgp_rider_pending(void) { return g_rider.valid ? 1 : 0; }
int hb_sgp_rider_launch_alone(hipStream_t stream) {
  g_rider.armed = false;
  if (!g_rider.valid) return 0;
  g_rider.valid = false;
  return sgp_A_strip_launch(g_rider.a, g_rider.E, stream);
}
extern "C" int hb_sgp_rider_flush(void* stream) { return hb_sgp_rider_launch_alone((hipStream_t)stream); }

int hb_sgp_rider_launch_with(CpArgs& ca, const HbSideJobs& sj, hipStream_t stream) {
  if (!g_rider.valid) return 0;
  const SgpArgs<float>& a = g_rider.a;
  const long E = g_rider.E, nS = hb_cdiv(a.n, SGP_SN);
  static int cus = 0;   // (one device per process)
  if (cus == 0 && (hb_device_info(nullptr, 0, &cus) || cus <= 0)) cus = 256;
  const bool ok = a.Wf == ca.Wf && !ca.bf16x3 && E == ca.B && a.M == ca.M && a.M <= SGP_SM_MAX && a.d >= 1 && a.d <= SGP_E_DMAX &&
                  a.fin && !a.W3 && (long)ca.total + sj.total + E * nS <= 2L * cus && hb_debug_get("sgp_early", 1) != 0;
  if (!ok) return 0;   // (the caller launches the factorisation alone, then hb_sgp_rider_launch_alone)
  ca.early = 1 | ((int)(hb_debug_get("sgp_early_diag", 0) & 3) << 1), ca.nside = sj.total, ca.arrive = ca.total + (int)(E * nS);   // (diag: timing variants)
  g_rider.valid = false, g_rider.armed = false;
  ca.poll_naps = (int)hb_debug_get("sgp_early_poll_naps", 1);
  long nunits = E * nS;
  if (hb_debug_get("sgp_early_diag", 0) == 4) nunits = 0, ca.arrive = ca.total;   // (diagnostic: the fused binary without its strips)
  const dim3 grid((unsigned)(ca.total + sj.total + nunits));
  if (a.d == 1)
    hipLaunchKernelGGL((chol_sgp_fwd_kernel<1>), grid, dim3(512), 0, stream, ca, sj, a);
  else
    hipLaunchKernelGGL((chol_sgp_fwd_kernel<2>), grid, dim3(512), 0, stream, ca, sj, a);
  HB_LAUNCH_CHECK();
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Column-strip contraction with bf16x3 operands ("fp16-in / fp32-accumulate" variant of BASELINE cfg 5, made usable).
// Plain 16-bit operands are NOT usable for the whitened solve: entries of W = L^-1 reach +-30 and cancel, a single
// bf16 rounding of W and K leaves a 27 % error in A = W K (profiles/r01_bf16_split_study.txt).  Each fp32 operand
// is therefore split into three bf16 terms x = hi + mid + lo (W once per step by hb_cholesky_inverse, the RBF block
// as it is synthesised); the six products hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi carry all but O(2^-24) of
// the fp32 product and run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 12 MFMAs of 32 cycles per tile-step
// instead of 16 of 64.  Same decomposition and loop shape as sgp_A_strip2_kernel; the operand images are
// fragment-major so every load is one contiguous kilobyte per wave.
// ---------------------------------------------------------------------------------------------------------------
#define SGP_S3LD (SGP_SM_MAX + 8)  // bf16 row stride of the K planes: 1040 B, lanes of a column group hit distinct banks
template <int D>
__global__ void __launch_bounds__(SGP_STRIP_THREADS) sgp_A_strip3_kernel(SgpArgs<float> a) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef __bf16 B8 __attribute__((ext_vector_type(8)));
  typedef Mma<float> MM;
  __shared__ __attribute__((aligned(16))) __bf16 K3[3][SGP_SN][SGP_S3LD];
  __shared__ __attribute__((aligned(16))) float zs[SGP_SM_MAX * D];
  __shared__ float us[4][SGP_SM_MAX];
  __shared__ __attribute__((aligned(16))) float Tw[SGP_STRIP_THREADS / 64][32][SGP_TLD];
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const float* __restrict__ x = a.x + e * a.sx;
  const float* __restrict__ z = a.z + e * a.M * D;
  const float* __restrict__ ell = a.ell + e * a.dl;
  const __bf16* __restrict__ W3 = reinterpret_cast<const __bf16*>(a.W3) + e * a.M * a.M;
  float* __restrict__ A = a.A ? a.A + e * a.M * a.n : nullptr;
  const int M = (int)a.M, n = (int)a.n;
  const int col0 = bx * SGP_SN;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;

  // ---- K(z, x[strip]) -> LDS as three bf16 planes [term][column][k]
  {
    const int c = tid & 31, kq = tid >> 5;
    const int cc = col0 + c < n ? col0 + c : n - 1;
    float sc[D], xs[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      sc[dd] = float(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : dd];
      xs[dd] = x[cc * D + dd];   // raw: the difference is taken first, then scaled (see hb_exp2_neg in sgp_strip.cuh)
    }
    constexpr int NTH = SGP_STRIP_THREADS;
    constexpr int ZIT = (SGP_SM_MAX * D) / NTH, UIT = SGP_SM_MAX / NTH;
    float zt[ZIT], ut[4][UIT];
    const int npu = a.part ? ((int)a.P < 4 ? (int)a.P : 4) : 0;
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      zt[it] = z[i < M * D ? i : 0];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        ut[p][it] = p < npu ? a.u[e * a.P * a.M + (long)p * M + (i < M ? i : 0)] : 0.f;
      }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      if (i < M * D) zs[i] = zt[it];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        if (p < npu && i < M) us[p][i] = ut[p][it];
      }
    __syncthreads();
    // thread (c, kq) takes the groups of 8 consecutive k: kq*8, kq*8 + 128, ...
#pragma unroll 2
    for (int k8 = kq * 8; k8 < M; k8 += NTH / 4) {
      B8 p0, p1, p2;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float r2 = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const float tt = (zs[(k8 + q) * D + dd] - xs[dd]) * sc[dd];
          r2 += tt * tt;
        }
        const float kv = hb_exp2_neg<float>(r2);
        const __bf16 b0 = (__bf16)kv;
        const float r1 = kv - (float)b0;
        const __bf16 b1 = (__bf16)r1;
        p0[q] = b0, p1[q] = b1, p2[q] = (__bf16)(r1 - (float)b1);
      }
      *reinterpret_cast<B8*>(&K3[0][c][k8]) = p0;
      *reinterpret_cast<B8*>(&K3[1][c][k8]) = p1;
      *reinterpret_cast<B8*>(&K3[2][c][k8]) = p2;
    }
  }
  __syncthreads();

  const int nT = M / 32;
  const int t1 = nT - 1 - w, t0 = w;
  const int d0 = w < t1 ? w + 1 : 0;
  const int d1 = w <= t1 ? t1 + 1 : 0;
  const int nts = d0 + d1;
  const int npart = a.part ? (int)a.P : 0;
  float cs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int gc = col0 + li;
  typename MM::Acc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  struct Frag {
    B8 f[3][2];  // [term][k16-step]
  };
  auto load = [&](Frag& fr, int ts) {
    const int tc = ts < nts ? ts : nts - 1;
    const int tile = tc < d0 ? t0 : t1, Q = tc < d0 ? tc : tc - d0;
    const __bf16* p = W3 + ((long)(tile * nT + Q) << 10) + 8 * lane;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) fr.f[t][q] = *reinterpret_cast<const B8*>(p + (long)t * a.plane3 + 512 * q);
  };
  auto compute = [&](const Frag& fr, int ts) {
    if (ts >= nts) return;
    const int tile = ts < d0 ? t0 : t1, Q = ts < d0 ? ts : ts - d0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      B8 b[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const B8*>(&K3[t][li][32 * Q + 16 * q + 8 * h]);
      // smallest terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f[2][q], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f[0][q], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f[1][q], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f[1][q], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f[0][q], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f[0][q], b[0], acc, 0, 0, 0);
    }
    if (ts == d0 - 1 || ts == nts - 1) {
      if (a.Af) {
        // bf16x3 planes of A for the backward kernels (HB_PREC_BF16X3: A_frag holds 3 bf16 planes)
        float row16[16];
        sgp_tile_rows(Tw[w], acc, row16, col0, n, lane);
        const int nSs = (n + SGP_SN - 1) / SGP_SN;
        sgp_store_frag3_row(reinterpret_cast<__bf16*>(a.Af), (long)a.plane3 / a.M * 32 * nSs, row16, e, nT, nSs, tile, bx, lane);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * tile + MM::acc_row(lane, r);
        const float v = acc[r];
        if (a.A && gc < n) A[(long)row * n + gc] = v;
        if (a.part) {
          cs[0] += v * v;
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (p < npart) cs[1 + p] += us[p][row] * v;
        }
        acc[r] = 0.f;
      }
    }
  };
  if (nts > 0) {
    Frag fa, fb;
    load(fa, 0);
#pragma nounroll
    for (int ts = 0; ts < nts; ts += 2) {
      load(fb, ts + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(fa, ts);
      __builtin_amdgcn_sched_barrier(0);
      load(fa, ts + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(fb, ts + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (a.part) {
#pragma unroll
    for (int q = 0; q < 5; ++q) cs[q] += __shfl_xor(cs[q], 32);
    __syncthreads();
    float* red = reinterpret_cast<float*>(&K3[0][0][0]);  // [8 waves][5][32]
    if (lane < 32) {
#pragma unroll
      for (int q = 0; q < 5; ++q) red[(w * 5 + q) * 32 + lane] = cs[q];
    }
    __syncthreads();
    if (tid < 32 && col0 + tid < n) {
      float* pp = a.part + e * 5 * a.n + col0 + tid;
      for (int q = 0; q < 1 + npart; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < SGP_STRIP_THREADS / 64; ++ww) sum += red[(ww * 5 + q) * 32 + tid];
        pp[(long)q * n] = sum;
      }
    }
  }
}

// diagnostic switch: HB_SGP_NO_STRIP=1 forces the tiled kernels (A/B timing)
static inline bool hb_sgp_no_strip() {
  return hb_debug_get("sgp_no_strip", 0) != 0;   // diagnostic (hb_debug_set)
}
// The strip form wins when the tiled kernel would leave ~one workgroup per CU (cfg 2: 33.5 -> 28.8 us); with
// many experts / a long minibatch the tiled kernel runs several workgroups per CU and stays the faster one
// (cfg 5, E = 8, n = 65536: 1.68 ms tiled vs 1.76 ms strip, HB_SGP_FORCE_STRIP=1).
static inline bool sgp_strip_ok(long E, long n, long M, long d, const void* W) {
  const long tiled_wgs = E * hb_cdiv(n, SGP_BN) * ((hb_cdiv(M, SGP_BM) + 1) / 2);
  const bool force = hb_debug_get("sgp_force_strip", 0) != 0;  // diagnostic: strip form whenever it is applicable
  return M >= 32 && M <= SGP_SM_MAX && M % 32 == 0 && d <= SGP_DREG && ((uintptr_t)W % 16 == 0) && (tiled_wgs < 1024 || force);
}

// 1 when hb_sgp_fwd / hb_sgp_bwd, given the fragment-major images of W (Wfrag), run in column-strip form -- the form
// that can also exchange A and Kbar in fragment-major layout (A_frag / Kbar_frag)
extern "C" int hb_sgp_strip_path(long E, long n, long M, long d, long P, int prec) {
  if (hb_sgp_no_strip()) return 0;
  if (!(M >= 32 && M <= SGP_SM_MAX && M % 32 == 0 && d >= 1 && d <= SGP_DREG && P <= 4 && n > 0 && E >= 1)) return 0;
  if (prec == HB_PREC_BF16X3) return 1;
  // Round 1's crossover (tiled kernels for many experts / long minibatches: 1.68 vs 1.76 ms at cfg 5 with the first
  // strip kernel) no longer holds with the fragment-major W / A / Kbar exchange: cfg 5 (E = 8, n = 65536) runs
  // 6.03 ms per step tiled against 5.33 ms in strip form (sgp_grad 3742 -> 3371 us, sgp 1554 -> 1479 us).
  // HB_SGP_TILED_CROSSOVER=1 restores the old rule (diagnostic).
  const bool old_rule = hb_debug_get("sgp_tiled_crossover", 0) != 0;
  // (the backward's strip partials [nS][2d + P][M] and at least one Lbar slab must fit the 32 M^2 workspace per expert)
  if (!old_rule) return (E * hb_cdiv(n, SGP_SN) <= (1L << 22) && (long)hb_cdiv(n, SGP_SN) * (2 * d + P) <= 31 * M) ? 1 : 0;
  const long tiled_wgs = E * hb_cdiv(n, SGP_BN) * ((hb_cdiv(M, SGP_BM) + 1) / 2);
  return (tiled_wgs < 1024 && E * hb_cdiv(n, SGP_SN) <= 4096) ? 1 : 0;
}

static int sgp_A_strip_launch(SgpArgs<float> a, long E, hipStream_t stream) {
  dim3 grid = sgp_grid(hb_cdiv(a.n, SGP_SN), 1, E, a.efast);
  // third form (transposed accumulators, two workgroups per CU) whenever its one column mean suffices;
  // HB_SGP_STRIP_FORM2=1 keeps the second form (diagnostic: A/B timing, the two forms agree bit for bit in A)
  const bool form3 = (!a.part || a.P <= 1) && hb_debug_get("sgp_strip_form2", 0) == 0;
  // sixteen-wave form: where every CU gets at most ONE strip, so that the third form could not put two workgroups on it
  // (M >= 384: at least 12 of the 16 waves own a pair of sub-tiles)
  static int cus = 0;
  if (cus == 0 && (hb_device_info(nullptr, 0, &cus) || cus <= 0)) cus = 256;
  // OFF by default: stand-alone it is 5-7 % faster than the third form at cfg 2 (26.2 / 25.4 against 28.2 / 26.7 us), inside
  // the step it is 1.2 us SLOWER (201.5 against 200.3 us per step, tools/ab_step.py) -- hb_debug_set("sgp_form16", 1) selects it
  const bool form16 = a.M >= 384 && a.M % 32 == 0 && E * hb_cdiv(a.n, SGP_SN) <= cus + cus / 4 && hb_debug_get("sgp_form16", 0) != 0;
#define HB_STRIP(D_)                                                                                          \
  do {                                                                                                        \
    if (a.W3)                                                                                                 \
      hipLaunchKernelGGL((sgp_A_strip3_kernel<D_>), grid, dim3(SGP_STRIP_THREADS), 0, stream, a);             \
    else if (a.Wf && form3 && form16)                                                                         \
      hipLaunchKernelGGL((sgp_A_strip16_kernel<D_>), grid, dim3(SGP_S16_THREADS), 0, stream, a);              \
    else if (a.Wf && form3)                                                                                   \
      hipLaunchKernelGGL((sgp_A_strip2t_kernel<D_>), grid, dim3(SGP_STRIP_THREADS), 0, stream, a);            \
    else if (a.Wf)                                                                                            \
      hipLaunchKernelGGL((sgp_A_strip2_kernel<D_>), grid, dim3(SGP_STRIP_THREADS), 0, stream, a);             \
    else                                                                                                      \
      hipLaunchKernelGGL((sgp_A_strip_kernel<D_, false>), grid, dim3(SGP_STRIP_THREADS), 0, stream, a);       \
  } while (0)
  if (a.d == 1)
    HB_STRIP(1);
  else if (a.d == 2)
    HB_STRIP(2);
  else if (a.d == 3)
    HB_STRIP(3);
  else
    HB_STRIP(4);
#undef HB_STRIP
  HB_LAUNCH_CHECK();
  return 0;
}
static int sgp_A_strip_launch(SgpArgs<double>, long, hipStream_t) { return -1; }  // fp32 only

// D dispatch: z staged in LDS when d <= SGP_DREG; vector operand path when, in
// addition, M is a multiple of 16 and W is 16-byte aligned.
template <typename T>
static int sgp_A_launch(const SgpArgs<T>& a, dim3 grid, hipStream_t stream) {
  const bool lds_z = a.d <= SGP_DREG && a.M * a.d <= SGP_ZS_MAX;
  const bool vec = lds_z && a.M % 16 == 0 && ((uintptr_t)a.W % 16 == 0);
#define HB_SGP_A(D_)                                                                          \
  do {                                                                                        \
    if (vec)                                                                                  \
      hipLaunchKernelGGL((sgp_A_kernel<T, D_, true>), grid, dim3(256), 0, stream, a);         \
    else                                                                                      \
      hipLaunchKernelGGL((sgp_A_kernel<T, D_, false>), grid, dim3(256), 0, stream, a);        \
  } while (0)
  if (lds_z && a.d == 1)
    HB_SGP_A(1);
  else if (lds_z && a.d == 2)
    HB_SGP_A(2);
  else if (lds_z && a.d == 3)
    HB_SGP_A(3);
  else if (lds_z && a.d == 4)
    HB_SGP_A(4);
  else
    hipLaunchKernelGGL((sgp_A_kernel<T, 0, false>), grid, dim3(256), 0, stream, a);
#undef HB_SGP_A
  HB_LAUNCH_CHECK();
  return 0;
}

// per column j: mean_p = sum_m u_pm A_mj ; s = sum_m A_mj^2 ; v = 1 - s ;
// f_p = mean_p + sqrt|v| eps_j.   Block = 32 columns x 8 row groups; ONE pass over A
// accumulates s and up to 4 latent-function means (P > 4: further passes for the rest).
template <typename T>
__global__ void __launch_bounds__(256) sgp_finish_kernel(const T* __restrict__ A, const T* __restrict__ u,
                                                         const T* __restrict__ eps, T* __restrict__ f,
                                                         T* __restrict__ v, long n, long M, long P, int mode) {
  __shared__ T red[5][8][33];
  const long e = blockIdx.y;
  A += e * M * n;
  u += e * P * M;
  f += e * P * n;
  v += e * n;
  if (eps) eps += e * n;
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const long j = (long)blockIdx.x * 32 + cx;
  const bool ok = j < n;
  const long jc = ok ? j : n - 1;
  T scale = T(0);
  for (long p0 = 0; p0 < P || p0 == 0; p0 += 4) {
    const int np = (int)((P - p0) < 4 ? (P - p0) : 4);
    T sq = T(0), mac[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll 4
    for (long m = ry; m < M; m += 8) {
      const T av = A[m * n + jc];
      sq += av * av;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < np) mac[q] += u[(p0 + q) * M + m] * av;
    }
    red[4][ry][cx] = sq;
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][ry][cx] = mac[q];
    __syncthreads();
    if (ry == 0 && ok) {
      if (p0 == 0) {
        T ssum = T(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) ssum += red[4][q][cx];
        const T vv = T(1) - ssum;
        v[j] = vv;
        if (mode == HB_SGP_DIAGONAL) scale = hb_sqrt(hb_abs(vv)) * eps[j];
      }
      for (int q = 0; q < np; ++q) {
        T mean = T(0);
#pragma unroll
        for (int w = 0; w < 8; ++w) mean += red[q][w][cx];
        f[(p0 + q) * n + j] = mean + scale;
      }
    }
    __syncthreads();
  }
}

// f, v from the column partials written by sgp_A_kernel's epilogue; eps is drawn here (the same
// per-lane streams and pair order as sgp_rng_fill_kernel) or taken from eps_in.
// (sgp_finish_one / hb_sgp_finish_body: chain_bodies.cuh)
template <typename T>
__global__ void __launch_bounds__(256) sgp_finish_part_kernel(const T* __restrict__ part, int gy,
                                                              const T* __restrict__ eps_in, uint64_t* rng,
                                                              long nlanes, T* __restrict__ eps_out, T* __restrict__ f,
                                                              T* __restrict__ v, long total, long n, long P, int mode) {
  hb_sgp_finish_body<T>(part, gy, eps_in, rng, nlanes, eps_out, f, v, total, n, P, mode == HB_SGP_DIAGONAL,
                        (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

template <typename T>
__global__ void __launch_bounds__(256) sgp_rng_fill_kernel(uint64_t* state, long nlanes, T* out, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npairs = (n + 1) / 2;
  if (t >= nlanes || t >= npairs) return;
  HbRng g = rng_load(state, nlanes, t);
  for (long p = t; p < npairs; p += nlanes) {
    T z0, z1;
    g.normal2(z0, z1);
    out[2 * p] = (T)z0;
    if (2 * p + 1 < n) out[2 * p + 1] = (T)z1;
  }
  rng_store(state, nlanes, t, g);
}

extern "C" long hb_sgp_ws_elems(long E, long n, long M, long d, long P) {
  (void)P;
  const long gy = (M + SGP_BM - 1) / SGP_BM;  // most row blocks a column's statistics can be split over
  const long mm = 32 * E * M * M, part = 5 * E * gy * n;  // backward split-K slabs / forward column partials
  return E * n + E * M * d + (mm > part ? mm : part);
}

// paired grid when it already gives the chip ~1.5 workgroups per CU, else one row block per workgroup
static inline int sgp_grid_y(long E, long n, int nRB) {
  const long paired = (long)hb_cdiv(n, SGP_BN) * ((nRB + 1) / 2) * E;
  return (paired < 384 && nRB > 1) ? nRB : (nRB + 1) / 2;
}

// the likelihood head that may ride in the forward strip kernel (hb_sgp_fwd_gauss)
template <typename T>
struct SgpHead {
  const T* y = nullptr;
  const T* scale = nullptr;
  const T* var = nullptr;
  double post = 0.0;
  T* dmu = nullptr;
  T* fbar = nullptr;
  T* part = nullptr;
  long units = 0;
};
// can hb_sgp_fwd run its finishing pass (and a likelihood head) inside the third strip form for this call?
static inline bool sgp_fused_finish_ok(long E, long n, long M, long d, long P, int prec, bool has_wfrag, bool draw, long rng_lanes) {
  const bool nofuse = hb_debug_get("sgp_no_fused_finish", 0) != 0 || hb_debug_get("sgp_strip_form2", 0) != 0;   // (diagnostic)
  return !nofuse && has_wfrag && prec == HB_PREC_NATIVE && hb_sgp_strip_path(E, n, M, d, P, prec) && P <= 1 && n % 2 == 0 &&
         (!draw || rng_lanes >= (E * n + 1) / 2);
}
extern "C" long hb_sgp_head_units(long E, long n, long M, long d, long P, int prec, int has_wfrag, int draw, long rng_lanes) {
  if (P != 1 || !sgp_fused_finish_ok(E, n, M, d, P, prec, has_wfrag != 0, draw != 0, rng_lanes)) return 0;
  return E * hb_cdiv(n, SGP_SN);
}

template <typename T>
static int sgp_fwd(int kind, int mode, const T* x, long sx, const T* z, const T* ell, long dl, const T* W, const T* Wf,
                   int prec, const T* u, const T* eps_in, uint64_t* rng, long rng_lanes, T* eps_out, T* A, T* A_frag, T* f,
                   T* v, long E, long n, long M, long d, long P, T* ws, hipStream_t stream, const SgpHead<T>* head = nullptr) {
  HB_REQUIRE(!A_frag || (sizeof(T) == 4 && Wf && ws && hb_sgp_strip_path(E, n, M, d, P, prec)),
             "hb_sgp_fwd: a fragment-major A needs the column-strip form (fp32, Wfrag, hb_sgp_strip_path)");
  if (hb_chain_recording()) {
    const int crc = hb_chain_flush(stream);   // whatever was recorded before this call runs before its contraction
    if (crc) return crc;
  }
  HB_REQUIRE(prec == HB_PREC_NATIVE || prec == HB_PREC_BF16X3, "hb_sgp_fwd: unknown precision %d", prec);
  HB_REQUIRE(prec == HB_PREC_NATIVE || (sizeof(T) == 4 && Wf && M % 32 == 0 && M <= SGP_SM_MAX && d <= SGP_DREG && P <= 4 && ws),
             "hb_sgp_fwd: bf16x3 needs fp32, the bf16 images in Wfrag, M %% 32 == 0, M <= %d, d <= %d, P <= 4", SGP_SM_MAX, SGP_DREG);
  HB_REQUIRE(kind == HB_KERN_RBF, "hb_sgp_fwd: only the UnitRBF kernel is fused (kind=%d)", kind);
  HB_REQUIRE(mode == HB_SGP_NEGLECTED || mode == HB_SGP_DIAGONAL, "hb_sgp_fwd: unknown mode %d", mode);
  HB_REQUIRE(E >= 0 && n >= 0 && M >= 0 && d >= 1 && P >= 0, "hb_sgp_fwd: bad extents");
  HB_REQUIRE(dl == 1 || dl == d, "hb_sgp_fwd: lengthscales must have 1 or d entries");
  HB_REQUIRE(x && z && ell && W && u && (A || A_frag) && f && v, "hb_sgp_fwd: NULL pointer");
  HB_REQUIRE(E <= 65535, "hb_sgp_fwd: too many experts");
  HB_REQUIRE(M * n < 2147483647L && M * M < 2147483647L && n * d < 2147483647L, "hb_sgp_fwd: matrix too large for 32-bit indexing");
  if (E * n == 0) return 0;
  const bool draw = mode == HB_SGP_DIAGONAL && !eps_in;
  if (draw) HB_REQUIRE(rng && rng_lanes > 0 && eps_out, "hb_sgp_fwd: need eps_in, or rng and eps_out");
  HB_REQUIRE(!head || (M > 0 && P == 1 && ws && Wf), "hb_sgp_fwd_gauss: needs Wfrag, a workspace and P == 1");
  const int nRB = hb_cdiv(M, SGP_BM);
  const int gy = sgp_grid_y(E, n, nRB);
  // fused path: the contraction kernel leaves per-column partial sums, one small kernel finishes f, v (and draws eps)
  if (M > 0 && P <= 4 && ws) {
    SgpArgs<T> a;
    a.x = x; a.sx = sx; a.z = z; a.ell = ell; a.dl = dl; a.W = W; a.Wf = Wf; a.u = u; a.A = A;
    a.n = n; a.M = M; a.d = d; a.P = P;
    a.W3 = prec == HB_PREC_BF16X3 ? (const void*)(Wf + 2 * E * M * M) : nullptr;
    a.plane3 = E * M * M;
    a.Af = A_frag;
    a.part = ws + E * n + E * M * d;
    const bool strip = sizeof(T) == 4 && ((Wf && hb_sgp_strip_path(E, n, M, d, P, prec)) ||
                                          (sgp_strip_ok(E, n, M, d, W) && !hb_sgp_no_strip()));
    const int gyp = strip ? 1 : gy;  // partial rows of the column statistics
    int rc;
    a.fin = 0;
    if (strip) {
      // third strip form with one column mean: the finishing pass runs inside the contraction kernel
      if (sizeof(T) == 4 && sgp_fused_finish_ok(E, n, M, d, P, prec, a.Wf != nullptr && !a.W3, draw, rng_lanes)) {
        a.fin = 1;
        a.fin_diag = mode == HB_SGP_DIAGONAL;
        a.fin_eps_in = mode == HB_SGP_DIAGONAL ? eps_in : (const T*)nullptr;
        a.fin_rng = draw ? rng : (uint64_t*)nullptr;
        a.fin_lanes = rng_lanes;
        a.fin_eps_out = mode == HB_SGP_DIAGONAL ? eps_out : (T*)nullptr;
        a.fin_f = f;
        a.fin_v = v;
        if (head) {
          HB_REQUIRE(P == 1 && head->units == E * hb_cdiv(n, SGP_SN), "hb_sgp_fwd_gauss: units must be hb_sgp_head_units(...)");
          a.head = 1;
          a.hy = head->y, a.hscale = head->scale, a.hvar = head->var;
          a.hdmu = head->dmu, a.hfbar = head->fbar, a.hpost = (T)head->post;
          a.hpart = head->part, a.hunits = head->units;
        }
      }
      HB_REQUIRE(!head || a.fin, "hb_sgp_fwd_gauss: this call cannot carry the head (hb_sgp_head_units(...) == 0)");
      if constexpr (std::is_same<T, float>::value) {
        if (g_rider.armed) {
          // recorded, not launched: the factorisation that produces W / Wfrag launches it inside its own grid (early-start form)
          HB_REQUIRE(a.fin && !a.W3 && d <= SGP_E_DMAX, "hb_sgp_fwd after hb_sgp_rider_begin: this call cannot ride (hb_sgp_rider_supported)");
          sgp_grid(hb_cdiv(a.n, SGP_SN), 1, E, a.efast);
          g_rider.a = a, g_rider.E = E, g_rider.valid = true, g_rider.armed = false;
          return 0;
        }
      }
      rc = sgp_A_strip_launch(a, E, stream);
      if (rc) return rc;
      if (a.fin) return 0;
    } else {
      dim3 grid = sgp_grid(hb_cdiv(n, SGP_BN), gy, E, a.efast);
      rc = sgp_A_launch<T>(a, grid, stream);
    }
    if (rc) return rc;
    const long total = E * n;
    if (hb_chain_recording() && total <= HB_CHAIN_FINISH_MAX_N) {
      // the finishing pass opens a serial chain (the likelihood head and its elementwise cluster follow in the same launch)
      HbChainJob j;
      j.kind = HB_CHAIN_SGP_FINISH;
      j.is64 = sizeof(T) == 8;
      j.p[0] = a.part;
      j.p[1] = mode == HB_SGP_DIAGONAL ? eps_in : (const T*)nullptr;
      j.p[2] = draw ? rng : (uint64_t*)nullptr;
      j.p[3] = mode == HB_SGP_DIAGONAL ? eps_out : (T*)nullptr;
      j.p[4] = f;
      j.p[5] = v;
      j.l[0] = gyp, j.l[1] = rng_lanes, j.l[2] = total, j.l[3] = n, j.l[4] = P, j.l[5] = mode == HB_SGP_DIAGONAL;
      return hb_chain_push(j, stream);
    }
    const int fgrid = draw ? hb_cdiv(rng_lanes, 256) : hb_stream_grid((total + 1) / 2, 256);
    hipLaunchKernelGGL(sgp_finish_part_kernel<T>, dim3(fgrid), dim3(256), 0, stream, a.part, gyp,
                       mode == HB_SGP_DIAGONAL ? eps_in : (const T*)nullptr, draw ? rng : (uint64_t*)nullptr, rng_lanes,
                       mode == HB_SGP_DIAGONAL ? eps_out : (T*)nullptr, f, v, total, n, P, mode);
    HB_LAUNCH_CHECK();
    return 0;
  }
  const T* eps = eps_in;
  if (mode == HB_SGP_DIAGONAL) {
    if (draw) {
      hipLaunchKernelGGL(sgp_rng_fill_kernel<T>, dim3(hb_cdiv(rng_lanes, 256)), dim3(256), 0, stream, rng, rng_lanes,
                         eps_out, E * n);
      HB_LAUNCH_CHECK();
      eps = eps_out;
    } else if (eps_out && eps_out != eps_in) {
      HB_HIP(hb_copy_async(eps_out, eps_in, sizeof(T) * E * n, stream));
    }
  }
  if (M > 0) {
    SgpArgs<T> a;
    a.x = x; a.sx = sx; a.z = z; a.ell = ell; a.dl = dl; a.W = W; a.Wf = nullptr; a.u = u; a.A = A;
    a.n = n; a.M = M; a.d = d; a.P = P;
    a.W3 = nullptr; a.plane3 = 0; a.Af = nullptr;
    a.part = nullptr;
    dim3 grid = sgp_grid(hb_cdiv(n, SGP_BN), gy, E, a.efast);
    int rc = sgp_A_launch<T>(a, grid, stream);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(sgp_finish_kernel<T>, dim3(hb_cdiv(n, 32), (unsigned)E), dim3(256), 0, stream, A, u, eps, f, v, n,
                     M, P, mode);
  HB_LAUNCH_CHECK();
  return 0;
}

// A = W K(z,x) alone (posterior-prediction callers need A without a draw; also
// lets bench.py time the contraction kernel in isolation).
template <typename T>
static int sgp_A_only(int kind, const T* x, long sx, const T* z, const T* ell, long dl, const T* W, const T* Wf, int prec,
                      T* A, long E, long n, long M, long d, hipStream_t stream) {
  HB_REQUIRE(prec == HB_PREC_NATIVE || prec == HB_PREC_BF16X3, "hb_sgp_A: unknown precision %d", prec);
  HB_REQUIRE(prec == HB_PREC_NATIVE || (sizeof(T) == 4 && Wf && M % 32 == 0 && M <= SGP_SM_MAX && d <= SGP_DREG),
             "hb_sgp_A: bf16x3 needs fp32, the bf16 images in Wfrag, M %% 32 == 0, M <= %d, d <= %d", SGP_SM_MAX, SGP_DREG);
  HB_REQUIRE(kind == HB_KERN_RBF, "hb_sgp_A: only the UnitRBF kernel is fused (kind=%d)", kind);
  HB_REQUIRE(E >= 0 && n >= 0 && M >= 0 && d >= 1, "hb_sgp_A: bad extents");
  HB_REQUIRE(dl == 1 || dl == d, "hb_sgp_A: lengthscales must have 1 or d entries");
  HB_REQUIRE(x && z && ell && W && A, "hb_sgp_A: NULL pointer");
  HB_REQUIRE(E <= 65535, "hb_sgp_A: too many experts");
  HB_REQUIRE(M * n < 2147483647L && M * M < 2147483647L && n * d < 2147483647L, "hb_sgp_A: matrix too large");
  if (E * n * M == 0) return 0;
  SgpArgs<T> a;
  a.x = x; a.sx = sx; a.z = z; a.ell = ell; a.dl = dl; a.W = W; a.Wf = Wf; a.u = nullptr; a.A = A;
  a.n = n; a.M = M; a.d = d; a.P = 0;
  a.W3 = prec == HB_PREC_BF16X3 ? (const void*)(Wf + 2 * E * M * M) : nullptr;
  a.plane3 = E * M * M;
  a.Af = nullptr;
  a.part = nullptr;
  if (sizeof(T) == 4 && ((sgp_strip_ok(E, n, M, d, W) && !hb_sgp_no_strip()) || prec == HB_PREC_BF16X3))
    return sgp_A_strip_launch(a, E, stream);
  const int nRB = hb_cdiv(M, SGP_BM);
  dim3 grid = sgp_grid(hb_cdiv(n, SGP_BN), sgp_grid_y(E, n, nRB), E, a.efast);
  return sgp_A_launch<T>(a, grid, stream);
}
extern "C" int hb_sgp_A_f32(int kind, const float* x, long sx, const float* z, const float* ell, long dl,
                            const float* W, const float* Wfrag, int prec, float* A, long E, long n, long M, long d,
                            void* stream) {
  return sgp_A_only<float>(kind, x, sx, z, ell, dl, W, Wfrag, prec, A, E, n, M, d, (hipStream_t)stream);
}
extern "C" int hb_sgp_A_f64(int kind, const double* x, long sx, const double* z, const double* ell, long dl,
                            const double* W, const double* Wfrag, int prec, double* A, long E, long n, long M, long d,
                            void* stream) {
  return sgp_A_only<double>(kind, x, sx, z, ell, dl, W, Wfrag, prec, A, E, n, M, d, (hipStream_t)stream);
}

extern "C" int hb_sgp_fwd_f32(int kind, int mode, const float* x, long sx, const float* z, const float* ell, long dl,
                              const float* W, const float* Wfrag, int prec, const float* u, const float* eps_in,
                              uint64_t* rng, long rng_lanes, float* eps_out, float* A, float* A_frag, float* f, float* v,
                              long E, long n, long M, long d, long P, float* ws, void* stream) {
  return sgp_fwd<float>(kind, mode, x, sx, z, ell, dl, W, Wfrag, prec, u, eps_in, rng, rng_lanes, eps_out, A, A_frag, f, v,
                        E, n, M, d, P, ws, (hipStream_t)stream);
}
extern "C" int hb_sgp_fwd_f64(int kind, int mode, const double* x, long sx, const double* z, const double* ell,
                              long dl, const double* W, const double* Wfrag, int prec, const double* u,
                              const double* eps_in, uint64_t* rng, long rng_lanes, double* eps_out, double* A,
                              double* A_frag, double* f, double* v, long E, long n, long M, long d, long P, double* ws,
                              void* stream) {
  return sgp_fwd<double>(kind, mode, x, sx, z, ell, dl, W, Wfrag, prec, u, eps_in, rng, rng_lanes, eps_out, A, A_frag, f,
                         v, E, n, M, d, P, ws, (hipStream_t)stream);
}

// hb_sgp_fwd with the per-point part of the Gaussian likelihood head behind it inside the same launch (the forward strip
// kernel's finishing pass already holds f_j): only where hb_sgp_head_units(...) > 0.  The partial sums are folded by
// hb_gauss_ll_fold -- typically as a job of the step's last serial chain.
extern "C" int hb_sgp_fwd_gauss_f32(int kind, int mode, const float* x, long sx, const float* z, const float* ell, long dl,
                                    const float* W, const float* Wfrag, int prec, const float* u, const float* eps_in,
                                    uint64_t* rng, long rng_lanes, float* eps_out, float* A, float* A_frag, float* f, float* v,
                                    long E, long n, long M, long d, long P, float* ws, const float* y, const float* scale,
                                    const float* var, double post, float* dmu, float* fbar, float* head_part, long units,
                                    void* stream) {
  HB_REQUIRE(y && var && dmu && head_part && units > 0, "hb_sgp_fwd_gauss: NULL pointer");
  SgpHead<float> h;
  h.y = y, h.scale = scale, h.var = var, h.post = post, h.dmu = dmu, h.fbar = fbar, h.part = head_part, h.units = units;
  return sgp_fwd<float>(kind, mode, x, sx, z, ell, dl, W, Wfrag, prec, u, eps_in, rng, rng_lanes, eps_out, A, A_frag, f, v,
                        E, n, M, d, P, ws, (hipStream_t)stream, &h);
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// c_j = -eps_j sign(v_j)/sqrt|v_j| * sum_p fbar_pj   (0 for NEGLECTED): the chain through the
// diagonal residual  f += sqrt|v| eps,  v = 1 - colsum(A^2).
// d sqrt|v| / dv = sign(v) / (2 sqrt|v|) is 0/0 at v == 0 (reachable in fp32 when x sits on an
// inducing point); the reference has no guard there (gp/gp.py:131) and would propagate NaN --
// take the sub-gradient 0 instead.
template <typename T>
__device__ __forceinline__ T sgp_resid_coef(const T* __restrict__ eps, const T* __restrict__ v,
                                            const T* __restrict__ fbar, long n, long P, int mode, long j) {
  if (mode != HB_SGP_DIAGONAL) return T(0);
  T cs = T(0);
  for (long p = 0; p < P; ++p) cs += fbar[p * n + j];
  const T vv = v[j];
  const T av = hb_abs(vv);
  return av > T(0) ? -eps[j] * hb_sign(vv) / hb_sqrt(av) * cs : T(0);
}

template <typename T>
struct SgpBwdArgs {
  const T* W;     // [E, M, M]
  const T* u;     // [E, P, M]
  const T* A;     // [E, M, n]
  const T* fbar;  // [E, P, n]
  const T* eps;   // [E, n] (DIAGONAL mode)
  const T* v;     // [E, n]
  T* Kbar;        // [E, M, n]
  long n, M, P;
  int mode;
  long efast;     // expert-fastest grid (see sgp_block)
  // column-strip form (sgp_kbar_strip_kernel)
  const T* x;     // [E?, n, d]
  long sx;
  const T* z;     // [E, M, d]
  const T* ell;   // [E, dl]
  long dl, d;
  const T* WTf;   // fragment-major fp32 image of W^T
  const void* WT3;  // bf16x3 images of W^T (3 planes) or nullptr
  long plane3;
  T* part;        // [E, nS, 2d + P, M] strip partials of zbar, ell, ubar
  const T* Af;    // fragment-major A (written by the forward strip kernel), or nullptr: read A row-major
  T* Kf;          // fragment-major Kbar output for the Lbar contraction, or nullptr
};

template <typename T>
struct SgpRawB {
  T a, u;
};
template <typename T>
struct SgpRawB4 {
  typedef T VT __attribute__((ext_vector_type(16 / sizeof(T))));
  VT a;
  T u;
};

// Kbar = W^T Abar,  Abar_kj = sum_p u_pk fbar_pj + A_kj c_j  (built in the loader)
template <typename T, bool FAST>
__global__ void __launch_bounds__(256) sgp_kbar_kernel(SgpBwdArgs<T> a) {
  typedef TileGemm<T, SGP_BM, SGP_BN, 16, 2, 2> G;
  __shared__ __attribute__((aligned(16))) T lds[G::LDS_ELEMS];
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const T* W = a.W + e * a.M * a.M;
  const T* u = a.u + e * a.P * a.M;
  const T* A = a.A + e * a.M * a.n;
  const T* fbar = a.fbar + e * a.P * a.n;
  const T* eps = a.eps ? a.eps + e * a.n : nullptr;
  const T* vres = a.v + e * a.n;
  T* Kbar = a.Kbar + e * a.M * a.n;
  const int M = (int)a.M, n = (int)a.n;
  const int col0 = bx * SGP_BN;
  const int nRB = (M + SGP_BM - 1) / SGP_BM;
  const int jcol = col0 + (threadIdx.x % SGP_BN);
  const bool jok = jcol < n;
  const int jc = jok ? jcol : n - 1;
  const int Mm1 = M - 1;
  const T cj = sgp_resid_coef<T>(eps, vres, fbar, a.n, a.P, a.mode, jc);
  const bool preg = a.P == 1;
  const T fb0 = a.P > 0 ? fbar[jc] : T(0);
  // vector path: this thread's B-operand group is always the same VEC columns
  constexpr int VECK = TileGemm<T, SGP_BM, SGP_BN, 16, 2, 2>::VEC;
  const int n4 = col0 + (threadIdx.x % (SGP_BN / VECK)) * VECK;
  const int n4c = n4 + VECK <= n ? n4 : (n >= VECK ? n - VECK : 0);
  T c4[VECK], f4[VECK];
#pragma unroll
  for (int q = 0; q < VECK; ++q) {
    const bool ok = FAST && (n4 + q < n);
    c4[q] = ok ? sgp_resid_coef<T>(eps, vres, fbar, a.n, a.P, a.mode, n4 + q) : T(0);  // out of range: no contribution
    f4[q] = ok ? fbar[n4 + q] : T(0);
  }

  for (int half = 0; half < 2; ++half) {
    int rb;
    if (!sgp_row_block(half, nRB, rb)) break;
    const int row0 = rb * SGP_BM;
    G g;
    g.zero();
    auto la = [&](int m, int k) -> T {
      const int r = row0 + m;
      return W[k * M + (r < M ? r : Mm1)];
    };
    auto fa = [&](T raw, int m, int k) -> T {
      const int r = row0 + m;
      return ((r < M) & (k >= r)) ? raw : T(0);
    };
    auto lb = [&](int k, int nn) -> SgpRawB<T> {
      SgpRawB<T> r;
      r.a = A[k * n + jc];
      r.u = u[k];
      return r;
    };
    auto fb = [&](SgpRawB<T> raw, int k, int nn) -> T {
      T val = raw.a * cj;
      if (preg) {
        val += raw.u * fb0;
      } else {
        for (long p = 0; p < a.P; ++p) val += u[p * M + k] * fbar[p * n + jc];
      }
      return jok ? val : T(0);
    };
    if constexpr (FAST) {
      // vector path (M % 16 == 0, n % VEC == 0, P == 1): W^T and A/c/fbar by 16-byte loads
      typedef typename G::VT VT;
      constexpr int VEC = G::VEC;
      auto la4 = [&](int m, int k) -> VT {
        const int r = row0 + m;
        return *reinterpret_cast<const VT*>(&W[k * M + (r < M ? r : M - VEC)]);
      };
      auto fa4 = [&](VT raw, int m, int k) -> VT {
        VT v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = ((row0 + m + q < M) & (k >= row0 + m + q)) ? raw[q] : T(0);
        return v;
      };
      auto lb4 = [&](int k, int nn) -> SgpRawB4<T> {
        SgpRawB4<T> r;
        r.a = *reinterpret_cast<const typename SgpRawB4<T>::VT*>(&A[k * n + n4c]);
        r.u = u[k];
        return r;
      };
      auto fb4 = [&](SgpRawB4<T> raw, int k, int nn) -> VT {
        VT v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = raw.a[q] * c4[q] + raw.u * f4[q];
        return v;
      };
      g.template run_vec<HB_MC, HB_MC>(row0, M, la4, fa4, lb4, fb4, lds);
    } else {
      g.template run<false, false>(row0, M, la, fa, lb, fb, lds);
    }
    g.for_each([&](int row, int col, T v) {
      const int r = row0 + row, cc = col0 + col;
      if (r < M && cc < n) Kbar[(long)r * n + cc] = v;
    });
  }
}

// Sum over each 32-lane half of the wave with DPP adds (VALU only): afterwards every lane of rows 1 and 3 (lanes
// 16..31 and 48..63) holds the total of its half-wave.
__device__ __forceinline__ float hb_half_wave_sum_dpp(float v) {
  auto dpp = [](float x, int ctrl_unused) { return x; };
  (void)dpp;
  int vi;
#define HB_DPP_ADD(CTRL, ROWMASK)                                                                        \
  vi = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, true);             \
  v += __builtin_bit_cast(float, vi);
  HB_DPP_ADD(0xB1, 0xF)   // quad_perm [1,0,3,2]
  HB_DPP_ADD(0x4E, 0xF)   // quad_perm [2,3,0,1]
  HB_DPP_ADD(0x141, 0xF)  // row_half_mirror
  HB_DPP_ADD(0x140, 0xF)  // row_mirror: every lane of a 16-lane row now holds the row total
  HB_DPP_ADD(0x142, 0xA)  // row_bcast:15 -> rows 1 and 3 add the total of the row before
#undef HB_DPP_ADD
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward in column-strip form (fp32, fragment-major W^T from hb_cholesky_inverse; BF3: bf16x3 operands).
//
// One workgroup owns 32 data columns and all M rows, like the forward strip kernel: it builds
//   Abar[:, strip] = u^T fbar + A diag(c)            (into LDS, from the A columns of the strip)
//   Kbar[:, strip] = W^T Abar[:, strip]              (W^T is upper triangular: row tile t contracts chunks t .. nT-1)
// and, as each 32 x 32 tile of Kbar completes, folds it into the row gradients of the strip
//   zbar_m += sum_j Kbar_mj dK_mj/dz_m ,  ell_m += sum_j Kbar_mj dK_mj/dell ,  ubar_pm += sum_j fbar_pj A_mj
// (K re-synthesised from the staged coordinates; the tile is turned row-per-lane through a per-wave LDS transpose so
// a lane sums its 16 columns in registers).  That replaces sgp_kbar_kernel + sgp_rowgrad_vec_kernel, i.e. the second
// full read of Kbar and A (33.5 MB) disappears; the strips' partial sums are folded by sgp_strip_finish_kernel.
// ---------------------------------------------------------------------------------------------------------------
template <int D, bool BF3>
__global__ void __launch_bounds__(SGP_STRIP_THREADS, (BF3 || D >= 3) ? 2 : 4) sgp_kbar_strip_kernel(SgpBwdArgs<float> a) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef __bf16 B8 __attribute__((ext_vector_type(8)));
  typedef Mma<float> MM;
  constexpr int KS_BYTES = BF3 ? 3 * SGP_SN * SGP_S3LD * 2 : SGP_SN * SGP_SLD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char ks_raw[KS_BYTES];
  __shared__ float lred[(SGP_STRIP_THREADS / 64) * D];
  __shared__ __attribute__((aligned(16))) float zs[SGP_SM_MAX * D];
  __shared__ __attribute__((aligned(16))) float xss[SGP_SN * D];
  __shared__ float us[4][SGP_SM_MAX];
  __shared__ float cjs[SGP_SN], fbs[4][SGP_SN];
  float (*Ks)[SGP_SLD] = reinterpret_cast<float (*)[SGP_SLD]>(ks_raw);
  __bf16 (*K3)[SGP_SN][SGP_S3LD] = reinterpret_cast<__bf16 (*)[SGP_SN][SGP_S3LD]>(ks_raw);
  long e;
  int bx;
  sgp_block(a.efast, e, bx);
  const float* __restrict__ x = a.x + e * a.sx;
  const float* __restrict__ z = a.z + e * a.M * D;
  const float* __restrict__ ell = a.ell + e * a.dl;
  const float* __restrict__ A = a.A ? a.A + e * a.M * a.n : nullptr;
  const float* __restrict__ fbar = a.fbar + e * a.P * a.n;
  float* __restrict__ Kbar = a.Kbar ? a.Kbar + e * a.M * a.n : nullptr;
  const int M = (int)a.M, n = (int)a.n, P = (int)a.P;
  const int col0 = bx * SGP_SN, nS = (n + SGP_SN - 1) / SGP_SN;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  const int nq = 2 * D + P;
  float* __restrict__ part = a.part + ((e * nS + bx) * (long)nq) * M;   // [2D + P][M] of this strip

  // ---- stage z, the strip's x (both raw: differences are scaled, not coordinates), u, the per-column residual coefficient and fbar
  float sc[D];
#pragma unroll
  for (int dd = 0; dd < D; ++dd) sc[dd] = float(SGP_EXP2_SCALE) / ell[a.dl == 1 ? 0 : dd];
  {
    constexpr int NTH = SGP_STRIP_THREADS;
    constexpr int ZIT = (SGP_SM_MAX * D) / NTH, UIT = SGP_SM_MAX / NTH;
    float zt[ZIT], ut[4][UIT];
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      zt[it] = z[i < M * D ? i : 0];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        ut[p][it] = p < P ? a.u[e * a.P * a.M + (long)p * M + (i < M ? i : 0)] : 0.f;
      }
    if (tid < SGP_SN) {
      const int cc = col0 + tid;
      const bool ok = cc < n;
      const int cj = ok ? cc : n - 1;
      cjs[tid] = ok ? sgp_resid_coef<float>(a.eps ? a.eps + e * a.n : nullptr, a.v + e * a.n, fbar, a.n, a.P, a.mode, cj) : 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) fbs[p][tid] = (ok && p < P) ? fbar[(long)p * n + cj] : 0.f;
#pragma unroll
      for (int dd = 0; dd < D; ++dd) xss[tid * D + dd] = x[cj * D + dd];
    }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int i = tid + NTH * it;
      if (i < M * D) zs[i] = zt[it];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int it = 0; it < UIT; ++it) {
        const int i = tid + NTH * it;
        if (p < P && i < M) us[p][i] = ut[p][it];
      }
  }
  __syncthreads();

  // ---- Abar[:, strip] -> LDS ([column][k]); ubar partials of the strip on the way.
  if (a.Af) {
    // fragment-major A: wave w takes the row tiles w, w + 8, ...; lane (li, h) ends up with A[32t + li][16h .. 16h+15]
    // (fp32: four contiguous-kilobyte loads per tile; bf16x3: the three planes, hi + mid + lo)
    const int nTp = M / 32;
    const long fplane = (long)a.plane3 / a.M * 32 * nS;   // elements per bf16 plane of a fragment-major operand
    for (int t = w; t < nTp; t += SGP_STRIP_THREADS / 64) {
      float avals[16];
      if (BF3) {
        const __bf16* blk = reinterpret_cast<const __bf16*>(a.Af) + ((((long)e * nTp + t) * nS + bx) << 10) + 8 * lane;
        B8 pl[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int q = 0; q < 2; ++q) pl[p][q] = *reinterpret_cast<const B8*>(blk + p * fplane + 512 * q);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 8; ++j) avals[8 * q + j] = ((float)pl[2][q][j] + (float)pl[1][q][j]) + (float)pl[0][q][j];
      } else {
        const float* blk = a.Af + ((((long)e * nTp + t) * nS + bx) << 10) + 4 * lane;
        V4 av[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) av[v] = *reinterpret_cast<const V4*>(blk + 256 * v);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) avals[4 * v + s2] = av[v][s2];
      }
      const int k = 32 * t + li;
      float uk[4], usum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 4; ++p) uk[p] = p < P ? us[p][k] : 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const V4 cj4 = *reinterpret_cast<const V4*>(&cjs[16 * h + 4 * v]);
        V4 fb4[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) fb4[p] = *reinterpret_cast<const V4*>(&fbs[p][16 * h + 4 * v]);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          const int c = 16 * h + 4 * v + s2;
          const float aval = avals[4 * v + s2];
          float val = aval * cj4[s2];
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (p < P) {
              val += uk[p] * fb4[p][s2];
              usum[p] += fb4[p][s2] * aval;
            }
          if (BF3) {
            const __bf16 b0 = (__bf16)val;
            const float r1 = val - (float)b0;
            const __bf16 b1 = (__bf16)r1;
            K3[0][c][k] = b0, K3[1][c][k] = b1, K3[2][c][k] = (__bf16)(r1 - (float)b1);
          } else {
            Ks[c][k] = val;
          }
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (p < P) {
          const float tsum = usum[p] + __shfl_xor(usum[p], 32);
          if (h == 0) part[(long)(2 * D + p) * M + k] = tsum;
        }
    }
  } else {
    // row-major A.  Thread (c, kq) takes groups of G consecutive rows k for its column c: the global loads of a wave
    // cover two 128-byte row segments, the LDS store is one conflict-free 16-byte vector per group.
    constexpr int G = BF3 ? 8 : 4;
    const int c = tid & 31, kq = tid >> 5;           // 16 row-group lanes
    const bool cok = col0 + c < n;
    const int cc = cok ? col0 + c : n - 1;
    const float cj = cjs[c];
    float fb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) fb[p] = fbs[p][c];
    constexpr int NG = SGP_SM_MAX / (16 * G);         // groups per thread (M = 512: 8 of 4, or 4 of 8)
    float av[NG][G];
#pragma unroll
    for (int it = 0; it < NG; ++it)
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int k = G * (kq + 16 * it) + i;
        av[it][i] = (k < M && cok) ? A[(long)k * n + cc] : 0.f;
      }
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int kb = G * (kq + 16 * it);
      if (kb < M) {                                   // (M % 32 == 0: a group is wholly inside or outside)
        float val[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
          val[i] = av[it][i] * cj;
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (p < P) val[i] += us[p][kb + i] * fb[p];
        }
        if (BF3) {
          B8 p0, p1, p2;
#pragma unroll
          for (int i = 0; i < G; ++i) {
            const __bf16 b0 = (__bf16)val[i];
            const float r1 = val[i] - (float)b0;
            const __bf16 b1 = (__bf16)r1;
            p0[i & 7] = b0, p1[i & 7] = b1, p2[i & 7] = (__bf16)(r1 - (float)b1);
          }
          *reinterpret_cast<B8*>(&K3[0][c][kb]) = p0;
          *reinterpret_cast<B8*>(&K3[1][c][kb]) = p1;
          *reinterpret_cast<B8*>(&K3[2][c][kb]) = p2;
        } else {
          V4 q;
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] = val[i & 3];
          *reinterpret_cast<V4*>(&Ks[c][kb]) = q;
        }
        // ubar_pk (strip) = sum over the strip's 32 columns of fbar_pc A_kc: the 32 lanes of a half-wave hold one k;
        // reduced with DPP adds (no LDS traffic); lanes 31 / 63 end up with the two half-wave totals
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p < P) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
              const float t = hb_half_wave_sum_dpp(fb[p] * av[it][i]);
              if (c == 31) part[(long)(2 * D + p) * M + kb + i] = t;
            }
          }
      }
    }
  }
  __syncthreads();

  // ---- tile-steps: W^T is upper triangular: row tile t contracts chunks Q = t .. nT-1
  const int nT = M / 32;
  const int ta = nT - 1 - w, tb = w;                 // shallow tile first (w + 1 chunks), then the deep one (nT - w)
  const int d0 = ta > tb ? w + 1 : 0;                // the middle tile of an odd count is taken once, as tb
  const int d1 = tb <= ta ? nT - w : 0;
  const int nts = d0 + d1;
  typename MM::Acc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int gc = col0 + li;

  auto tile_of = [&](int ts) { return ts < d0 ? ta : tb; };
  auto chunk_of = [&](int ts) { return ts < d0 ? ta + ts : tb + (ts - d0); };
  float lwave[D];
#pragma unroll
  for (int dd = 0; dd < D; ++dd) lwave[dd] = 0.f;

  // fold a finished tile into the outputs.  The tile product is computed TRANSPOSED (sgp_strip.cuh): `acc` holds row
  // li of the tile on the lane, so the row-per-lane view -- lane (li, h): row li, columns 16h .. 16h+15 -- is eight
  // permlane swaps away (the first form went through a per-wave LDS transpose: 16 stores + 4 loads per tile and 36.8 KB
  // of LDS that kept a second workgroup off the CU)
  auto retire = [&](int tile) {
    sgp_acc_t_settle(acc);
    float kb[16];
    sgp_acc_t_rows(acc, kb);
    if (col0 + SGP_SN > n) {   // (only the last strip has columns past n: the masks stay off the vector pipe elsewhere)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (col0 + 16 * h + i >= n) kb[i] = 0.f;
    }
    if (a.Kbar) {   // row-major Kbar (not requested by the fragment-major pipeline): 64 contiguous bytes per lane
      float* kp = Kbar + (long)(32 * tile + li) * n + col0 + 16 * h;
      if ((n & 3) == 0 && col0 + SGP_SN <= n) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          V4 q;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) q[s2] = kb[4 * v + s2];
          *reinterpret_cast<V4*>(kp + 4 * v) = q;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (col0 + 16 * h + i < n) kp[i] = kb[i];
      }
    }
    if (a.Kf && !BF3) sgp_store_frag_rows(a.Kf, kb, e, nT, nS, tile, bx, col0, n, lane);   // one contiguous KB per store
    if (BF3 && a.Kf)   // ... as three bf16 planes for the bf16x3 Lbar contraction
      sgp_store_frag3_row(reinterpret_cast<__bf16*>(a.Kf), (long)a.plane3 / a.M * 32 * nS, kb, e, nT, nS, tile, bx, lane);
    const int row = 32 * tile + li;
    float zr[D], zacc[D], lacc[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) zr[dd] = zs[row * D + dd], zacc[dd] = 0.f, lacc[dd] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = 16 * h + i;
      float tt[D], r2 = 0.f;
#pragma unroll
      for (int dd = 0; dd < D; ++dd) {
        tt[dd] = (zr[dd] - xss[c * D + dd]) * sc[dd];   // difference first, then the exp2 scale (coordinates staged raw)
        r2 += tt[dd] * tt[dd];
      }
      const float gk = kb[i] * hb_exp2_neg<float>(r2);   // (columns past n hold zeros)
#pragma unroll
      for (int dd = 0; dd < D; ++dd) {
        zacc[dd] -= gk * tt[dd];
        lacc[dd] += gk * tt[dd] * tt[dd];
      }
    }
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      zacc[dd] += __shfl_xor(zacc[dd], 32);
      // tt is in exp2-scaled units (t' = t * S, S = SGP_EXP2_SCALE): zbar = -(1/ell) sum g t, ell = (1/ell) sum g t^2
      if (h == 0) part[(long)dd * M + row] = zacc[dd] * sc[dd] * float(1.0 / (SGP_EXP2_SCALE * SGP_EXP2_SCALE));
      lwave[dd] += lacc[dd];   // the lengthscale gradient is summed over the rows too: per lane, folded at the end
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };

  if (!BF3) {
    const float* __restrict__ Wf = a.WTf + e * a.M * a.M;
    auto load = [&](V4 (&f)[4], int ts) {
      const int tc = ts < nts ? ts : nts - 1;
      const float* p = Wf + ((long)(tile_of(tc) * nT + chunk_of(tc)) << 10) + 4 * lane;
#pragma unroll
      for (int v = 0; v < 4; ++v) f[v] = *reinterpret_cast<const V4*>(p + 256 * v);
    };
    auto compute = [&](const V4 (&f)[4], int ts) {
      if (ts >= nts) return;
      const int Q = chunk_of(ts);
      V4 bv[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) bv[v] = *reinterpret_cast<const V4*>(&Ks[li][32 * Q + 16 * h + 4 * v]);
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) acc = MM::mma(bv[v][s2], f[v][s2], acc);   // transposed tile
      if (ts == d0 - 1 || ts == nts - 1) retire(tile_of(ts));
    };
    if (nts > 0) {
      V4 fa[4], fb2[4];
      load(fa, 0);
#pragma nounroll
      for (int ts = 0; ts < nts; ts += 2) {
        load(fb2, ts + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(fa, ts);
        __builtin_amdgcn_sched_barrier(0);
        load(fa, ts + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(fb2, ts + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    const __bf16* __restrict__ W3 = reinterpret_cast<const __bf16*>(a.WT3) + e * a.M * a.M;
    struct Frag {
      B8 f[3][2];
    };
    auto load = [&](Frag& fr, int ts) {
      const int tc = ts < nts ? ts : nts - 1;
      const __bf16* p = W3 + ((long)(tile_of(tc) * nT + chunk_of(tc)) << 10) + 8 * lane;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) fr.f[t][q] = *reinterpret_cast<const B8*>(p + (long)t * a.plane3 + 512 * q);
    };
    auto compute = [&](const Frag& fr, int ts) {
      if (ts >= nts) return;
      const int Q = chunk_of(ts);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        B8 b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const B8*>(&K3[t][li][32 * Q + 16 * q + 8 * h]);
        // (operands swapped against the forward: the tile comes out transposed, see retire)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], fr.f[2][q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2], fr.f[0][q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], fr.f[1][q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], fr.f[1][q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], fr.f[0][q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], fr.f[0][q], acc, 0, 0, 0);
      }
      if (ts == d0 - 1 || ts == nts - 1) retire(tile_of(ts));
    };
    if (nts > 0) {
      Frag fa, fb2;
      load(fa, 0);
#pragma nounroll
      for (int ts = 0; ts < nts; ts += 2) {
        load(fb2, ts + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(fa, ts);
        __builtin_amdgcn_sched_barrier(0);
        load(fa, ts + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(fb2, ts + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // ---- the strip's lengthscale partial: sum over all rows (lanes, waves) in a fixed order
  __syncthreads();
  float* red = lred;
#pragma unroll
  for (int dd = 0; dd < D; ++dd) {
    const float t = wave_sum(lwave[dd]);
    if (lane == 0) red[w * D + dd] = t;
  }
  __syncthreads();
#pragma unroll
  for (int dd = 0; dd < D; ++dd)
    if (tid == dd) {
      float t = 0.f;
      for (int ww = 0; ww < SGP_STRIP_THREADS / 64; ++ww) t += red[ww * D + dd];
      part[(long)(D + dd) * M] = t * sc[dd] * float(1.0 / (SGP_EXP2_SCALE * SGP_EXP2_SCALE * SGP_EXP2_SCALE));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Lbar = -tril(Kbar A^T) from the FRAGMENT-MAJOR Kbar and A that the strip kernels leave (contraction over the data
// axis, M^2 n flops on the lower tiles).  Workgroup = one 32 x 32 output tile (t >= t') x one slab of 32 strips; its
// eight waves take four strips each -- all eight operand loads of a wave are issued before its 64 MFMAs, and every
// load is one contiguous kilobyte -- and meet in LDS; the workgroup writes ONE partial tile.  sgp_lbar_finish_kernel
// sums the slabs (fixed order), negates and clears the upper triangle.  Replaces the LDS-staged split-K GEMM over
// the row-major operands (36.8 + 5.4 us at cfg 2, 33 % of its LDS cycles bank conflicts).
// ---------------------------------------------------------------------------------------------------------------
#ifndef HB_LSTAMP
#define HB_LSTAMP(i)
#endif
// Work split.  The operands are re-read once per output tile they meet, and at 32 x 32 tiles per wave that traffic
// (285 MB out of L2 at cfg 2 for 33.5 MB of operands) -- not the MFMAs -- set the time (32-41 us in two variants).
// A WAVE therefore owns a 64 x 64 block (2 x 2 tiles, four accumulators): per strip it loads two Kbar and two A
// fragment sets (16 contiguous-kilobyte loads) for 64 MFMAs, halving the bytes per flop; the four waves of a
// workgroup split a slab of strips and meet in LDS; S slabs give ~2 waves per SIMD.  On the diagonal blocks the
// tile above the diagonal is skipped.
template <bool BF3>
__global__ void __launch_bounds__(256) sgp_lbar_frag_kernel(const float* __restrict__ Kf, const float* __restrict__ Af,
                                                            float* __restrict__ slabs, int M, int nS, int S, long E,
                                                            int pairs) {
  typedef __bf16 B8 __attribute__((ext_vector_type(8)));
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  __shared__ float red[4][32][33];   // one tile at a time (16.5 KB: several workgroups per CU)
  const int nT = M / 32, nB = (nT + 1) / 2;             // 64-row blocks (the last one may hold a single tile)
  // XCD-aware unit order: workgroup w runs on XCD w % 8 (round-robin dispatch), each XCD has its own L2, and the
  // units that share operand strips are the `pairs` blocks of one (expert, slab).  The units are therefore laid out
  // (expert, slab)-major and XCD x takes the CONTIGUOUS range [x U/8, (x+1) U/8): an XCD works inside one or two slabs
  // and re-reads of a strip hit its L2 (with slab = w % S every XCD streamed every slab: 151 MB out of L2 per launch
  // at cfg 2 for 33.5 MB of operands, profiles/r02_pmc_cfg2_kernels.txt).
  const int U = gridDim.x, xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int uq = U >> 3, ur = U & 7;
  const int unit = xcd * uq + (xcd < ur ? xcd : ur) + kx;
  const int group = unit / pairs, pair = unit - group * pairs;
  const int slab = group % S;
  const long e = group / S;
  int bi = (int)((sqrtf(8.f * (float)pair + 1.f) - 1.f) * 0.5f);
  while (bi * (bi + 1) / 2 > pair) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= pair) ++bi;
  const int bj = pair - bi * (bi + 1) / 2;
  (void)nB;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int base = nS / S, rem = nS % S;
  const int sb = slab * base + (slab < rem ? slab : rem), se = sb + base + (slab < rem ? 1 : 0);
  const int cnt = se - sb, wq = cnt / 4, wr = cnt % 4;
  const int s0 = sb + w * wq + (w < wr ? w : wr), s1 = s0 + wq + (w < wr ? 1 : 0);
  // tiles of the block: rows 2bi, 2bi+1 (the second may not exist), columns 2bj, 2bj+1
  const int ti0 = 2 * bi, ti1 = 2 * bi + 1 < nT ? 2 * bi + 1 : ti0, tj0 = 2 * bj, tj1 = 2 * bj + 1 < nT ? 2 * bj + 1 : tj0;
  const bool has_i1 = 2 * bi + 1 < nT, has_j1 = 2 * bj + 1 < nT;
  const bool upper01 = bi == bj;                         // tile (ti0, tj1) lies above the diagonal
  const long tstride = (long)nS << 10;
  const float* __restrict__ kbase = Kf + (long)e * nT * tstride + 4 * lane;
  const float* __restrict__ abase = Af + (long)e * nT * tstride + 4 * lane;
  HB_LSTAMP(0);
  typename MM::Acc acc[4];                               // [2 * row + col]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  if (!BF3) {
    struct Frag {
      V4 a0[4], a1[4], b0[4], b1[4];
    };
    auto load = [&](Frag& f, int sidx) {
      const int sc = sidx < s1 ? sidx : (s1 > s0 ? s1 - 1 : s0);
      const long off = (long)sc << 10;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        f.a0[v] = *reinterpret_cast<const V4*>(kbase + ti0 * tstride + off + 256 * v);
        f.a1[v] = *reinterpret_cast<const V4*>(kbase + ti1 * tstride + off + 256 * v);
        f.b0[v] = *reinterpret_cast<const V4*>(abase + tj0 * tstride + off + 256 * v);
        f.b1[v] = *reinterpret_cast<const V4*>(abase + tj1 * tstride + off + 256 * v);
      }
    };
    auto compute = [&](const Frag& f, int sidx) {
      if (sidx >= s1) return;  // uniform
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          acc[0] = MM::mma(f.a0[v][s2], f.b0[v][s2], acc[0]);
          acc[2] = MM::mma(f.a1[v][s2], f.b0[v][s2], acc[2]);
          acc[3] = MM::mma(f.a1[v][s2], f.b1[v][s2], acc[3]);
        }
      if (!upper01) {   // (uniform; kept out of the MFMA stream above)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) acc[1] = MM::mma(f.a0[v][s2], f.b1[v][s2], acc[1]);
      }
    };
    if (s1 > s0) {
      Frag fa, fb;
      load(fa, s0);
#pragma nounroll
      for (int sidx = s0; sidx < s1; sidx += 2) {
        load(fb, sidx + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(fa, sidx);
        __builtin_amdgcn_sched_barrier(0);
        load(fa, sidx + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(fb, sidx + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    // bf16x3 operands: three bf16 planes per operand (written by the strip kernels), six cross products per k16-step
    const long plane = (long)E * nT * tstride;
    const __bf16* __restrict__ k3 = reinterpret_cast<const __bf16*>(Kf) + (long)e * nT * tstride + 8 * lane;
    const __bf16* __restrict__ a3 = reinterpret_cast<const __bf16*>(Af) + (long)e * nT * tstride + 8 * lane;
    struct Frag3 {
      B8 a0[3][2], a1[3][2], b0[3][2], b1[3][2];
    };
    auto load = [&](Frag3& f, int sidx) {
      const int sc = sidx < s1 ? sidx : (s1 > s0 ? s1 - 1 : s0);
      const long off = (long)sc << 10;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f.a0[p][q] = *reinterpret_cast<const B8*>(k3 + p * plane + ti0 * tstride + off + 512 * q);
          f.a1[p][q] = *reinterpret_cast<const B8*>(k3 + p * plane + ti1 * tstride + off + 512 * q);
          f.b0[p][q] = *reinterpret_cast<const B8*>(a3 + p * plane + tj0 * tstride + off + 512 * q);
          f.b1[p][q] = *reinterpret_cast<const B8*>(a3 + p * plane + tj1 * tstride + off + 512 * q);
        }
    };
    auto six = [&](const B8 (&x)[3][2], const B8 (&y)[3][2], int q, typename MM::Acc c) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2][q], y[0][q], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0][q], y[2][q], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1][q], y[1][q], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1][q], y[0][q], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0][q], y[1][q], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0][q], y[0][q], c, 0, 0, 0);
      return c;
    };
    auto compute = [&](const Frag3& f, int sidx) {
      if (sidx >= s1) return;  // uniform
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        acc[0] = six(f.a0, f.b0, q, acc[0]);
        acc[2] = six(f.a1, f.b0, q, acc[2]);
        acc[3] = six(f.a1, f.b1, q, acc[3]);
        if (!upper01) acc[1] = six(f.a0, f.b1, q, acc[1]);
      }
    };
    if (s1 > s0) {
      Frag3 fa, fb;
      load(fa, s0);
#pragma nounroll
      for (int sidx = s0; sidx < s1; sidx += 2) {
        load(fb, sidx + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(fa, sidx);
        __builtin_amdgcn_sched_barrier(0);
        load(fa, sidx + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(fb, sidx + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  HB_LSTAMP(1);
  float* out = slabs + (((long)slab * E + e) * M) * M;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ti = (q >> 1) ? ti1 : ti0, tj = (q & 1) ? tj1 : tj0;
    const bool live = ((q >> 1) == 0 || has_i1) && ((q & 1) == 0 || has_j1) && !(q == 1 && upper01);
    if (!live) continue;  // uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][MM::acc_row(lane, r)][lane & 31] = acc[q][r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + 256 * k, r = idx >> 5, c = idx & 31;
      const float v = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
      out[(long)(32 * ti + r) * M + 32 * tj + c] = v;
    }
    __syncthreads();
  }
  HB_LSTAMP(2);
}

// ---------------------------------------------------------------------------------------------------------------
// Lbar for MANY blocks (experts x long minibatches; round 3): a 128 x 128 block per workgroup, operand tiles through LDS.
//
// With a 64 x 64 block per wave (kernel above) every wave fetches its own two Kbar and two A tiles of each strip: 16
// flop per byte out of L2.  At cfg-5 size the units of a slab no longer run side by side, 76 % of those fetches miss
// the L2 and the launch runs at the memory system's rate -- 7.4 GB fetched for 2.1 GB of operands, 4.5 TB/s, 1.64 ms
// (profiles/r03_pmc_cfg5_contractions.txt).  Here the four waves of a workgroup own the four 64 x 64 quarters of ONE
// 128 x 128 block and share its eight operand tiles of a strip (4 of Kbar, 4 of A: 32 KB), staged once per workgroup in
// LDS -- double-buffered, one barrier per strip, fragment-major tiles copied as they are (contiguous kilobytes in,
// conflict-free 16-byte LDS accesses both ways): half the bytes per flop.  Every unit (expert, slab, block) is resident
// at once (two workgroups per CU), each wave writes its partial quarter itself; sgp_bwd_finish_kernel folds the slabs
// as before.  fp32 operands only.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) sgp_lbar_lds_kernel(const float* __restrict__ Kf, const float* __restrict__ Af,
                                                              float* __restrict__ slabs, int M, int nS, int S, long E,
                                                              int pairs) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  typedef Mma<float> MM;
  __shared__ __attribute__((aligned(16))) float buf[2][8][1024];   // [stage][4 Kbar tiles, 4 A tiles][fragment-major tile]
  const int nT = M / 32;
  // XCD-aware unit order as in sgp_lbar_frag_kernel: XCD x takes a contiguous range of (expert, slab)-major units
  const int U = gridDim.x, xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int uq = U >> 3, ur = U & 7;
  const int unit = xcd * uq + (xcd < ur ? xcd : ur) + kx;
  const int group = unit / pairs, pair = unit - group * pairs;
  const int slab = group % S;
  const long e = group / S;
  int bi = (int)((sqrtf(8.f * (float)pair + 1.f) - 1.f) * 0.5f);
  while (bi * (bi + 1) / 2 > pair) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= pair) ++bi;
  const int bj = pair - bi * (bi + 1) / 2;            // block (bi, bj), bj <= bi, 128 rows / columns each
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  const int base = nS / S, rem = nS % S;
  const int s0 = slab * base + (slab < rem ? slab : rem), s1 = s0 + base + (slab < rem ? 1 : 0);
  const long tstride = (long)nS << 10;
  const float* __restrict__ kbase = Kf + (long)e * nT * tstride;
  const float* __restrict__ abase = Af + (long)e * nT * tstride;
  // staging: V4 index q = i * 256 + tid covers the eight tiles (q >> 8 = tile, q & 255 = 16-byte group inside it)
  const float* gsrc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = i < 4 ? 4 * bi + i : 4 * bj + (i - 4);
    const int tc = t < nT ? t : nT - 1;                // (tiles past the matrix: a valid tile, never used)
    gsrc[i] = (i < 4 ? kbase : abase) + tc * tstride + 4 * tid;
  }
  // this wave's quarter: row tiles ti0, ti1 = 4 bi + 2 wr (+1), column tiles tj0, tj1 = 4 bj + 2 wc (+1)
  const int ti0 = 4 * bi + 2 * wr, tj0 = 4 * bj + 2 * wc;
  bool live[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ti = ti0 + (q >> 1), tj = tj0 + (q & 1);
    live[q] = ti < nT && tj < nT && tj <= ti;           // (tiles wholly above the diagonal are skipped)
  }
  const bool any_live = live[0] || live[1] || live[2] || live[3];
  typename MM::Acc acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  V4 st[8];
  auto gload = [&](int sidx) {
    const long off = (long)sidx << 10;
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = *reinterpret_cast<const V4*>(gsrc[i] + off);
  };
  auto lstore = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<V4*>(&buf[stage][i][4 * tid]) = st[i];
  };
  auto compute = [&](int stage) {
    if (!any_live) return;   // (uniform per wave)
    const float* k0 = &buf[stage][2 * wr][4 * lane];
    const float* k1 = &buf[stage][2 * wr + 1][4 * lane];
    const float* a0 = &buf[stage][4 + 2 * wc][4 * lane];
    const float* a1 = &buf[stage][4 + 2 * wc + 1][4 * lane];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const V4 fk0 = *reinterpret_cast<const V4*>(k0 + 256 * v), fk1 = *reinterpret_cast<const V4*>(k1 + 256 * v);
      const V4 fa0 = *reinterpret_cast<const V4*>(a0 + 256 * v), fa1 = *reinterpret_cast<const V4*>(a1 + 256 * v);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        if (live[0]) acc[0] = MM::mma(fk0[s2], fa0[s2], acc[0]);
        if (live[1]) acc[1] = MM::mma(fk0[s2], fa1[s2], acc[1]);
        if (live[2]) acc[2] = MM::mma(fk1[s2], fa0[s2], acc[2]);
        if (live[3]) acc[3] = MM::mma(fk1[s2], fa1[s2], acc[3]);
      }
    }
  };
  if (s1 > s0) {
    gload(s0);
    lstore(0);
    __syncthreads();
#pragma nounroll
    for (int sidx = s0; sidx < s1; ++sidx) {
      const int stage = (sidx - s0) & 1;
      const bool more = sidx + 1 < s1;
      if (more) gload(sidx + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(stage);
      __builtin_amdgcn_sched_barrier(0);
      if (more) lstore(stage ^ 1);
      __syncthreads();
    }
  }
  float* out = slabs + (((long)slab * E + e) * M) * M;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!live[q]) continue;
    const int ti = ti0 + (q >> 1), tj = tj0 + (q & 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long)(32 * ti + MM::acc_row(lane, r)) * M + 32 * tj + (lane & 31)] = acc[q][r];
  }
}

template <typename T>
__device__ __forceinline__ void sgp_lbar_finish_body(const T* __restrict__ slabs, int S, long E, long M, T* __restrict__ Lbar,
                                                     long vblock, long nvblocks) {
  const long total = E * M * M;
  const long stride = nvblocks * blockDim.x;
  for (long t = vblock * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long rem = t % (M * M);
    const long r = rem / M, c = rem - r * M;
    T acc = T(0);
    if (c <= r) {
      T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
      int s = 0;
      for (; s + 4 <= S; s += 4) {
        const T v0 = slabs[(long)s * total + t], v1 = slabs[(long)(s + 1) * total + t];
        const T v2 = slabs[(long)(s + 2) * total + t], v3 = slabs[(long)(s + 3) * total + t];
        a0 += v0, a1 += v1, a2 += v2, a3 += v3;
      }
      for (; s < S; ++s) a0 += slabs[(long)s * total + t];
      acc = -((a0 + a1) + (a2 + a3));
    }
    Lbar[t] = acc;
  }
}
template <typename T>
__global__ void __launch_bounds__(256) sgp_lbar_finish_kernel(const T* __restrict__ slabs, int S, long E, long M,
                                                              T* __restrict__ Lbar) {
  sgp_lbar_finish_body<T>(slabs, S, E, M, Lbar, blockIdx.x, gridDim.x);
}

// Fold the strips' partial sums (sgp_kbar_strip_kernel):  zbar[e, m, q] = sum_s part[e, s, q, m];
// ubar[e, p, m] = sum_s part[e, s, 2D + p, m];  ellbar[e, q | 0] = sum_s lstrip[e, s, q] (already summed over the rows
// by the strip kernel).  Block (x = 64-row block, y = quantity q, z = expert): 256 threads = 64 rows x 4 strip groups,
// 16 loads in flight per thread; the extra block row y = 2D + P - D... (see launch) sums the ell partials.  Fixed
// summation order: deterministic.
template <typename T>
__device__ __forceinline__ void sgp_strip_finish_body(const T* __restrict__ part, int nS, long M, long d, long dl, long P,
                                                      T* __restrict__ zbar, T* __restrict__ ellbar, T* __restrict__ ubar,
                                                      int bx, int q, long e, T (*red)[64], T* smem) {
  const int nq = (int)(2 * d + P);
  part += e * nS * (long)nq * M;
  if (q >= (int)d && q < 2 * (int)d) {
    // ell: per-strip totals were left in row 0 of this quantity's rows ([s][D + q][0]) by the strip kernel
    if (bx != 0) return;
    T acc = T(0);
    for (int s = threadIdx.x; s < nS; s += 256) acc += part[((long)s * nq + q) * M];
    acc = block_sum(acc, smem);
    // dl == 1: every dimension adds into the one lengthscale -- one block per dimension would race; dimension 0's
    // block therefore takes all of them
    if (dl == 1) {
      if (q != (int)d) return;
      T tot = acc;
      for (int qq = (int)d + 1; qq < 2 * (int)d; ++qq) {
        T a2 = T(0);
        for (int s = threadIdx.x; s < nS; s += 256) a2 += part[((long)s * nq + qq) * M];
        tot += block_sum(a2, smem);
      }
      if (threadIdx.x == 0) ellbar[e] = tot;
    } else if (threadIdx.x == 0) {
      ellbar[e * dl + (q - d)] = acc;
    }
    return;
  }
  const int ml = threadIdx.x & 63, sg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long m = (long)bx * 64 + ml;
  T acc = T(0);
  if (m < M) {
    const T* base = part + (long)q * M + m;
    const long sstride = (long)nq * M;
    int s = sg;
    for (; s + 4 * 15 < nS; s += 4 * 16) {
      T v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = base[(long)(s + 4 * i) * sstride];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += v[i];
    }
    for (; s < nS; s += 4) acc += base[(long)s * sstride];
  }
  red[sg][ml] = acc;
  __syncthreads();
  if (sg == 0 && m < M) {
    const T tot = (red[0][ml] + red[1][ml]) + (red[2][ml] + red[3][ml]);
    if (q < (int)d)
      zbar[(e * M + m) * d + q] = tot;
    else
      ubar[(e * P + (q - 2 * d)) * M + m] = tot;
  }
}
template <typename T>
__global__ void __launch_bounds__(256) sgp_strip_finish_kernel(const T* __restrict__ part, int nS, long M, long d, long dl,
                                                               long P, T* __restrict__ zbar, T* __restrict__ ellbar,
                                                               T* __restrict__ ubar) {
  __shared__ T red[4][64];
  __shared__ T smem[16];
  sgp_strip_finish_body<T>(part, nS, M, d, dl, P, zbar, ellbar, ubar, (int)blockIdx.x, (int)blockIdx.y, (long)blockIdx.z, red, smem);
}
// Both finishing passes of the fragment-major backward in ONE launch (they are independent: one folds the strip
// kernel's partial row gradients, the other the Lbar slabs): blocks [0, nb_strip) take the first job (flattened
// (64-row block, quantity, expert) index), the rest the second.
template <typename T>
__global__ void __launch_bounds__(256) sgp_bwd_finish_kernel(const T* __restrict__ part, int nS, long M, long d, long dl,
                                                             long P, T* __restrict__ zbar, T* __restrict__ ellbar,
                                                             T* __restrict__ ubar, int nb_strip, int nbx,
                                                             const T* __restrict__ slabs, int S, long E,
                                                             T* __restrict__ Lbar) {
  __shared__ T red[4][64];
  __shared__ T smem[16];
  if ((int)blockIdx.x < nb_strip) {
    const int nq = (int)(2 * d + P);
    const int vb = blockIdx.x;
    sgp_strip_finish_body<T>(part, nS, M, d, dl, P, zbar, ellbar, ubar, vb % nbx, (vb / nbx) % nq, (long)(vb / (nbx * nq)), red,
                             smem);
  } else {
    sgp_lbar_finish_body<T>(slabs, S, E, M, Lbar, (long)blockIdx.x - nb_strip, (long)gridDim.x - nb_strip);
  }
}

// One block per inducing row m:  zbar_m, ell partial, ubar_pm  (deterministic
// block reductions along the data axis).
//   dK_mj/dz_mq   = -K_mj (z_mq - x_jq)/ell_q^2
//   dK_mj/dell_q  =  K_mj (z_mq - x_jq)^2/ell_q^3
// The first pass over the row handles up to SGP_DREG input dims and up to 4 latent
// functions together (one read of Kbar[m,:] and A[m,:]); wider problems take more passes.
template <typename T>
__global__ void __launch_bounds__(256) sgp_rowgrad_kernel(const T* __restrict__ x, long sx, const T* __restrict__ z,
                                                          const T* __restrict__ ell, long dl,
                                                          const T* __restrict__ Kbar, const T* __restrict__ A,
                                                          const T* __restrict__ fbar, T* __restrict__ zbar,
                                                          T* __restrict__ ellpart, T* __restrict__ ubar, long n,
                                                          long M, long d, long P) {
  __shared__ T smem[16];
  const long e = blockIdx.y, m = blockIdx.x;
  x += e * sx;
  z += (e * M + m) * d;
  ell += e * dl;
  Kbar += (e * M + m) * n;
  A += (e * M + m) * n;
  fbar += e * P * n;
  const long npass_p = (P + 3) / 4, npass_q = (d + SGP_DREG - 1) / SGP_DREG;
  const long npass = npass_p > npass_q ? npass_p : npass_q;
  for (long pass = 0; pass < npass; ++pass) {
    const long p0 = pass * 4, q0 = pass * SGP_DREG;
    const int np = p0 < P ? (int)((P - p0) < 4 ? (P - p0) : 4) : 0;
    const int nq = q0 < d ? (int)((d - q0) < SGP_DREG ? (d - q0) : SGP_DREG) : 0;
    T uacc[4] = {T(0), T(0), T(0), T(0)};
    T zacc[SGP_DREG], lacc[SGP_DREG];
#pragma unroll
    for (int q = 0; q < SGP_DREG; ++q) zacc[q] = lacc[q] = T(0);
#pragma unroll 2
    for (long j = threadIdx.x; j < n; j += blockDim.x) {
      if (np > 0) {
        const T av = A[j];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < np) uacc[q] += fbar[(p0 + q) * n + j] * av;
      }
      if (nq > 0) {
        const T* xj = x + j * d;
        T r2 = T(0);
        for (long q = 0; q < d; ++q) {
          const T t = (z[q] - xj[q]) / ell[dl == 1 ? 0 : q];
          r2 += t * t;
        }
        const T gk = Kbar[j] * hb_exp(T(-0.5) * r2);
#pragma unroll
        for (int q = 0; q < SGP_DREG; ++q) {
          if (q < nq) {
            const T il = T(1) / ell[dl == 1 ? 0 : q0 + q];
            const T t = (z[q0 + q] - xj[q0 + q]) * il;
            zacc[q] += -gk * t * il;
            lacc[q] += gk * t * t * il;
          }
        }
      }
    }
    for (int q = 0; q < np; ++q) {
      const T us = block_sum(uacc[q], smem);
      if (threadIdx.x == 0) ubar[(e * P + p0 + q) * M + m] = us;
    }
    for (int q = 0; q < nq; ++q) {
      const T zs = block_sum(zacc[q], smem);
      const T ls = block_sum(lacc[q], smem);
      if (threadIdx.x == 0) {
        zbar[(e * M + m) * d + q0 + q] = zs;
        ellpart[(e * M + m) * d + q0 + q] = ls;
      }
    }
  }
}

// Vector form of the row pass (d <= SGP_DREG, P <= 4, n a multiple of the 16-byte group, aligned
// operands): each thread takes 16-byte groups of columns, the loop is unrolled so that the loads
// of four groups (A, Kbar, x, fbar) are in flight together -- with two waves per SIMD the scalar
// form above is bound by its dependent load round trips, not by bandwidth.
template <typename T, int D>
__global__ void __launch_bounds__(256) sgp_rowgrad_vec_kernel(const T* __restrict__ x, long sx,
                                                              const T* __restrict__ z, const T* __restrict__ ell,
                                                              long dl, const T* __restrict__ Kbar,
                                                              const T* __restrict__ A, const T* __restrict__ fbar,
                                                              T* __restrict__ zbar, T* __restrict__ ellpart,
                                                              T* __restrict__ ubar, long n, long M, long P) {
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef T VT __attribute__((ext_vector_type(VEC)));
  __shared__ T smem[16];
  const long e = blockIdx.y, m = blockIdx.x;
  x += e * sx;
  z += (e * M + m) * D;
  ell += e * dl;
  Kbar += (e * M + m) * n;
  A += (e * M + m) * n;
  fbar += e * P * n;
  const int np = (int)P;
  T zm[D], il[D];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    il[q] = T(1) / ell[dl == 1 ? 0 : q];
    zm[q] = z[q] * il[q];
  }
  T uacc[4] = {T(0), T(0), T(0), T(0)};
  T zacc[D], lacc[D];
#pragma unroll
  for (int q = 0; q < D; ++q) zacc[q] = lacc[q] = T(0);
  const long ngroups = n / VEC;
#pragma unroll 4
  for (long g = threadIdx.x; g < ngroups; g += 256) {
    const long j0 = g * VEC;
    const VT av = *reinterpret_cast<const VT*>(A + j0);
    const VT kv = *reinterpret_cast<const VT*>(Kbar + j0);
    T xq[VEC][D];
    if (D == 1) {
      const VT xv = *reinterpret_cast<const VT*>(x + j0);
#pragma unroll
      for (int c = 0; c < VEC; ++c) xq[c][0] = xv[c];
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c)
#pragma unroll
        for (int q = 0; q < D; ++q) xq[c][q] = x[(j0 + c) * D + q];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (p < np) {
        const VT fv = *reinterpret_cast<const VT*>(fbar + p * n + j0);
#pragma unroll
        for (int c = 0; c < VEC; ++c) uacc[p] += fv[c] * av[c];
      }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      T tt[D], r2 = T(0);
#pragma unroll
      for (int q = 0; q < D; ++q) {
        tt[q] = zm[q] - xq[c][q] * il[q];
        r2 += tt[q] * tt[q];
      }
      const T gk = kv[c] * hb_exp_fast<T>(T(-0.5) * r2);
#pragma unroll
      for (int q = 0; q < D; ++q) {
        zacc[q] -= gk * tt[q];
        lacc[q] += gk * tt[q] * tt[q];
      }
    }
  }
  for (int p = 0; p < np; ++p) {
    const T us = block_sum(uacc[p], smem);
    if (threadIdx.x == 0) ubar[(e * P + p) * M + m] = us;
  }
#pragma unroll
  for (int q = 0; q < D; ++q) {
    const T zs = block_sum(zacc[q], smem) * il[q];
    const T ls = block_sum(lacc[q], smem) * il[q];
    if (threadIdx.x == 0) {
      zbar[(e * M + m) * D + q] = zs;
      ellpart[(e * M + m) * D + q] = ls;
    }
  }
}

// ellbar[e, c] = sum_m ellpart[e, m, c] (all columns when dl == 1)
template <typename T>
__global__ void __launch_bounds__(256) sgp_ell_finish_kernel(const T* __restrict__ part, long M, long d, long dl,
                                                             T* __restrict__ ellbar) {
  __shared__ T smem[16];
  const long e = blockIdx.y, c = blockIdx.x;
  part += e * M * d;
  T acc = T(0);
  if (dl == 1) {
#pragma unroll 4
    for (long t = threadIdx.x; t < M * d; t += blockDim.x) acc += part[t];
  } else {
#pragma unroll 4
    for (long m = threadIdx.x; m < M; m += blockDim.x) acc += part[m * d + c];
  }
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) ellbar[e * dl + c] = acc;
}

// xbar_jq = sum_m Kbar_mj K_mj (z_mq - x_jq)/ell_q^2 : one thread per (j), loop m
template <typename T>
__global__ void __launch_bounds__(256) sgp_xbar_kernel(const T* __restrict__ x, long sx, const T* __restrict__ z,
                                                       const T* __restrict__ ell, long dl, const T* __restrict__ Kbar,
                                                       T* __restrict__ xbar, long n, long M, long d) {
  const long e = blockIdx.y;
  x += e * sx;
  z += e * M * d;
  ell += e * dl;
  Kbar += e * M * n;
  xbar += e * n * d;
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const T* xj = x + j * d;
  for (long q0 = 0; q0 < d; q0 += SGP_DREG) {
    T acc[SGP_DREG];
#pragma unroll
    for (int q = 0; q < SGP_DREG; ++q) acc[q] = T(0);
    for (long m = 0; m < M; ++m) {
      const T* zm = z + m * d;
      T r2 = T(0);
      for (long q = 0; q < d; ++q) {
        const T t = (zm[q] - xj[q]) / ell[dl == 1 ? 0 : q];
        r2 += t * t;
      }
      const T gk = Kbar[m * n + j] * hb_exp(T(-0.5) * r2);
#pragma unroll
      for (int q = 0; q < SGP_DREG; ++q) {
        if (q0 + q < d) {
          const T il = T(1) / ell[dl == 1 ? 0 : q0 + q];
          acc[q] += gk * (zm[q0 + q] - xj[q0 + q]) * il * il;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < SGP_DREG; ++q)
      if (q0 + q < d) xbar[j * d + q0 + q] = acc[q];
  }
}

static inline int sgp_matmul(const float* A, const float* B, float* C, long batch, long M, long N, long K, long lda,
                             long ldb, long ldc, long sA, long sB, long sC, int tA, int tB, double alpha, int flags,
                             float* ws, long wse, void* stream) {
  return hb_matmul_f32(A, B, C, batch, M, N, K, lda, ldb, ldc, sA, sB, sC, tA, tB, alpha, 0.0, nullptr, 0, HB_ACT_NONE,
                       flags, ws, wse, stream);
}
static inline int sgp_matmul(const double* A, const double* B, double* C, long batch, long M, long N, long K, long lda,
                             long ldb, long ldc, long sA, long sB, long sC, int tA, int tB, double alpha, int flags,
                             double* ws, long wse, void* stream) {
  return hb_matmul_f64(A, B, C, batch, M, N, K, lda, ldb, ldc, sA, sB, sC, tA, tB, alpha, 0.0, nullptr, 0, HB_ACT_NONE,
                       flags, ws, wse, stream);
}

static int sgp_bwd_strip_launch(SgpBwdArgs<float> a, long E, long nS, hipStream_t stream) {
  dim3 grid = sgp_grid(nS, 1, E, a.efast);
#define HB_KBS(D_)                                                                                               \
  do {                                                                                                           \
    if (a.WT3)                                                                                                   \
      hipLaunchKernelGGL((sgp_kbar_strip_kernel<D_, true>), grid, dim3(SGP_STRIP_THREADS), 0, stream, a);        \
    else                                                                                                         \
      hipLaunchKernelGGL((sgp_kbar_strip_kernel<D_, false>), grid, dim3(SGP_STRIP_THREADS), 0, stream, a);       \
  } while (0)
  if (a.d == 1)
    HB_KBS(1);
  else if (a.d == 2)
    HB_KBS(2);
  else if (a.d == 3)
    HB_KBS(3);
  else
    HB_KBS(4);
#undef HB_KBS
  HB_LAUNCH_CHECK();
  return 0;
}
static int sgp_bwd_strip_launch(SgpBwdArgs<double>, long, long, hipStream_t) { return -1; }  // fp32 only

// Launches the contraction only; *S_out slabs of E*M*M partial tiles are left at `slabs` for the finish pass.
static int sgp_lbar_frag_launch(const float* Kf, const float* Af, float* slabs, long slab_cap, long E, long M, long nS,
                                int prec, int* S_out, hipStream_t stream) {
  const int nT = (int)(M / 32);
  const long nB = (nT + 1) / 2, pairs = nB * (nB + 1) / 2;   // lower 64 x 64 blocks
  // ONE workgroup per CU (256 of them), i.e. one wave per SIMD: fp32 MFMAs occupy the SIMD's vector issue, so a
  // second wave's reduce / store phase crawls under its partner's MFMA stream (tools/lbar_stamps.hip: 20k cycles of
  // MFMAs followed by 20k cycles for a 2k-cycle epilogue), and fewer slabs mean fewer partial tiles to write and fold
  // (S = 7 at M = 512: 70.7 us for the whole backward against 75.7 at S = 14)
  long S = 256 / (pairs * E);
  if (S < 1) S = hb_cdiv(2048, pairs * E);   // more blocks than CUs (experts): ~8 workgroups per CU for balance
  if (S > nS / 8) S = nS / 8;
  if (S > slab_cap) S = slab_cap;   // the slabs live in what is left of the 32*E*M*M-element workspace
  if (S < 1) S = 1;
  {
    const long fs = hb_debug_get("lbar_force_s", 0);  // diagnostic
    if (fs) S = fs;
  }
  {
    // many blocks (experts x a long minibatch): 128 x 128 blocks with LDS-staged operand tiles, every unit resident
    const bool nolds = hb_debug_get("lbar_no_lds", 0) != 0;   // diagnostic
    const long nB2 = (nT + 3) / 4, pairs2 = nB2 * (nB2 + 1) / 2;
    if (prec != HB_PREC_BF16X3 && !nolds && pairs * E >= 256 && nS >= 64) {
      long S2 = 512 / (pairs2 * E);
      if (S2 > nS / 8) S2 = nS / 8;
      if (S2 > slab_cap) S2 = slab_cap;
      if (S2 >= 1) {
        dim3 grid2((unsigned)(pairs2 * S2 * E), 1, 1);
        hipLaunchKernelGGL(sgp_lbar_lds_kernel, grid2, dim3(256), 0, stream, Kf, Af, slabs, (int)M, (int)nS, (int)S2, E, (int)pairs2);
        HB_LAUNCH_CHECK();
        *S_out = (int)S2;
        return 0;
      }
    }
  }
  dim3 grid((unsigned)(pairs * S * E), 1, 1);
  if (prec == HB_PREC_BF16X3)
    hipLaunchKernelGGL(sgp_lbar_frag_kernel<true>, grid, dim3(256), 0, stream, Kf, Af, slabs, (int)M, (int)nS, (int)S, E,
                       (int)pairs);
  else
    hipLaunchKernelGGL(sgp_lbar_frag_kernel<false>, grid, dim3(256), 0, stream, Kf, Af, slabs, (int)M, (int)nS, (int)S, E,
                       (int)pairs);
  HB_LAUNCH_CHECK();
  *S_out = (int)S;
  return 0;
}
static int sgp_lbar_frag_launch(const double*, const double*, double*, long, long, long, long, int, int*, hipStream_t) { return -1; }

template <typename T>
static int sgp_bwd(int kind, int mode, const T* x, long sx, const T* z, const T* ell, long dl, const T* W,
                   const T* Wfrag, int prec, const T* u, const T* eps, const T* A, const T* A_frag, const T* v,
                   const T* fbar, T* Kbar, T* Kbar_frag, T* Lbar, T* ubar, T* zbar, T* ellbar, T* xbar, long E, long n,
                   long M, long d, long P, T* ws, hipStream_t stream) {
  HB_REQUIRE(prec == HB_PREC_NATIVE || prec == HB_PREC_BF16X3, "hb_sgp_bwd: unknown precision %d", prec);
  HB_REQUIRE(kind == HB_KERN_RBF, "hb_sgp_bwd: only the UnitRBF kernel is fused (kind=%d)", kind);
  HB_REQUIRE(mode == HB_SGP_NEGLECTED || mode == HB_SGP_DIAGONAL, "hb_sgp_bwd: unknown mode %d", mode);
  HB_REQUIRE(E >= 0 && n >= 0 && M >= 0 && d >= 1 && P >= 0, "hb_sgp_bwd: bad extents");
  HB_REQUIRE(dl == 1 || dl == d, "hb_sgp_bwd: lengthscales must have 1 or d entries");
  HB_REQUIRE(x && z && ell && W && u && (A || A_frag) && v && fbar && (Kbar || Kbar_frag) && Lbar && ubar && zbar && ellbar && ws,
             "hb_sgp_bwd: NULL pointer");
  HB_REQUIRE((A_frag != nullptr) == (Kbar_frag != nullptr), "hb_sgp_bwd: A_frag and Kbar_frag go together");
  HB_REQUIRE(mode == HB_SGP_NEGLECTED || eps, "hb_sgp_bwd: eps required for the diagonal mode");
  HB_REQUIRE(E <= 65535, "hb_sgp_bwd: too many experts");
  HB_REQUIRE(M * n < 2147483647L && M * M < 2147483647L && n * d < 2147483647L, "hb_sgp_bwd: matrix too large for 32-bit indexing");
  if (E * M == 0) return 0;
  T* ellpart = ws + E * n;  // (the first E*n elements of ws are unused since the residual coefficient is computed in-kernel)
  T* mmws = ellpart + E * M * d;
  const long mmws_elems = 32 * E * M * M;
  // column-strip form: fp32, fragment-major W^T available, M a multiple of 32 up to SGP_SM_MAX, small d and P, no xbar
  const long nS = hb_cdiv(n, SGP_SN);
  const bool strip = sizeof(T) == 4 && Wfrag && !xbar && hb_sgp_strip_path(E, n, M, d, P, prec) &&
                     nS * (2 * d + P) * M * E <= mmws_elems;
  HB_REQUIRE(!A_frag || strip, "hb_sgp_bwd: fragment-major A / Kbar need the column-strip form (see hb_sgp_strip_path)");
  HB_REQUIRE(strip || (A && Kbar), "hb_sgp_bwd: row-major A and Kbar required outside the column-strip form");
  HB_REQUIRE(prec == HB_PREC_NATIVE || strip, "hb_sgp_bwd: bf16x3 needs the column-strip form (fp32, Wfrag, M %% 32 == 0, M <= %d, "
             "d <= %d, P <= 4, no xbar)", SGP_SM_MAX, SGP_DREG);
  if (strip) {
    SgpBwdArgs<T> a;
    a.W = W; a.u = u; a.A = A; a.fbar = fbar; a.eps = eps; a.v = v; a.Kbar = Kbar;
    a.n = n; a.M = M; a.P = P; a.mode = mode;
    a.x = x; a.sx = sx; a.z = z; a.ell = ell; a.dl = dl; a.d = d;
    a.WTf = Wfrag + E * M * M;
    a.WT3 = prec == HB_PREC_BF16X3 ? (const void*)(reinterpret_cast<const unsigned short*>(Wfrag + 2 * E * M * M) + 3 * E * M * M)
                                   : nullptr;
    a.plane3 = E * M * M;
    a.part = mmws;   // consumed by the finish kernel before the Lbar contraction reuses the space
    a.Af = A_frag; a.Kf = Kbar_frag;
    if (A_frag) a.Kbar = nullptr;   // Kbar only feeds the Lbar contraction: the fragment-major copy is enough
    int rc = sgp_bwd_strip_launch(a, E, nS, stream);
    if (rc) return rc;
    if (A_frag) {
      // strip partials and Lbar slabs side by side in the workspace; ONE finish launch folds both
      const long npart = ((nS * (2 * d + P) * M * E + 63) / 64) * 64;
      const long slab_cap = (mmws_elems - npart) / (E * M * M);
      HB_REQUIRE(slab_cap >= 1, "hb_sgp_bwd: workspace too small for the Lbar slabs");
      int S = 0;
      rc = sgp_lbar_frag_launch(Kbar_frag, A_frag, mmws + npart, slab_cap > 32 ? 32 : slab_cap, E, M, nS, prec, &S, stream);
      if (rc) return rc;
      const int nbx = (int)hb_cdiv(M, 64), nb_strip = nbx * (int)(2 * d + P) * (int)E;
      hipLaunchKernelGGL(sgp_bwd_finish_kernel<T>, dim3((unsigned)(nb_strip + hb_stream_grid(E * M * M, 256))), dim3(256), 0, stream,
                         mmws, (int)nS, M, d, dl, P, zbar, ellbar, ubar, nb_strip, nbx, mmws + npart, S, E, Lbar);
      HB_LAUNCH_CHECK();
      return 0;
    }
    hipLaunchKernelGGL(sgp_strip_finish_kernel<T>, dim3((unsigned)hb_cdiv(M, 64), (unsigned)(2 * d + P), (unsigned)E), dim3(256), 0,
                       stream, mmws, (int)nS, M, d, dl, P, zbar, ellbar, ubar);
    HB_LAUNCH_CHECK();
    int rc2 = sgp_matmul(Kbar, A, Lbar, E, M, M, n, n, n, M, M * n, M * n, M * M, 0, 1, -1.0, HB_MM_TRIL_OUT, mmws, mmws_elems,
                         (void*)stream);
    return rc2;
  }
  if (n > 0) {
    SgpBwdArgs<T> a;
    a.W = W; a.u = u; a.A = A; a.fbar = fbar; a.eps = eps; a.v = v; a.Kbar = Kbar;
    a.n = n; a.M = M; a.P = P; a.mode = mode;
    a.x = nullptr; a.sx = 0; a.z = nullptr; a.ell = nullptr; a.dl = 0; a.d = d; a.WTf = nullptr; a.WT3 = nullptr;
    a.plane3 = 0; a.part = nullptr; a.Af = nullptr; a.Kf = nullptr;
    const int nRB = hb_cdiv(M, SGP_BM);
    dim3 grid = sgp_grid(hb_cdiv(n, SGP_BN), sgp_grid_y(E, n, nRB), E, a.efast);
    constexpr long VECH = 16 / sizeof(T);
    const bool vec = P == 1 && M % 16 == 0 && n % VECH == 0 && ((uintptr_t)W % 16 == 0) && ((uintptr_t)A % 16 == 0);
    if (vec)
      hipLaunchKernelGGL((sgp_kbar_kernel<T, true>), grid, dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((sgp_kbar_kernel<T, false>), grid, dim3(256), 0, stream, a);
    HB_LAUNCH_CHECK();
  }
  {
    constexpr long VECR = 16 / sizeof(T);
    const bool rvec = d <= SGP_DREG && P <= 4 && n > 0 && n % VECR == 0 && ((uintptr_t)A % 16 == 0) &&
                      ((uintptr_t)Kbar % 16 == 0) && ((uintptr_t)fbar % 16 == 0) && ((uintptr_t)x % 16 == 0) &&
                      (sx * (long)sizeof(T)) % 16 == 0;
    dim3 rgrid((unsigned)M, (unsigned)E);
#define HB_SGP_ROWGRAD(D_)                                                                                           \
  hipLaunchKernelGGL((sgp_rowgrad_vec_kernel<T, D_>), rgrid, dim3(256), 0, stream, x, sx, z, ell, dl, Kbar, A, fbar, \
                     zbar, ellpart, ubar, n, M, P)
    if (rvec && d == 1)
      HB_SGP_ROWGRAD(1);
    else if (rvec && d == 2)
      HB_SGP_ROWGRAD(2);
    else if (rvec && d == 3)
      HB_SGP_ROWGRAD(3);
    else if (rvec && d == 4)
      HB_SGP_ROWGRAD(4);
    else
      hipLaunchKernelGGL(sgp_rowgrad_kernel<T>, rgrid, dim3(256), 0, stream, x, sx, z, ell, dl, Kbar, A, fbar, zbar,
                         ellpart, ubar, n, M, d, P);
#undef HB_SGP_ROWGRAD
  }
  HB_LAUNCH_CHECK();
  hipLaunchKernelGGL(sgp_ell_finish_kernel<T>, dim3((unsigned)dl, (unsigned)E), dim3(256), 0, stream, ellpart, M, d, dl,
                     ellbar);
  HB_LAUNCH_CHECK();
  if (xbar && n > 0) {
    hipLaunchKernelGGL(sgp_xbar_kernel<T>, dim3(hb_cdiv(n, 256), (unsigned)E), dim3(256), 0, stream, x, sx, z, ell, dl,
                       Kbar, xbar, n, M, d);
    HB_LAUNCH_CHECK();
  }
  // Lbar = -tril(Kbar A^T): contraction over the data axis, split-K, lower tiles only
  if (n > 0) {
    int rc = sgp_matmul(Kbar, A, Lbar, E, M, M, n, n, n, M, M * n, M * n, M * M, 0, 1, -1.0, HB_MM_TRIL_OUT, mmws,
                        mmws_elems, (void*)stream);
    if (rc) return rc;
  } else {
    HB_HIP(hb_zero_async(Lbar, sizeof(T) * E * M * M, stream));
  }
  return 0;
}

extern "C" int hb_sgp_bwd_f32(int kind, int mode, const float* x, long sx, const float* z, const float* ell, long dl,
                              const float* W, const float* Wfrag, int prec, const float* u, const float* eps,
                              const float* A, const float* A_frag, const float* v, const float* fbar, float* Kbar,
                              float* Kbar_frag, float* Lbar, float* ubar, float* zbar, float* ellbar, float* xbar, long E,
                              long n, long M, long d, long P, float* ws, void* stream) {
  return sgp_bwd<float>(kind, mode, x, sx, z, ell, dl, W, Wfrag, prec, u, eps, A, A_frag, v, fbar, Kbar, Kbar_frag, Lbar,
                        ubar, zbar, ellbar, xbar, E, n, M, d, P, ws, (hipStream_t)stream);
}
extern "C" int hb_sgp_bwd_f64(int kind, int mode, const double* x, long sx, const double* z, const double* ell,
                              long dl, const double* W, const double* Wfrag, int prec, const double* u, const double* eps,
                              const double* A, const double* A_frag, const double* v, const double* fbar, double* Kbar,
                              double* Kbar_frag, double* Lbar, double* ubar, double* zbar, double* ellbar, double* xbar,
                              long E, long n, long M, long d, long P, double* ws, void* stream) {
  return sgp_bwd<double>(kind, mode, x, sx, z, ell, dl, W, Wfrag, prec, u, eps, A, A_frag, v, fbar, Kbar, Kbar_frag, Lbar,
                         ubar, zbar, ellbar, xbar, E, n, M, d, P, ws, (hipStream_t)stream);
}
