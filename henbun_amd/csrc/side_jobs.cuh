// Side jobs: small independent launches that ride on another kernel's launch as extra workgroups.
//
// A kernel boundary inside the captured step costs ~4.5 us even for a kernel that moves a few KB, and several of the
// step's small kernels are independent of everything around them: the minibatch index draw + row gather (needed only
// by the sparse-GP contraction, after the Cholesky chain), the reparameterised sample of q(u) + its KL (same), the
// sampler's VJP (needed by Adam only).  Their entry points have a `hb_side_push_*` twin that RECORDS the job instead
// of launching it; the next "host" launch of the same thread -- launch 0 of the Cholesky chain (8 workgroups on 256
// CUs), the in-workgroup split-K GEMM -- appends the recorded jobs' workgroups to its own grid (blocks past its own
// run hb_side_run).  hb_side_flush launches whatever is still pending as one kernel of its own, so deferring is
// always safe; the caller (graph.py) only defers a job whose outputs nobody reads before the host has run.
// fp32 only; up to HB_SIDE_MAX jobs per host launch.  The bodies below ARE the kernels of the stand-alone entry points
// (same code, the block index and grid size passed in).
#ifndef HB_SIDE_JOBS_CUH
#define HB_SIDE_JOBS_CUH
#include "common.cuh"
#include "rng_pairs.cuh"

// Several [nsrc, row_a] arrays gathered by one index vector (see elementwise.hip)
#define HB_GATHER_MAX 8
template <typename T>
struct GatherMultiArgs {
  const T* src[HB_GATHER_MAX];
  T* dst[HB_GATHER_MAX];
  long row[HB_GATHER_MAX];
  long start[HB_GATHER_MAX + 1];  // prefix sums of n*row[a]: element range of array a in the flattened work list
  int narr;
};

// minibatch index draw + gather: thread group of G lanes per output row r; its first lane owns RNG lane r
template <typename T>
__device__ __forceinline__ void gather_draw_body(const GatherMultiArgs<T>& g, uint64_t* __restrict__ state, long nlanes, long lo,
                                                 uint64_t range, long* __restrict__ idx_out, const long* __restrict__ perm,
                                                 long n, long nsrc, int* __restrict__ err, int Gs, long vblock) {
  const bool vec16 = Gs < 0;
  const int G = vec16 ? -Gs : Gs;
  const long t = vblock * 256 + threadIdx.x;
  const long r = t / G;
  const int sub = (int)(t % G);
  long j = 0;
  if (sub == 0 && r < n) {
    HbRng rg = rng_load(state, nlanes, r);
    j = lo + (long)__umul64hi(rg.next(), range);
    rng_store(state, nlanes, r, rg);
    idx_out[r] = j;
  }
  if (G > 1) {
    // the leader is lane (lane - sub) of the same wave (G divides 64)
    const int leader = (threadIdx.x & 63) - sub;
    const int jl = __shfl((int)(j & 0xffffffffL), leader), jh = __shfl((int)(j >> 32), leader);
    j = ((long)jh << 32) | (unsigned int)jl;
  }
  if (r >= n) return;
  if (perm) j = perm[j];
  const bool bad = j < 0 || j >= nsrc;
  if (bad && sub == 0 && err) *err = 1;
  // G < 0: every array has rows of whole 16-byte groups at 16-byte aligned addresses -- a lane moves 16 bytes per trip
  // (one float per lane made this launch instruction bound: 13.6 us for 2 x 8.4 MB at cfg 4 against 7.5 us for the plain
  // gather)
  if (vec16) {
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T VT __attribute__((ext_vector_type(VEC)));
    const VT zero = {};
#pragma unroll
    for (int a = 0; a < HB_GATHER_MAX; ++a) {
      if (a >= g.narr) break;
      const long w = g.row[a] / VEC;
      for (long c = sub; c < w; c += G)
        reinterpret_cast<VT*>(g.dst[a])[r * w + c] = bad ? zero : reinterpret_cast<const VT*>(g.src[a])[j * w + c];
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < HB_GATHER_MAX; ++a) {
    if (a >= g.narr) break;
    const long w = g.row[a];
    for (long c = sub; c < w; c += G) g.dst[a][r * w + c] = bad ? T(0) : g.src[a][j * w + c];
  }
}

// x = mu + exp(s) u, kl partial (variational.hip); `nvblocks` == 1: the block finishes kl itself
template <typename T>
__device__ __forceinline__ void diag_fwd_body(const T* __restrict__ mu, const T* __restrict__ s, const T* __restrict__ u_in,
                                              uint64_t* rng, long rng_lanes, T* __restrict__ u_out, T* __restrict__ x,
                                              T* __restrict__ partial, T* __restrict__ kl, long n, long L, long ldm, long lds_,
                                              long vblock, long nvblocks, T* smem) {
  // element i = (row i / L, column i % L); mu / s may be column blocks of a wider row-major matrix (row strides ldm,
  // lds_: the mean and log-std halves of an encoder output, read in place); x and u are dense
  const bool dense = ldm == L && lds_ == L;
  auto src = [&](long i, long ld) -> long {
    if (dense) return i;
    const long r = i / L;
    return r * ld + (i - r * L);
  };
  const long t = vblock * 256 + threadIdx.x;
  const long nthreads = rng ? rng_lanes : nvblocks * 256;
  const long npairs = (n + 1) / 2;
  T acc = T(0);
  const bool active = t < nthreads && t < npairs;
  HbRng g;
  if (rng && active) g = rng_load(rng, rng_lanes, t);
  if (active) {
    for (long p = t; p < npairs; p += nthreads) {
      const long i0 = 2 * p, i1 = 2 * p + 1;
      T u0, u1 = T(0);
      if (rng) {
        T z0, z1;
        g.normal2(z0, z1);
        u0 = (T)z0;
        u1 = (T)z1;
      } else {
        u0 = u_in[i0];
        if (i1 < n) u1 = u_in[i1];
      }
      {
        const T sv = s[src(i0, lds_)];
        const T xv = mu[src(i0, ldm)] + hb_exp(sv) * u0;
        x[i0] = xv;
        if (u_out) u_out[i0] = u0;
        acc += T(2) * sv + u0 * u0 - xv * xv;
      }
      if (i1 < n) {
        const T sv = s[src(i1, lds_)];
        const T xv = mu[src(i1, ldm)] + hb_exp(sv) * u1;
        x[i1] = xv;
        if (u_out) u_out[i1] = u1;
        acc += T(2) * sv + u1 * u1 - xv * xv;
      }
    }
  }
  if (rng && active) rng_store(rng, rng_lanes, t, g);
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) {
    if (nvblocks == 1)
      kl[0] = T(-0.5) * acc;  // one block covers everything: no finishing pass
    else
      partial[vblock] = acc;
  }
}

// VJP of the diagonal sampler: mubar = xbar + klbar x ; sbar = mubar exp(s) u - klbar
template <typename T>
__device__ __forceinline__ void diag_bwd_body(const T* __restrict__ s, const T* __restrict__ u, const T* __restrict__ x,
                                              const T* __restrict__ xbar, const T* __restrict__ klbar, T* __restrict__ mubar,
                                              T* __restrict__ sbar, long n, long L, long lds_, long ldo, long vblock,
                                              long nvblocks) {
  const T kb = klbar ? klbar[0] : T(0);
  const long stride = nvblocks * 256;
  const bool dense = lds_ == L && ldo == L;
  for (long i = vblock * 256 + threadIdx.x; i < n; i += stride) {
    // s read from, and the two gradients written into, column blocks of wider row-major matrices (see diag_fwd_body)
    long is = i, io = i;
    if (!dense) {
      const long r = i / L, c = i - r * L;
      is = r * lds_ + c;
      io = r * ldo + c;
    }
    const T mb = (xbar ? xbar[i] : T(0)) + kb * x[i];
    mubar[io] = mb;
    sbar[io] = mb * hb_exp(s[is]) * u[i] - kb;
  }
}

// ---- the recorded jobs
#define HB_SIDE_MAX 3
enum { HB_SIDE_GATHER_DRAW = 1, HB_SIDE_DIAG_FWD = 2, HB_SIDE_DIAG_BWD = 3 };
struct HbSideGather {
  GatherMultiArgs<float> g;
  uint64_t* state;
  long nlanes, lo;
  uint64_t range;
  long* idx_out;
  const long* perm;
  long n, nsrc;
  int* err;
  int G;
};
struct HbSideDiagFwd {
  const float *mu, *s, *u_in;
  uint64_t* rng;
  long rng_lanes;
  float *u_out, *x, *kl;
  long n, L, ldm, lds;
};
struct HbSideDiagBwd {
  const float *s, *u, *x, *xbar, *klbar;
  float *mubar, *sbar;
  long n, L, lds, ldo;
};
struct HbSideJob {
  int kind, nblocks;
  union {
    HbSideGather gather;
    HbSideDiagFwd dfwd;
    HbSideDiagBwd dbwd;
  };
};
struct HbSideJobs {
  int n, total;   // jobs recorded, sum of their workgroups
  HbSideJob job[HB_SIDE_MAX];
};

// one side workgroup (256 threads; `vb` = its index among the side workgroups of the launch)
__device__ __forceinline__ void hb_side_run(const HbSideJobs& J, int vb) {
  __shared__ float side_smem[16];
  // a host with more than four waves per block (chol_persist_kernel: 512 threads, its upper half leaves before this
  // call) makes block_sum read as many slots as the block has waves: the slots nobody writes must hold zeros
  if (threadIdx.x < 16) side_smem[threadIdx.x] = 0.f;
#pragma unroll
  for (int q = 0; q < HB_SIDE_MAX; ++q) {
    if (q >= J.n) return;
    const HbSideJob& j = J.job[q];
    if (vb < j.nblocks) {
      if (j.kind == HB_SIDE_GATHER_DRAW)
        gather_draw_body<float>(j.gather.g, j.gather.state, j.gather.nlanes, j.gather.lo, j.gather.range, j.gather.idx_out,
                                j.gather.perm, j.gather.n, j.gather.nsrc, j.gather.err, j.gather.G, vb);
      else if (j.kind == HB_SIDE_DIAG_FWD)
        diag_fwd_body<float>(j.dfwd.mu, j.dfwd.s, j.dfwd.u_in, j.dfwd.rng, j.dfwd.rng_lanes, j.dfwd.u_out, j.dfwd.x, nullptr,
                             j.dfwd.kl, j.dfwd.n, j.dfwd.L, j.dfwd.ldm, j.dfwd.lds, vb, 1, side_smem);
      else if (j.kind == HB_SIDE_DIAG_BWD)
        diag_bwd_body<float>(j.dbwd.s, j.dbwd.u, j.dbwd.x, j.dbwd.xbar, j.dbwd.klbar, j.dbwd.mubar, j.dbwd.sbar, j.dbwd.n,
                             j.dbwd.L, j.dbwd.lds, j.dbwd.ldo, vb, j.nblocks);
      return;
    }
    vb -= j.nblocks;
  }
}

// host side (runtime.hip): the pending list of the calling thread
int hb_side_push(const HbSideJob& job, hipStream_t stream);   // records; launches the pending ones first when the list is full
HbSideJobs hb_side_take();                                    // pending jobs for a host launch (list cleared)

#endif  // HB_SIDE_JOBS_CUH
