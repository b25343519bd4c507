// K1/K2: reparameterised Gaussian sampler fused with the Monte-Carlo KL term.
//
// Reference: Henbun/variationals.py:131-153 (_sample), :178-186 (logdet),
// :225-230 (Normal._KL = -0.5*sum(logdet + u^2 - x^2)).  One pass over
// mu, s(S), u: HBM-bound; algorithmic bytes = 3*n*B (diag: read mu,s; write x;
// u in registers when drawn in-kernel) or (size^2+3*size)*B (full rank).
#include "common.cuh"
#include "rng_pairs.cuh"
#include "side_jobs.cuh"  // diag_fwd_body, diag_bwd_body, the side-job list
#include "../../include/henbun_hip.h"

#define HB_KL_MAX_PARTIALS 2048

// ---------------------------------------------------------------------------
// diagonal
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) diag_fwd_kernel(const T* __restrict__ mu, const T* __restrict__ s,
                                                       const T* __restrict__ u_in, uint64_t* rng, long rng_lanes,
                                                       T* __restrict__ u_out, T* __restrict__ x,
                                                       T* __restrict__ partial, T* __restrict__ kl, long n, long L,
                                                       long ldm, long lds_) {
  __shared__ T smem[16];
  diag_fwd_body<T>(mu, s, u_in, rng, rng_lanes, u_out, x, partial, kl, n, L, ldm, lds_, (long)blockIdx.x, (long)gridDim.x, smem);
}

template <typename T>
__global__ void __launch_bounds__(256) kl_finish_kernel(const T* __restrict__ partial, int np, T* __restrict__ kl) {
  __shared__ T smem[16];
  T acc = T(0);
  for (int i = threadIdx.x; i < np; i += blockDim.x) acc += partial[i];
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) kl[0] = T(-0.5) * acc;
}

template <typename T>
static int diag_fwd(const T* mu, const T* s, const T* u_in, uint64_t* rng, long rng_lanes, T* u_out, T* x, T* kl,
                    long n, long L, long ldm, long lds_, T* ws, hipStream_t stream, bool defer = false) {
  HB_REQUIRE(n >= 0, "hb_diag_sample_kl_fwd: n < 0");
  HB_REQUIRE(L >= 1 && ldm >= L && lds_ >= L && n % L == 0, "hb_diag_sample_kl_fwd: bad row layout (L=%ld, ld=%ld/%ld, n=%ld)",
             L, ldm, lds_, n);
  HB_REQUIRE(mu && s && x && kl && ws, "hb_diag_sample_kl_fwd: NULL pointer");
  HB_REQUIRE(u_in || (rng && rng_lanes > 0), "hb_diag_sample_kl_fwd: neither u_in nor rng given");
  if (u_in) rng = nullptr;  // injected noise wins
  int grid;
  if (rng) {
    grid = hb_cdiv(rng_lanes, 256);
    HB_REQUIRE(grid <= HB_KL_MAX_PARTIALS, "hb_diag_sample_kl_fwd: rng_lanes too large");
  } else {
    grid = hb_stream_grid((n + 1) / 2, 256);
  }
  // blocks past the last pair would only contribute zeros
  const int need = hb_cdiv((n + 1) / 2, 256);
  if (grid > need) grid = need > 0 ? need : 1;
  if constexpr (sizeof(T) == 4) {
    if (defer && grid == 1) {   // one workgroup: recorded for the next host launch (side_jobs.cuh)
      HbSideJob j;
      j.kind = HB_SIDE_DIAG_FWD;
      j.nblocks = 1;
      j.dfwd.mu = mu; j.dfwd.s = s; j.dfwd.u_in = u_in; j.dfwd.rng = rng; j.dfwd.rng_lanes = rng_lanes; j.dfwd.u_out = u_out;
      j.dfwd.x = x; j.dfwd.kl = kl; j.dfwd.n = n; j.dfwd.L = L; j.dfwd.ldm = ldm; j.dfwd.lds = lds_;
      return hb_side_push(j, stream);
    }
  }
  hipLaunchKernelGGL(diag_fwd_kernel<T>, dim3(grid), dim3(256), 0, stream, mu, s, u_in, rng, rng_lanes, u_out, x, ws,
                     kl, n, L, ldm, lds_);
  HB_LAUNCH_CHECK();
  if (grid > 1) {
    hipLaunchKernelGGL(kl_finish_kernel<T>, dim3(1), dim3(256), 0, stream, ws, grid, kl);
    HB_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int hb_diag_sample_kl_fwd_f32(const float* mu, const float* s, const float* u_in, uint64_t* rng,
                                         long rng_lanes, float* u_out, float* x, float* kl, long n, long L, long ld_mu,
                                         long ld_s, float* ws, void* stream) {
  return diag_fwd<float>(mu, s, u_in, rng, rng_lanes, u_out, x, kl, n, L, ld_mu, ld_s, ws, (hipStream_t)stream);
}
extern "C" int hb_side_push_diag_fwd_f32(const float* mu, const float* s, const float* u_in, uint64_t* rng, long rng_lanes,
                                        float* u_out, float* x, float* kl, long n, long L, long ld_mu, long ld_s, float* ws,
                                        void* stream) {
  return diag_fwd<float>(mu, s, u_in, rng, rng_lanes, u_out, x, kl, n, L, ld_mu, ld_s, ws, (hipStream_t)stream, true);
}
extern "C" int hb_diag_sample_kl_fwd_f64(const double* mu, const double* s, const double* u_in, uint64_t* rng,
                                         long rng_lanes, double* u_out, double* x, double* kl, long n, long L,
                                         long ld_mu, long ld_s, double* ws, void* stream) {
  return diag_fwd<double>(mu, s, u_in, rng, rng_lanes, u_out, x, kl, n, L, ld_mu, ld_s, ws, (hipStream_t)stream);
}

template <typename T>
__global__ void __launch_bounds__(256) diag_bwd_kernel(const T* __restrict__ s, const T* __restrict__ u,
                                                       const T* __restrict__ x, const T* __restrict__ xbar,
                                                       const T* __restrict__ klbar, T* __restrict__ mubar,
                                                       T* __restrict__ sbar, long n, long L, long lds_, long ldo) {
  diag_bwd_body<T>(s, u, x, xbar, klbar, mubar, sbar, n, L, lds_, ldo, (long)blockIdx.x, (long)gridDim.x);
}

template <typename T>
static int diag_bwd(const T* s, const T* u, const T* x, const T* xbar, const T* klbar, T* mubar, T* sbar, long n,
                    long L, long lds_, long ldo, hipStream_t stream, bool defer = false) {
  HB_REQUIRE(n >= 0, "hb_diag_sample_kl_bwd: n < 0");
  HB_REQUIRE(L >= 1 && lds_ >= L && ldo >= L && n % L == 0, "hb_diag_sample_kl_bwd: bad row layout (L=%ld, ld=%ld/%ld, n=%ld)", L,
             lds_, ldo, n);
  HB_REQUIRE(s && u && x && mubar && sbar, "hb_diag_sample_kl_bwd: NULL pointer");
  if (n == 0) return 0;
  if constexpr (sizeof(T) == 4) {
    if (defer && hb_stream_grid(n, 256) <= 64) {   // small: recorded for the next host launch (side_jobs.cuh)
      HbSideJob j;
      j.kind = HB_SIDE_DIAG_BWD;
      j.nblocks = hb_stream_grid(n, 256);
      j.dbwd.s = s; j.dbwd.u = u; j.dbwd.x = x; j.dbwd.xbar = xbar; j.dbwd.klbar = klbar; j.dbwd.mubar = mubar; j.dbwd.sbar = sbar;
      j.dbwd.n = n; j.dbwd.L = L; j.dbwd.lds = lds_; j.dbwd.ldo = ldo;
      return hb_side_push(j, stream);
    }
  }
  hipLaunchKernelGGL(diag_bwd_kernel<T>, dim3(hb_stream_grid(n, 256)), dim3(256), 0, stream, s, u, x, xbar, klbar,
                     mubar, sbar, n, L, lds_, ldo);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_diag_sample_kl_bwd_f32(const float* s, const float* u, const float* x, const float* xbar,
                                         const float* klbar, float* mubar, float* sbar, long n, long L, long ld_s,
                                         long ld_out, void* stream) {
  return diag_bwd<float>(s, u, x, xbar, klbar, mubar, sbar, n, L, ld_s, ld_out, (hipStream_t)stream);
}
extern "C" int hb_side_push_diag_bwd_f32(const float* s, const float* u, const float* x, const float* xbar, const float* klbar,
                                        float* mubar, float* sbar, long n, long L, long ld_s, long ld_out, void* stream) {
  return diag_bwd<float>(s, u, x, xbar, klbar, mubar, sbar, n, L, ld_s, ld_out, (hipStream_t)stream, true);
}
extern "C" int hb_diag_sample_kl_bwd_f64(const double* s, const double* u, const double* x, const double* xbar,
                                         const double* klbar, double* mubar, double* sbar, long n, long L, long ld_s,
                                         long ld_out, void* stream) {
  return diag_bwd<double>(s, u, x, xbar, klbar, mubar, sbar, n, L, ld_s, ld_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// full rank: x_r = mu_r + tril(S_r) u_r
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) rng_fill_kernel(uint64_t* state, long nlanes, T* out, long n) {
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

// one wave per output (r,k): coalesced dot over j <= k.  Used for size > 64.
// Row k of block r of S: dense [rows, size, size] storage, or the packed lower triangle [rows, size(size+1)/2]
// in row-major order (row k holds its k+1 entries at offset k(k+1)/2: the order of numpy's tril_indices, which is
// what the reference's disabled vec_to_tri / LowerTriangular hook uses, tf_wraps.py:50-71, transforms.py:182-269).
// Packed storage halves the bytes the sampler reads (the full-rank q_sqrt of cfg 3 is 4.2 MB dense).
template <typename T>
__device__ __forceinline__ const T* fullrank_row(const T* S, long r, long k, long size, int packed) {
  return packed ? S + r * (size * (size + 1) / 2) + k * (k + 1) / 2 : S + (r * size + k) * size;
}

template <typename T>
__global__ void __launch_bounds__(256) fullrank_fwd_wave_kernel(const T* __restrict__ mu, const T* __restrict__ S,
                                                                const T* __restrict__ u, T* __restrict__ x,
                                                                T* __restrict__ partial, long rows, long size,
                                                                int packed) {
  __shared__ T smem[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long nout = rows * size;
  const long wstride = (long)gridDim.x * 4;
  T acc = T(0);
  for (long o = (long)blockIdx.x * 4 + w; o < nout; o += wstride) {
    const long r = o / size, k = o - r * size;
    const T* Srow = fullrank_row(S, r, k, size, packed);
    const T* ur = u + r * size;
    // four row segments in flight per step (same summation order as the plain loop: deterministic)
    T dot = T(0);
    long j = lane;
    for (; j + 192 <= k; j += 256) {
      const T s0 = Srow[j], s1 = Srow[j + 64], s2 = Srow[j + 128], s3 = Srow[j + 192];
      const T u0 = ur[j], u1 = ur[j + 64], u2 = ur[j + 128], u3 = ur[j + 192];
      dot += s0 * u0;
      dot += s1 * u1;
      dot += s2 * u2;
      dot += s3 * u3;
    }
    for (; j <= k; j += 64) dot += Srow[j] * ur[j];
    dot = wave_sum(dot);
    if (lane == 0) {
      const T xv = mu[o] + dot;
      x[o] = xv;
      const T skk = Srow[k], uk = ur[k];
      acc += hb_log(skk * skk) + uk * uk - xv * xv;
    }
  }
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// one thread per output: for small blocks (local variationals, size <= 64)
template <typename T>
__global__ void __launch_bounds__(256) fullrank_fwd_thread_kernel(const T* __restrict__ mu, const T* __restrict__ S,
                                                                  const T* __restrict__ u, T* __restrict__ x,
                                                                  T* __restrict__ partial, long rows, long size,
                                                                  int packed) {
  __shared__ T smem[16];
  const long nout = rows * size;
  const long stride = (long)gridDim.x * blockDim.x;
  T acc = T(0);
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < nout; o += stride) {
    const long r = o / size, k = o - r * size;
    const T* Srow = fullrank_row(S, r, k, size, packed);
    const T* ur = u + r * size;
    T dot = T(0);
    for (long j = 0; j <= k; ++j) dot += Srow[j] * ur[j];
    const T xv = mu[o] + dot;
    x[o] = xv;
    const T skk = Srow[k], uk = ur[k];
    acc += hb_log(skk * skk) + uk * uk - xv * xv;
  }
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// One launch (round 4): noise, sample and KL of a full-rank block of up to 1024 dimensions (cfg 3: q(u), M = 1024).
// The three-launch form above (fill u, rows, fold) is 17 us for 2 MB of triangle: launch boundaries, not bytes.  Here
//   * every workgroup draws the WHOLE u itself into LDS from the unchanged generator states (lane t draws the pairs
//     t, t + lanes, ... exactly as rng_fill_kernel does: same variates), while the loads of its rows of S -- issued before
//     the first variate is drawn -- are in flight; workgroup 0 also writes u_out;
//   * a wave owns the rows k and size - 1 - k of one block (size + 1 elements together: balanced), every lane has all its
//     elements of both rows in flight at once, and sums them in the order of the three-launch form;
//   * the workgroups meet at an arrival counter (`sync`: one zero word, left zero): the LAST one folds the partial sums
//     in a fixed order, writes kl, and advances the generator states -- nobody reads them any more.
// ---------------------------------------------------------------------------------------------------------------
#define HB_FR1_MAXSIZE 1024
#define HB_FR1_MAXN 8192
template <typename T>
__global__ void __launch_bounds__(256) fullrank_fwd_one_kernel(const T* __restrict__ mu, const T* __restrict__ S,
                                                               const T* __restrict__ u_in, uint64_t* rng, long nlanes,
                                                               T* __restrict__ u_out, T* __restrict__ x, T* partial,
                                                               T* __restrict__ kl, unsigned* sync, long rows, long size, int packed) {
  __shared__ T us[HB_FR1_MAXN];
  __shared__ T smem[16];
  __shared__ unsigned s_last;
  constexpr int PER = HB_FR1_MAXSIZE / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long n = rows * size, npairs = (n + 1) / 2;
  const long half = (size + 1) / 2, units = rows * half;
  const long unit = (long)blockIdx.x * 4 + w;
  const bool live = unit < units;
  const long r = live ? unit / half : 0, kp = live ? unit - r * half : 0;
  const long ka = kp, kb = size - 1 - kp;                  // (the middle row of an odd size: ka == kb, taken once)
  const T* Sa = fullrank_row(S, r, ka, size, packed);
  const T* Sb = fullrank_row(S, r, kb, size, packed);
  T sa[PER], sb[PER];
  HbRng gs[HB_FR1_MAXN / 512];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const long j = lane + 64 * i;
    sa[i] = (live && j <= ka) ? Sa[j] : T(0);
    sb[i] = (live && j <= kb && kb != ka) ? Sb[j] : T(0);
  }
  // ---- u: injected, or drawn (all of it, by every workgroup, from the states as they stand)
  if (u_in) {
    for (long i = threadIdx.x; i < n; i += 256) {
      const T v = u_in[i];
      us[i] = v;
      if (blockIdx.x == 0 && u_out && u_out != u_in) u_out[i] = v;
    }
  } else {
    // (a thread keeps the advanced states of its lanes -- at most HB_FR1_MAXN / 2 / 256 = 16 -- for the last workgroup to store)
#pragma unroll
    for (int q = 0; q < HB_FR1_MAXN / 512; ++q) {
      const long t = threadIdx.x + 256L * q;
      if (t < nlanes && t < npairs) {
        gs[q] = rng_load(rng, nlanes, t);
        for (long p = t; p < npairs; p += nlanes) {
          T z0, z1;
          gs[q].normal2(z0, z1);
          us[2 * p] = z0;
          if (2 * p + 1 < n) us[2 * p + 1] = z1;
          if (blockIdx.x == 0) {
            u_out[2 * p] = z0;
            if (2 * p + 1 < n) u_out[2 * p + 1] = z1;
          }
        }
      }
    }
  }
  __syncthreads();
  T acc = T(0);
  if (live) {
    const T* ur = us + r * size;
    T da = T(0), db = T(0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const long j = lane + 64 * i;
      if (j <= ka) da += sa[i] * ur[j];
      if (j <= kb && kb != ka) db += sb[i] * ur[j];
    }
    da = wave_sum(da), db = wave_sum(db);
    if (lane == 0) {
      const long oa = r * size + ka, ob = r * size + kb;
      const T xa = mu[oa] + da;
      x[oa] = xa;
      const T ska = Sa[ka], uka = ur[ka];
      acc += hb_log(ska * ska) + uka * uka - xa * xa;
      if (kb != ka) {
        const T xb = mu[ob] + db;
        x[ob] = xb;
        const T skb = Sb[kb], ukb = ur[kb];
        acc += hb_log(skb * skb) + ukb * ukb - xb * xb;
      }
    }
  }
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) {
    // (write-through store, drained, then the count: the hand-off of cdna_hip_programming.md Guideline 16 R1, no L2 write-back)
    __hip_atomic_store(partial + blockIdx.x, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup: fold (fixed order), advance the generator, leave the counter zero
  T tot = T(0);
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) tot += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tot = block_sum(tot, smem);
  if (threadIdx.x == 0) {
    kl[0] = T(-0.5) * tot;
    __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!u_in) {
#pragma unroll
    for (int q = 0; q < HB_FR1_MAXN / 512; ++q) {
      const long t = threadIdx.x + 256L * q;
      if (t < nlanes && t < npairs) rng_store(rng, nlanes, t, gs[q]);
    }
  }
}

template <typename T>
static int fullrank_fwd_one(const T* mu, const T* S, const T* u_in, uint64_t* rng, long rng_lanes, T* u_out, T* x, T* kl,
                            long rows, long size, int packed, T* ws, unsigned* sync, hipStream_t stream) {
  const long units = rows * ((size + 1) / 2);
  const int grid = (int)((units + 3) / 4);
  hipLaunchKernelGGL(fullrank_fwd_one_kernel<T>, dim3(grid), dim3(256), 0, stream, mu, S, u_in, u_in ? (uint64_t*)nullptr : rng,
                     rng_lanes, u_out, x, ws, kl, sync, rows, size, packed);
  HB_LAUNCH_CHECK();
  return 0;
}
// 1 when hb_fullrank_sample_kl_fwd1 takes its one-launch form for this shape (otherwise it runs the three-launch form)
extern "C" int hb_fullrank_one_launch_shape(long rows, long size) {
  return (size > 64 && size <= HB_FR1_MAXSIZE && rows * size <= HB_FR1_MAXN && rows * ((size + 1) / 2) <= 4L * HB_KL_MAX_PARTIALS &&
          hb_debug_get("fullrank_three_launches", 0) == 0)
             ? 1
             : 0;
}

template <typename T>
static int fullrank_fwd(const T* mu, const T* S, const T* u_in, uint64_t* rng, long rng_lanes, T* u_out, T* x, T* kl,
                        long rows, long size, int packed, T* ws, hipStream_t stream) {
  HB_REQUIRE(rows >= 0 && size >= 0, "hb_fullrank_sample_kl_fwd: negative extent");
  HB_REQUIRE(mu && S && x && kl && ws, "hb_fullrank_sample_kl_fwd: NULL pointer");
  HB_REQUIRE(u_in || (rng && rng_lanes > 0 && u_out), "hb_fullrank_sample_kl_fwd: need u_in, or rng and u_out");
  const long n = rows * size;
  const T* u = u_in;
  if (!u_in) {
    if (n > 0) {
      hipLaunchKernelGGL(rng_fill_kernel<T>, dim3(hb_cdiv(rng_lanes, 256)), dim3(256), 0, stream, rng, rng_lanes,
                         u_out, n);
      HB_LAUNCH_CHECK();
    }
    u = u_out;
  } else if (u_out && u_out != u_in && n > 0) {
    HB_HIP(hb_copy_async(u_out, u_in, sizeof(T) * n, stream));
  }
  int grid;
  if (size > 64) {
    grid = (int)((n + 3) / 4);
    if (grid > HB_KL_MAX_PARTIALS) grid = HB_KL_MAX_PARTIALS;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(fullrank_fwd_wave_kernel<T>, dim3(grid), dim3(256), 0, stream, mu, S, u, x, ws, rows, size,
                       packed);
  } else {
    grid = hb_stream_grid(n, 256);
    hipLaunchKernelGGL(fullrank_fwd_thread_kernel<T>, dim3(grid), dim3(256), 0, stream, mu, S, u, x, ws, rows, size,
                       packed);
  }
  HB_LAUNCH_CHECK();
  hipLaunchKernelGGL(kl_finish_kernel<T>, dim3(1), dim3(256), 0, stream, ws, grid, kl);
  HB_LAUNCH_CHECK();
  return 0;
}

extern "C" int hb_fullrank_sample_kl_fwd_f32(const float* mu, const float* S, const float* u_in, uint64_t* rng,
                                             long rng_lanes, float* u_out, float* x, float* kl, long rows, long size,
                                             int packed, float* ws, void* stream) {
  return fullrank_fwd<float>(mu, S, u_in, rng, rng_lanes, u_out, x, kl, rows, size, packed, ws, (hipStream_t)stream);
}
extern "C" int hb_fullrank_sample_kl_fwd_f64(const double* mu, const double* S, const double* u_in, uint64_t* rng,
                                             long rng_lanes, double* u_out, double* x, double* kl, long rows,
                                             long size, int packed, double* ws, void* stream) {
  return fullrank_fwd<double>(mu, S, u_in, rng, rng_lanes, u_out, x, kl, rows, size, packed, ws, (hipStream_t)stream);
}

// `sync`: ONE zero 32-bit word owned by the caller for this stream (zero at entry, zero at exit); NULL or a shape outside
// hb_fullrank_one_launch_shape: the three-launch form.
extern "C" int hb_fullrank_sample_kl_fwd1_f32(const float* mu, const float* S, const float* u_in, uint64_t* rng,
                                              long rng_lanes, float* u_out, float* x, float* kl, long rows, long size,
                                              int packed, float* ws, unsigned* sync, void* stream) {
  if (!sync || !hb_fullrank_one_launch_shape(rows, size))
    return fullrank_fwd<float>(mu, S, u_in, rng, rng_lanes, u_out, x, kl, rows, size, packed, ws, (hipStream_t)stream);
  HB_REQUIRE(mu && S && x && kl && ws, "hb_fullrank_sample_kl_fwd1: NULL pointer");
  HB_REQUIRE(u_in || (rng && rng_lanes > 0 && u_out), "hb_fullrank_sample_kl_fwd1: need u_in, or rng and u_out");
  return fullrank_fwd_one<float>(mu, S, u_in, rng, rng_lanes, u_out, x, kl, rows, size, packed, ws, sync, (hipStream_t)stream);
}
extern "C" int hb_fullrank_sample_kl_fwd1_f64(const double* mu, const double* S, const double* u_in, uint64_t* rng,
                                              long rng_lanes, double* u_out, double* x, double* kl, long rows, long size,
                                              int packed, double* ws, unsigned* sync, void* stream) {
  if (!sync || !hb_fullrank_one_launch_shape(rows, size))
    return fullrank_fwd<double>(mu, S, u_in, rng, rng_lanes, u_out, x, kl, rows, size, packed, ws, (hipStream_t)stream);
  HB_REQUIRE(mu && S && x && kl && ws, "hb_fullrank_sample_kl_fwd1: NULL pointer");
  HB_REQUIRE(u_in || (rng && rng_lanes > 0 && u_out), "hb_fullrank_sample_kl_fwd1: need u_in, or rng and u_out");
  return fullrank_fwd_one<double>(mu, S, u_in, rng, rng_lanes, u_out, x, kl, rows, size, packed, ws, sync, (hipStream_t)stream);
}

template <typename T>
__global__ void __launch_bounds__(256) fullrank_bwd_kernel(const T* __restrict__ S, const T* __restrict__ u,
                                                           const T* __restrict__ x, const T* __restrict__ xbar,
                                                           const T* __restrict__ klbar, T* __restrict__ mubar,
                                                           T* __restrict__ Sbar, long rows, long size) {
  const T kb = klbar ? klbar[0] : T(0);
  const long total = rows * size * size;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long rk = t / size;         // r*size + k
    const long j = t - rk * size;
    const long r = rk / size, k = rk - r * size;
    const T mb = (xbar ? xbar[rk] : T(0)) + kb * x[rk];
    if (j == 0) mubar[rk] = mb;
    T g = T(0);
    if (j < k) {
      g = mb * u[r * size + j];
    } else if (j == k) {
      g = mb * u[rk] - kb / S[t];
    }
    Sbar[t] = g;
  }
}

// index of the row that holds packed offset p (p = k(k+1)/2 + j, j <= k)
__device__ __forceinline__ long tri_row_of(long p) {
  long k = (long)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
  while (k * (k + 1) / 2 > p) --k;
  while ((k + 1) * (k + 2) / 2 <= p) ++k;
  return k;
}

// the same gradient written as the packed lower triangle [rows, size(size+1)/2] (S is packed too)
template <typename T>
__global__ void __launch_bounds__(256) fullrank_bwd_packed_kernel(const T* __restrict__ S, const T* __restrict__ u,
                                                                  const T* __restrict__ x, const T* __restrict__ xbar,
                                                                  const T* __restrict__ klbar, T* __restrict__ mubar,
                                                                  T* __restrict__ Sbar, long rows, long size) {
  const T kb = klbar ? klbar[0] : T(0);
  const long tri = size * (size + 1) / 2;
  const long total = rows * tri;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long r = t / tri, p = t - r * tri;
    const long k = tri_row_of(p), j = p - k * (k + 1) / 2;
    const long rk = r * size + k;
    const T mb = (xbar ? xbar[rk] : T(0)) + kb * x[rk];
    if (j == 0) mubar[rk] = mb;
    Sbar[t] = j < k ? mb * u[r * size + j] : mb * u[rk] - kb / S[t];
  }
}

template <typename T>
static int fullrank_bwd(const T* S, const T* u, const T* x, const T* xbar, const T* klbar, T* mubar, T* Sbar,
                        long rows, long size, int packed, hipStream_t stream) {
  HB_REQUIRE(rows >= 0 && size >= 0, "hb_fullrank_sample_kl_bwd: negative extent");
  HB_REQUIRE(S && u && x && mubar && Sbar, "hb_fullrank_sample_kl_bwd: NULL pointer");
  const long total = packed ? rows * (size * (size + 1) / 2) : rows * size * size;
  if (total == 0) return 0;
  if (packed)
    hipLaunchKernelGGL(fullrank_bwd_packed_kernel<T>, dim3(hb_stream_grid(total, 256)), dim3(256), 0, stream, S, u, x,
                       xbar, klbar, mubar, Sbar, rows, size);
  else
    hipLaunchKernelGGL(fullrank_bwd_kernel<T>, dim3(hb_stream_grid(total, 256)), dim3(256), 0, stream, S, u, x, xbar,
                       klbar, mubar, Sbar, rows, size);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_fullrank_sample_kl_bwd_f32(const float* S, const float* u, const float* x, const float* xbar,
                                             const float* klbar, float* mubar, float* Sbar, long rows, long size,
                                             int packed, void* stream) {
  return fullrank_bwd<float>(S, u, x, xbar, klbar, mubar, Sbar, rows, size, packed, (hipStream_t)stream);
}
extern "C" int hb_fullrank_sample_kl_bwd_f64(const double* S, const double* u, const double* x, const double* xbar,
                                             const double* klbar, double* mubar, double* Sbar, long rows, long size,
                                             int packed, void* stream) {
  return fullrank_bwd<double>(S, u, x, xbar, klbar, mubar, Sbar, rows, size, packed, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// vec_to_tri / tri_to_vec: the reference's (disabled) native op pair, Henbun/tf_wraps.py:50-71 -- a [B, N(N+1)/2]
// vector <-> [B, N, N] lower-triangular matrices, entries in numpy tril_indices order (transforms.py:225-244).
// Each is the other's gradient (tf_wraps.py:56-58).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) vec_to_tri_kernel(const T* __restrict__ v, T* __restrict__ tri, long B, long N) {
  const long total = B * N * N, tsz = N * (N + 1) / 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long b = t / (N * N), rem = t - b * N * N;
    const long i = rem / N, j = rem - i * N;
    tri[t] = j <= i ? v[b * tsz + i * (i + 1) / 2 + j] : T(0);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) tri_to_vec_kernel(const T* __restrict__ tri, T* __restrict__ v, long B, long N) {
  const long tsz = N * (N + 1) / 2, total = B * tsz;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long b = t / tsz, p = t - b * tsz;
    const long i = tri_row_of(p), j = p - i * (i + 1) / 2;
    v[t] = tri[(b * N + i) * N + j];
  }
}
template <typename T>
static int tri_pack_launch(const T* src, T* dst, long B, long N, bool to_tri, hipStream_t stream) {
  HB_REQUIRE(B >= 0 && N >= 0, "hb_vec_to_tri / hb_tri_to_vec: negative extent");
  const long total = to_tri ? B * N * N : B * (N * (N + 1) / 2);
  if (total == 0) return 0;
  HB_REQUIRE(src && dst, "hb_vec_to_tri / hb_tri_to_vec: NULL pointer");
  if (to_tri)
    hipLaunchKernelGGL(vec_to_tri_kernel<T>, dim3(hb_stream_grid(total, 256)), dim3(256), 0, stream, src, dst, B, N);
  else
    hipLaunchKernelGGL(tri_to_vec_kernel<T>, dim3(hb_stream_grid(total, 256)), dim3(256), 0, stream, src, dst, B, N);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_vec_to_tri_f32(const float* v, float* tri, long B, long N, void* stream) {
  return tri_pack_launch<float>(v, tri, B, N, true, (hipStream_t)stream);
}
extern "C" int hb_vec_to_tri_f64(const double* v, double* tri, long B, long N, void* stream) {
  return tri_pack_launch<double>(v, tri, B, N, true, (hipStream_t)stream);
}
extern "C" int hb_tri_to_vec_f32(const float* tri, float* v, long B, long N, void* stream) {
  return tri_pack_launch<float>(tri, v, B, N, false, (hipStream_t)stream);
}
extern "C" int hb_tri_to_vec_f64(const double* tri, double* v, long B, long N, void* stream) {
  return tri_pack_launch<double>(tri, v, B, N, false, (hipStream_t)stream);
}
