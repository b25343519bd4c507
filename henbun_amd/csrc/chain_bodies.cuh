// Bodies of the SMALL kernels of the step -- the flat-buffer Adam update, the one-workgroup Gaussian likelihood head, the
// sparse-GP finishing pass, the lengthscale-gradient fold -- written once, as device functions of a (first thread,
// thread count) pair, and used twice:
//   * by their ahead-of-time kernels (adam.hip, elementwise.hip, sgp.hip, gram.hip), which pass their own grid;
//   * by run-time generated SERIAL CHAIN kernels (csrc/jit.hip, hb_chain_*): several dependent small launches of a plan
//     (sparse-GP finish -> likelihood head -> its elementwise cluster;  lengthscale fold -> gradient cluster -> Adam)
//     become ONE workgroup of 1024 threads that runs them back to back with a barrier in between -- a kernel boundary
//     inside the captured step costs ~4.5 us, more than any of these bodies takes.
// Part of the hiprtc prelude: keep this file self-contained (needs ew_math.cuh and rng_core.cuh before it; no #include).
#ifndef HB_CHAIN_BODIES_CUH
#define HB_CHAIN_BODIES_CUH

// Floating-point contraction OFF in this file (as in ew_math.cuh / ew_apply.cuh): the same body must return the same
// bits from its ahead-of-time kernel and from a run-time generated chain, whatever each compiler would have fused.
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------------------------- Adam
// Failure containment (reference behaviour: tf.cholesky raises inside session.run BEFORE apply_gradients, so the
// parameters stay at the last good step).  Calls are asynchronous here, so the update itself looks at this step's
// factorisation status words (`info[n_info]`, LAPACK convention, written earlier in the same stream / graph), at
// the all-reduced failure flag of the other ranks (`dpflag`, nullable) and at the sticky record `fail[2]`
// (nullable): if any is non-zero the launch is a no-op -- theta, m, v and the step counter keep their values --
// and `fail` records the first failing step {t+1, first non-zero status seen}.  Every later step is then a no-op
// too, until the host clears `fail`.
template <typename T>
__device__ __forceinline__ int adam_step_blocked(const long* t, const int* info, long n_info, const T* dpflag,
                                                 long* fail, bool record) {
  int bad = 0, what = 0;
  if (fail != nullptr && fail[0] != 0) bad = 1;
  if (dpflag != nullptr && dpflag[0] != (T)0) { bad = 1; what = -1; }
  for (long i = threadIdx.x; i < n_info; i += blockDim.x) {
    const int w = info[i];
    if (w != 0) { bad = 1; what = w; }
  }
  const int any = __syncthreads_or(bad);
  if (any && record && fail != nullptr) {
    // one writer: the lowest thread that saw a status word (or thread 0 for the flag-only case)
    __shared__ int who;
    if (threadIdx.x == 0) who = blockDim.x;
    __syncthreads();
    if (what != 0) atomicMin(&who, (int)threadIdx.x);
    __syncthreads();
    const int writer = who == (int)blockDim.x ? 0 : who;
    if ((int)threadIdx.x == writer && fail[0] == 0) {
      fail[1] = (long)what;
      fail[0] = t[0] + 1;
    }
  }
  return any;
}

// i0 / stride: this thread's first element and the thread count of the whole launch; `record`: this workgroup writes the
// failure record; `owner`: this workgroup is the only one of the launch (it may advance the step counter itself)
template <typename T>
__device__ __forceinline__ void hb_adam_body(T* __restrict__ theta, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v,
                                             long n, double lr, double b1, double b2, double eps, double gscale, long* t,
                                             int tick, const int* info, long n_info, const T* dpflag, long* fail, long i0,
                                             long stride, bool record, bool owner) {
  // The first batch of operands (8 elements per thread: a 2048-parameter model in one go) and the step counter are
  // requested BEFORE the status check: the check, the counter and the update were three dependent memory round
  // trips in a kernel whose arithmetic is a few hundred cycles.
  constexpr int U = 8;
  T g0[U], m0[U], v0[U], th0[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const long i = i0 + k * stride, ic = i < n ? i : (n > 0 ? n - 1 : 0);
    g0[k] = g[ic], m0[k] = m[ic], v0[k] = v[ic], th0[k] = theta[ic];
  }
  const long tnow = t[0];
  if (adam_step_blocked<T>(t, info, n_info, dpflag, fail, record)) return;
  const double tt = (double)(tnow + 1);
  const T lr_t = (T)(lr * sqrt(1.0 - pow(b2, tt)) / (1.0 - pow(b1, tt)));
  const T c1 = (T)b1, c2 = (T)b2, d1 = (T)(1.0 - b1), d2 = (T)(1.0 - b2), e = (T)eps, gs = (T)gscale;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const long i = i0 + k * stride;
    if (i < n) {
      const T gi = g0[k] * gs;
      const T mi = c1 * m0[k] + d1 * gi;
      const T vi = c2 * v0[k] + d2 * gi * gi;
      m[i] = mi;
      v[i] = vi;
      theta[i] = th0[k] - lr_t * mi / (hb_sqrt(vi) + e);
    }
  }
  for (long i = i0 + U * stride; i < n; i += stride) {
    const T gi = g[i] * gs;
    const T mi = c1 * m[i] + d1 * gi;
    const T vi = c2 * v[i] + d2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    theta[i] -= lr_t * mi / (hb_sqrt(vi) + e);
  }
  if (owner && tick) {
    // a single block owns the whole update: it advances the step counter itself (every thread has read
    // t[0] by the barrier), saving the separate tick launch
    __syncthreads();
    if (threadIdx.x == 0) t[0] += 1;
  }
}

// Large parameter sets (several workgroups): 16 bytes per lane and array, every operand of a thread requested before the
// first is used.  theta, g, m, v 16-byte aligned; the n % (16 / sizeof(T)) trailing elements are updated one by one by the
// first threads of the launch.  One status check per workgroup.
template <typename T>
__device__ __forceinline__ void hb_adam_body_vec(T* __restrict__ theta, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v,
                                                 long n, double lr, double b1, double b2, double eps, double gscale, long* t,
                                                 const int* info, long n_info, const T* dpflag, long* fail, long i0, long stride,
                                                 bool record) {
  constexpr int VEC = 16 / (int)sizeof(T), U = 2;
  typedef T VT __attribute__((ext_vector_type(VEC)));
  const long nv = n / VEC;
  VT g0[U], m0[U], v0[U], th0[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const long i = i0 + k * stride, ic = i < nv ? i : (nv > 0 ? nv - 1 : 0);
    g0[k] = reinterpret_cast<const VT*>(g)[ic], m0[k] = reinterpret_cast<const VT*>(m)[ic];
    v0[k] = reinterpret_cast<const VT*>(v)[ic], th0[k] = reinterpret_cast<const VT*>(theta)[ic];
  }
  const long tnow = t[0];
  if (adam_step_blocked<T>(t, info, n_info, dpflag, fail, record)) return;
  const double tt = (double)(tnow + 1);
  const T lr_t = (T)(lr * sqrt(1.0 - pow(b2, tt)) / (1.0 - pow(b1, tt)));
  const T c1 = (T)b1, c2 = (T)b2, d1 = (T)(1.0 - b1), d2 = (T)(1.0 - b2), e = (T)eps, gs = (T)gscale;
  auto upd = [&](VT gv, VT mv, VT vv, VT tv, long i) {
    VT mo, vo, to;
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const T gi = gv[q] * gs;
      const T mi = c1 * mv[q] + d1 * gi;
      const T vi = c2 * vv[q] + d2 * gi * gi;
      mo[q] = mi, vo[q] = vi, to[q] = tv[q] - lr_t * mi / (hb_sqrt(vi) + e);
    }
    reinterpret_cast<VT*>(m)[i] = mo;
    reinterpret_cast<VT*>(v)[i] = vo;
    reinterpret_cast<VT*>(theta)[i] = to;
  };
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const long i = i0 + k * stride;
    if (i < nv) upd(g0[k], m0[k], v0[k], th0[k], i);
  }
  for (long i = i0 + U * stride; i < nv; i += stride)
    upd(reinterpret_cast<const VT*>(g)[i], reinterpret_cast<const VT*>(m)[i], reinterpret_cast<const VT*>(v)[i],
        reinterpret_cast<const VT*>(theta)[i], i);
  const long it = nv * VEC + i0;
  if (it < n) {
    const T gi = g[it] * gs;
    const T mi = c1 * m[it] + d1 * gi;
    const T vi = c2 * v[it] + d2 * gi * gi;
    m[it] = mi;
    v[it] = vi;
    theta[it] -= lr_t * mi / (hb_sqrt(vi) + e);
  }
}

// ------------------------------------------------------------------------------------- Gaussian likelihood head (one WG)
// sum_j log N(x_j | f_j * scale, var) with its gradients, n <= 16 * blockDim.x: every load of a thread is in flight before
// the first use.  blockDim.x = 1024.  fbar (nullable): also scale * (post * dmu_j), the gradient of post * ll w.r.t. f.
template <typename T>
__device__ __forceinline__ void hb_gauss_ll_single_body(const T* __restrict__ x, const T* __restrict__ f, const T* __restrict__ scale,
                                                        const T* __restrict__ var, long n, T* __restrict__ dmu, T* __restrict__ ll,
                                                        T* __restrict__ dscale, T* __restrict__ dvar, T* smem,
                                                        T* __restrict__ fbar = nullptr, T post = T(0)) {
  constexpr int PER = 16;
  const T s = scale ? scale[0] : T(1), v = var[0];
  const T iv = T(1) / v, lc = T(-0.91893853320467274178) - T(0.5) * hb_log(v);
  T xv[PER], fv[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const long j = q * 1024 + threadIdx.x;
    const long jc = j < n ? j : n - 1;
    xv[q] = x[jc];
    fv[q] = f[jc];
  }
  T all = T(0), asc = T(0), avr = T(0);
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const long j = q * 1024 + threadIdx.x;
    const T dlt = xv[q] - fv[q] * s;
    const T g = dlt * iv;
    if (j < n) {
      dmu[j] = g;
      if (fbar) fbar[j] = s * (post * g);   // the gradient handed to the producer of f: scale * (upstream * dmu), same order as the graph's ops
      all += lc - T(0.5) * dlt * g;
      asc += g * fv[q];
      avr += T(-0.5) * iv + T(0.5) * g * g;
    }
  }
  all = block_sum(all, smem);
  asc = block_sum(asc, smem);
  avr = block_sum(avr, smem);
  if (threadIdx.x == 0) {
    ll[0] = all;
    dscale[0] = asc;
    dvar[0] = avr;
  }
}

// One point of the head (the operations of hb_gauss_ll_single_body's loop, for callers outside this file: contraction
// is off HERE, so the forward strip kernel of csrc/sgp.hip gets the head's bits from it)
template <typename T>
__device__ __forceinline__ void hb_gauss_point(T xv, T fv, T s, T iv, T lc, T& g, T& all, T& asc, T& avr) {
  const T dlt = xv - fv * s;
  g = dlt * iv;
  all += lc - T(0.5) * dlt * g;
  asc += g * fv;
  avr += T(-0.5) * iv + T(0.5) * g * g;
}

// Fold of the per-unit partial sums (ll, dscale, dvar) that a multi-workgroup head leaves as partial[3][nb] -- the
// partial-sum kernel of hb_gauss_ll, or the forward strip kernel of hb_sgp_fwd_gauss, one unit per strip.  One workgroup.
template <typename T>
__device__ __forceinline__ void hb_gauss_fold_body(const T* __restrict__ partial, long nb, T* __restrict__ ll, T* __restrict__ dscale,
                                                   T* __restrict__ dvar, T* smem) {
  T a0 = T(0), a1 = T(0), a2 = T(0);
  for (long i = threadIdx.x; i < nb; i += blockDim.x) {
    a0 += partial[i];
    a1 += partial[nb + i];
    a2 += partial[2 * nb + i];
  }
  a0 = block_sum(a0, smem);
  a1 = block_sum(a1, smem);
  a2 = block_sum(a2, smem);
  if (threadIdx.x == 0) {
    ll[0] = a0;
    dscale[0] = a1;
    dvar[0] = a2;
  }
}

// -------------------------------------------------------------------------------------------- sparse-GP finishing pass
// f, v (and the residual noise) from the column partials of the forward contraction: v = 1 - sum A^2, f = u A + sqrt|v| eps.
// The noise is drawn here (same per-lane streams and pair order as the stand-alone fill) or taken from eps_in.
template <typename T>
__device__ __forceinline__ void sgp_finish_one(const T* __restrict__ part, int gy, long idx, T epsv, T* __restrict__ f,
                                               T* __restrict__ v, T* __restrict__ eps_out, long n, long P, int diagonal) {
  const long e = idx / n, j = idx - e * n;
  const T* pp = part + e * gy * 5 * n + j;
  T s = T(0);
  for (int y = 0; y < gy; ++y) s += pp[(long)y * 5 * n];
  const T vv = T(1) - s;
  v[idx] = vv;
  if (eps_out) eps_out[idx] = epsv;
  const T scale = diagonal ? hb_sqrt(hb_abs(vv)) * epsv : T(0);
  for (long p = 0; p < P; ++p) {
    T mean = T(0);
    for (int y = 0; y < gy; ++y) mean += pp[((long)y * 5 + 1 + p) * n];
    f[(e * P + p) * n + j] = mean + scale;
  }
}

// t0 / nthreads: this thread's index in, and the size of, the whole launch.  With an RNG, thread t serves the lanes
// t, t + nthreads, ... (a launch of >= min(nlanes, npairs) threads gives one lane per thread, as the stand-alone kernel does).
template <typename T>
__device__ __forceinline__ void hb_sgp_finish_body(const T* __restrict__ part, int gy, const T* __restrict__ eps_in, uint64_t* rng,
                                                   long nlanes, T* __restrict__ eps_out, T* __restrict__ f, T* __restrict__ v,
                                                   long total, long n, long P, int diagonal, long t0, long nthreads) {
  const long npairs = (total + 1) / 2;
  if (rng) {
    const long nact = nlanes < npairs ? nlanes : npairs;
    for (long t = t0; t < nact; t += nthreads) {
      HbRng g = rng_load(rng, nlanes, t);
      for (long p = t; p < npairs; p += nlanes) {
        T z0, z1;
        g.normal2(z0, z1);
        sgp_finish_one<T>(part, gy, 2 * p, z0, f, v, eps_out, n, P, diagonal);
        if (2 * p + 1 < total) sgp_finish_one<T>(part, gy, 2 * p + 1, z1, f, v, eps_out, n, P, diagonal);
      }
      rng_store(rng, nlanes, t, g);
    }
  } else {
    for (long idx = t0; idx < total; idx += nthreads)
      sgp_finish_one<T>(part, gy, idx, eps_in ? eps_in[idx] : T(0), f, v, (eps_out != eps_in) ? eps_out : nullptr, n, P, diagonal);
  }
}

// ---------------------------------------------------------------------------------- lengthscale-gradient fold (gram_bwd)
// ellbar[c] = sum_r partial[r, c (or all columns when dl == 1)] for one (column c, batch entry): one workgroup
template <typename T>
__device__ __forceinline__ void hb_gram_ell_body(const T* __restrict__ partial, long rows, long d, long dl, long c,
                                                 T* __restrict__ ellbar, T* smem) {
  T acc = T(0);
  if (dl == 1) {
#pragma unroll 4
    for (long t = threadIdx.x; t < rows * d; t += blockDim.x) acc += partial[t];
  } else {
#pragma unroll 4
    for (long r = threadIdx.x; r < rows; r += blockDim.x) acc += partial[r * d + c];
  }
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) ellbar[c] = acc;
}

#pragma clang fp contract(fast)
#endif  // HB_CHAIN_BODIES_CUH
