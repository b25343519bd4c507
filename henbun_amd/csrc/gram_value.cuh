// One entry of a stationary Gram matrix (reference Henbun/gp/kernels.py:54-84,110-111,122-131): shared by gram_fwd_kernel
// (csrc/gram.hip) and by the persistent Cholesky, which synthesises its tiles of K(z, z) + jitter I itself when the planner
// hands it the inducing points instead of a materialised matrix (csrc/chol_persist.cuh) -- the same function, the same bits.
#pragma once
#include "common.cuh"
#include "../../include/henbun_hip.h"

template <typename T>
__device__ __forceinline__ T gram_value(int kind, const T* __restrict__ xi, const T* __restrict__ xj,
                                        const T* __restrict__ ell, long dl, long d) {
  T r2 = T(0), r2m = T(0);
  for (long k = 0; k < d; ++k) {
    const T il = T(1) / ell[dl == 1 ? 0 : k];
    const T a = xi[k] * il, b = xj[k] * il;
    r2 += (a - b) * (a - b);
    r2m += (a + b) * (a + b);
  }
  if (kind == HB_KERN_SQDIST) return r2;
  T v = hb_exp(T(-0.5) * r2);
  if (kind == HB_KERN_CSYM_RBF) v += hb_exp(T(-0.5) * r2m);
  return v;
}

