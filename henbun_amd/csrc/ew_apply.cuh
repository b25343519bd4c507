// The elementwise op table (HB_EW_* of include/henbun_hip.h): used by ew_kernel, by the interpreted programs and, compiled
// at run time by hiprtc, by the generated kernels of csrc/jit.hip (where the op code is a literal and the switch
// folds away).  Self-contained apart from ew_math.cuh and the HB_EW_* enum, which must be visible before this file.
#ifndef HB_EW_APPLY_CUH
#define HB_EW_APPLY_CUH

// Floating-point contraction is switched OFF for everything in this file and back to the HIP default at its end.
// Under the default (fast) a multiply and an add fuse whenever both carry the `contract` flag -- including inside the
// backend's expansion of logf & co., whose flags depend on which identical calls the optimiser happened to merge:
// the same op then rounds differently from one compilation context to the next (seen: 1 ulp in log, 2e-5 in the
// digamma series between the ahead-of-time interpreter and a run-time compiled program).  With contraction off an
// elementwise op returns the same bits wherever it is compiled; fused multiply-adds that are WANTED are written
// hb_fma().
#pragma clang fp contract(off)

template <typename T>
__device__ __forceinline__ T hb_digamma(T x) {
  // recurrence up to x >= 6, then asymptotic series
  T r = T(0);
  while (x < T(6)) {
    r -= T(1) / x;
    x += T(1);
  }
  const T f = T(1) / (x * x);
  return r + hb_log(x) - T(0.5) / x -
         f * (T(1.0 / 12) - f * (T(1.0 / 120) - f * (T(1.0 / 252) - f * (T(1.0 / 240) - f * T(1.0 / 132)))));
}

template <typename T>
__device__ __forceinline__ void ew_apply(int op, T a, T b, T c, T d, const double* p,
                                         T& o0, T& o1, T& o2) {
  switch (op) {
    // ---- unary ----
    case HB_EW_NEG: o0 = -a; break;
    case HB_EW_EXP: o0 = hb_exp(a); break;
    case HB_EW_LOG: o0 = hb_log(a); break;
    case HB_EW_SQRT: o0 = hb_sqrt(a); break;
    case HB_EW_SQUARE: o0 = a * a; break;
    case HB_EW_ABS: o0 = hb_abs(a); break;
    case HB_EW_SIGN: o0 = hb_sign(a); break;
    case HB_EW_SIGMOID: o0 = hb_sigmoid(a); break;
    case HB_EW_RELU: o0 = a > T(0) ? a : T(0); break;
    case HB_EW_SOFTPLUS: o0 = hb_softplus(a); break;
    case HB_EW_TANH: o0 = hb_tanh(a); break;
    case HB_EW_RECIP: o0 = T(1) / a; break;
    case HB_EW_RSQRT: o0 = T(1) / hb_sqrt(a); break;
    case HB_EW_STEP: o0 = a > T(0) ? T(1) : T(0); break;
    case HB_EW_AFFINE: o0 = hb_fma(T(p[0]), a, T(p[1])); break;
    case HB_EW_CLIP: o0 = a < T(p[0]) ? T(p[0]) : (a > T(p[1]) ? T(p[1]) : a); break;
    case HB_EW_CLIPMASK: o0 = (a >= T(p[0]) && a <= T(p[1])) ? T(1) : T(0); break;
    case HB_EW_LGAMMA: o0 = hb_lgamma(a); break;
    case HB_EW_POWC: o0 = hb_pow(a, T(p[0])); break;
    case HB_EW_LOG1P: o0 = hb_log1p(a); break;
    case HB_EW_COPY: o0 = a; break;
    case HB_EW_DIGAMMA: o0 = hb_digamma(a); break;
    // ---- binary ----
    case HB_EW_ADD: o0 = a + b; break;
    case HB_EW_SUB: o0 = a - b; break;
    case HB_EW_MUL: o0 = a * b; break;
    case HB_EW_DIV: o0 = a / b; break;
    case HB_EW_MAX: o0 = a > b ? a : b; break;
    case HB_EW_MIN: o0 = a < b ? a : b; break;
    case HB_EW_POW: o0 = hb_pow(a, b); break;
    case HB_EW_GT: o0 = a > b ? T(1) : T(0); break;
    case HB_EW_GE: o0 = a >= b ? T(1) : T(0); break;
    case HB_EW_LT: o0 = a < b ? T(1) : T(0); break;
    case HB_EW_LE: o0 = a <= b ? T(1) : T(0); break;
    case HB_EW_EQ: o0 = a == b ? T(1) : T(0); break;
    case HB_EW_SIGMOID_GRAD: o0 = b * a * (T(1) - a); break;           // a = y, b = g
    case HB_EW_TANH_GRAD: o0 = b * (T(1) - a * a); break;              // a = y, b = g
    case HB_EW_RELU_GRAD: o0 = a > T(0) ? b : T(0); break;             // a = x, b = g
    case HB_EW_SOFTPLUS_GRAD: o0 = b * hb_sigmoid(a); break;           // a = x, b = g
    case HB_EW_CLIP_GRAD: o0 = (a >= T(p[0]) && a <= T(p[1])) ? b : T(0); break;  // a = x, b = g
    // ---- ternary ----
    case HB_EW_WHERE: o0 = a != T(0) ? b : c; break;
    case HB_EW_FMA: o0 = hb_fma(a, b, c); break;
    case HB_EW_GAUSS_LOGPDF: {
      // densities.gaussian(x=a, mu=b, var=c)   (reference densities.py:25-27)
      const T dlt = b - a;
      o0 = T(-0.91893853320467274178) - T(0.5) * hb_log(c) - T(0.5) * dlt * dlt / c;
    } break;
    // ---- 4 in, 3 out ----
    case HB_EW_GAUSS_LOGPDF_GRAD: {
      // a = x, b = mu, c = var, d = upstream g
      const T dlt = b - a;  // mu - x
      const T iv = T(1) / c;
      o0 = d * dlt * iv;                                   // d/dx
      o1 = -d * dlt * iv;                                  // d/dmu
      o2 = d * (T(-0.5) * iv + T(0.5) * dlt * dlt * iv * iv);  // d/dvar
    } break;
    default: o0 = T(0); break;
  }
}

#pragma clang fp contract(fast)
#endif  // HB_EW_APPLY_CUH
