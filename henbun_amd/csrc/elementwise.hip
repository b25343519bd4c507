// Generic tensor plumbing kernels: n-ary broadcast elementwise, reductions,
// strided copies, row gather, small matrix utilities.
//
// These replace the ~10^2 tiny TensorFlow elementwise ops the reference builds
// per step; they are launch-bound at the sizes on the path and
// are written for correctness + coalescing, not for a roofline.
#include "common.cuh"
#include "rng_pairs.cuh"
#include "side_jobs.cuh"  // GatherMultiArgs, gather_draw_body, the side-job list
#include <string.h>
#include "../../include/henbun_hip.h"

// ---------------------------------------------------------------------------
// n-ary broadcast elementwise
// ---------------------------------------------------------------------------
struct EwArgs {
  int ndim;
  int nin, nout;
  int op;
  long n;
  int shape[HB_MAX_DIMS];
  long istride[4][HB_MAX_DIMS];
  const void* in[4];
  void* out[3];
  double p[4];
};

#include "ew_apply.cuh"  // hb_digamma, ew_apply

template <typename T>
__global__ void __launch_bounds__(256) ew_kernel(EwArgs A) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += stride) {
    long off[4] = {0, 0, 0, 0};
    long r = i;
#pragma unroll
    for (int d = HB_MAX_DIMS - 1; d >= 0; --d) {
      if (d < A.ndim) {
        const long q = r / A.shape[d];
        const long c = r - q * A.shape[d];
        r = q;
#pragma unroll
        for (int k = 0; k < 4; ++k) off[k] += c * A.istride[k][d];
      }
    }
    // unconditional loads (an absent operand re-reads operand 0) + select: conditional loads compile to a branch
    // and a full vmcnt wait each, i.e. one serialised memory round trip per operand
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int kk = k < A.nin ? k : 0;
      const T ld = ((const T*)A.in[kk])[off[kk]];
      v[k] = k < A.nin ? ld : T(0);
    }
    T o0 = T(0), o1 = T(0), o2 = T(0);
    ew_apply<T>(A.op, v[0], v[1], v[2], v[3], A.p, o0, o1, o2);
    ((T*)A.out[0])[i] = o0;
    if (A.nout > 1) ((T*)A.out[1])[i] = o1;
    if (A.nout > 2) ((T*)A.out[2])[i] = o2;
  }
}

template <typename T>
static int ew_launch(int op, int nin, const void* const* in, const long* istrides, int nout,
                     void* const* out, int ndim, const long* shape, const double* params,
                     hipStream_t stream) {
  HB_REQUIRE(nin >= 1 && nin <= 4, "hb_ewise: nin=%d out of range", nin);
  HB_REQUIRE(nout >= 1 && nout <= 3, "hb_ewise: nout=%d out of range", nout);
  HB_REQUIRE(ndim >= 0 && ndim <= HB_MAX_DIMS, "hb_ewise: ndim=%d out of range", ndim);
  EwArgs A;
  A.op = op;
  A.nin = nin;
  A.nout = nout;
  long n = 1;
  // collapse dims: merge dim d into d-1 when every input is mergeable
  // (stride[d-1] == stride[d]*shape[d], which also covers 0/0 broadcast).
  long shp[HB_MAX_DIMS];
  long st[4][HB_MAX_DIMS];
  int nd = 0;
  for (int d = 0; d < ndim; ++d) {
    HB_REQUIRE(shape[d] >= 0, "hb_ewise: negative dim");
    n *= shape[d];
    if (shape[d] == 1) continue;  // size-1 dims carry no information
    bool merge = nd > 0;
    if (merge) {
      for (int k = 0; k < nin; ++k)
        if (st[k][nd - 1] != istrides[k * ndim + d] * shape[d]) merge = false;
    }
    if (merge) {
      shp[nd - 1] *= shape[d];
      for (int k = 0; k < nin; ++k) st[k][nd - 1] = istrides[k * ndim + d];
    } else {
      shp[nd] = shape[d];
      for (int k = 0; k < nin; ++k) st[k][nd] = istrides[k * ndim + d];
      ++nd;
    }
  }
  if (n == 0) return 0;
  A.ndim = nd;
  A.n = n;
  for (int d = 0; d < HB_MAX_DIMS; ++d) {
    A.shape[d] = d < nd ? (int)shp[d] : 1;
    for (int k = 0; k < 4; ++k) A.istride[k][d] = (k < nin && d < nd) ? st[k][d] : 0;
  }
  for (int k = 0; k < 4; ++k) A.in[k] = k < nin ? in[k] : nullptr;
  for (int k = 0; k < 3; ++k) A.out[k] = k < nout ? out[k] : nullptr;
  for (int k = 0; k < 4; ++k) A.p[k] = params ? params[k] : 0.0;
  hipLaunchKernelGGL(ew_kernel<T>, dim3(hb_stream_grid(n, 256)), dim3(256), 0, stream, A);
  HB_LAUNCH_CHECK();
  return 0;
}

extern "C" int hb_ewise_f32(int op, int nin, const void* const* in, const long* istrides, int nout,
                            void* const* out, int ndim, const long* shape, const double* params,
                            void* stream) {
  return ew_launch<float>(op, nin, in, istrides, nout, out, ndim, shape, params, (hipStream_t)stream);
}
extern "C" int hb_ewise_f64(int op, int nin, const void* const* in, const long* istrides, int nout,
                            void* const* out, int ndim, const long* shape, const double* params,
                            void* stream) {
  return ew_launch<double>(op, nin, in, istrides, nout, out, ndim, shape, params, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// fused elementwise program: a whole cluster of elementwise graph nodes (same
// broadcast iteration space) evaluated by ONE launch.  Each thread interprets a
// short register program for its element; inputs are read with broadcast
// strides, outputs whose shape is smaller than the iteration space are written
// only by the threads whose index along the broadcast dims is 0.
// On the ELBO path these chains act on scalars or [1,n] rows and are purely
// launch-bound (~4.5 us per launch in a hipGraph), so collapsing ~30 launches
// into a handful matters more than per-element speed.
// ---------------------------------------------------------------------------
#include "ew_prog.cuh"  // HB_PROG_* limits, struct ProgArgs
#include "chain.cuh"    // serial chains: the one-workgroup likelihood head may be recorded instead of launched

template <typename T>
__device__ __forceinline__ void ew_prog_body(const ProgArgs& A) {
  __shared__ T regs[HB_PROG_MAX_REGS][256];
  __shared__ T red_smem[16];
  T racc[HB_PROG_MAX_OUT];
#pragma unroll
  for (int k = 0; k < HB_PROG_MAX_OUT; ++k) racc[k] = T(0);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += stride) {
    int idx[HB_PROG_MAX_DIMS] = {0, 0, 0, 0};
    long r = i;
#pragma unroll
    for (int d = HB_PROG_MAX_DIMS - 1; d >= 0; --d) {
      if (d < A.ndim) {
        const long q = r / A.shape[d];
        idx[d] = (int)(r - q * A.shape[d]);
        r = q;
      }
    }
    // the register file lives in LDS ([reg][thread]: conflict-free), not in a private array: runtime-
    // indexed private arrays go to scratch memory, whose set-up alone doubled the launch cost
    T(*reg_)[256] = regs;
#define reg(r_) reg_[r_][threadIdx.x]
    {
      // every input load is issued before the first one is consumed (a load -> LDS-store loop pays one dependent
      // memory round trip per input: ~3 us of the ~8 us these launches used to take)
      T rin[HB_PROG_MAX_IN];
#pragma unroll
      for (int k = 0; k < HB_PROG_MAX_IN; ++k) {
        const int kk = k < A.nin ? k : 0;
        long off = 0;
#pragma unroll
        for (int d = 0; d < HB_PROG_MAX_DIMS; ++d) off += (long)idx[d] * A.istr[kk][d];
        rin[k] = ((const T*)A.in[kk])[off];
      }
#pragma unroll
      for (int k = 0; k < HB_PROG_MAX_IN; ++k)
        if (k < A.nin) reg(k) = rin[k];
    }
    for (int q = 0; q < A.ninstr; ++q) {
      const int op = A.code[q][0], dst = A.code[q][1];
      const T a = reg(A.code[q][2]), b = reg(A.code[q][3]), c = reg(A.code[q][4]);
      // the 4-input op carries its 4th operand's register number in params[q][0]
      const T d = (op == HB_EW_GAUSS_LOGPDF_GRAD) ? reg((int)A.params[q][0]) : T(0);
      T o0 = T(0), o1 = T(0), o2 = T(0);
      ew_apply<T>(op, a, b, c, d, A.params[q], o0, o1, o2);
      reg(dst) = o0;
      if (op == HB_EW_GAUSS_LOGPDF_GRAD) {
        reg(dst + 1) = o1;
        reg(dst + 2) = o2;
      }
    }
#pragma unroll
    for (int k = 0; k < HB_PROG_MAX_OUT; ++k) {
      if (k >= A.nout) break;
      if (A.out_reg[k] & HB_PROG_SUM) {
        racc[k] += reg(A.out_reg[k] & (HB_PROG_SUM - 1));
        continue;
      }
      long off = 0;
      bool write = true;
#pragma unroll
      for (int d = 0; d < HB_PROG_MAX_DIMS; ++d) {
        if (d < A.ndim) {
          if (A.ostr[k][d] == 0 && A.shape[d] > 1 && idx[d] != 0) write = false;  // broadcast dim: index 0 writes
          off += (long)idx[d] * A.ostr[k][d];
        }
      }
      if (write) ((T*)A.out[k])[off] = reg(A.out_reg[k]);
    }
  }
#undef reg
  // sum-reduced outputs: the launcher runs such programs as ONE workgroup, so a block reduction finishes them
#pragma unroll
  for (int k = 0; k < HB_PROG_MAX_OUT; ++k) {
    if (k < A.nout && (A.out_reg[k] & HB_PROG_SUM)) {
      const T s = block_sum(racc[k], red_smem);
      if (threadIdx.x == 0) ((T*)A.out[k])[0] = s;
    }
  }
}

// By-value form (one-off launches): the interpreter walks the ~1.5 KB descriptor in the kernarg segment with
// dependent scalar loads, one cold line after another (measured ~8.5 us per launch, 4 us above a trivial kernel).
template <typename T>
__global__ void __launch_bounds__(256) ew_prog_kernel(ProgArgs A) {
  ew_prog_body<T>(A);
}
// Device-resident form (replayed plans): the descriptor was uploaded once; the workgroup copies it into LDS with ONE
// parallel round trip and interprets from there.
template <typename T>
__global__ void __launch_bounds__(256) ew_prog_image_kernel(const unsigned int* __restrict__ image) {
  __shared__ __attribute__((aligned(16))) unsigned int simg[(sizeof(ProgArgs) + 3) / 4];
  constexpr int NW = (int)((sizeof(ProgArgs) + 3) / 4);
#pragma unroll
  for (int i = 0; i < (NW + 255) / 256; ++i) {
    const int k = threadIdx.x + 256 * i;
    if (k < NW) simg[k] = image[k];
  }
  __syncthreads();
  ew_prog_body<T>(*reinterpret_cast<const ProgArgs*>(simg));
}

// validate + fill the descriptor; *reduces_out: the program has sum-reduced outputs (single workgroup)
static int ew_prog_fill(ProgArgs& A, bool* reduces_out, int ninstr, const int* code, const double* params, int nin,
                        const void* const* in, const long* istrides, int nout, void* const* out, const int* out_regs,
                        const long* ostrides, int ndim, const long* shape) {
  HB_REQUIRE(ninstr >= 1 && ninstr <= HB_PROG_MAX_INSTR, "hb_ewise_prog: %d instructions (max %d)", ninstr,
             HB_PROG_MAX_INSTR);
  HB_REQUIRE(nin >= 0 && nin <= HB_PROG_MAX_IN && nout >= 1 && nout <= HB_PROG_MAX_OUT, "hb_ewise_prog: bad nin/nout");
  HB_REQUIRE(ndim >= 0 && ndim <= HB_PROG_MAX_DIMS, "hb_ewise_prog: ndim=%d out of range", ndim);
  memset(&A, 0, sizeof(A));
  A.ninstr = ninstr; A.nin = nin; A.nout = nout; A.ndim = ndim;
  long n = 1;
  for (int d = 0; d < HB_PROG_MAX_DIMS; ++d) {
    A.shape[d] = d < ndim ? (int)shape[d] : 1;
    if (d < ndim) {
      HB_REQUIRE(shape[d] >= 0, "hb_ewise_prog: negative dim");
      n *= shape[d];
    }
  }
  A.n = n;
  *reduces_out = false;
  if (n == 0) return 0;
  for (int k = 0; k < HB_PROG_MAX_IN; ++k) {
    A.in[k] = k < nin ? in[k] : nullptr;
    for (int d = 0; d < HB_PROG_MAX_DIMS; ++d) A.istr[k][d] = (k < nin && d < ndim) ? istrides[k * ndim + d] : 0;
  }
  for (int k = 0; k < HB_PROG_MAX_OUT; ++k) {
    A.out[k] = k < nout ? out[k] : nullptr;
    A.out_reg[k] = k < nout ? out_regs[k] : 0;
    for (int d = 0; d < HB_PROG_MAX_DIMS; ++d) A.ostr[k][d] = (k < nout && d < ndim) ? ostrides[k * ndim + d] : 0;
  }
  int maxreg = nin;
  for (int q = 0; q < ninstr; ++q) {
    for (int f = 0; f < 5; ++f) A.code[q][f] = (short)code[q * 5 + f];
    A.params[q][0] = params[q * 2];
    A.params[q][1] = params[q * 2 + 1];
    const int extra = code[q * 5] == HB_EW_GAUSS_LOGPDF_GRAD ? 2 : 0;
    HB_REQUIRE(code[q * 5 + 1] >= 0 && code[q * 5 + 1] + extra < HB_PROG_MAX_REGS, "hb_ewise_prog: register out of range");
    for (int f = 2; f < 5; ++f)
      HB_REQUIRE(code[q * 5 + f] >= 0 && code[q * 5 + f] < HB_PROG_MAX_REGS, "hb_ewise_prog: operand register out of range");
    if (extra) HB_REQUIRE(params[q * 2] >= 0 && params[q * 2] < HB_PROG_MAX_REGS, "hb_ewise_prog: 4th operand register out of range");
    if (code[q * 5 + 1] + extra + 1 > maxreg) maxreg = code[q * 5 + 1] + extra + 1;
  }
  bool reduces = false;
  for (int k = 0; k < nout; ++k) {
    const int r = out_regs[k] & (HB_PROG_SUM - 1);
    HB_REQUIRE(out_regs[k] >= 0 && out_regs[k] < 2 * HB_PROG_SUM && r < maxreg, "hb_ewise_prog: output register never written");
    reduces = reduces || (out_regs[k] & HB_PROG_SUM);
  }
  HB_REQUIRE(!reduces || n <= HB_PROG_SUM_MAX_N, "hb_ewise_prog: sum-reduced outputs need a space of at most %d elements",
             HB_PROG_SUM_MAX_N);
  *reduces_out = reduces;
  return 0;
}

template <typename T>
static int ew_prog_launch(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                          const long* istrides, int nout, void* const* out, const int* out_regs, const long* ostrides,
                          int ndim, const long* shape, hipStream_t stream) {
  ProgArgs A;
  bool reduces;
  int rc = ew_prog_fill(A, &reduces, ninstr, code, params, nin, in, istrides, nout, out, out_regs, ostrides, ndim, shape);
  if (rc || A.n == 0) return rc;
  hipLaunchKernelGGL(ew_prog_kernel<T>, dim3(reduces ? 1 : hb_stream_grid(A.n, 256)), dim3(256), 0, stream, A);
  HB_LAUNCH_CHECK();
  return 0;
}

extern "C" long hb_ewise_prog_image_bytes(void) { return (long)sizeof(ProgArgs); }
extern "C" int hb_ewise_prog_build(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                                   const long* istrides, int nout, void* const* out, const int* out_regs,
                                   const long* ostrides, int ndim, const long* shape, void* image_host, long* n_out,
                                   int* reduces_out) {
  HB_REQUIRE(image_host && n_out && reduces_out, "hb_ewise_prog_build: NULL pointer");
  ProgArgs A;
  bool reduces;
  int rc = ew_prog_fill(A, &reduces, ninstr, code, params, nin, in, istrides, nout, out, out_regs, ostrides, ndim, shape);
  if (rc) return rc;
  memcpy(image_host, &A, sizeof(A));
  *n_out = A.n;
  *reduces_out = reduces ? 1 : 0;
  return 0;
}
template <typename T>
static int ew_prog_run(const void* image_dev, long n, int reduces, hipStream_t stream) {
  HB_REQUIRE(image_dev && n >= 0 && ((uintptr_t)image_dev % 4) == 0, "hb_ewise_prog_run: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ew_prog_image_kernel<T>, dim3(reduces ? 1 : hb_stream_grid(n, 256)), dim3(256), 0, stream,
                     (const unsigned int*)image_dev);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_ewise_prog_run_f32(const void* image_dev, long n, int reduces, void* stream) {
  return ew_prog_run<float>(image_dev, n, reduces, (hipStream_t)stream);
}
extern "C" int hb_ewise_prog_run_f64(const void* image_dev, long n, int reduces, void* stream) {
  return ew_prog_run<double>(image_dev, n, reduces, (hipStream_t)stream);
}

extern "C" int hb_ewise_prog_f32(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                                 const long* istrides, int nout, void* const* out, const int* out_regs,
                                 const long* ostrides, int ndim, const long* shape, void* stream) {
  return ew_prog_launch<float>(ninstr, code, params, nin, in, istrides, nout, out, out_regs, ostrides, ndim, shape,
                               (hipStream_t)stream);
}
extern "C" int hb_ewise_prog_f64(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                                 const long* istrides, int nout, void* const* out, const int* out_regs,
                                 const long* ostrides, int ndim, const long* shape, void* stream) {
  return ew_prog_launch<double>(ninstr, code, params, nin, in, istrides, nout, out, out_regs, ostrides, ndim, shape,
                                (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// reductions over the middle axis of a contiguous [K1, R, K2] view
// ---------------------------------------------------------------------------
template <typename T, int OP>
__device__ __forceinline__ T red_combine(T a, T b) {
  if (OP == HB_RED_SUM) return a + b;
  return a > b ? a : b;
}
template <typename T, int OP>
__device__ __forceinline__ T red_identity() {
  if (OP == HB_RED_SUM) return T(0);
  return -INFINITY;
}

// K2 == 1: one block per (k1, split); threads stride over R (coalesced).
template <typename T, int OP>
__global__ void __launch_bounds__(256) reduce_rows_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                          long R, int S) {
  __shared__ T smem[16];
  const long k1 = blockIdx.x;
  const int s = blockIdx.y;
  const long chunk = (R + S - 1) / S;
  const long beg = s * chunk;
  long end = beg + chunk;
  if (end > R) end = R;
  const T* row = in + k1 * R;
  T acc = red_identity<T, OP>();
  long i = beg + threadIdx.x;
  // 8 independent loads in flight per thread: a single block reducing a long row is latency-bound
  for (; i + 7 * (long)blockDim.x < end; i += 8 * (long)blockDim.x) {
    T v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = row[i + q * (long)blockDim.x];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = red_combine<T, OP>(acc, v[q]);
  }
  for (; i < end; i += blockDim.x) acc = red_combine<T, OP>(acc, row[i]);
  if (OP == HB_RED_SUM)
    acc = block_sum(acc, smem);
  else
    acc = block_max(acc, smem);
  if (threadIdx.x == 0) out[k1 * S + s] = acc;
}

// K2 > 1: block = 64 columns x 4 row-lanes; grid (ceil(K2/64), K1, S).
template <typename T, int OP>
__global__ void __launch_bounds__(256) reduce_cols_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                          long K1, long R, long K2, int S) {
  __shared__ T smem[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + cx;
  const long k1 = blockIdx.y;
  const int s = blockIdx.z;
  const long chunk = (R + S - 1) / S;
  const long beg = s * chunk;
  long end = beg + chunk;
  if (end > R) end = R;
  T acc = red_identity<T, OP>();
  if (col < K2) {
    const T* base = in + k1 * R * K2 + col;
    // eight rows in flight per step (a one-load-per-iteration loop is bound by the load round trip, not bandwidth)
    long r = beg + ry;
    for (; r + 28 < end; r += 32) {
      T v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = base[(r + 4 * q) * K2];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = red_combine<T, OP>(acc, v[q]);
    }
    for (; r < end; r += 4) acc = red_combine<T, OP>(acc, base[r * K2]);
  }
  smem[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < K2) {
    T v = smem[0][cx];
    v = red_combine<T, OP>(v, smem[1][cx]);
    v = red_combine<T, OP>(v, smem[2][cx]);
    v = red_combine<T, OP>(v, smem[3][cx]);
    // partial layout [S, K1, K2] so that a second pass is again a column reduce
    out[((long)s * K1 + k1) * K2 + col] = v;
  }
}

template <typename T, int OP>
static int reduce_launch(const T* in, T* out, long K1, long R, long K2, T* ws, long ws_elems,
                         hipStream_t stream) {
  if (K1 * K2 == 0) return 0;
  if (R == 0) {
    // empty reduction: fill identity (sum -> 0) through a trivial launch
    HB_HIP(hb_zero_async(out, sizeof(T) * K1 * K2, stream));
    return 0;
  }
  if (K2 == 1) {
    int S = 1;
    if (K1 < 64 && R > 16384) {
      S = (int)(R / 4096);
      long cap = 256 / (K1 > 0 ? K1 : 1);
      if (cap < 1) cap = 1;
      if (S > cap) S = (int)cap;
      if ((long)S * K1 > ws_elems) S = 1;
    }
    if (S > 1) {
      HB_REQUIRE(ws != nullptr, "hb_reduce: workspace required");
      hipLaunchKernelGGL((reduce_rows_kernel<T, OP>), dim3(K1, S), dim3(256), 0, stream, in, ws, R, S);
      HB_LAUNCH_CHECK();
      hipLaunchKernelGGL((reduce_rows_kernel<T, OP>), dim3(K1, 1), dim3(256), 0, stream, ws, out, (long)S, 1);
    } else {
      hipLaunchKernelGGL((reduce_rows_kernel<T, OP>), dim3(K1, 1), dim3(256), 0, stream, in, out, R, 1);
    }
    HB_LAUNCH_CHECK();
    return 0;
  }
  const int gx = hb_cdiv(K2, 64);
  int S = 1;
  const long blocks = (long)gx * K1;
  if (blocks < 128 && R > 2048) {
    S = (int)(R / 512);
    long cap = 512 / blocks;
    if (cap < 1) cap = 1;
    if (S > cap) S = (int)cap;
    if ((long)S * K1 * K2 > ws_elems) S = 1;
  }
  HB_REQUIRE(K1 <= 65535, "hb_reduce: K1=%ld too large for grid.y", K1);
  if (S > 1) {
    HB_REQUIRE(ws != nullptr, "hb_reduce: workspace required");
    hipLaunchKernelGGL((reduce_cols_kernel<T, OP>), dim3(gx, K1, S), dim3(256), 0, stream, in, ws, K1, R, K2, S);
    HB_LAUNCH_CHECK();
    // second pass: [S, K1*K2] column reduce over S
    const long KK = K1 * K2;
    hipLaunchKernelGGL((reduce_cols_kernel<T, OP>), dim3(hb_cdiv(KK, 64), 1, 1), dim3(256), 0, stream, ws, out,
                       1L, (long)S, KK, 1);
  } else {
    hipLaunchKernelGGL((reduce_cols_kernel<T, OP>), dim3(gx, K1, 1), dim3(256), 0, stream, in, out, K1, R, K2, 1);
  }
  HB_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int reduce_dispatch(int op, const T* in, T* out, long K1, long R, long K2, T* ws, long ws_elems,
                           hipStream_t stream) {
  HB_REQUIRE(K1 >= 0 && R >= 0 && K2 >= 0, "hb_reduce: negative extent");
  if (op == HB_RED_SUM) return reduce_launch<T, HB_RED_SUM>(in, out, K1, R, K2, ws, ws_elems, stream);
  if (op == HB_RED_MAX) return reduce_launch<T, HB_RED_MAX>(in, out, K1, R, K2, ws, ws_elems, stream);
  HB_REQUIRE(false, "hb_reduce: unknown op %d", op);
}

extern "C" int hb_reduce_f32(int op, const float* in, float* out, long K1, long R, long K2, float* ws,
                             long ws_elems, void* stream) {
  return reduce_dispatch<float>(op, in, out, K1, R, K2, ws, ws_elems, (hipStream_t)stream);
}
extern "C" int hb_reduce_f64(int op, const double* in, double* out, long K1, long R, long K2, double* ws,
                             long ws_elems, void* stream) {
  return reduce_dispatch<double>(op, in, out, K1, R, K2, ws, ws_elems, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// strided n-d copy (transpose / slice / broadcast / concat placement)
// ---------------------------------------------------------------------------
struct CopyArgs {
  int ndim;
  long n;
  int shape[HB_MAX_DIMS];
  long is[HB_MAX_DIMS], os[HB_MAX_DIMS];
};

template <typename T>
__global__ void __launch_bounds__(256) copy_nd_kernel(const T* __restrict__ in, T* __restrict__ out, CopyArgs A) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += stride) {
    long r = i, io = 0, oo = 0;
#pragma unroll
    for (int d = HB_MAX_DIMS - 1; d >= 0; --d) {
      if (d < A.ndim) {
        const long q = r / A.shape[d];
        const long c = r - q * A.shape[d];
        r = q;
        io += c * A.is[d];
        oo += c * A.os[d];
      }
    }
    out[oo] = in[io];
  }
}

template <typename T>
static int copy_nd_launch(const T* in, const long* istr, T* out, const long* ostr, int ndim, const long* shape,
                          hipStream_t stream) {
  HB_REQUIRE(ndim >= 0 && ndim <= HB_MAX_DIMS, "hb_copy_nd: ndim=%d out of range", ndim);
  CopyArgs A;
  long n = 1;
  int nd = 0;
  for (int d = 0; d < ndim; ++d) {
    HB_REQUIRE(shape[d] >= 0, "hb_copy_nd: negative dim");
    n *= shape[d];
    if (shape[d] == 1) continue;
    if (nd > 0 && A.is[nd - 1] == istr[d] * shape[d] && A.os[nd - 1] == ostr[d] * shape[d]) {
      A.shape[nd - 1] *= (int)shape[d];
      A.is[nd - 1] = istr[d];
      A.os[nd - 1] = ostr[d];
    } else {
      A.shape[nd] = (int)shape[d];
      A.is[nd] = istr[d];
      A.os[nd] = ostr[d];
      ++nd;
    }
  }
  if (n == 0) return 0;
  for (int d = nd; d < HB_MAX_DIMS; ++d) {
    A.shape[d] = 1;
    A.is[d] = A.os[d] = 0;
  }
  A.ndim = nd;
  A.n = n;
  hipLaunchKernelGGL(copy_nd_kernel<T>, dim3(hb_stream_grid(n, 256)), dim3(256), 0, stream, in, out, A);
  HB_LAUNCH_CHECK();
  return 0;
}

extern "C" int hb_copy_nd_f32(const float* in, const long* istr, float* out, const long* ostr, int ndim,
                              const long* shape, void* stream) {
  return copy_nd_launch<float>(in, istr, out, ostr, ndim, shape, (hipStream_t)stream);
}
extern "C" int hb_copy_nd_f64(const double* in, const long* istr, double* out, const long* ostr, int ndim,
                              const long* shape, void* stream) {
  return copy_nd_launch<double>(in, istr, out, ostr, ndim, shape, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// fill
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) fill_kernel(T* out, long n, T v) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}
extern "C" int hb_fill_f32(float* out, long n, double v, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fill_kernel<float>, dim3(hb_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, out, n, (float)v);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_fill_f64(double* out, long n, double v, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fill_kernel<double>, dim3(hb_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, out, n, v);
  HB_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// K0: device-resident minibatch gather  dst[i,:] = src[perm[idx[i]],:]
// (replaces the host fancy-index + H2D of reference param.py:733-739)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gather_rows_kernel(const T* __restrict__ src, const long* __restrict__ idx,
                                                          const long* __restrict__ perm, T* __restrict__ dst,
                                                          long n, long row, long nsrc, int* __restrict__ err) {
  const long total = n * row;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / row, c = i - r * row;
    long j = idx[r];
    if (perm) j = perm[j];
    if (j < 0 || j >= nsrc) {
      if (err) *err = 1;
      dst[i] = T(0);
    } else {
      dst[i] = src[j * row + c];
    }
  }
}
extern "C" int hb_gather_rows_f32(const float* src, long nsrc, long row, const long* idx, const long* perm, long n,
                                  float* dst, int* err, void* stream) {
  HB_REQUIRE(n >= 0 && row >= 0 && nsrc >= 0, "hb_gather_rows: negative extent");
  if (n * row == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(hb_stream_grid(n * row, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, idx, perm, dst, n, row, nsrc, err);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_gather_rows_f64(const double* src, long nsrc, long row, const long* idx, const long* perm, long n,
                                  double* dst, int* err, void* stream) {
  HB_REQUIRE(n >= 0 && row >= 0 && nsrc >= 0, "hb_gather_rows: negative extent");
  if (n * row == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel<double>, dim3(hb_stream_grid(n * row, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, idx, perm, dst, n, row, nsrc, err);
  HB_LAUNCH_CHECK();
  return 0;
}

// Several arrays gathered by the same index vector in ONE launch (a model's MinibatchData arrays all take the same
// rows; one launch per array is ~4 us of kernel boundary each).  Up to HB_GATHER_MAX arrays, each with its own
// row width; all arrays have `nsrc` rows.
template <typename T>
__global__ void __launch_bounds__(256) gather_rows_multi_kernel(GatherMultiArgs<T> g, const long* __restrict__ idx,
                                                                const long* __restrict__ perm, long n, long nsrc,
                                                                int* __restrict__ err) {
  const long total = g.start[g.narr];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int a = 0;
#pragma unroll
    for (int q = 1; q < HB_GATHER_MAX; ++q)
      if (q < g.narr && i >= g.start[q]) a = q;
    const long li = i - g.start[a];
    const long r = li / g.row[a], c = li - r * g.row[a];
    long j = idx[r];
    if (perm) j = perm[j];
    if (j < 0 || j >= nsrc) {
      if (err) *err = 1;
      g.dst[a][li] = T(0);
    } else {
      g.dst[a][li] = g.src[a][j * g.row[a] + c];
    }
  }
}
template <typename T>
static int gather_rows_multi(int narr, const T* const* srcs, const long* rows, T* const* dsts, long nsrc,
                             const long* idx, const long* perm, long n, int* err, hipStream_t stream) {
  HB_REQUIRE(narr >= 1 && narr <= HB_GATHER_MAX, "hb_gather_rows_multi: %d arrays (max %d)", narr, HB_GATHER_MAX);
  HB_REQUIRE(n >= 0 && nsrc >= 0 && srcs && rows && dsts && idx, "hb_gather_rows_multi: bad arguments");
  GatherMultiArgs<T> g;
  g.narr = narr;
  g.start[0] = 0;
  for (int a = 0; a < HB_GATHER_MAX; ++a) {
    g.src[a] = a < narr ? srcs[a] : nullptr;
    g.dst[a] = a < narr ? dsts[a] : nullptr;
    g.row[a] = a < narr ? rows[a] : 1;
    if (a < narr) HB_REQUIRE(rows[a] >= 1 && srcs[a] && dsts[a], "hb_gather_rows_multi: bad array %d", a);
    g.start[a + 1] = g.start[a] + (a < narr ? n * rows[a] : 0);
  }
  if (g.start[narr] == 0) return 0;
  hipLaunchKernelGGL(gather_rows_multi_kernel<T>, dim3(hb_stream_grid(g.start[narr], 256)), dim3(256), 0, stream, g, idx,
                     perm, n, nsrc, err);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_gather_rows_multi_f32(int narr, const float* const* srcs, const long* rows, float* const* dsts,
                                        long nsrc, const long* idx, const long* perm, long n, int* err, void* stream) {
  return gather_rows_multi<float>(narr, srcs, rows, dsts, nsrc, idx, perm, n, err, (hipStream_t)stream);
}
extern "C" int hb_gather_rows_multi_f64(int narr, const double* const* srcs, const long* rows, double* const* dsts,
                                        long nsrc, const long* idx, const long* perm, long n, int* err, void* stream) {
  return gather_rows_multi<double>(narr, srcs, rows, dsts, nsrc, idx, perm, n, err, (hipStream_t)stream);
}

// The minibatch draw and the gather in one launch (n <= nlanes): G threads per row; the group's first thread owns
// RNG lane r -- it draws idx[r] exactly as rng_randint_kernel does (one value per lane, state advanced once), stores
// it for inspection, and hands it to the group, which then copies row perm[idx[r]] of every array.
template <typename T, int G>
__global__ void __launch_bounds__(256) gather_rows_multi_draw_kernel(GatherMultiArgs<T> g, uint64_t* __restrict__ state,
                                                                     long nlanes, long lo, uint64_t range,
                                                                     long* __restrict__ idx_out,
                                                                     const long* __restrict__ perm, long n, long nsrc,
                                                                     int* __restrict__ err) {
  gather_draw_body<T>(g, state, nlanes, lo, range, idx_out, perm, n, nsrc, err, G, (long)blockIdx.x);
}
template <typename T>
static int gather_rows_multi_draw(int narr, const T* const* srcs, const long* rows, T* const* dsts, long nsrc,
                                  uint64_t* state, long nlanes, long lo, long hi, long* idx_out, const long* perm,
                                  long n, int* err, hipStream_t stream, bool defer = false) {
  HB_REQUIRE(narr >= 1 && narr <= HB_GATHER_MAX, "hb_gather_rows_multi_draw: %d arrays (max %d)", narr, HB_GATHER_MAX);
  HB_REQUIRE(n >= 0 && nsrc >= 0 && srcs && rows && dsts && state && idx_out, "hb_gather_rows_multi_draw: bad arguments");
  HB_REQUIRE(n <= nlanes, "hb_gather_rows_multi_draw: n=%ld exceeds the %ld RNG lanes (draw with hb_rng_randint instead)", n,
             nlanes);
  HB_REQUIRE(hi > lo, "hb_gather_rows_multi_draw: empty range [%ld,%ld)", lo, hi);
  GatherMultiArgs<T> g;
  g.narr = narr;
  g.start[0] = 0;
  long wmax = 1;
  for (int a = 0; a < HB_GATHER_MAX; ++a) {
    g.src[a] = a < narr ? srcs[a] : nullptr;
    g.dst[a] = a < narr ? dsts[a] : nullptr;
    g.row[a] = a < narr ? rows[a] : 1;
    if (a < narr) HB_REQUIRE(rows[a] >= 1 && srcs[a] && dsts[a], "hb_gather_rows_multi_draw: bad array %d", a);
    if (a < narr && rows[a] > wmax) wmax = rows[a];
    g.start[a + 1] = g.start[a] + (a < narr ? n * rows[a] : 0);
  }
  if (n == 0) return 0;
  const uint64_t range = (uint64_t)(hi - lo);
  // rows of whole 16-byte groups at aligned addresses: a lane moves 16 bytes (group size counted in 16-byte units,
  // passed NEGATIVE to the body)
  constexpr long VEC = 16 / (long)sizeof(T);
  bool vec16 = true;
  for (int a = 0; a < narr; ++a)
    vec16 = vec16 && rows[a] % VEC == 0 && ((uintptr_t)srcs[a] % 16) == 0 && ((uintptr_t)dsts[a] % 16) == 0;
  const long wsel = vec16 ? wmax / VEC : wmax;
  const int Gsel = wsel <= 2 ? 1 : (wsel <= 8 ? 4 : (wsel <= 32 ? 16 : 64));
  if constexpr (sizeof(T) == 4) {
    if (defer) {   // recorded for the next host launch (side_jobs.cuh)
      HbSideJob j;
      j.kind = HB_SIDE_GATHER_DRAW;
      j.gather.G = vec16 ? -Gsel : Gsel;
      j.nblocks = hb_cdiv(n * Gsel, 256);
      if (j.nblocks <= 1024) {
        memcpy(&j.gather.g, &g, sizeof(g));
        j.gather.state = state; j.gather.nlanes = nlanes; j.gather.lo = lo; j.gather.range = range;
        j.gather.idx_out = idx_out; j.gather.perm = perm; j.gather.n = n; j.gather.nsrc = nsrc; j.gather.err = err;
        return hb_side_push(j, stream);
      }
    }
  }
#define HB_GDRAW(G_)                                                                                                   \
  hipLaunchKernelGGL((gather_rows_multi_draw_kernel<T, G_>), dim3((unsigned)hb_cdiv(n * (G_ < 0 ? -(G_) : (G_)), 256)), dim3(256), 0, \
                     stream, g, state, nlanes, lo, range, idx_out, perm, n, nsrc, err)
  if (vec16) {
    if (Gsel == 1)
      HB_GDRAW(-1);
    else if (Gsel == 4)
      HB_GDRAW(-4);
    else if (Gsel == 16)
      HB_GDRAW(-16);
    else
      HB_GDRAW(-64);
  } else if (Gsel == 1)
    HB_GDRAW(1);
  else if (Gsel == 4)
    HB_GDRAW(4);
  else if (Gsel == 16)
    HB_GDRAW(16);
  else
    HB_GDRAW(64);
#undef HB_GDRAW
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_gather_rows_multi_draw_f32(int narr, const float* const* srcs, const long* rows, float* const* dsts,
                                             long nsrc, uint64_t* state, long nlanes, long lo, long hi, long* idx_out,
                                             const long* perm, long n, int* err, void* stream) {
  return gather_rows_multi_draw<float>(narr, srcs, rows, dsts, nsrc, state, nlanes, lo, hi, idx_out, perm, n, err,
                                       (hipStream_t)stream);
}
extern "C" int hb_side_push_gather_draw_f32(int narr, const float* const* srcs, const long* rows, float* const* dsts, long nsrc,
                                           uint64_t* state, long nlanes, long lo, long hi, long* idx_out, const long* perm,
                                           long n, int* err, void* stream) {
  return gather_rows_multi_draw<float>(narr, srcs, rows, dsts, nsrc, state, nlanes, lo, hi, idx_out, perm, n, err,
                                       (hipStream_t)stream, true);
}
extern "C" int hb_gather_rows_multi_draw_f64(int narr, const double* const* srcs, const long* rows, double* const* dsts,
                                             long nsrc, uint64_t* state, long nlanes, long lo, long hi, long* idx_out,
                                             const long* perm, long n, int* err, void* stream) {
  return gather_rows_multi_draw<double>(narr, srcs, rows, dsts, nsrc, state, nlanes, lo, hi, idx_out, perm, n, err,
                                        (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// Fused Gaussian log-likelihood head:  ll = sum_j log N(x_j | mu_j, var),  mu_j = f_j * scale   (scale, var scalars)
// together with everything its backward pass needs, in the same pass over the data:
//   dmu_j  = d ll / d mu_j = (x_j - mu_j) / var            [n]
//   dscale = d ll / d scale = sum_j dmu_j f_j               [1]
//   dvar   = d ll / d var  = sum_j (-1/(2 var) + (x_j - mu_j)^2 / (2 var^2))   [1]
// (reference densities.py:25-27 + tf.reduce_sum + TF autodiff: a chain of eight launches when lowered op by op --
// two elementwise programs, three reductions, the 3-output gradient op and their glue).  Two launches here: per-block
// partial sums, then one block folds the three sums (deterministic order).
// ---------------------------------------------------------------------------
#define HB_GLL_BLOCK_ELEMS 1024
template <typename T>
__global__ void __launch_bounds__(256) gauss_ll_kernel(const T* __restrict__ x, const T* __restrict__ f,
                                                       const T* __restrict__ scale, const T* __restrict__ var,
                                                       long n, T* __restrict__ dmu, T* __restrict__ partial,
                                                       T* __restrict__ fbar, T post) {
  __shared__ T smem[16];
  const T s = scale ? scale[0] : T(1), v = var[0];
  const T iv = T(1) / v, lc = T(-0.91893853320467274178) - T(0.5) * hb_log(v);
  T all = T(0), asc = T(0), avr = T(0);
  const long base = (long)blockIdx.x * HB_GLL_BLOCK_ELEMS;
#pragma unroll
  for (int q = 0; q < HB_GLL_BLOCK_ELEMS / 256; ++q) {
    const long j = base + q * 256 + threadIdx.x;
    const long jc = j < n ? j : n - 1;
    const T xv = x[jc], fv = f[jc];
    const T dlt = xv - fv * s;
    const T g = dlt * iv;
    if (j < n) {
      dmu[j] = g;
      if (fbar) fbar[j] = s * (post * g);
      all += lc - T(0.5) * dlt * g;
      asc += g * fv;
      avr += T(-0.5) * iv + T(0.5) * g * g;
    }
  }
  all = block_sum(all, smem);
  asc = block_sum(asc, smem);
  avr = block_sum(avr, smem);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = all;
    partial[gridDim.x + blockIdx.x] = asc;
    partial[2 * gridDim.x + blockIdx.x] = avr;
  }
}
template <typename T>
__global__ void __launch_bounds__(256) gauss_ll_finish_kernel(const T* __restrict__ partial, int nb, T* __restrict__ ll,
                                                              T* __restrict__ dscale, T* __restrict__ dvar) {
  __shared__ T smem[16];
  T a0 = T(0), a1 = T(0), a2 = T(0);
  for (int i = threadIdx.x; i < nb; i += 256) {
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
// n <= HB_GLL_SINGLE_N: one 1024-thread workgroup does the whole head, sums included (a second launch costs more
// than streaming 16 elements per thread from one CU); every load of a thread is in flight before the first use
#define HB_GLL_SINGLE_N 16384
template <typename T>
__global__ void __launch_bounds__(1024) gauss_ll_single_kernel(const T* __restrict__ x, const T* __restrict__ f,
                                                               const T* __restrict__ scale, const T* __restrict__ var,
                                                               long n, T* __restrict__ dmu, T* __restrict__ ll,
                                                               T* __restrict__ dscale, T* __restrict__ dvar,
                                                               T* __restrict__ fbar, T post) {
  __shared__ T smem[16];
  static_assert(HB_GLL_SINGLE_N == 16 * 1024, "hb_gauss_ll_single_body: 16 elements per thread of a 1024-thread workgroup");
  hb_gauss_ll_single_body<T>(x, f, scale, var, n, dmu, ll, dscale, dvar, smem, fbar, post);
}

template <typename T>
static int gauss_ll(const T* x, const T* f, const T* scale, const T* var, long n, T* ll, T* dmu, T* dscale, T* dvar,
                    T* ws, long ws_elems, hipStream_t stream, double post = 0.0, T* fbar = nullptr) {
  HB_REQUIRE(n >= 0 && x && f && var && ll && dmu && dscale && dvar && ws, "hb_gauss_ll: bad arguments");
  const int nb = n > 0 ? hb_cdiv(n, HB_GLL_BLOCK_ELEMS) : 0;
  HB_REQUIRE(ws_elems >= 3L * (nb > 0 ? nb : 1), "hb_gauss_ll: workspace of 3*ceil(n/%d) elements required",
             HB_GLL_BLOCK_ELEMS);
  if (hb_chain_recording()) {
    if (n > 0 && n <= HB_CHAIN_GLL_MAX_N) {
      HbChainJob j;
      j.kind = HB_CHAIN_GLL;
      j.is64 = sizeof(T) == 8;
      j.p[0] = x, j.p[1] = f, j.p[2] = scale, j.p[3] = var, j.p[4] = dmu, j.p[5] = ll, j.p[6] = dscale, j.p[7] = dvar;
      j.p[8] = fbar;
      j.l[0] = n;
      j.d[0] = post;
      return hb_chain_push(j, stream);
    }
    const int crc = hb_chain_flush(stream);
    if (crc) return crc;
  }
  if (n > 0 && n <= HB_GLL_SINGLE_N) {
    hipLaunchKernelGGL(gauss_ll_single_kernel<T>, dim3(1), dim3(1024), 0, stream, x, f, scale, var, n, dmu, ll, dscale,
                       dvar, fbar, (T)post);
    HB_LAUNCH_CHECK();
    return 0;
  }
  if (nb > 0) {
    hipLaunchKernelGGL(gauss_ll_kernel<T>, dim3(nb), dim3(256), 0, stream, x, f, scale, var, n, dmu, ws, fbar, (T)post);
    HB_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(gauss_ll_finish_kernel<T>, dim3(1), dim3(256), 0, stream, ws, nb, ll, dscale, dvar);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_gauss_ll_f32(const float* x, const float* f, const float* scale, const float* var, long n, float* ll,
                               float* dmu, float* dscale, float* dvar, float* ws, long ws_elems, void* stream) {
  return gauss_ll<float>(x, f, scale, var, n, ll, dmu, dscale, dvar, ws, ws_elems, (hipStream_t)stream);
}
extern "C" int hb_gauss_ll_f64(const double* x, const double* f, const double* scale, const double* var, long n,
                               double* ll, double* dmu, double* dscale, double* dvar, double* ws, long ws_elems,
                               void* stream) {
  return gauss_ll<double>(x, f, scale, var, n, ll, dmu, dscale, dvar, ws, ws_elems, (hipStream_t)stream);
}
// hb_gauss_ll_fold: partial[3][nb] -> ll, dscale, dvar (the second half of a head whose per-point part ran elsewhere:
// hb_sgp_fwd_gauss).  Chain-aware: recorded into a serial chain when one is open.
template <typename T>
__global__ void __launch_bounds__(256) gauss_fold_kernel(const T* __restrict__ partial, long nb, T* __restrict__ ll,
                                                         T* __restrict__ dscale, T* __restrict__ dvar) {
  __shared__ T smem[16];
  hb_gauss_fold_body<T>(partial, nb, ll, dscale, dvar, smem);
}
template <typename T>
static int gauss_fold(const T* partial, long nb, T* ll, T* dscale, T* dvar, hipStream_t stream) {
  HB_REQUIRE(partial && nb >= 1 && ll && dscale && dvar, "hb_gauss_ll_fold: bad arguments");
  if (hb_chain_recording()) {
    if (nb <= HB_CHAIN_GLL_FOLD_MAX_N) {
      HbChainJob j;
      j.kind = HB_CHAIN_GLL_FOLD;
      j.is64 = sizeof(T) == 8;
      j.p[0] = partial, j.p[1] = ll, j.p[2] = dscale, j.p[3] = dvar;
      j.l[0] = nb;
      return hb_chain_push(j, stream);
    }
    const int crc = hb_chain_flush(stream);
    if (crc) return crc;
  }
  hipLaunchKernelGGL(gauss_fold_kernel<T>, dim3(1), dim3(256), 0, stream, partial, nb, ll, dscale, dvar);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_gauss_ll_fold_f32(const float* partial, long nb, float* ll, float* dscale, float* dvar, void* stream) {
  return gauss_fold<float>(partial, nb, ll, dscale, dvar, (hipStream_t)stream);
}
extern "C" int hb_gauss_ll_fold_f64(const double* partial, long nb, double* ll, double* dscale, double* dvar, void* stream) {
  return gauss_fold<double>(partial, nb, ll, dscale, dvar, (hipStream_t)stream);
}

extern "C" int hb_gauss_ll_post_f32(const float* x, const float* f, const float* scale, const float* var, long n, float* ll,
                                    float* dmu, float* dscale, float* dvar, double post, float* fbar, float* ws,
                                    long ws_elems, void* stream) {
  HB_REQUIRE(fbar, "hb_gauss_ll_post: fbar is NULL");
  return gauss_ll<float>(x, f, scale, var, n, ll, dmu, dscale, dvar, ws, ws_elems, (hipStream_t)stream, post, fbar);
}
extern "C" int hb_gauss_ll_post_f64(const double* x, const double* f, const double* scale, const double* var, long n,
                                    double* ll, double* dmu, double* dscale, double* dvar, double post, double* fbar,
                                    double* ws, long ws_elems, void* stream) {
  HB_REQUIRE(fbar, "hb_gauss_ll_post: fbar is NULL");
  return gauss_ll<double>(x, f, scale, var, n, ll, dmu, dscale, dvar, ws, ws_elems, (hipStream_t)stream, post, fbar);
}

// ---------------------------------------------------------------------------
// small matrix utilities on batched row-major [B, R, C]
// ---------------------------------------------------------------------------
// mode 0: band part   out = in where (lower<0 || i-j<=lower) && (upper<0 || j-i<=upper) else 0
// mode 1: add alpha to the diagonal (out = in + alpha*I)
// mode 2: Cholesky-gradient Phi: lower triangle with halved diagonal
// mode 3: symmetrise 0.5*(in + in^T) (square only)
// mode 4: symmetric from half the lower triangle: out[i][j] = 0.5 * in[max(i,j)][min(i,j)] (square only)
template <typename T>
__global__ void __launch_bounds__(256) matutil_kernel(const T* __restrict__ in, T* __restrict__ out, long B, long R,
                                                      long C, int mode, long lower, long upper, T alpha) {
  const long total = B * R * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long b = t / (R * C);
    const long rem = t - b * R * C;
    const long i = rem / C, j = rem - i * C;
    T v = in[t];
    if (mode == 0) {
      const bool keep = (lower < 0 || i - j <= lower) && (upper < 0 || j - i <= upper);
      v = keep ? v : T(0);
    } else if (mode == 1) {
      if (i == j) v += alpha;
    } else if (mode == 2) {
      v = (i > j) ? v : (i == j ? T(0.5) * v : T(0));
    } else if (mode == 3) {
      v = T(0.5) * (v + in[b * R * C + j * C + i]);
    } else if (mode == 4) {
      v = T(0.5) * (i >= j ? v : in[b * R * C + j * C + i]);
    }
    out[t] = v;
  }
}
template <typename T>
static int matutil_launch(const T* in, T* out, long B, long R, long C, int mode, long lower, long upper, double alpha,
                          hipStream_t stream) {
  HB_REQUIRE(B >= 0 && R >= 0 && C >= 0, "hb_matutil: negative extent");
  HB_REQUIRE(mode >= 0 && mode <= 4, "hb_matutil: bad mode %d", mode);
  HB_REQUIRE((mode != 3 && mode != 4) || (R == C && in != out), "hb_matutil: symmetrise needs square, out-of-place");
  const long n = B * R * C;
  if (n == 0) return 0;
  hipLaunchKernelGGL(matutil_kernel<T>, dim3(hb_stream_grid(n, 256)), dim3(256), 0, stream, in, out, B, R, C, mode,
                     lower, upper, (T)alpha);
  HB_LAUNCH_CHECK();
  return 0;
}
extern "C" int hb_matutil_f32(const float* in, float* out, long B, long R, long C, int mode, long lower, long upper,
                              double alpha, void* stream) {
  return matutil_launch<float>(in, out, B, R, C, mode, lower, upper, alpha, (hipStream_t)stream);
}
extern "C" int hb_matutil_f64(const double* in, double* out, long B, long R, long C, int mode, long lower, long upper,
                              double alpha, void* stream) {
  return matutil_launch<double>(in, out, B, R, C, mode, lower, upper, alpha, (hipStream_t)stream);
}
