/*
 * henbun_hip.h -- C ABI of libhenbun_hip.so, the MI355X (gfx950) numeric
 * backend of henbun_amd.
 *
 * What this replaces.  The reference (fujii-team/Henbun) has NO native
 * boundary of its own: every numeric op on the ELBO path is a TensorFlow 1.x
 * graph node reached through `session.run` (reference Henbun/model.py:81,96,
 * 225-227,250,265) and the thin shim module Henbun/tf_wraps.py:26-48.  The
 * only native-op precedent is the disabled `tf.load_op_library('tfops/
 * matpackops.so')` hook (Henbun/tf_wraps.py:50-71).  This header is the
 * boundary the north star asks for in its place: a flat `extern "C"` surface,
 * plain device pointers and sizes, loaded from Python with ctypes
 * (henbun_amd/_lib.py).  Each entry cites the reference call site(s) whose TF
 * op(s) it stands in for.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer unless marked (host);
 *     the library allocates nothing persistent; scratch comes in through
 *     `ws`/`ws_elems` (elements of the op's dtype);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous
 *     on it and never synchronises;
 *   - return 0 = ok, <0 = bad argument (message in hb_last_error_string()),
 *     >0 = hipError_t from the runtime;
 *   - numerical failure (non-positive-definite Cholesky) is reported through
 *     a device-side `int* info` (LAPACK convention: k+1 = leading minor k+1
 *     is not positive definite), since calls are asynchronous;
 *   - `_f32` / `_f64` suffix = arithmetic type; layouts are row-major,
 *     contiguous unless strides are given; extents are `long` (int64).
 *   - samplers accept injected noise (`u_in`, nullable) and export the noise
 *     they used (`u_out`), so parity tests never depend on the RNG stream.
 */
#ifndef HENBUN_HIP_H
#define HENBUN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history: 1 = rounds 1-3 (the round that removed hb_cholesky_inverse_sgp_f32 and hb_sgp_finish_* should have
 * bumped it and did not); 2 = round 4: hb_debug_set / hb_debug_clear, hb_cholesky_inverse_ws_elems and the workspace
 * contract of hb_cholesky_inverse_f32 (exchange + sync area, zero-filled once by its owner); added without changing an
 * existing signature: hb_cholesky_persistent_shape, hb_gram_cholesky_inverse_f32, hb_mlp2_sample_*, hb_matmul_gauss_*,
 * hb_matmul_gram_vjp_*, hb_gram_ell_fold_*, hb_sgp_rider_*, hb_fullrank_sample_kl_fwd1_* / hb_fullrank_one_launch_shape. */
#define HB_ABI_VERSION 2

/* ---- runtime ----------------------------------------------------------- */
int hb_version(void);
const char* hb_last_error_string(void);
/* Diagnostic switches (A/B forms of a dispatch rule, forced tile sizes; tests and tools/ use them): a process-wide
 * key -> value table.  The library never reads the environment; without a call here every dispatch rule is the
 * shipped one.  Keys: chol_persist (0: the launch-chain Cholesky), chol_no64, mm_no_wgk, mm_no_rowsreg, mm_no_rows,
 * mm_force_bt, mm_force_s, sgp_no_strip, sgp_force_strip, sgp_tiled_crossover, sgp_strip_form2, sgp_no_fused_finish,
 * lbar_force_s, lbar_no_lds.  hb_debug_clear() drops every entry. */
int hb_debug_set(const char* key, long value);
int hb_debug_clear(void);
/* device name / arch of the current device into (host) buf; returns 0 or hipError */
int hb_device_info(char* buf, int buflen, int* cu_count);

/* hipGraph capture of a launch sequence (replaces the per-op dispatch of
 * session.run, reference model.py:265-266).  exec is a (host) handle. */
int hb_graph_begin_capture(void* stream);
int hb_graph_end_capture(void* stream, void** exec_out);
int hb_graph_launch(void* exec, void* stream);
int hb_graph_destroy(void* exec);

/* ---- generic tensor plumbing (tf elementwise ops, reduce_sum, transpose,
 *      slice, tile; reference call sites listed in SURVEY.md 2.2) ---------- */
enum {
  /* unary */
  HB_EW_NEG = 1, HB_EW_EXP, HB_EW_LOG, HB_EW_SQRT, HB_EW_SQUARE, HB_EW_ABS, HB_EW_SIGN,
  HB_EW_SIGMOID, HB_EW_RELU, HB_EW_SOFTPLUS, HB_EW_TANH, HB_EW_RECIP, HB_EW_RSQRT,
  HB_EW_STEP, HB_EW_AFFINE /* p0*x+p1 */, HB_EW_CLIP /* [p0,p1] */, HB_EW_CLIPMASK,
  HB_EW_LGAMMA, HB_EW_POWC /* x^p0 */, HB_EW_LOG1P, HB_EW_COPY, HB_EW_DIGAMMA,
  /* binary */
  HB_EW_ADD = 32, HB_EW_SUB, HB_EW_MUL, HB_EW_DIV, HB_EW_MAX, HB_EW_MIN, HB_EW_POW,
  HB_EW_GT, HB_EW_GE, HB_EW_LT, HB_EW_LE, HB_EW_EQ,
  HB_EW_SIGMOID_GRAD /* (y,g) */, HB_EW_TANH_GRAD /* (y,g) */, HB_EW_RELU_GRAD /* (x,g) */,
  HB_EW_SOFTPLUS_GRAD /* (x,g) */, HB_EW_CLIP_GRAD /* (x,g) p0,p1 */,
  /* ternary */
  HB_EW_WHERE = 64 /* (c,a,b) */, HB_EW_FMA /* a*b+c */,
  HB_EW_GAUSS_LOGPDF /* (x,mu,var): reference densities.py:25-27 */,
  /* 4 in, 3 out */
  HB_EW_GAUSS_LOGPDF_GRAD = 80 /* (x,mu,var,g) -> (gx,gmu,gvar) */
};

/* out[k][i] = op(in[0][bcast(i)], ...); every output is contiguous with the
 * broadcast `shape[ndim]`; `istrides` is [nin][ndim] in elements (0 = broadcast).
 * in/out/istrides/shape/params are (host) arrays; params has 4 doubles or NULL. */
int hb_ewise_f32(int op, int nin, const void* const* in, const long* istrides, int nout,
                 void* const* out, int ndim, const long* shape, const double* params, void* stream);
int hb_ewise_f64(int op, int nin, const void* const* in, const long* istrides, int nout,
                 void* const* out, int ndim, const long* shape, const double* params, void* stream);

/* Fused Gaussian log-likelihood head (reference densities.py:25-27 under tf.reduce_sum, plus its TF gradients):
 *   ll = sum_j log N(x_j | f_j*scale, var)            -> ll[1]
 *   dmu_j = (x_j - f_j*scale)/var                      -> dmu[n]   (d ll / d mu_j)
 *   dscale = sum_j dmu_j f_j, dvar = sum_j (-1/(2var) + (x_j-mu_j)^2/(2var^2))   -> dscale[1], dvar[1]
 * x, f: n contiguous elements; scale (nullable = 1) and var: one element each, device.
 * ws >= 3*ceil(n/1024) elements.  Two launches; sums in a fixed order. */
int hb_gauss_ll_f32(const float* x, const float* f, const float* scale, const float* var, long n, float* ll,
                    float* dmu, float* dscale, float* dvar, float* ws, long ws_elems, void* stream);
int hb_gauss_ll_f64(const double* x, const double* f, const double* scale, const double* var, long n,
                    double* ll, double* dmu, double* dscale, double* dvar, double* ws, long ws_elems,
                    void* stream);
/* The same head with the gradient handed back to the producer of f written by the same launch:
 * fbar_j = scale * (post * dmu_j) = d(post * ll) / d f_j (post: the constant upstream factor of the likelihood term, N / n
 * in every notebook ELBO).  Replaces the two elementwise ops TF autodiff appends to the head (and their launch). */
int hb_gauss_ll_post_f32(const float* x, const float* f, const float* scale, const float* var, long n, float* ll,
                         float* dmu, float* dscale, float* dvar, double post, float* fbar, float* ws, long ws_elems,
                         void* stream);
int hb_gauss_ll_post_f64(const double* x, const double* f, const double* scale, const double* var, long n, double* ll,
                         double* dmu, double* dscale, double* dvar, double post, double* fbar, double* ws, long ws_elems,
                         void* stream);

/* A whole cluster of elementwise ops in one launch: a register program interpreted per element
 * of the broadcast iteration space `shape[ndim]` (ndim <= 4).  Registers 0..nin-1 hold the inputs
 * (read with `istrides`); instruction q = code[q] = {op, dst, a, b, c} (HB_EW_* op on registers a,b,c;
 * HB_EW_GAUSS_LOGPDF_GRAD takes the register number of its 4th operand in params[q][0] and writes
 * dst..dst+2) with params[q][2]; output k = register out_regs[k] stored with `ostrides` (a 0 stride on a dim of extent > 1
 * marks a broadcast dim: only index 0 writes); out_regs[k] + HB_EW_PROG_SUM instead writes the SUM of that
 * register over the whole space to out[k][0] (tf.reduce_sum of a chain's result; the program then runs as one
 * workgroup, space <= 65536 elements).  All array arguments are (host) arrays.  Replaces the
 * chains of tiny TF elementwise ops of the reference's graph at one launch per chain. */
enum { HB_EW_PROG_SUM = 256 };
int hb_ewise_prog_f32(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                      const long* istrides, int nout, void* const* out, const int* out_regs,
                      const long* ostrides, int ndim, const long* shape, void* stream);
int hb_ewise_prog_f64(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                      const long* istrides, int nout, void* const* out, const int* out_regs,
                      const long* ostrides, int ndim, const long* shape, void* stream);
/* Device-resident form for programs that are replayed (a captured plan): hb_ewise_prog_build validates the
 * same arguments and writes the launch descriptor (hb_ewise_prog_image_bytes() bytes, independent of the element
 * type) to HOST memory; the caller copies it to the device once (4-byte aligned) and every launch is
 * hb_ewise_prog_run_* with that pointer and the `n` / `reduces` build returned.  The kernel pulls the descriptor
 * into LDS with one parallel load; the by-value form above walks it with dependent scalar loads (~4 us slower). */
long hb_ewise_prog_image_bytes(void);
int hb_ewise_prog_build(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                        const long* istrides, int nout, void* const* out, const int* out_regs,
                        const long* ostrides, int ndim, const long* shape, void* image_host, long* n_out,
                        int* reduces_out);
int hb_ewise_prog_run_f32(const void* image_dev, long n, int reduces, void* stream);
int hb_ewise_prog_run_f64(const void* image_dev, long n, int reduces, void* stream);
/* Compiled form (the one a plan uses when it can): the same program description is turned into HIP source --
 * shape, strides, op codes and constants as literals, every op the library's own definition with the op code known
 * at compile time -- and compiled for gfx950 at plan-build time by hiprtc (looked up with dlopen; where it is
 * absent hb_ewise_jit_available() returns 0 and callers keep the interpreted form above).  build returns an opaque
 * handle holding the loaded kernel and its pointer arguments (NULL for an empty iteration space); run launches it
 * on `stream` (capturable); identical programs share one compiled module.  source_out (nullable): receives the
 * generated kernel text, truncated to source_cap bytes, for inspection.  handle_out == NULL: dry run -- generate and
 * compile only (needs no device).  The interpreter took 8.8 us per launch at
 * cfg 2 for 7-11 scalar ops; replaces the same tf.* elementwise chains. */
int hb_ewise_jit_available(void);
int hb_ewise_jit_build_f32(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                           const long* istrides, int nout, void* const* out, const int* out_regs,
                           const long* ostrides, int ndim, const long* shape, void** handle_out, long* n_out,
                           int* reduces_out, char* source_out, long source_cap);
int hb_ewise_jit_build_f64(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                           const long* istrides, int nout, void* const* out, const int* out_regs,
                           const long* ostrides, int ndim, const long* shape, void** handle_out, long* n_out,
                           int* reduces_out, char* source_out, long source_cap);
int hb_ewise_jit_run(void* handle, void* stream);
int hb_ewise_jit_destroy(void* handle);
/* Column programs (round 3): a fused program over a SHORT-AND-WIDE space [R, n] (R <= 16 rows, e.g. the E experts of a
 * softmax gate over a minibatch of n points) in which reductions over the ROW axis are ordinary instructions.  One
 * thread owns one column: a register is R values (operands [R, n]) or one value (operands [1, n], scalars, and the
 * result of a row reduction), every row loop is unrolled in the thread, so the softmax gate of the expert mixture
 * (reference notebooks/Expert_GPR.ipynb:139-147: reduce_max, exp, reduce_sum, divide, weighted sum -- eight launches
 * op by op, three of them tf.reduce_* over axis 0) and its whole VJP are ONE launch each and every operand is read once.
 *   code[q] = {op, dst, a, b, c}: op is an HB_EW_* code (not HB_EW_GAUSS_LOGPDF_GRAD) or HB_COLPROG_SUM / HB_COLPROG_MAX
 *             (dst[0] = sum / max over the rows of register a, rows in order 0..R-1); registers 0..nin-1 are the inputs.
 *   istrides[k] = {row stride, column stride} of input k in elements: {ld, 1} an [R, n] operand (ld >= n: a row block of a
 *             taller matrix is passed by pointer offset, no copy), {0, 1} a row [1, n], {0, 0} a scalar.
 *   out_regs / ostrides likewise; an output with column stride 0 is written by the thread of column 0.
 * A register is R-high exactly when one of its operands is; handle_out / source_out / dry run as for
 * hb_ewise_jit_build; the handle is launched with hb_ewise_jit_run and released with hb_ewise_jit_destroy. */
enum { HB_COLPROG_SUM = -1, HB_COLPROG_MAX = -2 };
int hb_ewise_colprog_build_f32(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                               const long* istrides, int nout, void* const* out, const int* out_regs,
                               const long* ostrides, long R, long n, void** handle_out, char* source_out,
                               long source_cap);
int hb_ewise_colprog_build_f64(int ninstr, const int* code, const double* params, int nin, const void* const* in,
                               const long* istrides, int nout, void* const* out, const int* out_regs,
                               const long* ostrides, long R, long n, void** handle_out, char* source_out,
                               long source_cap);

enum { HB_RED_SUM = 0, HB_RED_MAX = 1 };
/* out[K1,K2] = reduce over R of contiguous in[K1,R,K2]  (tf.reduce_sum / reduce_max) */
int hb_reduce_f32(int op, const float* in, float* out, long K1, long R, long K2, float* ws,
                  long ws_elems, void* stream);
int hb_reduce_f64(int op, const double* in, double* out, long K1, long R, long K2, double* ws,
                  long ws_elems, void* stream);

/* strided n-d copy, strides in elements (tf.transpose / slice / tile / concat) */
int hb_copy_nd_f32(const float* in, const long* istr, float* out, const long* ostr, int ndim,
                   const long* shape, void* stream);
int hb_copy_nd_f64(const double* in, const long* istr, double* out, const long* ostr, int ndim,
                   const long* shape, void* stream);

int hb_fill_f32(float* out, long n, double v, void* stream);
int hb_fill_f64(double* out, long n, double v, void* stream);

/* K0: dst[i,:] = src[perm[idx[i]],:]  (perm nullable).  Device-resident form
 * of MinibatchData.get_feed_dict's `self.data[minibatch_index]`, reference
 * param.py:733-739, with Indexer's train/test permutation (model.py:147-153)
 * applied on device.  *err is set to 1 on an out-of-range index. */
int hb_gather_rows_f32(const float* src, long nsrc, long row, const long* idx, const long* perm,
                       long n, float* dst, int* err, void* stream);
int hb_gather_rows_f64(const double* src, long nsrc, long row, const long* idx, const long* perm,
                       long n, double* dst, int* err, void* stream);
/* The same gather for up to 8 arrays that share the index vector (all with nsrc rows; rows[a] = row width of
 * array a), in one launch: a model's MinibatchData arrays (reference param.py:733-739 feeds them one by one). */
int hb_gather_rows_multi_f32(int narr, const float* const* srcs, const long* rows, float* const* dsts,
                             long nsrc, const long* idx, const long* perm, long n, int* err, void* stream);
int hb_gather_rows_multi_f64(int narr, const double* const* srcs, const long* rows, double* const* dsts,
                             long nsrc, const long* idx, const long* perm, long n, int* err, void* stream);

/* The minibatch draw and that gather in ONE launch: idx[i] ~ U{lo..hi-1} is drawn from RNG lane i exactly as
 * hb_rng_randint(state, nlanes, idx_out, n, lo, hi) would (same values, same state afterwards; n <= nlanes), written
 * to idx_out, and row perm[idx[i]] of every array is gathered.  Replaces Indexer.train_index + the feed of every
 * MinibatchData (reference param.py:715-739) per step. */
int hb_gather_rows_multi_draw_f32(int narr, const float* const* srcs, const long* rows, float* const* dsts,
                                  long nsrc, uint64_t* state, long nlanes, long lo, long hi, long* idx_out,
                                  const long* perm, long n, int* err, void* stream);
int hb_gather_rows_multi_draw_f64(int narr, const double* const* srcs, const long* rows, double* const* dsts,
                                  long nsrc, uint64_t* state, long nlanes, long lo, long hi, long* idx_out,
                                  const long* perm, long n, int* err, void* stream);

/* batched [B,R,C] matrix utilities.  mode 0 = tf.matrix_band_part(lower,upper)
 * (reference variationals.py:145); 1 = + alpha*I (kernels.py:100 jitter);
 * 2 = Phi (lower triangle, halved diagonal) of the Cholesky gradient;
 * 3 = 0.5*(A + A^T);
 * 4 = symmetric matrix built from HALF the lower triangle: out[i][j] = out[j][i] = 0.5*A[max(i,j)][min(i,j)]
 *     (= (Phi(A) + Phi(A)^T)/2, the operand that makes the Cholesky gradient's L^-T Phi(.) L^-1 symmetric by
 *     construction, so no symmetrising pass over the result is needed). */
int hb_matutil_f32(const float* in, float* out, long B, long R, long C, int mode, long lower,
                   long upper, double alpha, void* stream);
int hb_matutil_f64(const double* in, double* out, long B, long R, long C, int mode, long lower,
                   long upper, double alpha, void* stream);

/* ---- RNG: xoroshiro128+ per lane (replaces tf.random_normal, reference
 *      variationals.py:107,127; gp/gp.py:132,138,142; and np.random.randint
 *      of model.py:147-153) ------------------------------------------------ */
/* state = 2*nlanes uint64 (s0[nlanes] then s1[nlanes]); seeded by splitmix64
 * from (seed, stream_id, lane). */
int hb_rng_init(uint64_t* state, long nlanes, uint64_t seed, uint64_t stream_id, void* stream);
int hb_rng_normal_f32(uint64_t* state, long nlanes, float* out, long n, void* stream);
int hb_rng_normal_f64(uint64_t* state, long nlanes, double* out, long n, void* stream);
/* uniform integers in [lo, hi), with replacement */
int hb_rng_randint(uint64_t* state, long nlanes, long* out, long n, long lo, long hi, void* stream);

/* ---- K1/K2: reparameterised Gaussian sampler fused with the Monte-Carlo KL
 *      (reference variationals.py:131-153 _sample, :178-186 logdet,
 *      :225-230 Normal._KL) ------------------------------------------------ */
/* diagonal q: x = mu + exp(s)*u ; kl = -0.5*sum(2 s + u^2 - x^2) over all n
 * elements.  u_in nullable (then drawn from rng, which must be non-null);
 * u_out, x: [n] dense; kl: 1 element; ws >= 2048 elements.
 * mu and s are n/L rows of L elements with row strides ld_mu, ld_s (>= L): L = n, ld = n is the flat case; an
 * encoder-fed (LOCAL) posterior passes the mean and log-std halves of the encoder's [rows, 2L] output in place
 * (reference variationals.py:70-80 slices them out of the fed tensor). */
int hb_diag_sample_kl_fwd_f32(const float* mu, const float* s, const float* u_in, uint64_t* rng,
                              long rng_lanes, float* u_out, float* x, float* kl, long n, long L,
                              long ld_mu, long ld_s, float* ws, void* stream);
int hb_diag_sample_kl_fwd_f64(const double* mu, const double* s, const double* u_in, uint64_t* rng,
                              long rng_lanes, double* u_out, double* x, double* kl, long n, long L,
                              long ld_mu, long ld_s, double* ws, void* stream);
/* VJP.  xbar nullable (= 0); klbar = d loss / d kl, one device scalar
 * (nullable = 0).  mubar = xbar + klbar*x ; sbar = mubar*exp(s)*u - klbar.
 * s has row stride ld_s; mubar and sbar are written with row stride ld_out (both halves of one [rows, 2L]
 * gradient of the encoder output, or dense with ld_out = L). */
int hb_diag_sample_kl_bwd_f32(const float* s, const float* u, const float* x, const float* xbar,
                              const float* klbar, float* mubar, float* sbar, long n, long L, long ld_s,
                              long ld_out, void* stream);
int hb_diag_sample_kl_bwd_f64(const double* s, const double* u, const double* x, const double* xbar,
                              const double* klbar, double* mubar, double* sbar, long n, long L, long ld_s,
                              long ld_out, void* stream);
/* full-rank q over `rows` independent blocks: x_r = mu_r + tril(S_r) u_r ;
 * kl = -0.5*sum(log S_kk^2 + u^2 - x^2).  mu/u/x: [rows,size].  S: packed == 0: dense
 * [rows,size,size] (upper part ignored); packed != 0: the lower triangle only,
 * [rows, size(size+1)/2] in numpy tril_indices (row-major) order -- half the bytes. */
int hb_fullrank_sample_kl_fwd_f32(const float* mu, const float* S, const float* u_in, uint64_t* rng,
                                  long rng_lanes, float* u_out, float* x, float* kl, long rows,
                                  long size, int packed, float* ws, void* stream);
int hb_fullrank_sample_kl_fwd_f64(const double* mu, const double* S, const double* u_in,
                                  uint64_t* rng, long rng_lanes, double* u_out, double* x,
                                  double* kl, long rows, long size, int packed, double* ws,
                                  void* stream);
/* The same in ONE launch for a block of up to 1024 dimensions (rows * size <= 8192; cfg 3's q(u)): every workgroup draws u
 * itself from the unchanged generator states, a wave owns rows k and size - 1 - k, the last workgroup to arrive folds the
 * KL partials and advances the generator.  `sync`: one zero 32-bit word owned by the caller for this stream (zero at
 * entry, left zero).  Shapes outside hb_fullrank_one_launch_shape, or sync == NULL: the three-launch form above.  Same
 * variates bit for bit; x and kl agree with the three-launch form to rounding. */
int hb_fullrank_one_launch_shape(long rows, long size);
int hb_fullrank_sample_kl_fwd1_f32(const float* mu, const float* S, const float* u_in, uint64_t* rng,
                                   long rng_lanes, float* u_out, float* x, float* kl, long rows, long size,
                                   int packed, float* ws, unsigned* sync, void* stream);
int hb_fullrank_sample_kl_fwd1_f64(const double* mu, const double* S, const double* u_in, uint64_t* rng,
                                   long rng_lanes, double* u_out, double* x, double* kl, long rows, long size,
                                   int packed, double* ws, unsigned* sync, void* stream);
/* mubar = xbar + klbar*x ; Sbar = tril(mubar u^T) - klbar*diag(1/S_kk) ; strictly upper = 0
 * (dense) or absent (packed: Sbar has S's packed layout) */
int hb_fullrank_sample_kl_bwd_f32(const float* S, const float* u, const float* x, const float* xbar,
                                  const float* klbar, float* mubar, float* Sbar, long rows,
                                  long size, int packed, void* stream);
int hb_fullrank_sample_kl_bwd_f64(const double* S, const double* u, const double* x,
                                  const double* xbar, const double* klbar, double* mubar,
                                  double* Sbar, long rows, long size, int packed, void* stream);
/* The reference's one (disabled) native-op pair, Henbun/tf_wraps.py:50-71: v [B, N(N+1)/2] <-> lower-
 * triangular tri [B,N,N] (zeros above the diagonal), entries in numpy tril_indices order
 * (transforms.py:225-244); each is the other's gradient (tf_wraps.py:56-58). */
int hb_vec_to_tri_f32(const float* v, float* tri, long B, long N, void* stream);
int hb_vec_to_tri_f64(const double* v, double* tri, long B, long N, void* stream);
int hb_tri_to_vec_f32(const float* tri, float* v, long B, long N, void* stream);
int hb_tri_to_vec_f64(const double* tri, double* v, long B, long N, void* stream);

/* ---- K3: stationary Gram matrices (reference gp/kernels.py:54-84
 *      square_dist, :110-111 UnitRBF.K, :122-131 UnitCsymRBF) -------------- */
enum { HB_KERN_RBF = 0, HB_KERN_CSYM_RBF = 1, HB_KERN_SQDIST = 2 /* the scaled squared distance itself */,
       HB_KERN_KBAR_SYMMETRIC = 256 /* hb_gram_bwd only, OR-ed into `kind`: Kbar is symmetric (the Cholesky VJP's output):
                                       the one-pass form need not read the transposed entries */ };
/* K[b,i,j] = k(X[b,i,:], X2[b,j,:]); sX/sX2 = batch strides in elements (0 =
 * shared); ell has dl = 1 (scalar) or d entries (ARD), post-transform; sEll = 0
 * (one kernel for the whole batch) or dl (a lengthscale vector per batch entry:
 * independent experts). */
/* diag_add is added to K[b,i,i] (the jitter of kern.Cholesky, gp/kernels.py:101; 0 otherwise). */
int hb_gram_fwd_f32(int kind, const float* X, long sX, const float* X2, long sX2, const float* ell,
                    long sEll, long dl, float* K, long B, long n, long n2, long d, double diag_add,
                    void* stream);
int hb_gram_fwd_f64(int kind, const double* X, long sX, const double* X2, long sX2,
                    const double* ell, long sEll, long dl, double* K, long B, long n, long n2, long d,
                    double diag_add, void* stream);
/* VJP: Xbar[B,n,d], X2bar[B,n2,d] (either nullable), ellbar[dl] or [B,dl] when sEll != 0
 * (nullable).  ws >= B*n*d elements when ellbar != NULL.  X2bar == Xbar (with X2 == X):
 * the total gradient w.r.t. the shared points is written once (kind | HB_KERN_KBAR_SYMMETRIC: for a symmetric Kbar,
 * without the strided reads of the transposed entries). */
int hb_gram_bwd_f32(int kind, const float* X, long sX, const float* X2, long sX2, const float* ell,
                    long sEll, long dl, const float* Kbar, float* Xbar, float* X2bar, float* ellbar,
                    long B, long n, long n2, long d, float* ws, void* stream);
int hb_gram_bwd_f64(int kind, const double* X, long sX, const double* X2, long sX2,
                    const double* ell, long sEll, long dl, const double* Kbar, double* Xbar,
                    double* X2bar, double* ellbar, long B, long n, long n2, long d, double* ws,
                    void* stream);

/* ---- dense linear algebra on MFMA (f32: v_mfma_f32_32x32x2_f32, f64:
 *      v_mfma_f64_16x16x4_f64) --------------------------------------------- */
enum {
  HB_MM_LOWER_OUT = 1, /* only tiles touching the lower triangle of C are computed; the rest of C is left alone */
  HB_MM_TRIL_OUT = 2,  /* C = tril(result): as LOWER_OUT, and the strict upper triangle is written as zero */
  HB_MM_PHI_OUT = 4,   /* C = Phi(result): strict lower kept, diagonal halved, strict upper zero (Cholesky VJP) */
  HB_MM_SYM_OUT = 8,   /* C = (R + R^T)/2 of the square result R (needs the workspace: ws_elems >= batch*M*N;
                          bias/act/beta are not applied) */
  HB_MM_ACTGRAD = 16   /* C = alpha * result * act'(Y): `bias` then points to Y = the activation OUTPUT of the layer
                          being differentiated, [batch, M, N] contiguous (sBias = M*N or 0), and `act` names the
                          activation: y(1-y), [y > 0], 1-y^2.  The MLP backward's "dx GEMM, then activation-gradient
                          pass" in one launch; beta must be 0 and no other flag may be set */,
  HB_MM_SYMLOW_OUT = 32, /* C[i][j] = C[j][i] = 0.5 * result[max(i,j)][min(i,j)] (hb_matutil mode 4 as an epilogue): only the
                          tiles touching the lower triangle are computed; square result, no bias / activation / beta */
  HB_MM_COLSUM_B = 64   /* (set by hb_matmul_colsum_*) the column sums of op(B) are produced next to C */
};
enum { HB_ACT_NONE = 0, HB_ACT_SIGMOID = 1, HB_ACT_RELU = 2, HB_ACT_TANH = 3 };
/* C[b] = act(alpha * op(A[b]) op(B[b]) + bias[b]) + beta * C[b], op = transpose if trans?.
 * A is M x K (after op), B is K x N, C is M x N; ld* = row strides, s* = batch
 * strides (0 = broadcast), bias nullable = per-column [N] (sBias batch stride).
 * Replaces tf.matmul (reference gp/gp.py:50,122,125,171; nn.py:32) and, with
 * bias/act, the fused MatBias layer clip(x@w+b) + activation (nn.py:31-32,79-84).
 * ws: split-K scratch (nullable -> no split). */
int hb_matmul_f32(const float* A, const float* B, float* C, long batch, long M, long N, long K,
                  long lda, long ldb, long ldc, long sA, long sB, long sC, int transA, int transB,
                  double alpha, double beta, const float* bias, long sBias, int act, int flags,
                  float* ws, long ws_elems, void* stream);
int hb_matmul_f64(const double* A, const double* B, double* C, long batch, long M, long N, long K,
                  long lda, long ldb, long ldc, long sA, long sB, long sC, int transA, int transB,
                  double alpha, double beta, const double* bias, long sBias, int act, int flags,
                  double* ws, long ws_elems, void* stream);

/* C = A^T B together with colsum[n] = sum_k B[k][n]: the weight gradient dW = X^T G of a MatBias layer AND its bias
 * gradient db = sum of the rows of G (reference nn.py:31-32 backward: tf.gradients of x @ w + b) from ONE pass over
 * G -- the workgroups of the first tile row fold the columns of B while it streams through them, the split-K finish
 * launch folds the slab partials; no stand-alone reduction launches.  A: [K, M] (lda), B: [K, N] (ldb), C: [M, N]; one
 * matrix; ws as in hb_matmul plus 64*N elements.  Shapes the fused form does not take (unaligned operands, K % 16 != 0)
 * fall back to hb_matmul followed by hb_reduce: same results up to summation order. */
int hb_matmul_colsum_f32(const float* A, const float* B, float* C, float* colsum, long M, long N, long K, long lda, long ldb,
                         long ldc, float* ws, long ws_elems, void* stream);
int hb_matmul_colsum_f64(const double* A, const double* B, double* C, double* colsum, long M, long N, long K, long lda,
                         long ldb, long ldc, double* ws, long ws_elems, void* stream);

/* ---- K7 fused: the amortised encoder in one launch per direction (csrc/mlp.hip; round 4).
 * o = act(y w0 + b0) w1 + b1   [n, 32] = [q_mu (16) | q_sqrt = log-std (16)]   (NeuralNet of two MatBias layers,
 * reference nn.py:31-32,73-84, fed to a LOCAL diagonal Normal, variationals.py:121-129), x = mu + exp(s) u and
 * kl = -0.5 sum(2 s + u^2 - x^2) (variationals.py:138-142,225-230); the hidden layer is never written.  The backward
 * recomputes it from y and returns the four weight gradients (no gradient for y: a data operand).
 * fp32; hb_mlp2_sample_supported(n, din, hid, nout, rng_lanes, has_u) == 1: din in {32, 64}, hid in {128, 256}, nout = 32,
 * n % 32 == 0 and, for in-kernel noise, rng_lanes >= 2 n (one generator lane per row and half: its draw differs from
 * hb_diag_sample_kl's lane assignment).  u_in nullable -> drawn from rng; u_out receives the noise used.
 * ws: hb_mlp2_sample_ws_elems(n, din, hid) elements (KL partials; the backward's per-chunk partial sums). */
int hb_mlp2_sample_supported(long n, long din, long hid, long nout, long rng_lanes, int has_u);
long hb_mlp2_sample_ws_elems(long n, long din, long hid);
int hb_mlp2_sample_fwd_f32(const float* y, const float* w0, const float* b0, const float* w1, const float* b1, int act,
                           const float* u_in, uint64_t* rng, long rng_lanes, float* x, float* kl, float* u_out, float* o,
                           long n, long din, long hid, float* ws, void* stream);
/* xbar [n,16] (nullable) and klbar [1] (nullable): gradients w.r.t. x and kl; o, u, x: the forward's outputs. */
int hb_mlp2_sample_bwd_f32(const float* y, const float* w0, const float* b0, const float* w1, int act, const float* o,
                           const float* u, const float* x, const float* xbar, const float* klbar, float* dw0, float* db0,
                           float* dw1, float* db1, long n, long din, long hid, float* ws, void* stream);

/* K4: L = chol(A), lower, batched [B,M,M]; the strict upper triangle of L is
 * zeroed; info[B] (device) receives 0 or k+1.  Replaces tf.cholesky
 * (reference gp/kernels.py:101; gp/gp.py:135).  A and L must NOT alias (workgroups
 * re-read diagonal tiles of A while L is being written); only the lower triangle
 * of A is significant beyond its diagonal tiles. */
int hb_cholesky_f32(const float* A, float* L, long B, long M, int* info, void* stream);
int hb_cholesky_f64(const double* A, double* L, long B, long M, int* info, void* stream);

/* K4+K6 fused: L = chol(A) and W = L^-1 from the same launches (the identity is
 * eliminated alongside A: the inverse costs extra width per launch, no extra
 * depth).  fp32 with M % 64 == 0 runs as ONE persistent launch (csrc/chol_persist.cuh): workgroups keep their tiles
 * in registers for the whole factorisation and hand finished 16-column chunks of a panel to each other through ws.
 * ws: hb_cholesky_inverse_ws_elems(B, M, sizeof element) elements (B*M*M, plus the persistent launch's sync words).
 * WORKSPACE CONTRACT (fp32, M % 64 == 0): the sync words behind the first B*M*M elements must be ZERO when a call
 * starts.  The caller zero-fills a workspace once, after allocating it (or whenever something else wrote into it);
 * every call leaves them zero again, so the same workspace serves call after call (graph replays included) as long
 * as it is used with the same (B, M) and by one stream at a time.  info[b] = -1 reports a synchronisation timeout
 * (dirty sync words, or a device that stopped making progress for 2 s): the outputs are then invalid.
 * Replaces the tf.cholesky +
 * tf.matrix_triangular_solve(Lm, .) pair of SparseGP.samples (reference
 * gp/gp.py:135,162,169).  A, L, W must not alias.
 * Wfrag (nullable; 2*B*M*M elements, needs M % 32 == 0): fragment-major copies of W and of W^T --
 * [B][M/32 row tiles t][M/32 k chunks Q][4 v][64 lanes (li + 32 h)][4 s] = W[32t+li][32Q+16h+4v+s] (then the
 * same for W^T) -- the order in which the MFMA operand loads of hb_sgp_fwd / hb_sgp_bwd consume them, so that
 * every load instruction reads one contiguous kilobyte.
 * frag_bf16x3 != 0 (fp32 only): Wfrag has 5*B*M*M elements and, behind the two fp32 images, receives the bf16x3
 * operand images of W and W^T (3 + 3 planes of B*M*M bf16: each fp32 entry split into hi + mid + lo bf16 terms;
 * [term][B][t][Q][k16-step q][64 lanes][8] = X[32t+li][32Q+16q+8h+j]) for the HB_PREC_BF16X3 contractions. */
long hb_cholesky_inverse_ws_elems(long B, long M, int elem_bytes);
/* 1 when hb_cholesky_inverse_f32 takes the persistent launch for this (B, M) (and the diagnostic switch leaves it on). */
int hb_cholesky_persistent_shape(long B, long M, int elem_bytes);
/* hb_gram_fwd with X2 = X and a diag_add, followed by hb_cholesky_inverse, as one launch: the persistent kernel synthesises its tiles of
 * K(X, X) + diag_add I from the points (the same per-entry function as hb_gram_fwd: the same bits), K itself is never
 * written.  kern.Cholesky(z) feeding SparseGP's whitening, reference gp/kernels.py:93-101 + gp/gp.py:159-162.
 * Only where hb_cholesky_persistent_shape(B, M, 4) is 1; X [B or 1][M, d] (sX = 0: shared), ell as in hb_gram_fwd. */
int hb_gram_cholesky_inverse_f32(int kind, const float* X, long sX, const float* ell, long sEll, long dl, long d,
                                 double diag_add, float* L, float* W, long B, long M, int* info, float* ws,
                                 float* Wfrag, int frag_bf16x3, void* stream);
int hb_cholesky_inverse_f32(const float* A, float* L, float* W, long B, long M, int* info, float* ws,
                            float* Wfrag, int frag_bf16x3, void* stream);
int hb_cholesky_inverse_f64(const double* A, double* L, double* W, long B, long M, int* info,
                            double* ws, double* Wfrag, int frag_bf16x3, void* stream);
/* W = L^{-1} (lower triangular inverse), batched.  Used in place of
 * tf.matrix_triangular_solve(Lm, .) (reference gp/gp.py:162,169): the
 * reference's own batched branch forms the explicit inverse the same way.
 * ws >= B*M*M elements (nullable when M <= 32); W must not alias L. */
int hb_trinv_f32(const float* L, float* W, long B, long M, float* ws, void* stream);
int hb_trinv_f64(const double* L, double* W, long B, long M, double* ws, void* stream);

/* ---- K5/K6: fused sparse-GP conditional (reference gp/gp.py:99-143 samples,
 *      :146-162 _effective_LT, :177-189 _additional_cov 'diagonal') -------- */
enum { HB_SGP_NEGLECTED = 0, HB_SGP_DIAGONAL = 1 };
/* Operand precision of the M^2 n contraction.  NATIVE: the arithmetic type (fp32 / fp64 MFMA).  BF16X3 (fp32
 * only; BASELINE cfg 5's "fp16-with-fp32-accum" variant in a usable form): every fp32 operand is split into three
 * bf16 terms and the six significant cross products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation --
 * fp32-level accuracy (plain 16-bit operands leave a 27 % error in L^-1 K) at the bf16 matrix rate.  Needs the
 * bf16 images of hb_cholesky_inverse_f32 (frag_bf16x3 = 1) in Wfrag. */
enum { HB_PREC_NATIVE = 0, HB_PREC_BF16X3 = 1 };
/* Per expert e < E (all arrays carry a leading E; x may be shared: sx = 0):
 *   Kmn = k(z, x)            [M,n]   (never written to memory)
 *   A   = W Kmn              [M,n]   W = chol(Kmm + jitter I)^{-1}
 *   mean= u A                [P,n]
 *   v   = kdiag(x) - sum_m A^2        [n]
 *   f   = mean + sqrt(|v|) * eps      (DIAGONAL; eps [n] shared by the P rows,
 *                                      reference gp/gp.py:131-132)  or mean (NEGLECTED)
 * eps_in nullable -> drawn from rng; eps_out [E,n] receives the noise used.
 * Wfrag (nullable): the fragment-major copies hb_cholesky_inverse wrote for this W (faster operand loads).
 * ws >= hb_sgp_ws_elems(...) elements. */
long hb_sgp_ws_elems(long E, long n, long M, long d, long P);
/* Column-strip form and fragment-major exchange.  With Wfrag and hb_sgp_strip_path(...) == 1 the contraction runs
 * as column strips (one workgroup = 32 data columns x all M rows) and A may be left -- additionally (A_frag), or
 * only (A == NULL) -- in FRAGMENT-MAJOR layout for hb_sgp_bwd:
 *   A_frag [E][M/32 row tiles t][nS = ceil(n/32) strips s][4 v][64 lanes (li + 32 h)][4 s'] = A[32t+li][32s+16h+4v+s']
 * (zeros past column n): the MFMA operand fragments of the Lbar contraction over the data axis, in load order. */
int hb_sgp_strip_path(long E, long n, long M, long d, long P, int prec);
int hb_sgp_fwd_f32(int kind, int mode, const float* x, long sx, const float* z, const float* ell,
                   long dl, const float* W, const float* Wfrag, int prec, const float* u,
                   const float* eps_in, uint64_t* rng, long rng_lanes, float* eps_out, float* A,
                   float* A_frag, float* f, float* v, long E, long n, long M, long d, long P, float* ws,
                   void* stream);
int hb_sgp_fwd_f64(int kind, int mode, const double* x, long sx, const double* z, const double* ell,
                   long dl, const double* W, const double* Wfrag, int prec, const double* u,
                   const double* eps_in, uint64_t* rng, long rng_lanes, double* eps_out, double* A,
                   double* A_frag, double* f, double* v, long E, long n, long M, long d, long P,
                   double* ws, void* stream);
/* The Gaussian likelihood head riding in the forward contraction (fp32, column-strip form, one latent function): where
 * hb_sgp_head_units(...) > 0 the strip kernel that finishes f_j also computes the per-point part of
 * hb_gauss_ll_post -- dmu_j, fbar_j = scale * (post * dmu_j) -- and leaves its strip's partial sums of
 * (ll, dscale, dvar) in head_part[3][units]; hb_gauss_ll_fold_* folds them (chain-aware: a job of the step's last serial
 * chain).  Replaces a launch of its own between the forward and the backward contraction (hb_gauss_ll: 5.6 us of the
 * cfg-2 step).  y [E, n] like f; draw = 1 when eps is drawn from rng (eps_in == NULL, DIAGONAL). */
long hb_sgp_head_units(long E, long n, long M, long d, long P, int prec, int has_wfrag, int draw, long rng_lanes);
int hb_sgp_fwd_gauss_f32(int kind, int mode, const float* x, long sx, const float* z, const float* ell, long dl,
                         const float* W, const float* Wfrag, int prec, const float* u, const float* eps_in,
                         uint64_t* rng, long rng_lanes, float* eps_out, float* A, float* A_frag, float* f, float* v,
                         long E, long n, long M, long d, long P, float* ws, const float* y, const float* scale,
                         const float* var, double post, float* dmu, float* fbar, float* head_part, long units,
                         void* stream);
/* Early-start forward (round 4): the forward contraction INSIDE the launch of the factorisation that feeds it.  Row block j
 * of W = L^-1 is final as soon as panel j of the persistent factorisation is done, so the strips of A = W K(z, x) can
 * take their row tiles while the factorisation is still running on a quarter of the chip (csrc/sgp.hip, early-start form).
 * Protocol, per thread:  hb_sgp_rider_begin();  ONE hb_sgp_fwd_f32 / hb_sgp_fwd_gauss_f32 call -- recorded, not launched
 * (its W / Wfrag are the buffers the factorisation is about to write; hb_sgp_rider_supported(...) must be 1 for its
 * arguments, otherwise the call fails);  then hb_cholesky_inverse_f32 / hb_gram_cholesky_inverse_f32 with the same
 * Wfrag: ONE launch runs the factorisation, this thread's pending side jobs and the recorded forward.  If the
 * factorisation cannot take it (other buffers, batch, size; bf16x3 images; more workgroups than two per CU) it launches
 * as usual and the recorded forward right behind it, so the results never depend on whether the ride happened (bits of A:
 * the last two row tiles are summed in three partial sums when it did).  hb_sgp_rider_flush launches a recorded
 * forward that no factorisation picked up.  Reference: gp/gp.py:159-172 (L^-1 K(z, x) behind tf.cholesky). */
int hb_sgp_rider_supported(long E, long n, long M, long d, long P, int prec, int has_wfrag, int draw, long rng_lanes);
int hb_sgp_rider_begin(void);
int hb_sgp_rider_pending(void);
int hb_sgp_rider_flush(void* stream);
/* The Gram VJP in the epilogue of the product that computes Kbar (round 4).  The Cholesky VJP ends in S = L^-T Phi L^-1
 * (reference: TF's _CholeskyGrad behind tf.cholesky gp/kernels.py:101), whose only reader is the VJP of K(X, X) + jitter I
 * (gp/kernels.py:54-101): C[batch, M, M] = op(A) op(B) as hb_matmul computes it (in-workgroup split-K form: fp32, M % 32 == 0,
 * K % 128 == 0, <= 1024 tiles -- hb_matmul_gram_vjp_ok) AND, from the same launch, Xbar[batch, M, d] and the lengthscale
 * row partials ell_partial[batch * M, d] of hb_gram_bwd's one-pass symmetric form (UnitRBF, Kbar symmetric).  `part`:
 * hb_matmul_gram_vjp_ws_elems(batch, M, d) scratch elements; `counters`: batch * M / 32 zero 32-bit words, left zero.
 * hb_gram_ell_fold_* folds the partials into ellbar (the last launch of hb_gram_bwd on its own; chain-aware). */
int hb_matmul_gram_vjp_ok(long M, long K, long batch, long d);
long hb_matmul_gram_vjp_ws_elems(long batch, long M, long d);
int hb_matmul_gram_vjp_f32(const float* A, const float* B, float* C, long batch, long M, long K, long lda, long ldb,
                           long ldc, long sA, long sB, long sC, int transA, int transB, const float* X, long sX,
                           const float* ell, long sEll, long dl, long d, float* Xbar, float* ell_partial,
                           float* part, unsigned* counters, void* stream);
int hb_gram_ell_fold_f32(const float* partial, long rows, long d, long dl, long groups, float* ellbar, void* stream);
int hb_gram_ell_fold_f64(const double* partial, long rows, long d, long dl, long groups, double* ellbar, void* stream);
/* The Gaussian likelihood head of a MatBias layer in the layer's own launch (round 4; reference nn.py:31-32 feeding
 * densities.py:25-27 under tf.reduce_sum): f = A[n, K] B[K, N] + bias[N] is consumed in the epilogue of the row-streaming
 * product and never written -- dmu[n, N] = (y - f s) / var, fbar = s (post dmu) when fbar != NULL, and one partial triple
 * of (ll, dscale, dvar) per wave in part[3][units] for hb_gauss_ll_fold.  units = hb_matmul_gauss_units(n, K, N); 0: this
 * shape takes the product and the head as two launches.  Supported: fp32, n >= 2048, 32 < N <= 256, K == 16 or K in {32, 64, 128}.  dmu / fbar agree
 * with hb_gauss_ll on the materialised f to fp32 rounding (the same per-point arithmetic on an f that was never rounded
 * through memory); the three sums are taken in another (fixed) order. */
long hb_matmul_gauss_units(long n, long K, long N);
int hb_matmul_gauss_f32(const float* A, long lda, const float* B, long ldb, const float* bias, const float* y,
                        const float* scale, const float* var, double post, float* dmu, float* fbar, float* part,
                        long units, long n, long K, long N, void* stream);
int hb_gauss_ll_fold_f32(const float* partial, long nb, float* ll, float* dscale, float* dvar, void* stream);
int hb_gauss_ll_fold_f64(const double* partial, long nb, double* ll, double* dscale, double* dvar, void* stream);
/* The contraction alone, A = W k(z,x) (posterior-prediction callers; isolated timing). */
int hb_sgp_A_f32(int kind, const float* x, long sx, const float* z, const float* ell, long dl,
                 const float* W, const float* Wfrag, int prec, float* A, long E, long n, long M, long d,
                 void* stream);
int hb_sgp_A_f64(int kind, const double* x, long sx, const double* z, const double* ell, long dl,
                 const double* W, const double* Wfrag, int prec, double* A, long E, long n, long M,
                 long d, void* stream);
/* VJP given fbar [E,P,n]:
 *   Abar = u^T fbar + A diag(c),  c = -eps sign(v)/sqrt|v| * sum_p fbar_p
 *   Kbar = W^T Abar            [E,M,n]  (scratch output, kept for Lbar)
 *   Lbar = -tril(Kbar A^T)     [E,M,M]
 *   ubar = fbar A^T            [E,P,M]
 * Wfrag (nullable) / prec: as for hb_sgp_fwd; with Wfrag (fp32, M %% 32 == 0, M <= 512, d <= 4, P <= 4, no xbar) Kbar
 * and the row gradients come from ONE column-strip kernel (no second pass over Kbar and A).
 * A_frag / Kbar_frag (both or neither; need hb_sgp_strip_path): A is read, and Kbar written, in the fragment-major
 * layout of hb_sgp_fwd's A_frag (A / Kbar may then be NULL), and Lbar comes from a contraction whose every operand
 * load is one contiguous kilobyte (E*M*32*ceil(n/32) elements each).
 *   zbar, ellbar (and xbar, nullable) through Kmn = k(z,x). */
int hb_sgp_bwd_f32(int kind, int mode, const float* x, long sx, const float* z, const float* ell,
                   long dl, const float* W, const float* Wfrag, int prec, const float* u, const float* eps,
                   const float* A, const float* A_frag, const float* v, const float* fbar, float* Kbar,
                   float* Kbar_frag, float* Lbar, float* ubar, float* zbar, float* ellbar, float* xbar,
                   long E, long n, long M, long d, long P, float* ws, void* stream);
int hb_sgp_bwd_f64(int kind, int mode, const double* x, long sx, const double* z, const double* ell,
                   long dl, const double* W, const double* Wfrag, int prec, const double* u,
                   const double* eps, const double* A, const double* A_frag, const double* v,
                   const double* fbar, double* Kbar, double* Kbar_frag, double* Lbar, double* ubar,
                   double* zbar, double* ellbar, double* xbar, long E, long n, long M, long d, long P,
                   double* ws, void* stream);

/* ---- K9: flat-buffer Adam, TensorFlow-1 formula (reference model.py:206,220
 *      tf.train.AdamOptimizer via optimizer.minimize) --------
 *   t <- t+1 ; lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
 *   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; theta <- theta - lr_t m/(sqrt(v)+eps)
 * g is read as gscale*g (gscale = 1/world_size for data-parallel means).
 * t is a device-side step counter (one int64), incremented by the call when `tick` != 0 (several
 * calls on disjoint segments of one parameter set share a step: only the last one ticks).
 * Failure containment: tf.cholesky raises inside session.run BEFORE apply_gradients (reference
 * model.py:265-266), leaving the parameters at the last good step.  Calls here are asynchronous,
 * so the update reads this step's factorisation status words `info[n_info]` (nullable, LAPACK
 * convention), the other ranks' all-reduced failure flag `dpflag` (nullable, one element) and the
 * sticky record `fail[2]` (nullable): when any is non-zero the call changes nothing (theta, m, v,
 * t) and `fail` keeps {step number t+1, status} of the FIRST blocked step until the host clears it. */
int hb_adam_step_f32(float* theta, const float* g, float* m, float* v, long n, double lr, double b1,
                     double b2, double eps, double gscale, long* t, int tick, const int* info,
                     long n_info, const float* dpflag, long* fail, void* stream);
int hb_adam_step_f64(double* theta, const double* g, double* m, double* v, long n, double lr,
                     double b1, double b2, double eps, double gscale, long* t, int tick,
                     const int* info, long n_info, const double* dpflag, long* fail, void* stream);

/* ---- side jobs: small independent launches riding on another kernel's launch (csrc/side_jobs.cuh) --------------
 * hb_side_push_* take the arguments of their stand-alone twins (hb_gather_rows_multi_draw_f32,
 * hb_diag_sample_kl_fwd_f32, hb_diag_sample_kl_bwd_f32) but RECORD the job in a per-thread list instead of launching
 * it (a job that does not fit the side form -- more than one workgroup for the sampler, fp64 -- is launched at
 * once, as the twin would).  The next host launch issued by the same thread -- launch 0 of hb_cholesky /
 * hb_cholesky_inverse on the 64-column fp32 path, hb_matmul on its in-workgroup split-K path -- appends the recorded
 * jobs' workgroups to its own grid.  hb_side_flush launches whatever is still pending as one kernel of its own, so
 * push ... host ... flush is always equivalent to the stand-alone calls provided nobody reads the jobs' outputs
 * before the host launch has run.  hb_side_pending: number of recorded jobs (at most 3; a fourth push flushes). */
int hb_side_push_gather_draw_f32(int narr, const float* const* srcs, const long* rows, float* const* dsts, long nsrc,
                                 uint64_t* state, long nlanes, long lo, long hi, long* idx_out, const long* perm,
                                 long n, int* err, void* stream);
int hb_side_push_diag_fwd_f32(const float* mu, const float* s, const float* u_in, uint64_t* rng, long rng_lanes,
                              float* u_out, float* x, float* kl, long n, long L, long ld_mu, long ld_s, float* ws,
                              void* stream);
int hb_side_push_diag_bwd_f32(const float* s, const float* u, const float* x, const float* xbar, const float* klbar,
                              float* mubar, float* sbar, long n, long L, long ld_s, long ld_out, void* stream);
int hb_side_pending(void);
int hb_side_flush(void* stream);
/* Drops the calling thread's recorded jobs without running them (returns how many): for a caller whose launch sequence
 * was abandoned between a push and its host launch; the recorded raw pointers must not ride on a later, unrelated
 * launch.  henbun_amd's Plan calls it before and after every execution / capture of a plan. */
int hb_side_discard(void);

/* ---- serial chains: several small DEPENDENT launches run as one generated kernel (csrc/chain.cuh, csrc/jit.hip) ----
 * Between hb_chain_begin() and hb_chain_end(stream) the calling thread records the small launches that the chain-aware
 * entry points would issue (each up to a few thousand elements: a chain is one workgroup, csrc/chain.cuh has the limits) --
 * hb_ewise_jit_run, hb_gauss_ll_*, hb_adam_step_*, the finishing pass of hb_sgp_fwd_*, the lengthscale fold of hb_gram_bwd_* --
 * and runs them, in
 * call order, as ONE hiprtc-compiled kernel: one workgroup of 1024 threads, a barrier between jobs.  A launch that cannot
 * be recorded first runs what is recorded, so the order of the calls is the order of execution; ONLY the entry points
 * named above may be called between begin and end.  A kernel boundary inside a captured step costs ~4.5 us, more than
 * any of these bodies takes: at BASELINE cfg 2 the lengthscale fold, the last gradient cluster and Adam are one launch.  Without hiprtc in the process
 * (hb_ewise_jit_available() == 0) nothing is recorded and every call launches as usual.
 * hb_chain_discard: drop what is recorded without running it and stop recording (returns the number of jobs dropped).
 * hb_chain_source: the generated part of the source the recorded jobs would compile to (diagnostics; launches nothing). */
int hb_chain_begin(void);
int hb_chain_end(void* stream);
int hb_chain_discard(void);
int hb_chain_source(char* out, long cap);
/* hb_chain_compile_dry: compile the kernel of the recorded jobs for gfx950 without loading or launching it (no device
 * needed), drop the jobs and stop recording: the build / CPU check of the generated chains. */
int hb_chain_compile_dry(void);

/* ---- data-parallel exchange step (no reference counterpart: the reference is one tf.Session on one
 *      device, model.py:57,255-269) --------------------------------------------------
 * One process per GPU; the ranks exchange ONE all-reduce (sum) of the flat gradient buffer per Adam
 * step, over RCCL (xGMI inside a node), issued on the caller's stream -- capturable into the step's
 * hipGraph between the backward kernels and hb_adam_step.
 *   hb_comm_available   1 when RCCL could be resolved in this process (it is looked up among the already
 *                       loaded libraries first, then dlopen'ed: the library has no link-time dependency).
 *   hb_comm_unique_id   (host) 128-byte rendezvous token, created on rank 0 and handed to the other ranks
 *                       by the launcher's own channel (henbun_amd.parallel uses torch.distributed for that).
 *   hb_comm_init        collective over all ranks; *comm_out is a (host) handle.
 *   hb_allreduce_sum    in place over `n` elements, asynchronous on `stream`.
 *   hb_dp_pack          tail[0] = objective[0] (nullable -> 0), tail[1] = any(info != 0): the two words that
 *                       ride behind the gradient so that every rank sees the mean objective and a failed
 *                       factorisation on ANY rank blocks the update on EVERY rank (hb_adam_step's dpflag). */
#define HB_COMM_ID_BYTES 128
int hb_comm_available(void);
int hb_comm_unique_id(char* id128);
int hb_comm_init(const char* id128, int rank, int world, void** comm_out);
int hb_comm_destroy(void* comm);
int hb_allreduce_sum_f32(float* buf, long n, void* comm, void* stream);
int hb_allreduce_sum_f64(double* buf, long n, void* comm, void* stream);
int hb_dp_pack_f32(float* tail, const float* objective, const int* info, long n_info, void* stream);
int hb_dp_pack_f64(double* tail, const double* objective, const int* info, long n_info, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HENBUN_HIP_H */
