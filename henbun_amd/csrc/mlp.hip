// K7 fused: the amortised encoder as ONE launch per direction (round 4).
//
// Reference: nn.py:31-32,73-84 (MatBias / NeuralNet: clip(x @ w + b), an activation between the layers, none after
// the last) feeding a LOCAL diagonal Normal through Variational.feed (variationals.py:121-129: the encoder output's
// columns are [q_mu (L) | q_sqrt = log-std (L)], sorted-name order of param.py:516-537), sample x = mu + exp(s) u and
// Monte-Carlo KL -0.5 sum(2 s + u^2 - x^2) (variationals.py:138-142,225-230).
//
// Op by op that is four launches forward (two GEMMs, sampler, KL fold) and five backward (sampler VJP, the input-
// gradient GEMM with the activation derivative, two weight-gradient GEMMs with their finish launches), and the hidden
// layer h [n, H] -- 33.5 MB at cfg 4 -- is written once and read three times.  Here h never exists in memory:
//
//   forward  (mlp2_fwd_kernel): a wave owns 32 minibatch rows.  Layer 0 is computed TRANSPOSED on
//     v_mfma_f32_32x32x2_f32 -- D[hidden][row] = W0^T[hidden][k] y^T[k][row] -- so the accumulator has the row on the
//     lane and the hidden unit in the registers; after the activation (in registers) an accumulator register IS the B
//     operand of the second layer's MFMA (contraction over the register index: no lane movement, no LDS):
//     D2[out][row] += W1^T[out][hidden] h^T[hidden][row].  Its accumulator holds, per lane, mu_l and s_l of the SAME
//     latent dimensions (registers r and r + 8), so the reparameterised sample and the KL terms are lane-local.
//     Weights sit in LDS (96 KB at [64, 256, 32]); written: o = [mu | s], x, u and one KL partial per workgroup.
//   backward (mlp2_bwd_kernel): h is RECOMPUTED from y (1.07 GFLOP at cfg 4: ~7 us at the fp32 MFMA peak, against
//     reading 33.5 MB twice).  A workgroup owns 64 hidden units and a chunk of rows; each of its four waves walks its own
//     32-row tiles.  Layer 0 in NATURAL orientation -- D[row][hidden] -- puts the hidden unit on the lane and the rows in
//     the registers, so that the accumulator is, untouched, the A operand of dW1 += h^T do and, after the in-register
//     product with the activation derivative, the B operand of dW0 += y^T dh (both contract over rows = the register
//     index).  do = [mubar | sbar] comes from (xbar, x, u, s) per row (the sampler's VJP) into a wave-private LDS tile.
//     Per-chunk partial sums of (dW0, db0, dW1, db1) are folded by mlp2_bwd_finish_kernel in fixed order.
// fp32, L = 16 (32 encoder outputs), Din in {32, 64}, H in {128, 256}, n % 32 == 0; everything else is lowered to
// the op-by-op launches by the planner (henbun_amd/graph.py: mlp2_sample_kl).
#include "common.cuh"
#include "../../include/henbun_hip.h"

typedef float MlV4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ T mlp_act(int act, T v) {
  switch (act) {
    case HB_ACT_SIGMOID: return hb_sigmoid(v);
    case HB_ACT_RELU: return v > T(0) ? v : T(0);
    case HB_ACT_TANH: return hb_tanh(v);
    default: return v;
  }
}
template <typename T>
__device__ __forceinline__ T mlp_act_grad(int act, T y) {   // through the activation's OUTPUT
  switch (act) {
    case HB_ACT_SIGMOID: return y * (T(1) - y);
    case HB_ACT_RELU: return y > T(0) ? T(1) : T(0);
    case HB_ACT_TANH: return T(1) - y * y;
    default: return T(1);
  }
}

struct Mlp2FwdArgs {
  const float *y, *w0, *b0, *w1, *b1, *u_in;
  uint64_t* rng;
  long rng_lanes;
  float *x, *u_out, *o, *klpart;
  long n;
  int act;
};

#define MLP_L 16
#define MLP_O 32   // encoder outputs = 2 L

template <int DIN, int HID>
__global__ void __launch_bounds__(256) mlp2_fwd_kernel(Mlp2FwdArgs a) {
  typedef Mma<float> MM;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* W0s = sm;                    // [DIN][HID]
  float* W1s = W0s + DIN * HID;       // [HID][32]
  float* b0s = W1s + HID * MLP_O;     // [HID]
  float* b1s = b0s + HID;             // [32]
  float* red = b1s + MLP_O;           // [4]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < DIN * HID / 4; i += 256) reinterpret_cast<MlV4*>(W0s)[i] = reinterpret_cast<const MlV4*>(a.w0)[i];
  for (int i = tid; i < HID * MLP_O / 4; i += 256) reinterpret_cast<MlV4*>(W1s)[i] = reinterpret_cast<const MlV4*>(a.w1)[i];
  for (int i = tid; i < HID; i += 256) b0s[i] = a.b0[i];
  if (tid < MLP_O) b1s[tid] = a.b1[tid];
  __syncthreads();
  const int li = lane & 31, half = lane >> 5;
  constexpr int HT = HID / 32, KH = DIN / 2;   // hidden tiles; contraction entries per lane half
  float klacc = 0.f;
  const long ntiles = a.n / 32;
  for (long tile = (long)blockIdx.x * 4 + w; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + li;
    // B operand of layer 0: this lane's row, entries [half KH, half KH + KH) of the contraction (permuted k: 16-byte loads)
    float yr[KH];
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const MlV4 t = *reinterpret_cast<const MlV4*>(a.y + row * DIN + half * KH + 4 * v);
      yr[4 * v] = t[0], yr[4 * v + 1] = t[1], yr[4 * v + 2] = t[2], yr[4 * v + 3] = t[3];
    }
    typename MM::Acc acc[HT];
#pragma unroll
    for (int T = 0; T < HT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[T][r] = b0s[32 * T + MM::acc_row(lane, r)];
    // (operands of step s + 1 are read while step s runs; the scheduling barriers keep the compiler from hoisting ALL
    // LDS reads of the unrolled loop to its top: 256 VGPRs + 256 AGPRs and 138 spills without them)
    {
      float wc[HT], wn[HT];
#pragma unroll
      for (int T = 0; T < HT; ++T) wc[T] = W0s[(half * KH) * HID + 32 * T + li];
#pragma unroll
      for (int s = 0; s < KH; ++s) {
        if (s + 1 < KH) {
#pragma unroll
          for (int T = 0; T < HT; ++T) wn[T] = W0s[(half * KH + s + 1) * HID + 32 * T + li];
        }
#pragma unroll
        for (int T = 0; T < HT; ++T) acc[T] = MM::mma(wc[T], yr[s], acc[T]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 0; T < HT; ++T) wc[T] = wn[T];
      }
    }
    typename MM::Acc oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = b1s[MM::acc_row(lane, r)];
    {
      float vc[16], vn[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vc[r] = W1s[MM::acc_row(lane, r) * MLP_O + li];
#pragma unroll
      for (int T = 0; T < HT; ++T) {
        if (T + 1 < HT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vn[r] = W1s[(32 * (T + 1) + MM::acc_row(lane, r)) * MLP_O + li];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float h = mlp_act<float>(a.act, acc[T][r]);
          oacc = MM::mma(vc[r], h, oacc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) vc[r] = vn[r];
      }
    }
    // oacc[r], r < 8: mu of latent dimension acc_row(lane, r); oacc[r + 8]: its log-std
    float z[8];
    if (a.u_in) {
      const MlV4 t0 = *reinterpret_cast<const MlV4*>(a.u_in + row * MLP_L + 4 * half);
      const MlV4 t1 = *reinterpret_cast<const MlV4*>(a.u_in + row * MLP_L + 8 + 4 * half);
      z[0] = t0[0], z[1] = t0[1], z[2] = t0[2], z[3] = t0[3], z[4] = t1[0], z[5] = t1[1], z[6] = t1[2], z[7] = t1[3];
    } else {
      // one generator lane per (row, half): four steps, eight normals
      HbRng g = rng_load(a.rng, a.rng_lanes, row * 2 + half);
#pragma unroll
      for (int p = 0; p < 4; ++p) g.normal2(z[2 * p], z[2 * p + 1]);
      rng_store(a.rng, a.rng_lanes, row * 2 + half, g);
    }
    float xv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float sv = oacc[r + 8];
      xv[r] = oacc[r] + hb_exp(sv) * z[r];
      klacc += 2.f * sv + z[r] * z[r] - xv[r] * xv[r];
    }
    *reinterpret_cast<MlV4*>(a.x + row * MLP_L + 4 * half) = MlV4{xv[0], xv[1], xv[2], xv[3]};
    *reinterpret_cast<MlV4*>(a.x + row * MLP_L + 8 + 4 * half) = MlV4{xv[4], xv[5], xv[6], xv[7]};
    *reinterpret_cast<MlV4*>(a.u_out + row * MLP_L + 4 * half) = MlV4{z[0], z[1], z[2], z[3]};
    *reinterpret_cast<MlV4*>(a.u_out + row * MLP_L + 8 + 4 * half) = MlV4{z[4], z[5], z[6], z[7]};
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<MlV4*>(a.o + row * MLP_O + 8 * g4 + 4 * half) = MlV4{oacc[4 * g4], oacc[4 * g4 + 1], oacc[4 * g4 + 2], oacc[4 * g4 + 3]};
  }
  klacc = wave_sum(klacc);
  if (lane == 0) red[w] = klacc;
  __syncthreads();
  if (tid == 0) a.klpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) mlp2_kl_finish_kernel(const float* __restrict__ part, int np, float* __restrict__ kl) {
  __shared__ float smem[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) acc += part[i];
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) kl[0] = -0.5f * acc;
}

static size_t mlp2_fwd_lds(int din, int hid) { return (size_t)(din * hid + hid * MLP_O + hid + MLP_O + 4) * sizeof(float); }
extern "C" int hb_mlp2_sample_supported(long n, long din, long hid, long nout, long rng_lanes, int has_u) {
  const bool shape = (din == 32 || din == 64) && (hid == 128 || hid == 256) && nout == MLP_O && n > 0 && n % 32 == 0;
  return shape && (has_u || rng_lanes >= 2 * n) ? 1 : 0;
}
extern "C" long hb_mlp2_sample_ws_elems(long n, long din, long hid) {
  // forward: one KL partial per workgroup; backward: partial sums per row chunk (see hb_mlp2_sample_bwd_f32)
  const long chunks = 128;
  return 1024 + chunks * (din * hid + hid + hid * MLP_O + MLP_O);
}

extern "C" int hb_mlp2_sample_fwd_f32(const float* y, const float* w0, const float* b0, const float* w1, const float* b1, int act,
                                      const float* u_in, uint64_t* rng, long rng_lanes, float* x, float* kl, float* u_out,
                                      float* o, long n, long din, long hid, float* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  HB_REQUIRE(y && w0 && b0 && w1 && b1 && x && kl && u_out && o && ws, "hb_mlp2_sample_fwd: NULL pointer");
  HB_REQUIRE(hb_mlp2_sample_supported(n, din, hid, MLP_O, u_in ? 0 : rng_lanes, u_in != nullptr),
             "hb_mlp2_sample_fwd: unsupported shape n=%ld din=%ld hid=%ld (hb_mlp2_sample_supported)", n, din, hid);
  HB_REQUIRE(u_in || rng, "hb_mlp2_sample_fwd: neither u_in nor rng given");
  HB_REQUIRE(((uintptr_t)y | (uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)x | (uintptr_t)u_out | (uintptr_t)o | (uintptr_t)u_in) % 16 == 0,
             "hb_mlp2_sample_fwd: operands must be 16-byte aligned");
  Mlp2FwdArgs a = {y, w0, b0, w1, b1, u_in, u_in ? nullptr : rng, rng_lanes, x, u_out, o, ws, n, act};
  long g = (n / 32 + 3) / 4;
  if (g > 1024) g = 1024;
  const size_t lds = mlp2_fwd_lds((int)din, (int)hid);
#define HB_MLP_FWD(D_, H_)                                                                                        \
  do {                                                                                                            \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) {                                                                                              \
      HB_HIP(hipFuncSetAttribute((const void*)mlp2_fwd_kernel<D_, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL((mlp2_fwd_kernel<D_, H_>), dim3((unsigned)g), dim3(256), lds, stream, a);                  \
  } while (0)
  if (din == 64 && hid == 256) HB_MLP_FWD(64, 256);
  else if (din == 64 && hid == 128) HB_MLP_FWD(64, 128);
  else if (din == 32 && hid == 256) HB_MLP_FWD(32, 256);
  else HB_MLP_FWD(32, 128);
#undef HB_MLP_FWD
  HB_LAUNCH_CHECK();
  hipLaunchKernelGGL(mlp2_kl_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, (int)g, kl);
  HB_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
struct Mlp2BwdArgs {
  const float *y, *w0, *b0, *w1, *o, *u, *x, *xbar, *klbar;
  float* part;      // [chunks][DIN*HID + HID + HID*32 + 32]
  long n;
  int act, chunks;
};
#define MLP_DOLD 33   // row stride of the do tile in LDS (conflict-free row-per-lane reads)

template <int DIN, int HID>
__global__ void __launch_bounds__(256, 2) mlp2_bwd_kernel(Mlp2BwdArgs a) {
  typedef Mma<float> MM;
  constexpr int KH = DIN / 2, KT = DIN / 32, YLD = DIN + 1;
  __shared__ __attribute__((aligned(16))) float W0s[DIN][64];          // this workgroup's 64 hidden columns of W0
  __shared__ __attribute__((aligned(16))) float W1Ts[MLP_O][64 + 4];   // W1^T restricted to them
  __shared__ float b0s[64];
  __shared__ float dos[4][32 * MLP_DOLD];                              // per wave: do = [mubar | sbar] of its 32 rows
  __shared__ float ys[4][32 * YLD];                                    // per wave: its y tile, row major
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int hg = blockIdx.x, chunk = blockIdx.y;                       // hidden group (64 units), row chunk
  const int h0 = 64 * hg;
  for (int i = tid; i < DIN * 64; i += 256) W0s[i >> 6][i & 63] = a.w0[(size_t)(i >> 6) * HID + h0 + (i & 63)];
  for (int i = tid; i < 64 * MLP_O; i += 256) W1Ts[i & 31][i >> 5] = a.w1[(size_t)(h0 + (i >> 5)) * MLP_O + (i & 31)];
  if (tid < 64) b0s[tid] = a.b0[h0 + tid];
  __syncthreads();
  const int li = lane & 31, half = lane >> 5;
  const float kb = a.klbar ? a.klbar[0] : 0.f;
  typename MM::Acc dW0[KT][2], dW1[2];
  float db0[2] = {0.f, 0.f}, db1 = 0.f;
#pragma unroll
  for (int T = 0; T < 2; ++T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dW1[T][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW0[kt][T][r] = 0.f;
  }
  const long ntiles = a.n / 32;
  const long per = (ntiles + a.chunks - 1) / a.chunks;
  const long t_begin = (long)chunk * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
  float* dow = dos[w];
  float* yw = ys[w];
  for (long tile = t_begin + w; tile < t_end; tile += 4) {
    const long row = tile * 32 + li;
    // y: the A operand of layer 0 (lane = row, this half's KH contraction entries), also staged row major for dW0
    float yr[KH];
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const MlV4 t = *reinterpret_cast<const MlV4*>(a.y + row * DIN + half * KH + 4 * v);
      yr[4 * v] = t[0], yr[4 * v + 1] = t[1], yr[4 * v + 2] = t[2], yr[4 * v + 3] = t[3];
    }
#pragma unroll
    for (int s = 0; s < KH; ++s) yw[li * YLD + half * KH + s] = yr[s];
    // do: the sampler's VJP for this lane's row, latent dimensions 8 half .. 8 half + 7
    //   mubar = xbar + klbar x ;  sbar = mubar exp(s) u - klbar      (variational.hip: diag_bwd_body)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int l0 = 8 * half + 4 * v;
      const MlV4 xb = a.xbar ? *reinterpret_cast<const MlV4*>(a.xbar + row * MLP_L + l0) : MlV4{0.f, 0.f, 0.f, 0.f};
      const MlV4 xx = *reinterpret_cast<const MlV4*>(a.x + row * MLP_L + l0);
      const MlV4 uu = *reinterpret_cast<const MlV4*>(a.u + row * MLP_L + l0);
      const MlV4 ss = *reinterpret_cast<const MlV4*>(a.o + row * MLP_O + MLP_L + l0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float mb = xb[e] + kb * xx[e];
        dow[li * MLP_DOLD + l0 + e] = mb;
        dow[li * MLP_DOLD + MLP_L + l0 + e] = mb * hb_exp(ss[e]) * uu[e] - kb;
      }
    }
    // (wave-private LDS tiles: the LDS executes a wave's instructions in order; the compiler orders its own accesses)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (hg == 0 && lane < MLP_O) {
      float sdo = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) sdo += dow[rr * MLP_DOLD + lane];
      db1 += sdo;
    }
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      // (a) h tile, natural orientation: lane = hidden unit 32 T + li of the group, register r = row acc_row(lane, r)
      typename MM::Acc hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[r] = b0s[32 * T + li];
#pragma unroll
      for (int s = 0; s < KH; ++s) hh = MM::mma(yr[s], W0s[half * KH + s][32 * T + li], hh);
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[r] = mlp_act<float>(a.act, hh[r]);
      // (b) dW1[hidden][out] += sum_rows h[row][hidden] do[row][out]: the accumulator is the A operand as it stands
#pragma unroll
      for (int r = 0; r < 16; ++r) dW1[T] = MM::mma(hh[r], dow[MM::acc_row(lane, r) * MLP_DOLD + li], dW1[T]);
      // (c) dh = (do W1^T) o act'(h), same layout as h
      typename MM::Acc dh;
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
      for (int s = 0; s < MLP_O / 2; ++s)
        dh = MM::mma(dow[li * MLP_DOLD + half * (MLP_O / 2) + s], W1Ts[half * (MLP_O / 2) + s][32 * T + li], dh);
      float sb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dh[r] *= mlp_act_grad<float>(a.act, hh[r]);
        sb += dh[r];
      }
      db0[T] += sb;
      // (d) dW0[k][hidden] += sum_rows y[row][k] dh[row][hidden]: dh is the B operand as it stands
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW0[kt][T] = MM::mma(yw[MM::acc_row(lane, r) * YLD + 32 * kt + li], dh[r], dW0[kt][T]);
    }
  }
  // ---- fold the four waves (fixed order) and write this workgroup's partial sums
  __syncthreads();
  float* red = &ys[0][0];   // reused (4 x 32 x (DIN + 1) floats >= DIN * 64 and >= 64 * 32 + 96)
  float* part = a.part + (size_t)chunk * ((size_t)DIN * HID + HID + HID * MLP_O + MLP_O);
  float* pW0 = part;
  float* pb0 = pW0 + (size_t)DIN * HID;
  float* pW1 = pb0 + HID;
  float* pb1 = pW1 + (size_t)HID * MLP_O;
  // dW0: [DIN][64] slice; accumulator (kt, T) register r of lane (li, half) = dW0[32 kt + acc_row][32 T + li]
  for (int pass = 0; pass < 4; ++pass) {
    if (w == pass) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* p = &red[(32 * kt + MM::acc_row(lane, r)) * 64 + 32 * T + li];
            *p = pass == 0 ? dW0[kt][T][r] : *p + dW0[kt][T][r];
          }
    }
    __syncthreads();
  }
  for (int i = tid; i < DIN * 64; i += 256) pW0[(size_t)(i >> 6) * HID + h0 + (i & 63)] = red[i];
  __syncthreads();
  // dW1: [64][32] slice; accumulator T register r of lane = dW1[32 T + acc_row][li]; db0: hidden 32 T + li, both halves
  for (int pass = 0; pass < 4; ++pass) {
    if (w == pass) {
#pragma unroll
      for (int T = 0; T < 2; ++T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* p = &red[(32 * T + MM::acc_row(lane, r)) * MLP_O + li];
          *p = pass == 0 ? dW1[T][r] : *p + dW1[T][r];
        }
        const float both = db0[T] + __shfl_xor(db0[T], 32, 64);
        if (half == 0) {
          float* p = &red[64 * MLP_O + 32 * T + li];
          *p = pass == 0 ? both : *p + both;
        }
      }
      if (hg == 0 && lane < MLP_O) {
        float* p = &red[64 * MLP_O + 64 + lane];
        *p = pass == 0 ? db1 : *p + db1;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < 64 * MLP_O; i += 256) pW1[(size_t)(h0 + (i >> 5)) * MLP_O + (i & 31)] = red[i];
  if (tid < 64) pb0[h0 + tid] = red[64 * MLP_O + tid];
  if (hg == 0 && tid < MLP_O) pb1[tid] = red[64 * MLP_O + 64 + tid];
}

// out[e] = sum over the row chunks of part[chunk][e], fixed order, four accumulators in flight
__global__ void __launch_bounds__(256) mlp2_bwd_finish_kernel(const float* __restrict__ part, int chunks, long stride, float* __restrict__ dw0,
                                                              long nw0, float* __restrict__ db0, long nb0, float* __restrict__ dw1,
                                                              long nw1, float* __restrict__ db1, long nb1) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= stride) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    s0 += part[(size_t)c * stride + e];
    s1 += part[(size_t)(c + 1) * stride + e];
    s2 += part[(size_t)(c + 2) * stride + e];
    s3 += part[(size_t)(c + 3) * stride + e];
  }
  for (; c < chunks; ++c) s0 += part[(size_t)c * stride + e];
  const float v = (s0 + s1) + (s2 + s3);
  if (e < nw0) dw0[e] = v;
  else if (e < nw0 + nb0) db0[e - nw0] = v;
  else if (e < nw0 + nb0 + nw1) dw1[e - nw0 - nb0] = v;
  else db1[e - nw0 - nb0 - nw1] = v;
}

extern "C" int hb_mlp2_sample_bwd_f32(const float* y, const float* w0, const float* b0, const float* w1, int act, const float* o,
                                      const float* u, const float* x, const float* xbar, const float* klbar, float* dw0,
                                      float* db0, float* dw1, float* db1, long n, long din, long hid, float* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  HB_REQUIRE(y && w0 && b0 && w1 && o && u && x && dw0 && db0 && dw1 && db1 && ws, "hb_mlp2_sample_bwd: NULL pointer");
  HB_REQUIRE(hb_mlp2_sample_supported(n, din, hid, MLP_O, 0, 1), "hb_mlp2_sample_bwd: unsupported shape n=%ld din=%ld hid=%ld", n, din, hid);
  HB_REQUIRE(((uintptr_t)y | (uintptr_t)o | (uintptr_t)u | (uintptr_t)x | (uintptr_t)xbar) % 16 == 0, "hb_mlp2_sample_bwd: operands must be 16-byte aligned");
  const long ntiles = n / 32;
  int chunks = (int)(ntiles / 8 < 1 ? 1 : ntiles / 8);   // two 32-row tiles per wave and chunk
  if (chunks > 128) chunks = 128;
  Mlp2BwdArgs a = {y, w0, b0, w1, o, u, x, xbar, klbar, ws + 1024, n, act, chunks};
  const dim3 grid((unsigned)(hid / 64), (unsigned)chunks);
  if (din == 64 && hid == 256) hipLaunchKernelGGL((mlp2_bwd_kernel<64, 256>), grid, dim3(256), 0, stream, a);
  else if (din == 64 && hid == 128) hipLaunchKernelGGL((mlp2_bwd_kernel<64, 128>), grid, dim3(256), 0, stream, a);
  else if (din == 32 && hid == 256) hipLaunchKernelGGL((mlp2_bwd_kernel<32, 256>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((mlp2_bwd_kernel<32, 128>), grid, dim3(256), 0, stream, a);
  HB_LAUNCH_CHECK();
  const long stride = din * hid + hid + hid * MLP_O + MLP_O;
  hipLaunchKernelGGL(mlp2_bwd_finish_kernel, dim3((unsigned)hb_cdiv(stride, 256)), dim3(256), 0, stream, (const float*)(ws + 1024), chunks,
                     stride, dw0, din * hid, db0, hid, dw1, hid * MLP_O, db1, (long)MLP_O);
  HB_LAUNCH_CHECK();
  return 0;
}
