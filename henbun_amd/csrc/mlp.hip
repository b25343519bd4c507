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

#ifdef HB_MLP_STAMPS   // diagnostic build (tools/mlp_stamps.hip): s_memtime stamps held in registers, dumped at the end
extern unsigned long long* hb_mlp_stamps_buffer;
#define MLP_STAMP(i) mst[i] = __builtin_amdgcn_s_memtime()
#define MLP_STAMP_DECL unsigned long long mst[24]; _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) mst[i_] = 0;
#define MLP_STAMP_DUMP(buf, wgid)                                                                      \
  if ((buf) && lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) (buf)[((size_t)(wgid) * 4 + w) * 24 + i_] = mst[i_]; }
#else
#define MLP_STAMP(i)
#define MLP_STAMP_DECL
#define MLP_STAMP_DUMP(buf, wgid)
#endif

// The activation is a TEMPLATE parameter: switched per element at run time inside the unrolled MFMA loops it becomes a
// branch ladder per accumulator register (12 000 lines of ISA for the forward kernel, ~150 cycles per MFMA).
template <int ACT>
__device__ __forceinline__ float mlp_act(float v) {
  if (ACT == HB_ACT_SIGMOID) return hb_sigmoid(v);
  if (ACT == HB_ACT_RELU) return v > 0.f ? v : 0.f;
  if (ACT == HB_ACT_TANH) return hb_tanh(v);
  return v;
}
template <int ACT>
__device__ __forceinline__ float mlp_act_grad(float y) {   // through the activation's OUTPUT
  if (ACT == HB_ACT_SIGMOID) return y * (1.f - y);
  if (ACT == HB_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (ACT == HB_ACT_TANH) return 1.f - y * y;
  return 1.f;
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

template <int DIN, int HID, int ACT>
__global__ void __launch_bounds__(256) mlp2_fwd_kernel(Mlp2FwdArgs a) {
  typedef Mma<float> MM;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* W0s = sm;                    // [DIN][HID]
  float* W1s = W0s + DIN * HID;       // [HID][32]
  float* b0s = W1s + HID * MLP_O;     // [HID]
  float* b1s = b0s + HID;             // [32]
  float* red = b1s + MLP_O;           // [4]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  constexpr int HT = HID / 32, KH = DIN / 2;   // hidden tiles; contraction entries per lane half
  const long ntiles = a.n / 32;
  // the y fragments of the wave's first tile are requested BEFORE the weights are staged (one wave per SIMD: nothing else
  // hides an HBM round trip)
  MlV4 yv[KH / 4];
  {
    const long tile0 = (long)blockIdx.x * 4 + w;
    const long row0 = (tile0 < ntiles ? tile0 : 0) * 32 + li;
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) yv[v] = *reinterpret_cast<const MlV4*>(a.y + row0 * DIN + half * KH + 4 * v);
  }
  for (int i = tid; i < DIN * HID / 4; i += 256) reinterpret_cast<MlV4*>(W0s)[i] = reinterpret_cast<const MlV4*>(a.w0)[i];
  for (int i = tid; i < HID * MLP_O / 4; i += 256) reinterpret_cast<MlV4*>(W1s)[i] = reinterpret_cast<const MlV4*>(a.w1)[i];
  for (int i = tid; i < HID; i += 256) b0s[i] = a.b0[i];
  if (tid < MLP_O) b1s[tid] = a.b1[tid];
  __syncthreads();
  float klacc = 0.f;
  for (long tile = (long)blockIdx.x * 4 + w; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + li;
    // B operand of layer 0: this lane's row, entries [half KH, half KH + KH) of the contraction (permuted k: 16-byte loads)
    float yr[KH];
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) yr[4 * v] = yv[v][0], yr[4 * v + 1] = yv[v][1], yr[4 * v + 2] = yv[v][2], yr[4 * v + 3] = yv[v][3];
    {
      // (the next tile of this wave, if any: requested now, consumed after this tile's 384 MFMAs)
      const long nt = tile + (long)gridDim.x * 4;
      const long nrow = (nt < ntiles ? nt : tile) * 32 + li;
#pragma unroll
      for (int v = 0; v < KH / 4; ++v) yv[v] = *reinterpret_cast<const MlV4*>(a.y + nrow * DIN + half * KH + 4 * v);
    }
    typename MM::Acc acc[HT];
#pragma unroll
    for (int T = 0; T < HT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[T][r] = b0s[32 * T + MM::acc_row(lane, r)];
    // (operands of step s + 1 are read while step s runs; the scheduling barriers keep the compiler from hoisting ALL
    // LDS reads of the unrolled loop to its top: 256 VGPRs + 256 AGPRs and 138 spills without them)
    {
      const float* pW0 = &W0s[(half * KH) * HID + li];   // + compile-time offsets: immediate-offset LDS reads
      float wc[HT], wn[HT];
#pragma unroll
      for (int T = 0; T < HT; ++T) wc[T] = pW0[32 * T];
#pragma unroll
      for (int s = 0; s < KH; ++s) {
        if (s + 1 < KH) {
#pragma unroll
          for (int T = 0; T < HT; ++T) wn[T] = pW0[(s + 1) * HID + 32 * T];
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
      const float* pW1 = &W1s[(4 * half) * MLP_O + li];
      float vc[16], vn[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vc[r] = pW1[((r & 3) + 8 * (r >> 2)) * MLP_O];
#pragma unroll
      for (int T = 0; T < HT; ++T) {
        if (T + 1 < HT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vn[r] = pW1[(32 * (T + 1) + (r & 3) + 8 * (r >> 2)) * MLP_O];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float h = mlp_act<ACT>(acc[T][r]);
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
#define HB_MLP_FWD3(D_, H_, A_)                                                                                   \
  do {                                                                                                            \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) {                                                                                              \
      HB_HIP(hipFuncSetAttribute((const void*)mlp2_fwd_kernel<D_, H_, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL((mlp2_fwd_kernel<D_, H_, A_>), dim3((unsigned)g), dim3(256), lds, stream, a);              \
  } while (0)
#define HB_MLP_FWD(D_, H_)                                     \
  do {                                                         \
    if (act == HB_ACT_SIGMOID) HB_MLP_FWD3(D_, H_, HB_ACT_SIGMOID); \
    else if (act == HB_ACT_RELU) HB_MLP_FWD3(D_, H_, HB_ACT_RELU);  \
    else HB_MLP_FWD3(D_, H_, HB_ACT_TANH);                     \
  } while (0)
  HB_REQUIRE(act == HB_ACT_SIGMOID || act == HB_ACT_RELU || act == HB_ACT_TANH, "hb_mlp2_sample_fwd: activation %d", act);
  if (din == 64 && hid == 256) HB_MLP_FWD(64, 256);
  else if (din == 64 && hid == 128) HB_MLP_FWD(64, 128);
  else if (din == 32 && hid == 256) HB_MLP_FWD(32, 256);
  else HB_MLP_FWD(32, 128);
#undef HB_MLP_FWD
#undef HB_MLP_FWD3
  HB_LAUNCH_CHECK();
  hipLaunchKernelGGL(mlp2_kl_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, (int)g, kl);
  HB_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
struct Mlp2BwdArgs {
  unsigned long long* stamps;
  const float *y, *w0, *b0, *w1, *o, *u, *x, *xbar, *klbar;
  float* part;      // [chunks][DIN*HID + HID + HID*32 + 32]
  long n;
  int act, chunks;
};
#define MLP_DOLD 33   // row stride of the do tile in LDS (conflict-free row-per-lane reads)

// OCC: waves per SIMD the register allocation aims at (2: two workgroups per CU; the [64, 256] instance needs ~280
// registers with its operand prefetch buffers and runs one workgroup per CU instead of spilling)
template <int DIN, int HID, int OCC, int ACT>
__global__ void __launch_bounds__(256, OCC) mlp2_bwd_kernel(Mlp2BwdArgs a) {
  typedef Mma<float> MM;
  constexpr int KH = DIN / 2, KT = DIN / 32, YLD = DIN + 1;
  __shared__ __attribute__((aligned(16))) float W0s[DIN][64];          // this workgroup's 64 hidden columns of W0
  __shared__ __attribute__((aligned(16))) float W1Ts[MLP_O][64 + 4];   // W1^T restricted to them
  __shared__ float b0s[64];
  // per wave: its y tile (row major) and do = [mubar | sbar] of its 32 rows; one array: the final reduction reuses it
  constexpr int YS = 32 * YLD, DS = 32 * MLP_DOLD;
  __shared__ float tiles[4 * (YS + DS)];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hg = blockIdx.x, chunk = blockIdx.y;                       // hidden group (64 units), row chunk
  const int h0 = 64 * hg;
  MLP_STAMP_DECL
  MLP_STAMP(0);
  {
    // (16-byte loads, all requested before the first is stored: the scalar form was 16 + 8 dependent round trips, 3.6 us)
    MlV4 t0[DIN * 16 / 256], t1[2];
#pragma unroll
    for (int k = 0; k < DIN * 16 / 256; ++k) {
      const int i = tid + 256 * k;     // group i: row i >> 4, columns 4 (i & 15) ..
      t0[k] = *reinterpret_cast<const MlV4*>(a.w0 + (size_t)(i >> 4) * HID + h0 + 4 * (i & 15));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + 256 * k;     // group i: hidden unit i >> 3 of the group, outputs 4 (i & 7) ..
      t1[k] = *reinterpret_cast<const MlV4*>(a.w1 + (size_t)(h0 + (i >> 3)) * MLP_O + 4 * (i & 7));
    }
#pragma unroll
    for (int k = 0; k < DIN * 16 / 256; ++k) {
      const int i = tid + 256 * k;
      *reinterpret_cast<MlV4*>(&W0s[i >> 4][4 * (i & 15)]) = t0[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + 256 * k;
#pragma unroll
      for (int e = 0; e < 4; ++e) W1Ts[4 * (i & 7) + e][i >> 3] = t1[k][e];
    }
  }
  if (tid < 64) b0s[tid] = a.b0[h0 + tid];
  __syncthreads();
  MLP_STAMP(1);
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
  float* yw = tiles + w * (YS + DS);
  float* dow = yw + YS;
  // The raw operands of a tile (its y rows; xbar, x, u, s of its rows for the sampler's VJP) are REQUESTED one tile ahead,
  // right before the previous tile's 192 MFMAs, and turned into the two LDS tiles at the top of their own iteration: one
  // wave per SIMD has nothing else to hide an HBM round trip with (stamps: 4 800 of 20 200 cycles per tile before).
  MlV4 ry[KH / 4], rxb[2], rxx[2], ruu[2], rss[2];
  auto request = [&](long tile_) {
    const long row_ = tile_ * 32 + li;
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) ry[v] = *reinterpret_cast<const MlV4*>(a.y + row_ * DIN + half * KH + 4 * v);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int l0 = 8 * half + 4 * v;
      rxb[v] = a.xbar ? *reinterpret_cast<const MlV4*>(a.xbar + row_ * MLP_L + l0) : MlV4{0.f, 0.f, 0.f, 0.f};
      rxx[v] = *reinterpret_cast<const MlV4*>(a.x + row_ * MLP_L + l0);
      ruu[v] = *reinterpret_cast<const MlV4*>(a.u + row_ * MLP_L + l0);
      rss[v] = *reinterpret_cast<const MlV4*>(a.o + row_ * MLP_O + MLP_L + l0);
    }
  };
  if (t_begin + w < t_end) request(t_begin + w);
  int tcount = 0;
  for (long tile = t_begin + w; tile < t_end; tile += 4, ++tcount) {
    if (tcount == 1) MLP_STAMP(2);
    // y: staged row major (the A operand of layer 0 and of dW0 both come from this tile)
#pragma unroll
    for (int v = 0; v < KH / 4; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) yw[li * YLD + half * KH + 4 * v + e] = ry[v][e];
    // do: the sampler's VJP for this lane's row, latent dimensions 8 half .. 8 half + 7
    //   mubar = xbar + klbar x ;  sbar = mubar exp(s) u - klbar      (variational.hip: diag_bwd_body)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int l0 = 8 * half + 4 * v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float mb = rxb[v][e] + kb * rxx[v][e];
        dow[li * MLP_DOLD + l0 + e] = mb;
        dow[li * MLP_DOLD + MLP_L + l0 + e] = mb * hb_exp(rss[v][e]) * ruu[v][e] - kb;
      }
    }
    if (tile + 4 < t_end) request(tile + 4);
    // (wave-private LDS tiles: the LDS executes a wave's instructions in order; the compiler orders its own accesses)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (hg == 0 && lane < MLP_O) {
      float sdo = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) sdo += dow[rr * MLP_DOLD + lane];
      db1 += sdo;
    }
    // Every MFMA below takes one operand from LDS.  Left to the compiler the reads sit right in front of their MFMAs
    // (register pressure: 240 VGPRs), and a wave then pays an LDS round trip per MFMA (50.6 us per launch at cfg 4,
    // 0.40 of the fp32 MFMA peak).  The 16 reads of the NEXT block of 16 MFMAs are issued before the current block runs.
    float pa[16], pb[16], pc[16], pd[16];
    // (every LDS operand address = one of six per-lane bases + a compile-time constant, so that the reads carry immediate
    // offsets: indexed by expressions, each of the ~330 reads of a tile got an address register of its own, parked in
    // the accumulator file and fetched back with v_accvgpr_read in front of its read: ~580 extra instructions per tile)
    const float* pW0 = &W0s[half * KH][li];
    const float* pyA = &yw[li * YLD + half * KH];
    const float* pdoR = &dow[(4 * half) * MLP_DOLD + li];
    const float* pdoL = &dow[li * MLP_DOLD + half * (MLP_O / 2)];
    const float* pW1 = &W1Ts[half * (MLP_O / 2)][li];
    const float* pyR = &yw[(4 * half) * YLD + li];
#define MLP_ROWOFF(i_) (((i_) & 3) + 8 * ((i_) >> 2))   /* acc_row(lane, i) - 4 half */
#define MLP_LOAD_YA(BUF, S0_)                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) BUF[i_] = pyA[(S0_) + i_];
#define MLP_LOAD_A(BUF, T_, S0_)                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) BUF[i_] = pW0[((S0_) + i_) * 64 + 32 * (T_)];
#define MLP_LOAD_DO_ROWS(BUF)                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) BUF[i_] = pdoR[MLP_ROWOFF(i_) * MLP_DOLD];
#define MLP_LOAD_DO_LANE(BUF)                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) BUF[i_] = pdoL[i_];
#define MLP_LOAD_W1T(BUF, T_)                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) BUF[i_] = pW1[i_ * (64 + 4) + 32 * (T_)];
#define MLP_LOAD_Y(BUF, KT_)                                                 \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) BUF[i_] = pyR[MLP_ROWOFF(i_) * YLD + 32 * (KT_)];
    static_assert(KH == 16 || KH == 32, "mlp2_bwd_kernel: Din in {32, 64}");
    if (tcount == 1) MLP_STAMP(3);
    MLP_LOAD_A(pa, 0, 0)
    MLP_LOAD_YA(pd, 0)
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      // (a) h tile, natural orientation: lane = hidden unit 32 T + li of the group, register r = row acc_row(lane, r)
      typename MM::Acc hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[r] = b0s[32 * T + li];
      if (KH == 32) {
        MLP_LOAD_A(pb, T, 16)
        MLP_LOAD_YA(pc, 16)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) hh = MM::mma(pd[i], pa[i], hh);
        __builtin_amdgcn_sched_barrier(0);
        MLP_LOAD_DO_ROWS(pa)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) hh = MM::mma(pc[i], pb[i], hh);
      } else {
        MLP_LOAD_DO_ROWS(pb)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) hh = MM::mma(pd[i], pa[i], hh);
#pragma unroll
        for (int i = 0; i < 16; ++i) pa[i] = pb[i];
      }
      if (tcount == 1 && T == 0) MLP_STAMP(4);
      __builtin_amdgcn_sched_barrier(0);
      MLP_LOAD_DO_LANE(pb)
      MLP_LOAD_W1T(pc, T)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[r] = mlp_act<ACT>(hh[r]);
      // (b) dW1[hidden][out] += sum_rows h[row][hidden] do[row][out]: the accumulator is the A operand as it stands
#pragma unroll
      for (int r = 0; r < 16; ++r) dW1[T] = MM::mma(hh[r], pa[r], dW1[T]);
      if (tcount == 1 && T == 0) MLP_STAMP(5);
      __builtin_amdgcn_sched_barrier(0);
      MLP_LOAD_Y(pa, 0)
      __builtin_amdgcn_sched_barrier(0);
      // (c) dh = (do W1^T) o act'(h), same layout as h
      typename MM::Acc dh;
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
      for (int i = 0; i < MLP_O / 2; ++i) dh = MM::mma(pb[i], pc[i], dh);
      float sb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dh[r] *= mlp_act_grad<ACT>(hh[r]);
        sb += dh[r];
      }
      db0[T] += sb;
      if (tcount == 1 && T == 0) MLP_STAMP(6);
      // (d) dW0[k][hidden] += sum_rows y[row][k] dh[row][hidden]: dh is the B operand as it stands
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < KT) {
          MLP_LOAD_Y(pb, kt + 1)
        } else if (T == 0) {
          MLP_LOAD_A(pb, 1, 0)
          MLP_LOAD_YA(pd, 0)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) dW0[kt][T] = MM::mma(pa[r], dh[r], dW0[kt][T]);
#pragma unroll
        for (int i = 0; i < 16; ++i) pa[i] = pb[i];
      }
    }
#undef MLP_LOAD_A
#undef MLP_ROWOFF
#undef MLP_LOAD_YA
#undef MLP_LOAD_DO_ROWS
#undef MLP_LOAD_DO_LANE
#undef MLP_LOAD_W1T
#undef MLP_LOAD_Y
    if (tcount == 1) MLP_STAMP(7);
  }
  MLP_STAMP(8);
  // ---- fold the four waves and write this workgroup's partial sums.  Two regions of the (now free) tile area: waves 0 and
  // 1 store into regions 0 and 1, waves 2 and 3 add into them, every thread then sums the two while writing out:
  // ((w0 + w2) + (w1 + w3)), fixed order, two passes instead of four.
  __syncthreads();
  constexpr int NW0 = DIN * 64, NW1 = 64 * MLP_O, RSZ = NW0 + NW1 + 64 + MLP_O;
  static_assert(2 * RSZ <= 4 * (YS + DS), "mlp2_bwd_kernel: reduction regions exceed the tile area");
  float* part = a.part + (size_t)chunk * ((size_t)DIN * HID + HID + HID * MLP_O + MLP_O);
  float* pW0 = part;
  float* pb0 = pW0 + (size_t)DIN * HID;
  float* pW1 = pb0 + HID;
  float* pb1 = pW1 + (size_t)HID * MLP_O;
  // accumulator (kt, T) register r of lane (li, half) = dW0[32 kt + acc_row][32 T + li]; dW1[32 T + acc_row][li];
  // db0: hidden 32 T + li, both halves
  for (int pass = 0; pass < 2; ++pass) {
    if ((w >> 1) == pass) {
      float* red = tiles + (w & 1) * RSZ;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* p = &red[(32 * kt + MM::acc_row(lane, r)) * 64 + 32 * T + li];
            *p = pass == 0 ? dW0[kt][T][r] : *p + dW0[kt][T][r];
          }
#pragma unroll
      for (int T = 0; T < 2; ++T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* p = &red[NW0 + (32 * T + MM::acc_row(lane, r)) * MLP_O + li];
          *p = pass == 0 ? dW1[T][r] : *p + dW1[T][r];
        }
        const float both = db0[T] + __shfl_xor(db0[T], 32, 64);
        if (half == 0) {
          float* p = &red[NW0 + NW1 + 32 * T + li];
          *p = pass == 0 ? both : *p + both;
        }
      }
      if (lane < MLP_O) {
        float* p = &red[NW0 + NW1 + 64 + lane];
        *p = pass == 0 ? db1 : *p + db1;
      }
    }
    __syncthreads();
  }
  const float* r0 = tiles;
  const float* r1 = tiles + RSZ;
  for (int i = tid; i < NW0 / 4; i += 256) {
    const MlV4 va = *reinterpret_cast<const MlV4*>(r0 + 4 * i), vb = *reinterpret_cast<const MlV4*>(r1 + 4 * i);
    *reinterpret_cast<MlV4*>(pW0 + (size_t)(i >> 4) * HID + h0 + 4 * (i & 15)) = va + vb;
  }
  for (int i = tid; i < NW1 / 4; i += 256) {
    const MlV4 va = *reinterpret_cast<const MlV4*>(r0 + NW0 + 4 * i), vb = *reinterpret_cast<const MlV4*>(r1 + NW0 + 4 * i);
    *reinterpret_cast<MlV4*>(pW1 + (size_t)(h0 + (i >> 3)) * MLP_O + 4 * (i & 7)) = va + vb;
  }
  if (tid < 64) pb0[h0 + tid] = r0[NW0 + NW1 + tid] + r1[NW0 + NW1 + tid];
  if (hg == 0 && tid < MLP_O) pb1[tid] = r0[NW0 + NW1 + 64 + tid] + r1[NW0 + NW1 + 64 + tid];
  MLP_STAMP(9);
  MLP_STAMP_DUMP(a.stamps, blockIdx.y * gridDim.x + blockIdx.x)
}

// out[e] = sum over the row chunks of part[chunk][e], fixed order, four accumulators in flight
__global__ void __launch_bounds__(256) mlp2_bwd_finish_kernel(const float* __restrict__ part, int chunks, long stride, float* __restrict__ dw0,
                                                              long nw0, float* __restrict__ db0, long nb0, float* __restrict__ dw1,
                                                              long nw1, float* __restrict__ db1, long nb1) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= stride) return;
  // sixteen partials in flight per round (four made this launch four dependent-load rounds per 16 chunks: 11.2 us)
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  int c = 0;
  for (; c + 16 <= chunks; c += 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += part[(size_t)(c + i) * stride + e];
  }
  for (; c < chunks; ++c) acc[0] += part[(size_t)c * stride + e];
  const float v = (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))) +
                  (((acc[8] + acc[9]) + (acc[10] + acc[11])) + ((acc[12] + acc[13]) + (acc[14] + acc[15])));
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
  // ~440 registers with the operand prefetch buffers and the next tile's raw operands: one workgroup per CU (at two
  // waves per SIMD the allocation spills 20 .. 200 registers).  Row chunks so that the grid is about 256 workgroups, at
  // least one 32-row tile per wave, at most 128 chunks (the workspace)
  long want = 256L / (hid / 64);
  if (want > ntiles / 4) want = ntiles / 4;
  int chunks = (int)(want < 1 ? 1 : (want > 128 ? 128 : want));
#ifdef HB_MLP_STAMPS
  unsigned long long* stamps = hb_mlp_stamps_buffer;
#else
  unsigned long long* stamps = nullptr;
#endif
  Mlp2BwdArgs a = {stamps, y, w0, b0, w1, o, u, x, xbar, klbar, ws + 1024, n, act, chunks};
  const dim3 grid((unsigned)(hid / 64), (unsigned)chunks);
  HB_REQUIRE(act == HB_ACT_SIGMOID || act == HB_ACT_RELU || act == HB_ACT_TANH, "hb_mlp2_sample_bwd: activation %d", act);
#define HB_MLP_BWD(D_, H_, O_)                                                                                     \
  do {                                                                                                             \
    if (act == HB_ACT_SIGMOID) hipLaunchKernelGGL((mlp2_bwd_kernel<D_, H_, O_, HB_ACT_SIGMOID>), grid, dim3(256), 0, stream, a); \
    else if (act == HB_ACT_RELU) hipLaunchKernelGGL((mlp2_bwd_kernel<D_, H_, O_, HB_ACT_RELU>), grid, dim3(256), 0, stream, a);  \
    else hipLaunchKernelGGL((mlp2_bwd_kernel<D_, H_, O_, HB_ACT_TANH>), grid, dim3(256), 0, stream, a);            \
  } while (0)
  if (din == 64 && hid == 256) HB_MLP_BWD(64, 256, 1);
  else if (din == 64 && hid == 128) HB_MLP_BWD(64, 128, 1);
  else if (din == 32 && hid == 256) HB_MLP_BWD(32, 256, 1);
  else HB_MLP_BWD(32, 128, 1);
#undef HB_MLP_BWD
  HB_LAUNCH_CHECK();
  const long stride = din * hid + hid + hid * MLP_O + MLP_O;
  hipLaunchKernelGGL(mlp2_bwd_finish_kernel, dim3((unsigned)hb_cdiv(stride, 256)), dim3(256), 0, stream, (const float*)(ws + 1024), chunks,
                     stride, dw0, din * hid, db0, hid, dw1, hid * MLP_O, db1, (long)MLP_O);
  HB_LAUNCH_CHECK();
  return 0;
}
