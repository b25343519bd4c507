// Cholesky (+ inverse) as ONE persistent launch (fp32, M % 64 == 0); round 4.
//
// The 8-launch right-looking chain (chol_rl64_kernel) spends half of every launch on things that exist only because
// the factorisation is cut into launches: the kernel boundary, reloading the trailing matrix, storing it again
// (profiles/r03_chol_panel_phase_stamps.txt: ~16 000 of ~33 000 cycles per launch), and its in-panel phase is a chain
// paced by ONE wave (elimination of 8 columns, then four dependent MFMAs, ~1 750 cycles per 8 columns).  Here:
//
//   * Every workgroup OWNS its tiles for the whole factorisation.  Workgroup (j, s) holds, in registers, column
//     block j (64 columns) of one 64-row block of the stacked matrix [A; I] (its "strip": an A row block i > j, or a
//     row block i' <= j of the identity, whose elimination yields Y = L^-T) plus -- redundantly, like the factor
//     workgroups of the chain -- the 64 x 64 diagonal block (j, j).  nb = M / 64 strips per column block, nb^2
//     workgroups per matrix, 8 waves each: wave (r, q) holds 64 rows (r = 0 diagonal block, r = 1 strip) x 16
//     columns (group q).  Nothing but finished panels ever travels.
//   * A finished panel travels through an EXCHANGE buffer (the workspace: nb^2 blocks of 64 x 64 per matrix, exactly
//     B M^2 elements) in the order its consumers' MFMA operand loads want it, 16 columns at a time: the strip wave
//     that has finished column group q writes its 4 KB with write-through (sc1) stores, drains them (vmcnt(0)) and
//     raises that chunk's flag word; a consumer wave polls the flag (relaxed agent-scope load, s_sleep) and then
//     reads the chunk with sc1 loads -- the hand-off form of cdna_hip_programming.md Guideline 16 R1 with one storing
//     wave per flag, no L2 write-back (buffer_wbl2) and no L1 invalidate anywhere.  Consumers of column block j + 1
//     apply chunk q of panel j while chunk q + 1 is still being factored: when the last chunk lands, one rank-16
//     update (16 MFMAs) separates it from the next in-panel phase.
//   * The trailing update runs on v_mfma_f32_16x16x4_f32 with the matrix ROW on the lane (lane & 15): a wave's four
//     16 x 16 accumulators become "lane = row, 16 columns in 16 registers" with eight v_permlane32_swap + eight
//     v_permlane16_swap (a 4 x 4 block transpose over the lane groups), no LDS.
//   * In-panel phase, row per lane: the owner of a column group eliminates its 16 columns with v_readlane (pivot and
//     multipliers are rows of its own lanes) and publishes every finished column to LDS at once (64 values + the
//     reciprocal pivot, then a monotonic counter; LDS executes a wave's instructions in order, so no wait sits
//     between them); all other waves follow column by column -- L_ic from the published column (their own rows:
//     per-lane read), the multipliers L[c2][c] of their 16 columns as uniform 16-byte reads, 16 FMAs -- on the other
//     three SIMDs.  The chain is one readlane -> rsq -> mul -> readlane -> fma sequence per column plus one LDS round
//     trip per change of owner, instead of elimination + MFMA update alternating in one wave.
//   * Deadlock freedom does not depend on dispatch order or residency: a workgroup takes its identity from a ticket
//     (atomic counter) in START order, identities are numbered column block by column block, and a workgroup waits
//     only for workgroups of earlier column blocks -- i.e. only for tickets that have already started.  Every spin
//     is bounded (s_memrealtime deadline): on expiry the launch sets a timeout word, every wait falls through, and
//     info[] = -1.
//   * State: flags, ticket, failure words live behind the exchange area in the caller's workspace.  They must be zero
//     when the kernel starts; the LAST workgroup to finish (arrival counter) writes info[] and zeroes them again, so a
//     workspace is zero-filled once by its owner and then reused call after call (graph replays included).
#pragma once
#include "common.cuh"
#include "side_jobs.cuh"
#include "gram_value.cuh"

#define CP_NB 64       // panel width = rows per row block
#define CP_G 16        // columns per wave
#define CP_LD 68       // LDS stride of a published column (floats): 16-byte aligned rows, conflict-free transposed reads
#define CP_HDR 4       // sync words: ticket, finished, timeout, reserved
#define CP_FAILBIG 0x40000000u
#define CP_TIMEOUT_TICKS 200000000ull   // s_memrealtime ticks (100 MHz): 2 s

typedef float CpV4 __attribute__((ext_vector_type(4)));
typedef unsigned CpU4 __attribute__((ext_vector_type(4)));
typedef float CpV2 __attribute__((ext_vector_type(2)));

struct CpArgs {
  const float* A;
  float* L;
  float* W;          // nullptr: plain factorisation (no identity rows)
  float* X;          // exchange area, B*M*M floats
  unsigned* sync;    // CP_HDR + roundup4(B) + 4*B*nb*nb words, zero at entry, zero at exit
  int* info;
  // A == nullptr: the matrix is K(X, X) + gdiag I of a stationary kernel, synthesised tile by tile (gram_value.cuh)
  const float* gX;
  const float* gell;
  long gsX, gsEll, gdl, gd;
  int gkind;
  float gdiag;
  float* Wf;         // nullable: the fragment-major images of W and W^T (and the bf16x3 planes behind them)
  int bf16x3;
  int M, B, nb, total;   // total = number of workgroups of the factorisation (side-job blocks come after them)
  // early-start consumers in the same launch (csrc/sgp.hip: chol_sgp_fwd_kernel).  early & 1 (the other bits are diagnostic switches): the W image is stored
  // write-through and every identity-strip workgroup of column block j raises wready[b][j] behind its stores (row block j
  // of W is final after panel j: j + 1 contributions); `arrive` workgroups take part in the arrival count (the
  // factorisation's plus the consumers'), `nside` side-job workgroups raise sync[3] when their outputs are released.
  int early, nside, arrive;
  int poll_naps;   // consumers: naps of 512 cycles between two polls of a row-block counter
  unsigned long long* stamps;   // diagnostic builds only (HB_CP_STAMPS)
};

// (the row-block counters of the early-start consumers sit in cache lines of their own: they are polled by every consumer
// workgroup, the panel flags in front of them are the factorisation's critical path)
__host__ __device__ static inline long cp_wready_off(long B, long nb) { return ((CP_HDR + ((B + 3) / 4) * 4 + 4 * B * nb * nb + 31) / 32) * 32 + 32; }
__host__ __device__ static inline long cp_sync_words_nb(long B, long nb) { return cp_wready_off(B, nb) + ((B * nb + 31) / 32) * 32; }
static inline long cp_sync_words(long B, long M) { return cp_sync_words_nb(B, M / CP_NB); }
static inline int cp_strips(int nb, int j, int inv) {
  const int n = (nb - 1 - j) + (inv ? j + 1 : 0);
  return n > 0 ? n : 1;
}
static inline long cp_total(long B, int nb, int inv) {
  long t = 0;
  for (int j = 0; j < nb; ++j) t += B * cp_strips(nb, j, inv);
  return t;
}

#ifdef HB_CP_STAMPS
// diagnostic build: stamps stay in registers (a stamp that stores to memory costs ~250 cycles and distorts what it
// measures) and leave through a buffer of their own at the end of the kernel
#define CP_NSTAMP 16
#define CP_STAMP(slot) cpst[slot] = __builtin_amdgcn_s_memtime()
#else
#define CP_STAMP(slot)
#endif

__device__ __forceinline__ void cp_swap32(float& a, float& b) {   // a = [a.lo32, b.lo32], b = [a.hi32, b.hi32]
  unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  a = __builtin_bit_cast(float, x);
  b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ void cp_swap16(float& a, float& b) {   // rows of 16 lanes: a = [a0, b0, a2, b2], b = [a1, b1, a3, b3]
  unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  a = __builtin_bit_cast(float, x);
  b = __builtin_bit_cast(float, y);
}

// bounded wait for a flag word of another workgroup (one storing wave per word; see the header)
struct CpWait {
  unsigned* tmo;               // the launch's timeout word
  unsigned long long deadline;
  bool dead;
  __device__ __forceinline__ void wait(const unsigned* f, unsigned atleast = 1u) {
    if (dead) return;
    unsigned spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < atleast) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 127u) == 0u) {
        if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
            __builtin_amdgcn_s_memrealtime() > deadline) {
          __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dead = true;
          return;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the payload loads below the poll
  }
};

// LDS of one factorisation workgroup (the kernels that host the body declare the bytes and hand them in, so that a
// launch with other roles -- csrc/sgp.hip -- can overlay them with its own)
struct CpLds {
  float colbuf[2][CP_NB][CP_LD];   // [diag | strip][column][row]
  float pibuf[CP_NB];              // reciprocal pivots
  int done[2];                     // columns of the diagonal block (0) / of the strip (1) that are published
  unsigned s_last;
};

// arrival of one workgroup (any role); the LAST one writes info[] and leaves the sync words zero for the next call
__device__ __forceinline__ void cp_arrive(const CpArgs& a, unsigned& s_last) {
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(&a.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == (unsigned)a.arrive - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (s_last) {
    const unsigned tmo = __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int bb = tid; bb < a.B; bb += 512) {
      const unsigned f = __hip_atomic_load(&a.sync[CP_HDR + bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.info[bb] = tmo ? -1 : (f ? (int)(CP_FAILBIG - f) : 0);
    }
    __syncthreads();
    const long nw = cp_sync_words_nb(a.B, a.nb);
    for (long t = tid; t < nw; t += 512) __hip_atomic_store(&a.sync[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One factorisation workgroup; `ticket` = its identity in start order (the hosting kernel draws it from sync[0]).
__device__ __forceinline__ void chol_persist_body(const CpArgs& a, CpLds& sh, const unsigned ticket) {
  float (&colbuf)[2][CP_NB][CP_LD] = sh.colbuf;
  float (&pibuf)[CP_NB] = sh.pibuf;
  int (&done)[2] = sh.done;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = w >> 2;                                   // 0: diagonal block, 1: strip
  // column group.  Wave w runs on SIMD w & 3; the work of a wave grows with q (it follows 16 q columns before its own
  // 16), so the diagonal wave of group q shares its SIMD with the strip wave of group 3 - q: 5 units of 16 columns each
  // (a.early & 32, diagnostic: the pairing (q, q - 1 mod 4), in which the strip wave next to the pivot wave of chunk q >= 1 has
  // finished its own group before that chunk starts -- measured 1.3 us per cfg-2 step SLOWER, 197.5 against 196.1: the balance wins)
  const int q = r == 0 ? (w & 3) : ((a.early & 32) ? ((w & 3) + 3) & 3 : 3 - (w & 3));
  const int i16 = lane & 15, g4 = lane >> 4;              // MFMA 16x16x4 lane coordinates
  const int M = a.M, nb = a.nb, inv = a.W != nullptr;

  if (tid == 0) done[0] = 0, done[1] = 0;
  __syncthreads();
  // ticket -> (column block j, matrix b, strip s), column block major
  int j = 0, b, s;
  {
    unsigned t = ticket;
    for (;; ++j) {
      const unsigned per = (unsigned)a.B * (unsigned)((nb - 1 - j) + (inv ? j + 1 : 0) > 0 ? (nb - 1 - j) + (inv ? j + 1 : 0) : 1);
      if (t < per || j == nb - 1) break;
      t -= per;
    }
    const int ns = (nb - 1 - j) + (inv ? j + 1 : 0) > 0 ? (nb - 1 - j) + (inv ? j + 1 : 0) : 1;
    b = (int)(t / (unsigned)ns);
    s = (int)(t % (unsigned)ns);
  }
  const int nA = nb - 1 - j;                 // A strips of this column block
  const bool stripA = s < nA;
  const bool stripY = !stripA && inv;
  const int irow = stripA ? j + 1 + s : s - nA;   // row block of the strip (A row block, or identity row block)
  const bool strip_live = stripA || stripY;
  const bool ydiag = stripY && irow == j;    // rows of the identity that start in this column block

  const size_t mm = (size_t)M * M;
  const float* Ab = a.A ? a.A + (size_t)b * mm : nullptr;
  float* Lb = a.L + (size_t)b * mm;
  float* Wb = inv ? a.W + (size_t)b * mm : nullptr;
  unsigned* fail = a.sync + CP_HDR + b;
  unsigned* flags = a.sync + CP_HDR + ((a.B + 3) / 4) * 4 + (size_t)b * nb * nb * 4;   // [panel k][strip s'][chunk]
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(a.X + (size_t)b * mm, 0, (int)(mm * sizeof(float)), 0x00020000);
  CpWait wt = {a.sync + 2, __builtin_amdgcn_s_memrealtime() + CP_TIMEOUT_TICKS, false};
#ifdef HB_CP_STAMPS
  unsigned long long cpst[CP_NSTAMP];
#pragma unroll
  for (int i = 0; i < CP_NSTAMP; ++i) cpst[i] = 0;
#endif
  CP_STAMP(0);

  // ---- accumulators: tile t = rows 16t .. 16t+15 of the wave's row block, columns 16q .. 16q+15;
  // lane (i16, g4) register e = element (row 16t + i16, column 16q + 4 g4 + e)
  CpV4 R[4];
  {
    const bool fromA = r == 0 || stripA;
    const int rb = r == 0 ? j : irow;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (fromA && (r == 1 || t >= q)) {   // tiles of the diagonal block strictly above the diagonal are never read
        const int row = CP_NB * rb + 16 * t + i16, col0 = CP_NB * j + CP_G * q + 4 * g4;
        if (a.A) {
          R[t] = *reinterpret_cast<const CpV4*>(Ab + (size_t)row * M + col0);
        } else {
          const float* Xb = a.gX + (size_t)b * a.gsX;
          const float* eb = a.gell + (size_t)b * a.gsEll;
          CpV4 kv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = gram_value<float>(a.gkind, Xb + (size_t)row * a.gd, Xb + (size_t)(col0 + e) * a.gd, eb, a.gdl, a.gd);
            kv[e] = row == col0 + e ? v + a.gdiag : v;
          }
          R[t] = kv;
        }
      } else {
        CpV4 z = {0.f, 0.f, 0.f, 0.f};
        if (r == 1 && ydiag) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (16 * t + i16 == CP_G * q + 4 * g4 + e) z[e] = 1.f;
        }
        R[t] = z;
      }
    }
  }

  // ---- trailing updates by panels 0 .. j-1, 16 columns (one chunk) at a time
  for (int k = 0; k < j; ++k) {
    if (r == 1 && (!strip_live || ydiag || (stripY && k < irow))) continue;   // identity rows are zero left of their 1s
    const int sB = j - (k + 1);                                       // producer of L(j, k) in column block k
    const int sA = r == 0 ? sB : (stripA ? irow - (k + 1) : (nb - 1 - k) + irow);
    const unsigned* fB = flags + ((size_t)k * nb + sB) * 4;
    const unsigned* fA = flags + ((size_t)k * nb + sA) * 4;
    const int offB = (k * nb + sB) * (CP_NB * CP_NB * 4), offA = (k * nb + sA) * (CP_NB * CP_NB * 4);   // bytes
#pragma unroll 1
    for (int c4 = 0; c4 < 4; ++c4) {
      wt.wait(fB + c4);
      if (r == 1) wt.wait(fA + c4);
      if (k == j - 1 && c4 == 3) CP_STAMP(5);
      // chunk layout [chunk][v = k-quad][row][4 floats]: lane (i16, g4) takes k-quad g4 of its rows
      const int chunk = ((c4 * 4 + g4) * CP_NB) * 16;
      const CpV4 av = __builtin_bit_cast(CpV4, __builtin_amdgcn_raw_buffer_load_b128(xr, offB + chunk + (CP_G * q + i16) * 16, 0, 16));
      CpV4 bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (r == 1 || t >= q)
          bv[t] = __builtin_bit_cast(CpV4, __builtin_amdgcn_raw_buffer_load_b128(xr, offA + chunk + (16 * t + i16) * 16, 0, 16));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (r == 1 || t >= q) R[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[e], bv[t][e], R[t], 0, 0, 0);
      if (k == j - 1 && c4 == 3) CP_STAMP(6);
    }
  }
  CP_STAMP(1);

  // ---- in-panel phase.
  // (a) FOLLOW the column groups to the left, still in the MFMA layout, four published columns at a time: one rank-4
  //     MFMA per tile, operands straight from the published columns (A: the multipliers L[cq + m][c], B: the wave's own
  //     rows' L[row][c]) -- 5 LDS reads + 4 MFMAs per four columns.
  const int cq = CP_G * q;
  const bool in_panel = r == 0 || strip_live;
  int seen0 = 0, seen1 = 0;
  auto wait0 = [&](int upto) {   // columns [0, upto) of the diagonal block are published
    while (seen0 < upto) {
      seen0 = __hip_atomic_load(&done[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (seen0 < upto && r == 1) __builtin_amdgcn_s_sleep(1);   // (a diagonal wave is the next pivot: it polls without sleeping)
    }
    asm volatile("" ::: "memory");
  };
  auto wait1 = [&](int upto) {   // ... of the strip (published after the same columns of the diagonal block)
    while (seen1 < upto) {
      seen1 = __hip_atomic_load(&done[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (seen1 < upto) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  if (in_panel) {
    if (r == 0) __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
    for (int c0 = 0; c0 < cq; c0 += 4) {
      if (r == 0) wait0(c0 + 4); else wait1(c0 + 4);   // (a strip column is published after the same column of the diagonal block)
      const float av = colbuf[0][c0 + g4][cq + i16];
      float bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (r == 1 || t >= q) bv[t] = colbuf[r][c0 + g4][16 * t + i16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (r == 1 || t >= q) R[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(-av, bv[t], R[t], 0, 0, 0);
    }
  }
  CP_STAMP(2);

  // (b) row per lane: a 4 x 4 block transpose over (lane group, tile); afterwards lane l holds row l of its row
  // block, x[4 s + e] = column 16 q + 4 s + e
  float x[16];
  {
    float T[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = R[t][e];
        T[t][e] = v;
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cp_swap32(T[0][e], T[2][e]);
      cp_swap32(T[1][e], T[3][e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cp_swap16(T[0][e], T[1][e]);
      cp_swap16(T[2][e], T[3][e]);
    }
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
      for (int e = 0; e < 4; ++e) x[4 * sl + e] = T[sl][e];
  }

  // (c) the wave's own column group.  Measured (tools/xlane_cost.hip, profiles/r04_xlane_cost.txt): a lone wave issues
  // one vector instruction per ~6.6 cycles whatever it is (v_readlane + v_fma pair 16.7 dependent, v_fmac_f32_dpp 11.6),
  // the chain readlane -> rsq -> mul -> readlane -> fma is ~50 cycles per column -- so the INSTRUCTION COUNT of the pivot
  // wave paces the panel, not the latency of its cross-lane operations.
  //   Pivot wave (diagonal block, all 64 rows at once): per column 3 chain instructions, 2 LDS writes, ONE eager update
  //   (the next column, multiplier by v_readlane) -- every later column takes the update one step later, multipliers read
  //   back from the published column as uniform 16-byte LDS reads (LDS executes a wave's instructions in order, so the
  //   read-back sees the wave's own write) and applied two columns per instruction (v_pk_fma_f32).
  //   Strip wave: all reads of a four-column sub-group first (one LDS latency per sub-group), then the arithmetic (packed),
  //   then its four stores and the counter.
  CP_STAMP(11);
  if (in_panel) {
    CpV2 xx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xx[i] = CpV2{x[2 * i], x[2 * i + 1]};
    if (r == 0) {
      __builtin_amdgcn_s_setprio(3);
      CpV4 mb[2][4];
      float xprev = 0.f;
#pragma unroll
      for (int p = 0; p < CP_G; ++p) {
        const int c = cq + p;
        float xp = (p & 1) ? xx[p >> 1][1] : xx[p >> 1][0];
        const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xp), c));
        float lcc, pi;
        pivot_sqrt(d, lcc, pi);
        (void)lcc;
        xp *= pi;
        colbuf[0][c][lane] = xp;
        pibuf[c] = pi;
        if ((p & 3) == 3) {
          asm volatile("" ::: "memory");
          __hip_atomic_store(&done[0], c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
          CP_STAMP(7 + (p >> 2));
        }
        if (p + 1 < CP_G) {
          // eager: the next column (it is the next pivot)
          const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xp), c + 1));
          float xn = ((p + 1) & 1) ? xx[(p + 1) >> 1][1] : xx[(p + 1) >> 1][0];
          xn = __builtin_fmaf(-xp, m1, xn);
          if ((p + 1) & 1) xx[(p + 1) >> 1][1] = xn; else xx[(p + 1) >> 1][0] = xn;
        }
        // multipliers of this column for the columns p + 2 .. 15, read back for the next step
#pragma unroll
        for (int v = (p + 2) >> 2; v < 4; ++v) mb[p & 1][v] = *reinterpret_cast<const CpV4*>(&colbuf[0][c][cq + 4 * v]);
        // lazy: column p - 1 applied to the columns p + 1 .. 15
        if (p >= 1) {
#pragma unroll
          for (int i = (p + 1) >> 1; i < 8; ++i) {
            const CpV4 mv = mb[(p - 1) & 1][i >> 1];
            const CpV2 m2 = (i & 1) ? CpV2{mv[2], mv[3]} : CpV2{mv[0], mv[1]};
            if (2 * i >= p + 1) {
              xx[i] = __builtin_elementwise_fma(CpV2{-xprev, -xprev}, m2, xx[i]);
            } else {   // 2 i == p: only the odd element (column p + 1)
              const float t1 = xx[i][1];
              xx[i][1] = __builtin_fmaf(-xprev, m2[1], t1);
            }
          }
        }
        if (p & 1) xx[p >> 1][1] = xp; else xx[p >> 1][0] = xp;
        xprev = xp;
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        wait0(cq + 4 * sg + 4);
        const CpV4 pi4 = *reinterpret_cast<const CpV4*>(&pibuf[cq + 4 * sg]);
        CpV4 mv[4][4];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
          for (int v = sg; v < 4; ++v) mv[pp][v] = *reinterpret_cast<const CpV4*>(&colbuf[0][cq + 4 * sg + pp][cq + 4 * v]);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const int p = 4 * sg + pp;
          float xp = (p & 1) ? xx[p >> 1][1] : xx[p >> 1][0];
          xp *= pi4[pp];
          if (p & 1) xx[p >> 1][1] = xp; else xx[p >> 1][0] = xp;
#pragma unroll
          for (int i = (p + 1) >> 1; i < 8; ++i) {
            const CpV4 m4 = mv[pp][i >> 1];
            const CpV2 m2 = (i & 1) ? CpV2{m4[2], m4[3]} : CpV2{m4[0], m4[1]};
            if (2 * i >= p + 1) {
              xx[i] = __builtin_elementwise_fma(CpV2{-xp, -xp}, m2, xx[i]);
            } else {
              const float t1 = xx[i][1];
              xx[i][1] = __builtin_fmaf(-xp, m2[1], t1);
            }
          }
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const int p = 4 * sg + pp;
          colbuf[1][cq + p][lane] = (p & 1) ? xx[p >> 1][1] : xx[p >> 1][0];
        }
        asm volatile("" ::: "memory");
        __hip_atomic_store(&done[1], cq + 4 * sg + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        // the four finished columns ARE k-quad sg of the chunk: on their way to the later column blocks at once (only the
        // last quad's store is still in flight when the chunk's flag is due)
        if (j < nb - 1 && !(a.early & 16)) {
          const CpV4 o = {xx[2 * sg][0], xx[2 * sg][1], xx[2 * sg + 1][0], xx[2 * sg + 1][1]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(CpU4, o), xr,
                                                 (j * nb + s) * (CP_NB * CP_NB * 4) + ((q * 4 + sg) * CP_NB + lane) * 16, 0, 16);
        }
        CP_STAMP(7 + sg);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[2 * i] = xx[i][0], x[2 * i + 1] = xx[i][1];
    if (r == 1) {
      // ---- hand the finished chunk to the later column blocks (write-through stores, drained, then its flag)
      if (j < nb - 1) {
        if (a.early & 16) {   // (diagnostic: the chunk's four stores at the end, as before)
          const int off = (j * nb + s) * (CP_NB * CP_NB * 4) + ((q * 4) * CP_NB + lane) * 16;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const CpV4 o = {x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(CpU4, o), xr, off + v * CP_NB * 16, 0, 16);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CP_STAMP(12);
        __hip_atomic_store(flags + ((size_t)j * nb + s) * 4 + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  CP_STAMP(3);
  __syncthreads();

  // ---- results, from the published columns in LDS (colbuf[.][column][row]).  Every 64 x 64 block of L, W and of the
  // fragment-major images (include/henbun_hip.h: hb_cholesky_inverse, Wfrag) has exactly one writer, zero blocks of the
  // strict upper triangles included -- there is no finishing pass.
  //   A strip (i, j), i > j:  L(i, j);  zeros: W(j, i), its W image block, the W^T image block (i, j).
  //   identity strip (i', j), i' <= j:  W(j, i') = Y(i', j)^T and both image blocks;  zeros: L(i', j) when i' < j.
  const int nT = M / 32;
  float* Wf = a.Wf ? a.Wf + (size_t)b * mm : nullptr;                 // image of W;  + B*M*M: image of W^T
  const size_t tot = (size_t)a.B * mm;
  __bf16* W3 = (a.Wf && a.bf16x3) ? reinterpret_cast<__bf16*>(a.Wf + 2 * tot) + (size_t)b * mm : nullptr;   // planes: + p * tot
  const CpV4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // one 16-byte group of a fragment-major image block: tile row tt (32 rows), k chunk QQ (32 columns), group index g =
  // (v, lane): element s of the group is X[32 tt + li][32 QQ + 16 h + 4 v + s]
  auto frag_off = [&](int tt, int QQ, int g) -> size_t { return (((size_t)tt * nT + QQ) * 256 + g) * 4; };
  auto store_bf3 = [&](__bf16* plane0, size_t off8, const float (&x8)[8]) {
    // bf16x3 planes, [t][Q][q][64 lanes][8]: eight consecutive k of one lane, split into hi + mid + lo
    typedef __bf16 B8 __attribute__((ext_vector_type(8)));
    B8 h8, m8, l8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 h_, m_, l_;
      hb_split_bf16x3(x8[e], h_, m_, l_);
      h8[e] = h_, m8[e] = m_, l8[e] = l_;
    }
    *reinterpret_cast<B8*>(plane0 + off8) = h8;
    *reinterpret_cast<B8*>(plane0 + tot + off8) = m8;
    *reinterpret_cast<B8*>(plane0 + 2 * tot + off8) = l8;
  };
  if (stripA) {
    for (int idx = tid; idx < CP_NB * (CP_NB / 4); idx += 512) {
      const int rr = idx >> 4, c4 = (idx & 15) * 4;
      const CpV4 v = {colbuf[1][c4][rr], colbuf[1][c4 + 1][rr], colbuf[1][c4 + 2][rr], colbuf[1][c4 + 3][rr]};
      *reinterpret_cast<CpV4*>(Lb + (size_t)(CP_NB * irow + rr) * M + CP_NB * j + c4) = v;
      if (inv) *reinterpret_cast<CpV4*>(Wb + (size_t)(CP_NB * j + rr) * M + CP_NB * irow + c4) = zero4;
    }
    if (Wf) {
      for (int idx = tid; idx < 4 * 256; idx += 512) {
        const int sub = idx >> 8, g = idx & 255;
        *reinterpret_cast<CpV4*>(Wf + frag_off(2 * j + (sub >> 1), 2 * irow + (sub & 1), g)) = zero4;          // W(j, i)
        *reinterpret_cast<CpV4*>(Wf + tot + frag_off(2 * irow + (sub >> 1), 2 * j + (sub & 1), g)) = zero4;    // W^T(i, j)
      }
      if (W3) {
        const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int idx = tid; idx < 4 * 128; idx += 512) {
          const int sub = idx >> 7, g = idx & 127;
          store_bf3(W3, (((size_t)(2 * j + (sub >> 1)) * nT + 2 * irow + (sub & 1)) * 128 + g) * 8, z8);
          store_bf3(W3 + 3 * tot, (((size_t)(2 * irow + (sub >> 1)) * nT + 2 * j + (sub & 1)) * 128 + g) * 8, z8);
        }
      }
    }
  } else if (stripY) {
    if (Wf && (a.early & 1)) {
      // Early-start consumers (csrc/sgp.hip) read row block j of the W image as soon as it is final: written first,
      // write-through, drained, then this workgroup's contribution to the row block's counter.
      const __amdgpu_buffer_rsrc_t wfr = __builtin_amdgcn_make_buffer_rsrc(Wf, 0, (int)(mm * sizeof(float)), 0x00020000);
      for (int idx = tid; idx < 4 * 256; idx += 512) {
        const int sub = idx >> 8, g = idx & 255, v4 = g >> 6, l6 = g & 63, li = l6 & 31, h = l6 >> 5;
        const int tsub = sub >> 1, qsub = sub & 1;
        const int c = 32 * tsub + li, r4 = 32 * qsub + 16 * h + 4 * v4;
        CpV4 v = *reinterpret_cast<const CpV4*>(&colbuf[1][c][r4]);
        if (ydiag) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (r4 + e > c) v[e] = 0.f;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(CpU4, v), wfr, (int)(frag_off(2 * j + tsub, 2 * irow + qsub, g) * 4), 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_fetch_add(a.sync + cp_wready_off(a.B, nb) + (size_t)b * nb + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // W(j, irow) = Y(irow, j)^T: row c of the block is the published column c
    for (int idx = tid; idx < CP_NB * (CP_NB / 4); idx += 512) {
      const int c = idx >> 4, r4 = (idx & 15) * 4;
      CpV4 v = *reinterpret_cast<const CpV4*>(&colbuf[1][c][r4]);
      if (ydiag) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (r4 + e > c) v[e] = 0.f;
      }
      *reinterpret_cast<CpV4*>(Wb + (size_t)(CP_NB * j + c) * M + CP_NB * irow + r4) = v;
      if (!ydiag) *reinterpret_cast<CpV4*>(Lb + (size_t)(CP_NB * irow + c) * M + CP_NB * j + r4) = zero4;   // L(i', j), i' < j
    }
    if (Wf) {
      for (int idx = tid; idx < 4 * 256; idx += 512) {
        const int sub = idx >> 8, g = idx & 255, v4 = g >> 6, l6 = g & 63, li = l6 & 31, h = l6 >> 5;
        const int tsub = sub >> 1, qsub = sub & 1;
        if (!(a.early & 1)) {
          // W image: row 32 (2j + tsub) + li = column 32 tsub + li of the block; k = 64 i' + 32 qsub + 16 h + 4 v4 + s
          const int c = 32 * tsub + li, r4 = 32 * qsub + 16 * h + 4 * v4;
          CpV4 v = *reinterpret_cast<const CpV4*>(&colbuf[1][c][r4]);
          if (ydiag) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (r4 + e > c) v[e] = 0.f;
          }
          *reinterpret_cast<CpV4*>(Wf + frag_off(2 * j + tsub, 2 * irow + qsub, g)) = v;
        }
        {
          // W^T image: row 32 (2 i' + tsub) + li = row 32 tsub + li of the strip; k = 64 j + 32 qsub + 16 h + 4 v4 + s
          const int rr = 32 * tsub + li, c4 = 32 * qsub + 16 * h + 4 * v4;
          CpV4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (ydiag && rr > c4 + e) ? 0.f : colbuf[1][c4 + e][rr];
          *reinterpret_cast<CpV4*>(Wf + tot + frag_off(2 * irow + tsub, 2 * j + qsub, g)) = v;
        }
      }
      if (W3) {
        for (int idx = tid; idx < 4 * 128; idx += 512) {
          const int sub = idx >> 7, g = idx & 127, q2 = g >> 6, l6 = g & 63, li = l6 & 31, h = l6 >> 5;
          const int tsub = sub >> 1, qsub = sub & 1;
          float x8[8];
          {
            const int c = 32 * tsub + li, r8 = 32 * qsub + 16 * q2 + 8 * h;
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[e] = (ydiag && r8 + e > c) ? 0.f : colbuf[1][c][r8 + e];
            store_bf3(W3, (((size_t)(2 * j + tsub) * nT + 2 * irow + qsub) * 128 + g) * 8, x8);
          }
          {
            const int rr = 32 * tsub + li, c8 = 32 * qsub + 16 * q2 + 8 * h;
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[e] = (ydiag && rr > c8 + e) ? 0.f : colbuf[1][c8 + e][rr];
            store_bf3(W3 + 3 * tot, (((size_t)(2 * irow + tsub) * nT + 2 * j + qsub) * 128 + g) * 8, x8);
          }
        }
      }
    }
  }
  if (s == 0) {
    // L(j, j), lower triangle; and the first failed pivot of this column block (a pivot <= 0 or NaN leaves a NaN on
    // the diagonal at its own column, and only NaNs after it)
    for (int idx = tid; idx < CP_NB * (CP_NB / 4); idx += 512) {
      const int rr = idx >> 4, c4 = (idx & 15) * 4;
      CpV4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = c4 + e <= rr ? colbuf[0][c4 + e][rr] : 0.f;
      *reinterpret_cast<CpV4*>(Lb + (size_t)(CP_NB * j + rr) * M + CP_NB * j + c4) = v;
    }
    if (w == 0) {
      const float dg = colbuf[0][lane][lane];
      const unsigned long long bad = __ballot(!(dg == dg));
      if (bad != 0ull && lane == 0) {
        const unsigned col = (unsigned)(CP_NB * j + __builtin_ctzll(bad) + 1);   // LAPACK's info
        __hip_atomic_fetch_max(fail, CP_FAILBIG - col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  CP_STAMP(4);
#ifdef HB_CP_STAMPS
  if (a.stamps && lane == 0) {
#pragma unroll
    for (int i = 0; i < CP_NSTAMP; ++i) a.stamps[((size_t)ticket * 8 + w) * CP_NSTAMP + i] = cpst[i];
    a.stamps[((size_t)a.total * 8 + (size_t)ticket * 8 + w)] = __builtin_amdgcn_s_memrealtime();   // (a clock all XCDs share)
  }
#endif

  // ---- arrival; the last workgroup writes info[] and leaves the sync words zero for the next call
  cp_arrive(a, sh.s_last);
}
