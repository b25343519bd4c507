// Microbenchmark: cycles per instruction of the cross-lane forms a factorisation chain can be built from, one wave on
// one SIMD (s_memtime around an unrolled block, 16 lanes active like the diagonal-tile phase or all 64).
// hipcc -O3 --offload-arch=gfx950 tools/xlane_cost.hip -o tools/_bin/xlane_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__global__ void k(float* p, unsigned long long* out, int mask16) {
  __shared__ float lds[1024];
  const int lane = threadIdx.x;
  float x0 = p[lane], x1 = p[lane + 64], x2 = p[lane + 128], x3 = p[lane + 192], nx = -x0;
  lds[lane] = x0;
  __syncthreads();
  if (mask16 && lane >= 16) return;
  unsigned long long t0, t1;
  // (a) dependent chain of v_fmac_f32_dpp row_newbcast
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; ++i) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x1) : "v"(nx));
  t1 = now();
  if (lane == 0) out[0] = t1 - t0;
  // (b) independent v_fmac_f32_dpp (4 accumulators, same source)
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x1) : "v"(x0), "v"(nx));
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x2) : "v"(x0), "v"(nx));
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x3) : "v"(x0), "v"(nx));
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x1) : "v"(x0), "v"(nx));
  }
  t1 = now();
  if (lane == 0) out[1] = t1 - t0;
  // (c) dependent plain v_fma chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x1) : "v"(nx));
  t1 = now();
  if (lane == 0) out[2] = t1 - t0;
  // (d) independent plain v_fma
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(x0), "v"(nx));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(x0), "v"(nx));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(x0), "v"(nx));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(x0), "v"(nx));
  }
  t1 = now();
  if (lane == 0) out[3] = t1 - t0;
  // (e) v_readlane -> v_fma with the SGPR, dependent chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    float s;
    asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(x1));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "s"(s), "v"(nx));
  }
  t1 = now();
  if (lane == 0) out[4] = t1 - t0;
  // (f) v_readlane -> v_rsq(SGPR) -> v_mul chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    float s, r;
    asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(x2));
    asm volatile("v_rsq_f32 %0, %1\n\ts_nop 0" : "=v"(r) : "s"(s));
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(r));
  }
  t1 = now();
  if (lane == 0) out[5] = t1 - t0;
  // (g) v_rsq_dpp -> v_mul chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    float r;
    asm volatile("s_nop 1\n\tv_rsq_f32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0" : "=v"(r) : "v"(x3));
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x3) : "v"(r));
  }
  t1 = now();
  if (lane == 0) out[6] = t1 - t0;
  // (h) ds_write_b32 stream (no waits)
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; ++i) lds[(i * 64 + lane) & 1023] = x1;
  t1 = now();
  if (lane == 0) out[7] = t1 - t0;
  // (i) uniform ds_read_b128 -> 4 fma, dependent through the address? no: independent reads, wait once
  t0 = now();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    const float4 m = *reinterpret_cast<const float4*>(&lds[16 * i]);
    acc = __builtin_fmaf(m.x, x0, acc); acc = __builtin_fmaf(m.y, x0, acc); acc = __builtin_fmaf(m.z, x0, acc); acc = __builtin_fmaf(m.w, x0, acc);
  }
  t1 = now();
  if (lane == 0) out[8] = t1 - t0;
  // (j) write -> read back (uniform b128) -> fma: one LDS round trip per iteration
  t0 = now();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lds[lane] = x1;
    const float4 m = *reinterpret_cast<const float4*>(&lds[4]);
    x1 = __builtin_fmaf(m.x, nx, x1);
  }
  t1 = now();
  if (lane == 0) out[9] = t1 - t0;
  // (k) ds_bpermute broadcast -> fma chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(12, __builtin_bit_cast(int, x2)));
    x2 = __builtin_fmaf(b, nx, x2);
  }
  t1 = now();
  if (lane == 0) out[10] = t1 - t0;
  p[lane] = x0 + x1 + x2 + x3 + acc;
}
int main() {
  float* p; unsigned long long* out;
  (void)hipMalloc(&p, 4096); (void)hipMalloc(&out, 128);
  (void)hipMemset(p, 0, 4096);
  const char* names[] = {"dependent v_fmac_dpp row_newbcast (+s_nop 1)", "independent v_fmac_dpp", "dependent v_fma", "independent v_fma",
                         "v_readlane -> v_fma(SGPR) dependent pair", "v_readlane -> v_rsq(SGPR) -> v_mul", "v_rsq_dpp -> v_mul",
                         "ds_write_b32 stream", "uniform ds_read_b128 + 4 fma (per fma)", "ds_write -> uniform read back -> fma (per round trip, 16 iters)", "ds_bpermute -> fma (per iter, 16 iters)"};
  const int reps[] = {REP, REP, REP, REP, REP, REP, REP, REP, REP, 16, 16};
  for (int m16 = 0; m16 < 2; ++m16) {
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, out, m16); (void)hipDeviceSynchronize(); }
    unsigned long long h[16]; (void)hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
    printf("%s lanes active (cycles per iteration, stamp overhead ~40 per block not removed):\n", m16 ? "16" : "64");
    for (int i = 0; i < 11; ++i) printf("  %-66s %7.1f\n", names[i], (double)h[i] / reps[i]);
  }
  return 0;
}
