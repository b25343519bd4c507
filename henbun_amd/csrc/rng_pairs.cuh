// Load/store of one lane's xoroshiro state (structure-of-arrays layout).
#pragma once
#include "common.cuh"

__device__ __forceinline__ HbRng rng_load(const uint64_t* state, long nlanes, long t) {
  HbRng g;
  g.s0 = state[t];
  g.s1 = state[nlanes + t];
  return g;
}
__device__ __forceinline__ void rng_store(uint64_t* state, long nlanes, long t, const HbRng& g) {
  state[t] = g.s0;
  state[nlanes + t] = g.s1;
}
