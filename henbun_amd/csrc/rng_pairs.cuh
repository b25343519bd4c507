// Load/store of one lane's xoroshiro state: see rng_core.cuh (kept as the include the kernel files use).
#pragma once
#include "common.cuh"
