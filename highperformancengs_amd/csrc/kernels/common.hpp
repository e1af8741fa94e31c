// common.hpp -- device helpers shared by the gfx950 scan kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpngs.h"

namespace hpn {

constexpr int kWave = 64;  // CDNA wavefront

typedef unsigned long long u64;
typedef uint32_t u32 __attribute__((ext_vector_type(4)));  // 16-byte vector = one dwordx4 per lane

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// Sum over the 64 lanes of a wave; result valid in every lane.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, kWave);
    return v;
}

// Streaming 16-byte load: the data is read once, keep it out of the way of
// anything worth caching.
__device__ __forceinline__ u32 load_stream16(const u32 *p) { return __builtin_nontemporal_load(p); }

// splitmix64 finaliser: the counter-based generator of SURVEY.md §8d.
// (The CPU checker restates the same published function in oracle/.)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
constexpr uint64_t kGold = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kStep = 0xD1B54A32D192ED03ull;

}  // namespace hpn
