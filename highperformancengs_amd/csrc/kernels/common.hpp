// common.hpp -- device helpers shared by the gfx950 scan kernels.
#pragma once
#include "../host/knobs.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpngs.h"

namespace hpn {

constexpr int kWave = 64;  // CDNA wavefront

typedef unsigned long long u64;
typedef uint32_t u32 __attribute__((ext_vector_type(4)));  // 16-byte vector = one dwordx4 per lane

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// Sum / OR over the 64 lanes of a wave; result valid in every lane (it comes back through an SGPR).  DPP moves: shifts by
// 1, 2, 4, 8 inside rows of 16 lanes, lane 15 of a row into the next row (rows 1 and 3), lane 31 into rows 2 and 3; lane 63
// then holds the whole wave.  Full-rate VALU instructions; the __shfl_xor form is an LDS round trip (ds_bpermute) per step.
#define HPN_DPP_STEP(x, op, ctrl, rows) x op (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, false)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    HPN_DPP_STEP(v, +=, 0x111, 0xf);
    HPN_DPP_STEP(v, +=, 0x112, 0xf);
    HPN_DPP_STEP(v, +=, 0x114, 0xf);
    HPN_DPP_STEP(v, +=, 0x118, 0xf);
    HPN_DPP_STEP(v, +=, 0x142, 0xa);
    HPN_DPP_STEP(v, +=, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, kWave - 1);
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    HPN_DPP_STEP(v, |=, 0x111, 0xf);
    HPN_DPP_STEP(v, |=, 0x112, 0xf);
    HPN_DPP_STEP(v, |=, 0x114, 0xf);
    HPN_DPP_STEP(v, |=, 0x118, 0xf);
    HPN_DPP_STEP(v, |=, 0x142, 0xa);
    HPN_DPP_STEP(v, |=, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, kWave - 1);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)      // (bound_ctrl off: a lane without a source keeps its own value)
{
#define HPN_DPP_MAXSTEP(ctrl, rows)                                                                     \
    {                                                                                                   \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false); \
        v = t > v ? t : v;                                                                              \
    }
    HPN_DPP_MAXSTEP(0x111, 0xf)
    HPN_DPP_MAXSTEP(0x112, 0xf)
    HPN_DPP_MAXSTEP(0x114, 0xf)
    HPN_DPP_MAXSTEP(0x118, 0xf)
    HPN_DPP_MAXSTEP(0x142, 0xa)
    HPN_DPP_MAXSTEP(0x143, 0xc)
#undef HPN_DPP_MAXSTEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, kWave - 1);
}
#undef HPN_DPP_STEP

// n / W for n < 2^32 with M = 0xffffffff / W (made once): the estimate is short by at most one.  A 32-bit division is
// ~30 vector instructions.
__device__ __forceinline__ uint32_t div_by(uint32_t n, uint32_t W, uint32_t M)
{
    const uint32_t q = __umulhi(n, M);
    return q + (n - q * W >= W ? 1u : 0u);
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
