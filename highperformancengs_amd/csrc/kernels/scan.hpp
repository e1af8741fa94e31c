// scan.hpp -- single-pass device-wide prefix sums for gfx950 (decoupled look-back).
//
// One workgroup = one tile, taken in ticket order.  A tile publishes one 8-byte
// granule {flag:2, value:62} per tile: first its own aggregate, later its
// inclusive prefix.  Flag and payload share the granule, so one agent-scope
// relaxed store / load is a complete hand-off (no separate flag, no release /
// acquire fence needed: MI355X_MICROARCH.md "R2 granule").  The per-XCD L2s are
// not coherent, hence the agent-scope (sc1) accesses.  Spins are bounded: a
// hand-off that never arrives sets *err instead of hanging the GPU.
#pragma once
#include "common.hpp"

namespace hpn {

constexpr u64 kScanInvalid = 0, kScanAggregate = 1, kScanPrefix = 2;
constexpr u64 kScanValueMask = (1ull << 62) - 1;
constexpr uint32_t kScanSpinLimit = 1u << 24;

__device__ __forceinline__ void scan_publish(u64 *status, uint64_t tile, u64 flag, u64 value)
{
    __hip_atomic_store(&status[tile], (flag << 62) | (value & kScanValueMask), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL lanes of wave 0 of the workgroup. Returns the exclusive prefix of
// `tile` (sum of the aggregates of tiles 0..tile-1) in every lane, and publishes
// this tile's inclusive prefix.
__device__ __forceinline__ u64 scan_lookback(u64 *status, uint64_t tile, u64 aggregate, uint32_t *err)
{
    if (tile == 0) {
        if (lane_id() == 0) scan_publish(status, 0, kScanPrefix, aggregate);
        return 0;
    }
    if (lane_id() == 0) scan_publish(status, tile, kScanAggregate, aggregate);
    u64 exclusive = 0;
    int64_t idx = (int64_t)tile - 1;  // lane L inspects tile idx-L
    for (;;) {
        const int64_t j = idx - lane_id();
        u64 w = (kScanPrefix << 62);  // tiles before 0: prefix 0
        uint32_t spins = 0;
        for (;;) {
            if (j >= 0) w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot((w >> 62) == kScanInvalid) == 0) break;
            if (++spins > kScanSpinLimit) {
                if (lane_id() == 0) atomicOr(err, 1u);
                return exclusive;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const u64 has_prefix = __ballot((w >> 62) == kScanPrefix);
        const int stop = has_prefix ? __builtin_ctzll(has_prefix) : kWave;  // nearest tile with a full prefix
        u64 v = lane_id() <= stop ? (w & kScanValueMask) : 0;
        // 64-bit wave sum
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
        exclusive += v;
        if (has_prefix) break;
        idx -= kWave;
    }
    if (lane_id() == 0) scan_publish(status, tile, kScanPrefix, exclusive + aggregate);
    return exclusive;
}

// Exclusive scan of one value per lane across the wave (returns exclusive, sets total).
__device__ __forceinline__ u64 wave_excl_scan(u64 v, u64 &total)
{
    u64 inc = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const u64 t = __shfl_up(inc, o, kWave);
        if (lane_id() >= o) inc += t;
    }
    total = __shfl(inc, kWave - 1, kWave);
    return inc - v;
}

}  // namespace hpn
