// tally_util.hpp -- helpers shared by the fastq tally kernels (fastq_scan.hip, fastq_tally.hip).
#pragma once
#include "common.hpp"

namespace hpn {

// Bytes are < 128 inside the domain, so x+75 sets bit 7 exactly when x >= 53
// and x+65 exactly when x >= 63, with no carry between bytes (statQ's thresholds,
// fastq_count.c:124).  `hi` collects bit 7 of every input byte: set = domain violation.
__device__ __forceinline__ void swar16(u32 v, uint32_t &c20, uint32_t &c30, uint32_t &hi)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = v[k];
        hi |= w;
        c20 += __builtin_popcount((w + 0x4b4b4b4bu) & 0x80808080u);
        c30 += __builtin_popcount((w + 0x41414141u) & 0x80808080u);
    }
}

// Keep bytes [a, b) of a 16-byte vector, zero the rest (a zero byte counts nowhere).
__device__ __forceinline__ u32 mask_bytes(u32 v, int a, int b)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lo = min(max(a - 4 * k, 0), 4), hi = min(max(b - 4 * k, 0), 4);
        const uint32_t mh = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
        const uint32_t ml = lo >= 4 ? 0xffffffffu : ((1u << (8 * lo)) - 1u);
        v[k] &= mh & ~ml;
    }
    return v;
}

// Add one length per active lane to an LDS histogram.  Reads of one run have one
// length almost always: then a single lane adds the whole wave's count instead of
// 64 lanes serialising on one LDS address.
__device__ __forceinline__ void hist_len(uint32_t *s_hist, bool valid, uint32_t len)
{
    const u64 act = __ballot(valid);
    if (act == 0) return;
    const int leader = __builtin_ctzll(act);
    const uint32_t first = __shfl(len, leader, kWave);
    const u64 same = __ballot(valid && len == first);
    if (same == act) {
        if (lane_id() == leader) atomicAdd(&s_hist[first], (uint32_t)__builtin_popcountll(act));
    } else if (valid) {
        atomicAdd(&s_hist[len], 1u);
    }
}

}  // namespace hpn
