// synth.hip -- synthetic FASTQ batches generated directly in HBM.
//
// Counter-based: byte k of record r of stream `seed` depends only on
// (seed, r, k), so a shard of the 1e9-read workload can be produced on any GPU
// without the others, and any sub-range can be regenerated on the CPU for
// checking.  Definition (all arithmetic mod 2^64, mix64 = splitmix64 finaliser):
//   key(r)   = mix64(seed + r * 0x9E3779B97F4A7C15)
//   wq(r,j)  = mix64(key + (2j+1) * 0xD1B54A32D192ED03)   quality word of bytes 4j..4j+3
//   wb(r,j)  = mix64(key + (2j+2) * 0xD1B54A32D192ED03)   base word
//   u        = 16-bit field (k & 3) of the word
//   qual     = 35 + (u*40 >> 16)          Phred 2..41 (+33), uniform
//   x        = u*10000 >> 16;  base = x < 100 ? 'N' : "ACGT"[(x-100)/2475]
#include "common.hpp"

namespace hpn {

constexpr int kSynthThreads = 256;

template <bool kBase>
__device__ __forceinline__ uint32_t synth_byte(uint64_t key, uint32_t k)
{
    const uint64_t j = k >> 2;
    const uint64_t w = mix64(key + (2 * j + (kBase ? 2 : 1)) * kStep);
    const uint32_t u = (uint32_t)(w >> (16 * (k & 3))) & 0xffffu;
    if (!kBase) return 35u + ((u * 40u) >> 16);
    const uint32_t x = (u * 10000u) >> 16;
    const uint32_t c = x < 100 ? 4u : (x - 100u) / 2475u;
    return (0x4e54474341ull >> (8 * c)) & 0xffu;  // 'A','C','G','T','N'
}

template <bool kBase>
__global__ __launch_bounds__(kSynthThreads) void k_synth_bytes(uint64_t seed, uint64_t first, uint64_t total,
                                                              uint32_t len, uint8_t *__restrict__ out)
{
    const uint64_t nvec = (total + 15) >> 4;
    for (uint64_t v = (uint64_t)blockIdx.x * kSynthThreads + threadIdx.x; v < nvec;
         v += (uint64_t)gridDim.x * kSynthThreads) {
        const uint64_t g = v << 4;
        uint64_t r = g / len;
        uint32_t k = (uint32_t)(g - r * len);
        uint64_t key = mix64(seed + (first + r) * kGold);
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            w[i >> 2] |= synth_byte<kBase>(key, k) << (8 * (i & 3));
            if (++k == len) {
                k = 0;
                ++r;
                key = mix64(seed + (first + r) * kGold);
            }
        }
        if (g + 16 <= total) {
            *reinterpret_cast<u32 *>(out + g) = u32{w[0], w[1], w[2], w[3]};
        } else {
            for (uint64_t i = 0; g + i < total; ++i) out[g + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        }
    }
}

__global__ __launch_bounds__(kSynthThreads) void k_synth_off(uint64_t n, uint32_t len, uint64_t *__restrict__ off)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kSynthThreads + threadIdx.x; i <= n;
         i += (uint64_t)gridDim.x * kSynthThreads)
        off[i] = i * len;
}

hipError_t launch_synth_fastq(uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint8_t *d_qual,
                              uint8_t *d_base, uint64_t *d_off, int n_cu, hipStream_t st)
{
    const uint64_t total = n * (uint64_t)len;
    const uint64_t nvec = (total + 15) >> 4;
    uint64_t want = nvec / kSynthThreads + 1;
    const unsigned grid = (unsigned)(want < (uint64_t)n_cu * 16 ? want : (uint64_t)n_cu * 16);
    if (total) {
        hipLaunchKernelGGL((k_synth_bytes<false>), dim3(grid), dim3(kSynthThreads), 0, st, seed, first, total, len, d_qual);
        if (d_base)
            hipLaunchKernelGGL((k_synth_bytes<true>), dim3(grid), dim3(kSynthThreads), 0, st, seed, first, total, len, d_base);
    }
    want = n / kSynthThreads + 1;
    const unsigned g2 = (unsigned)(want < (uint64_t)n_cu * 16 ? want : (uint64_t)n_cu * 16);
    hipLaunchKernelGGL(k_synth_off, dim3(g2), dim3(kSynthThreads), 0, st, n, len, d_off);
    return hipGetLastError();
}

}  // namespace hpn
