// fastq_gc.hip -- per-read GC fraction, the STATSEQ macro of the reference's R plugin
// (Rgzfastq_uniq.c:50-57): GC = #{'G','C'} / L as a double per read (upper case only, as
// there; L = 0 gives 0/0 = NaN, as there).  The per-cycle Nucleotide / Quality matrices of
// the same macro family come from k_tally_hist (fastq_tally.hip).
//
// 16 lanes per read, four reads per wave-instruction: unaligned 16-byte pieces, SWAR byte
// compares, popcount, 4-step shuffle reduction inside the 16-lane group; the group's first
// lane writes the quotient.  Bound: HBM (Σlen + 8 B/record read, 8 B/record written).
#include "common.hpp"

namespace hpn {

constexpr int kGcThreads = 256;

__device__ __forceinline__ uint32_t eq_bytes(uint32_t x, uint32_t pat)  // 0x80 in the bytes of x equal to pat's
{
    const uint32_t y = x ^ pat;
    return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
}

__global__ __launch_bounds__(kGcThreads) void k_read_gc(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                        uint64_t n, double *__restrict__ gc)
{
    const uint64_t nwaves = (uint64_t)gridDim.x * (kGcThreads / kWave);
    const uint64_t wave = (uint64_t)blockIdx.x * (kGcThreads / kWave) + wave_id();
    const int lane = lane_id(), sub = lane & 15, g = lane >> 4;
    for (uint64_t r0 = wave * kWave; r0 < n; r0 += nwaves * kWave) {
        const uint64_t r = r0 + lane;
        uint64_t a = 0;
        uint32_t len = 0;
        if (r < n) {
            a = off[r];
            len = (uint32_t)(off[r + 1] - a);
        }
#pragma unroll 2
        for (int it = 0; it < kWave / 4; ++it) {
            const int j = 4 * it + g;  // the read this 16-lane group serves now
            const uint64_t aj = __shfl(a, j, kWave);
            const uint32_t lj = __shfl(len, j, kWave);
            uint32_t c = 0;
            if (lj >= 16u) {
                for (uint32_t i = 16u * (uint32_t)sub; i < lj; i += 256u) {
                    // the last piece is moved back so that it ends with the read; its first
                    // (i - o) bytes belong to the piece before and are not counted again
                    const uint32_t o = min(i, lj - 16u), skip = i - o;
                    u32 v;
                    __builtin_memcpy(&v, seq + aj + o, 16);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        uint32_t m = eq_bytes(v[d], 0x47474747u) | eq_bytes(v[d], 0x43434343u);  // 'G' | 'C'
                        const int ignore = (int)skip - 4 * d;
                        if (ignore > 0) m = ignore >= 4 ? 0u : m & (0xffffffffu << (8 * ignore));
                        c += (uint32_t)__builtin_popcount(m);
                    }
                }
            } else if ((uint32_t)sub < lj) {
                const uint32_t b = seq[aj + sub];
                c = (b == 'G' || b == 'C') ? 1u : 0u;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
            if (sub == 0 && r0 + (uint64_t)j < n) gc[r0 + j] = (double)c / (double)lj;  // (GC)/=L, :56
        }
    }
}

hipError_t launch_read_gc(const uint8_t *d_seq, const uint64_t *d_off, uint64_t n, double *d_gc, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t want = (n + kGcThreads - 1) / kGcThreads;
    const uint64_t cap = (uint64_t)n_cu * 8;
    hipLaunchKernelGGL(k_read_gc, dim3((unsigned)(want < cap ? want : cap)), dim3(kGcThreads), 0, st, d_seq, d_off, n, d_gc);
    return hipGetLastError();
}

}  // namespace hpn
