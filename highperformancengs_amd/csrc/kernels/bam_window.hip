// bam_window.hip -- gfx950 kernel behind hpn_window_* (include/hpngs.h).
//
// Replaces fetch_func + cal_GC of bam_sliding_count (reference
// bam_sliding_count.c:84-124): per record with tid >= 0 and !(flag & 4)
//     slot = win_off[tid] + (unsigned short)(pos / W)
//     bins[slot] += 1;  gc[slot] += #{4-bit codes == 2 (C) or == 4 (G)};  len[slot] += l_qseq
// All integer here; the float32 arithmetic of calc_winGC (:126-138) is replayed
// on the host in reference order (csrc/host/report.cpp).
//
// One lane per record.  The packed sequence is read with aligned 16-byte loads
// and counted with nibble-wise SWAR.  Records of a coordinate-sorted BAM that are
// near each other fall into the same window, so a workgroup first adds into a
// small LDS table indexed by (slot - lowest slot of its chunk) and flushes that
// with one global atomic per touched window; slots outside the table (unsorted
// input) go to global atomics directly.
// Bound: HBM read of 16 B + ceil(l_qseq/2) B per record.
#include "common.hpp"

namespace hpn {

constexpr int kWinThreads = 256;
constexpr int kWinPer = 4;                        // records per lane per chunk
constexpr int kWinChunk = kWinThreads * kWinPer;  // 1024 records
constexpr int kWinTable = 256;                    // LDS window slots per chunk

// number of nibbles of w equal to 2 or 4 (exact per nibble, no carries)
__device__ __forceinline__ uint32_t gc_nibbles(uint32_t w)
{
    const uint32_t a = w ^ 0x22222222u, b = w ^ 0x44444444u;
    const uint32_t za = ~(((a & 0x77777777u) + 0x77777777u) | a | 0x77777777u);  // 0x8 where nibble == 2
    const uint32_t zb = ~(((b & 0x77777777u) + 0x77777777u) | b | 0x77777777u);  // 0x8 where nibble == 4
    return __builtin_popcount(za) + __builtin_popcount(zb);
}

// GC count of bases [0, l_qseq) of the packed sequence starting at byte s of seq4
// (base i = high nibble of byte i/2 for even i, bam1_seqi, bam.h:260).
__device__ __forceinline__ uint32_t gc_of_record(const uint8_t *__restrict__ seq4, uint64_t s, int32_t l_qseq)
{
    if (l_qseq <= 0) return 0;
    const uint64_t end = s + (uint64_t)((l_qseq + 1) >> 1);
    const uintptr_t base = (uintptr_t)seq4;
    const uintptr_t a0 = (base + s) & ~(uintptr_t)15, a1 = base + end;
    uint32_t gc = 0;
    for (uintptr_t v = a0; v < a1; v += 16) {
        const u32 q = *reinterpret_cast<const u32 *>(seq4 + (v - base));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t w0 = (int64_t)(v + 4 * j) - (int64_t)(base + s);   // record-relative byte of the word's byte 0
            const int64_t nb = (int64_t)(end - s);
            const int lo = (int)min(max(-w0, (int64_t)0), (int64_t)4);
            const int hi = (int)min(max(nb - w0, (int64_t)0), (int64_t)4);
            if (hi <= lo) continue;
            const uint32_t mh = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
            const uint32_t ml = lo >= 4 ? 0xffffffffu : ((1u << (8 * lo)) - 1u);
            uint32_t w = q[j] & mh & ~ml;
            // odd length: the low nibble of the last byte is padding, not a base
            if ((l_qseq & 1) && nb - 1 - w0 >= 0 && nb - 1 - w0 < 4) w &= ~(0xfu << (8 * (int)(nb - 1 - w0)));
            gc += gc_nibbles(w);
        }
    }
    return gc;
}

struct WinLds {
    uint32_t bins[kWinTable];
    uint32_t len[kWinTable];
    uint32_t gc[kWinTable];
    unsigned long long base;  // lowest slot of the chunk
};

__global__ __launch_bounds__(kWinThreads) void k_window_add(
    const int32_t *__restrict__ rec_tid, const int32_t *__restrict__ rec_pos, const uint32_t *__restrict__ rec_flag,
    const int32_t *__restrict__ l_qseq, const uint64_t *__restrict__ seq_off, const uint8_t *__restrict__ seq4,
    uint64_t n, uint32_t W, int32_t n_targets, const uint64_t *__restrict__ win_off, uint32_t *__restrict__ bins,
    u64 *__restrict__ gc, uint32_t *__restrict__ len, uint32_t *__restrict__ touched, u64 *__restrict__ n_count,
    uint32_t *__restrict__ bad)
{
    __shared__ WinLds s;
    const int tid = threadIdx.x;
    u64 counted = 0;
    const uint64_t nchunk = (n + kWinChunk - 1) / kWinChunk;
    for (uint64_t ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        for (int i = tid; i < kWinTable; i += kWinThreads) s.bins[i] = 0, s.len[i] = 0, s.gc[i] = 0;
        if (tid == 0) s.base = ~0ull;
        __syncthreads();
        uint64_t slot[kWinPer];
        uint32_t g[kWinPer], lq[kWinPer];
        bool ok[kWinPer];
#pragma unroll
        for (int k = 0; k < kWinPer; ++k) {
            const uint64_t r = ch * kWinChunk + (uint64_t)k * kWinThreads + tid;
            ok[k] = false;
            slot[k] = 0, g[k] = 0, lq[k] = 0;
            if (r >= n) continue;
            const int32_t t = rec_tid[r];
            if (t < 0 || (rec_flag[r] & 4u)) continue;        // :96-97
            if (t >= n_targets) { atomicOr(bad, 1u); continue; }
            // c->pos / window in int, then (unsigned short) (:117)
            const uint32_t w16 = (uint32_t)(uint16_t)(rec_pos[r] / (int32_t)W);
            const uint64_t lo = win_off[t], hi = win_off[t + 1];
            if (lo + w16 >= hi) { atomicOr(bad, 2u); continue; }   // the reference would write out of bounds
            ok[k] = true;
            slot[k] = lo + w16;
            lq[k] = (uint32_t)l_qseq[r];
            g[k] = (uint32_t)(uint16_t)gc_of_record(seq4, seq_off[r], l_qseq[r]);  // unsigned short current_GC (:118)
            touched[t] = 1u;                                   // benign same-value race
            ++counted;
            atomicMin(&s.base, (unsigned long long)slot[k]);
        }
        __syncthreads();
        const uint64_t base = s.base;
#pragma unroll
        for (int k = 0; k < kWinPer; ++k) {
            if (!ok[k]) continue;
            const uint64_t rel = slot[k] - base;
            if (rel < kWinTable) {
                atomicAdd(&s.bins[rel], 1u);
                atomicAdd(&s.len[rel], lq[k]);
                atomicAdd(&s.gc[rel], g[k]);
            } else {
                atomicAdd(&bins[slot[k]], 1u);
                atomicAdd(&len[slot[k]], lq[k]);
                atomicAdd(&gc[slot[k]], (u64)g[k]);
            }
        }
        __syncthreads();
        if (base != ~0ull) {
            for (int i = tid; i < kWinTable; i += kWinThreads) {
                if (s.bins[i]) {
                    atomicAdd(&bins[base + i], s.bins[i]);
                    atomicAdd(&len[base + i], s.len[i]);
                    atomicAdd(&gc[base + i], (u64)s.gc[i]);
                }
            }
        }
        __syncthreads();
    }
    // n_count (:104)
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) counted += __shfl_xor(counted, o, kWave);
    if (lane_id() == 0 && counted) atomicAdd(n_count, counted);
}

hipError_t launch_window_add(const int32_t *tid_a, const int32_t *pos, const uint32_t *flag, const int32_t *l_qseq,
                             const uint64_t *seq_off, const uint8_t *seq4, uint64_t n, uint32_t W, int32_t n_targets,
                             const uint64_t *win_off, uint32_t *bins, u64 *gc, uint32_t *len, uint32_t *touched,
                             u64 *n_count, uint32_t *bad, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t want = (n + kWinChunk - 1) / kWinChunk;
    const uint64_t cap = (uint64_t)n_cu * 8;
    hipLaunchKernelGGL(k_window_add, dim3((unsigned)(want < cap ? want : cap)), dim3(kWinThreads), 0, st, tid_a, pos, flag,
                       l_qseq, seq_off, seq4, n, W, n_targets, win_off, bins, gc, len, touched, n_count, bad);
    return hipGetLastError();
}

}  // namespace hpn
