// fastq_trim.hip -- gfx950 kernels behind hpn_fastq_trim (include/hpngs.h).
//
// Replaces the cut of readNextNode (reference fastq_trim.c:76-77,83-84):
//   out = line[min(S,len) .. min(E,len))   for the sequence and the quality line.
//
//   k_trim_scan  new length of every record + device-wide exclusive scan
//                (single pass, decoupled look-back, scan.hpp) -> out_off[n+1]
//   k_trim_copy  substring gather: one wave walks 64 records whose boundaries it
//                loaded with one coalesced read; per record the 64 lanes copy
//                consecutive bytes (source and destination spans are both
//                contiguous, so every wave access is one or two cache lines).
// Bound: HBM, read 2*sum(len) + 16 n, write 2*sum(newlen) + 8 n bytes.
#include <stdlib.h>

#include "scan.hpp"

namespace hpn {

constexpr int kTrimThreads = 256;
constexpr int kTrimPerThread = 16;   // 4096 records per tile: every tile costs ~12 ns of chain (ticket + hand-off), 4 per thread made 195 K tiles = 2.4 ms per 2e8 records
#ifndef HPN_TRIM_SCAN_THREADS
#define HPN_TRIM_SCAN_THREADS 256
#endif
constexpr int kTrimScanThreads = HPN_TRIM_SCAN_THREADS;   // the scan's workgroup (256 / 512 / 1024: see profiles/r02/k3_k4_sweeps.txt)
constexpr int kTrimTile = kTrimScanThreads * kTrimPerThread;

__device__ __forceinline__ uint64_t cut_len(uint64_t len, uint64_t S, uint64_t E)
{
    const uint64_t b = S < len ? S : len, e = E < len ? E : len;
    return e > b ? e - b : 0;
}

__global__ __launch_bounds__(kTrimScanThreads) void k_trim_scan(const uint64_t *__restrict__ off, uint64_t n,
                                                           uint64_t S, uint64_t E,
                                                           const uint32_t *__restrict__ pbeg,
                                                           const uint32_t *__restrict__ pend,
                                                           uint64_t *__restrict__ out_off,
                                                           u64 *__restrict__ status,
                                                           uint32_t *__restrict__ ticket,
                                                           uint32_t *__restrict__ err)
{
    __shared__ u64 s_wave[kTrimScanThreads / kWave];
    __shared__ u64 s_excl;
    __shared__ uint32_t s_tile;
    // A thread scans 16 CONSECUTIVE records, but a wave's loads and stores should cover whole lines: the tile's offsets
    // come in (and its results go out) with consecutive lanes on consecutive 8 bytes, through this buffer; slot i of the
    // tile lies at i + i / 16, so that the lanes' 128-byte-strided accesses to it spread over the banks.
    __shared__ u64 s_io[kTrimTile + 1 + (kTrimTile + 1) / kTrimPerThread + 1];
    auto slot = [](uint32_t i) { return i + i / kTrimPerThread; };
    const int tid = threadIdx.x;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);  // tiles in start order: look-back never waits on a tile not yet running
    __syncthreads();
    const uint64_t tile = s_tile;
    const uint64_t tile0 = tile * kTrimTile;
    for (uint32_t i = (uint32_t)tid; i <= (uint32_t)kTrimTile; i += kTrimScanThreads) s_io[slot(i)] = tile0 + i <= n ? off[tile0 + i] : 0;
    __syncthreads();
    const uint64_t base = tile0 + (uint64_t)tid * kTrimPerThread;
    uint64_t o[kTrimPerThread + 1];
#pragma unroll
    for (int k = 0; k <= kTrimPerThread; ++k) o[k] = s_io[slot((uint32_t)tid * kTrimPerThread + k)];
    uint64_t nl[kTrimPerThread];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kTrimPerThread; ++k) {
        // fixed cycles [S,E) of fastq_trim, or this record's own points (quality-threshold trim)
        nl[k] = base + k < n ? cut_len(o[k + 1] - o[k], pbeg ? pbeg[base + k] : S, pend ? pend[base + k] : E) : 0;
        mine += nl[k];
    }
    u64 wtotal;
    const u64 wexcl = wave_excl_scan(mine, wtotal);
    if (lane_id() == kWave - 1) s_wave[wave_id()] = wtotal;
    __syncthreads();                                  // (also: every thread has read its offsets out of s_io)
    u64 before = 0, aggregate = 0;
#pragma unroll
    for (int w = 0; w < kTrimScanThreads / kWave; ++w) {
        if (w < wave_id()) before += s_wave[w];
        aggregate += s_wave[w];
    }
    if (wave_id() == 0) {
        const u64 ex = scan_lookback(status, tile, aggregate, err);
        if (lane_id() == 0) s_excl = ex;
    }
    __syncthreads();
    u64 run = s_excl + before + wexcl;
#pragma unroll
    for (int k = 0; k < kTrimPerThread; ++k) {
        s_io[slot((uint32_t)tid * kTrimPerThread + k)] = run;
        run += nl[k];
    }
    if (tid == kTrimScanThreads - 1) s_io[slot(kTrimTile)] = run;   // the tile's end = the next tile's first value (or out_off[n])
    __syncthreads();
    // records tile0 .. min(tile0 + kTrimTile, n) - 1, and out_off[n] by the tile that holds record n - 1
    for (uint32_t i = (uint32_t)tid; i <= (uint32_t)kTrimTile; i += kTrimScanThreads) {
        const uint64_t r = tile0 + i;
        if (r < n && i < (uint32_t)kTrimTile) out_off[r] = s_io[slot(i)];
        else if (r == n && (i > 0 || n == 0)) out_off[n] = s_io[slot(i)];
    }
}

__global__ __launch_bounds__(kTrimThreads) void k_trim_copy(const uint8_t *__restrict__ seq,
                                                           const uint8_t *__restrict__ qual,
                                                           const uint64_t *__restrict__ off,
                                                           const uint64_t *__restrict__ out_off, uint64_t n,
                                                           uint64_t S, uint64_t E,
                                                           const uint32_t *__restrict__ pbeg,
                                                           const uint32_t *__restrict__ pend,
                                                           uint8_t *__restrict__ out_seq,
                                                           uint8_t *__restrict__ out_qual)
{
    const uint64_t nwaves = (uint64_t)gridDim.x * (kTrimThreads / kWave);
    const uint64_t wave = (uint64_t)blockIdx.x * (kTrimThreads / kWave) + wave_id();
    const int lane = lane_id();
    for (uint64_t r0 = wave * kWave; r0 < n; r0 += nwaves * kWave) {
        const uint64_t r = r0 + lane;
        uint64_t a = 0, len = 0, d = 0, Sr = S, Er = E;
        if (r < n) {
            a = off[r];
            len = off[r + 1] - a;
            d = out_off[r];
            if (pbeg) Sr = pbeg[r];
            if (pend) Er = pend[r];
        }
        const uint64_t b = Sr < len ? Sr : len;
        const uint64_t e = Er < len ? Er : len;
        const uint32_t cnt = e > b ? (uint32_t)(e - b) : 0u;
        const uint64_t src = a + b;
        const int sub = lane & 15, g = lane >> 4;
        const uint32_t c0 = __shfl(cnt, 0, kWave);
        const uint64_t s0 = __shfl(src, 0, kWave), d0 = __shfl(d, 0, kWave);
        if (c0 >= 16 && c0 <= 1024 && __ballot(cnt != c0 || ((src - s0) >> 32) || ((d - d0) >> 32)) == 0) {
            // every record of the wave keeps the same number of bytes (fixed-length reads, always): P = ceil(cnt / 16) lanes
            // serve one record and floor(64 / P) records go per wave-instruction -- 7 at 135 bytes, 63 of 64 lanes busy,
            // 10 steps per 64 records where 16 lanes per record took 16 steps with 9 of 16 busy.  The last piece of a record
            // overlaps the one before it (same bytes written twice) instead of a byte tail.
            const int P = (int)((c0 + 15u) >> 4), rps = kWave / P, steps = (kWave + rps - 1) / rps;
            const int pg = lane / P, ps = lane - pg * P;
            const uint32_t o = min(16u * (uint32_t)ps, c0 - 16u);
            const uint32_t srel = (uint32_t)(src - s0), drel = (uint32_t)(d - d0);   // 64 neighbouring records: within 4 GiB (checked)
            for (int t = 0; t < steps; ++t) {
                const int rec = t * rps + pg;
                const bool on = pg < rps && rec < kWave;
                const uint32_t sj = __shfl(srel, on ? rec : 0, kWave), dj = __shfl(drel, on ? rec : 0, kWave);
                if (on) {
                    u32 s4, q4;
                    __builtin_memcpy(&s4, seq + s0 + sj + o, 16);
                    __builtin_memcpy(&q4, qual + s0 + sj + o, 16);
                    __builtin_memcpy(out_seq + d0 + dj + o, &s4, 16);
                    __builtin_memcpy(out_qual + d0 + dj + o, &q4, 16);
                }
            }
            continue;
        }
        // Mixed lengths.  Four records per wave-instruction: the 16 lanes of quarter g serve record
        // 4*it + g, 16 bytes per lane through unaligned dwordx4 accesses.  A span that is
        // not a multiple of 16 ends with one overlapping 16-byte piece (same bytes written
        // twice); spans shorter than 16 are copied bytewise by the quarter's first lanes.
#pragma unroll 2
        for (int it = 0; it < kWave / 4; ++it) {
            const int j = 4 * it + g;
            const uint64_t sj = __shfl(src, j, kWave), dj = __shfl(d, j, kWave);
            const uint32_t cj = __shfl(cnt, j, kWave);
            if (cj >= 16) {
                for (uint32_t i = 16u * sub; i < cj; i += 256) {
                    const uint32_t o = min(i, cj - 16);
                    u32 s4, q4;
                    __builtin_memcpy(&s4, seq + sj + o, 16);
                    __builtin_memcpy(&q4, qual + sj + o, 16);
                    __builtin_memcpy(out_seq + dj + o, &s4, 16);
                    __builtin_memcpy(out_qual + dj + o, &q4, 16);
                }
            } else if ((uint32_t)sub < cj) {
                out_seq[dj + sub] = seq[sj + sub];
                out_qual[dj + sub] = qual[sj + sub];
            }
        }
    }
}

hipError_t launch_trim(const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint64_t S,
                       uint64_t E, const uint32_t *d_beg, const uint32_t *d_end, uint8_t *d_out_seq, uint8_t *d_out_qual, uint64_t *d_out_off, u64 *d_status,
                       uint32_t *d_ticket_err, int n_cu, hipStream_t st)
{
    const uint64_t ntile = n / kTrimTile + 1;
    hipError_t e = hipMemsetAsync(d_status, 0, ntile * sizeof(u64), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_ticket_err, 0, 2 * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_trim_scan, dim3((unsigned)ntile), dim3(kTrimScanThreads), 0, st, d_off, n, S, E, d_beg, d_end, d_out_off, d_status,
                       d_ticket_err, d_ticket_err + 1);
    if (n) {
        uint64_t want = (n + kTrimThreads - 1) / kTrimThreads;
        const char *eg = test_env("HPN_TRIM_WG_PER_CU");  // A/B knob (scripts/k2_sweep.py)
        const uint64_t cap = (uint64_t)n_cu * (eg ? (uint64_t)atoi(eg) : 8);
        hipLaunchKernelGGL(k_trim_copy, dim3((unsigned)(want < cap ? want : cap)), dim3(kTrimThreads), 0, st, d_seq,
                           d_qual, d_off, d_out_off, n, S, E, d_beg, d_end, d_out_seq, d_out_qual);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// EXTENSION (no reference counterpart): quality-threshold trim points.
// Per record: beg = index of the first quality byte >= T, end = 1 + index of the last
// one (beg = end = 0 when there is none).  One wave per record: 64 bytes per step, one
// ballot, first / last set bit by ffs / clz.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kTrimThreads) void k_qtrim_points(const uint8_t *__restrict__ qual,
                                                              const uint64_t *__restrict__ off, uint64_t n,
                                                              uint32_t T, uint32_t *__restrict__ out_beg,
                                                              uint32_t *__restrict__ out_end)
{
    const uint64_t nwaves = (uint64_t)gridDim.x * (kTrimThreads / kWave);
    const uint64_t wave = (uint64_t)blockIdx.x * (kTrimThreads / kWave) + wave_id();
    const int lane = lane_id();
    for (uint64_t r0 = wave * kWave; r0 < n; r0 += nwaves * kWave) {
        // the wave owns 64 records: boundaries by one coalesced read, then record by record
        const uint64_t r = r0 + lane;
        uint64_t a = 0, len = 0;
        if (r < n) {
            a = off[r];
            len = off[r + 1] - a;
        }
        uint32_t my_beg = 0, my_end = 0;
        const int m = (int)min((uint64_t)kWave, n - r0);
        for (int j = 0; j < m; ++j) {
            const uint64_t aj = __shfl(a, j, kWave), lj = __shfl(len, j, kWave);
            uint32_t first = 0xffffffffu, last = 0;
            for (uint64_t base = 0; base < lj; base += kWave) {
                const uint64_t i = base + lane;
                const uint32_t q = i < lj ? qual[aj + i] : 0u;
                const u64 hit = __ballot(i < lj && q >= T);
                if (hit) {
                    if (first == 0xffffffffu) first = (uint32_t)base + (uint32_t)__builtin_ctzll(hit);
                    last = (uint32_t)base + 64u - (uint32_t)__builtin_clzll(hit);
                }
            }
            if (lane == j) my_beg = first == 0xffffffffu ? 0u : first, my_end = last;
        }
        if (r < n) out_beg[r] = my_beg, out_end[r] = my_end;
    }
}

hipError_t launch_qtrim_points(const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint32_t T, uint32_t *d_beg,
                               uint32_t *d_end, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t want = (n + kTrimThreads - 1) / kTrimThreads;
    const uint64_t cap = (uint64_t)n_cu * 8;
    hipLaunchKernelGGL(k_qtrim_points, dim3((unsigned)(want < cap ? want : cap)), dim3(kTrimThreads), 0, st, d_qual, d_off, n,
                       T, d_beg, d_end);
    return hipGetLastError();
}

uint64_t trim_status_words(uint64_t n) { return n / kTrimTile + 1; }

}  // namespace hpn
