// bam_depth.hip -- gfx950 kernels behind hpn_depth_* (include/hpngs.h).
//
// Replaces, for one target (chromosome) at a time:
//   fetch_func      (reference bam2depth.c:86-110)  -> k_depth_scatter (K3)
//   hash2BedGraph   (bam2depth.c:203-236) + overlap (:132-176) -> k_depth_scan (K4)
//
// The reference keeps two string-keyed hash tables (Start/End breakpoints), sorts
// the union of keys and sweeps it.  Here the breakpoints live in a dense
// difference array diff[0 .. target_len + slack) of int32 in HBM:
//   K3: one lane per record: filter, walk the (short) CIGAR, atomicAdd +1 / -1.
//   K4: ONE pass over diff: inclusive prefix sum = coverage (never materialised),
//       change points -> runs (start, end, depth) of depth > 0 written in order,
//       per-window sums of coverage.  Two decoupled look-back chains (scan.hpp)
//       carry the coverage prefix and the run count between workgroups.
// Bounds: K3 atomic rate (2 x 4 B per M block); K4 HBM read of 4 B per position,
// + 12 B per run + 8 B per window written.
#include <stdlib.h>

#include "scan.hpp"

namespace hpn {

// ---------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------
constexpr int kScatThreads = 256;

__global__ __launch_bounds__(kScatThreads) void k_depth_scatter(
    const int32_t *__restrict__ rec_tid, const int32_t *__restrict__ rec_pos, const uint32_t *__restrict__ rec_flag,
    const uint32_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar, uint64_t n, int32_t tid,
    uint32_t flag_mask, int32_t *__restrict__ diff, uint64_t slots, uint32_t *__restrict__ bad)
{
    for (uint64_t r = (uint64_t)blockIdx.x * kScatThreads + threadIdx.x; r < n;
         r += (uint64_t)gridDim.x * kScatThreads) {
        // bam2depth.c:90: flag & BAM_DEF_MASK or tid < 0 -> skipped; other targets are not ours
        if (rec_tid[r] != tid || tid < 0 || (rec_flag[r] & flag_mask)) continue;
        uint64_t p = (uint32_t)rec_pos[r];  // unsigned int temp_start = c->pos (:93)
        const uint32_t c0 = cigar_off[r], c1 = cigar_off[r + 1];
        for (uint32_t k = c0; k < c1; ++k) {
            const uint32_t w = cigar[k], op = w & 0xfu, len = w >> 4;
            if (op == 2u || op == 3u) {          // D, N: advance only
                p += len;
            } else if (op == 0u) {               // M: +1 at the block start, -1 one past its end
                const uint64_t e = p + len;
                if (e >= slots) {                // breakpoint beyond the dense array (>= 2^28 or huge overhang)
                    atomicOr(bad, 1u);
                    break;
                }
                atomicAdd(&diff[p], 1);
                atomicAdd(&diff[e], -1);
                p = e;
            }                                    // I, S, H, P, =, X: neither counted nor advanced (:94-107)
        }
    }
}

// ---------------------------------------------------------------------------
// K4
// ---------------------------------------------------------------------------
// Tile size: a look-back hop resolves at most 64 tiles (one per lane) and takes ~0.8 us of agent-scope
// round trips, and with every resident workgroup waiting on the same chain the prefix can only advance by
// that much per hop -- measured 12 ns per tile on top of a 0.8 ms streaming floor for chr1, whatever the
// ticket or the dispatch order (a persistent grid and ticket-free tiles changed nothing).  Hence few, large
// tiles: 256 x 16 -> 1.55 ms, 512 x 16 -> 1.26, 1024 x 16 -> 1.00 (1024 x 20 and up spill).
constexpr int kDsThreads = 1024;
constexpr int kDsPer = 16;                       // positions per lane, four 16-byte loads
constexpr int kDsTile = kDsThreads * kDsPer;     // 16384 positions = 64 KiB per workgroup

struct DepthOut {
    hpn_run *runs;
    uint64_t runs_cap;
    u64 *n_runs;          // total number of runs (written by the last tile)
    u64 *win_sum;         // [target_len / W + 1]
};

__global__ __launch_bounds__(kDsThreads) void k_depth_scan(const int32_t *__restrict__ diff, uint64_t slots,
                                                          uint32_t target_len, uint32_t W, DepthOut out,
                                                          u64 *__restrict__ st_cov, u64 *__restrict__ st_cnt,
                                                          uint32_t *__restrict__ ticket, uint32_t *__restrict__ err)
{
    __shared__ u64 s_w[kDsThreads / kWave];
    __shared__ u64 s_x;
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint64_t tile = s_tile;
    const uint64_t p0 = tile * kDsTile + (uint64_t)tid * kDsPer;  // this lane's first position

    int32_t d[kDsPer];
    if (p0 + kDsPer <= slots) {
        const u32 *v = reinterpret_cast<const u32 *>(diff + p0);
#pragma unroll
        for (int k = 0; k < kDsPer / 4; ++k) {
            const u32 q = v[k];
            d[4 * k] = (int32_t)q[0], d[4 * k + 1] = (int32_t)q[1], d[4 * k + 2] = (int32_t)q[2], d[4 * k + 3] = (int32_t)q[3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < kDsPer; ++k) d[k] = p0 + k < slots ? diff[p0 + k] : 0;
    }

    // ---- chain 1: coverage prefix ------------------------------------------------
    int64_t mine = 0;
#pragma unroll
    for (int k = 0; k < kDsPer; ++k) mine += d[k];
    u64 wtot;
    const u64 wex = wave_excl_scan((u64)mine, wtot);
    if (lane_id() == kWave - 1) s_w[wave_id()] = wtot;
    __syncthreads();
    u64 before = 0, agg = 0;
#pragma unroll
    for (int w = 0; w < kDsThreads / kWave; ++w) {
        if (w < wave_id()) before += s_w[w];
        agg += s_w[w];
    }
    if (wave_id() == 0) {
        const u64 ex = scan_lookback(st_cov, tile, agg, err);
        if (lane_id() == 0) s_x = ex;
    }
    __syncthreads();
    // coverage just before this lane's first position (true value is >= 0 and small: the
    // 62-bit modular arithmetic of the chain is exact for it)
    const int64_t cov_in = (int64_t)((s_x + before + wex) & kScanValueMask);
    __syncthreads();  // s_w / s_x are reused below

    // ---- change points, run starts, window sums -------------------------------------
    int32_t cov[kDsPer];
    uint32_t starts = 0;
    {
        int64_t c = cov_in;
#pragma unroll
        for (int k = 0; k < kDsPer; ++k) {
            c += d[k];
            cov[k] = (int32_t)c;
            starts += (d[k] != 0 && c > 0);  // coverage changed here to a positive depth: a run starts
        }
    }
    // ---- chain 2: number of runs started before each lane ------------------------------
    const u64 wex2 = wave_excl_scan((u64)starts, wtot);
    if (lane_id() == kWave - 1) s_w[wave_id()] = wtot;
    __syncthreads();
    before = 0, agg = 0;
#pragma unroll
    for (int w = 0; w < kDsThreads / kWave; ++w) {
        if (w < wave_id()) before += s_w[w];
        agg += s_w[w];
    }
    if (wave_id() == 0) {
        const u64 ex = scan_lookback(st_cnt, tile, agg, err);
        if (lane_id() == 0) s_x = ex;
    }
    __syncthreads();
    u64 idx = ((s_x & kScanValueMask) + before + wex2);  // runs started before this lane's first position
    // A run [s, e) of depth c: at s coverage becomes c > 0; at e it changes again.  With idx = number
    // of runs started before position p: a start at p is run idx, a run ending at p is run idx-1.
    // At 30x nearly every run starts and ends inside one lane's 16 positions: such a run is
    // written as ONE 12-byte store; only runs that cross into another lane are written in two
    // pieces ({start, -, depth} here, `end` by the lane that sees the next change point).
    {
        typedef int32_t i32x3 __attribute__((ext_vector_type(3)));
        int64_t prev = cov_in;
        bool pending = false;  // a run started in this lane and not yet closed
        int32_t rs = 0, rd = 0;
#pragma unroll
        for (int k = 0; k < kDsPer; ++k) {
            if (d[k] != 0) {
                const int32_t p = (int32_t)(p0 + k);
                if (prev > 0 && idx - 1 < out.runs_cap) {
                    if (pending) *reinterpret_cast<i32x3 *>(&out.runs[idx - 1]) = i32x3{rs, p, rd};
                    else out.runs[idx - 1].end = p;
                }
                pending = false;
                if (cov[k] > 0) {
                    rs = p, rd = cov[k], pending = true;
                    ++idx;
                }
            }
            prev = cov[k];
        }
        if (pending && idx - 1 < out.runs_cap) {
            out.runs[idx - 1].start = rs;
            out.runs[idx - 1].depth = rd;
        }
    }
    // ---- window sums (overlap(), bam2depth.c:132-176): sum of coverage per window, clipped at
    // target_len.  Done last, so that these atomics do not sit in front of the look-back loads in
    // the wave's vmcnt queue.  A wave covers 1024 consecutive positions: when those touch at most
    // two windows (W >= 1024, the tool's default is 20000) it reduces both partial sums and issues
    // at most two atomics; per-lane atomics on one address cost ~4 ms per chr1-sized pass.
    if (W) {
        const uint32_t q0 = (uint32_t)p0;                                   // positions are < 2^28
        const uint32_t wave_lo = (uint32_t)(tile * kDsTile) + (uint32_t)wave_id() * (kWave * kDsPer);
        const uint32_t wave_hi = (uint32_t)min((uint64_t)wave_lo + kWave * kDsPer, (uint64_t)target_len);
        if (wave_lo < wave_hi) {
            const uint32_t w0 = wave_lo / W;                                 // one 32-bit division per lane
            const uint32_t nb = (w0 + 1) * W;                                // first position of window w0+1
            if (wave_hi - 1 - w0 * W < 2 * (uint64_t)W) {
                u64 sa = 0, sb = 0;
#pragma unroll
                for (int k = 0; k < kDsPer; ++k) {
                    const uint32_t p = q0 + k;
                    const u64 c = p < target_len ? (u64)(uint32_t)cov[k] : 0;
                    if (p < nb) sa += c;
                    else sb += c;
                }
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) sa += __shfl_xor(sa, o, kWave), sb += __shfl_xor(sb, o, kWave);
                if (lane_id() == 0) {
                    if (sa) atomicAdd(&out.win_sum[w0], sa);
                    if (sb) atomicAdd(&out.win_sum[w0 + 1], sb);
                }
            } else {  // many small windows under one wave: per-lane segments
                u64 s = 0;
                uint32_t w = q0 / W;
                uint32_t nbl = (w + 1) * W;
#pragma unroll
                for (int k = 0; k < kDsPer; ++k) {
                    const uint32_t p = q0 + k;
                    if (p >= target_len) break;
                    if (p == nbl) {
                        if (s) atomicAdd(&out.win_sum[w], s);
                        s = 0, ++w, nbl += W;
                    }
                    s += (u64)(uint32_t)cov[k];
                }
                if (s) atomicAdd(&out.win_sum[w], s);
            }
        }
    }
    if (tile == (slots - 1) / kDsTile && tid == kDsThreads - 1) *out.n_runs = idx;
}

hipError_t launch_depth_scatter(const int32_t *tid_a, const int32_t *pos, const uint32_t *flag, const uint32_t *cigar_off,
                                const uint32_t *cigar, uint64_t n, int32_t tid, uint32_t flag_mask, int32_t *diff,
                                uint64_t slots, uint32_t *bad, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t want = (n + kScatThreads - 1) / kScatThreads;
    const uint64_t cap = (uint64_t)n_cu * 8;
    hipLaunchKernelGGL(k_depth_scatter, dim3((unsigned)(want < cap ? want : cap)), dim3(kScatThreads), 0, st, tid_a, pos,
                       flag, cigar_off, cigar, n, tid, flag_mask, diff, slots, bad);
    return hipGetLastError();
}

uint64_t depth_scan_tiles(uint64_t slots) { return (slots + kDsTile - 1) / kDsTile; }

// ws: [0] ticket, [1] err (uint32 each), then u64 n_runs, then st_cov[tiles], st_cnt[tiles]
hipError_t launch_depth_scan(const int32_t *diff, uint64_t slots, uint32_t target_len, uint32_t W, hpn_run *runs,
                             uint64_t runs_cap, u64 *win_sum, void *ws, hipStream_t st)
{
    const uint64_t tiles = depth_scan_tiles(slots);
    hipError_t e = hipMemsetAsync(ws, 0, 16 + 2 * tiles * sizeof(u64), st);
    if (e != hipSuccess) return e;
    uint32_t *ticket = (uint32_t *)ws;
    u64 *n_runs = (u64 *)ws + 1;
    u64 *st_cov = (u64 *)ws + 2, *st_cnt = st_cov + tiles;
    DepthOut out{runs, runs_cap, n_runs, win_sum};
    hipLaunchKernelGGL(k_depth_scan, dim3((unsigned)tiles), dim3(kDsThreads), 0, st, diff, slots, target_len, W, out, st_cov,
                       st_cnt, ticket, ticket + 1);
    return hipGetLastError();
}

}  // namespace hpn
