// bam_depth.hip -- gfx950 kernels behind hpn_depth_* (include/hpngs.h).
//
// Replaces, for one target (chromosome) at a time:
//   fetch_func      (reference bam2depth.c:86-110)  -> K3: k_depth_index + k_depth_tiles (+ k_depth_fill, k_depth_far)
//   hash2BedGraph   (bam2depth.c:203-236) + overlap (:132-176) -> K4: k_depth_scan
//
// The reference keeps two string-keyed hash tables (Start/End breakpoints), sorts the union of keys and
// sweeps it.  Here the breakpoints live in a dense difference array diff[0 .. target_len + slack) of int32
// in HBM, cut into tiles of 16384 positions (64 KiB), the unit of both kernels.
//
// K3.  bam2depth needs an index, so its input is coordinate-sorted, and a sorted batch turns the scatter
// into a gather: a workgroup OWNS one tile of positions, looks up which records can put a breakpoint
// into it (a contiguous range of the batch: pos in [tile start - kReach, tile end)), walks their CIGARs,
// adds +1 / -1 into an LDS image of the tile and writes the tile with plain coalesced 16-byte stores.
// No global atomic, no zero-fill of the 1 GB array beforehand: a per-tile `written` word says whether the
// tile holds data (then the flush adds to it) or has never been touched (then the scan reads it as zeros
// without loading it).  The per-record atomics this replaces ran at the memory-side atomic rate with 64
// lanes in ~20 different 64-byte segments (1.86 ms per 5e7 records).
//   k_depth_index   one pass over (tid, pos): is the batch sorted by (tid, pos)?  where do the records of
//                   the wanted target start and end, and for every tile threshold inside their position
//                   range: the first record at or beyond it (two lattices: tile start, tile start - kReach).
//   k_depth_tiles   the gather described above.  A breakpoint further than kReach behind its record's
//                   pos (long D / N operations) is "far": the tile that owns the record's pos counts it
//                   and marks the target tile as needed, k_depth_fill zero-fills needed tiles that were never
//                   written and k_depth_far adds the far breakpoints with global atomics (it returns at
//                   once when there are none).  An unsorted batch takes the same two kernels for ALL its
//                   breakpoints: the per-record atomic scatter of round 1.
// K4.  ONE pass over diff: prefix sum = coverage (never materialised), change points -> runs (start, end,
// depth) of depth > 0 written in order, per-window sums of coverage.  ONE decoupled look-back chain
// carries (coverage, runs started): see DepthSum.  A 512-thread workgroup takes four sub-tiles of 8192 positions
// behind one chain entry; a sub-tile's runs are staged in LDS as they lie in memory and flushed as whole 16-byte
// pieces; wave scans are DPP moves.
// Bounds: HBM.  K3 reads 16 B + 4 B x n_cigar per record (x 1.125 for the reach overlap) and writes 4 B per
// position of the tiles it touches; K4 reads 4 B per written position and writes 12 B per run + 8 B per window.
#include <stdlib.h>
#include <string.h>

#include "scan.hpp"

namespace hpn {

// Tile shapes were swept in one session (profiles/r02/k3_k4_sweeps.txt): K3 runs at 0.59-0.62 ms per 5e7 records for
// tiles of 4096 .. 16384 positions and any unroll (it moves 2.2 GB at 4.7 TB/s, reads and writes mixed).
#ifndef HPN_TILE
#define HPN_TILE 16384
#endif
constexpr int kTile = HPN_TILE;             // positions per K3 tile = granularity of the `written` words (A/B builds: 8192)
constexpr int kTileThreads = kTile / 16;    // 16 positions per lane in the flush
constexpr uint32_t kReach = 2048;           // breakpoints up to this far behind pos are gathered by the owner tile
constexpr int kTileUnroll = 4;              // records per lane in flight in k_depth_tiles

// ---- record accessors: SoA batch (hpn_bam_batch) or records in place in inflated BGZF blocks ----------
struct SoaRecs {
    // the CIGAR operations of a batch lie in one array: the reach is BOUNDED by (most operations of a record) x (longest M / D / N
    // operation of the batch) from two streaming passes, instead of walked record by record
    static constexpr bool kCigarArray = true;
    const int32_t *tid, *pos;
    const uint32_t *flag, *cigar_off, *cigar;
    __device__ __forceinline__ void key(uint64_t r, int32_t &t, uint32_t &p) const { t = tid[r], p = (uint32_t)pos[r]; }
    __device__ __forceinline__ bool open(uint64_t r, int32_t want, uint32_t mask, uint32_t &p, const uint32_t *&cig, uint32_t &n) const
    {
        const int32_t t = tid[r];                                 // all five loads issue together
        const uint32_t f = flag[r], c0 = cigar_off[r], c1 = cigar_off[r + 1];
        p = (uint32_t)pos[r];                                     // unsigned int temp_start = c->pos (:93)
        cig = cigar + c0, n = c1 - c0;
        return t == want && !(f & mask);                          // bam2depth.c:90
    }
    __device__ __forceinline__ void cigar_of(uint64_t r, const uint32_t *&cig, uint32_t &n) const
    {
        const uint32_t c0 = cigar_off[r], c1 = cigar_off[r + 1];
        cig = cigar + c0, n = c1 - c0;
    }
    // (the pointer went through a select with nullptr and lost its address space: say it again, a flat load waits on the LDS counter too)
    static __device__ __forceinline__ uint32_t word(const uint32_t *cig, uint32_t k) { return ((const __attribute__((address_space(1))) uint32_t *)cig)[k]; }
};

__device__ __forceinline__ uint32_t ld32u(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// bam1_core_t on disk behind block_size (bam.h:178-187): refID @4, pos @8, l_read_name @12, n_cigar_op @16, flag @18
struct RawRecs {
    static constexpr bool kCigarArray = false;
    const uint8_t *raw;
    const uint64_t *rec_off;
    __device__ __forceinline__ void key(uint64_t r, int32_t &t, uint32_t &p) const
    {
        const uint8_t *q = raw + rec_off[r];
        t = (int32_t)ld32u(q + 4), p = ld32u(q + 8);
    }
    __device__ __forceinline__ bool open(uint64_t r, int32_t want, uint32_t mask, uint32_t &p, const uint32_t *&cig, uint32_t &n) const
    {
        const uint8_t *q = raw + rec_off[r];
        const uint32_t flag_nc = ld32u(q + 16), t = ld32u(q + 4), l_name = q[12];
        p = ld32u(q + 8);
        cig = reinterpret_cast<const uint32_t *>(q + 36u + l_name);
        n = flag_nc & 0xffffu;
        return (int32_t)t == want && !((flag_nc >> 16) & mask);
    }
    __device__ __forceinline__ void cigar_of(uint64_t r, const uint32_t *&cig, uint32_t &n) const
    {
        const uint8_t *q = raw + rec_off[r];
        cig = reinterpret_cast<const uint32_t *>(q + 36u + q[12]);
        n = ld32u(q + 16) & 0xffffu;
    }
    static __device__ __forceinline__ uint32_t word(const uint32_t *cig, uint32_t k)
    {
        uint32_t v;
        __builtin_memcpy(&v, (const __attribute__((address_space(1))) uint8_t *)cig + 4u * k, 4);
        return v;
    }
};

// ---------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------
// head[]: per-add state, cleared before k_depth_index
enum { kHdFlags = 0, kHdFar = 1, kHdR0 = 2, kHdR1 = 3, kHdPmin = 4, kHdPmax = 5, kHdReach = 6, kHdNcig = 7, kHdOplen = 8, kHdWords = 12 };
constexpr uint32_t kUnsorted = 1u;
constexpr uint32_t kLate = 2u;      // a record of the target lies in front of the sweep's frontier

// The sweep (k_depth_sweep): state that lives from hpn_depth_begin to hpn_depth_finish, across hpn_depth_add calls.
//   ctl[kSwFrontier]  first tile not yet swept: everything in front of it has become runs / window sums, its difference
//                     array is never looked at again
//   ctl[kSwLate]      records arrived behind the frontier (input not in coordinate order across calls): the result is void
//   ctl[kSwTicket]    tiles handed out in the current call (start order; reset by k_depth_commit)
//   ctl[kSwErr]       look-back timed out (1) / coverage >= 2^30 (2), as k_depth_scan's err word
//   ctl[kSwEnabled]   0: every batch takes the two-pass route (HPN_DEPTH_ANY_ORDER)
//   status            the look-back chain, ONE entry per tile of kTile positions (k_depth_scan's format), kept across calls:
//                     the entry of tile frontier - 1 is where the next call, and finally k_depth_scan, pick the chain up
enum { kSwFrontier = 0, kSwLate = 1, kSwTicket = 2, kSwErr = 3, kSwEnabled = 4, kSwWords = 16 };
struct SweepState {
    uint32_t *ctl;
    u64 *status;
};

struct TileIndex {
    uint32_t *head;       // [kHdWords]
    uint32_t *first_hi;   // [ntiles + 1]  first record with pos >= t * kTile          (valid for pmin < threshold <= pmax)
    uint32_t *first_lo;   // [ntiles + 1]  first record with pos >= t * kTile - kReach (valid likewise)
    uint32_t *written;    // [ntiles]      the tile holds data
    uint32_t *need;       // [ntiles]      a far breakpoint targets the tile
    uint32_t ntiles;
};

#ifndef HPN_IDX_UNROLL
#define HPN_IDX_UNROLL 4
#endif
constexpr int kIdxThreads = 256;

constexpr int kIdxUnroll = HPN_IDX_UNROLL;   // records per thread whose keys are loaded side by side (one at a time: a dependent round trip each)

// ... and, for the sweep: the batch's REACH = the largest distance of an M block's end from its record's pos (walked from
// the CIGARs: a batch whose breakpoints all lie within kReach of their records' pos has no "far" breakpoints, which is what
// lets a tile be swept by the workgroup that gathered it), and whether a record of the target lies behind the frontier.
template <typename Recs>
__global__ __launch_bounds__(kIdxThreads) void k_depth_index(Recs recs, uint64_t n, int32_t want, TileIndex ix, SweepState sw)
{
    const uint64_t stride = (uint64_t)gridDim.x * kIdxThreads;
    const bool sweeping = sw.ctl[kSwEnabled] != 0;
    const u64 swept = (u64)sw.ctl[kSwFrontier] * kTile;     // positions in front of this are final
    uint32_t reach = 0;
    for (uint64_t i0 = (uint64_t)blockIdx.x * kIdxThreads + threadIdx.x; i0 < n; i0 += stride * kIdxUnroll) {
        int32_t t[kIdxUnroll], tp[kIdxUnroll], tn[kIdxUnroll];
        uint32_t p[kIdxUnroll], pp[kIdxUnroll], pn[kIdxUnroll];
        const uint32_t *cg[kIdxUnroll];
        uint32_t nc[kIdxUnroll], w0[kIdxUnroll];
#pragma unroll
        for (int u = 0; u < kIdxUnroll; ++u) {
            const uint64_t i = i0 + u * stride;
            t[u] = tp[u] = tn[u] = 0, p[u] = pp[u] = pn[u] = 0, cg[u] = nullptr, nc[u] = 0;
            if (i < n) {
                recs.key(i, t[u], p[u]);
                if (i) recs.key(i - 1, tp[u], pp[u]);
                if (i + 1 < n) recs.key(i + 1, tn[u], pn[u]);
                if (sweeping) recs.cigar_of(i, cg[u], nc[u]);     // (where the CIGAR lies: issued beside the keys, not behind them)
            }
        }
        if (sweeping && !Recs::kCigarArray) {
#pragma unroll
            for (int u = 0; u < kIdxUnroll; ++u) w0[u] = nc[u] ? Recs::word(cg[u], 0) : 0u;   // the first operations of all records side by side
        }
#pragma unroll
        for (int u = 0; u < kIdxUnroll; ++u) {
            const uint64_t i = i0 + u * stride;
            if (i >= n) break;
            if (i) {
                // sort order of a BAM: refID as unsigned (unmapped, -1, last), then pos
                const u64 ka = ((u64)(uint32_t)tp[u] << 32) | pp[u], kb = ((u64)(uint32_t)t[u] << 32) | p[u];
                if (ka > kb) atomicOr(&ix.head[kHdFlags], kUnsorted);
            }
            if (t[u] != want) continue;
            if (sweeping) {
                if (p[u] < swept) atomicOr(&ix.head[kHdFlags], kLate);
                uint32_t q = 0, last = 0;                        // relative to pos; saturating is not needed below 2^32 / op
                if (Recs::kCigarArray) last = nc[u];             // (only counted here: k_depth_oplen has the lengths)
                else for (uint32_t k = 0; k < nc[u]; ++k) {
                    const uint32_t w = k ? Recs::word(cg[u], k) : w0[u], op = w & 0xfu, len = w >> 4;
                    if (op == 0u) q += len, last = q;
                    else if (op == 2u || op == 3u) q += len;
                    if (q > (1u << 30)) break;                   // (absurd: reported as a far batch; the two-pass route has the domain check)
                }
                reach = last > reach ? last : reach;
            }
            const bool first = i == 0 || tp[u] != want;
            const bool last = i + 1 == n || tn[u] != want;
            if (first) ix.head[kHdR0] = (uint32_t)i, ix.head[kHdPmin] = p[u];
            if (last) ix.head[kHdR1] = (uint32_t)i + 1u, ix.head[kHdPmax] = p[u];
            if (!first && pp[u] < p[u]) {
                // thresholds in (pp, p]: this is the first record at or beyond them
                for (u64 k = (u64)pp[u] / kTile + 1; k <= (u64)p[u] / kTile && k <= ix.ntiles; ++k) ix.first_hi[k] = (uint32_t)i;
                for (u64 k = ((u64)pp[u] + kReach) / kTile + 1; k <= ((u64)p[u] + kReach) / kTile && k <= ix.ntiles; ++k)
                    ix.first_lo[k] = (uint32_t)i;
            }
        }
    }
    if (sweeping) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const uint32_t other = __shfl_xor(reach, o, kWave);
            reach = other > reach ? other : reach;
        }
        // (one atomic per wave on ONE word is 32 K serialised atomics per chr1, ~0.4 ms: nearly every wave finds its value there already)
        uint32_t *word = &ix.head[Recs::kCigarArray ? kHdNcig : kHdReach];
        if (lane_id() == 0 && reach > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, reach);
    }
}

// longest M / D / N operation among cigar[cigar_off[0] .. cigar_off[n]) (SoA batches): one streaming pass
__global__ __launch_bounds__(256) void k_depth_oplen(const uint32_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar, uint64_t n,
                                                    TileIndex ix, SweepState sw)
{
    if (sw.ctl[kSwEnabled] == 0) return;
    const uint32_t c0 = cigar_off[0], c1 = cigar_off[n];
    uint32_t best = 0;
    for (uint64_t i = (uint64_t)c0 + ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < c1; i += (uint64_t)gridDim.x * 256 * 4) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (i + 4 <= c1) __builtin_memcpy(w, cigar + i, 16);
        else
            for (uint64_t k = i; k < c1; ++k) w[k - i] = cigar[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t op = w[k] & 0xfu, len = w[k] >> 4;
            if ((op == 0u || op == 2u || op == 3u) && len > best) best = len;
        }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(best, o, kWave);
        best = other > best ? other : best;
    }
    if (lane_id() == 0 && best > __hip_atomic_load(&ix.head[kHdOplen], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ix.head[kHdOplen], best);
}

// Does this batch go through the sweep?  The same answer in every kernel of the call (it is a function of the head).
__device__ __forceinline__ bool sweep_takes(const TileIndex &ix, const SweepState &sw)
{
    // reach: walked exactly (records in place), or bounded by operations x longest operation (SoA)
    return sw.ctl[kSwEnabled] != 0 && !(ix.head[kHdFlags] & (kUnsorted | kLate)) && ix.head[kHdReach] <= kReach &&
           (u64)ix.head[kHdNcig] * ix.head[kHdOplen] <= kReach;
}

// first record of the wanted target with pos >= thr (thr in positions): the head's values, or the table entry `tab`
// (loaded by the caller beside the head, whether it is needed or not: one round trip instead of two dependent ones)
__device__ __forceinline__ uint32_t first_at(uint32_t tab, u64 thr, uint32_t r0, uint32_t r1, uint32_t pmin, uint32_t pmax)
{
    if (r0 == r1 || thr <= pmin) return r0;
    if (thr > pmax) return r1;
    return tab;
}

template <typename Recs>
__global__ __launch_bounds__(kTileThreads) void k_depth_tiles(Recs recs, int32_t want, uint32_t flag_mask, int32_t *__restrict__ diff,
                                                             uint64_t slots, TileIndex ix, SweepState sw, uint32_t *__restrict__ bad)
{
    __shared__ int32_t s_d[kTile];
    const uint32_t t = blockIdx.x;
    if (sweep_takes(ix, sw)) return;             // k_depth_sweep did the batch
    // head and table entries in ONE round trip (the table words are garbage outside the batch's position range and then unused)
    const uint32_t h_flags = ix.head[kHdFlags], h_r0 = ix.head[kHdR0], h_r1 = ix.head[kHdR1], h_pmin = ix.head[kHdPmin], h_pmax = ix.head[kHdPmax];
    const uint32_t t_lo = ix.first_lo[t], t_hi = ix.first_hi[t], t_hi1 = ix.first_hi[t + 1], was_written = ix.written[t];
    if (h_flags & kUnsorted) return;             // k_depth_far does the whole batch
    const u64 lo = (u64)t * kTile, hi = lo + kTile;
    const uint32_t r_first = first_at(t_lo, lo > kReach ? lo - kReach : 0, h_r0, h_r1, h_pmin, h_pmax);
    const uint32_t r_home = first_at(t_hi, lo, h_r0, h_r1, h_pmin, h_pmax);
    const uint32_t r_end = t + 1 == ix.ntiles ? h_r1 : first_at(t_hi1, hi, h_r0, h_r1, h_pmin, h_pmax);
    if (r_first >= r_end) return;                // nothing can land here: the tile is not touched
    const int tid = threadIdx.x;
    const bool add = was_written != 0;           // read by every wave before the first barrier; set again after the flush
    {
        u32 *z = reinterpret_cast<u32 *>(s_d);
#pragma unroll
        for (int k = 0; k < kTile / 4 / kTileThreads; ++k) z[k * kTileThreads + tid] = u32{0, 0, 0, 0};
    }
    __syncthreads();
    uint32_t far = 0;
    // kTileUnroll records per lane at a time: their fields and first CIGAR words are loaded side by side
    // (one record after the other is a chain of four dependent loads, ~25 us per tile at 30x)
    for (uint32_t base = r_first; base < r_end; base += kTileUnroll * kTileThreads) {
        uint32_t p[kTileUnroll], n[kTileUnroll], w0[kTileUnroll], w1[kTileUnroll];
        const uint32_t *cig[kTileUnroll];
        bool home[kTileUnroll];
#pragma unroll
        for (int u = 0; u < kTileUnroll; ++u) {
            const uint32_t r = base + u * kTileThreads + tid;
            n[u] = 0, p[u] = 0, cig[u] = nullptr;
            if (r < r_end && !recs.open(r, want, flag_mask, p[u], cig[u], n[u])) n[u] = 0;
            home[u] = r >= r_home;               // the tile that owns pos answers for the record's far breakpoints
        }
#pragma unroll
        for (int u = 0; u < kTileUnroll; ++u) {
            w0[u] = n[u] > 0 ? Recs::word(cig[u], 0) : 0u;
            w1[u] = n[u] > 1 ? Recs::word(cig[u], 1) : 0u;
        }
#pragma unroll
        for (int u = 0; u < kTileUnroll; ++u) {
            u64 q = p[u];
            for (uint32_t k = 0; k < n[u]; ++k) {
                const uint32_t w = k == 0 ? w0[u] : k == 1 ? w1[u] : Recs::word(cig[u], k), op = w & 0xfu, len = w >> 4;
                if (op == 2u || op == 3u) {          // D, N: advance only
                    q += len;
                } else if (op == 0u) {               // M: +1 at the block start, -1 one past its end
                    const u64 e = q + len;
                    if (e >= slots) {                // breakpoint beyond the dense array (>= 2^28 or huge overhang)
                        if (home[u]) atomicOr(bad, 1u);
                        break;
                    }
                    if (q - p[u] <= kReach) {
                        if (q >= lo && q < hi) __hip_atomic_fetch_add(&s_d[q - lo], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (home[u]) {
                        ++far, ix.need[q / kTile] = 1u;
                    }
                    if (e - p[u] <= kReach) {
                        if (e >= lo && e < hi) __hip_atomic_fetch_add(&s_d[e - lo], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (home[u]) {
                        ++far, ix.need[e / kTile] = 1u;
                    }
                    q = e;
                }                                    // I, S, H, P, =, X: neither counted nor advanced (:94-107)
            }
        }
    }
    far = wave_sum(far);
    if (far && lane_id() == 0) atomicAdd(&ix.head[kHdFar], far);
    __syncthreads();
    // flush: consecutive lanes write consecutive 16 bytes, so every store instruction of a wave covers whole lines
    // (a lane writing its own 64 bytes leaves each line to four instructions: partial-line stores)
    if (hi <= slots) {
        u32 *g = reinterpret_cast<u32 *>(diff + lo);
        const u32 *sv = reinterpret_cast<const u32 *>(s_d);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32 v = sv[k * kTileThreads + tid];
            if (add) v += g[k * kTileThreads + tid];
            g[k * kTileThreads + tid] = v;
        }
    } else {
        for (u64 q = lo + tid; q < slots; q += kTileThreads) diff[q] = s_d[q - lo] + (add ? diff[q] : 0);
    }
    if (tid == 0) ix.written[t] = 1u;
}

// tiles a far breakpoint targets (or every tile, for an unsorted batch) that have never been written: zero them
__global__ __launch_bounds__(256) void k_depth_fill(int32_t *__restrict__ diff, uint64_t slots, TileIndex ix, SweepState sw)
{
    const uint32_t t = blockIdx.x;
    if (sweep_takes(ix, sw) || t < sw.ctl[kSwFrontier]) return;   // (swept tiles are never read again)
    const bool all = ix.head[kHdFlags] & kUnsorted;
    if (ix.written[t] || !(all || ix.need[t])) return;
    const u64 lo = (u64)t * kTile;
    for (int k = threadIdx.x * 4; k < kTile; k += 256 * 4) {
        if (lo + k + 4 <= slots) *reinterpret_cast<u32 *>(diff + lo + k) = u32{0, 0, 0, 0};
        else
            for (int j = 0; j < 4 && lo + k + j < slots; ++j) diff[lo + k + j] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) ix.written[t] = 1u, ix.need[t] = 0u;
}

// the breakpoints k_depth_tiles left out (far ones; all of them for an unsorted batch), with global atomics:
// fetch_func's loop one lane per record, as in round 1
constexpr int kFarThreads = 256;
template <typename Recs>
__global__ __launch_bounds__(kFarThreads) void k_depth_far(Recs recs, uint64_t n, int32_t want, uint32_t flag_mask, int32_t *__restrict__ diff,
                                                          uint64_t slots, TileIndex ix, SweepState sw, uint32_t *__restrict__ bad)
{
    if (sweep_takes(ix, sw)) return;
    const bool all = ix.head[kHdFlags] & kUnsorted;
    if (!all && ix.head[kHdFar] == 0) return;
    for (uint64_t r = (uint64_t)blockIdx.x * kFarThreads + threadIdx.x; r < n; r += (uint64_t)gridDim.x * kFarThreads) {
        uint32_t p, nc;
        const uint32_t *cig;
        if (!recs.open(r, want, flag_mask, p, cig, nc)) continue;
        u64 q = p;
        for (uint32_t k = 0; k < nc; ++k) {
            const uint32_t w = Recs::word(cig, k), op = w & 0xfu, len = w >> 4;
            if (op == 2u || op == 3u) {
                q += len;
            } else if (op == 0u) {
                const u64 e = q + len;
                if (e >= slots) {
                    atomicOr(bad, 1u);
                    break;
                }
                if (all || q - p > kReach) atomicAdd(&diff[q], 1);
                if (all || e - p > kReach) atomicAdd(&diff[e], -1);
                q = e;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K4
// ---------------------------------------------------------------------------
// What a stretch of positions tells whoever comes after it, with the coverage c_in at its start unknown:
//   s   sum of its differences (coverage after it = c_in + s)
//   m   smallest inclusive prefix inside it
//   z   positions with a non-zero difference whose inclusive prefix equals m
//   nz  positions with a non-zero difference
// Coverage is never negative, so c_in + m >= 0, and a change point lands on coverage 0 only where the prefix
// equals m and c_in == -m.  Runs started inside the stretch = nz - (c_in == -m ? z : 0).  Two stretches
// compose associatively, so ONE look-back chain carries both the coverage prefix and the number of runs
// started (round 1 ran two chains, one after the other: the second needs the first's result).
struct DepthSum {
    int32_t s, m;
    uint32_t z, nz;
};
constexpr int32_t kMInf = 0x3fffffff;   // m of the empty stretch; |s| < 2^30 keeps s + kMInf inside int32

__device__ __forceinline__ DepthSum ds_identity() { return DepthSum{0, kMInf, 0u, 0u}; }
__device__ __forceinline__ DepthSum ds_compose(const DepthSum &a, const DepthSum &b)   // a first, then b
{
    const int32_t mb = a.s + b.m, m = a.m < mb ? a.m : mb;
    return DepthSum{a.s + b.s, m, (a.m == m ? a.z : 0u) + (mb == m ? b.z : 0u), a.nz + b.nz};
}
__device__ __forceinline__ DepthSum ds_shfl_up(const DepthSum &v, int o)
{
    return DepthSum{__shfl_up(v.s, o, kWave), __shfl_up(v.m, o, kWave), __shfl_up(v.z, o, kWave), __shfl_up(v.nz, o, kWave)};
}
__device__ __forceinline__ DepthSum ds_shfl_down(const DepthSum &v, int o)
{
    return DepthSum{__shfl_down(v.s, o, kWave), __shfl_down(v.m, o, kWave), __shfl_down(v.z, o, kWave), __shfl_down(v.nz, o, kWave)};
}
// Scans across a wave with DPP moves (full-rate VALU; __shfl_up is an LDS round trip per field and step): shifts by
// 1, 2, 4, 8 inside rows of 16 lanes, then lane 15 of a row into the next row (rows 1, 3), then lane 31 into rows 2, 3.
// Lanes without a source keep `old` = the identity.
template <int kCtrl, int kRows>
__device__ __forceinline__ DepthSum ds_dpp(const DepthSum &v)
{
    return DepthSum{__builtin_amdgcn_update_dpp(0, v.s, kCtrl, kRows, 0xf, false), __builtin_amdgcn_update_dpp(kMInf, v.m, kCtrl, kRows, 0xf, false),
                    (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, kCtrl, kRows, 0xf, false),
                    (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.nz, kCtrl, kRows, 0xf, false)};
}
__device__ __forceinline__ DepthSum ds_wave_inclusive(DepthSum v);
__device__ __forceinline__ uint32_t ds_starts(const DepthSum &v, int64_t c_in) { return v.nz - ((int64_t)v.m == -c_in ? v.z : 0u); }

__device__ __forceinline__ DepthSum ds_wave_inclusive(DepthSum v)
{
    v = ds_compose(ds_dpp<0x111, 0xf>(v), v);
    v = ds_compose(ds_dpp<0x112, 0xf>(v), v);
    v = ds_compose(ds_dpp<0x114, 0xf>(v), v);
    v = ds_compose(ds_dpp<0x118, 0xf>(v), v);
    v = ds_compose(ds_dpp<0x142, 0xa>(v), v);
    v = ds_compose(ds_dpp<0x143, 0xc>(v), v);
    return v;
}
__device__ __forceinline__ DepthSum ds_lane_before(const DepthSum &inc) { return ds_dpp<0x138, 0xf>(inc); }   // wave_shr:1; lane 0: the identity

constexpr int kLbWaves = 1;    // waves of a workgroup that poll side by side, 64 tiles each (2 / 4 / 8 / 16: 0.80 / 0.82 / 0.84 / 0.89 ms)
constexpr int kStStride = 8;   // u64 words between the status entries of consecutive tiles: two entries per 128-byte line
                               // (0.85 -> 0.80 ms against packed entries: fewer pollers per line)


// Chain status: 16 bytes per tile = two 8-byte granules {flag:2, a:31, b:31}, (s, m) and (z, nz), each written and
// read with one agent-scope relaxed access (scan.hpp); a reader takes a tile only when both flags agree.
__device__ __forceinline__ void ds_publish(u64 *status, uint64_t tile, u64 flag, const DepthSum &v)
{
    const u64 h0 = (flag << 62) | ((u64)((uint32_t)v.s & 0x7fffffffu) << 31) | (u64)((uint32_t)v.m & 0x7fffffffu);
    const u64 h1 = (flag << 62) | ((u64)(v.z & 0x7fffffffu) << 31) | (u64)(v.nz & 0x7fffffffu);
    __hip_atomic_store(&status[kStStride * tile], h0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&status[kStStride * tile + 1], h1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t sext31(uint32_t x) { return (int32_t)(x << 1) >> 1; }

// One look-back window, all lanes of one wave: lane L inspects tile newest - L.  Spins until those tiles have
// published at least their aggregate, then returns (in every lane) the composition, oldest first, of the tiles from
// the nearest one that holds a full prefix up to `newest`; *has_prefix = such a tile exists in the window.
// Tiles before 0 count as the empty prefix.
#ifdef DIAG_SWEEP_STAMPS
__device__ uint32_t g_diag_polls[1024];   // polls per tile slot (tile & 1023): one writer at a time, summed by the reader
#endif
__device__ __forceinline__ DepthSum ds_window(u64 *status, int64_t newest, bool *has_prefix, uint32_t *err)
{
    const int64_t t = newest - lane_id();
    u64 h0 = kScanPrefix << 62, h1 = kScanPrefix << 62;
    uint32_t spins = 0;
    for (;;) {
        if (t >= 0) {
            h0 = __hip_atomic_load(&status[kStStride * t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h1 = __hip_atomic_load(&status[kStStride * t + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool ready = (h0 >> 62) != kScanInvalid && (h0 >> 62) == (h1 >> 62);
        if (__ballot(!ready) == 0) break;
        if (++spins > kScanSpinLimit) {           // a hand-off that never arrives: report, and end the look-back
            if (lane_id() == 0) atomicOr(err, 1u);
            *has_prefix = true;
            return ds_identity();
        }
        __builtin_amdgcn_s_sleep(1);
    }
#ifdef DIAG_SWEEP_STAMPS
    if (lane_id() == 0) g_diag_polls[(newest + 1) & 1023] = spins + 1u;
#endif
    const u64 pfx = __ballot((h0 >> 62) == kScanPrefix);
    const int stop = pfx ? __builtin_ctzll(pfx) : kWave;                      // lane of the nearest full prefix
    DepthSum v = ds_identity();
    if (lane_id() <= stop && t >= 0)
        v = DepthSum{sext31((uint32_t)(h0 >> 31) & 0x7fffffffu), sext31((uint32_t)h0 & 0x7fffffffu),
                     (uint32_t)(h1 >> 31) & 0x7fffffffu, (uint32_t)h1 & 0x7fffffffu};
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {         // ordered reduction: higher lanes hold older tiles and come first
        const DepthSum older = ds_shfl_down(v, o);
        if (lane_id() + o < kWave) v = ds_compose(older, v);
    }
    *has_prefix = pfx != 0;
    return DepthSum{__shfl(v.s, 0, kWave), __shfl(v.m, 0, kWave), __shfl(v.z, 0, kWave), __shfl(v.nz, 0, kWave)};
}

constexpr int kDsThreads = 512;                  // two workgroups per CU (61 KB of LDS and 8 waves of <= 128 VGPRs each): one's loads, look-back
                                                 // and store drain overlap the other's arithmetic.  1024 / 768 / 512 / 256: 0.52 / 0.57 / 0.48 / 0.51 ms
constexpr int kDsPer = 16;                       // positions per lane, four 16-byte loads (32: 128 VGPRs + spills, 0.87 ms)
constexpr int kDsTile = kDsThreads * kDsPer;     // positions per workgroup
static_assert(kTile % kDsPer == 0, "a lane's 16 positions lie in one K3 tile (whatever the scan's own tile)");

struct DepthOut {
    hpn_run *runs;
    uint64_t runs_cap;
    u64 *n_runs;          // total number of runs (written by the last tile)
    u64 *win_sum;         // [target_len / W + 1]
};

// what this lane's 16 positions tell their successors
__device__ __forceinline__ DepthSum scan_lane_sum(const int32_t (&d)[kDsPer])
{
    int32_t pre = 0, m = INT32_MAX;
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < kDsPer; ++k) {
        pre += d[k];
        m = pre < m ? pre : m;
        nz += d[k] != 0;
    }
    uint32_t z = 0;
    pre = 0;
#pragma unroll
    for (int k = 0; k < kDsPer; ++k) {
        pre += d[k];
        z += (d[k] != 0 && pre == m);
    }
    return DepthSum{pre, m, z, nz};
}

// Runs and window sums of one lane's 16 positions (first one p0), given everything before them.  Returns the number of
// runs started up to and including these positions.
// Runs are staged in LDS and written out by scan_flush() as whole 16-byte pieces in lane order.  The image holds the
// sub-tile's stretch of runs[] dword for dword, from the 16-byte piece that holds the `end` of run base-1 (base = runs
// started before the sub-tile: that run may be open on entry and closed here): word j of the image is dword origin + j of
// runs[].  Runs beyond the image (a sub-tile of dense change points) go to memory directly.
constexpr uint32_t kStageRuns = 5100;             // 61 KB of LDS; a sub-tile of 8192 positions at 30x holds ~2200 runs
constexpr uint32_t kStageWords = 3 * (kStageRuns + 1) + 8;
__device__ __forceinline__ int32_t stage_origin(uint32_t base) { return ((int32_t)(3u * base) - 2) & ~3; }   // (-4 for base 0: never touched)

// Runs and window sums of one lane's 16 positions (first one p0), given everything before them.  Returns the number of
// runs started up to and including these positions.
// (kRuns: runs the staging area holds -- kStageRuns for k_depth_scan's sub-tiles, kSwStageRuns for a tile of the sweep)
template <uint32_t kRuns = kStageRuns>
__device__ __forceinline__ uint32_t scan_emit(const int32_t (&d)[kDsPer], const DepthSum &before, uint64_t p0, uint64_t sub_first,
                                              uint32_t target_len, uint32_t W, uint32_t Wm, const DepthOut &out, uint32_t *__restrict__ err,
                                              uint32_t *__restrict__ stage, uint32_t base)
{
    // 32-bit from here on: coverage stays below 2^30 (else `over` reports the target as out of domain) and a target has
    // fewer than 2^28 runs -- the 64-bit forms of these sixteen steps were a third of the kernel's VALU instructions
    const int32_t cov_in = before.s;             // coverage just before this lane's first position
    uint32_t idx = ds_starts(before, 0);         // runs started before it (the target starts at coverage 0)
    const int32_t origin = stage_origin(base);
    const int32_t pb = (int32_t)p0;
    uint32_t over = 0, tot = 0;                  // tot: sum of the 16 coverages (exact in 32 bits while they stay below 2^27)
    bool tot_ok = false;

    // ---- change points -> runs ------------------------------------------------------------
    // A run [s, e) of depth c: at s coverage becomes c > 0; at e it changes again.  With idx = number
    // of runs started before position p: a start at p is run idx, a run ending at p is run idx-1.
    if (__ballot(idx + kDsPer - base > kRuns) == 0) {
        // every piece lies inside the image: no branches, three predicated LDS stores per position
        uint32_t *q = stage + ((int32_t)(3u * idx) - 2 - origin);   // `end` of run idx-1; run idx starts two words on
        uint32_t *const q0 = q;
        int32_t c = cov_in;
#pragma unroll
        for (int k = 0; k < kDsPer; ++k) {
            const bool was = c > 0;
            c += d[k];
            over |= (uint32_t)c;
            tot += (uint32_t)c;
            const bool chg = d[k] != 0;
            if (chg && was) q[0] = (uint32_t)(pb + k);
            if (chg && c > 0) {
                q[2] = (uint32_t)(pb + k);
                q[4] = (uint32_t)c;
                q += 3;
            }
        }
        idx += (uint32_t)(q - q0) / 3u;
        tot_ok = true;
    } else {
        // At 30x nearly every run starts and ends inside one lane's 16 positions: such a run is
        // written as ONE 12-byte store; only runs that cross into another lane are written in two
        // pieces ({start, -, depth} here, `end` by the lane that sees the next change point).
        typedef int32_t i32x3 __attribute__((ext_vector_type(3)));
        const uint32_t cap32 = out.runs_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)out.runs_cap;
        int32_t c = cov_in;
        bool pending = false;  // a run started in this lane and not yet closed
        int32_t rs = 0, rd = 0;
#pragma unroll
        for (int k = 0; k < kDsPer; ++k) {
            const int32_t prev = c;
            c += d[k];
            over |= (uint32_t)c;                 // bit 30 or 31 set: coverage >= 2^30 (or negative): outside the chain's 31-bit fields
            if (d[k] != 0) {
                const int32_t p = pb + k;
                if (prev > 0) {
                    const uint32_t r = idx - 1u;               // >= base - 1
                    if (r + 1u - base <= kRuns) {
                        uint32_t *q = stage + ((int32_t)(3u * r) - origin);
                        if (pending) q[0] = (uint32_t)rs, q[2] = (uint32_t)rd;
                        q[1] = (uint32_t)p;
                    } else if (r < cap32) {
                        if (pending) *reinterpret_cast<i32x3 *>(&out.runs[r]) = i32x3{rs, p, rd};
                        else out.runs[r].end = p;
                    }
                }
                pending = false;
                if (c > 0) {
                    rs = p, rd = c, pending = true;
                    ++idx;
                }
            }
        }
        if (pending) {
            const uint32_t r = idx - 1u;
            if (r + 1u - base <= kRuns) {
                uint32_t *q = stage + ((int32_t)(3u * r) - origin);
                q[0] = (uint32_t)rs, q[2] = (uint32_t)rd;
            } else if (r < cap32) {
                out.runs[r].start = rs;
                out.runs[r].depth = rd;
            }
        }
    }
#ifndef DIAG_NOCHAIN
    if (over >> 30) atomicOr(err, 2u);
#endif
#ifdef DIAG_NOWIN
    W = 0;
#endif
    // ---- window sums (overlap(), bam2depth.c:132-176): sum of coverage per window, clipped at
    // target_len.  A wave covers 1024 consecutive positions: when those touch at most two windows
    // (W >= 1024, the tool's default is 20000) it reduces both partial sums and issues at most two
    // atomics; per-lane atomics on one address cost ~4 ms per chr1-sized pass.
    if (W) {
        const uint32_t q0 = (uint32_t)p0;                                   // positions are < 2^28
        const uint32_t wave_lo = (uint32_t)sub_first + (uint32_t)wave_id() * (kWave * kDsPer);
        const uint32_t wave_hi = (uint32_t)min((uint64_t)wave_lo + kWave * kDsPer, (uint64_t)target_len);
        if (wave_lo < wave_hi) {
            const uint32_t w0 = div_by(wave_lo, W, Wm);
            const uint32_t nb = (w0 + 1) * W;                                // first position of window w0+1
            if (tot_ok && wave_lo + kWave * kDsPer <= min(nb, target_len) && __ballot(over >> 27) == 0) {
                // the usual wave: inside one window, the sums already taken above
                tot = wave_sum(tot);
                if (lane_id() == 0 && tot) atomicAdd(&out.win_sum[w0], (u64)tot);
            } else if (wave_hi - 1 - w0 * W < 2 * (uint64_t)W) {
                u64 sa = 0, sb = 0;
                {
                    int32_t c = cov_in;
                    uint32_t a32 = 0, b32 = 0;           // sixteen coverages below 2^30 may pass 2^32: carried into 64 bits twice
#pragma unroll
                    for (int k = 0; k < kDsPer; ++k) {
                        const uint32_t p = q0 + k;
                        c += d[k];
                        const uint32_t cc = p < target_len ? (uint32_t)c : 0u;
                        if (p < nb) a32 += cc;
                        else b32 += cc;
                        if (k == kDsPer / 4 - 1 || k == kDsPer / 2 - 1 || k == 3 * kDsPer / 4 - 1 || k == kDsPer - 1) sa += a32, sb += b32, a32 = 0, b32 = 0;
                    }
                }
                // coverage below 2^21 everywhere under the wave (always, outside pile-ups): the 1024-position sums fit 32 bits
                if (__ballot((sa | sb) >> 25) == 0) {
                    uint32_t a32 = (uint32_t)sa, b32 = (uint32_t)sb;
#pragma unroll
                    for (int o = kWave / 2; o > 0; o >>= 1) a32 += __shfl_xor(a32, o, kWave), b32 += __shfl_xor(b32, o, kWave);
                    sa = a32, sb = b32;
                } else {
#pragma unroll
                    for (int o = kWave / 2; o > 0; o >>= 1) sa += __shfl_xor(sa, o, kWave), sb += __shfl_xor(sb, o, kWave);
                }
                if (lane_id() == 0) {
                    if (sa) atomicAdd(&out.win_sum[w0], sa);
                    if (sb) atomicAdd(&out.win_sum[w0 + 1], sb);
                }
            } else {  // many small windows under one wave: per-lane segments
                u64 s = 0;
                uint32_t w = div_by(q0, W, Wm);
                uint32_t nbl = (w + 1) * W;
                int32_t c = cov_in;
#pragma unroll
                for (int k = 0; k < kDsPer; ++k) {
                    const uint32_t p = q0 + k;
                    if (p >= target_len) break;
                    if (p == nbl) {
                        if (s) atomicAdd(&out.win_sum[w], s);
                        s = 0, ++w, nbl += W;
                    }
                    c += d[k];
                    s += (u64)(uint32_t)c;
                }
                if (s) atomicAdd(&out.win_sum[w], s);
            }
        }
    }
    return idx;
}

// The staged runs [base, next) of a sub-tile -> memory, 16 bytes per lane, consecutive lanes consecutive pieces.  Left alone:
// the `end` of a run still open where the sub-tile ends (cov_end > 0; whoever sees the next change point writes it), and of
// run base-1 all but its `end`, and that only when the sub-tile closed it (closes_prev).
template <int kThreads = kDsThreads, uint32_t kRuns = kStageRuns>
__device__ __forceinline__ void scan_flush(const uint32_t *__restrict__ stage, uint32_t base, uint32_t next, bool closes_prev,
                                           int32_t cov_end, const DepthOut &out)
{
    uint32_t n = next - base;
    n = n < kRuns ? n : kRuns;
    const int32_t origin = stage_origin(base);
    const int32_t g_lo = (int32_t)(3u * base);                           // in dwords of runs[]; < 2^30
    const uint64_t lim = 3ull * out.runs_cap;
    int32_t g_hi = g_lo + (int32_t)(3u * n);
    if ((uint64_t)g_hi > lim) g_hi = (int32_t)lim;
    const int32_t prev_end = closes_prev && (uint64_t)g_lo <= lim ? g_lo - 2 : -8;
    if (g_hi <= g_lo && prev_end < 0) return;
    const int32_t skip = (cov_end > 0 && next - base - 1u < kRuns) ? (int32_t)(3u * (next - 1u)) + 1 : -8;
    uint32_t *__restrict__ dst = reinterpret_cast<uint32_t *>(out.runs);
    for (int32_t g = origin + 4 * (int32_t)threadIdx.x; g < g_hi; g += 4 * kThreads) {
        const u32 v = *reinterpret_cast<const u32 *>(stage + (g - origin));
        if (g >= g_lo && g + 4 <= g_hi && (uint32_t)(skip - g) >= 4u) {
            *reinterpret_cast<u32 *>(dst + g) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (((g + e >= g_lo && g + e < g_hi) || g + e == prev_end) && g + e != skip) dst[g + e] = v[e];
        }
    }
}

// kDsSub sub-tiles of 8192 positions per workgroup, ONE ticket, ONE chain entry, one drain of the stores: the loads of all
// sub-tiles are in flight together, and what a tile pays once whatever its size -- the ticket's round trip, the look-back's
// (3.3 us under load), the wait for its stores before the workgroup may leave (4.3 us) -- is paid per 32768 positions.
// (32 positions per lane in ONE scan step spilled; steps of 16 with their own registers do not.)
constexpr int kDsSub = 4;                        // with 1024 threads, 1 / 2 / 3 / 4 sub-tiles: 0.81 / 0.74 / 0.68 / 0.67 ms; 64 registers of differences
constexpr int kDsGroup = kDsSub * kDsTile;       // positions per workgroup = per chain entry
static_assert(kDsSub * (kDsThreads / kWave) <= kWave, "wave 0 scans one (sub-tile, wave) total per lane");

// Starts at the sweep's frontier (position pos0 = frontier * kTile; 0 when nothing was swept): the chain entry of the tile in
// front of it (sw.status, decoded by ds_load_prefix) is what the first group here continues from.
__device__ __forceinline__ DepthSum ds_load_prefix(const u64 *status, uint64_t tile)
{
    const u64 h0 = __hip_atomic_load(&status[kStStride * tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 h1 = __hip_atomic_load(&status[kStStride * tile + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return DepthSum{sext31((uint32_t)(h0 >> 31) & 0x7fffffffu), sext31((uint32_t)h0 & 0x7fffffffu), (uint32_t)(h1 >> 31) & 0x7fffffffu,
                    (uint32_t)h1 & 0x7fffffffu};
}

__global__ __launch_bounds__(kDsThreads) void k_depth_scan(const int32_t *__restrict__ diff, const uint32_t *__restrict__ written,
                                                          uint64_t slots, uint32_t target_len, uint32_t W, DepthOut out,
                                                          u64 *__restrict__ status, uint32_t *__restrict__ ticket,
                                                          uint32_t *__restrict__ err, SweepState sw)
{
    constexpr int kWaves = kDsThreads / kWave;
    const uint32_t frontier = sw.ctl[kSwFrontier];
    const uint64_t pos0 = (uint64_t)frontier * kTile;
    if (pos0 + (uint64_t)blockIdx.x * kDsGroup >= slots) return;   // the grid is sized for the whole target
    __shared__ DepthSum s_w[kDsSub * kWaves + 1]; // [sub][wave]: in stream order; the last one: the whole group
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[kStageWords];
    __shared__ DepthSum s_lb[kLbWaves];
    __shared__ uint32_t s_lbp[kLbWaves];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint64_t tile = s_tile;                // the group's index = its chain entry
    const uint32_t Wm = W ? 0xffffffffu / W : 0u; // div_by()

    int32_t d[kDsSub][kDsPer];
    uint64_t p0[kDsSub];
#pragma unroll
    for (int sb = 0; sb < kDsSub; ++sb) {
        p0[sb] = pos0 + tile * kDsGroup + (uint64_t)sb * kDsTile + (uint64_t)tid * kDsPer;  // this lane's first position in the sub-tile
        const bool have = p0[sb] < slots && written[p0[sb] / kTile] != 0;
        if (!have) {                             // never touched by K3: zeros, nothing to load
#pragma unroll
            for (int k = 0; k < kDsPer; ++k) d[sb][k] = 0;
        } else if (p0[sb] + kDsPer <= slots) {
            const u32 *v = reinterpret_cast<const u32 *>(diff + p0[sb]);
#pragma unroll
            for (int k = 0; k < kDsPer / 4; ++k) {
                const u32 q = v[k];
                d[sb][4 * k] = (int32_t)q[0], d[sb][4 * k + 1] = (int32_t)q[1], d[sb][4 * k + 2] = (int32_t)q[2], d[sb][4 * k + 3] = (int32_t)q[3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kDsPer; ++k) d[sb][k] = p0[sb] + k < slots ? diff[p0[sb] + k] : 0;
        }
    }
    // ---- this lane's stretches, then lanes -> waves -> group -> chain ----------------------
    DepthSum lanes_before[kDsSub];
#pragma unroll
    for (int sb = 0; sb < kDsSub; ++sb) {
        const DepthSum inc = ds_wave_inclusive(scan_lane_sum(d[sb]));
        lanes_before[sb] = ds_lane_before(inc);
        if (lane_id() == kWave - 1) s_w[sb * kWaves + wave_id()] = inc;
    }
    __syncthreads();
    // wave 0: the (sub-tile, wave) totals in stream order -> what lies before each of them inside the group, and the group's
    // aggregate, published at once
    DepthSum excl = ds_identity(), agg = ds_identity();   // (wave 0 only)
    if (wave_id() == 0) {
        const DepthSum w = ds_wave_inclusive(lane_id() < kDsSub * kWaves ? s_w[lane_id()] : ds_identity());
        agg = DepthSum{__shfl(w.s, kDsSub * kWaves - 1, kWave), __shfl(w.m, kDsSub * kWaves - 1, kWave),
                       __shfl(w.z, kDsSub * kWaves - 1, kWave), __shfl(w.nz, kDsSub * kWaves - 1, kWave)};
        excl = ds_lane_before(w);                      // what comes before this lane's (sub-tile, wave)
        if (tile > 0 && lane_id() == 0) ds_publish(status, tile, kScanAggregate, agg);
    }
    // The chain.  Per-tile stamps (profiles/r02/k3_k4_sweeps.txt): a tile finds a full prefix within one hop (1.03 hops,
    // 1.26 polls on average) -- it does not wait for its predecessors -- but that one round trip of agent-scope loads
    // takes 3.3 us under streaming load (two 512-thread workgroups per CU overlap each other's).  Wider hops
    // (more tiles per lane, several polling waves), more workgroups per CU and larger tiles were all measured slower.
    DepthSum exclusive = ds_identity();               // of tiles 0 .. tile-1; the same in every thread
    if (tile == 0 && frontier > 0) exclusive = ds_load_prefix(sw.status, frontier - 1u);   // what the sweep left
#ifdef DIAG_NOCHAIN
    if (false) {
#else
    if (tile > 0) {
#endif
        int64_t newest = (int64_t)tile - 1;
        for (;;) {
            if (wave_id() < kLbWaves) {
                bool hp = false;
                const DepthSum win = ds_window(status, newest - (int64_t)kWave * wave_id(), &hp, err);
                if (lane_id() == 0) s_lb[wave_id()] = win, s_lbp[wave_id()] = hp ? 1u : 0u;
            }
            __syncthreads();
            bool found = false;
#pragma unroll
            for (int v = 0; v < kLbWaves; ++v) {
                if (!found) {
                    exclusive = ds_compose(s_lb[v], exclusive);   // older windows come first
                    found = s_lbp[v] != 0;
                }
            }
            if (found) break;
            newest -= (int64_t)kWave * kLbWaves;
            __syncthreads();                          // s_lb is written again
        }
    }
    if (wave_id() == 0) {
        if (lane_id() == 0) {
            const DepthSum all = ds_compose(exclusive, agg);
            ds_publish(status, tile, kScanPrefix, all);
            s_w[kDsSub * kWaves] = all;
        }
        if (lane_id() < kDsSub * kWaves) s_w[lane_id()] = ds_compose(exclusive, excl);   // everything before (sub-tile, wave) `lane`
    }
    __syncthreads();
    uint32_t idx = 0;
#pragma unroll
    for (int sb = 0; sb < kDsSub; ++sb) {
        const DepthSum before = ds_compose(s_w[sb * kWaves + wave_id()], lanes_before[sb]);   // everything before this lane's positions
        const DepthSum upto0 = s_w[sb * kWaves];                             // everything before the sub-tile
        const DepthSum upto = s_w[(sb + 1) * kWaves];                        // everything up to its end
        const uint32_t base = ds_starts(upto0, 0);                           // runs started before the sub-tile
        idx = scan_emit(d[sb], before, p0[sb], pos0 + tile * kDsGroup + (uint64_t)sb * kDsTile, target_len, W, Wm, out, err, s_stage, base);
        __syncthreads();
#ifndef DIAG_NOFLUSH
        scan_flush(s_stage, base, ds_starts(upto, 0), upto0.s > 0 && upto.nz != upto0.nz, upto.s, out);
#endif
        if (sb + 1 < kDsSub) __syncthreads();     // the image is written again
    }
    if (tile == (slots - 1 - pos0) / kDsGroup && tid == kDsThreads - 1) *out.n_runs = idx;   // (< 2^28: one run needs a position)
}

// ---------------------------------------------------------------------------
// K3 + K4 in one: the sweep
// ---------------------------------------------------------------------------
// The difference array's round trip through HBM -- written by k_depth_tiles, read back by k_depth_scan: 2 GB of the two
// kernels' 3.3 GB per chr1 -- is what is left between them and their roofline.  In a coordinate-sorted input a tile that the
// batch's LAST record lies beyond can receive nothing more: the workgroup that gathered it (as k_depth_tiles does) goes
// straight on and sweeps it from LDS (as k_depth_scan does): prefix sum, runs, window sums, with the look-back chain kept
// across hpn_depth_add calls.  Only the tile the batch ends in (and the few behind it its last records reach) go to HBM, to be
// completed by the next call; hpn_depth_finish's k_depth_scan starts at the frontier and sees only those.
// Takes a batch that is sorted, has no far breakpoints (reach <= kReach, proven by k_depth_index from the CIGARs) and
// does not start behind the frontier; every other batch goes the two-pass way as before, whenever it comes.
// LDS image of the tile, laid out for what comes after the gather: lane l of sub-tile sb holds positions 16 l .. 16 l + 15
// as four quads, and quad j of all lanes lies together (conflict-free 16-byte reads):
//   position p -> word (p & ~8191) | (((p >> 2 & 3) * 512 + ((p & 8191) >> 4)) << 2) | (p & 3)
// One workgroup of kTile / 16 = 1024 threads per tile, 16 positions per lane, two workgroups per CU (64 KB of LDS and <= 64
// VGPRs each): 32 waves per CU keep the record loads of the gather in flight, as k_depth_tiles does.
//   position p -> word (((p >> 2 & 3) * 1024 + (p >> 4)) << 2) | (p & 3)
constexpr int kSwThreads = kTile / kDsPer;
#ifndef HPN_SW_LB_WAVES
#define HPN_SW_LB_WAVES 1
#endif
constexpr int kSwLbWaves = HPN_SW_LB_WAVES;                  // waves of a tile's workgroup that look back side by side
constexpr uint32_t kSwStageRuns = (kTile - 8) / 3 - 1;       // runs of one tile the image's LDS can stage (5457; a tile at 30x holds ~4400)
static_assert(kSwThreads == kTileThreads && 3 * (kSwStageRuns + 1) + 8 <= (uint32_t)kTile, "the tile image doubles as the runs' staging area");
constexpr int kSwLog = kTile == 16384 ? 10 : kTile == 8192 ? 9 : kTile == 4096 ? 8 : -1;   // log2(lanes of a tile's workgroup)
static_assert(kSwLog > 0 && (1 << kSwLog) == kTile / kDsPer, "tile sizes the sweep's LDS image is laid out for");
__device__ __forceinline__ uint32_t sw_word(uint32_t p) { return (((((p >> 2) & 3u) << kSwLog) + (p >> 4)) << 2) | (p & 3u); }

template <typename Recs>
__global__ __launch_bounds__(kSwThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_depth_sweep(Recs recs, int32_t want, uint32_t flag_mask, int32_t *__restrict__ diff, uint64_t slots, TileIndex ix, SweepState sw,
                   uint32_t target_len, uint32_t W, DepthOut out, uint32_t *__restrict__ bad)
{
    constexpr int kWaves = kSwThreads / kWave;
#ifdef DIAG_SWEEP_STAMPS
    uint32_t st_[12];      // (32-bit stamps in registers, totals by atomics at the end: a printf in the kernel quadruples its time)
    int st_n = 0;
#define SW_STAMP() st_[st_n++] = (uint32_t)wall_clock64()
#else
#define SW_STAMP()
#endif
    SW_STAMP();
    __shared__ __attribute__((aligned(16))) int32_t s_d[kTile];
    __shared__ DepthSum s_w[kWaves + 1];
    __shared__ DepthSum s_lb[kSwLbWaves];
    __shared__ uint32_t s_lbp[kSwLbWaves];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x;
    if (!sweep_takes(ix, sw)) return;
    const uint32_t h_r0 = ix.head[kHdR0], h_r1 = ix.head[kHdR1], h_pmin = ix.head[kHdPmin], h_pmax = ix.head[kHdPmax];
    if (h_r0 == h_r1) return;                                  // no record of the target in this batch
    const uint32_t frontier = sw.ctl[kSwFrontier];
    // tiles in front of `closed` end before the batch's last record: final.  The target's last tile is always left to k_depth_scan.
    uint32_t closed = h_pmax / kTile, t_end = (uint32_t)(((u64)h_pmax + kReach) / kTile);
    closed = closed < ix.ntiles - 1u ? closed : ix.ntiles - 1u;
    t_end = t_end < ix.ntiles - 1u ? t_end : ix.ntiles - 1u;
    if (blockIdx.x > t_end - frontier) return;                 // (frontier <= tile of pmin <= t_end: the batch is not late)
    // The tile's table entries are fetched for the tile this workgroup will most likely get (workgroups start in blockIdx order)
    // beside the ticket, not behind it: one round trip instead of two dependent ones (~2 us of a tile's ~25).
    const uint32_t tg = frontier + blockIdx.x;
    uint32_t t_lo = ix.first_lo[tg], t_hi1 = ix.first_hi[tg + 1], was_written = ix.written[tg];
    if (tid == 0) s_tile = atomicAdd(&sw.ctl[kSwTicket], 1u);   // tiles in start order: the look-back only waits on running workgroups
    __syncthreads();
    const uint32_t t = frontier + s_tile;
    if (t != tg) t_lo = ix.first_lo[t], t_hi1 = ix.first_hi[t + 1], was_written = ix.written[t];
    const u64 lo = (u64)t * kTile, hi = lo + kTile;
    const uint32_t r_first = first_at(t_lo, lo > kReach ? lo - kReach : 0, h_r0, h_r1, h_pmin, h_pmax);
    const uint32_t r_end = t + 1 == ix.ntiles ? h_r1 : first_at(t_hi1, hi, h_r0, h_r1, h_pmin, h_pmax);
    const bool sweep = t < closed, gather = r_first < r_end;
    if (!sweep && !gather) return;                             // an open tile nothing lands in
    SW_STAMP();
    if (gather) {
        u32 *z = reinterpret_cast<u32 *>(s_d);
#pragma unroll
        for (int k = 0; k < kTile / 4 / kSwThreads; ++k) z[k * kSwThreads + tid] = u32{0, 0, 0, 0};
        __syncthreads();
        for (uint32_t base = r_first; base < r_end; base += kTileUnroll * kSwThreads) {
            uint32_t p[kTileUnroll], n[kTileUnroll], w0[kTileUnroll], w1[kTileUnroll], w2[kTileUnroll];
            const uint32_t *cig[kTileUnroll];
#pragma unroll
            for (int u = 0; u < kTileUnroll; ++u) {
                const uint32_t r = base + u * kSwThreads + tid;
                n[u] = 0, p[u] = 0, cig[u] = nullptr;
                if (r < r_end && !recs.open(r, want, flag_mask, p[u], cig[u], n[u])) n[u] = 0;
            }
#ifdef DIAG_SWEEP_STAMPS
            { uint32_t sink = 0;
              for (int u = 0; u < kTileUnroll; ++u) sink += n[u] + p[u];
              if (sink == 0xdeadbeefu) st_[11] = 1; }
            uint32_t g1 = (uint32_t)wall_clock64();
#endif
#pragma unroll
            for (int u = 0; u < kTileUnroll; ++u) {
                w0[u] = n[u] > 0 ? Recs::word(cig[u], 0) : 0u;
                w1[u] = n[u] > 1 ? Recs::word(cig[u], 1) : 0u;
                w2[u] = n[u] > 2 ? Recs::word(cig[u], 2) : 0u;   // (a third operation loaded inside the walk is a memory round trip per record and lane, one after the other)
            }
#ifdef DIAG_SWEEP_STAMPS
            { uint32_t sink = 0;
              for (int u = 0; u < kTileUnroll; ++u) sink += w0[u] + w1[u] + w2[u];
              if (sink == 0xdeadbeefu) st_[11] = 2; }
            uint32_t g2 = (uint32_t)wall_clock64();
            st_[8] = g1, st_[9] = g2;
#endif
        // positions in 32 bits: slots <= 2^28 and an operation is shorter than 2^28, so q + len cannot wrap while q is kept below 2^31
        const uint32_t lo32 = (uint32_t)lo, slots32 = (uint32_t)slots;
#pragma unroll
            for (int u = 0; u < kTileUnroll; ++u) {
                uint32_t q = p[u] < 0x80000000u ? p[u] : 0x80000000u;
                for (uint32_t k = 0; k < n[u]; ++k) {
                    const uint32_t w = k == 0 ? w0[u] : k == 1 ? w1[u] : k == 2 ? w2[u] : Recs::word(cig[u], k), op = w & 0xfu, len = w >> 4;
                    if (op == 2u || op == 3u) {          // D, N: advance only
                        q += len;
                        q = q < 0x80000000u ? q : 0x80000000u;
                    } else if (op == 0u) {               // M: +1 at the block start, -1 one past its end (bam2depth.c:94-107)
                        const uint32_t e = q + len;
                        if (e >= slots32) {              // breakpoint beyond the dense array (>= 2^28 or huge overhang)
                            atomicOr(bad, 1u);
                            break;
                        }
                        if (q - lo32 < (uint32_t)kTile) __hip_atomic_fetch_add(&s_d[sw_word(q - lo32)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (e - lo32 < (uint32_t)kTile) __hip_atomic_fetch_add(&s_d[sw_word(e - lo32)], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        q = e;
                    }                                    // I, S, H, P, =, X: neither counted nor advanced
                }
            }
        }
        __syncthreads();
    }
    if (!sweep) {
        // the batch's records do not end beyond this tile: to HBM in natural order (+ what earlier calls left), for the next call
        if (hi <= slots) {
            u32 *g = reinterpret_cast<u32 *>(diff + lo);
#pragma unroll
            for (int k = 0; k < kTile / 4 / kSwThreads; ++k) {
                const uint32_t Q = (uint32_t)(k * kSwThreads + tid);
                u32 v = *reinterpret_cast<const u32 *>(&s_d[sw_word(4u * Q)]);
                if (was_written) v += g[Q];
                g[Q] = v;
            }
        } else {
            for (u64 q = lo + tid; q < slots; q += kSwThreads) diff[q] = s_d[sw_word((uint32_t)(q - lo))] + (was_written ? diff[q] : 0);
        }
        if (tid == 0) ix.written[t] = 1u;
        return;
    }
    SW_STAMP();
    // ---- sweep: this lane's 16 positions out of LDS (+ the tile's earlier content in HBM), then k_depth_scan's steps ----
    int32_t d[kDsPer];
    const uint64_t p0 = lo + (uint64_t)tid * kDsPer;
#pragma unroll
    for (int j = 0; j < kDsPer / 4; ++j) {
        u32 q = u32{0, 0, 0, 0};
        if (gather) q = reinterpret_cast<const u32 *>(s_d)[j * kSwThreads + tid];
        d[4 * j] = (int32_t)q[0], d[4 * j + 1] = (int32_t)q[1], d[4 * j + 2] = (int32_t)q[2], d[4 * j + 3] = (int32_t)q[3];
    }
    if (was_written) {                                         // (closed tiles lie wholly inside the array: the last tile is never swept)
        const u32 *v = reinterpret_cast<const u32 *>(diff + p0);
#pragma unroll
        for (int j = 0; j < kDsPer / 4; ++j) {
            const u32 q = v[j];
            d[4 * j] += (int32_t)q[0], d[4 * j + 1] += (int32_t)q[1], d[4 * j + 2] += (int32_t)q[2], d[4 * j + 3] += (int32_t)q[3];
        }
    }
    uint32_t *const s_stage = reinterpret_cast<uint32_t *>(s_d);   // the image is in registers: its LDS stages the runs
    const DepthSum inc = ds_wave_inclusive(scan_lane_sum(d));
    const DepthSum lanes_before = ds_lane_before(inc);
    if (lane_id() == kWave - 1) s_w[wave_id()] = inc;
    __syncthreads();                                           // (also: every lane has read its quads)
    DepthSum excl = ds_identity(), agg = ds_identity();        // (wave 0 only)
    if (wave_id() == 0) {
        const DepthSum w = ds_wave_inclusive(lane_id() < kWaves ? s_w[lane_id()] : ds_identity());
        agg = DepthSum{__shfl(w.s, kWaves - 1, kWave), __shfl(w.m, kWaves - 1, kWave), __shfl(w.z, kWaves - 1, kWave), __shfl(w.nz, kWaves - 1, kWave)};
        excl = ds_lane_before(w);
        if (t > 0 && lane_id() == 0) ds_publish(sw.status, t, kScanAggregate, agg);
    }
    SW_STAMP();
    DepthSum exclusive = ds_identity();                        // of tiles 0 .. t-1 (earlier calls' included); the same in every thread
    if (t > 0) {
        // A tile's aggregate exists only after its gather (~10 us in), so the 512 tiles in flight reach this point at about the same
        // time and wait for each other: 10-15 us per tile on average, the sweep's longest step (stamps in profiles/r03).  More
        // waves looking back side by side (64 tiles each) only add polling traffic: 1 / 2 / 4 / 16 waves: 0.94 / 0.96 / 0.99 / 1.14 ms.
        // A tile's aggregate exists only after its gather (~10 us in), so the 512 tiles in flight reach this point at about the same
        // time and wait for each other: 10-15 us per tile, the sweep's longest step (stamps in profiles/r03/sweep_stamps.txt).
        // Step 1, one wave: the 64 tiles in front of this one, polled until each has at least its aggregate.  Step 2, only when none
        // of them holds a prefix yet: kSwLbWaves waves look further back side by side.  Measured per chr1 (index + sweep): all
        // windows polled by 1 / 2 / 4 / 16 waves from the start 0.94 / 0.96 / 0.99 / 1.14 ms; this two-step form with 8 waves 0.97:
        // the wait is for the aggregates themselves, not for the prefix to travel, and more polling only takes from the gathers.
        int64_t newest = (int64_t)t - 1;
        if (wave_id() == 0) {
            bool hp = false;
            const DepthSum win = ds_window(sw.status, newest, &hp, &sw.ctl[kSwErr]);
            if (lane_id() == 0) s_lb[0] = win, s_lbp[0] = hp ? 1u : 0u;
        }
        __syncthreads();
        exclusive = s_lb[0];
        bool found = s_lbp[0] != 0;
        newest -= kWave;
        while (!found) {
            __syncthreads();                                   // s_lb is written again
            if (wave_id() < kSwLbWaves) {
                bool hp = false;
                const DepthSum win = ds_window(sw.status, newest - (int64_t)kWave * wave_id(), &hp, &sw.ctl[kSwErr]);
                if (lane_id() == 0) s_lb[wave_id()] = win, s_lbp[wave_id()] = hp ? 1u : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < kSwLbWaves; ++v) {
                if (!found) {
                    exclusive = ds_compose(s_lb[v], exclusive);   // older windows come first
                    found = s_lbp[v] != 0;
                }
            }
            newest -= (int64_t)kWave * kSwLbWaves;
        }
    }
    if (wave_id() == 0) {
        if (lane_id() == 0) {
            const DepthSum all = ds_compose(exclusive, agg);
            ds_publish(sw.status, t, kScanPrefix, all);
            s_w[kWaves] = all;
        }
        if (lane_id() < kWaves) s_w[lane_id()] = ds_compose(exclusive, excl);   // everything before wave `lane`
    }
    __syncthreads();
    SW_STAMP();
    const uint32_t Wm = W ? 0xffffffffu / W : 0u;
    const DepthSum before = ds_compose(s_w[wave_id()], lanes_before);
    const DepthSum upto0 = s_w[0];
    const DepthSum upto = s_w[kWaves];
    const uint32_t base = ds_starts(upto0, 0);
    (void)scan_emit<kSwStageRuns>(d, before, p0, lo, target_len, W, Wm, out, &sw.ctl[kSwErr], s_stage, base);
    __syncthreads();
    SW_STAMP();
    scan_flush<kSwThreads, kSwStageRuns>(s_stage, base, ds_starts(upto, 0), upto0.s > 0 && upto.nz != upto0.nz, upto.s, out);
#ifdef DIAG_SWEEP_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    SW_STAMP();
    // (the tile's stretch of the difference array is free once it is swept: the stamps go there, plain stores, nothing shared;
    // hpn_depth_finish reads them back when HPN_SWEEP_DIAG is set)
    if (tid == 0) {
        uint32_t *o = reinterpret_cast<uint32_t *>(diff + lo);
        o[0] = 0x5354414du, o[1] = st_[1] - st_[0], o[2] = st_[2] - st_[1], o[3] = st_[3] - st_[2], o[4] = st_[4] - st_[3], o[5] = st_[5] - st_[4],
        o[6] = st_[6] - st_[5], o[7] = g_diag_polls[t & 1023u];
    }
#endif
}

// After the sweep of a call: the frontier moves up to the tile the batch's last record lies in, the tickets start over.
__global__ void k_depth_commit(TileIndex ix, SweepState sw)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ix.head[kHdFlags] & kLate) sw.ctl[kSwLate] = 1u;
    if (sweep_takes(ix, sw) && ix.head[kHdR0] != ix.head[kHdR1]) {
        uint32_t closed = ix.head[kHdPmax] / kTile;
        closed = closed < ix.ntiles - 1u ? closed : ix.ntiles - 1u;
        if (closed > sw.ctl[kSwFrontier]) sw.ctl[kSwFrontier] = closed;
    }
    sw.ctl[kSwTicket] = 0u;
}

// Window sums of coverage from the runs (what overlap() does, bam2depth.c:132-176), for a window size other than the one
// the sweep was told: sum over runs of depth x overlap with [kW, min((k+1)W, target_len)).  Runs are sorted, so the 64 runs
// under a wave mostly lie in one window: one atomic per wave then, else per lane and window.
__global__ __launch_bounds__(256) void k_win_from_runs(const hpn_run *__restrict__ runs, uint64_t n_runs, uint32_t target_len, uint32_t W,
                                                      u64 *__restrict__ win_sum)
{
    const uint32_t Wm = 0xffffffffu / W;
    for (uint64_t r0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) & ~(uint64_t)(kWave - 1); r0 < n_runs; r0 += (uint64_t)gridDim.x * 256) {
        const uint64_t r = r0 + lane_id();
        uint32_t s = 0, e = 0, dp = 0;
        if (r < n_runs) {
            typedef int32_t i32x3 __attribute__((ext_vector_type(3)));
            const i32x3 v = *reinterpret_cast<const i32x3 *>(&runs[r]);
            s = (uint32_t)v[0], e = (uint32_t)v[1], dp = (uint32_t)v[2];
            e = e < target_len ? e : target_len;               // windows are clipped at the contig end (:132-176); runs may overhang it
            if (s >= e) s = e = 0, dp = 0;
        }
        const uint32_t k0 = div_by(s, W, Wm), k1 = e > s ? div_by(e - 1u, W, Wm) : k0;
        const uint32_t w_first = __shfl(k0, 0, kWave);
        if (__ballot(dp != 0 && (k0 != w_first || k1 != w_first)) == 0) {
            u64 tot = (u64)(e - s) * dp;
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, kWave);
            if (lane_id() == 0 && tot) atomicAdd(&win_sum[w_first], tot);
        } else if (dp) {
            for (uint32_t k = k0; k <= k1; ++k) {
                const uint32_t a = s > k * W ? s : k * W, b = e < (k + 1) * W ? e : (k + 1) * W;
                if (b > a) atomicAdd(&win_sum[k], (u64)(b - a) * dp);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// bedGraph text on the device: fprintf(bedGraph, "%s\t%d\t%d\t%d\n", chr, start, end, depth) per run
// (bam2depth.c:217).  hg38 at 30x is ~1e9 such lines = ~25 GB of text: formatted by one host thread that is the
// longest step of the tool once the scan runs on the GPU.  Here a workgroup takes 512 runs, sizes their lines,
// finds where its text starts (decoupled look-back over the tiles' byte counts, scan.hpp), builds the text in
// LDS and copies it out in 16-byte pieces.
// ---------------------------------------------------------------------------
constexpr int kFmtThreads = 256, kFmtPer = 2, kFmtSub = kFmtThreads * kFmtPer;   // 512 runs are staged at a time ...
#ifndef HPN_BG_SUBS
#define HPN_BG_SUBS 16
#endif
constexpr int kFmtSubs = HPN_BG_SUBS, kFmtTile = kFmtSub * kFmtSubs;   // ... and a workgroup takes 8 such pieces in a row: one chain entry
                                                             // per 4096 runs (one per 512 was 131 K entries x ~14 ns = 1.9 of 2.0 ms)
// bytes of text a piece may stage (512 lines of up to 78 bytes).  With the kernel's other ~600 bytes this stays within 32 of the
// CU's LDS granules of 1,280 bytes (scripts/micro/lds_occupancy.hip): four workgroups per CU; 40,960 bytes of text were 33 granules
// and three.
#ifndef HPN_BG_LDS
#define HPN_BG_LDS 24576
#endif
constexpr int kFmtLds = HPN_BG_LDS;                   // (A/B builds: smaller -> more workgroups per CU; a piece that does not fit goes straight to memory)
constexpr int kFmtMaxName = 44;                       // longest target name the staged path takes
constexpr int kFmtWaveLds = (kFmtLds / (kFmtThreads / kWave)) & ~15;   // every wave's own part (128 lines)

__device__ __forceinline__ int dec_digits(uint32_t v)
{
    return 1 + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}
// Decimal digits without a division per digit (v % 10, v / 10 is a quarter-rate 32-bit multiply-high and a multiply-low
// per digit; the formatting, not the memory traffic, is 0.8 of the kernel's 1.1-1.3 ms: ~400 vector instructions per line).  The number is cut into
// v = hi * 10^8 + mid * 10^4 + lo with one multiply-high, and every 4-digit group becomes four ASCII bytes with 24-bit
// multiplies (full rate): x / 100 = x * 5243 >> 19 for x < 43699, y / 10 = y * 103 >> 10 for y < 179.
__device__ __forceinline__ uint32_t ascii4(uint32_t x)   // x < 10000 -> "dddd", thousands in byte 0
{
    const uint32_t h = __umul24(x, 5243u) >> 19, l = x - __umul24(h, 100u);
    const uint32_t ht = __umul24(h, 103u) >> 10, ho = h - __umul24(ht, 10u);
    const uint32_t lt = __umul24(l, 103u) >> 10, lo = l - __umul24(lt, 10u);
    return (ht | ho << 8 | lt << 16 | lo << 24) + 0x30303030u;
}
// decimal digits of v into p[0 .. nd), most significant first (nd = dec_digits(v))
template <typename P>
__device__ __forceinline__ void put_dec(P p, uint32_t v, int nd)
{
    if (__ballot(v >= 10000u) == 0) {                                     // (depths: one group of four digits for the whole wave)
        const uint32_t s2 = ascii4(v), skip = 4u - (uint32_t)nd;
        P q = p - skip;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
            if (j >= skip) q[j] = (uint8_t)(s2 >> (8 * j));
        return;
    }
    const uint32_t hi = v / 100000000u;                                   // 0 .. 42
    const uint32_t r = v - (__umul24(hi, 390625u) << 8);                  // 10^8 = 390625 * 256
    // r / 10000 for r < 10^8: (r >> 4) / 625 with a 24-bit multiply-high (13743896 = ceil(2^33 / 625); exact below 2^23)
    const uint32_t mid = (uint32_t)(((u64)((r >> 4) & 0x7fffffu) * 13743896ull) >> 33);   // v_mul_hi_u32_u24
    const uint32_t lo = r - __umul24(mid, 10000u);
    // the ten characters "hhmmmmllll", the last nd of them to p[0 .. nd): character j goes to (p - skip)[j] for j >= skip
    const uint32_t ht = __umul24(hi, 103u) >> 10;                           // hi < 43: two digits
    const uint32_t s0 = ((ht | (hi - __umul24(ht, 10u)) << 8) << 16) + 0x30300000u, s1 = ascii4(mid), s2 = ascii4(lo);   // of s0, the last two characters
    const uint32_t skip = 10u - (uint32_t)nd;                              // 0 .. 9 leading zeros to drop
    P q = p - skip;
#pragma unroll
    for (uint32_t j = 0; j < 10; ++j) {
        if (j >= skip) q[j] = (uint8_t)((j < 2 ? s0 : j < 6 ? s1 : s2) >> (8 * ((j + 2) & 3)));
    }
}
// (runs travel as three scalars: an array of hpn_run structs indexed in unrolled loops was kept in scratch memory)
// Digit count from a table indexed by the bit length b of v: tab[b] = d << 32 | 10^d with d = digits of 2^(b-1); the count is
// d, or d + 1 from 10^d on.  Two LDS reads and four instructions instead of nine compares and adds.
__device__ __forceinline__ int dec_digits_tab(uint32_t v, const u64 *tab)
{
    const u64 e = ((const __attribute__((address_space(3))) u64 *)tab)[32 - __builtin_clz(v | 1u)];   // (LDS: a ds_read, not a flat load)
    return (int)(e >> 32) + (v >= (uint32_t)e);
}
__device__ __forceinline__ void dec_digits_fill(u64 *tab, int tid)   // threads 0 .. 32 of the workgroup, before a barrier
{
    if (tid <= 32) {
        const u64 lo = tid ? 1ull << (tid - 1) : 0;            // smallest value of bit length tid (0: v == 0 is looked up as 1)
        uint32_t d = 1;
        u64 p10 = 10;
        while (p10 <= lo) p10 *= 10, ++d;
        tab[tid] = (u64)d << 32 | (p10 > 0xffffffffull ? 0xffffffffull : p10);
    }
}

// length of the line in bits 0..15, the digit counts of its three numbers in bits 16..19, 20..23, 24..27 (counted once)
__device__ __forceinline__ uint32_t line_info(int32_t start, int32_t end, int32_t depth, int name_len, const u64 *tab)
{
    const int neg = (start < 0) + (end < 0) + (depth < 0);   // never, for runs the scan emits; kept printf-exact
    const int n1 = dec_digits_tab((uint32_t)abs(start), tab), n2 = dec_digits_tab((uint32_t)abs(end), tab), n3 = dec_digits_tab((uint32_t)abs(depth), tab);
    return (uint32_t)(name_len + 4 + neg + n1 + n2 + n3) | (uint32_t)n1 << 16 | (uint32_t)n2 << 20 | (uint32_t)n3 << 24;
}
template <typename P>
__device__ __forceinline__ int put_field(P p, int at, int32_t f, int nd)
{
    p[at++] = '\t';
    if (f < 0) p[at++] = '-';
    put_dec(p + at, (uint32_t)abs(f), nd);
    return at + nd;
}
// The target's name: its first eight characters from two SGPRs (chr1 .. chr22, chrX, chrM: the whole name), the rest from
// `name` (LDS in the staged path).  Read from the kernel argument character by character, the name was a memory round
// trip per character and line -- most of the kernel's time.
template <typename P>
__device__ __forceinline__ void put_line(P p, int32_t start, int32_t end, int32_t depth, uint32_t info, uint32_t n0, uint32_t n1, const uint8_t *name,
                                         int name_len)
{
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < name_len) p[k] = (uint8_t)((k < 4 ? n0 : n1) >> (8 * (k & 3)));   // (name_len is uniform: a scalar branch)
    for (int k = 8; k < name_len; ++k) p[k] = name[k];
    int at = put_field(p, name_len, start, (int)(info >> 16) & 15);
    at = put_field(p, at, end, (int)(info >> 20) & 15);
    at = put_field(p, at, depth, (int)(info >> 24) & 15);
    p[at] = '\n';
}

// ---- lines of ONE layout (round 4) ----------------------------------------------------------------------------------------------
// What made the byte-wise line expensive was not the bytes but deciding, per character and lane, whether to store it (digits are
// right-aligned: ~100 compare / mask / store groups per line).  In a wave of neighbouring runs the three numbers of a line nearly
// always have the same digit counts in every lane (coordinates cross a power of ten a handful of times per chromosome; depths mostly
// stay within 10..99), and then the line has ONE layout for the whole wave: every field sits at a wave-uniform offset, the
// digits -- shifted to the front of their registers with wave-uniform shifts -- are stored without any predicate, and a lane's
// address arithmetic is one add per field.  A start that equals the end of the line before (adjacent runs: the rule at 30x) takes
// that line's digits.  Waves whose lines differ take put_line as before.
struct Dig {
    uint32_t a, b, c;        // characters 0..3, 4..7, 8..9 of the number, most significant first
};
// the nd (wave-uniform, 1..10) decimal digits of v, left-justified
__device__ __forceinline__ Dig dec_left(uint32_t v, int nd)
{
    const uint32_t hi = v / 100000000u;                                   // 0 .. 42
    const uint32_t r = v - (__umul24(hi, 390625u) << 8);                  // 10^8 = 390625 * 256
    const uint32_t mid = (uint32_t)(((u64)((r >> 4) & 0x7fffffu) * 13743896ull) >> 33);   // r / 10000 (see put_dec)
    const uint32_t lo = r - __umul24(mid, 10000u);
    const uint32_t ht = __umul24(hi, 103u) >> 10;
    const uint32_t s0 = (ht | (hi - __umul24(ht, 10u)) << 8) + 0x3030u, s1 = ascii4(mid), s2 = ascii4(lo);   // "hh", "mmmm", "llll"
    // the ten characters as a stream x0 x1 x2 ("hhmm", "mmll", "ll"), then moved to the front by skip = 10 - nd characters
    const uint32_t x0 = s0 | s1 << 16, x1 = __builtin_amdgcn_alignbyte(s2, s1, 2), x2 = s2 >> 16;
    const int skip = 10 - nd, d = skip >> 2, b = skip & 3;                // wave-uniform: scalar selects
    const uint32_t a0 = d == 0 ? x0 : d == 1 ? x1 : x2, a1 = d == 0 ? x1 : d == 1 ? x2 : 0u, a2 = d == 0 ? x2 : 0u;
    return Dig{__builtin_amdgcn_alignbyte(a1, a0, (uint32_t)b), __builtin_amdgcn_alignbyte(a2, a1, (uint32_t)b), a2 >> (8 * b)};
}
// the first n (wave-uniform, <= 4) characters of w to p[0..n): byte stores, the odd bytes through one shift
__device__ __forceinline__ void put_chars(uint8_t *p, uint32_t w, int n)
{
    const uint32_t o = w >> 8;
    if (n > 0) p[0] = (uint8_t)w;
    if (n > 1) p[1] = (uint8_t)o;
    if (n > 2) p[2] = (uint8_t)(w >> 16);
    if (n > 3) p[3] = (uint8_t)(o >> 16);
}
__device__ __forceinline__ void put_dig(uint8_t *p, const Dig g, int nd)
{
    put_chars(p, g.a, nd < 4 ? nd : 4);
    if (nd > 4) put_chars(p + 4, g.b, nd - 4 < 4 ? nd - 4 : 4);
    if (nd > 8) put_chars(p + 8, g.c, nd - 8);
}
// One line whose digit counts n1, n2, n3 (wave-uniform) and name (<= 8 characters, in n0 / n1w) are the same in every lane;
// depth < 10000.  S: the digits of start (the caller's: converted, or the previous line's end).
__device__ __forceinline__ void put_line_uniform(uint8_t *p, const Dig S, const Dig E, uint32_t depth, int n1, int n2, int n3, uint32_t n0, uint32_t n1w,
                                                 int name_len)
{
    put_chars(p, n0, name_len < 4 ? name_len : 4);
    if (name_len > 4) put_chars(p + 4, n1w, name_len - 4);
    uint8_t *q = p + name_len;
    q[0] = '\t';
    put_dig(q + 1, S, n1);
    q += n1 + 1;
    q[0] = '\t';
    put_dig(q + 1, E, n2);
    q += n2 + 1;
    q[0] = '\t';
    put_chars(q + 1, ascii4(depth) >> (8 * (4 - n3)), n3);
    q[n3 + 1] = '\n';
}

// ---- lines of one layout, stored as aligned words (round 4) --------------------------------------------------------------------
// The byte-wise store of a line is ~27 ds_write_b8 and as many address adds.  With the layout known at COMPILE time (name length and
// the digit counts of start and end: ten instantiations cover chr1 .. chr22 / X / Y / M from position 10^4 to 10^9; the depth's
// 1 .. 4 digits close the line and stay a run-time value) the line is put together in registers with constant shifts, moved to its
// place inside the words it covers by one v_perm per word (selector from the lane's byte offset & 3), and stored as whole aligned
// words.  A word that two lines share is completed before the store: by the lane itself between its two lines, from the lane
// below (DPP wave shift) between lanes; only the wave's first and last line have an edge left, which goes byte by byte.
constexpr int kLineWords = 9;      // a line of up to 8 + 1 + 9 + 1 + 9 + 1 + 4 + 1 = 34 characters
// ORs the first NB characters of the left-justified field w0 w1 w2 (zero behind its last character) into the line at character POS
template <int POS, int NB>
__device__ __forceinline__ void line_or(uint32_t (&L)[kLineWords], uint32_t w0, uint32_t w1, uint32_t w2)
{
    const uint32_t w[3] = {w0, w1, w2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (4 * j < NB) {
            constexpr int dummy = 0;
            (void)dummy;
            const int at = POS + 4 * j, i = at >> 2, sh = at & 3;
            L[i] |= w[j] << (8 * sh);
            if (sh && i + 1 < kLineWords) L[i + 1] |= w[j] >> (8 * (4 - sh));
        }
    }
}
// name (NAME characters in n0 n1w) \t S (N1 digits) \t E (N2 digits) \t depth (n3 digits, < 10000) \n   -> L; the line's length
template <int NAME, int N1, int N2>
__device__ __forceinline__ int line_words(uint32_t (&L)[kLineWords], uint32_t n0, uint32_t n1w, const Dig S, const Dig E, uint32_t depth, int n3)
{
    constexpr int P1 = NAME, P2 = P1 + 1 + N1, P3 = P2 + 1 + N2;     // where the three tabs stand
#pragma unroll
    for (int i = 0; i < kLineWords; ++i) L[i] = 0;
    L[P1 >> 2] |= 9u << (8 * (P1 & 3)), L[P2 >> 2] |= 9u << (8 * (P2 & 3)), L[P3 >> 2] |= 9u << (8 * (P3 & 3));
    constexpr uint32_t m0 = NAME >= 4 ? 0xffffffffu : (1u << (8 * (NAME & 3))) - 1u;
    constexpr uint32_t m1 = NAME >= 8 ? 0xffffffffu : NAME > 4 ? (1u << (8 * (NAME & 3))) - 1u : 0u;
    line_or<0, NAME>(L, n0 & m0, n1w & m1, 0u);
    line_or<P1 + 1, N1>(L, S.a, S.b, S.c);
    line_or<P2 + 1, N2>(L, E.a, E.b, E.c);
    // the depth's digits, left-justified, and the newline behind them: five characters at the most
    const uint32_t d = ascii4(depth) >> (8 * (4 - n3));
    line_or<P3 + 1, 5>(L, n3 < 4 ? d | 10u << (8 * n3) : d, n3 < 4 ? 0u : 10u, 0u);
    return P3 + 2 + n3;
}
// One line's words at byte offset `at` of the staging buffer: T[i] = word i of ([at & 3 zero bytes] + line), c = the words it
// covers, the last of them incomplete when (at + len) & 3 != 0.
struct LinePlace {
    uint32_t T[kLineWords + 1];
    uint32_t a, tb;        // bytes of the first word that belong to the line before; bytes of the last word that are this line's (0: all)
    uint32_t tail;         // the last word when it is incomplete, else 0
};
__device__ __forceinline__ void line_place(LinePlace &pl, const uint32_t (&L)[kLineWords], uint32_t at, int len)
{
    pl.a = at & 3u, pl.tb = (at + (uint32_t)len) & 3u;
    const uint32_t sel = 0x03020100u + 0x01010101u * (4u - pl.a);       // bytes 4 - a .. 7 - a of {L[i], L[i - 1]}
    pl.T[0] = __builtin_amdgcn_perm(L[0], 0u, sel);
#pragma unroll
    for (int i = 1; i < kLineWords; ++i) pl.T[i] = __builtin_amdgcn_perm(L[i], L[i - 1], sel);
    pl.T[kLineWords] = __builtin_amdgcn_perm(0u, L[kLineWords - 1], sel);
}
// word `idx` of T for idx in [LO, LO + 2] (wave-uniform base, lane-varying idx): selects, no indexed registers
template <int LO>
__device__ __forceinline__ uint32_t line_pick(const uint32_t (&T)[kLineWords + 1], uint32_t idx)
{
    const uint32_t x0 = T[LO < kLineWords + 1 ? LO : kLineWords], x1 = T[LO + 1 < kLineWords + 1 ? LO + 1 : kLineWords],
                   x2 = T[LO + 2 < kLineWords + 1 ? LO + 2 : kLineWords];
    return idx == (uint32_t)LO ? x0 : idx == (uint32_t)LO + 1u ? x1 : x2;
}
// The two lines of a lane (line 0 at `at`, line 1 right behind it), both of the layout <NAME, N1, N2> in every lane of the wave;
// `part` is the wave's own staging area (its lane 0 starts at byte 0).  One line at a time, so that only one line's words are
// live: line 0 stores its whole words at once and keeps its first word when that is shared with the lane below (h0) and its
// last when that is incomplete (t0); line 1 completes t0 with its own first bytes, and its own incomplete last word with the
// h0 of the lane above (DPP wave shift; the wave's last lane: with zeros -- behind the wave's text, never copied out).
template <int NAME, int N1, int N2>
__device__ __forceinline__ void put_pair_words(uint8_t *part, uint32_t at, uint32_t n0, uint32_t n1w, const Dig S0, const Dig E0, uint32_t d0, int n30,
                                               const Dig S1, const Dig E1, uint32_t d1, int n31)
{
    constexpr int kMin = NAME + 3 + N1 + N2 + 2;                     // shortest line of the layout (one depth digit)
    constexpr int kLo = (kMin + 3) / 4 - 1;                          // index of a line's last word: >= kLo (a = 0, one digit), <= kLo + 2
    uint32_t L[kLineWords];
    uint32_t h0, t0;
    uint32_t at1;
    {
        LinePlace p0;
        const int len0 = line_words<NAME, N1, N2>(L, n0, n1w, S0, E0, d0, n30);
        line_place(p0, L, at, len0);
        at1 = at + (uint32_t)len0;
        const uint32_t c0 = (p0.a + (uint32_t)len0 + 3u) >> 2, n_w0 = c0 - (p0.tb ? 1u : 0u);      // words the line covers / stores whole
        t0 = p0.tb ? line_pick<kLo>(p0.T, c0 - 1u) : 0u;
        h0 = p0.a ? p0.T[0] : 0u;                                    // shared with the lane below: that lane stores it
        uint32_t *w0 = reinterpret_cast<uint32_t *>(part + (at & ~3u));
        if (p0.a == 0u) w0[0] = p0.T[0];
#pragma unroll
        for (int i = 1; i < kLineWords + 1; ++i) {
            if (i < kLo) w0[i] = p0.T[i];                            // (every lane's line reaches this far)
            else if ((uint32_t)i < n_w0 && i <= kLo + 2) w0[i] = p0.T[i];
        }
    }
    {
        LinePlace p1;
        const int len1 = line_words<NAME, N1, N2>(L, n0, n1w, S1, E1, d1, n31);
        line_place(p1, L, at1, len1);
        const uint32_t c1 = (p1.a + (uint32_t)len1 + 3u) >> 2, n_w1 = c1 - (p1.tb ? 1u : 0u);
        const uint32_t above = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0x130, 0xf, 0xf, false);   // wave_shl:1: lane l + 1's h0 (lane 63: 0)
        uint32_t *w1 = reinterpret_cast<uint32_t *>(part + (at1 & ~3u));
        w1[0] = p1.T[0] | t0;
#pragma unroll
        for (int i = 1; i < kLineWords + 1; ++i) {
            if (i < kLo) w1[i] = p1.T[i];
            else if ((uint32_t)i < n_w1 && i <= kLo + 2) w1[i] = p1.T[i];
        }
        if (p1.tb) w1[c1 - 1u] = line_pick<kLo>(p1.T, c1 - 1u) | above;
    }
}

struct FmtName {
    uint32_t w[16];       // 64 characters
};

// inclusive prefix sum over the lanes of a wave (DPP: shifts inside rows of 16, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_u32(uint32_t v)
{
#define HPN_DPP_ADD(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false)
    HPN_DPP_ADD(0x111, 0xf);
    HPN_DPP_ADD(0x112, 0xf);
    HPN_DPP_ADD(0x114, 0xf);
    HPN_DPP_ADD(0x118, 0xf);
    HPN_DPP_ADD(0x142, 0xa);
    HPN_DPP_ADD(0x143, 0xc);
#undef HPN_DPP_ADD
    return v;
}

// Three kernels (round 4; one kernel with a look-back chain before): the lines' SIZES per (tile, piece, wave) and per tile, the
// tiles' offsets (one workgroup), then the text -- every wave of the third kernel knows where its 128 lines go and needs no other
// wave: no ticket, no chain, no workgroup barrier.  Measured on the one-kernel form (timing-only builds, profiles/r04/
// bedgraph_breakdown.txt): sizes 0.30 + chain wait 0.12 + LDS traffic 0.10 + formatting 0.14 + stores 0.31 = the kernel's 0.97 ms
// -- the parts ADD UP, a tile's phases wait for one another and four to six workgroups per CU do not hide that.
typedef int32_t i32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void bg_load_pair(const hpn_run *__restrict__ runs, uint64_t n_runs, uint64_t r0, i32x3 (&v)[kFmtPer], uint32_t (&ln)[kFmtPer])
{
#pragma unroll
    for (int k = 0; k < kFmtPer; ++k) {
        v[k] = i32x3{0, 0, 0};
        ln[k] = r0 + k < n_runs ? 1u : 0u;                     // (a line at all; its size when the values are there: bg_size_pair)
        if (ln[k]) v[k] = *reinterpret_cast<const i32x3 *>(&runs[r0 + k]);
    }
}
__device__ __forceinline__ uint32_t bg_size_pair(const i32x3 (&v)[kFmtPer], uint32_t (&ln)[kFmtPer], int name_len, const u64 *dig)
{
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kFmtPer; ++k) {
        if (ln[k]) ln[k] = line_info(v[k][0], v[k][1], v[k][2], name_len, dig);
        mine += (ln[k] & 0xffffu);
    }
    return mine;
}
constexpr int kFmtWaves = kFmtThreads / kWave;
static_assert(kFmtWaves == 4, "a piece's four wave totals travel as one 16-byte word");

// wave_tot[(tile * kFmtSubs + piece) * 4 + wave] = bytes of that wave's 128 lines; tile_tot[tile] = bytes of the tile
__global__ __launch_bounds__(kFmtThreads) void k_bedgraph_sizes(const hpn_run *__restrict__ runs, uint64_t n_runs, int name_len, uint64_t n_tiles,
                                                                uint32_t *__restrict__ wave_tot, u64 *__restrict__ tile_tot)
{
    __shared__ u64 s_dig[33];
    __shared__ uint32_t s_sum[kFmtWaves];
    const int tid = threadIdx.x;
    dec_digits_fill(s_dig, tid);
    __syncthreads();
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t sum = 0;
#pragma unroll
        for (int sb = 0; sb < kFmtSubs; ++sb) {
            i32x3 v[kFmtPer];
            uint32_t ln[kFmtPer];
            bg_load_pair(runs, n_runs, tile * kFmtTile + (uint64_t)sb * kFmtSub + (uint64_t)tid * kFmtPer, v, ln);
            const uint32_t upto = wave_incl_u32(bg_size_pair(v, ln, name_len, s_dig));
            if (lane_id() == kWave - 1) wave_tot[(tile * kFmtSubs + (uint64_t)sb) * kFmtWaves + (uint64_t)wave_id()] = upto;
            sum += upto;                                       // (lane 63's is the wave's)
        }
        if (lane_id() == kWave - 1) s_sum[wave_id()] = sum;
        __syncthreads();
        if (tid == 0) tile_tot[tile] = (u64)s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        __syncthreads();
    }
}

// tile_base[t] = bytes in front of tile t (in place of tile_tot[t]); *total = all bytes.  One workgroup: a chromosome has ~10^4 tiles.
__global__ __launch_bounds__(1024) void k_bedgraph_offsets(u64 *__restrict__ tile_tot, uint64_t n_tiles, u64 *__restrict__ total)
{
    __shared__ u64 s_wave[16];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t i0 = 0; i0 < n_tiles; i0 += 1024u) {
        const uint64_t i = i0 + threadIdx.x;
        const u64 v = i < n_tiles ? tile_tot[i] : 0;
        u64 inc = v;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const u64 t = __shfl_up(inc, o, kWave);
            if (lane_id() >= o) inc += t;
        }
        if (lane_id() == kWave - 1) s_wave[wave_id()] = inc;
        __syncthreads();
        u64 before = s_carry;
        for (int w = 0; w < wave_id(); ++w) before += s_wave[w];
        if (i < n_tiles) tile_tot[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

#ifndef HPN_BG_EU
#define HPN_BG_EU 6
#endif
#ifndef HPN_BG_VGPR
#define HPN_BG_VGPR 80
#endif
__global__ __launch_bounds__(kFmtThreads) __attribute__((amdgpu_waves_per_eu(HPN_BG_EU, 8), amdgpu_num_vgpr(HPN_BG_VGPR)))
void k_bedgraph_text(const hpn_run *__restrict__ runs, uint64_t n_runs, FmtName name, int name_len,
                                                               const uint8_t *__restrict__ long_name, uint8_t *__restrict__ out,
                                                               const uint32_t *__restrict__ wave_tot, const u64 *__restrict__ tile_base)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_text[kFmtLds];
    __shared__ uint32_t s_name[16];
    __shared__ u64 s_dig[33];
    const int tid = threadIdx.x;
    if (tid < 16) s_name[tid] = name.w[tid];
    dec_digits_fill(s_dig, tid);
    __syncthreads();
    const uint64_t tile = blockIdx.x;
    const uint32_t n0 = name.w[0], n1 = name.w[1];
    const uint8_t *nm = name_len <= 64 ? reinterpret_cast<const uint8_t *>(s_name) : long_name;
    auto load_pair = [&](int sb, i32x3 (&v)[kFmtPer], uint32_t (&ln)[kFmtPer]) {
        bg_load_pair(runs, n_runs, tile * kFmtTile + (uint64_t)sb * kFmtSub + (uint64_t)tid * kFmtPer, v, ln);
    };
    auto size_pair = [&](const i32x3 (&v)[kFmtPer], uint32_t (&ln)[kFmtPer]) { return bg_size_pair(v, ln, name_len, s_dig); };
    const u32 *const tots = reinterpret_cast<const u32 *>(wave_tot) + tile * kFmtSubs;   // (the four wave totals of a piece: one word)
    i32x3 cur[kFmtPer], nxt[kFmtPer];
    uint32_t cur_ln[kFmtPer], nxt_ln[kFmtPer];
    load_pair(0, cur, cur_ln);
    u64 piece_base = tile_base[tile];                         // byte offset of the piece being written
#pragma unroll 1
    for (int sb = 0; sb < kFmtSubs; ++sb) {
        if (sb + 1 < kFmtSubs) load_pair(sb + 1, nxt, nxt_ln);   // the next piece's runs are on their way while this one is written
        int32_t rs[kFmtPer], re[kFmtPer], rd[kFmtPer];
        uint32_t len[kFmtPer];
#pragma unroll
        for (int k = 0; k < kFmtPer; ++k) rs[k] = cur[k][0], re[k] = cur[k][1], rd[k] = cur[k][2], len[k] = cur_ln[k];
        const uint32_t mine = size_pair(cur, len), wex = wave_incl_u32(mine) - mine;
        const u32 wt = tots[sb];
        u64 before = 0, piece = 0;
#pragma unroll
        for (int w = 0; w < kFmtWaves; ++w) {
            if (w < wave_id()) before += wt[w];
            piece += wt[w];
        }
        // Every wave stages ITS 128 lines in its own part of the buffer and copies them out itself: no workgroup barrier in this
        // loop (round 4; two per piece before), the waves drift apart and one's copy runs beside another's formatting.
        const uint32_t wave_bytes = wave_id() == 0 ? wt[0] : wave_id() == 1 ? wt[1] : wave_id() == 2 ? wt[2] : wt[3];
        uint8_t *const my = s_text + (uint32_t)wave_id() * (uint32_t)kFmtWaveLds;
        uint32_t at = wex;                                    // this lane's first line inside the wave's part
        // (Round 3 tried lines built in registers -- digits shifted to the front with v_alignbyte, fields written with two overlapping
        // exact-length 8- / 4- / 2-byte stores at their own byte offsets: 270 M instead of 313 M vector instructions per chr1, but the
        // unaligned stores stall in LDS (SQ_LDS_UNALIGNED_STALL 394 M cycles: 1.15 ms against 1.03) and are worse still straight to
        // memory (2.27 ms): profiles/r03/bedgraph_text_variants.txt.  Round 4: whole ALIGNED words, put_pair_words.)
        if (name_len <= kFmtMaxName && wave_bytes + 4u <= (uint32_t)kFmtWaveLds) {   // staged: build in LDS, copy out in 16-byte pieces
            Dig prevE{0, 0, 0};
            bool have_prev = false;                           // prevE holds the digits of line k - 1's end (wave-uniform)
            // both lines of every lane of ONE layout the word path is compiled for?  (chr1 .. chrM names, 5 .. 9 digits)
            bool by_words = false;
#ifdef DIAG_BG_NOFMT
            by_words = true;                                  // (timing only: nothing is formatted; the part is written so that the copy stays)
            for (uint32_t o = (uint32_t)lane_id() * 16u; o < wave_bytes + 32u; o += (uint32_t)kWave * 16u)
                *reinterpret_cast<u32 *>(my + o) = u32{(uint32_t)rs[0], (uint32_t)re[0], (uint32_t)rd[0], o};
#elif !defined(HPN_BG_BYTES)
            {
                static_assert(kFmtPer == 2, "put_pair_words takes a lane's two lines");
                const uint32_t pk = len[0] >> 16, pk0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk);
                const int n1d = (int)(pk0 & 15u), n2d = (int)((pk0 >> 4) & 15u);
                const bool fits = (name_len == 4 || name_len == 5) && n1d == n2d && n1d >= 5 && n1d <= 9 && pk0 != 0;
                if (fits && __ballot((len[0] >> 16 & 0xffu) != (pk0 & 0xffu) || (len[1] >> 16 & 0xffu) != (pk0 & 0xffu) ||
                                     (uint32_t)rd[0] >= 10000u || (uint32_t)rd[1] >= 10000u || (rs[0] | re[0] | rs[1] | re[1]) < 0) == 0) {
                    by_words = true;
                    const int n30 = (int)(len[0] >> 24) & 15, n31 = (int)(len[1] >> 24) & 15;   // (lane-varying: the depth's digits)
                    const Dig S0 = dec_left((uint32_t)rs[0], n1d), E0 = dec_left((uint32_t)re[0], n1d), E1 = dec_left((uint32_t)re[1], n1d);
                    Dig S1 = E0;                               // adjacent runs: line 1 starts where line 0 ends
                    if (__ballot(rs[1] != re[0])) S1 = dec_left((uint32_t)rs[1], n1d);
#define HPN_BG_CASE(NM, ND) \
    case NM * 16 + ND: put_pair_words<NM, ND, ND>(my, at, n0, n1, S0, E0, (uint32_t)rd[0], n30, S1, E1, (uint32_t)rd[1], n31); break;
                    switch (name_len * 16 + n1d) {
                        HPN_BG_CASE(4, 5) HPN_BG_CASE(4, 6) HPN_BG_CASE(4, 7) HPN_BG_CASE(4, 8) HPN_BG_CASE(4, 9)
                        HPN_BG_CASE(5, 5) HPN_BG_CASE(5, 6) HPN_BG_CASE(5, 7) HPN_BG_CASE(5, 8) HPN_BG_CASE(5, 9)
                    }
#undef HPN_BG_CASE
                }
            }
#endif
            if (!by_words) {
#pragma unroll
                for (int k = 0; k < kFmtPer; ++k) {
                    // one layout for the whole wave?  (digit counts, a line in every lane, nothing negative, depth of at most four digits)
                    const uint32_t pk = len[k] >> 16, pk0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk);
                    const bool same = name_len <= 8 && pk0 != 0 &&
                                      __ballot(pk != pk0 || (uint32_t)rd[k] >= 10000u || (rs[k] | re[k]) < 0) == 0;
                    if (same) {
                        const int n1d = (int)(pk0 & 15u), n2d = (int)((pk0 >> 4) & 15u), n3d = (int)((pk0 >> 8) & 15u);
                        const Dig E = dec_left((uint32_t)re[k], n2d);
                        Dig S = prevE;                                               // adjacent runs: the line before ended where this one starts
                        if (!have_prev || __ballot(rs[k] != re[k > 0 ? k - 1 : 0])) S = dec_left((uint32_t)rs[k], n1d);
                        put_line_uniform(my + at, S, E, (uint32_t)rd[k], n1d, n2d, n3d, n0, n1, name_len);
                        prevE = E, have_prev = true;
                    } else {
                        if (len[k]) put_line(my + at, rs[k], re[k], rd[k], len[k], n0, n1, nm, name_len);
                        have_prev = false;
                    }
                    at += (len[k] & 0xffffu);
                }
            }
            // (what a lane copies out is what OTHER lanes of the wave wrote: the wave's LDS operations run in order, the fence keeps
            // the compiler from moving them)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (16-byte stores at whatever byte the wave's text begins: placing the text in LDS as it lies in memory modulo 16, so that
            // the stores are aligned, was measured and is slower -- 1.02 against 0.97 ms: the stores run at memory speed either way)
            uint8_t *dst = out + piece_base + before;
#ifdef DIAG_BG_NOCOPY
            {                                                 // (timing only: the part is read, nothing is stored unless its words XOR to a magic value)
                uint32_t x = 0;
                for (uint32_t o = (uint32_t)lane_id() * 16u; o < wave_bytes; o += (uint32_t)kWave * 16u) {
                    const u32 v = *reinterpret_cast<const u32 *>(my + o);
                    x ^= v[0] ^ v[1] ^ v[2] ^ v[3];
                }
                if (x == 0x12345678u) dst[lane_id()] = 1;
            }
            if (false)
#endif
            for (uint32_t o = (uint32_t)lane_id() * 16u; o < wave_bytes; o += (uint32_t)kWave * 16u) {
                if (o + 16u <= wave_bytes) {
                    const u32 v = *reinterpret_cast<const u32 *>(my + o);
                    __builtin_memcpy(dst + o, &v, 16);
                } else {
                    for (uint32_t b2 = o; b2 < wave_bytes; ++b2) dst[b2] = my[b2];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the part is rewritten by the wave's next piece
            __builtin_amdgcn_wave_barrier();
        } else {                                              // a very long target name (or numbers): straight to memory
            at += (uint32_t)before;
#pragma unroll
            for (int k = 0; k < kFmtPer; ++k) {
                if (len[k]) put_line(out + piece_base + at, rs[k], re[k], rd[k], len[k], n0, n1, nm, name_len);
                at += (len[k] & 0xffffu);
            }
        }
        piece_base += piece;
#pragma unroll
        for (int k = 0; k < kFmtPer; ++k) cur[k] = nxt[k], cur_ln[k] = nxt_ln[k];
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
uint64_t depth_tiles(uint64_t slots) { return (slots + kTile - 1) / kTile; }
static uint64_t scan_tiles(uint64_t slots) { return (slots + kDsGroup - 1) / kDsGroup; }

// bytes of the per-target index: head | first_hi | first_lo | written | need
size_t depth_index_bytes(uint64_t slots)
{
    const uint64_t nt = depth_tiles(slots);
    return sizeof(uint32_t) * (kHdWords + 2 * (nt + 1) + 2 * nt);
}

static TileIndex tile_index(void *ws, uint64_t slots)
{
    const uint32_t nt = (uint32_t)depth_tiles(slots);
    uint32_t *w = (uint32_t *)ws;
    return TileIndex{w, w + kHdWords, w + kHdWords + (nt + 1), w + kHdWords + 2 * (nt + 1), w + kHdWords + 2 * (nt + 1) + nt, nt};
}

// hpn_depth_begin: no tile has been written, none is needed
hipError_t depth_index_reset(void *ws, uint64_t slots, hipStream_t st)
{
    const TileIndex ix = tile_index(ws, slots);
    return hipMemsetAsync(ix.written, 0, 2 * sizeof(uint32_t) * (size_t)ix.ntiles, st);
}

const uint32_t *depth_written(void *ws, uint64_t slots) { return tile_index(ws, slots).written; }

// the sweep's state (SweepState): ctl words, then one chain entry per tile
size_t depth_sweep_bytes(uint64_t slots) { return kSwWords * sizeof(uint32_t) + kStStride * depth_tiles(slots) * sizeof(u64); }
static SweepState sweep_state(void *sws) { return SweepState{(uint32_t *)sws, (u64 *)((uint32_t *)sws + kSwWords)}; }

// hpn_depth_begin: nothing swept, no chain entry published; enabled = 0 keeps every batch on the two-pass route
hipError_t depth_sweep_reset(void *sws, uint64_t slots, bool enabled, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(sws, 0, depth_sweep_bytes(slots), st);
    if (e != hipSuccess || !enabled) return e;
    static const uint32_t one = 1u;
    return hipMemcpyAsync((uint32_t *)sws + kSwEnabled, &one, sizeof one, hipMemcpyHostToDevice, st);
}

// what the sweep writes besides the chain: runs (all of the target's, capacity runs_cap) and, when the window size is
// known while the records come (W != 0), the window sums of the swept part
struct SweepOut {
    hpn_run *runs;
    uint64_t runs_cap;
    u64 *win_sum;
    uint32_t target_len, W;
};

static void launch_oplen(const SoaRecs &recs, uint64_t n, const TileIndex &ix, const SweepState &sw, uint64_t cap, hipStream_t st)
{
    const uint64_t want = (n / 4 + 255) / 256 + 1;       // ~1.25 operations per record
    hipLaunchKernelGGL(k_depth_oplen, dim3((unsigned)(want < cap * 2 ? want : cap * 2)), dim3(256), 0, st, recs.cigar_off, recs.cigar, n, ix, sw);
}
static void launch_oplen(const RawRecs &, uint64_t, const TileIndex &, const SweepState &, uint64_t, hipStream_t) {}

template <typename Recs>
static hipError_t depth_add(const Recs &recs, uint64_t n, int32_t tid, uint32_t flag_mask, int32_t *diff, uint64_t slots, void *ws,
                            void *sws, const SweepOut &so, uint32_t *bad, int n_cu, hipStream_t st)
{
    if (n == 0 || tid < 0) return hipSuccess;    // tid < 0 never matches (bam2depth.c:90)
    const TileIndex ix = tile_index(ws, slots);
    const SweepState sw = sweep_state(sws);
    hipError_t e = hipMemsetAsync(ix.head, 0, kHdWords * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    const uint64_t cap = (uint64_t)n_cu * 8;
    const uint64_t wi = (n + kIdxThreads - 1) / kIdxThreads, wf = (n + kFarThreads - 1) / kFarThreads;
    hipLaunchKernelGGL(k_depth_index<Recs>, dim3((unsigned)(wi < cap * 4 ? wi : cap * 4)), dim3(kIdxThreads), 0, st, recs, n, tid, ix, sw);
    launch_oplen(recs, n, ix, sw, cap, st);
    // the sweep takes the batch (sorted, no far breakpoint, not behind the frontier) or leaves it to the three kernels behind it
    const DepthOut out{so.runs, so.runs_cap, nullptr, so.W ? so.win_sum : nullptr};
    hipLaunchKernelGGL(k_depth_sweep<Recs>, dim3(ix.ntiles), dim3(kSwThreads), 0, st, recs, tid, flag_mask, diff, slots, ix, sw, so.target_len,
                       so.W, out, bad);
    hipLaunchKernelGGL(k_depth_tiles<Recs>, dim3(ix.ntiles), dim3(kTileThreads), 0, st, recs, tid, flag_mask, diff, slots, ix, sw, bad);
    hipLaunchKernelGGL(k_depth_fill, dim3(ix.ntiles), dim3(256), 0, st, diff, slots, ix, sw);
    hipLaunchKernelGGL(k_depth_far<Recs>, dim3((unsigned)(wf < cap ? wf : cap)), dim3(kFarThreads), 0, st, recs, n, tid, flag_mask, diff,
                       slots, ix, sw, bad);
    hipLaunchKernelGGL(k_depth_commit, dim3(1), dim3(64), 0, st, ix, sw);
    return hipGetLastError();
}

hipError_t launch_depth_add(const int32_t *tid_a, const int32_t *pos, const uint32_t *flag, const uint32_t *cigar_off,
                            const uint32_t *cigar, uint64_t n, int32_t tid, uint32_t flag_mask, int32_t *diff, uint64_t slots,
                            void *ws, void *sws, hpn_run *runs, uint64_t runs_cap, u64 *win_sw, uint32_t target_len, uint32_t W,
                            uint32_t *bad, int n_cu, hipStream_t st)
{
    return depth_add(SoaRecs{tid_a, pos, flag, cigar_off, cigar}, n, tid, flag_mask, diff, slots, ws, sws,
                     SweepOut{runs, runs_cap, win_sw, target_len, W}, bad, n_cu, st);
}

hipError_t launch_depth_add_raw(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, int32_t tid, uint32_t flag_mask, int32_t *diff,
                                uint64_t slots, void *ws, void *sws, hpn_run *runs, uint64_t runs_cap, u64 *win_sw, uint32_t target_len,
                                uint32_t W, uint32_t *bad, int n_cu, hipStream_t st)
{
    return depth_add(RawRecs{raw, rec_off}, n, tid, flag_mask, diff, slots, ws, sws, SweepOut{runs, runs_cap, win_sw, target_len, W}, bad,
                     n_cu, st);
}

// window sums of all `n_runs` runs for window size W, added into win_sum
hipError_t launch_win_from_runs(const hpn_run *runs, uint64_t n_runs, uint32_t target_len, uint32_t W, u64 *win_sum, int n_cu, hipStream_t st)
{
    if (n_runs == 0) return hipSuccess;
    const uint64_t want = (n_runs + 255) / 256, cap = (uint64_t)n_cu * 8;
    hipLaunchKernelGGL(k_win_from_runs, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, runs, n_runs, target_len, W, win_sum);
    return hipGetLastError();
}

// ws: [0] ticket, [1] err (uint32 each), then u64 n_runs, then status[2 * tiles]
size_t depth_scan_bytes(uint64_t slots) { return 16 + kStStride * scan_tiles(slots) * sizeof(u64); }

// sws: the sweep's state -- the scan starts at its frontier and continues its chain (a frontier of 0: the whole target)
hipError_t launch_depth_scan(const int32_t *diff, const uint32_t *written, uint64_t slots, uint32_t target_len, uint32_t W, hpn_run *runs,
                             uint64_t runs_cap, u64 *win_sum, void *ws, void *sws, hipStream_t st)
{
    const uint64_t tiles = scan_tiles(slots);
    hipError_t e = hipMemsetAsync(ws, 0, depth_scan_bytes(slots), st);
    if (e != hipSuccess) return e;
    uint32_t *ticket = (uint32_t *)ws;
    u64 *n_runs = (u64 *)ws + 1;
    u64 *status = (u64 *)ws + 2;
    DepthOut out{runs, runs_cap, n_runs, win_sum};
    hipLaunchKernelGGL(k_depth_scan, dim3((unsigned)tiles), dim3(kDsThreads), 0, st, diff, written, slots, target_len, W, out, status,
                       ticket, ticket + 1, sweep_state(sws));
    return hipGetLastError();
}

// bedGraph text of `n_runs` runs into `out` (device; upper bound of the size: bedgraph_text_bound).
// ws: [0] ticket, [1] err (uint32 each), then u64 total, then status[tiles]
uint64_t bedgraph_text_bound(uint64_t n_runs, int name_len) { return n_runs * (uint64_t)(name_len + 34 + 3) + 64; }
static uint64_t bedgraph_tiles(uint64_t n_runs) { return (n_runs + kFmtTile - 1) / kFmtTile; }
// ws: [0] unused, [1] err (uint32 each), then u64 total, then tile_base[tiles] (u64), then wave_tot[tiles * pieces * 4] (uint32)
size_t bedgraph_ws_bytes(uint64_t n_runs)
{
    const uint64_t nt = bedgraph_tiles(n_runs);
    return 16 + (nt + 1) * sizeof(u64) + nt * kFmtSubs * kFmtWaves * sizeof(uint32_t) + 16;
}

hipError_t launch_bedgraph_text(const hpn_run *runs, uint64_t n_runs, const char *name, int name_len, const uint8_t *d_long_name,
                                uint8_t *out, void *ws, int n_cu, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(ws, 0, 16, st);
    if (e != hipSuccess || n_runs == 0) return e;
    FmtName nm;
    memset(&nm, 0, sizeof nm);
    memcpy(nm.w, name, (size_t)(name_len < 64 ? name_len : 64));   // (a longer name: its head here, all of it in d_long_name)
    const uint64_t nt = bedgraph_tiles(n_runs);
    u64 *tile_base = (u64 *)ws + 2;
    uint32_t *wave_tot = (uint32_t *)(tile_base + ((nt + 2) & ~(uint64_t)1));      // (16-byte aligned: read as whole words)
    // (a workgroup per tile: 0.167 -> 0.135 ms per 67 M lines against eight workgroups per CU looping over the tiles -- the pass is
    // short and latency-bound, more of it in flight is all it wants; the runs of several pieces asked for together: no gain)
    (void)n_cu;
    hipLaunchKernelGGL(k_bedgraph_sizes, dim3((unsigned)nt), dim3(kFmtThreads), 0, st, runs, n_runs, name_len, nt, wave_tot, tile_base);
    hipLaunchKernelGGL(k_bedgraph_offsets, dim3(1), dim3(1024), 0, st, tile_base, nt, (u64 *)ws + 1);
    hipLaunchKernelGGL(k_bedgraph_text, dim3((unsigned)nt), dim3(kFmtThreads), 0, st, runs, n_runs, nm, name_len, d_long_name, out, wave_tot, tile_base);
    return hipGetLastError();
}

uint32_t depth_tile_size() { return (uint32_t)kTile; }

}  // namespace hpn
