// bam_window.hip -- gfx950 kernel behind hpn_window_* (include/hpngs.h).
//
// Replaces fetch_func + cal_GC of bam_sliding_count (reference
// bam_sliding_count.c:84-124): per record with tid >= 0 and !(flag & 4)
//     slot = win_off[tid] + (unsigned short)(pos / W)
//     bins[slot] += 1;  gc[slot] += #{4-bit codes == 2 (C) or == 4 (G)};  len[slot] += l_qseq
// All integer here; the float32 arithmetic of calc_winGC (:126-138) is replayed
// on the host in reference order (csrc/host/report.cpp).
//
// Waves work on their own: one lane per record for the fields; the packed sequences are read by the wave together
// (eight lanes per record, 16-byte pieces: gc_of_wave_records) and counted with nibble-wise SWAR; the sums of the
// window a wave is in stay in registers and go to memory with three atomics when the window changes (records of a
// coordinate-sorted BAM that are near each other fall into the same window).  No LDS table, no barrier.
// Bound: HBM read of 16 B + ceil(l_qseq/2) B per record.
#include "common.hpp"

namespace hpn {

constexpr int kWinThreads = 256;

// number of nibbles of w equal to 2 (C) or 4 (G): bits 8 and 1 clear, exactly one of bits 2 and 4 set
__device__ __forceinline__ uint32_t gc_nibbles(uint32_t w)
{
    return __builtin_popcount(((w >> 1) ^ (w >> 2)) & ~(w >> 3) & ~w & 0x11111111u);
}
// The same with the verdict of each nibble at its bit 2 and the nibbles to count chosen by `keep` (a subset of 0x44444444:
// masking the verdicts instead of the input costs nothing), added to acc.  Seven instructions per word: shift; shift-or
// (bits 0 and 3 meet at bit 2); shift; xor (bits 1 and 2 meet at bit 2); and-not; and; count-and-add.
__device__ __forceinline__ uint32_t gc_nibbles_at2(uint32_t w, uint32_t keep, uint32_t acc)
{
    const uint32_t other = (w << 2) | (w >> 1);            // bit 2 of a nibble: its bit 0 | its bit 3
    const uint32_t one = (w << 1) ^ w;                     // bit 2 of a nibble: its bit 1 ^ its bit 2
    return acc + (uint32_t)__builtin_popcount(one & ~other & keep);
}

// GC count of the first `nbytes` (1..16) bytes of a 16-byte piece of a packed sequence (base i = high nibble of byte
// i/2 for even i, bam1_seqi, bam.h:260); clip: the piece ends the sequence of an odd-length read, whose last low
// nibble is padding, not a base.
__device__ __forceinline__ uint32_t gc_of_piece(const u32 q, int nbytes, bool clip)
{
    uint32_t gc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int hi = min(max(nbytes - 4 * j, 0), 4);                       // bytes of word j that count
        uint32_t m = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
        if (clip && hi > 0 && nbytes - 4 * j <= 4) m &= ~(0xfu << (8 * (hi - 1)));
        gc += gc_nibbles(q[j] & m);
    }
    return gc;
}

// 16 bytes from byte offset a of seq4 (any alignment); nothing at or beyond offset lim is touched (the last pieces of a
// batch: the caller's buffer may end with its last sequence byte)
__device__ __forceinline__ u32 load16u(const uint8_t *__restrict__ seq4, uint64_t a, uint64_t lim)
{
    u32 v;
    if (a + 16 <= lim) {
        __builtin_memcpy(&v, seq4 + a, 16);
    } else {
        v = u32{0, 0, 0, 0};
        for (int k = 0; k < 16 && a + k < lim; ++k) v[k >> 2] |= (uint32_t)seq4[a + k] << (8 * (k & 3));
    }
    return v;
}

// GC counts of 64 records, one per lane (s = byte offset of the packed sequence in seq4, l_qseq <= 0: none), computed by
// the wave together: in step t the eight lanes of group g read the sequence of record 8 t + g, lane i of the group its
// bytes [16 i, 16 i + 16) (unaligned 16-byte loads: only the last piece of a record needs a mask), so a wave
// instruction reads eight neighbouring records = one contiguous span.  One lane per record reads 64 scattered
// segments per instruction and spent ~10x the instructions on masks (0.26 of the HBM peak).  Eight steps, all their
// loads in flight together when no record of the wave is longer than 128 bytes (256 bases); longer reads loop.
// what a lane does in the passes of one read length (kept across passes: the division and the masks are made once per length)
struct GcLayout {
    int lq = -1, rps = 0, steps = 0, pg = 0, psub = 0;
    uint32_t m[4] = {0, 0, 0, 0};
};

__device__ __forceinline__ uint32_t gc_of_wave_records(const uint8_t *__restrict__ seq4, uint64_t s, int32_t l_qseq, uint64_t lim,
                                                       uint32_t *wave_gc /* 64 LDS words of this wave */, GcLayout &lay)
{
    const int lane = lane_id(), sub = lane & 7, grp = lane >> 3;
    const int my_nb = l_qseq > 0 ? (l_qseq + 1) >> 1 : 0;
    uint32_t mine = 0;
    const u64 has_seq = __ballot(l_qseq > 0);
    if (!has_seq) return 0;
    const int lq0 = __shfl(l_qseq, __builtin_ctzll(has_seq), kWave);   // records without a sequence ride along, their sum is dropped
    // byte offsets relative to the wave's first sequence: one 32-bit shuffle per step (records without a sequence read
    // the first one's; 64 neighbouring records 2 GiB apart do not happen -- if they do, the generic paths take the wave)
    const uint64_t s0 = __shfl((u64)s, __builtin_ctzll(has_seq), kWave);
    const bool rel_ok = l_qseq <= 0 || (s >= s0 && s - s0 < (1ull << 31) - (1ull << 16));
    const uint32_t srel = l_qseq > 0 ? (uint32_t)(s - s0) : 0u;
    if (lq0 <= 256 && __ballot((l_qseq > 0 && l_qseq != lq0) || !rel_ok) == 0) {
        // every record of the wave has the same length (the normal case): P = pieces per record lanes serve one record, so
        // floor(64 / P) records are read per step (12 at 150 bases: 60 of 64 lanes busy, 6 steps; eight lanes per record
        // kept 5 of 8 busy over 8 steps), and which bytes of its piece a lane counts depends on the lane only: the masks are
        // built once.  A record's pieces are summed with one LDS add per lane into the wave's own 64 words.
        if (lq0 != lay.lq) {               // a new read length: lane -> (record of the step, piece) and the byte masks
            const int nb = (lq0 + 1) >> 1, P = (nb + 15) >> 4;                // 1..8, the same in every lane
            lay.lq = lq0, lay.rps = kWave / P, lay.steps = (kWave + lay.rps - 1) / lay.rps;
            lay.pg = lane / P, lay.psub = lane - lay.pg * P;
            const int rem = nb - 16 * lay.psub;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hi = min(max(rem - 4 * j, 0), 4);
                lay.m[j] = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
                if ((lq0 & 1) && hi > 0 && rem - 4 * j <= 4) lay.m[j] &= ~(0xfu << (8 * (hi - 1)));
                lay.m[j] &= 0x44444444u;                   // the nibbles of word j that are bases of the read, as gc_nibbles_at2 wants them
            }
        }
        const int rps = lay.rps, steps = lay.steps, pg = lay.pg, psub = lay.psub;
        const bool used = pg < rps;
        const uint32_t *m = lay.m;
        uint32_t *acc = wave_gc + lane;
        __hip_atomic_store(acc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // The pieces come through a buffer descriptor that starts at the wave's first sequence and ends where seq4 may no
        // longer be read: address = descriptor + 32-bit lane offset (no 64-bit arithmetic, no compare against the end per
        // load -- the hardware drops what lies beyond).  A piece that STRADDLES the end would be dropped whole
        // (scripts/micro/buffer_range.hip): the batch's last piece, taken byte by byte below.
        const uint64_t room = lim > s0 ? lim - s0 : 0;
        const uint32_t nrec = room > 0x7fffffffull ? 0x7fffffffu : (uint32_t)room;   // (a wave's records lie within 2^31 - 2^16 bytes: rel_ok)
        const uint64_t b0 = (uint64_t)(uintptr_t)seq4 + s0;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b0)),
            0, (int)__builtin_amdgcn_readfirstlane(nrec), 0x00020000);
        u32 q[8];
        int rec[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            rec[t] = t * rps + pg;
            const bool on = t < steps && used && rec[t] < kWave;
            const uint32_t at = __shfl(srel, on ? rec[t] : 0, kWave) + 16u * (uint32_t)psub;
#ifdef DIAG_NOSEQ
            q[t] = u32{at, at * 3u, at * 5u, at * 7u};
#else
            q[t] = __builtin_bit_cast(u32, __builtin_amdgcn_raw_buffer_load_b128(rsrc, on ? at : 0x80000000u, 0, 0));   // an offset beyond any descriptor: no load
#endif
            if (__ballot(on && at < nrec && at + 16u > nrec)) {
                if (on && at < nrec && at + 16u > nrec) q[t] = load16u(seq4, s0 + at, lim);
            }
            if (!on) rec[t] = -1;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= steps) break;
            const uint32_t g = gc_nibbles_at2(q[t][0], m[0], gc_nibbles_at2(q[t][1], m[1], gc_nibbles_at2(q[t][2], m[2], gc_nibbles_at2(q[t][3], m[3], 0u))));
            if (rec[t] >= 0) __hip_atomic_fetch_add(wave_gc + rec[t], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                // the wave's own LDS operations: in order
        mine = __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (__ballot(my_nb > 128) == 0) {
        u32 q[8];
        int rem[8];
        bool clip[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int src = t * 8 + grp;
            const uint64_t st = __shfl((u64)s, src, kWave);
            const int lq = __shfl(l_qseq, src, kWave), nb = lq > 0 ? (lq + 1) >> 1 : 0;
            rem[t] = nb - 16 * sub;                                         // bytes of the record from this piece on
            clip[t] = (lq & 1) && rem[t] <= 16;
            q[t] = u32{0, 0, 0, 0};
            if (rem[t] > 0) q[t] = load16u(seq4, st + 16u * sub, lim);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint32_t g = rem[t] > 0 ? gc_of_piece(q[t], min(rem[t], 16), clip[t]) : 0u;
            g += __shfl_xor(g, 1, kWave), g += __shfl_xor(g, 2, kWave), g += __shfl_xor(g, 4, kWave);
            const uint32_t r = __shfl(g, sub * 8 + t, kWave);   // lane 8 t + g takes group g's sum of step t
            if (grp == t) mine = r;
        }
    } else {
        for (int t = 0; t < 8; ++t) {
            const int src = t * 8 + grp;
            const uint64_t st = __shfl((u64)s, src, kWave);
            const int lq = __shfl(l_qseq, src, kWave), nb = lq > 0 ? (lq + 1) >> 1 : 0;
            uint32_t g = 0;
            for (int at = 16 * sub; at < nb; at += 8 * 16)
                g += gc_of_piece(load16u(seq4, st + at, lim), min(nb - at, 16), (lq & 1) && nb - at <= 16);
            g += __shfl_xor(g, 1, kWave), g += __shfl_xor(g, 2, kWave), g += __shfl_xor(g, 4, kWave);
            const uint32_t r = __shfl(g, sub * 8 + t, kWave);
            if (grp == t) mine = r;
        }
    }
    return l_qseq > 0 ? mine : 0u;
}

constexpr int kWinSpan = 1024;                    // consecutive records per wave: one run of passes with private sums

// masked wave sums (valid in every lane)
__device__ __forceinline__ uint32_t wave_sum_if(bool in, uint32_t v) { return wave_sum(in ? v : 0u); }

__global__ __launch_bounds__(kWinThreads) void k_window_add(
    const int32_t *__restrict__ rec_tid, const int32_t *__restrict__ rec_pos, const uint32_t *__restrict__ rec_flag,
    const int32_t *__restrict__ l_qseq, const uint64_t *__restrict__ seq_off, const uint8_t *__restrict__ seq4,
    uint64_t n, uint64_t seq_end, uint32_t W, int32_t n_targets, const uint64_t *__restrict__ win_off,
    uint32_t *__restrict__ bins, u64 *__restrict__ gc, uint32_t *__restrict__ len, uint32_t *__restrict__ touched,
    u64 *__restrict__ n_count, uint32_t *__restrict__ bad)
{
    // Waves work on their own: no LDS table, no barrier.  A wave takes spans of 1024 consecutive records, 64 per pass, and
    // keeps the sums of the window it is in (a coordinate-sorted BAM stays in one window for thousands of records at WGS
    // depth); they go to memory with three atomics when the window changes.  Passes with more than eight different
    // windows (unsorted input) hand the rest to per-record atomics.
    __shared__ uint32_t s_gc[kWinThreads / kWave][kWave];   // per wave: the GC sums of the 64 records of a pass
    const int lane = lane_id();
    const uint64_t lim = seq_end ? seq_end : seq_off[n];   // first byte offset of seq4 that must not be read
    const uint64_t nspan = (n + kWinSpan - 1) / kWinSpan;
    const uint64_t wave0 = (uint64_t)blockIdx.x * (kWinThreads / kWave) + wave_id(), nwaves = (uint64_t)gridDim.x * (kWinThreads / kWave);
    uint32_t counted = 0;
    GcLayout lay;
    const uint32_t Wm = 0xffffffffu / W;                    // div_by()
    for (uint64_t span = wave0; span < nspan; span += nwaves) {
        u64 cur = ~0ull;                                   // window slot the sums belong to (same value in every lane)
        uint32_t a_bins = 0, a_len = 0, cur_tid = 0;
        u64 a_gc = 0;
        auto flush = [&]() {
            if (cur != ~0ull && lane == 0) {
                atomicAdd(&bins[cur], a_bins);
                atomicAdd(&len[cur], a_len);
                atomicAdd(&gc[cur], a_gc);
                touched[cur_tid] = 1u;
            }
        };
        // the fields of a pass are loaded one pass ahead, beside the sequence loads of the pass before: one exposed round
        // trip per pass instead of two (fields, then the sequences they point to)
        struct Fields {
            int32_t t = -1, p = 0, l = 0;
            uint32_t f = 0;
            uint64_t so = 0;
        };
        auto fields_of = [&](uint64_t r) {
            Fields x;
            if (r < n) x.t = rec_tid[r], x.p = rec_pos[r], x.l = l_qseq[r], x.f = rec_flag[r], x.so = seq_off[r];
            return x;
        };
        Fields nxt = fields_of(span * kWinSpan + lane);
        for (int pass = 0; pass < kWinSpan / kWave; ++pass) {
            const uint64_t r = span * kWinSpan + (uint64_t)pass * kWave + lane;
            if (__ballot(r < n) == 0) break;
            const Fields cf = nxt;
            if (pass + 1 < kWinSpan / kWave) nxt = fields_of(r + kWave);
            bool ok = false;
            uint64_t slot = 0, so = 0;
            uint32_t lq = 0, tt = 0;
            int32_t lqs = 0;
            if (r < n) {
                const int32_t t = cf.t, p = cf.p, l = cf.l;
                const uint32_t f = cf.f;
                so = cf.so, lqs = l;                           // skipped records are read too: a wave of one length stays one
                if (t >= 0 && !(f & 4u)) {                     // :96-97
                    if (t >= n_targets) {
                        atomicOr(bad, 1u);
                    } else {
                        // c->pos / window in int, then (unsigned short) (:117)
                        // (one multiplication when the position is not negative -- always, for a mapped record)
                        const uint32_t w16 = (uint32_t)(uint16_t)(p >= 0 ? (int32_t)div_by((uint32_t)p, W, Wm) : p / (int32_t)W);
                        const uint64_t lo = win_off[t], hi = win_off[t + 1];
                        if (lo + w16 >= hi) {
                            atomicOr(bad, 2u);                 // the reference would write out of bounds
                        } else {
                            ok = true, slot = lo + w16, lq = (uint32_t)l, tt = (uint32_t)t;
                        }
                    }
                }
            }
            // every lane takes part (lanes without a record contribute an empty sequence)
#ifdef DIAG_NOGC
            const uint32_t g = (uint32_t)so & 63u;
#else
            const uint32_t g = (uint32_t)(uint16_t)gc_of_wave_records(seq4, so, lqs, lim, s_gc[wave_id()], lay);   // unsigned short current_GC (:118)
#endif
            u64 rem = __ballot(ok);
            counted += (uint32_t)__builtin_popcountll(rem);   // n_count (:104); the same in every lane
            for (int it = 0; rem; ++it) {
                if (it == 8) {                                 // many windows under one pass: unsorted input
                    if (ok && ((rem >> lane) & 1)) {
                        atomicAdd(&bins[slot], 1u);
                        atomicAdd(&len[slot], lq);
                        atomicAdd(&gc[slot], (u64)g);
                        touched[tt] = 1u;                      // benign same-value race
                    }
                    break;
                }
                const int f = __builtin_ctzll(rem);
                const u64 s0 = __shfl((u64)slot, f, kWave);
                const bool in = ok && slot == s0;
                const u64 m = __ballot(in);
                if (s0 != cur) {
                    flush();
                    cur = s0, a_bins = 0, a_len = 0, a_gc = 0, cur_tid = __shfl(tt, f, kWave);
                }
                a_bins += (uint32_t)__builtin_popcountll(m);
                a_len += wave_sum_if(in, lq);
                a_gc += wave_sum_if(in, g);
                rem &= ~m;
            }
        }
        flush();
    }
    if (lane == 0 && counted) atomicAdd(n_count, (u64)counted);
}

hipError_t launch_window_add(const int32_t *tid_a, const int32_t *pos, const uint32_t *flag, const int32_t *l_qseq,
                             const uint64_t *seq_off, const uint8_t *seq4, uint64_t n, uint64_t seq_end, uint32_t W, int32_t n_targets,
                             const uint64_t *win_off, uint32_t *bins, u64 *gc, uint32_t *len, uint32_t *touched,
                             u64 *n_count, uint32_t *bad, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t want = ((n + kWinSpan - 1) / kWinSpan + kWinThreads / kWave - 1) / (kWinThreads / kWave);
    const uint64_t cap = (uint64_t)n_cu * 8;
    hipLaunchKernelGGL(k_window_add, dim3((unsigned)(want < cap ? want : cap)), dim3(kWinThreads), 0, st, tid_a, pos, flag,
                       l_qseq, seq_off, seq4, n, seq_end, W, n_targets, win_off, bins, gc, len, touched, n_count, bad);
    return hipGetLastError();
}

}  // namespace hpn
