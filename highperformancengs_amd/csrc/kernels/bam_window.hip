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

// A raw buffer descriptor over [base, base + bytes): loads through it take a 32-bit lane offset plus a scalar offset (no
// 64-bit address arithmetic in the vector unit) and return zero for anything that does not lie inside whole.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t span_rsrc(const void *base, uint64_t bytes)
{
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const uint32_t nb = bytes > 0x7fffffffull ? 0x7fffffffu : (uint32_t)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(
        (void *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b)),
        0, (int)__builtin_amdgcn_readfirstlane(nb), 0x00020000);
}

// GC count of one 16-byte piece added to acc; nk[j] = ~(the nibbles of word j to count, as a subset of 0x44444444).
// Per word: w ^ (w << 1) has b1 ^ b2 at bit 2 of every nibble, (w << 2) | nk puts "bit 0 set, or not to be counted" there (and
// ones everywhere else), w >> 1 puts bit 3 there; verdict = that xor, and not the other two (gfx950's three-input bit operation
// folds the logic into two instructions), then one count-and-add.  (3 w instead of the xor would save an instruction and is
// wrong: the sum carries from one nibble into the next.)
__device__ __forceinline__ uint32_t gc_add_piece(const u32 q, const uint32_t *nk, uint32_t acc)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = q[j];
        const uint32_t one = (w << 1) ^ w, b0n = (w << 2) | nk[j], b3 = w >> 1;
        const uint32_t v = one & ~b0n & ~b3;
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(v));      // count-and-add in one instruction, as a chain (no partial sums kept;
    }                                                               //  two interleaved chains measured the same: the kernel waits for memory)
    return acc;
}

// Offsets that can never lie inside a fast pass's descriptor (at most kFastReach bytes), whatever is added to them:
// kNoRecord marks a record that is skipped (flag & 4, tid < 0, beyond n), kNoLane a lane without a piece in a step.
constexpr uint32_t kFastReach = 0x3fffffffu, kNoRecord = 0x40000000u, kNoLane = 0x80000000u;
#ifndef HPN_K5_NT
#define HPN_K5_NT 2
#endif
#ifndef HPN_K5_CHUNK
#define HPN_K5_CHUNK 16
#endif
#ifndef HPN_K5_SPAN
#define HPN_K5_SPAN 1
#endif
#ifndef HPN_K5_SEAM
#define HPN_K5_SEAM 1
#endif
#ifndef HPN_K5_EU
#define HPN_K5_EU 4
#endif
constexpr int kNt = HPN_K5_NT;     // cache policy of a buffer load: 2 = non-temporal (the bytes are read once)

// The sequence loads and the count of a fast pass of STEPS steps: every lane fetches the offset of its record of each step from
// the lane that holds the record (ds_bpermute, all of them first), adds its piece's offset, loads 16 bytes and counts.  A
// skipped record hands out kNoRecord and a lane without a piece adds kNoLane: the hardware drops those loads and returns zeros,
// which count as nothing -- no mask, no branch, and the bytes of skipped records are not even fetched.
template <int STEPS>
__device__ __forceinline__ uint32_t fast_pass_gc(__amdgpu_buffer_rsrc_t rsrc, uint32_t srel, uint32_t bp_addr, int rps, uint32_t off_mid,
                                                 uint32_t off_last, const uint32_t *nk, uint32_t acc)
{
    uint32_t at[STEPS];
#pragma unroll
    for (int t = 0; t < STEPS; ++t) at[t] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(bp_addr + (uint32_t)(4 * t * rps)), (int)srel);
    __builtin_amdgcn_sched_barrier(0);        // (all the exchanges in flight together, then all the loads)
    u32 q[STEPS];
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        const uint32_t a = at[t] + (t + 1 < STEPS ? off_mid : off_last);
#ifdef DIAG_NOSEQ
        q[t] = u32{a, a * 3u, a * 5u, a * 7u};
#else
        q[t] = __builtin_bit_cast(u32, __builtin_amdgcn_raw_buffer_load_b128(rsrc, a, 0, 0));
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < STEPS; ++t) acc = gc_add_piece(q[t], nk, acc);
    return acc;
}

// bytes k .. 3 of a word (k <= 0: all, k >= 4: none)
__device__ __forceinline__ uint32_t bytes_from(int k) { return k <= 0 ? 0xffffffffu : k >= 4 ? 0u : 0xffffffffu << (8 * k); }

// A pass whose sequences lie back to back (a batch handed over as arrays: seq_off[r + 1] = seq_off[r] + bytes) and have an even
// number of bases: the pass's bytes are ONE span, read as aligned 16-byte pieces in lane order -- a wave instruction is 1 KiB of
// whole cache lines, a quarter of the requests the record-wise pieces put to the vector cache (whose request rate, not HBM and
// not instruction issue, bounded the kernel: 1.04 ms whether a pass cost 485 or 200 vector instructions).  Every nibble of the span
// is a base; only the first and the last piece need a mask (the bytes before the span in its first 16, after it in its last).
// total = bytes from the aligned start to the span's end, head = bytes of the first piece in front of the span.
template <int STEPS>
__device__ __forceinline__ uint32_t span_pass_gc(__amdgpu_buffer_rsrc_t rsrc, uint32_t head, uint32_t total, uint32_t acc)
{
    const uint32_t c0 = 16u * (uint32_t)lane_id();
    u32 q[STEPS];
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        const uint32_t c = c0 + 1024u * (uint32_t)t;
        const uint32_t off = t + 1 < STEPS || c < total ? c0 : kNoLane;     // (only the last step has lanes beyond the span)
#ifdef DIAG_NOSEQ
        q[t] = u32{off, off * 3u, off * 5u, off * 7u};
#else
        q[t] = __builtin_bit_cast(u32, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 1024 * t, kNt));   // read once: non-temporal
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t all[4] = {~0x44444444u, ~0x44444444u, ~0x44444444u, ~0x44444444u};
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        if (t == 0 || t == STEPS - 1) {
            const int c = (int)c0 + 1024 * t, lo = (int)head - c, hi = (int)total - c;     // the piece's bytes [lo, hi) belong to the span
            uint32_t nk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) nk[j] = ~(bytes_from(lo - 4 * j) & ~bytes_from(hi - 4 * j) & 0x44444444u);
            acc = gc_add_piece(q[t], nk, acc);
        } else {
            acc = gc_add_piece(q[t], all, acc);
        }
    }
    return acc;
}

__global__ __launch_bounds__(kWinThreads) __attribute__((amdgpu_waves_per_eu(HPN_K5_EU, 8))) void k_window_add(
    const int32_t *__restrict__ rec_tid, const int32_t *__restrict__ rec_pos, const uint32_t *__restrict__ rec_flag,
    const int32_t *__restrict__ l_qseq, const uint64_t *__restrict__ seq_off, const uint8_t *__restrict__ seq4,
    uint64_t n, uint64_t seq_end, uint32_t W, int32_t n_targets, const uint64_t *__restrict__ win_off,
    uint32_t *__restrict__ bins, u64 *__restrict__ gc, uint32_t *__restrict__ len, uint32_t *__restrict__ touched,
    u64 *__restrict__ n_count, uint32_t *__restrict__ todo /* [0]: passes left for k_window_rest, their numbers from [1] */)
{
    // Waves work on their own: no barrier.  A wave takes spans of 1024 consecutive records, 64 per pass, and keeps the sums of
    // the window it is in (a coordinate-sorted BAM stays in one window for thousands of records at WGS depth); they go to
    // memory with three atomics when the window changes.
    //
    // The pass the kernel is built around (round 4): 64 records of ONE target, ONE window and ONE read length, as in a
    // coordinate-sorted BAM of fixed-length reads.  Whether a pass is one is decided with wave-uniform state -- the target, the
    // window's position range [w_lo, w_lo + W) and the read length of the pass before sit in scalar registers, a lane only
    // compares its record against them (no division, no window-table loads) -- and then nothing is attributed to records at
    // all: a lane adds the GC of its pieces to a private register (pieces of records that are skipped, flag & 4 / tid < 0,
    // masked by one bit test), bins and len grow by popcount(ok) and popcount(ok) x length in scalar registers, and the
    // lanes' registers are summed once, when the window changes or the span ends.  The fields come through buffer
    // descriptors of the span (address = descriptor + lane x 4 + pass x 256: no vector address arithmetic).
    // Everything else -- mixed lengths, a window or target boundary inside the pass, unsorted input, reads over 256 bases, the
    // last bytes of the sequence buffer -- is left to k_window_rest: the pass's number goes on a list (one atomic per such
    // pass; ~2 % of the passes of a sorted WGS BAM at W = 20000).  Two kernels rather than two branches because the general
    // pass needs three times the registers: inlined here it halved this kernel's occupancy.
    const int lane = lane_id();
    const uint64_t lim = seq_end ? seq_end : seq_off[n];   // first byte offset of seq4 that must not be read
    const uint64_t nspan = (n + kWinSpan - 1) / kWinSpan;
    // (the wave's number through readfirstlane: the compiler cannot see that threadIdx.x >> 6 is the same in a wave's lanes, and
    //  would run every loop and branch below under an execution mask)
    const uint64_t wave0 = (uint64_t)blockIdx.x * (kWinThreads / kWave) + (uint32_t)__builtin_amdgcn_readfirstlane(wave_id()),
                   nwaves = (uint64_t)gridDim.x * (kWinThreads / kWave);
    uint32_t counted = 0;
    GcLayout lay;
    uint32_t off_mid = kNoLane, off_last = kNoLane;           // fast pass: 16 x piece number, or "no load" for a lane without a piece
    uint32_t bp_addr = 0;                                     // fast pass: 4 x (record of step 0) for ds_bpermute
    const uint32_t Wm = 0xffffffffu / W;                    // div_by()
    const uint32_t lane4 = (uint32_t)lane * 4u, lane8 = (uint32_t)lane * 8u;
    // what the pass before established (all wave-uniform)
    int32_t c_tid = -1;            // target whose window table entry is cached
    uint64_t c_lo = 0;             // win_off[c_tid]
    uint32_t c_nwin = 0;           // windows of c_tid (clamped to 2^32 - 1)
    uint32_t w_lo = 0;             // first position of the cached window
    bool w_valid = false;          // [w_lo, w_lo + W) belongs to slot c_slot of target c_tid
    uint64_t c_slot = 0;
    for (uint64_t span = wave0; span < nspan; span += nwaves) {
        u64 cur = ~0ull;                                   // window slot the sums belong to (same value in every lane)
        uint32_t a_bins = 0, a_len = 0, cur_tid = 0;
        u64 a_gc = 0;
        uint32_t lane_gc = 0;                              // this lane's share of the window's GC sum (fast passes)
        auto flush = [&]() {                               // wave-uniform control flow only: every lane takes part in the sum
            const uint32_t tot = wave_sum(lane_gc);
            lane_gc = 0;
            if (cur != ~0ull && lane == 0) {
                atomicAdd(&bins[cur], a_bins);
                atomicAdd(&len[cur], a_len);
                atomicAdd(&gc[cur], a_gc + tot);
                touched[cur_tid] = 1u;
            }
        };
        const uint64_t r0 = span * kWinSpan;
        const uint32_t in_span = (uint32_t)(n - r0 < (uint64_t)kWinSpan ? n - r0 : (uint64_t)kWinSpan);
        const __amdgpu_buffer_rsrc_t d_tid = span_rsrc(rec_tid + r0, (uint64_t)in_span * 4), d_pos = span_rsrc(rec_pos + r0, (uint64_t)in_span * 4),
                                     d_flag = span_rsrc(rec_flag + r0, (uint64_t)in_span * 4), d_lq = span_rsrc(l_qseq + r0, (uint64_t)in_span * 4),
                                     d_so = span_rsrc(seq_off + r0, (uint64_t)in_span * 8);
        // the fields of a pass are loaded one pass ahead, beside the sequence loads of the pass before: one exposed round
        // trip per pass instead of two (fields, then the sequences they point to)
        struct Fields {
            int32_t t = -1, p = 0, l = 0;
            uint32_t f = 0;
            uint64_t so = 0;
        };
        auto fields_of = [&](int pass) {
            Fields x;
            const int so4 = pass * (kWave * 4), so8 = pass * (kWave * 8);
            x.t = __builtin_amdgcn_raw_buffer_load_b32(d_tid, lane4, so4, kNt);
            x.p = __builtin_amdgcn_raw_buffer_load_b32(d_pos, lane4, so4, kNt);
            x.f = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(d_flag, lane4, so4, kNt);
            x.l = __builtin_amdgcn_raw_buffer_load_b32(d_lq, lane4, so4, kNt);
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(d_so, lane8, so8, kNt));
            x.so = ((uint64_t)v[1] << 32) | v[0];
            return x;
        };
        Fields nxt = fields_of(0);
        const int npass = (int)((in_span + kWave - 1) / kWave);
        for (int pass = 0; pass < npass; ++pass) {
            const Fields cf = nxt;
            if (pass + 1 < npass) nxt = fields_of(pass + 1);
            const uint32_t left = in_span - (uint32_t)pass * kWave;
            const uint32_t nvalid = left < (uint32_t)kWave ? left : (uint32_t)kWave;   // 1 .. 64 records in this pass; lanes at or beyond it hold none
            const bool valid = (uint32_t)lane < nvalid;
            const bool okf = valid && cf.t >= 0 && !(cf.f & 4u);                // :96-97
            const u64 okm = __ballot(okf);
            // ---- is this a fast pass?  (every test below is one or two vector compares against scalar registers) ----
            bool fast = false;
            int lq0 = 0;
            uint64_t s0 = 0;
            uint32_t srel = 0, srel_all = 0, srel_skip = 0, reach = 0, second_lo = 0;
            uint64_t second_slot = 0;
            u64 m_second = 0;                 // ok records of the pass that belong to the window after the cached one
            bool span = false;
            if (okm) {
                const int first = __builtin_ctzll(okm);
                const int32_t t0 = __builtin_amdgcn_readlane(cf.t, first);
                lq0 = __builtin_amdgcn_readlane(cf.l, 0);                           // (lane 0 is always valid)
                if (t0 != c_tid && t0 < n_targets) {                                // a new target: its window table entry, once
                    c_tid = t0;
                    c_lo = win_off[t0];
                    const uint64_t nw = win_off[t0 + 1] - c_lo;
                    c_nwin = nw > 0xffffffffull ? 0xffffffffu : (uint32_t)nw;
                    w_valid = false;
                }
                if (t0 == c_tid && lq0 > 0 && lq0 <= 256) {
                    // one target, one length (skipped records too: their pieces are loaded, and dropped by a bit test)
                    u64 odd = __ballot((okf && cf.t != t0) || (valid && cf.l != lq0) || (okf && cf.p < 0));
                    if (!odd && !(w_valid && __ballot(okf && (uint32_t)cf.p - w_lo >= W) == 0)) {
                        // not (known to be) inside the cached window: the window of the first record, by division
                        const uint32_t wfull = div_by((uint32_t)cf.p, W, Wm);
                        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)wfull, first);
                        odd = __ballot(okf && wfull != w0);
                        const uint32_t w16 = w0 & 0xffffu;                          // (unsigned short)(c->pos / window) (:117)
                        if (!odd && w16 < c_nwin) {
                            w_valid = true, w_lo = w0 * W, c_slot = c_lo + w16;
                        } else if (HPN_K5_SEAM && w16 < c_nwin && ((w0 + 1u) & 0xffffu) < c_nwin && __ballot(okf && wfull - w0 > 1u) == 0) {
                            // the pass crosses ONE window seam (the ~2 % of a sorted BAM's passes that are not of one window): the
                            // records of the first window now, those of the next as a second round of this pass
                            odd = 0;
                            m_second = __ballot(okf && wfull != w0);
                            w_valid = true, w_lo = w0 * W, c_slot = c_lo + w16;
                            second_lo = (w0 + 1u) * W, second_slot = c_lo + ((w0 + 1u) & 0xffffu);
                        } else {
                            odd = 1;                                                // several windows, or one the reference would write out of bounds
                            w_valid = false;
                        }
                    }
                    if (!odd) {
                        // the sequences relative to the first record's, inside what may be read, 2 GiB at most
                        s0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(cf.so >> 32), 0) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cf.so, 0);
                        const uint64_t room = lim > s0 ? lim - s0 : 0;
                        reach = room > kFastReach ? kFastReach : (uint32_t)room;
                        const uint64_t rel = cf.so - s0;                              // (wraps for a record in front of the first: then it is huge)
                        const uint32_t piece_bytes = 16u * (uint32_t)((((lq0 + 1) >> 1) + 15) >> 4);
                        fast = __ballot(valid && rel + piece_bytes > reach) == 0;     // no piece of the pass straddles the end of what may be read
                        srel = okf ? (uint32_t)rel : kNoRecord;                       // a skipped record's pieces are not loaded
                        srel_all = (uint32_t)rel;
                        // back to back, an even number of bases, at least a piece per record: the pass is one span (span_pass_gc)
                        const uint32_t nbytes = (uint32_t)lq0 >> 1;
                        span = HPN_K5_SPAN && !m_second && fast && !(lq0 & 1) && lq0 >= 32 && __ballot(valid && rel != (uint64_t)((uint32_t)lane * nbytes)) == 0;
                        srel_skip = valid && !okf ? (uint32_t)rel : kNoRecord;        // ... from which the skipped records are taken off again
                    }
                }
            }
            if (fast) {
                if (lq0 != lay.lq) {               // a new read length: lane -> (record of the step, piece) and the byte masks
                    const int nb = (lq0 + 1) >> 1, P = (nb + 15) >> 4;                // 1..8, the same in every lane
                    lay.lq = lq0, lay.rps = kWave / P, lay.steps = (kWave + lay.rps - 1) / lay.rps;
                    lay.pg = lane / P, lay.psub = lane - lay.pg * P;
                    const int rem = nb - 16 * lay.psub;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int hi = min(max(rem - 4 * j, 0), 4);
                        uint32_t m = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
                        if ((lq0 & 1) && hi > 0 && rem - 4 * j <= 4) m &= ~(0xfu << (8 * (hi - 1)));
                        lay.m[j] = ~(m & 0x44444444u);             // NOT the nibbles of word j that are bases of the read (gc_add_piece)
                    }
                    const bool used = lay.pg < lay.rps;
                    off_mid = used ? 16u * (uint32_t)lay.psub : kNoLane;
                    off_last = used && (lay.steps - 1) * lay.rps + lay.pg < kWave ? off_mid : kNoLane;
                    bp_addr = 4u * (uint32_t)lay.pg;
                }
                const __amdgpu_buffer_rsrc_t rsrc = span_rsrc(seq4 + s0, reach);
                if (m_second) {
                    // two windows under the pass: each round takes its records' pieces (the others hand out kNoRecord)
                    for (int round = 0; round < 2; ++round) {
                        const u64 m = round ? m_second : okm & ~m_second;
                        if (round) w_lo = second_lo, c_slot = second_slot;
                        if (c_slot != cur) {
                            flush();
                            cur = c_slot, a_bins = 0, a_len = 0, a_gc = 0, cur_tid = (uint32_t)c_tid;
                        }
                        const uint32_t nm = (uint32_t)__builtin_popcountll(m);
                        counted += nm, a_bins += nm, a_len += nm * (uint32_t)lq0;
                        const uint32_t sr = (m >> lane) & 1ull ? srel_all : kNoRecord;
#define HPN_FAST_STEPS(N) case N: lane_gc = fast_pass_gc<N>(rsrc, sr, bp_addr, lay.rps, off_mid, off_last, lay.m, lane_gc); break;
                        switch (lay.steps) {
                            HPN_FAST_STEPS(1) HPN_FAST_STEPS(2) HPN_FAST_STEPS(3) HPN_FAST_STEPS(4) HPN_FAST_STEPS(5) HPN_FAST_STEPS(6) HPN_FAST_STEPS(7)
                            default: lane_gc = fast_pass_gc<8>(rsrc, sr, bp_addr, lay.rps, off_mid, off_last, lay.m, lane_gc); break;
                        }
#undef HPN_FAST_STEPS
                    }
                    continue;
                }
                if (c_slot != cur) {
                    flush();
                    cur = c_slot, a_bins = 0, a_len = 0, a_gc = 0, cur_tid = (uint32_t)c_tid;
                }
                const uint32_t nok = (uint32_t)__builtin_popcountll(okm);
                counted += nok, a_bins += nok, a_len += nok * (uint32_t)lq0;     // n_count (:104), bins, len (:119-121)
                const u64 validm = __ballot(valid);
                uint32_t head = 0, total = 0;
                // (a pass with skipped records would need both fetches -- the span, and the skipped records' pieces to take them off
                //  again: measured slower than the record-wise pieces alone, 1.03 against 0.95 ms where 95 % of the passes hold one)
                span = span && okm == validm;
                if (span) {
                    head = (uint32_t)(((uint64_t)(uintptr_t)seq4 + s0) & 15u);
                    total = head + nvalid * ((uint32_t)lq0 >> 1);
                    span = total <= 8192u && (uint64_t)total - head + 16u <= (uint64_t)reach + 0u;    // eight steps at most; the last piece inside what may be read
                }
                uint32_t sub = 0;          // GC of the pieces fetched record-wise: the pass's (no span), or its skipped records' (span)
#define HPN_FAST_STEPS(N) case N: sub = fast_pass_gc<N>(rsrc, span ? srel_skip : srel, bp_addr, lay.rps, off_mid, off_last, lay.m, 0u); break;
                if (!span || okm != validm) {
                    switch (lay.steps) {          // ceil(64 / floor(64 / pieces per record)): 1, 2, 3, 4, 6 (150 bases), 7 or 8
                        HPN_FAST_STEPS(1) HPN_FAST_STEPS(2) HPN_FAST_STEPS(3) HPN_FAST_STEPS(4) HPN_FAST_STEPS(5) HPN_FAST_STEPS(6) HPN_FAST_STEPS(7)
                        default: sub = fast_pass_gc<8>(rsrc, span ? srel_skip : srel, bp_addr, lay.rps, off_mid, off_last, lay.m, 0u); break;
                    }
                }
#undef HPN_FAST_STEPS
                if (!span) {
                    lane_gc += sub;
                } else {
                    // the whole span, then the skipped records' share off again (lane sums modulo 2^32: the wave's sum is what counts)
                    const __amdgpu_buffer_rsrc_t rs2 = span_rsrc(seq4 + s0 - head, (uint64_t)reach + head);
#define HPN_SPAN_STEPS(N) case N: lane_gc = span_pass_gc<N>(rs2, head, total, lane_gc); break;
                    switch ((total + 1023u) >> 10) {
                        HPN_SPAN_STEPS(1) HPN_SPAN_STEPS(2) HPN_SPAN_STEPS(3) HPN_SPAN_STEPS(4) HPN_SPAN_STEPS(5) HPN_SPAN_STEPS(6) HPN_SPAN_STEPS(7)
                        default: lane_gc = span_pass_gc<8>(rs2, head, total, lane_gc); break;
                    }
#undef HPN_SPAN_STEPS
                    lane_gc -= sub;
                }
                continue;
            }
            // ---- not a fast pass: k_window_rest takes it ----
            if (okm && lane == 0) todo[1 + atomicAdd(&todo[0], 1u)] = (uint32_t)(r0 / kWave) + (uint32_t)pass;
        }
        flush();
    }
    if (lane == 0 && counted) atomicAdd(n_count, (u64)counted);
}

// The passes k_window_add left: 64 records each, any mixture.  Per-record GC through the wave's 64 LDS words
// (gc_of_wave_records), up to eight windows per pass with wave sums, then per-record atomics.  A wave takes four list entries at a
// time and keeps the sums of the window it is in across them (the list of an unsorted input is nearly in order).
__global__ __launch_bounds__(kWinThreads) void k_window_rest(
    const int32_t *__restrict__ rec_tid, const int32_t *__restrict__ rec_pos, const uint32_t *__restrict__ rec_flag,
    const int32_t *__restrict__ l_qseq, const uint64_t *__restrict__ seq_off, const uint8_t *__restrict__ seq4,
    uint64_t n, uint64_t seq_end, uint32_t W, int32_t n_targets, const uint64_t *__restrict__ win_off,
    uint32_t *__restrict__ bins, u64 *__restrict__ gc, uint32_t *__restrict__ len, uint32_t *__restrict__ touched,
    u64 *__restrict__ n_count, uint32_t *__restrict__ bad, const uint32_t *__restrict__ todo)
{
    __shared__ uint32_t s_gc[kWinThreads / kWave][kWave];   // per wave: the GC sums of the 64 records of a pass
    const uint32_t n_todo = todo[0];
    if (n_todo == 0) return;
    const int lane = lane_id();
    const uint64_t lim = seq_end ? seq_end : seq_off[n];   // first byte offset of seq4 that must not be read
    const uint32_t wave0 = blockIdx.x * (kWinThreads / kWave) + (uint32_t)__builtin_amdgcn_readfirstlane(wave_id()), nwaves = gridDim.x * (kWinThreads / kWave);
    uint32_t counted = 0;
    GcLayout lay;
    const uint32_t Wm = 0xffffffffu / W;                    // div_by()
    constexpr uint32_t kChunk = HPN_K5_CHUNK;      // (list entries per wave and turn: a sorted BAM leaves ~2 % of its passes, spread thin)
    for (uint32_t e0 = wave0 * kChunk; e0 < n_todo; e0 += nwaves * kChunk) {
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
        for (uint32_t e = e0; e < e0 + kChunk && e < n_todo; ++e) {
            const uint64_t r = (uint64_t)todo[1 + e] * kWave + lane;
            bool ok = false;
            uint64_t slot = 0, so = 0;
            uint32_t lq = 0, tt = 0;
            int32_t lqs = 0;
            if (r < n) {
                const int32_t t = rec_tid[r], p = rec_pos[r], l = l_qseq[r];
                const uint32_t f = rec_flag[r];
                so = seq_off[r], lqs = l;                      // skipped records are read too: a wave of one length stays one
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
            const uint32_t g = (uint32_t)(uint16_t)gc_of_wave_records(seq4, so, lqs, lim, s_gc[__builtin_amdgcn_readfirstlane(wave_id())], lay);   // unsigned short current_GC (:118)
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
                const u64 sl0 = __shfl((u64)slot, f, kWave);
                const bool in = ok && slot == sl0;
                const u64 m = __ballot(in);
                if (sl0 != cur) {
                    flush();
                    cur = sl0, a_bins = 0, a_len = 0, a_gc = 0, cur_tid = __shfl(tt, f, kWave);
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

size_t window_todo_words(uint64_t n) { return (size_t)((n + kWave - 1) / kWave) + 1; }   // the list of passes left for k_window_rest

hipError_t launch_window_add(const int32_t *tid_a, const int32_t *pos, const uint32_t *flag, const int32_t *l_qseq,
                             const uint64_t *seq_off, const uint8_t *seq4, uint64_t n, uint64_t seq_end, uint32_t W, int32_t n_targets,
                             const uint64_t *win_off, uint32_t *bins, u64 *gc, uint32_t *len, uint32_t *touched,
                             u64 *n_count, uint32_t *bad, uint32_t *todo /* window_todo_words(n) */, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    if (n >= ((uint64_t)1 << 32) * kWave) return hipErrorInvalidValue;      // pass numbers are 32-bit
    hipError_t e = hipMemsetAsync(todo, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    uint64_t want = ((n + kWinSpan - 1) / kWinSpan + kWinThreads / kWave - 1) / (kWinThreads / kWave);
    // exactly the workgroups the chip holds at once: the spans are dealt out round-robin, so a grid of 8 per CU on a kernel that
    // fits 7 runs a second round at one eighth of the chip (1.06 ms where one round takes half of that)
    static const int per_cu = [] {
        int b = 0;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_window_add, kWinThreads, 0) == hipSuccess && b > 0 ? b : 4;
    }();
    const uint64_t cap = (uint64_t)n_cu * (uint64_t)per_cu;
    hipLaunchKernelGGL(k_window_add, dim3((unsigned)(want < cap ? want : cap)), dim3(kWinThreads), 0, st, tid_a, pos, flag,
                       l_qseq, seq_off, seq4, n, seq_end, W, n_targets, win_off, bins, gc, len, touched, n_count, todo);
    // whatever the fast kernel left (nothing: its waves return at once)
    want = ((n + kWave - 1) / kWave + HPN_K5_CHUNK * (kWinThreads / kWave) - 1) / (HPN_K5_CHUNK * (kWinThreads / kWave));
    const uint64_t cap2 = (uint64_t)n_cu * 4;
    hipLaunchKernelGGL(k_window_rest, dim3((unsigned)(want < cap2 ? want : cap2)), dim3(kWinThreads), 0, st, tid_a, pos, flag,
                       l_qseq, seq_off, seq4, n, seq_end, W, n_targets, win_off, bins, gc, len, touched, n_count, bad, todo);
    return hipGetLastError();
}

}  // namespace hpn
