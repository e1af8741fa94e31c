// fastq_tally.hip -- K1L `k_tally_hist`: the full per-(symbol, cycle) histograms.
//
// Replaces AssignQuality (reference fastq_count.c:29-35) when the whole
// Quality[128][512] matrix is wanted (`fastq_count_kthread -L`, printQ :52-64) and
// the per-cycle nucleotide tally of Rgzfastq_uniq.c:50-57 (Nucleotide[5][512]).
// The default report needs neither: that is K1, fastq_scan.hip.
//
// One 1024-thread workgroup per CU owns a histogram image in LDS: 32-bit counters,
// 128 symbol rows x cycles 0..255, 256 dwords per row.  With group g = cycle >> 2:
//     word(row, cycle) = row*256 + k*64 + (g & 1)*32 + ((g >> 1) + 8k) % 32,  k = cycle & 3        (lds_word())
// Cycles 256..511 (reads longer than 256 bases) go to the global matrix directly.
// The image is flushed once, at the end of the kernel (a 32-bit counter cannot wrap
// within one launch); sum / Q20 / Q30 are row sums taken during that flush.
//
//   records of one length, 16 <= len <= 256 (the normal case; up to 4 x 4096 records per workgroup turn):
//     work item = (read r, group j) = cycles 8j..8j+7 OF THE READ, one unaligned 8-byte buffer load
//     (descriptor of the chunk + lane offset + m * step in an SGPR: no address arithmetic per load).
//     Byte 4e + k of the item goes to word row*256 + k*64 + 32e + (j + 8k) % 32, so in each of the
//     eight adds the lanes of a read hit different banks whatever the symbols are, and the reads under
//     one wave stay within the four lanes per bank that an LDS atomic handles at no extra cost
//     (scripts/micro/lds_atomic.hip: 4.3 clocks per wave ds_add_u32 = 16 lanes per clock per CU,
//     free up to 4 lanes per bank, +2 clocks per extra lane on one ADDRESS).  Reads shorter than 121
//     bases (more than four under a wave) take the bytes of a dword in an order rotated by the read's
//     number: with the 8k shift the wave's reads then work on four different windows of banks.
//     2 VALU + 1 LDS per byte, no per-byte compare: bit 7 of every byte is OR-ed into one
//     flag and the row index is masked.  A lane keeps its group j for the whole chunk; two sets of
//     eight items: one in flight while the other is tallied.
//   anything else (ragged lengths, very short or very long reads), 4096 records at a time:
//     aligned 16-byte vectors of the chunk's byte range, binary search over the LDS
//     boundaries per vector, per-byte walk.
//
// History, 2e8 x 150 bp: 22.5 ms aligned 16-B vectors + binary search; 14.9 lanes diverged on cycle
// parity / read boundaries; 12.1 read-aligned 16-B vectors (LDS at the random-scatter conflict
// rate); 11.1 byte loads; 10.1 predicate-free body; 8.8 dword items (round 1).  Round 2: 8.2 the
// len % 4 tail as a masked partial group; 7.5 static lane -> group mapping, 16 dwords in flight;
// 6.3 buffer loads with SGPR offsets (a wave64 VALU instruction holds its SIMD for 4 clocks: the
// address arithmetic of flat loads was a quarter of the kernel's issue slots); 6.06 8-byte items
// and one-length turns of 16384 records.  Timing-only builds of this form: without the LDS
// atomics 6.2 ms, without the loads 4.9 ms, without both 3.0 ms: what bounds it is the memory
// side at ~5 TB/s (K1 streams 6.9), not the atomics.  No MFMA: there is no contraction here.
// Round 3, the nucleotide pass (2e8 x 150 bp, qualities + bases): 13.8 -> 12.2 ms with the five rows counted in packed
// registers (stream_uniform): that pass was the one bound by LDS operations (a table read AND an atomic per base).
#include <stdlib.h>

#include "tally_util.hpp"

namespace hpn {

constexpr int kHistThreads = 1024;
constexpr int kHistWaves = kHistThreads / kWave;
constexpr int kHistRecs = 4096;    // records per chunk: 600 KB at 150 bp between barriers (2048 / 1024: +6 % / +12 %)
// Items per set and sets in the ring (A/B builds: -DHPN_SPAN1=.. -DHPN_SETS=..).  With buffer loads, dword items: 4 / 6 / 8 / 10 /
// 12 / 16 per set -> 7.6 / 6.5 / 6.4 / 6.7 / 7.3 / 7.0 ms; 2 / 3 / 4 sets of 8 -> 6.3 / 6.7 / 6.9: more loads in flight per
// wave do not help.  Chunks per one-length turn 1 / 2 / 4 / 8 / 16 -> 6.29 / 6.09 / 6.06 / 6.16 / 6.38.
#ifndef HPN_HIST_BIG
#define HPN_HIST_BIG 4
#endif
#ifndef HPN_SETS
#define HPN_SETS 2
#endif
#ifndef HPN_SPAN1
#define HPN_SPAN1 8
#endif
#ifndef HPN_SPAN2
#define HPN_SPAN2 8
#endif
constexpr int kSpanOne = HPN_SPAN1, kSpanBoth = HPN_SPAN2;
constexpr int kLdsCycles = 256;    // cycles held in LDS; later cycles go to global atomics
constexpr int kRowWords = kLdsCycles;

// Symbol code of a base for Nucleotide[5][512] (reference Rgzfastq_uniq.c:97-108:
// T/U 0, C 1, A 2, G 3, N and '.' 4, every other byte 0).
__device__ __forceinline__ uint32_t nuc_code(uint32_t b)
{
    const uint32_t u = b & 0xdfu;  // fold case for letters
    uint32_t c = 0;
    c = (u == 'C') ? 1u : c;
    c = (u == 'A') ? 2u : c;
    c = (u == 'G') ? 3u : c;
    c = (b == 'N' || b == '.') ? 4u : c;
    return c;
}

// junk words (one per lane of a wave) directly behind each image, addressed relative to the image's first word
constexpr uint32_t kJunkQ = HPN_QUAL_ROWS * kRowWords, kJunkN = HPN_NUC_CODES * kRowWords;

struct HistLds {
    uint32_t qh[HPN_QUAL_ROWS * kRowWords];
    uint32_t qjunk[64];
    uint32_t nh[HPN_NUC_CODES * kRowWords];
    uint32_t njunk[64];
    uint8_t nlut[256];       // nuc_code of every byte value: one LDS read per base instead of ten VALU operations
    uint32_t nlut6[256];     // 1 << 6 * nuc_code(b): what a base adds to a lane's packed counters (stream_uniform)
    uint32_t loff[kHistRecs + 1];
    uint32_t lhist[HPN_LEN_BINS + 1];
    u64 red[3][kHistWaves];
};

// Index of (row, cycle) inside the LDS image: see the head of the file.  A lane that holds the eight bytes of
// cycles 8j..8j+7 of a read adds byte 4e + k at word row*256 + k*64 + 32e + j.
__device__ __forceinline__ uint32_t lds_word(uint32_t row, uint32_t pos)
{
    const uint32_t g = pos >> 2, k = pos & 3u;
    return row * kRowWords + (k << 6) + ((g & 1u) << 5) + (((g >> 1) + 8u * k) & 31u);
}

struct HiTot {  // quality bytes tallied at cycles >= 256 (they bypass the LDS image and its row sums)
    uint32_t tot = 0, c20 = 0, c30 = 0;
};

// One symbol at one cycle: LDS image for cycles < 256, the global matrix beyond.
// A quality byte >= 128 has no row (the reference would write out of bounds): flagged.
template <bool kQual>
__device__ __forceinline__ void bump(uint32_t *hist, u64 *__restrict__ gacc, uint32_t byte, uint32_t pos, uint32_t &bad,
                                     HiTot &hi)
{
    if (kQual && byte >= HPN_QUAL_ROWS) {
        bad = 1;
        return;
    }
    const uint32_t row = kQual ? byte : nuc_code(byte);
    if (pos < (uint32_t)kLdsCycles) {
        atomicAdd(&hist[lds_word(row, pos)], 1u);
    } else {
        atomicAdd(&gacc[row * HPN_LEN_BINS + pos], (u64)1);
        if (kQual) hi.tot += 1, hi.c20 += byte >= 53, hi.c30 += byte >= 63;
    }
}

// 4 bytes from any address (hardware unaligned access: one global_load_dword).
__device__ __forceinline__ uint32_t load_unaligned4(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// Equal-length chunk, 16 <= len0 <= 256: cnt reads of len0 bytes starting at arr + base_off.
// Work item = (read r, group j) = cycles 4j..4j+3 of read r, one unaligned dword load (256 B per
// wave-instruction: four times the bytes in flight of byte loads, which ran into the per-CU limit on
// outstanding requests).  The workgroup takes floor(1024 / ngr) whole reads per round, so a lane keeps its
// group j for the whole chunk: its LDS column addresses are computed once, its load address advances by a
// constant, and the only per-item arithmetic left is one add and one compare (PMC, round 2: the item-index
// arithmetic of lanes that changed (r, j) every item was 5 of the kernel's 7.4 VALU operations per byte,
// and VALU issue, not the LDS, was what the kernel waited for).
// kPartial: len0 % 8 != 0 (the chunk's reads end in a partial group); kRot: short reads (fewer than 16 groups): the order in
// which a lane takes the four bytes of a dword is rotated by its read (see below); kSpanRound: 8-byte items per set
template <bool kQual, bool kPartial, bool kRot, int kSpanRound>
__device__ __forceinline__ void stream_uniform(HistLds &s, const uint8_t *arr, uint64_t base_off, uint64_t arr_end, uint32_t cnt,
                                               uint32_t len0, uint32_t &bad)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x2 item_t;
    uint32_t *hist = kQual ? s.qh : s.nh;
    const uint8_t *p0 = arr + base_off;
    // Work item = (read r, 8-cycle group j) = cycles 8j..8j+7 of the read = two 4-cycle groups 2j, 2j+1 = ONE unaligned
    // 8-byte load.  (The CU holds a bounded number of vector-memory INSTRUCTIONS in flight, not of bytes: with dword items
    // the kernel ran at the same 4.8 TB/s with and without its LDS atomics, whatever the number of loads a wave kept
    // outstanding.)  The last group holds only len0 % 8 cycles when that is not 0: its load reaches into the next read's
    // first bytes, which are masked.
    const uint32_t ngr = (len0 + 7u) >> 3;
    // reads per round (>= 32: ngr <= 32).  (Items that start off a dword boundary -- any length that is not a multiple of
    // 4 -- were also tried as three ALIGNED dwords per lane shifted with v_alignbyte: the same 6.1-6.2 ms at 150 and 151 bp.)
    const uint32_t rpr = kHistThreads / ngr;
    const uint32_t lr = threadIdx.x / ngr, j = threadIdx.x - lr * ngr;   // the one division per chunk
    const uint32_t nvalid = min(8u, len0 - 8u * j);       // bytes of this lane's group that belong to the read
    const bool lane_on = lr < rpr;                        // the last 1024 - rpr * ngr lanes have no item
    // byte b = 4e + k of every item of this lane goes to word row*256 + k*64 + 32e + j (col_of()): the lanes of a read hit
    // ngr different banks in every one of the eight adds, and the three or four reads under a wave stay within the four
    // lanes per bank that cost nothing (scripts/micro/lds_atomic.hip)
    // With fewer than 16 groups per read more than four reads lie under a wave, and their lanes of one group would meet on one
    // bank in every add (5 lanes per bank at 100 bp).  The column of byte k is shifted by 8k banks, and a lane takes the bytes
    // of a dword in the order k = (t + rot) % 4, t = 0..3, with rot = its read's number % 4: the reads of a wave then work on
    // four different windows of banks at a time.  (One more VALU instruction per byte: the shift is no longer a constant.)
    const uint32_t rot = kRot ? (lr & 3u) : 0u;
    uint32_t col[8], bmask[8], shift[8];                  // slot b8 = 4e + t; bmask: the byte's 7 (8) bits, or 0 for a byte behind the read
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
        const uint32_t e = b8 >> 2, k = ((uint32_t)(b8 & 3) + rot) & 3u;
        const bool mine = !kPartial || 4u * e + k < nvalid;
        shift[b8] = 8u * k;
        bmask[b8] = mine ? (kQual ? 0x7fu : 0xffu) : 0u;
        col[b8] = mine ? 64u * k + 32u * e + ((j + 8u * k) & 31u) : (kQual ? kJunkQ : kJunkN) + (threadIdx.x & 63u);
    }
    // The loads go through a buffer descriptor of the chunk: address = descriptor base + lane offset (VGPR) + m * step
    // (SGPR), so a set of kSpanRound loads costs ONE vector add (the 64-bit address arithmetic, the per-item compare and
    // select of flat loads were 1.3 of the kernel's 5.5 VALU operations per byte, and with four cycles per wave64
    // instruction VALU issue was what the kernel waited for).  Lanes without items carry an offset beyond the
    // descriptor: the hardware drops their loads.
    const uint32_t rounds = (cnt + rpr - 1) / rpr;        // uniform
    uint32_t left = 0;                                    // items this lane still has to fetch
    if (lane_on && lr < cnt) left = rounds - (lr + (rounds - 1u) * rpr >= cnt ? 1u : 0u);
    const uint64_t chunk_bytes = (uint64_t)cnt * len0, avail = arr_end - base_off;
    // The partial group of the batch's very last read cannot be loaded whole (it would reach past the batch): that
    // one item is left out of the sets and tallied byte by byte after the loop.
    bool last_mine = false;
    const uint64_t p0u = (uint64_t)(uintptr_t)p0;         // (the same in every lane: into SGPRs)
    const uint32_t last_off = (cnt - 1u) * len0 + 8u * j;
    if (kPartial && left && j == ngr - 1u && lr + (rounds - 1u) * rpr == cnt - 1u && (uint64_t)last_off + 8u > avail) last_mine = true, --left;
    const uint32_t nrec = (uint32_t)min(avail, chunk_bytes + 7u);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(p0u >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)p0u)),
        0, (int)__builtin_amdgcn_readfirstlane(nrec), 0x00020000);
    const uint32_t step = __builtin_amdgcn_readfirstlane(rpr * len0);   // bytes between a lane's items of consecutive rounds
    constexpr uint32_t kNowhere = 0x7ff00000u;            // an offset no descriptor of a chunk reaches
    uint32_t voff = left ? lr * len0 + 8u * j : kNowhere; // chunk-relative byte offset of this lane's next item
    constexpr int kSets = HPN_SETS;
    item_t v[kSets][kSpanRound];
    uint32_t nv[kSets] = {};                              // items of the set that exist (below kSpanRound only in a lane's last set)
    auto fetch = [&](item_t (&vs)[kSpanRound], uint32_t &n) {
        n = min(left, (uint32_t)kSpanRound);
        left -= n;
        // items beyond a lane's last one lie beyond the descriptor's end (the range check covers voffset + soffset, and a
        // load that straddles the end is dropped whole: scripts/micro/buffer_range.hip): no memory is touched for them
#pragma unroll
#ifdef DIAG_NOLOAD
        for (int m = 0; m < kSpanRound; ++m) vs[m][0] = (voff + m * 0x01030507u) & 0x3f3f3f3fu, vs[m][1] = (voff + m * 0x03050701u) & 0x3f3f3f3fu;
#else
        for (int m = 0; m < kSpanRound; ++m) vs[m] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, n ? voff : kNowhere, m * step, 0);
#endif
        voff += kSpanRound * step;
    };
    // A quality byte >= 128 has no row: bit 7 of every byte is OR-ed into `seen` (checked once
    // per chunk; the batch is then rejected) and the row index is masked to 7 bits so that the
    // LDS address stays in range: no compare, no exec-mask juggling per byte.
    uint32_t seen = 0, sink = 0;
    // Nucleotides: five rows only.  A lane keeps the counts of its eight cycles in registers, five 6-bit counters to a word
    // (what a base adds comes out of an LDS table: 1 << 6 * code), and adds them to the image every 56 items: one LDS read per
    // base instead of a read and an atomic (the nucleotide pass took 7.6 ms beside the quality pass's 6.0 for the same
    // bytes).  A slot behind the read (partial group) counts the next read's bytes and is dropped when the counters are poured.
    uint32_t packed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto pour = [&]() {
#pragma unroll
        for (int b8 = 0; b8 < 8; ++b8) {
            if (!kPartial || bmask[b8]) {
#pragma unroll
                for (int c = 0; c < HPN_NUC_CODES; ++c) {
                    const uint32_t v = (packed[b8] >> (6 * c)) & 63u;
                    if (v) atomicAdd(&hist[c * kRowWords + col[b8]], v);
                }
            }
            packed[b8] = 0;
        }
    };
    auto tally_item = [&](const item_t d) {
        if (kQual) seen |= d[0] | d[1];   // (the bytes behind a partial group are the next read's: quality bytes as well)
        if (!kQual) {
#pragma unroll
            for (int b8 = 0; b8 < 8; ++b8) {
                const uint32_t w = d[b8 >> 2];
                uint32_t at;                                           // 4 * byte: its word in nlut6
                if (kRot) at = ((w >> shift[b8]) & 0xffu) << 2;
                else at = (b8 & 3) == 0 ? (w << 2) & 0x3fcu : (w >> (8 * (b8 & 3) - 2)) & 0x3fcu;
                packed[b8] += *(const uint32_t *)((const uint8_t *)s.nlut6 + at);
            }
            return;
        }
#pragma unroll
        for (int b8 = 0; b8 < 8; ++b8) {
            // bytes behind a partial group go to this lane's junk word behind the image (row 0): no branch per byte
            const uint32_t byte = (kRot ? d[b8 >> 2] >> shift[b8] : d[b8 >> 2] >> (8 * (b8 & 3))) & bmask[b8];
            const uint32_t row = kQual ? byte : (uint32_t)s.nlut[byte];
#ifdef DIAG_NOATOM
            sink += row * kRowWords + col[b8];
#else
            atomicAdd(&hist[row * kRowWords + col[b8]], 1u);
#endif
        }
    };
    auto tally = [&](const item_t (&vs)[kSpanRound], uint32_t n) {
        if (__ballot(n != (uint32_t)kSpanRound && n != 0u) == 0) {
            if (n) {
#pragma unroll
                for (int m = 0; m < kSpanRound; ++m) tally_item(vs[m]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < kSpanRound; ++m) {
                if ((uint32_t)m >= n) break;             // a lane's items of a set are its first n
                tally_item(vs[m]);
            }
        }
    };
    // a ring of kSets sets: kSets - 1 of them in flight while one is tallied
    const uint32_t nsets = (rounds + kSpanRound - 1) / kSpanRound;       // uniform
    uint32_t fetched = 0;
#pragma unroll
    for (int t = 0; t < kSets - 1; ++t)
        if (fetched < nsets) fetch(v[t], nv[t]), ++fetched;
    uint32_t since = 0;                                   // sets tallied into `packed` (a 6-bit counter holds 63)
    for (uint32_t i = 0; i < nsets; i += kSets) {
#pragma unroll
        for (int t = 0; t < kSets; ++t) {
            if (i + t >= nsets) break;
            if (fetched < nsets) fetch(v[(t + kSets - 1) % kSets], nv[(t + kSets - 1) % kSets]), ++fetched;
            tally(v[t], nv[t]);
            if (!kQual && (since += kSpanRound) > 63u - kSpanRound) pour(), since = 0;
        }
    }
    if (!kQual) pour();
    if (kPartial && last_mine) {                          // at most one lane of the batch's last chunk
        for (uint32_t b8 = 0; b8 < nvalid; ++b8) {
            const uint32_t byte = p0[last_off + b8];
            if (kQual) seen |= byte;
            const uint32_t row = kQual ? (byte & 0x7fu) : (uint32_t)s.nlut[byte];
            atomicAdd(&hist[row * kRowWords + 64u * (b8 & 3u) + 32u * (b8 >> 2) + ((j + 8u * (b8 & 3u)) & 31u)], 1u);
        }
    }
    if (seen & 0x80808080u) bad = 1;
    if (sink == 0x12345u) bad = 1;
}

template <bool kQual, int kSp>
__device__ __forceinline__ void tally_uniform(HistLds &s, const uint8_t *arr, uint64_t base_off, uint64_t arr_end, uint32_t cnt, uint32_t len0,
                                              uint32_t &bad)
{
    if (len0 <= 120u) {   // fewer than 16 groups of 8 cycles
        (len0 & 7u) ? stream_uniform<kQual, true, true, kSp>(s, arr, base_off, arr_end, cnt, len0, bad)
                    : stream_uniform<kQual, false, true, kSp>(s, arr, base_off, arr_end, cnt, len0, bad);
    } else {
        (len0 & 7u) ? stream_uniform<kQual, true, false, kSp>(s, arr, base_off, arr_end, cnt, len0, bad)
                    : stream_uniform<kQual, false, false, kSp>(s, arr, base_off, arr_end, cnt, len0, bad);
    }
}

// Any chunk: aligned vectors of the byte range, each located by binary search.
// Records are [loff[lo-1], loff[lo]).
template <bool kQual>
__device__ __forceinline__ void stream_ragged(HistLds &s, u64 *__restrict__ gacc, const uint8_t *arr,
                                              uint64_t base_off, uint32_t cnt, uint32_t &bad, HiTot &over)
{
    const uint32_t B = s.loff[cnt];
    if (B == 0) return;
    uint32_t *hist = kQual ? s.qh : s.nh;
    const uint8_t *p0 = arr + base_off;
    const int a0 = (int)((uintptr_t)p0 & 15);
    const u32 *vec = reinterpret_cast<const u32 *>(p0 - a0);
    const uint32_t nvec = (uint32_t)((a0 + B + 15) >> 4);
#pragma unroll 1
    for (uint32_t j = threadIdx.x; j < nvec; j += kHistThreads) {
        const u32 v = load_stream16(vec + j);
        const int rel = (int)(16 * j) - a0;
        const int k0 = rel < 0 ? -rel : 0;
        const int k1 = min(16, (int)B - rel);
        uint32_t b = (uint32_t)(rel + k0);
        uint32_t lo = 1, hi = cnt;
        while (lo < hi) {  // first boundary above b
            const uint32_t mid = (lo + hi) >> 1;
            if (s.loff[mid] > b) hi = mid;
            else lo = mid + 1;
        }
        uint32_t nxt = s.loff[lo], pos = b - s.loff[lo - 1];
#pragma unroll 1
        for (int k = k0; k < k1; ++k) {
            if (b == nxt) {  // crossed into the next non-empty record
                do { ++lo; } while (s.loff[lo] <= b);
                nxt = s.loff[lo];
                pos = 0;
            }
            const uint32_t wd = k < 4 ? v[0] : k < 8 ? v[1] : k < 12 ? v[2] : v[3];
            bump<kQual>(hist, gacc, (wd >> (8 * (k & 3))) & 0xffu, pos, bad, over);
            ++b, ++pos;
        }
    }
}

// LDS image -> global matrix; returns this lane's share of (sum, sum over rows >= 53, >= 63).
__device__ __forceinline__ void hist_flush(const uint32_t *lds, int rows, u64 *__restrict__ gacc, u64 &tot, u64 &t20,
                                           u64 &t30)
{
    for (int w = threadIdx.x; w < rows * kRowWords; w += kHistThreads) {
        const uint32_t v = lds[w];
        if (v) {
            const int r = w / kRowWords, c = w - r * kRowWords;  // c = (cycle & 3) * 64 + (group & 1) * 32 + (group >> 1), group = cycle >> 2
            const int k = c >> 6, jj = ((c & 31) - 8 * k) & 31;   // c = k*64 + (group & 1)*32 + ((group >> 1) + 8k) % 32, cycle = 4*group + k
            atomicAdd(&gacc[r * HPN_LEN_BINS + 4 * (2 * jj + ((c >> 5) & 1)) + k], (u64)v);
            tot += v;
            if (r >= 53) t20 += v;
            if (r >= 63) t30 += v;
        }
    }
}

template <bool kQualHist, bool kNucHist>
__global__ __launch_bounds__(kHistThreads) void k_tally_hist(const uint8_t *__restrict__ qual,
                                                            const uint8_t *__restrict__ base,
                                                            const uint64_t *__restrict__ off, uint64_t n,
                                                            u64 *__restrict__ acc, uint32_t big)
{
    __shared__ HistLds s;
    constexpr int kSp = (kQualHist && kNucHist) ? kSpanBoth : kSpanOne;
    const int tid = threadIdx.x;
    for (int i = tid; i < HPN_QUAL_ROWS * kRowWords; i += kHistThreads) s.qh[i] = 0;
    for (int i = tid; i < HPN_NUC_CODES * kRowWords; i += kHistThreads) s.nh[i] = 0;
    for (int i = tid; i <= HPN_LEN_BINS; i += kHistThreads) s.lhist[i] = 0;
    if (tid < 256) s.nlut[tid] = (uint8_t)nuc_code((uint32_t)tid);
    if (tid < 256) s.nlut6[tid] = 1u << (6u * nuc_code((uint32_t)tid));
    __syncthreads();

    uint32_t bad = 0;
    u64 *gq = acc + HPN_TALLY_W_QUAL, *gn = acc + HPN_TALLY_W_NUC;
    HiTot hi;
    const uint64_t arr_end = off[n];          // first byte offset that is not the batch's
    // A workgroup takes `big` chunks of kHistRecs records at a time.  When they all have one length (the normal case: checked
    // from the offsets in registers, no LDS) they are streamed as ONE chunk: the offsets' round trip, the barriers and the
    // fill and drain of the load pipeline are paid once per big * 4096 records.  Otherwise chunk by chunk, boundaries in LDS.
    const uint64_t span = (uint64_t)big * kHistRecs;
    const uint64_t nspan = (n + span - 1) / span;
    for (uint64_t sp = blockIdx.x; sp < nspan; sp += gridDim.x) {
        const uint64_t q0 = sp * span;
        const uint32_t qcnt = (uint32_t)min(span, n - q0);
        const uint64_t q_off = off[q0];
        const uint64_t qlen0 = off[q0 + 1] - q_off;
        bool one_len = qlen0 >= 16 && qlen0 <= (uint64_t)kLdsCycles;
        if (one_len) {
            for (uint32_t i = tid; i < qcnt; i += kHistThreads) one_len = one_len && off[q0 + i + 1] - off[q0 + i] == qlen0;
        }
        if (__syncthreads_and((int)one_len)) {
            const uint32_t len0 = (uint32_t)qlen0;
            if (tid == 0) atomicAdd(&s.lhist[len0], qcnt);
            if (kQualHist) tally_uniform<true, kSp>(s, qual, q_off, arr_end, qcnt, len0, bad);
            if (kNucHist) tally_uniform<false, kSp>(s, base, q_off, arr_end, qcnt, len0, bad);
            continue;
        }
        for (uint64_t r0 = q0; r0 < q0 + qcnt; r0 += kHistRecs) {
            const uint32_t cnt = (uint32_t)min((uint64_t)kHistRecs, q0 + qcnt - r0);
            const uint64_t base_off = off[r0];
            // chunk-relative boundaries; a chunk spans < 4096*512 bytes inside the domain
            for (uint32_t i = tid; i <= cnt; i += kHistThreads) {
                const uint64_t d = off[r0 + i] - base_off;
                s.loff[i] = d > 0x7fffffffull ? 0x7fffffffu : (uint32_t)d;
            }
            __syncthreads();
            const uint32_t len0 = s.loff[1];
            bool all_same = true;
            for (uint32_t i = tid; i < (uint32_t)kHistRecs; i += kHistThreads) {  // same trip count in every lane (ballots inside)
                const bool valid = i < cnt;
                uint32_t len = 0;
                if (valid) {
                    len = s.loff[i + 1] - s.loff[i];
                    if (len >= HPN_LEN_BINS) len = HPN_LEN_BINS, bad = 1;
                    all_same = all_same && len == len0;
                }
                hist_len(s.lhist, valid, len);
            }
            // an over-long record poisons position tracking: stop tallying bytes, the batch
            // is rejected as a whole (HPN_E_DOMAIN) once `bad` is seen
            if (!__syncthreads_or((int)bad)) {
                const bool uniform = __syncthreads_and((int)all_same) && len0 >= 16 && len0 <= (uint32_t)kLdsCycles;
                if (uniform) {
                    if (kQualHist) tally_uniform<true, kSp>(s, qual, base_off, arr_end, cnt, len0, bad);
                    if (kNucHist) tally_uniform<false, kSp>(s, base, base_off, arr_end, cnt, len0, bad);
                } else {
                    if (kQualHist) stream_ragged<true>(s, gq, qual, base_off, cnt, bad, hi);
                    if (kNucHist) stream_ragged<false>(s, gn, base, base_off, cnt, bad, hi);
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    u64 tot = hi.tot, t20 = hi.c20, t30 = hi.c30, ntot = 0, n20 = 0, n30 = 0;
    if (kQualHist) hist_flush(s.qh, HPN_QUAL_ROWS, gq, tot, t20, t30);
    if (kNucHist) hist_flush(s.nh, HPN_NUC_CODES, gn, ntot, n20, n30);

#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        tot += __shfl_xor(tot, o, kWave);
        t20 += __shfl_xor(t20, o, kWave);
        t30 += __shfl_xor(t30, o, kWave);
    }
    bad = wave_or(bad);
    if (lane_id() == 0) {
        s.red[0][wave_id()] = tot;
        s.red[1][wave_id()] = t20;
        s.red[2][wave_id()] = t30;
        if (bad) atomicAdd(&acc[HPN_TALLY_W_BAD], (u64)1);
    }
    __syncthreads();
    if (tid == 0 && kQualHist) {
        u64 a = 0, b20 = 0, b30 = 0;
        for (int w = 0; w < kHistWaves; ++w) a += s.red[0][w], b20 += s.red[1][w], b30 += s.red[2][w];
        if (a) atomicAdd(&acc[HPN_TALLY_W_TOTAL], a);
        if (b20) atomicAdd(&acc[HPN_TALLY_W_Q20], b20);
        if (b30) atomicAdd(&acc[HPN_TALLY_W_Q30], b30);
    }
    // SeqLen comes from this kernel only when it replaces the flat scan (quality matrix
    // requested); a nucleotide-only launch runs beside k_tally_scan.
    for (int i = tid; i <= HPN_LEN_BINS; i += kHistThreads) {
        const uint32_t h = s.lhist[i];
        if (h && (kQualHist || i == HPN_LEN_BINS))
            atomicAdd(&acc[i < HPN_LEN_BINS ? HPN_TALLY_W_SEQLEN + i : HPN_TALLY_W_BAD], (u64)h);
    }
}

hipError_t launch_tally_hist(const uint8_t *d_qual, const uint8_t *d_base, const uint64_t *d_off, uint64_t n,
                             bool qual_hist, bool nuc_hist, u64 *d_acc, int n_cu, hipStream_t st)
{
    const uint64_t nchunk = (n + kHistRecs - 1) / kHistRecs;
    if (nchunk == 0) return hipSuccess;
    const unsigned grid = (unsigned)(nchunk < (uint64_t)n_cu ? nchunk : (uint64_t)n_cu);  // LDS: one image per CU
    // chunks per workgroup turn: up to kHistBig, fewer when the batch would otherwise leave workgroups without a turn or
    // make the last round of turns a large part of the whole
    uint32_t big = HPN_HIST_BIG;
    while (big > 1 && nchunk < (uint64_t)8 * big * grid) big >>= 1;
    if (const char *e = test_env("HPN_K1L_BIG")) {   // tests: turns of several chunks on batches that would not get them
        const int v = atoi(e);
        if (v >= 1 && v <= 64) big = (uint32_t)v;
    }
    if (qual_hist && nuc_hist)
        hipLaunchKernelGGL((k_tally_hist<true, true>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc, big);
    else if (qual_hist)
        hipLaunchKernelGGL((k_tally_hist<true, false>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc, big);
    else
        hipLaunchKernelGGL((k_tally_hist<false, true>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc, big);
    return hipGetLastError();
}

}  // namespace hpn
