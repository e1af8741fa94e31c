// inflate_core.hpp -- the wave-level DEFLATE machinery shared by the BGZF-block kernel (bgzf_inflate.hip) and the
// symbolic single-member-gzip kernel (gz_inflate.hip): LDS layout, wave-uniform bit reader over an LDS input ring,
// canonical Huffman table construction (inftrees.c's layout), table lookup.  See bgzf_inflate.hip for the design.
#pragma once
#include "common.hpp"

namespace hpn {

// The decoder's LDS footprint and register budget decide how many single-wave decoders a CU holds (scripts/micro/lds_occupancy.hip:
// LDS comes in granules of 1,280 bytes): same-session A/B builds set these from the command line (scripts/ab_build.sh).
#ifndef HPN_INF_RING
#define HPN_INF_RING 512
#endif
#ifndef HPN_INF_LIT
#define HPN_INF_LIT 852
#endif
#ifndef HPN_INF_WAVES
#define HPN_INF_WAVES 24
#endif
#ifndef HPN_INF_EU
#define HPN_INF_EU 6
#endif
constexpr uint32_t kRing = HPN_INF_RING;      // compressed-input ring (bytes), refilled by halves
// Root bits and table sizes (root + sub-tables).  The sub-tables are sized like zlib's (inftrees.c: per root prefix, for the
// longest code under it), so its bound holds: `enough 286 9 15` = 852 entries for the literal/length table; the distance table's
// worst case is exactly 400 (30 symbols, 8-bit root: every length distribution enumerated).  A 9-bit literal root instead of
// 10 bits (round 3: six LDS granules of 1,280 bytes instead of seven, 20 decoders per CU instead of 18); round 4: the literal
// table cut to zlib's bound (852 entries, not 1,024) and the input ring to 512 bytes -- 6,000 bytes = five granules -- and the
// kernels held to 80 registers: 24 decoders per CU, 6,144 on the chip (same-session A/B, scripts/ab_inflate.sh: gz 47.0 -> 49.8,
// BGZF 43.4 -> 46.2 GB/s; 20 % more decoders buy 6 %: the bound is instruction issue, not latency).  Codes longer
// than the root are looked up a second time by the lanes that hold them (window()).
constexpr uint32_t kLitRoot = 9, kDistRoot = 8;
constexpr uint32_t kLitSize = HPN_INF_LIT, kDistSize = 256 + 144;

// table entry: [31:16] value, [15:8] extra-bit count (or sub-table index bits), [7:4] kind, [3:0] code bits
enum { kLit = 0, kLit2 = 1, kLen = 2, kEob = 3, kSub = 4, kDist = 5, kBad = 15 };  // literal kinds first: one compare
__device__ __forceinline__ uint32_t mk(uint32_t value, uint32_t extra, uint32_t kind, uint32_t nbits)
{
    return value << 16 | extra << 8 | kind << 4 | nbits;
}

constexpr uint32_t kQueue = 128;               // decoded symbols waiting for their output positions (emit())
struct InfLds {
    uint32_t ring[kRing / 4 + 2];   // (+ copies of words 0 and 1 behind the last: a window reads three neighbouring words; FIRST in the
                                    // struct: a two-word LDS read has 8-bit offsets, the tables' single reads take theirs in 16 bits)
    uint32_t lit[kLitSize];
    uint32_t dist[kDistSize];
    union {
        struct {                    // while a block's code tables are built
            uint8_t lens[384];      // code lengths being assembled (19 code-length codes | 286 + 30 lengths from offset 32)
            uint16_t count[16], first[16], next[16];
        };
        uint32_t queue[kQueue];     // while its symbols are decoded (empty between blocks)
    };
    uint32_t spot[kWave];           // assemble(): which symbol starts at which output position of a chunk
};

constexpr uint32_t kInflateWavesPerCu = HPN_INF_WAVES;
static_assert(((sizeof(InfLds) + 1279) / 1280) * 1280 * kInflateWavesPerCu <= 160 * 1024, "LDS granules of the decoders of one CU");

struct Bits {  // wave-uniform bit reader over the LDS ring
    u64 bb = 0;
    uint32_t bc = 0;       // valid bits in bb
    uint32_t in_pos = 0;   // compressed bytes moved into bb
    uint32_t filled = 0;   // compressed bytes staged into the ring
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
// The decoder's state is the same in every lane and its loops are meant to run on the scalar unit (SGPR arithmetic, s_cbranch
// instead of exec masks).  The compiler does that only for values it can PROVE uniform, and one doubtful value in a loop's cycle
// (the error code, behind the table builds' lane-strided loops) moves the whole cycle into vector registers and its branches
// under exec masks.  pin() re-states the uniformity at the head of a loop (v_readfirstlane; nothing for a value in an SGPR).
__device__ __forceinline__ void pin(Bits &b) { b.bb = uni64(b.bb), b.bc = uni(b.bc), b.in_pos = uni(b.in_pos), b.filled = uni(b.filled); }
__device__ __forceinline__ void pin(uint32_t &v) { v = uni(v); }

// stage the next kRing/2 bytes of the block's compressed data (16 B per lane)
__device__ __forceinline__ void stage(InfLds &s, Bits &b, const uint8_t *__restrict__ in, uint32_t in_len)
{
    const uint32_t at = b.filled + 16u * (uint32_t)lane_id();
    if (16u * (uint32_t)lane_id() < kRing / 2) {
        u32 v = {0, 0, 0, 0};
        if (at < in_len + 16u) __builtin_memcpy(&v, in + at, 16);  // the buffer is padded by the host
        *(u32 *)((uint8_t *)s.ring + (at & (kRing - 1))) = v;
        if ((at & (kRing - 1)) == 0) s.ring[kRing / 4] = v[0], s.ring[kRing / 4 + 1] = v[1];
    }
    b.filled += kRing / 2;
}

__device__ __forceinline__ void refill(InfLds &s, Bits &b, const uint8_t *__restrict__ in, uint32_t in_len)
{
    if (b.bc > 32u) return;
    if (b.filled - b.in_pos < 8u + kRing / 4) {
        if (b.filled < in_len + 8u) stage(s, b, in, in_len);
    }
    const uint32_t p = b.in_pos;
    const uint32_t w0 = s.ring[(p >> 2) & (kRing / 4 - 1)], w1 = s.ring[((p >> 2) + 1) & (kRing / 4 - 1)];
    const uint32_t v = uni((uint32_t)((((u64)w1 << 32) | w0) >> (8u * (p & 3u))));
    b.bb |= (u64)v << b.bc;
    b.bc += 32u;
    b.in_pos = p + 4u;
}
__device__ __forceinline__ uint32_t peek(const Bits &b, uint32_t n) { return (uint32_t)b.bb & ((1u << n) - 1u); }
__device__ __forceinline__ void drop(Bits &b, uint32_t n) { b.bb >>= n, b.bc -= n; }
__device__ __forceinline__ uint32_t take(Bits &b, uint32_t n)
{
    const uint32_t v = peek(b, n);
    drop(b, n);
    return v;
}

__device__ __forceinline__ uint32_t rev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32u - len); }

// Canonical Huffman table from s.lens[base .. base+n): root-bit primary table, sub-tables for
// longer codes sized like zlib's (per prefix, for the longest code under it).  All lanes run
// this redundantly on uniform values; stores of the same value to the same address by every
// lane are intended.  payload(sym, nbits) supplies the entry.  false: over-subscribed or
// (beyond what zlib accepts) incomplete code, or a table that does not fit.
template <typename F>
__device__ bool build(InfLds &s, uint32_t *tab, uint32_t tab_size, uint32_t root, uint32_t base, uint32_t n, bool allow_single,
                      F payload)
{
    const int lane = lane_id();
    if (lane < 16) s.count[lane] = 0;
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t l = s.lens[base + i];
        if (l) s.count[l] = (uint16_t)(s.count[l] + 1);
        maxlen = l > maxlen ? l : maxlen;
    }
    uint32_t code = 0, left = 1;
    bool over = false;
    for (uint32_t l = 1; l <= 15; ++l) {
        const uint32_t c = s.count[l];
        left <<= 1;
        if (c > left) over = true;
        left -= over ? 0 : c;
        code = (code + (l > 1 ? s.count[l - 1] : 0)) << 1;
        s.first[l] = (uint16_t)code;
        s.next[l] = (uint16_t)code;
    }
    if (over) return false;
    if (left != 0 && !(allow_single && maxlen <= 1)) return false;  // inftrees.c: incomplete only with max == 1 (or no code at all)
    for (uint32_t i = (uint32_t)lane; i < tab_size; i += kWave) tab[i] = mk(0, 0, kBad, 0);
    if (maxlen == 0) return true;
    uint32_t sub_next = 1u << root;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t l = s.lens[base + i];
        if (!l) continue;
        const uint32_t c = s.next[l];
        s.next[l] = (uint16_t)(c + 1);
        const uint32_t r = rev(c, l);
        if (l <= root) {
            const uint32_t e = payload(i, l);
            for (uint32_t k = r + ((uint32_t)lane << l); k < (1u << root); k += (uint32_t)kWave << l) tab[k] = e;
        } else {
            const uint32_t prefix = r & ((1u << root) - 1u), top = c >> (l - root);  // the code's first root bits
            uint32_t pe = tab[prefix];
            if (((pe >> 4) & 15u) != kSub) {
                // codes under one prefix are consecutive in canonical order with non-decreasing
                // lengths: the sub-table is as wide as the longest length that reaches this prefix
                uint32_t sb = l - root;
                for (uint32_t m = maxlen; m > l; --m) {
                    const uint32_t cnt = s.count[m];
                    if (cnt && top >= ((uint32_t)s.first[m] >> (m - root)) && top <= (((uint32_t)s.first[m] + cnt - 1u) >> (m - root))) {
                        sb = m - root;
                        break;
                    }
                }
                if (sub_next + (1u << sb) > tab_size) return false;
                pe = mk(sub_next, sb, kSub, root);
                tab[prefix] = pe;
                sub_next += 1u << sb;
            }
            const uint32_t sb = (pe >> 8) & 255u, so = pe >> 16;
            const uint32_t e = payload(i, l - root);
            const uint32_t hi = r >> root, step = 1u << (l - root);
            for (uint32_t k = hi + ((uint32_t)lane * step); k < (1u << sb); k += (uint32_t)kWave * step) tab[so + k] = e;
        }
    }
    return true;
}

static __device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t lit_payload(uint32_t sym, uint32_t nbits)
{
    if (sym < 256u) return mk(sym, 0, kLit, nbits);
    if (sym == 256u) return mk(0, 0, kEob, nbits);
    if (sym < 286u) {  // RFC 1951 3.2.5, in closed form (a table in global memory would cost a load per symbol)
        const uint32_t k = sym - 257u;
        if (k < 8u) return mk(3u + k, 0, kLen, nbits);
        if (k == 28u) return mk(258u, 0, kLen, nbits);
        const uint32_t xb = (k - 4u) >> 2;
        return mk(3u + ((4u + (k & 3u)) << xb), xb, kLen, nbits);
    }
    return mk(0, 0, kBad, nbits);
}
__device__ __forceinline__ uint32_t dist_payload(uint32_t sym, uint32_t nbits)
{
    if (sym < 30u) {
        if (sym < 4u) return mk(1u + sym, 0, kDist, nbits);
        const uint32_t xb = (sym - 2u) >> 1;
        return mk(1u + ((2u + (sym & 1u)) << xb), xb, kDist, nbits);
    }
    return mk(0, 0, kBad, nbits);
}

__device__ __forceinline__ uint32_t lookup(const uint32_t *tab, uint32_t root, Bits &b)
{
    uint32_t e = uni(tab[peek(b, root)]);
    if (((e >> 4) & 15u) == kSub) {
        drop(b, root);
        e = uni(tab[(e >> 16) + peek(b, (e >> 8) & 255u)]);
    }
    drop(b, e & 15u);
    return e;
}

// ---- the symbols of a block, 64 bit offsets at a time ----------------------------------------------------------------------
// The serial loop (refill; look the next code up in LDS; wait; readfirstlane; branch; store) is one instruction stream per
// wave, and with 18 single-wave decoders per CU it is the CU's ONE scalar unit that has to carry them: 42 scalar
// instructions per literal code kept it 70 % busy (SQ_ACTIVE_INST_SCA / SQ_BUSY_CU_CYCLES), and neither moving the loop's
// arithmetic between the vector and the scalar unit, nor dropping the per-literal store, nor taking the LDS round trip off
// the chain changed the kernel's time (round 3 A/B runs, profiles/r03/inflate_ab.txt).  So the work per symbol is moved to
// the lanes, which were idle:
//   * lane k holds the 64 bits of the stream from bit k of the window on (three ring words, two funnel shifts) and works out
//     the symbol that WOULD start there, all by itself: literal(s), or length + distance (both codes, both runs of extra bits:
//     at most 15 + 5 + 15 + 13 = 48 bits), and where the code after it starts;
//   * the wave follows the true chain of symbol starts through those offsets -- v_readlane + s_bitset1 per symbol, nothing else;
//   * the lanes on the chain append their symbols to a queue in LDS (one word each: units of output | literals or distance).
// A window ends with the first symbol that STARTS at or behind bit 64 (round 5; until then a match had to lie inside the 64
// lanes with both its codes, because the distance code was fetched from the lane where it starts: windows took ~55 bits).
// Output positions are NOT worked out per window: when 64 symbols are queued they leave together (emit()): one prefix sum over
// 64 symbols, and assemble() -- whose cost is per 64 output positions -- runs on full chunks instead of one window's ~14 bytes
// (round 4: a window with a match paid for a whole chunk; that was 80 of its ~250 instructions).
struct Pos {             // consumed position in the chunk: `bit` (0..7) bits into byte `byte`
    uint32_t byte, bit;
};
__device__ __forceinline__ Pos pos_of(const Bits &b)
{
    return Pos{b.in_pos - ((b.bc + 7u) >> 3), (0u - b.bc) & 7u};
}
// the serial reader, restarted at p (the ring holds the bytes around it)
__device__ __forceinline__ void seek(InfLds &s, Bits &b, Pos p, const uint8_t *__restrict__ in, uint32_t in_len)
{
    b.in_pos = p.byte, b.bb = 0, b.bc = 0;
    refill(s, b, in, in_len);
    drop(b, p.bit);
}
// bits [at, at + n) of v, n = 0 .. 31 (one v_bfe_u32 with offset and width in registers; the shift-and-mask form is two or three)
__device__ __forceinline__ uint32_t bits_of(uint32_t v, uint32_t at, uint32_t n) { return __builtin_amdgcn_ubfe(v, at, n); }
struct Win {
    uint32_t lo, hi;     // the 64 bits of the stream from bit (p + lane) on
    uint32_t el;         // the literal/length entry for them (a code longer than the root table already resolved)
};
// bitpos: the window's first bit, counted from the chunk's first byte (only its low 12 bits matter).  The ring holds the bytes
// the window reads, up to byte p.byte + 20: decode_symbols stages ahead of it.
__device__ __forceinline__ Win window(const InfLds &s, uint32_t bitpos)
{
    const uint32_t q = bitpos + (uint32_t)lane_id();
    const uint32_t i0 = (q >> 5) & (kRing / 4 - 1);
    const uint32_t w0 = s.ring[i0], w1 = s.ring[i0 + 1u], w2 = s.ring[i0 + 2u];   // (ring[kRing / 4 + k] = ring[k])
    Win w;
    w.lo = __builtin_amdgcn_alignbit(w1, w0, q);      // (the shift is q & 31)
    w.hi = __builtin_amdgcn_alignbit(w2, w1, q);
    w.el = s.lit[w.lo & ((1u << kLitRoot) - 1u)];
    // a code longer than the root table: the lane that holds one takes the second-level entry itself and counts the root's
    // bits into it -- the walk sees a plain entry of up to 15 bits (rare length codes and the 256-symbol alphabets of BAM blocks)
    if ((w.el & 0xf0u) == (kSub << 4)) w.el = s.lit[(w.el >> 16) + bits_of(w.lo, kLitRoot, (w.el >> 8) & 15u)] + kLitRoot;
    return w;
}
__device__ __forceinline__ uint32_t lane_of(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t from_lane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(l << 2), (int)v); }
// inclusive prefix sum over the lanes of a wave (DPP: shifts inside rows of 16, then row broadcasts; see wave_sum)
__device__ __forceinline__ uint32_t wave_prefix(uint32_t v)
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

// inclusive prefix maximum over the lanes of a wave (the same DPP moves as wave_prefix)
__device__ __forceinline__ uint32_t wave_prefix_max(uint32_t v)
{
#define HPN_DPP_MAX(ctrl, rows)                                                                    \
    {                                                                                              \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false); \
        v = t > v ? t : v;                                                                         \
    }
    HPN_DPP_MAX(0x111, 0xf)
    HPN_DPP_MAX(0x112, 0xf)
    HPN_DPP_MAX(0x114, 0xf)
    HPN_DPP_MAX(0x118, 0xf)
    HPN_DPP_MAX(0x142, 0xa)
    HPN_DPP_MAX(0x143, 0xc)
#undef HPN_DPP_MAX
    return v;
}

// The output of a batch of symbols, 64 positions at a time, one position per lane.  The symbols sit in their lanes (`mine`):
// rel = where the symbol's output starts (from sink.op), el = its word (the literals in [31:16]), dist = its distance, 0 for
// literals.  Every output position finds its symbol (the symbols' lane numbers written to their start
// positions in 64 words of LDS, then a prefix maximum), takes the symbol's words from its lane (ds_bpermute) and is a
// literal byte, or a copy of position - dist: of memory (sink.fetch: waits for this wave's stores first where they are
// in the way; in front of a gzip stretch: a history placeholder), or of a lane of this very chunk -- those are followed by
// pointer jumping, which is also what unrolls an overlapping match (distance < length: every position refers to the one
// `dist` before it, in the same match).  One store instruction per chunk.  The serial form (per match: three v_readlane, a
// dozen scalar compares and branches, a load, a wait and a store of its own) was half of k_bgzf_inflate's time on BAM blocks
// (timing-only builds: 128 ms with, 66 ms without the matches; profiles/r03/inflate_ab.txt).
template <typename Sink>
__device__ __forceinline__ void assemble(InfLds &s, Sink &sink, bool mine, uint32_t rel, uint32_t el, uint32_t dist, uint32_t total)
{
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t *spot = s.spot;
    const uint32_t words = rel | dist << 16;                        // rel < 64 * 258, dist <= 32768
    uint32_t carry = 0;                                             // the symbol the chunk before ended in
    for (uint32_t c0 = 0; c0 < total; c0 += (uint32_t)kWave) {
        spot[lane] = 0;
        if (mine && rel - c0 < (uint32_t)kWave) spot[rel - c0] = lane + 1u;
        // (what a lane reads next is what OTHER lanes of the wave wrote: without the fence the compiler forwards the lane's own 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t x = spot[lane];
        if (lane == 0 && carry > x) x = carry;
        const uint32_t own = wave_prefix_max(x);                    // 1 + the lane of the symbol this position belongs to
        carry = lane_of(own, kWave - 1);
        const uint32_t w = from_lane(words, own - 1u), e = from_lane(el, own - 1u);
        const uint32_t p = c0 + lane, i = p - (w & 0xffffu), d = w >> 16;
        const bool live = p < total, copy = live && d != 0;
        uint32_t val = (e >> (16u + 8u * (i & 1u))) & 255u;         // a literal: byte i of its word
        const int32_t from = (int32_t)(sink.op + p) - (int32_t)d;   // a copy: of this position (may lie in front of a gzip stretch)
        const bool near = copy && from >= (int32_t)(sink.op + c0);
#ifndef DIAG_NOFETCH
        if (__builtin_amdgcn_ballot_w64(copy && !near)) val = sink.fetch(copy && !near, from, sink.op + c0, val);
#endif
        uint32_t ref = near ? (uint32_t)from - (sink.op + c0) : ~0u;   // lane of this chunk still to be copied from; ~0: val is final
#ifdef DIAG_NOJUMP
        while (false) {
#else
        while (__builtin_amdgcn_ballot_w64(ref != ~0u)) {
#endif
            const uint32_t rv = from_lane(val, ref), rr = from_lane(ref, ref);
            if (ref != ~0u) {
                if (rr == ~0u) val = rv, ref = ~0u;
                else ref = rr;
            }
        }
#ifndef DIAG_NOPUT
        sink.put(live, sink.op + p, val);
#endif
    }
}

// The symbols decoded and not yet written: words in s.queue[(head + i) % kQueue], i < n.  A word: [8:0] the units of output
// (1 or 2: literals, [23:16] and [31:24]; 3..258: a match, its distance in [31:16]).
struct Queue {
    uint32_t head, n;
};

// The first `take` (<= 64) queued symbols leave.  -> 0 or the decoder's error code.
// The sink: op / out_len -- units written / allowed; lits(mine, two, at, w) -- the lanes for which `mine` holds store the one
// (two) literal(s) of THEIR word at `at`; in_reach(at, dist) -- a match at `at` may refer `dist` back; fetch / put: see assemble().
template <typename Sink>
__device__ __forceinline__ uint32_t emit(InfLds &s, Sink &sink, Queue &q, uint32_t take)
{
    const uint32_t lane = (uint32_t)lane_id();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (the words were written by other lanes)
    __builtin_amdgcn_wave_barrier();
    const bool mine = lane < take;
    const uint32_t w = mine ? s.queue[(q.head + lane) & (kQueue - 1u)] : 0u;
    const uint32_t units = w & 0x1ffu;
    const bool is_match = units >= 3u;
    const uint32_t upto = wave_prefix(units), total = lane_of(upto, kWave - 1), rel = upto - units;
    const uint32_t dist = is_match ? w >> 16 : 0u;
    q.head = (q.head + take) & (kQueue - 1u), q.n -= take;
    const u64 matches = __builtin_amdgcn_ballot_w64(is_match);
    if constexpr (Sink::kDry) sink.lits(mine && !is_match, units == 2u, 0u, w);   // nothing is stored: the sink only looks at the literals
    if (__builtin_amdgcn_ballot_w64(is_match && !sink.in_reach(sink.op + rel, dist))) return 14;
    if (sink.op + total > sink.out_len) return 12;
    if constexpr (!Sink::kDry) {
        if (matches == 0) sink.lits(mine, units == 2u, sink.op + rel, w);
#ifndef DIAG_NOASSEMBLE
        else assemble(s, sink, mine, rel, w, dist, total);
#endif
    }
    sink.op += total;
    return 0;
}

// One window.  -> how it ended; o = bits consumed (the symbol at o is still to be decoded, except behind an end-of-block code).
enum { kWinNext = 0, kWinEob = 1, kWinError = 3, kWinErrLen = 4 };
__device__ __forceinline__ uint32_t walk(InfLds &s, const Win &w, Queue &q, uint32_t &o)
{
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t kind = (w.el >> 4) & 15u, bits = w.el & 15u;
    // what the symbol starting here would be, and where the one behind it starts
    uint32_t nxt = lane + bits, word = (w.el & 0xffff0000u) | (kind + 1u);
    // (which lanes hold a symbol this window could take, as a lane mask in scalar registers from the start: a bool that is merged
    // behind the branch below comes back from the compiler as v_cndmask + v_cmp per use)
    u64 emit_mask = __builtin_amdgcn_ballot_w64(kind <= kLit2);
    const u64 len_lanes = __builtin_amdgcn_ballot_w64(kind == kLen);
    if (len_lanes) {                                  // (same in every lane: windows of literals skip this)
        const uint32_t xb = (w.el >> 8) & 15u, o2 = bits + xb;                     // o2: where the distance code starts
        const uint32_t len = (w.el >> 16) + bits_of(w.lo, bits, xb);
        const uint32_t d32 = __builtin_amdgcn_alignbit(w.hi, w.lo, o2);            // 32 bits from there on (o2 <= 20)
        uint32_t ed = s.dist[d32 & ((1u << kDistRoot) - 1u)];
        if ((ed & 0xf0u) == (kSub << 4)) ed = s.dist[(ed >> 16) + bits_of(d32, kDistRoot, (ed >> 8) & 15u)] + kDistRoot;
        const uint32_t db = ed & 15u, dxb = (ed >> 8) & 15u;
        const uint32_t dist = (ed >> 16) + bits_of(d32, db, dxb);                  // db + dxb <= 28
        // anything but a distance code behind a length: the chain stops at that lane with an error
        emit_mask |= len_lanes & __builtin_amdgcn_ballot_w64((ed & 0xf0u) == (kDist << 4));
        if (kind == kLen) {
            nxt = lane + o2 + db + dxb;
            word = len | dist << 16;
        }
    }
    // the chain's fixed point: a lane whose symbol this window does not take, or whose symbol ends the window
    const uint32_t hop = __builtin_amdgcn_inverse_ballot_w64(emit_mask) && nxt < (uint32_t)kWave ? nxt : lane;
    u64 on = 1;
    uint32_t f = 0;
    // (measured and not kept, round 4: the chain's second, third and fourth successors worked out in the lanes first -- three
    // ds_bpermute -- so that the four v_readlane of a round select by the same scalar and only the first waits for it: 50.6 / 47.0
    // against 51.2 / 47.6 GB/s, profiles/r04/ab_inflate_hop4.txt)
    for (;;) {
        const uint32_t a = lane_of(hop, f), b = lane_of(hop, a), c = lane_of(hop, b), d = lane_of(hop, c);
        asm("s_bitset1_b64 %0, %1\n\ts_bitset1_b64 %0, %2\n\ts_bitset1_b64 %0, %3\n\ts_bitset1_b64 %0, %4" : "+s"(on) : "s"(a), "s"(b), "s"(c), "s"(d));
        f = d;
        if (d == c) break;
    }
    // (the mask stays in scalar registers: `taken` needs no vector instruction, and the inverse ballot makes it the store's exec
    // mask -- testing the lane's bit of a 64-bit scalar costs three vector instructions)
    const u64 taken = on & emit_mask;
    // the taken lanes' words, in stream order, behind what is queued
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(taken >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)taken, 0u));
    if (__builtin_amdgcn_inverse_ballot_w64(taken)) s.queue[(q.head + q.n + below) & (kQueue - 1u)] = word;
    q.n += (uint32_t)__builtin_popcountll(taken);
    // the chain's last lane: its symbol was taken and ends in or behind the window's last bit, or it is the end-of-block code,
    // or an error (a length without a distance code behind it: 13; not a code at all: 15)
    const uint32_t fe = lane_of(w.el, f), fn = lane_of(nxt, f), fk = (fe >> 4) & 15u;
    const bool last_taken = (taken >> f) & 1u;
    o = last_taken || fk == kEob ? fn : f;           // (an end-of-block code's nxt = lane + bits, too)
    return last_taken ? kWinNext : fk == kEob ? kWinEob : fk == kLen ? kWinErrLen : kWinError;     // (the error code: decode_symbols)
}

// Decodes symbols from p on until the block's end-of-block code (-> true: p is behind it) or an error (-> false, err set).
template <typename Sink>
__device__ __forceinline__ bool decode_symbols(InfLds &s, Bits &b, Pos &p, const uint8_t *__restrict__ in, uint32_t in_len, Sink &sink,
                                               uint32_t &err)
{
    // the ring is staged half a ring ahead of the windows: one compare per window (`ahead`: the byte position from which on
    // the next half is due; none once the chunk's last bytes are in)
    auto due = [&]() { return b.filled < in_len + 8u ? b.filled - (8u + kRing / 4) : 0xffffffffu; };
    b.filled = uni(b.filled);
    uint32_t ahead = due();
    Queue q{0u, 0u};
    for (;;) {
        // windows, until 64 symbols are queued or the block ends (the loop the decoder lives in: one exit test)
        uint32_t how;
        do {
            p.byte = uni(p.byte), p.bit = uni(p.bit), ahead = uni(ahead), q.head = uni(q.head), q.n = uni(q.n);
            if (p.byte > ahead) {
                stage(s, b, in, in_len);
                b.filled = uni(b.filled), ahead = due();
            }
            const Win w = window(s, p.byte * 8u + p.bit);
            uint32_t o;
            how = walk(s, w, q, o);
            p.byte += (p.bit + o) >> 3, p.bit = (p.bit + o) & 7u;
        } while (how == kWinNext && q.n < (uint32_t)kWave);
        if (how >= kWinError) {
            err = how == kWinErrLen ? 13 : 15;
            return false;
        }
        sink.pin_state();
        if (q.n >= (uint32_t)kWave) {                 // (a window adds at most 64: the queue holds 128)
            if ((err = emit(s, sink, q, kWave)) != 0) return false;
        }
        if (how == kWinEob) {
            while (q.n) {
                if ((err = emit(s, sink, q, q.n < (uint32_t)kWave ? q.n : (uint32_t)kWave)) != 0) return false;
            }
        }
        if constexpr (Sink::kDry) {
            if (sink.bad) {                              // (a trial decode of something that is not text: no need to go on)
                err = 16;
                return false;
            }
        }
        if (how == kWinEob) return true;
    }
}

// Two literals per lookup: where a root-table index starts with a literal code of l1 bits and the
// remaining root - l1 bits hold a whole second literal code, the entry delivers both
// ([23:16] first, [31:24] second, [15:8] l1, code bits = l1 + l2).  The decoder is bound by the
// latency of its dependent table lookups, and most of a BAM block's symbols are literals
// (qualities, names); this halves the lookups for them.  In place: an entry that was already
// paired still shows its first literal and l1, so the pass can run on all entries at once.
__device__ __forceinline__ void pair_literals(uint32_t *tab, uint32_t root)
{
    for (uint32_t i = (uint32_t)lane_id(); i < (1u << root); i += kWave) {
        const uint32_t e1 = tab[i], k1 = (e1 >> 4) & 15u;
        if (k1 != kLit) continue;
        const uint32_t l1 = e1 & 15u, rest = root - l1;
        if (rest == 0) continue;
        const uint32_t e2 = tab[i >> l1], k2 = (e2 >> 4) & 15u;  // the second code sees the remaining bits, zero-extended
        uint32_t l2, lit2;
        if (k2 == kLit) l2 = e2 & 15u, lit2 = e2 >> 16;
        else if (k2 == kLit2) l2 = (e2 >> 8) & 255u, lit2 = (e2 >> 16) & 255u;
        else continue;
        if (l2 > rest) continue;  // its code would need bits beyond the index
        tab[i] = mk((e1 >> 16) | lit2 << 8, l1, kLit2, l1 + l2);
    }
}

// The code tables of a block whose three header bits have been taken: fixed (type 1, RFC 1951 3.2.6) or dynamic (type 2, 3.2.7)
// codes, literal pairs included.  -> 0 or the decoder's error code.
__device__ __forceinline__ uint32_t block_tables(InfLds &s, Bits &b, const uint8_t *__restrict__ in, uint32_t in_len, uint32_t type)
{
    const int lane = lane_id();
    if (type == 1) {
        for (uint32_t i = (uint32_t)lane; i < 288u; i += kWave) s.lens[i] = i < 144u ? 8 : i < 256u ? 9 : i < 280u ? 7 : 8;
        for (uint32_t i = (uint32_t)lane; i < 32u; i += kWave) s.lens[288 + i] = 5;
        if (!build(s, s.lit, kLitSize, kLitRoot, 0, 288, true, lit_payload) ||
            !build(s, s.dist, kDistSize, kDistRoot, 288, 32, true, dist_payload))  // 30 used + 2 reserved: complete
            return 4;
    } else {
        refill(s, b, in, in_len);
        const uint32_t hlit = take(b, 5) + 257u, hdist = take(b, 5) + 1u, hclen = take(b, 4) + 4u;
        if (hlit > 286u || hdist > 30u) return 5;
        // The code-length code (RFC 1951 3.2.7: up to 19 symbols, lengths 0..7), built in registers (round 6).  build() below is
        // made for 286 symbols: serial loops whose every step is an LDS round trip -- ~28 K clocks for these 19, which is what a
        // false candidate of the block-start search cost above all (k_gz_find_starts tries ~40 a slice: 80 K clocks each,
        // profiles/r06/find_diag.txt).  Here lane i holds symbol i's length: how many codes a length has is a ballot, a symbol's
        // code is its length's first code + its rank among the lanes of that length, and the 128 entries are written symbol by
        // symbol by the lanes side by side.  Over-subscribed or incomplete: 6, as before (inftrees.c accepts neither here).
        u64 cl = 0;
        for (uint32_t got = 0; got < 3u * hclen;) {
            refill(s, b, in, in_len);                                   // (at least 32 bits)
            const uint32_t n = 3u * hclen - got < 30u ? 3u * hclen - got : 30u;
            cl |= (u64)take(b, n) << got;
            got += n;
        }
        {
            const uint32_t i = (uint32_t)lane;
            // where symbol i's length stands in the stream (the inverse of 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15)
            const uint32_t j = i >= 16u ? i - 16u : i == 0u ? 3u : i < 8u ? 19u - 2u * i : 2u * i - 12u;
            const uint32_t len = i < 19u && j < hclen ? (uint32_t)(cl >> (3u * j)) & 7u : 0u;
            uint32_t code = 0, left = 1, first = 0, rank = 0;
            bool over = false;
#pragma unroll
            for (uint32_t l = 1; l <= 7u; ++l) {
                const u64 m = __builtin_amdgcn_ballot_w64(len == l);
                const uint32_t c = (uint32_t)__builtin_popcountll(m);
                left <<= 1;
                over = over || c > left;
                left -= over ? 0u : c;
                if (len == l) first = code, rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                code = (code + c) << 1;
            }
            if (over || left != 0u) return 6;
            const uint32_t mine = rev(first + rank, len ? len : 1u), entry = mk(i, 0, kLit, len);
            for (u64 used = __builtin_amdgcn_ballot_w64(len != 0u); used; used &= used - 1u) {
                const uint32_t sym = (uint32_t)__builtin_ctzll(used);
                const uint32_t r = lane_of(mine, sym), e = lane_of(entry, sym), l = e & 15u;
                for (uint32_t k = (uint32_t)lane; k < (128u >> l); k += kWave) s.dist[r + (k << l)] = e;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the entries were written by other lanes)
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t i = 0, prev = 0;
        const uint32_t total = hlit + hdist;
        // Kraft sums of the two codes as their lengths come in (units of 2^-15): an over-subscribed code is refused HERE, with the
        // code build() would refuse it with -- but before the rest of the lengths are decoded and counted.  Real blocks never get
        // there; the block-start search's false candidates (k_gz_find_starts: ~40 per slice whose code-length code happens to be
        // complete) nearly all do, within their first dozen lengths: 98 K clocks per false candidate before (decode ~300 lengths,
        // count them in build()), profiles/r06/find_diag.txt.
        uint32_t k_lit = 0, k_dist = 0;
        while (i < total) {
            refill(s, b, in, in_len);
            const uint32_t e = lookup(s.dist, 7, b);
            if (((e >> 4) & 15u) != kLit) return 7;
            const uint32_t sym = e >> 16;
            uint32_t rep = 1, val = sym;
            if (sym == 16u) {
                if (i == 0) return 8;
                rep = 3u + take(b, 2), val = prev;
            } else if (sym == 17u) {
                rep = 3u + take(b, 3), val = 0;
            } else if (sym == 18u) {
                rep = 11u + take(b, 7), val = 0;
            }
            if (i + rep > total) return 9;
            if (val) {
                const uint32_t in_lit = i < hlit ? (rep < hlit - i ? rep : hlit - i) : 0u;
                k_lit += in_lit * (32768u >> val), k_dist += (rep - in_lit) * (32768u >> val);
                if (k_lit > 32768u || k_dist > 32768u) return 11;
            }
            for (uint32_t k = (uint32_t)lane; k < rep; k += kWave) s.lens[32 + i + k] = (uint8_t)val;
            i += rep, prev = val;
        }
        if (s.lens[32 + 256] == 0) return 10;  // no end-of-block code
        // incomplete codes: build() refuses them below unless the code is a single symbol of length 1 (sum 1/2) or, for distances,
        // empty -- sums that cannot be either are refused here, before the lengths are counted
        if ((k_lit < 32768u && k_lit != 16384u) || (k_dist < 32768u && k_dist != 16384u && k_dist != 0u)) return 11;
        if (!build(s, s.lit, kLitSize, kLitRoot, 32, hlit, true, lit_payload) ||
            !build(s, s.dist, kDistSize, kDistRoot, 32 + hlit, hdist, true, dist_payload))
            return 11;
    }
    pair_literals(s.lit, kLitRoot);
    return 0;
}

}  // namespace hpn
