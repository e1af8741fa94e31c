// inflate_core.hpp -- the wave-level DEFLATE machinery shared by the BGZF-block kernel (bgzf_inflate.hip) and the
// symbolic single-member-gzip kernel (gz_inflate.hip): LDS layout, wave-uniform bit reader over an LDS input ring,
// canonical Huffman table construction (inftrees.c's layout), table lookup.  See bgzf_inflate.hip for the design.
#pragma once
#include "common.hpp"

namespace hpn {

constexpr uint32_t kRing = 1024;             // compressed-input ring (bytes), refilled by halves
constexpr uint32_t kLitRoot = 10, kDistRoot = 8;
constexpr uint32_t kLitSize = 1024 + 384, kDistSize = 256 + 144;  // root + sub-tables (inftrees.c ENOUGH: 1332 for a 10-bit root)

// table entry: [31:16] value, [15:8] extra-bit count (or sub-table index bits), [7:4] kind, [3:0] code bits
enum { kLit = 0, kLit2 = 1, kLen = 2, kEob = 3, kSub = 4, kDist = 5, kBad = 15 };  // literal kinds first: one compare
__device__ __forceinline__ uint32_t mk(uint32_t value, uint32_t extra, uint32_t kind, uint32_t nbits)
{
    return value << 16 | extra << 8 | kind << 4 | nbits;
}

struct InfLds {
    uint32_t lit[kLitSize];
    uint32_t dist[kDistSize];
    uint32_t ring[kRing / 4];
    uint8_t lens[384];     // code lengths being assembled (19 code-length codes | 286 + 30 lengths from offset 32)
    uint16_t count[16], first[16], next[16];
};

struct Bits {  // wave-uniform bit reader over the LDS ring
    u64 bb = 0;
    uint32_t bc = 0;       // valid bits in bb
    uint32_t in_pos = 0;   // compressed bytes moved into bb
    uint32_t filled = 0;   // compressed bytes staged into the ring
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// stage the next kRing/2 bytes of the block's compressed data (16 B per lane)
__device__ __forceinline__ void stage(InfLds &s, Bits &b, const uint8_t *__restrict__ in, uint32_t in_len)
{
    const uint32_t at = b.filled + 16u * (uint32_t)lane_id();
    if (16u * (uint32_t)lane_id() < kRing / 2) {
        u32 v = {0, 0, 0, 0};
        if (at < in_len + 16u) __builtin_memcpy(&v, in + at, 16);  // the buffer is padded by the host
        *(u32 *)((uint8_t *)s.ring + (at & (kRing - 1))) = v;
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

}  // namespace hpn
