// fast_inflate.hpp -- a quicker DEFLATE decoder and CRC-32 for the host-side gzip readers.
//
// Everything that still has to be inflated on the host (plain .fastq.gz, the host fallbacks of the
// BGZF paths) goes through zlib in the reference (gzread, bgzf.c inflate_block) at ~0.3 GB/s of
// output per core, and zlib's table-driven crc32 at ~0.9 GB/s takes a third of that time.  This
// is the usual modern decoder shape instead: a 64-bit bit buffer refilled with one unaligned
// load, an 11-bit root table for literal/length codes (sub-tables beyond), up to three symbols
// per refill, word-wise match copies; and a carry-less-multiply CRC-32 (folding by 4 x 128 bits)
// where the CPU has PCLMULQDQ.  Streaming: run() fills the caller's buffer and is resumed at
// symbol granularity with the 32 KiB history kept in front of the buffer by the caller.
//
// Not a replacement for zlib's judgement: the readers check ISIZE and CRC-32 like gzread does,
// and anything this decoder rejects, or that fails those checks, is re-read with zlib itself.
// tests/test_host_ingest.py holds both to zlib byte for byte (levels, strategies, damage).
#pragma once
#include <stdint.h>
#include <string.h>
#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace hpn {

// ---- CRC-32 (the gzip polynomial) -----------------------------------------------------------------
#if defined(__x86_64__)
// state in / state out are the raw register contents (zlib's value inverted); len >= 64, multiple of 16
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul_raw(const uint8_t *buf, size_t len, uint32_t state)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(buf + 0)), x2 = _mm_loadu_si128((const __m128i *)(buf + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(buf + 32)), x4 = _mm_loadu_si128((const __m128i *)(buf + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    buf += 64, len -= 64;
    while (len >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11), x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11), x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(buf + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(buf + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(buf + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(buf + 48)));
        buf += 64, len -= 64;
    }
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x2), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x3), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x4), t);
    while (len >= 16) {
        t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), _mm_loadu_si128((const __m128i *)buf)), t);
        buf += 16, len -= 16;
    }
    // 128 -> 64 -> 32 bits (Barrett)
    const __m128i mask = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i y = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), y);
    y = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask), k5k0, 0x00), y);
    y = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask), poly, 0x10), mask);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(y, poly, 0x00));
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// zlib-compatible: crc32_fast(crc32_fast(0, a, n), b, m) == crc32 of a||b
inline uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len)
{
#if defined(__x86_64__)
    static const bool usable = [] {  // needs the instruction, and must agree with zlib on a probe
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t probe[208];
        for (size_t i = 0; i < sizeof probe; ++i) probe[i] = (uint8_t)(i * 131u + 7u);
        return ~crc32_clmul_raw(probe, sizeof probe, ~0u) == (uint32_t)::crc32(0, probe, sizeof probe);
    }();
    if (usable && len >= 64) {
        const size_t body = len & ~(size_t)15;
        crc = ~crc32_clmul_raw(buf, body, ~crc);
        buf += body, len -= body;
    }
#endif
    while (len) {  // zlib takes uInt
        const uInt k = len > (1u << 30) ? (1u << 30) : (uInt)len;
        crc = (uint32_t)::crc32(crc, buf, k);
        buf += k, len -= k;
    }
    return crc;
}

// ---- DEFLATE ------------------------------------------------------------------------------------------
// T = uint8_t: bytes.  T = uint16_t: the same decoder writing one 16-bit symbol per byte, so that a
// caller who does not know the 32 KiB of history yet can put placeholders (values >= 256) in front
// of the buffer and have the match copies carry them along (pgz_reader.hpp).
template <class T>
class FastInflateT {
public:
    enum { kBlockEnd = 2, kNeedOutput = 1, kDone = 0, kError = -1 };
    static constexpr size_t kOvershoot = 320;  // run() may write this many elements past dst_end (one match + word copy)

    // raw DEFLATE starting `bit` bits (0..7) into `in`; nothing at or beyond `limit` is read.
    // stop_blocks: run() also returns (kBlockEnd) after every non-final block.
    void begin(const uint8_t *in, const uint8_t *limit, uint32_t bit = 0, bool stop_blocks = false)
    {
        in_ = in, limit_ = limit;
        bb_ = 0, bc_ = 0;
        state_ = kHeader, last_ = false, stored_ = 0, stop_blocks_ = stop_blocks;
        if (bit) {
            refill();
            if (bc_ >= bit) drop(bit);
            else in_ = limit_, bb_ = 0, bc_ = 0;
        }
    }
    // first input byte not consumed (valid after kDone)
    const uint8_t *in_pos() const { return in_ - (bc_ >> 3); }
    // bit offset from `base` of the next unread bit (valid whenever run() has returned)
    uint64_t bit_pos(const uint8_t *base) const { return (uint64_t)(in_ - base) * 8 - bc_; }

    // Decode into [dst, dst_end) (+ kOvershoot of slack).  Bytes [hist, dst) are the history (earlier
    // output, up to 32 KiB of it is looked at).  Advances dst.
    int run(T *&dst, T *dst_end, const T *hist)
    {
        for (;;) {
            if (state_ == kHeader) {
                if (last_) return kDone;
                refill();
                if (bc_ < 3) return kError;
                last_ = bb_ & 1;
                const uint32_t type = (uint32_t)(bb_ >> 1) & 3;
                drop(3);
                if (type == 0) {
                    drop(bc_ & 7);                 // to the byte boundary
                    in_ -= bc_ >> 3, bb_ = 0, bc_ = 0;  // whole bytes go back to the input
                    if (limit_ - in_ < 4) return kError;
                    const uint32_t len = in_[0] | in_[1] << 8, nlen = in_[2] | in_[3] << 8;
                    if ((len ^ nlen) != 0xffff) return kError;
                    in_ += 4;
                    stored_ = len;
                    state_ = kStored;
                } else if (type == 1) {
                    if (!fixed_tables()) return kError;
                    state_ = kSymbols;
                } else if (type == 2) {
                    if (!dynamic_tables()) return kError;
                    state_ = kSymbols;
                } else {
                    return kError;
                }
            }
            if (state_ == kStored) {
                while (stored_) {
                    if (dst >= dst_end) return kNeedOutput;
                    size_t k = (size_t)(dst_end - dst) < stored_ ? (size_t)(dst_end - dst) : stored_;
                    if ((size_t)(limit_ - in_) < k) return kError;
                    if constexpr (sizeof(T) == 1) memcpy(dst, in_, k);
                    else
                        for (size_t i = 0; i < k; ++i) dst[i] = in_[i];
                    dst += k, in_ += k, stored_ -= (uint32_t)k;
                }
                state_ = kHeader;
                continue;
            }
            // ---- kSymbols ----
            // The decoder state lives in locals here: byte stores through `out` may alias any member, and
            // the compiler would otherwise reload and spill the bit buffer around every literal.
            const uint8_t *in = in_;
            uint64_t bb = bb_;
            uint32_t bc = bc_;
            T *out = dst;
            const uint32_t *const lit = lit_, *const dtab = dist_;
            const uint8_t *const limit = limit_;
            auto leave = [&](int rc) {
                in_ = in, bb_ = bb, bc_ = bc, dst = out;
                return rc;
            };
            auto fill = [&]() {
                if (__builtin_expect(limit - in >= 8, 1)) {  // one unaligned load; the bits above bc are real and are loaded again
                    uint64_t v;
                    memcpy(&v, in, 8);
                    bb |= v << bc;
                    in += (63 - bc) >> 3;
                    bc |= 56;
                } else {
                    while (bc <= 56 && in < limit) bb |= (uint64_t)*in++ << bc, bc += 8;
                }
            };
            auto put = [&](uint32_t e) {  // a literal entry: store both bytes, advance by one or two, consume its bits
                out[0] = (uint8_t)(e >> 16), out[1] = (T)(e >> 24);
                out += 1 + kind(e);
                bb >>= (e & 15), bc -= (e & 15);
            };
            constexpr uint32_t kLitMask = (1u << kLitRoot) - 1;
            for (;;) {
                if (out >= dst_end) return leave(kNeedOutput);
                fill();
                uint32_t e = lit[bb & kLitMask];
                // Literal entries carry one or two bytes (two where both codes fit in the root index).  Up to
                // three such entries per refill: each uses at most 11 of the 56+ buffered bits.
                if (kind(e) <= kLit2 && bc >= 48) {
                    put(e);
                    e = lit[bb & kLitMask];
                    if (kind(e) <= kLit2) {
                        put(e);
                        e = lit[bb & kLitMask];
                        if (kind(e) <= kLit2) {
                            put(e);
                            continue;
                        }
                    }
                }
                if (kind(e) == kSub) {
                    if (bc < kLitRoot) return leave(kError);
                    bb >>= kLitRoot, bc -= kLitRoot;
                    e = lit[(e >> 16) + ((uint32_t)bb & ((1u << ((e >> 8) & 255)) - 1))];
                }
                if ((e & 15) > bc) return leave(kError);  // the stream ends inside a code
                bb >>= (e & 15), bc -= (e & 15);
                const uint32_t k = kind(e);
                if (k <= kLit2) {
                    out[0] = (uint8_t)(e >> 16), out[1] = (T)(e >> 24);
                    out += 1 + k;
                    continue;
                }
                if (k == kEob) {
                    state_ = kHeader;
                    if (stop_blocks_ && !last_) return leave(kBlockEnd);
                    break;
                }
                if (k != kLen) return leave(kError);
                uint32_t xb = (e >> 8) & 255;
                if (bc < xb) return leave(kError);
                const uint32_t len = (e >> 16) + ((uint32_t)bb & ((1u << xb) - 1));
                bb >>= xb, bc -= xb;
                if (bc < 28) fill();  // a distance code and its extra bits: up to 15 + 13
                uint32_t d = dtab[bb & ((1u << kDistRoot) - 1)];
                if (kind(d) == kSub) {
                    if (bc < kDistRoot) return leave(kError);
                    bb >>= kDistRoot, bc -= kDistRoot;
                    d = dtab[(d >> 16) + ((uint32_t)bb & ((1u << ((d >> 8) & 255)) - 1))];
                }
                if (kind(d) != kDist || (d & 15) > bc) return leave(kError);
                bb >>= (d & 15), bc -= (d & 15);
                xb = (d >> 8) & 255;
                if (bc < xb) return leave(kError);
                const size_t distance = (d >> 16) + ((uint32_t)bb & ((1u << xb) - 1));
                bb >>= xb, bc -= xb;
                if (distance > (size_t)(out - hist)) return leave(kError);
                const T *src = out - distance;
                T *end = out + len;
                constexpr size_t kWord = 8 / sizeof(T);
                if (distance >= kWord) {  // word-wise, may run up to 7 bytes past `end` (slack / overwritten next)
                    do {
                        memcpy(out, src, 8);
                        out += kWord, src += kWord;
                    } while (out < end);
                } else if (distance == 1) {
                    const T v = *src;
                    if constexpr (sizeof(T) == 1) memset(out, v, len);
                    else
                        while (out < end) *out++ = v;
                } else {
                    while (out < end) *out++ = *src++;
                }
                out = end;
            }
            leave(0);
        }
    }

private:
    enum { kHeader, kStored, kSymbols };
    enum { kLit = 0, kLit2 = 1, kLen = 2, kEob = 3, kSub = 4, kDist = 5, kBad = 15 };  // literal kinds first: one compare
    static constexpr uint32_t kLitRoot = 11, kDistRoot = 8;
    static constexpr uint32_t kLitSize = 2048 + 512, kDistSize = 256 + 256;
    static uint32_t mk(uint32_t value, uint32_t extra, uint32_t kd, uint32_t nbits) { return value << 16 | extra << 8 | kd << 4 | nbits; }
    static uint32_t kind(uint32_t e) { return (e >> 4) & 15; }

    void drop(uint32_t n) { bb_ >>= n, bc_ -= n; }
    void refill()
    {
        if (limit_ - in_ >= 8) {  // one unaligned load; the bits above bc_ are real and are loaded again next time
            uint64_t v;
            memcpy(&v, in_, 8);
            bb_ |= v << bc_;
            in_ += (63 - bc_) >> 3;
            bc_ |= 56;
        } else {
            while (bc_ <= 56 && in_ < limit_) bb_ |= (uint64_t)*in_++ << bc_, bc_ += 8;
        }
    }
    uint32_t bits(uint32_t n)
    {
        const uint32_t v = (uint32_t)bb_ & ((1u << n) - 1);
        drop(n);
        return v;
    }

    static uint32_t lit_payload(uint32_t sym, uint32_t nb)
    {
        if (sym < 256) return mk(sym, 0, kLit, nb);
        if (sym == 256) return mk(0, 0, kEob, nb);
        if (sym < 286) {  // RFC 1951 3.2.5
            const uint32_t k = sym - 257;
            if (k < 8) return mk(3 + k, 0, kLen, nb);
            if (k == 28) return mk(258, 0, kLen, nb);
            const uint32_t xb = (k - 4) >> 2;
            return mk(3 + ((4 + (k & 3)) << xb), xb, kLen, nb);
        }
        return mk(0, 0, kBad, nb);
    }
    static uint32_t dist_payload(uint32_t sym, uint32_t nb)
    {
        if (sym < 30) {
            if (sym < 4) return mk(1 + sym, 0, kDist, nb);
            const uint32_t xb = (sym - 2) >> 1;
            return mk(1 + ((2 + (sym & 1)) << xb), xb, kDist, nb);
        }
        return mk(0, 0, kBad, nb);
    }
    static uint32_t cl_payload(uint32_t sym, uint32_t nb) { return mk(sym, 0, kLit, nb); }
    static uint32_t rev(uint32_t code, uint32_t len)
    {
        uint32_t r = 0;
        for (uint32_t i = 0; i < len; ++i) r |= ((code >> i) & 1) << (len - 1 - i);
        return r;
    }

    // canonical code -> root table + sub-tables sized per prefix (inftrees.c's layout); false: over-subscribed,
    // incomplete beyond what zlib accepts, or too large
    bool build(uint32_t *tab, uint32_t tab_size, uint32_t root, const uint8_t *lens, uint32_t n, bool allow_single,
               uint32_t (*payload)(uint32_t, uint32_t))
    {
        uint32_t count[16] = {0}, first[16], next[16], maxlen = 0;
        for (uint32_t i = 0; i < n; ++i) {
            count[lens[i]]++;
            if (lens[i] > maxlen) maxlen = lens[i];
        }
        count[0] = 0;
        uint32_t code = 0;
        int64_t left = 1;
        for (uint32_t l = 1; l <= 15; ++l) {
            left = (left << 1) - (int64_t)count[l];
            if (left < 0) return false;
            code = (code + count[l - 1]) << 1;
            first[l] = next[l] = code;
        }
        if (left > 0 && !(allow_single && maxlen <= 1)) return false;
        for (uint32_t i = 0; i < tab_size; ++i) tab[i] = mk(0, 0, kBad, 0);
        if (maxlen == 0) return true;
        uint32_t sub_next = 1u << root;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t l = lens[i];
            if (!l) continue;
            const uint32_t c = next[l]++, r = rev(c, l);
            if (l <= root) {
                const uint32_t e = payload(i, l);
                for (uint32_t k = r; k < (1u << root); k += 1u << l) tab[k] = e;
            } else {
                const uint32_t prefix = r & ((1u << root) - 1), top = c >> (l - root);
                uint32_t pe = tab[prefix];
                if (kind(pe) != kSub) {
                    uint32_t sb = l - root;
                    for (uint32_t m = maxlen; m > l; --m)
                        if (count[m] && top >= (first[m] >> (m - root)) && top <= ((first[m] + count[m] - 1) >> (m - root))) {
                            sb = m - root;
                            break;
                        }
                    if (sub_next + (1u << sb) > tab_size) return false;
                    pe = mk(sub_next, sb, kSub, root);
                    tab[prefix] = pe;
                    sub_next += 1u << sb;
                }
                const uint32_t sb = (pe >> 8) & 255, so = pe >> 16, e = payload(i, l - root);
                for (uint32_t k = r >> root; k < (1u << sb); k += 1u << (l - root)) tab[so + k] = e;
            }
        }
        return true;
    }
    // two literals per root entry where both codes fit in the index (in place: a paired entry still shows
    // its first literal and that literal's code length in the extra field)
    void pair_literals()
    {
        for (uint32_t i = 0; i < (1u << kLitRoot); ++i) {
            const uint32_t e1 = lit_[i];
            if (kind(e1) != kLit) continue;
            const uint32_t l1 = e1 & 15, rest = kLitRoot - l1;
            if (!rest) continue;
            const uint32_t e2 = lit_[i >> l1];
            uint32_t l2, b2;
            if (kind(e2) == kLit) l2 = e2 & 15, b2 = (e2 >> 16) & 255;
            else if (kind(e2) == kLit2) l2 = (e2 >> 8) & 255, b2 = (e2 >> 16) & 255;
            else continue;
            if (l2 > rest) continue;
            lit_[i] = mk(((e1 >> 16) & 255) | b2 << 8, l1, kLit2, l1 + l2);
        }
    }
    bool fixed_tables()
    {
        uint8_t lens[320];
        for (uint32_t i = 0; i < 288; ++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (uint32_t i = 0; i < 32; ++i) lens[288 + i] = 5;
        if (!build(lit_, kLitSize, kLitRoot, lens, 288, true, lit_payload) || !build(dist_, kDistSize, kDistRoot, lens + 288, 32, true, dist_payload))
            return false;
        pair_literals();
        return true;
    }
    bool dynamic_tables()
    {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        refill();
        if (bc_ < 14) return false;
        const uint32_t hlit = bits(5) + 257, hdist = bits(5) + 1, hclen = bits(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        uint8_t cl[19] = {0}, lens[320];
        for (uint32_t i = 0; i < hclen; ++i) {
            if (bc_ < 3) refill();
            if (bc_ < 3) return false;
            cl[order[i]] = (uint8_t)bits(3);
        }
        if (!build(dist_, kDistSize, 7, cl, 19, false, cl_payload)) return false;  // the code-length code borrows the distance table
        uint32_t i = 0, prev = 0;
        const uint32_t total = hlit + hdist;
        while (i < total) {
            if (bc_ < 14) refill();
            const uint32_t e = dist_[bb_ & 127];
            if (kind(e) != kLit || (e & 15) > bc_) return false;
            drop(e & 15);
            const uint32_t sym = e >> 16;
            uint32_t rep = 1, val = sym;
            if (sym == 16) {
                if (i == 0 || bc_ < 2) return false;
                rep = 3 + bits(2), val = prev;
            } else if (sym == 17) {
                if (bc_ < 3) return false;
                rep = 3 + bits(3), val = 0;
            } else if (sym == 18) {
                if (bc_ < 7) return false;
                rep = 11 + bits(7), val = 0;
            }
            if (i + rep > total) return false;
            memset(lens + i, (int)val, rep);
            i += rep, prev = val;
        }
        if (lens[256] == 0) return false;  // no end-of-block code
        if (!build(lit_, kLitSize, kLitRoot, lens, hlit, true, lit_payload) ||
            !build(dist_, kDistSize, kDistRoot, lens + hlit, hdist, true, dist_payload))
            return false;
        pair_literals();
        return true;
    }

    const uint8_t *in_ = nullptr, *limit_ = nullptr;
    uint64_t bb_ = 0;
    uint32_t bc_ = 0;
    int state_ = kHeader;
    bool last_ = false, stop_blocks_ = false;
    uint32_t stored_ = 0;
    uint32_t lit_[kLitSize], dist_[kDistSize];
};

using FastInflate = FastInflateT<uint8_t>;

}  // namespace hpn
