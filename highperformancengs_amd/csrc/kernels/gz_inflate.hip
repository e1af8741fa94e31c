// gz_inflate.hip -- ONE gzip member (the usual .fastq.gz) inflated on gfx950, two passes.
//
// A gzip member is a single DEFLATE stream in which every byte may refer to the 32 KiB before it;
// the reference reads it through gzread on one host thread (fastq_count.c:112-118 via gzgets).
// host/pgz_reader.hpp does the two-pass parallel decode on the host cores; this is the same
// scheme with the heavy passes on the device (the host only finds where deflate blocks start):
//
//   k_gz_sym_inflate  one wavefront per stretch [block start, next stretch's start): the decoder of
//                     bgzf_inflate.hip writing 16-bit symbols -- a byte value, or 256 + i for
//                     "byte i of the 32 KiB of history before this stretch", which a match that
//                     reaches in front of the stretch's own output produces.  The stretch must end
//                     on its given end bit exactly at a block boundary (or at the final block),
//                     which is what makes the next stretch's start a proven block boundary.
//   k_gz_windows      one small workgroup walks the stretches in order: the resolved last 32 KiB of a
//                     stretch are the history of the next; also the running text offsets.
//   k_gz_translate    symbols -> bytes through each stretch's history, compacted into one text.
//
// Bounds: k_gz_sym_inflate by instruction issue like k_bgzf_inflate (the same decoder: symbols 64 bit
// offsets at a time, inflate_core.hpp); the other two are small streaming passes.  CRC-32 is not checked here (ISIZE is,
// by the caller); anything malformed sets a status and the host readers take the file.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "inflate_core.hpp"

namespace hpn {

struct GzChunk {  // = hpn_gz_chunk
    uint64_t in_off;   // byte of the compressed buffer that holds the stretch's first bit
    uint64_t end_bit;  // where the stretch ends, in bits from in_off * 8; ~0: at the final block
    uint32_t in_len;   // bytes that may be read from in_off on
    uint32_t start_bit;
};
struct GzBound {   // a member that ended inside a stretch, and went on with the next member
    uint32_t chunk, n_out;   // symbols of the stretch that belong to members up to and including this one
    uint32_t isize, crc;      // ISIZE and CRC-32 of its trailer
};
struct GzMeta {
    uint32_t n_out, status, final_block, reserved;
    uint64_t end_bit;   // bit position reached (byte-aligned after a final block), from in_off * 8
    uint64_t text_off;  // filled by k_gz_windows
};

constexpr uint32_t kGzHist = 32768;

// where the symbols go: 16 bits each, straight to global memory; a match reads its source back from there, or names the
// byte of the 32 KiB in front of the stretch it would have copied
struct SymSink {
    static constexpr bool kDry = false;
    uint16_t *out;
    uint32_t out_len, op, safe;   // op: symbols decoded; symbols below `safe` are known to have reached memory
    __device__ __forceinline__ void pin_state() { op = uni(op), safe = uni(safe); }
    // entry e: [23:16] the literal, [31:24] the second one of a pair.  (Byte offsets in 32 bits: a stretch's symbols stay far
    // below 2 GiB.)
    __device__ __forceinline__ void lits(bool mine, bool two, uint32_t at, uint32_t e)
    {
        if (mine) {
            uint8_t *to = (uint8_t *)out + (at << 1);
            if (two) {
                const uint32_t v = ((e >> 16) & 255u) | (e >> 24) << 16;
                __builtin_memcpy(to, &v, 4);                    // (one store; `at` may be odd)
            } else {
                *(uint16_t *)to = (uint16_t)((e >> 16) & 255u);
            }
        }
    }
    __device__ __forceinline__ bool in_reach(uint32_t at, uint32_t dist) const { return dist <= at + kGzHist; }
    // symbols at `from` (< base, the first position of the chunk being assembled; negative: in the history in front of the
    // stretch, whose byte this position will be) for the lanes that ask
    __device__ __forceinline__ uint32_t fetch(bool ask, int32_t from, uint32_t base, uint32_t val)
    {
        if (__builtin_amdgcn_ballot_w64(ask && from >= (int32_t)safe)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            safe = base;
        }
        if (ask) val = from < 0 ? 256u + kGzHist + (uint32_t)from : (uint32_t)out[from];
        return val;
    }
    __device__ __forceinline__ void put(bool live, uint32_t at, uint32_t val)
    {
        if (live) *(uint16_t *)((uint8_t *)out + (at << 1)) = (uint16_t)val;
    }
};

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(HPN_INF_EU, 8))) void k_gz_sym_inflate(const uint8_t *__restrict__ comp, const GzChunk *__restrict__ chunks,
                                                          uint32_t n_chunks, uint16_t *__restrict__ symbuf, uint32_t sym_cap,
                                                          GzMeta *__restrict__ meta, GzBound *__restrict__ bounds, uint32_t bounds_cap,
                                                          uint32_t *__restrict__ n_bounds)
{
    __shared__ InfLds s;
    const int lane = lane_id();
    // Stretches are TAKEN (round 6; k_bgzf_inflate's blocks since round 4): a wave's first stretch is its own index, every further
    // one comes from a counter (n_bounds[1], cleared with the member count) -- with more stretches than decoder slots the launch no
    // longer ends with the wave that was dealt the slow ones.  No counter (a caller without a member list): dealt by stride.
    for (uint32_t ci = blockIdx.x; ci < n_chunks;) {
        const GzChunk ck = chunks[ci];
        const uint8_t *in = comp + ck.in_off;
        uint16_t *out = symbuf + (uint64_t)ci * sym_cap;
        const uint32_t in_len = ck.in_len, out_len = sym_cap;
        const uint64_t end_bit = ck.end_bit;
        Bits b;
        stage(s, b, in, in_len);
        stage(s, b, in, in_len);
        refill(s, b, in, in_len);
        drop(b, ck.start_bit);
        SymSink sink{out, out_len, 0u, 0u};
        uint32_t &op = sink.op;
        uint32_t err = 0;
        bool last = false, arrived = false;
        while (!last && !err) {
            pin(b), pin(err), sink.pin_state();
            const uint64_t pos = (uint64_t)b.in_pos * 8u - b.bc;  // a block boundary
            if (pos >= end_bit) {
                if (pos == end_bit) arrived = true;
                else err = 20;  // the given end is not a block boundary of this stream
                break;
            }
            refill(s, b, in, in_len);
            if (b.in_pos - (b.bc >> 3) > in_len) {
                err = 17;
                break;
            }
            last = take(b, 1) != 0;
            const uint32_t type = take(b, 2);
            if (type == 0) {  // stored
                drop(b, b.bc & 7u);
                refill(s, b, in, in_len);
                const uint32_t len = take(b, 16);
                refill(s, b, in, in_len);
                const uint32_t nlen = take(b, 16);
                if ((len ^ nlen) != 0xffffu || op + len > out_len) {
                    err = 1;
                    break;
                }
                const uint32_t src = b.in_pos - (b.bc >> 3);
                if (src + len > in_len) {
                    err = 2;
                    break;
                }
                for (uint32_t i = (uint32_t)lane; i < len; i += kWave) out[op + i] = in[src + i];
                op += len;
                b.bb = 0, b.bc = 0;
                b.in_pos = src + len;
                b.filled = b.in_pos & ~(kRing / 2 - 1);
                stage(s, b, in, in_len);
                stage(s, b, in, in_len);
                continue;
            }
            if (type == 3) {
                err = 3;
                break;
            }
            if ((err = block_tables(s, b, in, in_len, type)) != 0) break;
            // ---- symbols of this block: 64 bit offsets at a time (decode_symbols, inflate_core.hpp) ----
            {
                Pos p = pos_of(b);
                if (!decode_symbols(s, b, p, in, in_len, sink, err)) break;
                seek(s, b, p, in, in_len);
            }
            // ---- the member's final block is done: does another member follow (cat a.gz b.gz, pigz -i ...)? ----
            // 8 bytes of trailer (CRC-32, ISIZE), then a gzip header (RFC 1952) and the next member's first block: the
            // stretch goes on with it -- matches never reach back over a member's start, so nothing else changes.  No header
            // there (the end of the file, bytes that are not gzip): the stretch ends at this final block as before.
            if (last && !err) {
                drop(b, b.bc & 7u);
                const uint32_t at = b.in_pos - (b.bc >> 3);               // byte of the trailer, from in_off
                if (at + 8u + 10u + 1u <= in_len && bounds) {
                    const Bits keep = b;                                    // (the ring holds what lies 512 bytes back: see below)
                    auto byte = [&]() {
                        refill(s, b, in, in_len);
                        return take(b, 8);
                    };
                    uint32_t isize = 0, crc = 0, hdr_ok = 1;
                    for (int k = 0; k < 4; ++k) crc |= byte() << (8 * k);
                    for (int k = 0; k < 4; ++k) isize |= byte() << (8 * k);
                    hdr_ok = byte() == 0x1fu;
                    hdr_ok = (byte() == 0x8bu) && hdr_ok;
                    hdr_ok = (byte() == 8u) && hdr_ok;
                    const uint32_t flg = byte();
                    hdr_ok = hdr_ok && !(flg & 0xe0u);
                    if (hdr_ok) {
                        for (int k = 0; k < 6; ++k) (void)byte();          // MTIME, XFL, OS
                        uint32_t left = in_len - (b.in_pos - (b.bc >> 3));  // every loop below is bounded by the input
                        if (flg & 4u) {                                     // FEXTRA
                            uint32_t xlen = byte();
                            xlen |= byte() << 8;
                            for (uint32_t k = 0; k < xlen && left > 0; ++k, --left) (void)byte();
                        }
                        if (flg & 8u)                                       // FNAME
                            while (left > 0 && byte() != 0) --left;
                        if (flg & 16u)                                      // FCOMMENT
                            while (left > 0 && byte() != 0) --left;
                        if (flg & 2u) (void)byte(), (void)byte();           // FHCRC
                        hdr_ok = b.in_pos - (b.bc >> 3) + 1u <= in_len;
                    }
                    if (hdr_ok) {
                        uint32_t slot = 0;
                        if (lane == 0) slot = atomicAdd(n_bounds, 1u);
                        slot = uni(__shfl(slot, 0, kWave));
                        if (slot >= bounds_cap) {
                            err = 23;                                       // more members than the caller made room for
                        } else {
                            if (lane == 0) bounds[slot] = GzBound{ci, op, isize, crc};
                            last = false;                                   // the next member's first block
                        }
                    } else {
                        // not a member: un-read the trailer.  The bytes are still in the ring unless the header parse ran far
                        // (a long FNAME): then the stretch is reported as ending in something this decoder cannot place.
                        if (b.in_pos - keep.in_pos > kRing / 4) err = 24;
                        else b = keep;
                    }
                }
            }
        }
        uint64_t pos = (uint64_t)b.in_pos * 8u - b.bc;
        if (!err && last) pos = (pos + 7u) & ~(uint64_t)7u;          // the trailer is byte-aligned
        if (!err && last && pos > (uint64_t)in_len * 8u) err = 17;   // the final block ends behind the input (bytes there read as zeros: k_bgzf_inflate's note)
        if (!err && !last && !arrived) err = 21;                      // cannot happen: the loop only leaves on one of these
        // a final block inside a stretch that was given an end: the member ends before the next stretch's start, which
        // therefore is not proven by this one (a second member, or bytes behind the stream): the caller hands the file back
        if (!err && last && end_bit != ~0ull) err = 22;
        if (lane == 0) {
            GzMeta m;
            m.n_out = op, m.status = err, m.final_block = last ? 1u : 0u, m.reserved = 0;
            m.end_bit = pos, m.text_off = 0;
            meta[ci] = m;
        }
        if (n_bounds) {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(n_bounds + 1, 1u);
            ci = gridDim.x + uni(__shfl(t, 0, kWave));
        } else {
            ci += gridDim.x;
        }
    }
}

// ---- where do deflate blocks start? -----------------------------------------------------------------------------------------
// The stretches of a member begin at block starts nobody marks: host/pgz_reader.hpp's gz_find_block_start looks for them by
// trial (~0.4 ms per stretch and core: with 16 cores the search took as long as the device needed to inflate the file).  The
// same trial on the device, one wavefront per slice of the compressed bytes: the lanes test 64 bit positions at a time for a
// dynamic block's header (BFINAL 0, BTYPE 2, HLIT / HDIST in range, the code-length code complete: Kraft sum 1, RFC 1951 3.2.7),
// and what passes is decoded by the wave with the decoder above into a sink that stores nothing and only asks whether the
// literals are text: the block to its end, and the beginning of the one behind it.  Like every proposed start it is PROVEN only
// by the stretch before it arriving there exactly (k_gz_sym_inflate, status 20 otherwise).
struct GzSlice {       // bit positions in the compressed buffer
    uint64_t lo, hi;
};
struct TextCheck {     // a sink that keeps nothing
    static constexpr bool kDry = true;
    uint32_t out_len, op, safe, bad;
    __device__ __forceinline__ void pin_state() { op = uni(op), bad = uni(bad); }
    static __device__ __forceinline__ bool texty(uint32_t v) { return (v >= 32u && v < 127u) || v == '\n' || v == '\r' || v == '\t'; }
    __device__ __forceinline__ void lits(bool mine, bool two, uint32_t, uint32_t e)
    {
        if (__ballot(mine && !(texty((e >> 16) & 255u) && (!two || texty(e >> 24))))) bad = 1;
    }
    __device__ __forceinline__ bool in_reach(uint32_t at, uint32_t dist) const { return dist <= at + kGzHist; }
};

// How much of a proposed block is decoded before it is believed (units of output).  Round 4: 4,096 -- 0.6 ms of one wave per
// slice, as much as the scan in front of it.  A header that parses, three COMPLETE Huffman codes (the code-length code's Kraft sum,
// build()'s completeness test for the literal/length and the distance code) and 1,024 units of text in a row are no accident
// either, and a start is proven by the stretch before it arriving there anyway (k_gz_sym_inflate, status 20).
#ifndef HPN_GZ_TRIAL_SYMBOLS
#define HPN_GZ_TRIAL_SYMBOLS 1024
#endif
constexpr uint32_t kGzTrialSymbols = HPN_GZ_TRIAL_SYMBOLS;

// does a dynamic block that decodes to text start at bit p, with something that begins like a block behind it?
__device__ __forceinline__ bool gz_trial(InfLds &s, const uint8_t *__restrict__ comp, uint64_t comp_len, uint64_t p)
{
    const uint8_t *in = comp + (p >> 3);
    const uint64_t room = comp_len - (p >> 3);
    const uint32_t in_len = room > 0x7fffff00ull ? 0x7fffff00u : (uint32_t)room;
    Bits b;
    stage(s, b, in, in_len);
    stage(s, b, in, in_len);
    refill(s, b, in, in_len);
    drop(b, (uint32_t)(p & 7u));
    refill(s, b, in, in_len);
    pin(b);
    if (take(b, 1) != 0 || take(b, 2) != 2u) return false;
    if (block_tables(s, b, in, in_len, 2) != 0) return false;
    uint32_t err = 0;
    Pos at = pos_of(b);
#ifdef HPN_GZ_TRIAL_FULL
    TextCheck all{1u << 20, 0u, 0u, 0u};
    if (!decode_symbols(s, b, at, in, in_len, all, err) || all.op == 0 || all.bad) return false;
#else
    // The block's first kGzTrialSymbols units, not the block to its end (round 4): a block of FASTQ text is ~10^5 symbols, ~15 ms
    // of one wave -- the search re-decoded a quarter of what the inflate kernel decodes (k_gz_find_starts 30.7 ms beside 76 ms of
    // k_gz_sym_inflate per batch, profiles/r04/kernel_stats_gz_tool.csv).
    TextCheck all{kGzTrialSymbols, 0u, 0u, 0u};
    if (!decode_symbols(s, b, at, in, in_len, all, err)) return err == 12u && !all.bad && all.op != 0;   // (12: the symbols are through)
    if (all.op == 0 || all.bad) return false;
#endif
    // ... and the next block must at least begin like one (header parses, tables build, the first symbols decode to text):
    // decoding it to its end as well would double the cost for nothing -- whoever uses the start proves it anyway
    seek(s, b, at, in, in_len);
    refill(s, b, in, in_len);
    pin(b);
    if (b.in_pos - (b.bc >> 3) > in_len) return false;
    (void)take(b, 1);
    const uint32_t type = take(b, 2);
    if (type == 3u) return false;
    if (type == 0u) {
        drop(b, b.bc & 7u);
        refill(s, b, in, in_len);
        const uint32_t len = take(b, 16);
        refill(s, b, in, in_len);
        return (len ^ take(b, 16)) == 0xffffu;
    }
    if (block_tables(s, b, in, in_len, type) != 0) return false;
    TextCheck some{256u, 0u, 0u, 0u};
    at = pos_of(b);
    err = 0;
    if (!decode_symbols(s, b, at, in, in_len, some, err) && err != 12u) return false;   // (12: the 256 symbols are through)
    return !some.bad;
}

// The scan (round 6).  Round 4's loop looked at 64 bit positions per pass -- two unaligned 8-byte loads per lane and the whole
// header test, Kraft sum included, in every lane for every position: ~2.3 vector instructions per position behind a load's
// latency, and a batch's search took 13 ms of a wave per slice (profiles/r05/kernel_stats_gz_tool_final.csv: 19 % of the
// gzip route's device time).  Now a lane owns a BYTE: one 16-byte load (the next pass's is in flight while this one is
// looked at) serves its eight bit positions; the three cheap header tests (BFINAL 0 / BTYPE 2, HLIT <= 29, HDIST <= 29: one
// add and one masked compare, see below) leave one position in nine, and only those pay for the Kraft sum of the code-length
// code -- 512 positions per pass at ~0.8 instructions per position.  What passes is tried in stream order as before.
struct GzFindDiag {   // -DHPN_FIND_DIAG builds: what the search spent where (summed over the slices of a launch)
    unsigned long long passes, trials, hits, clk_scan, clk_false, clk_true;
};
#ifdef HPN_FIND_DIAG
__device__ GzFindDiag g_find_diag;
#endif

// Kraft sum of the code-length code whose 3-bit lengths start 17 bits behind bit `sh` of the 128 bits (a0 .. a3), `hclen` of
// them (RFC 1951 3.2.7): 128 = complete.  Lengths beyond hclen are masked to 0, and a length of 0 counts nothing.
__device__ __forceinline__ uint32_t gz_kraft(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t sh, uint32_t hclen)
{
    const uint32_t s = 17u + sh;                                                    // 17 .. 24
    uint32_t y0 = __builtin_amdgcn_alignbit(a1, a0, s), y1 = __builtin_amdgcn_alignbit(a2, a1, s);
    const uint32_t nb = 3u * hclen;                                                 // 12 .. 57 bits of lengths
    y0 &= nb >= 32u ? 0xffffffffu : (1u << nb) - 1u;
    y1 &= nb > 32u ? (1u << (nb - 32u)) - 1u : 0u;
    uint32_t k = 0;
#pragma unroll
    for (uint32_t i = 0; i < 10u; ++i) k += (128u >> ((y0 >> (3u * i)) & 7u)) & 127u;
    k += (128u >> ((y0 >> 30 | y1 << 2) & 7u)) & 127u;                              // length 10 lies across the two words
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) k += (128u >> ((y1 >> (1u + 3u * i)) & 7u)) & 127u;
    return k;
}

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(HPN_INF_EU, 8))) void k_gz_find_starts(const uint8_t *__restrict__ comp, uint64_t comp_len, const GzSlice *__restrict__ slices,
                                                          uint32_t n, uint64_t *__restrict__ found)
{
    __shared__ InfLds s;
    const uint32_t lane = (uint32_t)lane_id();
    for (uint32_t si = blockIdx.x; si < n; si += gridDim.x) {
        const uint64_t lo = uni64(slices[si].lo), hi = uni64(slices[si].hi);
        uint64_t hit = ~0ull;
#ifdef HPN_FIND_DIAG
        unsigned long long d_pass = 0, d_trial = 0, d_false = 0, d_true = 0;
        const unsigned long long d_t0 = clock64();
#endif
        // the 16 bytes from byte B on; nothing where a header's 81 bits would not fit any more (the old test: byte + 16 <= comp_len)
        auto fetch = [&](uint64_t B) {
            u32 v = {0, 0, 0, 0};
            if (B + 16u <= comp_len) __builtin_memcpy(&v, comp + B, 16);
            return v;
        };
        uint64_t b0 = lo >> 3;                              // the pass's first byte; lane k looks at byte b0 + k
        u32 cur = fetch(b0 + lane);
        for (; b0 * 8u < hi && hit == ~0ull; b0 += kWave) {
            const u32 w = cur;
            cur = fetch(b0 + kWave + lane);                 // (in flight while this pass is looked at)
            const uint64_t B = b0 + lane;
            // BFINAL = 0, BTYPE = 2 (bits 0-2 = 4), HLIT <= 29 (bits 3-7: not 30, 31 = bits 4-7 not all ones), HDIST <= 29 (bits
            // 8-12 likewise: bits 9-12): adding 1 to each of the two 4-bit fields carries into the (masked-out) bit above it
            // exactly when the field is all ones -- one mask, one add, one masked compare per position
            uint32_t m8 = 0;
#pragma unroll
            for (uint32_t sh = 0; sh < 8u; ++sh) {
                const uint32_t v = __builtin_amdgcn_alignbit(w[1], w[0], sh);
                const uint32_t t = (v & (7u | 0xfu << 4 | 0xfu << 9)) + (1u << 4 | 1u << 9);
                m8 |= (t & (7u | 1u << 8 | 1u << 13)) == 4u ? 1u << sh : 0u;
            }
            // positions outside [lo, hi) (the first and the last pass), and bytes that had no room
            {
                const uint64_t p = B * 8u;
                if (p < lo) m8 = p + 8u <= lo ? 0u : m8 & (0xffu << (uint32_t)(lo - p));
                if (p + 8u > hi) m8 = p >= hi ? 0u : m8 & ((1u << (uint32_t)(hi - p)) - 1u);
                if (B + 16u > comp_len) m8 = 0u;
            }
            uint32_t k8 = 0;
            while (__builtin_amdgcn_ballot_w64(m8 != 0u)) {
                if (m8) {
                    const uint32_t sh = (uint32_t)__builtin_ctz(m8);
                    m8 &= m8 - 1u;
                    const uint32_t hclen = ((__builtin_amdgcn_alignbit(w[1], w[0], sh) >> 13) & 15u) + 4u;
                    if (gz_kraft(w[0], w[1], w[2], sh, hclen) == 128u) k8 |= 1u << sh;
                }
            }
#ifdef HPN_FIND_DIAG
            ++d_pass;
#endif
            for (uint64_t m = __builtin_amdgcn_ballot_w64(k8 != 0u); m && hit == ~0ull; m &= m - 1) {
                const uint32_t l = (uint32_t)__builtin_ctzll(m);
                for (uint32_t km = lane_of(k8, l); km && hit == ~0ull; km &= km - 1u) {
                    const uint64_t q = (b0 + l) * 8u + (uint32_t)__builtin_ctz(km);
#ifdef HPN_FIND_DIAG
                    const unsigned long long c0 = clock64();
                    const bool ok = gz_trial(s, comp, comp_len, q);
                    ++d_trial;
                    (ok ? d_true : d_false) += clock64() - c0;
                    if (ok) hit = q;
#else
                    if (gz_trial(s, comp, comp_len, q)) hit = q;
#endif
                }
            }
        }
        if (lane == 0) found[si] = hit;
#ifdef HPN_FIND_DIAG
        if (lane == 0) {
            atomicAdd(&g_find_diag.passes, d_pass), atomicAdd(&g_find_diag.trials, d_trial), atomicAdd(&g_find_diag.hits, hit != ~0ull ? 1ull : 0ull);
            atomicAdd(&g_find_diag.clk_scan, clock64() - d_t0 - d_false - d_true), atomicAdd(&g_find_diag.clk_false, d_false), atomicAdd(&g_find_diag.clk_true, d_true);
        }
#endif
    }
}

// One small workgroup: histories in stream order.  windows + k * 32768 = the 32 KiB before stretch k (n_chunks + 1 of
// them: the last is the history after the batch).  The histories live in global memory, not LDS, and the workgroup is 4
// waves: it has to find room on a chip whose CUs are packed with inflate waves of other contexts (18 x 8.7 KiB of LDS),
// and with a 64 KiB LDS image it waited for up to a whole stretch time (0.17 s) to be placed.  Each step reads a region
// that was written in the step before (never read earlier, so no stale cache line), behind a barrier.
// (kGzWinThreads = 1024 when the call holds enough stretches to have had the chip to itself: 32 positions per thread and step.)
template <int kGzWinThreads>
__global__ __launch_bounds__(kGzWinThreads) void k_gz_windows(const uint16_t *__restrict__ symbuf, uint32_t sym_cap,
                                                              GzMeta *__restrict__ meta, uint32_t n_chunks,
                                                              const uint8_t *__restrict__ window_in, uint8_t *windows,
                                                              uint8_t *__restrict__ window_out, u64 *__restrict__ summary)
{
    const int tid = threadIdx.x;
    for (uint32_t j = (uint32_t)tid; j < kGzHist; j += kGzWinThreads) windows[j] = window_in ? window_in[j] : 0;
    __syncthreads();
    u64 text = 0;
    uint32_t bad = 0, bad_at = 0, fin = 0;
    // what position j0 of the history behind stretch k is, before the look-up: a symbol of the stretch's last 32 KiB, or --
    // the stretch is shorter than that -- the history in front of it itself
    // (a wave-uniform base + the lane's 32-bit j: one register per address)
    auto tail = [&](uint32_t k, uint32_t j) -> uint32_t {
        const int32_t d = (int32_t)meta[k].n_out - (int32_t)kGzHist;            // symbol j of the history = symbol d + j of the stretch
        const uint16_t *base = symbuf + (uint64_t)k * sym_cap + (int64_t)d;
        return (int32_t)j + d >= 0 ? (uint32_t)base[j] : 256u + (uint32_t)((int32_t)kGzHist + (int32_t)j + d);
    };
    if (kGzWinThreads >= 1024) {
        // 32 positions per thread, ALL their symbols loaded while the stretch before is resolved (they do not depend on it):
        // what is left on the chain per stretch is one round of look-ups, the stores and the barrier (~2.5 us; with the
        // symbols loaded inside the step, in four batches, it was eight dependent round trips = 12.8 us, 59 ms for 4,608
        // stretches beside 154 ms of inflating them)
        constexpr int kPer = (int)(kGzHist / 1024);
        uint32_t sv[kPer], nx[kPer / 2];                      // (the symbols in flight two to a register: 128 VGPRs are all there are)
#pragma unroll
        for (int q = 0; q < kPer; ++q) sv[q] = n_chunks ? tail(0, (uint32_t)tid + (uint32_t)q * 1024u) : 0u;
        for (uint32_t k = 0; k < n_chunks; ++k) {
            const uint32_t n = meta[k].n_out, st = meta[k].status;
            if (st && !bad) bad = st, bad_at = k;
            if (meta[k].final_block) fin = k + 1u;
            if (tid == 0) meta[k].text_off = text;
            text += n;
            const uint8_t *cur = windows + (uint64_t)k * kGzHist;
            uint8_t *nxt = windows + (uint64_t)(k + 1) * kGzHist;
            if (k + 1 < n_chunks) {
#pragma unroll
                for (int q = 0; q < kPer / 2; ++q)
                    nx[q] = tail(k + 1, (uint32_t)tid + (uint32_t)(2 * q) * 1024u) | tail(k + 1, (uint32_t)tid + (uint32_t)(2 * q + 1) * 1024u) << 16;
            }
            uint8_t v[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) v[q] = sv[q] < 256u ? (uint8_t)sv[q] : cur[sv[q] - 256u];
#pragma unroll
            for (int q = 0; q < kPer; ++q) nxt[(uint32_t)tid + (uint32_t)q * 1024u] = v[q];
#pragma unroll
            for (int q = 0; q < kPer; ++q) sv[q] = (q & 1) ? nx[q / 2] >> 16 : nx[q / 2] & 0xffffu;
            __syncthreads();
        }
    } else {
    for (uint32_t k = 0; k < n_chunks; ++k) {
        const uint32_t n = meta[k].n_out, st = meta[k].status;
        if (st && !bad) bad = st, bad_at = k;
        if (meta[k].final_block) fin = k + 1u;
        if (tid == 0) meta[k].text_off = text;
        text += n;
        const uint8_t *cur = windows + (uint64_t)k * kGzHist;
        uint8_t *nxt = windows + (uint64_t)(k + 1) * kGzHist;
        // 128 positions per thread, 16 at a time: 16 independent symbol loads, then 16 independent history look-ups
        constexpr int kBatch = 16;
        for (uint32_t j0 = (uint32_t)tid; j0 < kGzHist; j0 += kGzWinThreads * kBatch) {
            uint32_t sv[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; ++q) sv[q] = tail(k, j0 + (uint32_t)q * kGzWinThreads);
            uint8_t v[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; ++q) v[q] = sv[q] < 256u ? (uint8_t)sv[q] : cur[sv[q] - 256u];
#pragma unroll
            for (int q = 0; q < kBatch; ++q) nxt[j0 + (uint32_t)q * kGzWinThreads] = v[q];
        }
        __syncthreads();
    }
    }
    if (window_out) {
        const uint8_t *last = windows + (uint64_t)n_chunks * kGzHist;
        for (uint32_t j = (uint32_t)tid; j < kGzHist; j += kGzWinThreads) window_out[j] = last[j];
    }
    if (tid == 0) summary[0] = text, summary[1] = bad, summary[2] = bad_at, summary[3] = fin;
}

// The same walk with the history in LDS, for a call that has the chip to itself (one worker: nobody else's decoders hold the
// CUs' LDS, which is what made a 64 KiB workgroup wait for a whole stretch time when several contexts inflate side by side).
// A thread takes eight groups of four consecutive positions: the four symbols in one 8-byte load (the NEXT stretch's, in flight
// while this one is resolved), the placeholders among them looked up in LDS, the four bytes written as one dword to the other
// half of the LDS image and to the global copy k_gz_translate reads: 5,120 stretches 25.5 -> 17 ms (3.3 us per stretch; the
// prefetch's loads still end up on the chain: the compiler's s_waitcnt placement waits for the newest load where an older one
// is used, whichever way the sets are rotated -- deeper prefetching by hand gave the same 17 ms or worse).
__global__ __launch_bounds__(1024) void k_gz_windows_lds(const uint16_t *__restrict__ symbuf, uint32_t sym_cap, GzMeta *__restrict__ meta,
                                                         uint32_t n_chunks, const uint8_t *__restrict__ window_in, uint8_t *windows,
                                                         uint8_t *__restrict__ window_out, u64 *__restrict__ summary)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[2][kGzHist];
    constexpr int kGroups = (int)(kGzHist / (1024 * 4));           // 8
    const uint32_t tid = threadIdx.x;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    for (int g = 0; g < kGroups; ++g) {
        const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
        uint32_t v = 0;
        if (window_in) __builtin_memcpy(&v, window_in + j, 4);
        *(uint32_t *)(s_win[0] + j) = v;
        *(uint32_t *)(windows + j) = v;
    }
    // the four symbols of group g of the history behind stretch k, before the look-up (two to a register)
    auto tail4 = [&](uint32_t k, uint32_t j) -> u32x2 {
        const int32_t d = (int32_t)meta[k].n_out - (int32_t)kGzHist;   // symbol j of the history = symbol d + j of the stretch
        const uint16_t *base = symbuf + (uint64_t)k * sym_cap + (int64_t)d;
        u32x2 r;
        if ((int32_t)j + d >= 0) {
            __builtin_memcpy(&r, base + j, 8);
        } else {
            uint32_t e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = (int32_t)(j + i) + d >= 0 ? (uint32_t)base[j + i] : 256u + (uint32_t)((int32_t)kGzHist + (int32_t)(j + i) + d);
            r = u32x2{e[0] | e[1] << 16, e[2] | e[3] << 16};
        }
        return r;
    };
    u32x2 sv[kGroups], nx[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) sv[g] = n_chunks ? tail4(0, ((uint32_t)g * 1024u + tid) * 4u) : u32x2{0, 0};
    __syncthreads();
    u64 text = 0;
    uint32_t bad = 0, bad_at = 0, fin = 0;
    for (uint32_t k = 0; k < n_chunks; ++k) {
        const uint32_t n = meta[k].n_out, st = meta[k].status;
        if (st && !bad) bad = st, bad_at = k;
        if (meta[k].final_block) fin = k + 1u;
        if (tid == 0) meta[k].text_off = text;
        text += n;
        const uint8_t *cur = s_win[k & 1];
        uint8_t *nxt = s_win[(k + 1) & 1], *gout = windows + (uint64_t)(k + 1) * kGzHist;
        if (k + 1 < n_chunks) {
#pragma unroll
            for (int g = 0; g < kGroups; ++g) nx[g] = tail4(k + 1, ((uint32_t)g * 1024u + tid) * 4u);
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
            const uint32_t e0 = sv[g][0] & 0xffffu, e1 = sv[g][0] >> 16, e2 = sv[g][1] & 0xffffu, e3 = sv[g][1] >> 16;
            const uint32_t b0 = e0 < 256u ? e0 : cur[e0 - 256u], b1 = e1 < 256u ? e1 : cur[e1 - 256u];
            const uint32_t b2 = e2 < 256u ? e2 : cur[e2 - 256u], b3 = e3 < 256u ? e3 : cur[e3 - 256u];
            const uint32_t v = b0 | b1 << 8 | b2 << 16 | b3 << 24;
            *(uint32_t *)(nxt + j) = v;
            *(uint32_t *)(gout + j) = v;
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) sv[g] = nx[g];
        __syncthreads();
    }
    if (window_out) {
        const uint8_t *last = s_win[n_chunks & 1];
        for (int g = 0; g < kGroups; ++g) {
            const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
            *(uint32_t *)(window_out + j) = *(const uint32_t *)(last + j);
        }
    }
    if (tid == 0) summary[0] = text, summary[1] = bad, summary[2] = bad_at, summary[3] = fin;
}

// ---- the histories in three steps (round 4) ------------------------------------------------------------------------------------
// The walk above is a chain of n_chunks steps (3.3 - 3.8 us each: 23 ms per batch of 6,000 stretches, whatever their size -- a
// fifth of a batch's device time in the tools).  But what a stretch does to the history is a MAP: every byte of the history
// behind it is a literal or a position of the history in front of it -- and maps compose.  So:
//   k_gz_win_maps    a workgroup per GROUP of kGzGroup consecutive stretches composes the group's map (32,768 entries of the
//                    symbols' own encoding: < 256 a literal, else 256 + position in the history in front of the group), the
//                    stretches one after the other in LDS -- the groups side by side;
//   k_gz_win_chain   one workgroup walks the GROUPS (n / kGzGroup steps): the history in front of every group, the one behind
//                    the batch; and the stretches' text offsets, the first bad stretch, the final one (what the walk above
//                    gathered on its way);
//   k_gz_win_apply   a workgroup per group again: the walk above, from the group's own history, writing windows[k].
// Twice the look-ups, a chain of 2 x kGzGroup + n / kGzGroup steps instead of n.
constexpr uint32_t kGzGroup = 64;

// the four symbols of positions j .. j + 3 of the history behind stretch k, before the look-up (two to a register)
typedef uint32_t gz_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gz_u32x2 gz_tail4(const uint16_t *__restrict__ symbuf, uint32_t sym_cap, const GzMeta *__restrict__ meta, uint32_t k, uint32_t j)
{
    const int32_t d = (int32_t)meta[k].n_out - (int32_t)kGzHist;   // symbol j of the history = symbol d + j of the stretch
    const uint16_t *base = symbuf + (uint64_t)k * sym_cap + (int64_t)d;
    gz_u32x2 r;
    if ((int32_t)j + d >= 0) {
        __builtin_memcpy(&r, base + j, 8);
    } else {
        uint32_t e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = (int32_t)(j + i) + d >= 0 ? (uint32_t)base[j + i] : 256u + (uint32_t)((int32_t)kGzHist + (int32_t)(j + i) + d);
        r = gz_u32x2{e[0] | e[1] << 16, e[2] | e[3] << 16};
    }
    return r;
}

__global__ __launch_bounds__(1024) void k_gz_win_maps(const uint16_t *__restrict__ symbuf, uint32_t sym_cap, const GzMeta *__restrict__ meta,
                                                      uint32_t n_chunks, uint16_t *__restrict__ gmaps)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_map[2][kGzHist];       // 128 KB
    constexpr int kGroups = (int)(kGzHist / (1024 * 4));           // 8 groups of four consecutive positions per thread
    const uint32_t tid = threadIdx.x, k0 = blockIdx.x * kGzGroup, k1 = k0 + kGzGroup < n_chunks ? k0 + kGzGroup : n_chunks;
    for (int g = 0; g < kGroups; ++g) {
        const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
        *(gz_u32x2 *)(s_map[0] + j) = gz_u32x2{(256u + j) | (257u + j) << 16, (258u + j) | (259u + j) << 16};     // the identity
    }
    gz_u32x2 sv[kGroups], nx[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) sv[g] = gz_tail4(symbuf, sym_cap, meta, k0, ((uint32_t)g * 1024u + tid) * 4u);
    __syncthreads();
    for (uint32_t k = k0; k < k1; ++k) {
        const uint16_t *cur = s_map[(k - k0) & 1];
        uint16_t *nxt = s_map[(k - k0 + 1) & 1];
        if (k + 1 < k1) {
#pragma unroll
            for (int g = 0; g < kGroups; ++g) nx[g] = gz_tail4(symbuf, sym_cap, meta, k + 1, ((uint32_t)g * 1024u + tid) * 4u);
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
            const uint32_t e0 = sv[g][0] & 0xffffu, e1 = sv[g][0] >> 16, e2 = sv[g][1] & 0xffffu, e3 = sv[g][1] >> 16;
            const uint32_t m0 = e0 < 256u ? e0 : cur[e0 - 256u], m1 = e1 < 256u ? e1 : cur[e1 - 256u];
            const uint32_t m2 = e2 < 256u ? e2 : cur[e2 - 256u], m3 = e3 < 256u ? e3 : cur[e3 - 256u];
            *(gz_u32x2 *)(nxt + j) = gz_u32x2{m0 | m1 << 16, m2 | m3 << 16};
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) sv[g] = nx[g];
        __syncthreads();
    }
    const uint16_t *fin = s_map[(k1 - k0) & 1];
    uint16_t *out = gmaps + (uint64_t)blockIdx.x * kGzHist;
    for (int g = 0; g < kGroups; ++g) {
        const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
        *(gz_u32x2 *)(out + j) = *(const gz_u32x2 *)(fin + j);
    }
}

// group_in + g * 32768 = the history in front of group g; windows_last = the one behind the batch (windows[n_chunks])
__global__ __launch_bounds__(1024) void k_gz_win_chain(const uint16_t *__restrict__ gmaps, uint32_t n_groups, GzMeta *__restrict__ meta,
                                                       uint32_t n_chunks, const uint8_t *__restrict__ window_in, uint8_t *__restrict__ group_in,
                                                       uint8_t *__restrict__ windows_last, uint8_t *__restrict__ window_out, u64 *__restrict__ summary)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[2][kGzHist];
    __shared__ u64 s_wave[16];
    __shared__ u64 s_carry;
    __shared__ uint32_t s_bad_at, s_fin;
    constexpr int kGroups = (int)(kGzHist / (1024 * 4));
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_carry = 0, s_bad_at = 0xffffffffu, s_fin = 0;
    for (int g = 0; g < kGroups; ++g) {
        const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
        uint32_t v = 0;
        if (window_in) __builtin_memcpy(&v, window_in + j, 4);
        *(uint32_t *)(s_win[0] + j) = v;
    }
    __syncthreads();
    // the stretches' text offsets (an exclusive scan of their sizes), the first stretch that failed, the final one
    for (uint32_t i0 = 0; i0 < n_chunks; i0 += 1024u) {
        const uint32_t i = i0 + tid;
        const u64 v = i < n_chunks ? meta[i].n_out : 0;
        if (i < n_chunks) {
            if (meta[i].status) atomicMin(&s_bad_at, i);
            if (meta[i].final_block) atomicMax(&s_fin, i + 1u);
        }
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
        if (i < n_chunks) meta[i].text_off = before + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + inc;
        __syncthreads();
    }
    gz_u32x2 mv[kGroups], nx[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) mv[g] = n_groups ? *(const gz_u32x2 *)(gmaps + ((uint32_t)g * 1024u + tid) * 4u) : gz_u32x2{0, 0};
    for (uint32_t gi = 0; gi < n_groups; ++gi) {
        const uint8_t *cur = s_win[gi & 1];
        uint8_t *nxt = s_win[(gi + 1) & 1], *gout = group_in + (uint64_t)gi * kGzHist;
        if (gi + 1 < n_groups) {
#pragma unroll
            for (int g = 0; g < kGroups; ++g) nx[g] = *(const gz_u32x2 *)(gmaps + (uint64_t)(gi + 1) * kGzHist + ((uint32_t)g * 1024u + tid) * 4u);
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
            *(uint32_t *)(gout + j) = *(const uint32_t *)(cur + j);                 // the history in front of this group
            const uint32_t e0 = mv[g][0] & 0xffffu, e1 = mv[g][0] >> 16, e2 = mv[g][1] & 0xffffu, e3 = mv[g][1] >> 16;
            const uint32_t b0 = e0 < 256u ? e0 : cur[e0 - 256u], b1 = e1 < 256u ? e1 : cur[e1 - 256u];
            const uint32_t b2 = e2 < 256u ? e2 : cur[e2 - 256u], b3 = e3 < 256u ? e3 : cur[e3 - 256u];
            *(uint32_t *)(nxt + j) = b0 | b1 << 8 | b2 << 16 | b3 << 24;
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) mv[g] = nx[g];
        __syncthreads();
    }
    const uint8_t *last = s_win[n_groups & 1];
    for (int g = 0; g < kGroups; ++g) {
        const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
        const uint32_t v = *(const uint32_t *)(last + j);
        *(uint32_t *)(windows_last + j) = v;
        if (window_out) *(uint32_t *)(window_out + j) = v;
    }
    if (tid == 0) {
        const uint32_t bad_at = s_bad_at;
        summary[0] = s_carry, summary[1] = bad_at != 0xffffffffu ? meta[bad_at].status : 0u, summary[2] = bad_at != 0xffffffffu ? bad_at : 0u,
        summary[3] = s_fin;
    }
}

// windows + k * 32768 = the history in front of stretch k, for the stretches of group blockIdx.x
__global__ __launch_bounds__(1024) void k_gz_win_apply(const uint16_t *__restrict__ symbuf, uint32_t sym_cap, const GzMeta *__restrict__ meta,
                                                       uint32_t n_chunks, const uint8_t *__restrict__ group_in, uint8_t *__restrict__ windows)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[2][kGzHist];
    constexpr int kGroups = (int)(kGzHist / (1024 * 4));
    const uint32_t tid = threadIdx.x, k0 = blockIdx.x * kGzGroup, k1 = k0 + kGzGroup < n_chunks ? k0 + kGzGroup : n_chunks;
    for (int g = 0; g < kGroups; ++g) {
        const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
        const uint32_t v = *(const uint32_t *)(group_in + (uint64_t)blockIdx.x * kGzHist + j);
        *(uint32_t *)(s_win[0] + j) = v;
        *(uint32_t *)(windows + (uint64_t)k0 * kGzHist + j) = v;
    }
    gz_u32x2 sv[kGroups], nx[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) sv[g] = gz_tail4(symbuf, sym_cap, meta, k0, ((uint32_t)g * 1024u + tid) * 4u);
    __syncthreads();
    for (uint32_t k = k0; k + 1 < k1; ++k) {              // (the history behind the group's last stretch is the next group's: k_gz_win_chain)
        const uint8_t *cur = s_win[(k - k0) & 1];
        uint8_t *nxt = s_win[(k - k0 + 1) & 1], *gout = windows + (uint64_t)(k + 1) * kGzHist;
        if (k + 2 < k1) {
#pragma unroll
            for (int g = 0; g < kGroups; ++g) nx[g] = gz_tail4(symbuf, sym_cap, meta, k + 1, ((uint32_t)g * 1024u + tid) * 4u);
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const uint32_t j = ((uint32_t)g * 1024u + tid) * 4u;
            const uint32_t e0 = sv[g][0] & 0xffffu, e1 = sv[g][0] >> 16, e2 = sv[g][1] & 0xffffu, e3 = sv[g][1] >> 16;
            const uint32_t b0 = e0 < 256u ? e0 : cur[e0 - 256u], b1 = e1 < 256u ? e1 : cur[e1 - 256u];
            const uint32_t b2 = e2 < 256u ? e2 : cur[e2 - 256u], b3 = e3 < 256u ? e3 : cur[e3 - 256u];
            const uint32_t v = b0 | b1 << 8 | b2 << 16 | b3 << 24;
            *(uint32_t *)(nxt + j) = v;
            *(uint32_t *)(gout + j) = v;
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) sv[g] = nx[g];
        __syncthreads();
    }
}

// symbols -> bytes: blockIdx.y = stretch, the x blocks stride over its symbols (16 per thread and step).  The stretch's
// 32 KiB of history lie in LDS: a placeholder is a ds_read_u8, not a second trip to memory behind the symbols' own
// (names and other text repeated record after record are placeholders all the way down a stretch, not only in its
// first 32 KiB: a copy of a placeholder is a placeholder).
constexpr int kGzTrThreads = 256;
constexpr int kGzTrBlocks = 16;   // per stretch: the history is read into LDS once per ~100 KB of text
__global__ __launch_bounds__(kGzTrThreads) void k_gz_translate(const uint16_t *__restrict__ symbuf, uint32_t sym_cap,
                                                               const GzMeta *__restrict__ meta, const uint8_t *__restrict__ windows,
                                                               uint8_t *__restrict__ text)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[kGzHist];
    const uint32_t k = blockIdx.y;
    const uint32_t n = meta[k].n_out;
    const uint16_t *sym = symbuf + (uint64_t)k * sym_cap;  // 16-byte aligned when sym_cap % 8 == 0
    const uint8_t *wk = windows + (uint64_t)k * kGzHist;
    uint8_t *dst = text + meta[k].text_off;
    const uint32_t groups = n / 16u;
    if (blockIdx.x * kGzTrThreads >= groups && blockIdx.x != 0) return;
    for (uint32_t i = threadIdx.x * 16u; i < kGzHist; i += kGzTrThreads * 16u) *reinterpret_cast<u32 *>(s_win + i) = *reinterpret_cast<const u32 *>(wk + i);
    __syncthreads();
    auto byte_of = [&](uint32_t e) -> uint32_t { return e < 256u ? e : (uint32_t)s_win[(e - 256u) & (kGzHist - 1u)]; };
    for (uint32_t g = blockIdx.x * kGzTrThreads + threadIdx.x; g < groups; g += gridDim.x * kGzTrThreads) {
        const u32 v0 = *reinterpret_cast<const u32 *>(sym + (uint64_t)g * 16u), v1 = *reinterpret_cast<const u32 *>(sym + (uint64_t)g * 16u + 8u);
        u32 o;
        if (((v0[0] | v0[1] | v0[2] | v0[3] | v1[0] | v1[1] | v1[2] | v1[3]) & 0xff00ff00u) == 0u) {  // bytes only: the low bytes of the 16 halves
            o[0] = __builtin_amdgcn_perm(v0[1], v0[0], 0x06040200u);
            o[1] = __builtin_amdgcn_perm(v0[3], v0[2], 0x06040200u);
            o[2] = __builtin_amdgcn_perm(v1[1], v1[0], 0x06040200u);
            o[3] = __builtin_amdgcn_perm(v1[3], v1[2], 0x06040200u);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                o[q] = byte_of(v0[2 * q] & 0xffffu) | byte_of(v0[2 * q] >> 16) << 8 | byte_of(v0[2 * q + 1] & 0xffffu) << 16 | byte_of(v0[2 * q + 1] >> 16) << 24;
                o[2 + q] = byte_of(v1[2 * q] & 0xffffu) | byte_of(v1[2 * q] >> 16) << 8 | byte_of(v1[2 * q + 1] & 0xffffu) << 16 | byte_of(v1[2 * q + 1] >> 16) << 24;
            }
        }
        __builtin_memcpy(dst + (uint64_t)g * 16u, &o, 16);  // dst is not aligned in general
    }
    if (blockIdx.x == 0)
        for (uint32_t i = groups * 16u + threadIdx.x; i < n; i += kGzTrThreads) dst[i] = (uint8_t)byte_of(sym[i]);
}

// d_bounds: [0] = count (uint32, cleared here), entries from byte 16 on; nullptr: a final block ends its stretch (one member)
hipError_t launch_gz_sym_inflate(const uint8_t *d_comp, const void *d_chunks, uint32_t n_chunks, uint16_t *d_sym, uint32_t sym_cap,
                                 void *d_meta, void *d_bounds, uint32_t bounds_cap, int n_cu, hipStream_t st)
{
    if (n_chunks == 0) return hipSuccess;
    if (d_bounds) {
        hipError_t e = hipMemsetAsync(d_bounds, 0, 16, st);
        if (e != hipSuccess) return e;
    }
    const uint32_t cap = (uint32_t)n_cu * kInflateWavesPerCu;
    hipLaunchKernelGGL(k_gz_sym_inflate, dim3(n_chunks < cap ? n_chunks : cap), dim3(kWave), 0, st, d_comp, (const GzChunk *)d_chunks,
                       n_chunks, d_sym, sym_cap, (GzMeta *)d_meta, d_bounds ? (GzBound *)((uint8_t *)d_bounds + 16) : nullptr, bounds_cap,
                       (uint32_t *)d_bounds);
    return hipGetLastError();
}
hipError_t launch_gz_find_starts(const uint8_t *d_comp, uint64_t comp_len, const void *d_slices, uint32_t n, uint64_t *d_found, int n_cu,
                                 hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint32_t cap = (uint32_t)n_cu * kInflateWavesPerCu;
#ifdef HPN_FIND_DIAG
    GzFindDiag z{};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_find_diag), &z, sizeof z);
#endif
    hipLaunchKernelGGL(k_gz_find_starts, dim3(n < cap ? n : cap), dim3(kWave), 0, st, d_comp, comp_len, (const GzSlice *)d_slices, n, d_found);
#ifdef HPN_FIND_DIAG
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(&z, HIP_SYMBOL(g_find_diag), sizeof z);
    fprintf(stderr, "[find diag] %u slices: %llu passes of 512 positions, %llu trials (%llu hits); clocks per slice: scan %.0f, false trials %.0f, the true trial %.0f\n", n,
            z.passes, z.trials, z.hits, (double)z.clk_scan / n, (double)z.clk_false / n, (double)z.clk_true / n);
#endif
    return hipGetLastError();
}
size_t gz_groups_bytes(uint32_t n_chunks) { return (size_t)((n_chunks + kGzGroup - 1) / kGzGroup) * kGzHist * 3 + 64; }   // maps (u16) + histories (u8)
hipError_t launch_gz_windows(const uint16_t *d_sym, uint32_t sym_cap, void *d_meta, uint32_t n_chunks, const uint8_t *d_window_in,
                             uint8_t *d_windows, uint8_t *d_window_out, u64 *d_summary, void *d_groups, int n_cu, hipStream_t st)
{
    // (a call of more than half a chip-fill of stretches comes from a worker that has the device to itself: tally_gz_on_gpu
    // divides the chip's 5,120 slots among the workers in flight)
    const char *how = test_env("HPN_GZ_WINDOWS");                  // groups / lds / global: tests and A/B runs
    if (d_groups && (how ? !strcmp(how, "groups") : n_chunks > (uint32_t)n_cu * 12u)) {
        const uint32_t ng = (n_chunks + kGzGroup - 1) / kGzGroup;
        uint16_t *gmaps = (uint16_t *)d_groups;
        uint8_t *group_in = (uint8_t *)d_groups + (size_t)ng * kGzHist * sizeof(uint16_t);
        hipLaunchKernelGGL(k_gz_win_maps, dim3(ng), dim3(1024), 0, st, d_sym, sym_cap, (const GzMeta *)d_meta, n_chunks, gmaps);
        hipLaunchKernelGGL(k_gz_win_chain, dim3(1), dim3(1024), 0, st, gmaps, ng, (GzMeta *)d_meta, n_chunks, d_window_in, group_in,
                           d_windows + (size_t)n_chunks * kGzHist, d_window_out, d_summary);
        hipLaunchKernelGGL(k_gz_win_apply, dim3(ng), dim3(1024), 0, st, d_sym, sym_cap, (const GzMeta *)d_meta, n_chunks, group_in, d_windows);
        return hipGetLastError();
    }
    if (how ? !strcmp(how, "lds") : n_chunks > (uint32_t)n_cu * 12u)
        hipLaunchKernelGGL(k_gz_windows_lds, dim3(1), dim3(1024), 0, st, d_sym, sym_cap, (GzMeta *)d_meta, n_chunks, d_window_in, d_windows,
                           d_window_out, d_summary);
    else if (n_chunks > (uint32_t)n_cu * 6u)
        hipLaunchKernelGGL(k_gz_windows<1024>, dim3(1), dim3(1024), 0, st, d_sym, sym_cap, (GzMeta *)d_meta, n_chunks, d_window_in,
                           d_windows, d_window_out, d_summary);
    else
        hipLaunchKernelGGL(k_gz_windows<256>, dim3(1), dim3(256), 0, st, d_sym, sym_cap, (GzMeta *)d_meta, n_chunks, d_window_in,
                           d_windows, d_window_out, d_summary);
    return hipGetLastError();
}
hipError_t launch_gz_translate(const uint16_t *d_sym, uint32_t sym_cap, const void *d_meta, uint32_t n_chunks, const uint8_t *d_windows,
                               uint8_t *d_text, hipStream_t st)
{
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gz_translate, dim3(kGzTrBlocks, n_chunks), dim3(kGzTrThreads), 0, st, d_sym, sym_cap, (const GzMeta *)d_meta, d_windows,
                       d_text);
    return hipGetLastError();
}

uint32_t inflate_waves_per_cu() { return kInflateWavesPerCu; }

}  // namespace hpn
