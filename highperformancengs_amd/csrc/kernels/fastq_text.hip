// fastq_text.hip -- gfx950 kernels of the raw-text front end (hpn_fastq_text_*).
//
// The reference frames a FASTQ stream with four gzgets() calls per record into
// one 1024-byte buffer (fastq_count.c:112-118, fastq_trim.c:67-89).  On REGULAR
// text -- no NUL byte, every line at most 1022 characters + '\n', whole
// records, quality line not shorter than the sequence line -- that loop is a
// pure function of the newline positions, and these kernels compute it on the
// device from the raw bytes:
//
//   k_text_lines    newline index: every workgroup counts the '\n' of its 64 KiB
//                   tile (SWAR on dwordx4 loads), one decoupled look-back chain
//                   (scan.hpp) turns the counts into global line numbers, and the
//                   position of line j's '\n' lands in nl[j].
//   k_text_records  one thread per record = 4 lines: checks the regularity
//                   conditions above, derives the record's length (count) or
//                   output size (trim) and scans it into off[] with a second
//                   look-back chain.  Everything irregular raises a flag; the host
//                   then frames that stream with the exact gzgets emulation.
//   k_text_gather   count: quality (and sequence) lines -> packed arrays for
//                   k_tally_scan / k_tally_hist.
//   k_text_trim     trim: writes "name\nseq[S:E]\n+\nqual[S:E]\n" (fastq_trim.c:101)
//                   for every record straight into the output text.
//
// Bound: HBM (one read of the text per kernel, 4 B per line, 8 B per record);
// in the tools the PCIe copy of the text is ~100x slower than these kernels.
#include "scan.hpp"

namespace hpn {

constexpr int kTxtThreads = 256;
constexpr int kLinesThreads = 512;                              // k_text_lines (256: 0.295, 512: 0.266, 1024: 0.258 ms per GiB)
constexpr int kTxtRows = 16;                                    // 16-byte words per thread
constexpr uint32_t kTxtTile = kLinesThreads * kTxtRows * 16;    // bytes per workgroup of k_text_lines
constexpr int kRecPerThread = 4;
constexpr uint32_t kRecTile = kTxtThreads * kRecPerThread;      // records per workgroup

// device state block (uint32 words), zeroed before every chunk
enum { kTsLines = 0, kTsRecs, kTsFlags, kTsUnterminated, kTsConsumed, kTsTotalLo, kTsTotalHi, kTsErr, kTsTicket1, kTsTicket2, kTsOwnLines, kTsWords = 16 };

__device__ __forceinline__ uint32_t zero_bytes(uint32_t x)  // 0x80 in exactly the bytes of x that are 0
{
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
}
__device__ __forceinline__ uint32_t pack4(uint32_t m)  // 0x80 flags of 4 bytes -> 4 bits
{
    return ((((m >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xfu;
}

// 0x80 in exactly the bytes of x that are '\n' (the high bit of x ^ 0x0a.. is the high bit of x)
__device__ __forceinline__ uint32_t newline_flags(uint32_t x)
{
    return ~((((x ^ 0x0a0a0a0au) & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}
// the 0x80 flags of two dwords -> 8 bits in byte order.  g holds a's flags at bits 8j and b's at 8j + 4; the 24-bit
// multiply (full rate; v_mul_lo_u32 is not) moves bytes 0..2 of both to bits 16 + j and 20 + j -- every partial product
// lands on a bit of its own, so nothing carries --, byte 3 of both is shifted there.
__device__ __forceinline__ uint32_t pack8(uint32_t fa, uint32_t fb)
{
    const uint32_t g = (fa >> 7) | (fb >> 3);
    uint32_t prod;  // (written as g * 0x10204 the compiler drops the 24-bit mask -- the bits looked at below do not need it -- and takes v_mul_lo_u32)
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(prod) : "v"(g), "s"(0x10204u));
    return ((prod | ((g >> 5) & 0x00880000u)) >> 16) & 0xffu;
}

__device__ __forceinline__ uint32_t wave_excl_scan32(uint32_t v, uint32_t &total)
{
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, kWave);
        if (lane_id() >= o) inc += t;
    }
    total = __shfl(inc, kWave - 1, kWave);
    return inc - v;
}

// inclusive prefix sum over the lanes of a wave (DPP: shifts inside rows of 16, then row broadcasts; see wave_sum)
__device__ __forceinline__ uint32_t wave_incl_dpp(uint32_t v)
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

constexpr int kTxtLoads = 16;  // 16-byte loads a lane has in flight (8 at six waves per SIMD: no faster, it spills)
constexpr uint32_t kTxtWaveBytes = kTxtTile / (kLinesThreads / kWave);  // a wave's contiguous part of a tile
constexpr uint32_t kTxtRowBytes = kWave * 16u;                         // what one load instruction of a wave covers

// The '\n' masks of a lane's 16 words (word k at base + k * kTxtRowBytes).  kEdge: the tile holds `begin`, `end`,
// the stream's last byte or `own_end` -- every word is cut to [begin, end) and counted up to own_end; otherwise the
// words are whole and `own_all` says whether the tile lies in front of own_end.
template <bool kEdge>
__device__ __forceinline__ void line_masks(const uint8_t *__restrict__ slot, uint32_t base, uint32_t begin, uint32_t end, int last,
                                           uint32_t own_end, bool own_all, uint32_t (&mask)[kTxtRows], uint32_t &nul,
                                           uint32_t &own, uint32_t *__restrict__ st)
{
#pragma unroll
    for (int h = 0; h < kTxtRows; h += kTxtLoads) {
        u32 w[kTxtLoads];
#pragma unroll
        for (int i = 0; i < kTxtLoads; ++i) {
            const uint32_t p = base + (uint32_t)(h + i) * kTxtRowBytes;
            if (!kEdge || p < end) w[i] = load_stream16((const u32 *)(slot + p));
            else w[i] = u32{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < kTxtLoads; ++i) {
            const uint32_t p = base + (uint32_t)(h + i) * kTxtRowBytes;
            uint32_t m = 0, z = 0;
            if (!kEdge) {  // whole words: 3 + 2 instructions per dword, 7 per pair of dwords (the exact form below: ~26 per dword)
                m = pack8(newline_flags(w[i][0]), newline_flags(w[i][1])) | (pack8(newline_flags(w[i][2]), newline_flags(w[i][3])) << 8);
#pragma unroll
                for (int d = 0; d < 4; ++d) nul |= (w[i][d] - 0x01010101u) & ~w[i][d];  // & 0x80808080 != 0 <=> a byte of some word was 0
                if (own_all) own += (uint32_t)__builtin_popcount(m);
                mask[h + i] = m;
                continue;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                m |= pack4(zero_bytes(w[i][d] ^ 0x0a0a0a0au)) << (4 * d);
                z |= pack4(zero_bytes(w[i][d])) << (4 * d);
            }
            if (p < end) {
                const uint32_t lo = p < begin ? begin - p : 0u;
                const uint32_t hi = end - p >= 16u ? 16u : end - p;
                const uint32_t valid = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                m &= valid;
                if (z & valid) nul |= 0x80u;  // (the whole words keep their zero-byte flags in the bytes' high bits)
                if (own_end > p) own += (uint32_t)__builtin_popcount(own_end - p >= 16u ? m : m & ((1u << (own_end - p)) - 1u));
                if (last && end > begin && end - 1u >= p && end - 1u - p < 16u) {  // this word holds the last byte of the stream
                    if (!((m >> (end - 1u - p)) & 1u)) {
                        m |= 1u << hi;  // virtual '\n' at position end
                        st[kTsUnterminated] = 1u;
                    }
                }
            } else {
                m = 0;
            }
            mask[h + i] = m;
        }
    }
}

// text = slot[begin, end).  With `last`, a final line that lacks its '\n' is closed by a
// virtual one at position `end` (state word kTsUnterminated tells the record kernel).
// own_end (pieces of one stream framed by several contexts, hpn_fastq_text_piece_*): the '\n' at
// positions < own_end are also counted into kTsOwnLines -- what this piece adds to the stream's
// line count, and how many record starts it can own; 0: not a piece.
//
// A tile is 64 KiB, a wave's part of it 16 contiguous KiB: load k of a wave covers 1 KiB.  Newlines are numbered
// in (wave, load, lane, bit) order = by position.  Inside a wave: the counts of two loads share one DPP prefix sum
// (16 bits each: at most 17 per lane), eight of them before the look-back, so that what stays live across the
// look-back is 16 masks + 8 packed offsets (the first form of this kernel held 256 VGPRs: two waves per SIMD).
__global__ __launch_bounds__(kLinesThreads) void k_text_lines(const uint8_t *__restrict__ slot, uint32_t begin,
                                                            uint32_t end, int last, uint32_t own_end,
                                                            uint32_t *__restrict__ nl, uint32_t nl_cap,
                                                            u64 *__restrict__ status, uint32_t *__restrict__ st)
{
    __shared__ uint32_t s_w[kLinesThreads / kWave];
    __shared__ u64 s_excl;
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x;
    if (tid == 0) s_tile = atomicAdd(&st[kTsTicket1], 1u);  // tiles in start order (look-back never waits on a tile not yet running)
    __syncthreads();
    const uint32_t tile = s_tile;
    uint32_t nul = 0, own = 0;
    bool dense = false;
    const uint32_t lo_t = (begin & ~15u) + tile * kTxtTile;
    const u64 hi_t = (u64)lo_t + kTxtTile;
    const uint32_t base = lo_t + (uint32_t)wave_id() * kTxtWaveBytes + (uint32_t)lane_id() * 16u;
    const bool inner = lo_t >= begin && hi_t + (last ? 1u : 0u) <= end && (own_end == 0u || own_end >= hi_t || own_end <= lo_t);
    uint32_t mask[kTxtRows];
    if (inner) line_masks<false>(slot, base, begin, end, last, own_end, own_end >= hi_t, mask, nul, own, st);
    else line_masks<true>(slot, base, begin, end, last, own_end, false, mask, nul, own, st);

    uint32_t ex[kTxtRows / 2], rows_before = 0;  // ex: where the lane's newlines of two loads start inside the wave's
#pragma unroll
    for (int k = 0; k < kTxtRows; k += 2) {
        const uint32_t c0 = (uint32_t)__builtin_popcount(mask[k]), c1 = (uint32_t)__builtin_popcount(mask[k + 1]);
        const uint32_t inc = wave_incl_dpp(c0 | (c1 << 16));
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
        const uint32_t e0 = rows_before + (inc & 0xffffu) - c0;
        const uint32_t e1 = rows_before + (t & 0xffffu) + (inc >> 16) - c1;
        ex[k / 2] = e0 | (e1 << 16);
        rows_before += (t & 0xffffu) + (t >> 16);
    }
    if (lane_id() == 0) s_w[wave_id()] = rows_before;
    __syncthreads();
    uint32_t aggregate = 0, before = 0;
#pragma unroll
    for (int w = 0; w < kLinesThreads / kWave; ++w) {
        if (w == wave_id()) before = aggregate;
        aggregate += s_w[w];
    }
    if (wave_id() == 0) {
        const u64 e = scan_lookback(status, tile, aggregate, &st[kTsErr]);
        if (lane_id() == 0) s_excl = e;
    }
    __syncthreads();
    const u64 wave_base64 = s_excl + before;
    const uint32_t wave_base = wave_base64 < nl_cap ? (uint32_t)wave_base64 : nl_cap;  // (nl_cap + 17408 < 2^32: a chunk is < 2^31 bytes)
#pragma unroll
    for (int k = 0; k < kTxtRows; ++k) {
        const uint32_t p = base + (uint32_t)k * kTxtRowBytes;
        uint32_t g = wave_base + ((ex[k / 2] >> (16 * (k & 1))) & 0xffffu);
        uint32_t m = mask[k];
        while (m) {
            const int j = __builtin_ctz(m);
            m &= m - 1;
            if (g < nl_cap) nl[g] = p + (uint32_t)j;
            else dense = true;
            ++g;
        }
    }
    if (tile == gridDim.x - 1 && tid == 0) {
        const u64 n = s_excl + aggregate;
        st[kTsLines] = n > nl_cap ? nl_cap : (uint32_t)n;
    }
    if (nul & 0x80808080u) atomicOr(&st[kTsFlags], (uint32_t)HPN_TEXT_NUL);
    if (dense) atomicOr(&st[kTsFlags], (uint32_t)HPN_TEXT_DENSE);
    if (own_end) {   // (wave-uniform branch; one atomic per wave that saw any)
        const uint32_t wt = wave_sum(own);
        if (wt && lane_id() == 0) atomicAdd(&st[kTsOwnLines], wt);
    }
}

__device__ __forceinline__ uint32_t trim_cut(uint32_t l, uint32_t S, uint32_t E, uint32_t &b)
{
    b = S < l ? S : l;
    const uint32_t e = E < l ? E : l;
    return e > b ? e - b : 0u;
}

// How a chunk's lines group into records.  Chunk mode (first = -1, limit = 0): the text starts at a
// record, record r = lines 4r .. 4r+3, the lines of an unfinished last record are carried over.  Piece
// mode (limit != 0; hpn_fastq_text_piece_*): the text is a piece of a stream cut anywhere, handed over
// with the byte before the piece and a tail behind it; the stream has `L` lines before the text, so a
// record starts behind local line end i iff (L + i + 1) % 4 == 0 -- `first` is the smallest such i
// (-1: at `begin` itself, the stream's first byte) -- and the piece owns the records that START before
// `limit`; their lines reach into the tail.
struct TextFrame {
    int32_t first;    // local index of the line end in front of record 0; -1 = record 0 starts at begin
    uint32_t limit;   // piece mode: records starting at or beyond this position belong to the next piece; 0 = chunk mode
};

// Launched with an upper bound of tiles (the line count lives on the device).
template <bool kTrim>
__global__ __launch_bounds__(kTxtThreads) void k_text_records(const uint32_t *__restrict__ nl, uint32_t begin,
                                                              uint32_t end, int last, uint32_t S, uint32_t E,
                                                              uint32_t carry_cap, TextFrame fr,
                                                              uint64_t *__restrict__ off,
                                                              u64 *__restrict__ status, uint32_t *__restrict__ st)
{
    __shared__ u64 s_wave[kTxtThreads / kWave];
    __shared__ u64 s_excl;
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x;
    const uint32_t n_lines = st[kTsLines];
    const uint32_t unterminated = st[kTsUnterminated];
    const bool piece = fr.limit != 0;
    const uint32_t *__restrict__ nlp = nl + (fr.first + 1);   // record r's four line ends: nlp[4r .. 4r+3]; the one in front: nlp[4r-1]
    uint32_t n;
    bool incomplete = false;
    if (piece) {
        const uint32_t m = st[kTsOwnLines];   // line ends at positions < limit - 1: those a record of this piece can start behind
        if (fr.first < 0) n = begin < fr.limit ? 1u + m / 4u : 0u;
        else n = m > (uint32_t)fr.first ? (m - (uint32_t)fr.first + 3u) / 4u : 0u;
        // the last record's fourth line end is local line fr.first + 4n: it has to exist in the text handed over
        if (n && (uint32_t)(fr.first + 1) + 4u * n > n_lines) incomplete = true;
    } else {
        n = n_lines >> 2;
    }
    const uint32_t lines_used = (uint32_t)(fr.first + 1) + 4u * n;   // local lines up to the last record's end
    if (blockIdx.x == 0 && tid == 0) {
        uint32_t f = 0;
        if (piece) {
            if (incomplete) f |= last ? HPN_TEXT_PARTIAL : HPN_TEXT_LONG_LINE;   // (a tail of 4 KiB holds every regular record)
            st[kTsRecs] = incomplete ? 0u : n;
            st[kTsConsumed] = end;
        } else {
            st[kTsRecs] = n;
            uint32_t consumed = n ? nl[4u * n - 1u] + 1u : begin;
            if (consumed > end) consumed = end;  // the virtual newline
            st[kTsConsumed] = consumed;
            const uint32_t left = end - consumed;
            if (last && left) f |= HPN_TEXT_PARTIAL;
            if (!last && left > carry_cap) f |= HPN_TEXT_LONG_LINE;
        }
        if (kTrim && unterminated) f |= HPN_TEXT_PARTIAL;
        if (f) atomicOr(&st[kTsFlags], f);
        if (n == 0 || incomplete) {
            off[0] = 0;
            st[kTsTotalLo] = st[kTsTotalHi] = 0;
        }
    }
    if (incomplete) return;
    // The grid is an upper bound; only the workgroups that have records take a ticket (the
    // same-address atomic is the serial part of the kernel), in start order.
    if ((u64)blockIdx.x * kRecTile >= n) return;
    if (tid == 0) s_tile = atomicAdd(&st[kTsTicket2], 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * kRecTile + (uint32_t)tid * kRecPerThread;
    uint32_t val[kRecPerThread], flags = 0;
    uint32_t prev = begin - 1u;
    if (base < n && (base || fr.first >= 0)) prev = nlp[4 * (int)base - 1];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kRecPerThread; ++k) {
        const uint32_t r = base + k;
        val[k] = 0;
        if (r < n) {
            u32 e;  // the record's four line ends (16-byte aligned in chunk mode only)
            __builtin_memcpy(&e, nlp + 4u * r, 16);
            const uint32_t L1 = e[0] - prev, L2 = e[1] - e[0], L3 = e[2] - e[1];  // lengths with the '\n'
            uint32_t L4 = e[3] - e[2];
            if (L1 > 1023u || L2 > 1023u || L3 > 1023u || L4 > 1023u) flags |= HPN_TEXT_LONG_LINE;  // gzgets would split it
            const uint32_t len = L2 - 1u;
            if (kTrim) {
                if (L4 != L2) flags |= HPN_TEXT_RAGGED;
                if (S > len) flags |= HPN_TEXT_STALE;  // strncpy(buf + S) would run into the earlier lines' bytes
                uint32_t b;
                val[k] = L1 + 2u * trim_cut(len, S, E, b) + 4u;
            } else {
                if (unterminated && r == n - 1u && lines_used == n_lines) L4 -= 1u;  // no '\n' in the buffer after the last gzgets
                if (L4 < len) flags |= HPN_TEXT_RAGGED;  // the reference would tally stale buffer bytes
                if (len >= HPN_LEN_BINS) flags |= HPN_TEXT_LEN;
                val[k] = len;
            }
            prev = e[3];
        }
        mine += val[k];
    }
    if (flags) atomicOr(&st[kTsFlags], flags);
    u64 wtotal;
    const u64 wexcl = wave_excl_scan(mine, wtotal);
    if (lane_id() == kWave - 1) s_wave[wave_id()] = wtotal;
    __syncthreads();
    u64 before = 0, aggregate = 0;
#pragma unroll
    for (int w = 0; w < kTxtThreads / kWave; ++w) {
        if (w < wave_id()) before += s_wave[w];
        aggregate += s_wave[w];
    }
    if (wave_id() == 0) {
        const u64 ex = scan_lookback(status, tile, aggregate, &st[kTsErr]);
        if (lane_id() == 0) s_excl = ex;
    }
    __syncthreads();
    u64 run = s_excl + before + wexcl;
#pragma unroll
    for (int k = 0; k < kRecPerThread; ++k) {
        const uint32_t r = base + k;
        if (r < n) off[r] = run;
        run += val[k];
        if (r + 1u == n) {
            off[n] = run;
            st[kTsTotalLo] = (uint32_t)run;
            st[kTsTotalHi] = (uint32_t)(run >> 32);
        }
    }
}

// 16 lanes copy one span: 16-byte unaligned pieces, the last one overlapping its
// predecessor; spans shorter than 16 bytewise.
__device__ __forceinline__ void copy_span(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t cnt, int sub)
{
    if (cnt >= 16u) {
        for (uint32_t i = 16u * (uint32_t)sub; i < cnt; i += 256u) {
            const uint32_t o = min(i, cnt - 16u);
            u32 v;
            __builtin_memcpy(&v, src + o, 16);
            __builtin_memcpy(dst + o, &v, 16);
        }
    } else if ((uint32_t)sub < cnt) {
        dst[sub] = src[sub];
    }
}

__global__ __launch_bounds__(kTxtThreads) void k_text_gather(const uint8_t *__restrict__ slot,
                                                             const uint32_t *__restrict__ nl,
                                                             const uint64_t *__restrict__ off, uint32_t n,
                                                             uint8_t *__restrict__ out_qual,
                                                             uint8_t *__restrict__ out_seq)
{
    const uint32_t nwaves = gridDim.x * (kTxtThreads / kWave);
    const uint32_t wave = blockIdx.x * (kTxtThreads / kWave) + wave_id();
    const int lane = lane_id(), sub = lane & 15, g = lane >> 4;
    for (uint32_t r0 = wave * kWave; r0 < n; r0 += nwaves * kWave) {
        const uint32_t r = r0 + lane;
        uint32_t qs = 0, ss = 0, cnt = 0;
        uint64_t d = 0;
        if (r < n) {
            u32 e;
            __builtin_memcpy(&e, nl + 4u * r, 16);
            ss = e[0] + 1u, qs = e[2] + 1u, cnt = e[1] - e[0] - 1u;
            d = off[r];
        }
#pragma unroll 2
        for (int it = 0; it < kWave / 4; ++it) {  // four records per wave-instruction
            const int j = 4 * it + g;
            const uint32_t qj = __shfl(qs, j, kWave), sj = __shfl(ss, j, kWave), cj = __shfl(cnt, j, kWave);
            const uint64_t dj = __shfl(d, j, kWave);
            copy_span(slot + qj, out_qual + dj, cj, sub);
            if (out_seq) copy_span(slot + sj, out_seq + dj, cj, sub);
        }
    }
}

// nl = the first record's four line ends (k_text_records' nlp); at_begin: record 0 starts at `begin`, else behind nl[-1]
__global__ __launch_bounds__(kTxtThreads) void k_text_trim(const uint8_t *__restrict__ slot,
                                                           const uint32_t *__restrict__ nl, uint32_t begin, int at_begin,
                                                           const uint64_t *__restrict__ off, uint32_t n, uint32_t S,
                                                           uint32_t E, uint8_t *__restrict__ out)
{
    const uint32_t nwaves = gridDim.x * (kTxtThreads / kWave);
    const uint32_t wave = blockIdx.x * (kTxtThreads / kWave) + wave_id();
    const int lane = lane_id(), sub = lane & 15, g = lane >> 4;
    for (uint32_t r0 = wave * kWave; r0 < n; r0 += nwaves * kWave) {
        const uint32_t r = r0 + lane;
        uint32_t p0 = 0, L1 = 0, ss = 0, qs = 0, cut = 0;
        uint64_t d = 0;
        if (r < n) {
            u32 e;
            __builtin_memcpy(&e, nl + 4u * r, 16);
            p0 = r || !at_begin ? nl[4 * (int)r - 1] + 1u : begin;
            L1 = e[0] + 1u - p0;  // name line with its '\n'
            uint32_t b;
            cut = trim_cut(e[1] - e[0] - 1u, S, E, b);
            ss = e[0] + 1u + b, qs = e[2] + 1u + b;
            d = off[r];
        }
#pragma unroll 2
        for (int it = 0; it < kWave / 4; ++it) {
            const int j = 4 * it + g;
            const uint32_t pj = __shfl(p0, j, kWave), lj = __shfl(L1, j, kWave), sj = __shfl(ss, j, kWave);
            const uint32_t qj = __shfl(qs, j, kWave), cj = __shfl(cut, j, kWave);
            const uint64_t dj = __shfl(d, j, kWave);
            if (r0 + (uint32_t)j >= n) continue;
            uint8_t *o = out + dj;
            copy_span(slot + pj, o, lj, sub);               // "%s\n"  name
            copy_span(slot + sj, o + lj, cj, sub);          // "%s"    seq[S:min(E,len)]
            copy_span(slot + qj, o + lj + cj + 3u, cj, sub);  // "%s"    qual[S:min(E,len)]
            if (sub == 0) {
                o[lj + cj] = '\n', o[lj + cj + 1u] = '+', o[lj + cj + 2u] = '\n';  // "\n+\n"
                o[lj + 2u * cj + 3u] = '\n';
            }
        }
    }
}

uint64_t text_tiles1(uint32_t begin, uint32_t end)
{
    const uint64_t span = (uint64_t)end - (begin & ~15u);
    const uint64_t t = (span + kTxtTile - 1) / kTxtTile;
    return t ? t : 1;
}
uint64_t text_tiles2(uint32_t nl_cap) { return (uint64_t)(nl_cap / 4u) / kRecTile + 1; }

hipError_t launch_text_lines(const uint8_t *d_slot, uint32_t begin, uint32_t end, int last, uint32_t own_end, uint32_t *d_nl,
                             uint32_t nl_cap, u64 *d_status, uint32_t *d_state, hipStream_t st);
hipError_t launch_text_records(const uint32_t *d_nl, uint32_t begin, uint32_t end, int last, bool trim, uint32_t S, uint32_t E,
                               uint32_t carry_cap, int32_t first, uint32_t limit, uint32_t nl_cap, uint64_t *d_off,
                               u64 *d_status, uint32_t *d_state, hipStream_t st);

// frame: newline index + record validation / scan.  d_status holds text_tiles1 + text_tiles2 words.
hipError_t launch_text_frame(const uint8_t *d_slot, uint32_t begin, uint32_t end, int last, bool trim, uint32_t S,
                             uint32_t E, uint32_t carry_cap, uint32_t *d_nl, uint32_t nl_cap, uint64_t *d_off,
                             u64 *d_status, uint32_t *d_state, hipStream_t st)
{
    hipError_t e = launch_text_lines(d_slot, begin, end, last, 0u, d_nl, nl_cap, d_status, d_state, st);
    if (e != hipSuccess) return e;
    return launch_text_records(d_nl, begin, end, last, trim, S, E, carry_cap, -1, 0u, nl_cap, d_off, d_status, d_state, st);
}

// The two halves on their own (pieces: the line count of a piece is published before its records can be framed).
hipError_t launch_text_lines(const uint8_t *d_slot, uint32_t begin, uint32_t end, int last, uint32_t own_end, uint32_t *d_nl,
                             uint32_t nl_cap, u64 *d_status, uint32_t *d_state, hipStream_t st)
{
    const uint64_t t1 = text_tiles1(begin, end), t2 = text_tiles2(nl_cap);
    hipError_t e = hipMemsetAsync(d_status, 0, (t1 + t2) * sizeof(u64), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_state, 0, kTsWords * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_text_lines, dim3((unsigned)t1), dim3(kLinesThreads), 0, st, d_slot, begin, end, last, own_end, d_nl,
                       nl_cap, d_status, d_state);
    return hipGetLastError();
}

hipError_t launch_text_records(const uint32_t *d_nl, uint32_t begin, uint32_t end, int last, bool trim, uint32_t S, uint32_t E,
                               uint32_t carry_cap, int32_t first, uint32_t limit, uint32_t nl_cap, uint64_t *d_off,
                               u64 *d_status, uint32_t *d_state, hipStream_t st)
{
    const uint64_t t1 = text_tiles1(begin, end), t2 = text_tiles2(nl_cap);
    const TextFrame fr{first, limit};
    if (trim)
        hipLaunchKernelGGL(k_text_records<true>, dim3((unsigned)t2), dim3(kTxtThreads), 0, st, d_nl, begin, end, last, S, E,
                           carry_cap, fr, d_off, d_status + t1, d_state);
    else
        hipLaunchKernelGGL(k_text_records<false>, dim3((unsigned)t2), dim3(kTxtThreads), 0, st, d_nl, begin, end, last, S,
                           E, carry_cap, fr, d_off, d_status + t1, d_state);
    return hipGetLastError();
}

static unsigned copy_grid(uint32_t n, int n_cu)
{
    uint64_t want = ((uint64_t)n + kTxtThreads - 1) / kTxtThreads;
    const uint64_t cap = (uint64_t)n_cu * 8;
    return (unsigned)(want < cap ? want : cap);
}

hipError_t launch_text_gather(const uint8_t *d_slot, const uint32_t *d_nl, const uint64_t *d_off, uint32_t n,
                              uint8_t *d_out_qual, uint8_t *d_out_seq, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_text_gather, dim3(copy_grid(n, n_cu)), dim3(kTxtThreads), 0, st, d_slot, d_nl, d_off, n,
                       d_out_qual, d_out_seq);
    return hipGetLastError();
}

hipError_t launch_text_trim(const uint8_t *d_slot, const uint32_t *d_nl, uint32_t begin, int at_begin, const uint64_t *d_off,
                            uint32_t n, uint32_t S, uint32_t E, uint8_t *d_out, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_text_trim, dim3(copy_grid(n, n_cu)), dim3(kTxtThreads), 0, st, d_slot, d_nl, begin, at_begin, d_off, n,
                       S, E, d_out);
    return hipGetLastError();
}

}  // namespace hpn
