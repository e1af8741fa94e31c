// fastq_tally.hip -- K1L `k_tally_hist`: the full per-(symbol, cycle) histograms.
//
// Replaces AssignQuality (reference fastq_count.c:29-35) when the whole
// Quality[128][512] matrix is wanted (`fastq_count_kthread -L`, printQ :52-64) and
// the per-cycle nucleotide tally of Rgzfastq_uniq.c:50-57 (Nucleotide[5][512]).
// The default report needs neither: that is K1, fastq_scan.hip.
//
// One 1024-thread workgroup per CU owns a histogram image in LDS: 32-bit counters,
// 128 symbol rows x cycles 0..255, 256 dwords per row, cycles stored transposed in
// groups of four:  word(row, cycle) = row*256 + (cycle & 3)*64 + (cycle >> 2).
// Cycles 256..511 (reads longer than 256 bases) go to the global matrix directly.
// The image is flushed once, at the end of the kernel (a 32-bit counter cannot wrap
// within one launch); sum / Q20 / Q30 are row sums taken during that flush.
//
//   chunk of equal-length reads, 16 <= len <= 256 (the normal case):
//     work item = (read r, group j) = cycles 4j..4j+3 OF THE READ, one unaligned dword
//     load.  Byte k of the item goes to word row*256 + k*64 + j, so the 32 lanes of a
//     half-wave (consecutive j) hit 32 different banks for every k, whatever the symbols
//     are: conflict-free ds_add_u32 (16 lane-ops/clk/CU measured, against ~10.5 for
//     16-bytes-per-lane vectors and ~12.6 for random words, scripts/lds_atomic_ubench.hip),
//     2 VALU + 1 LDS per byte, no per-byte compare: bit 7 of every byte is OR-ed into one
//     flag and the row index is masked.  A lane's items are tid + 1024 m; (r, j) advance by
//     a per-chunk constant with carry (one division per chunk).  Eight dwords per lane are
//     in flight, one round ahead of the round being tallied, behind counted vmcnt waits.
//     The len%4 tail bytes of each read are walked by one lane per read.
//   anything else (ragged lengths, very short or very long reads):
//     aligned 16-byte vectors of the chunk's byte range, binary search over the LDS
//     boundaries per vector, per-byte walk.
//
// PMC history (profiles/r01b), 2e8 x 150 bp: 22.5 ms aligned 16-B vectors + binary search;
// 14.9 -> lanes diverged on cycle parity / read boundaries (2x LDS and VALU instruction
// counts); 12.1 read-aligned 16-B vectors, LDS at the random-scatter conflict rate (70 % of
// LDS cycles were conflict cycles); 11.1 byte loads, conflict-free but too few bytes in
// flight per request slot and 6 SALU + 11 VALU per byte; 10.1 predicate-free body;
// 8.8 ms this form.  No MFMA: there is no contraction here.
#include "tally_util.hpp"

namespace hpn {

constexpr int kHistThreads = 1024;
constexpr int kHistWaves = kHistThreads / kWave;
constexpr int kHistRecs = 4096;    // records per chunk: 600 KB at 150 bp between barriers (2048 / 1024: +6 % / +12 %)
// dword items a lane keeps in flight: 4 / 8 / 16 -> 9.05 / 8.22 / 7.79 ms per 2e8 x 150 bp (128 VGPRs at 16; 20 spills; the kernel that
// builds both matrices spills at 16)
constexpr int kSpanOne = 16, kSpanBoth = 12;   // (both matrices: 8 / 10 / 12 / 14 -> 17.8 / 19.1 / 17.4 / 18.7 ms)
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
    uint32_t loff[kHistRecs + 1];
    uint32_t lhist[HPN_LEN_BINS + 1];
    u64 red[3][kHistWaves];
};

// Index of (row, cycle) inside the LDS image.  Cycles are stored TRANSPOSED in groups of four:
//     word(row, cycle) = row * 256 + (cycle & 3) * 64 + (cycle >> 2)
// A lane that holds the four bytes of cycles 4j..4j+3 of a read adds byte k at word
// row*256 + k*64 + j: the 32 lanes of a half-wave (consecutive j) hit 32 different banks for
// every k, whatever the symbols are.
__device__ __forceinline__ uint32_t lds_word(uint32_t row, uint32_t pos)
{
    return row * kRowWords + ((pos & 3u) << 6) + (pos >> 2);
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
template <bool kQual, bool kPartial, int kSpanRound>   // kPartial: len0 % 4 != 0 (the chunk's reads end in a partial group); kSpanRound: dword items a lane keeps in flight
__device__ __forceinline__ void stream_uniform(HistLds &s, const uint8_t *arr, uint64_t base_off, uint64_t arr_end, uint32_t cnt,
                                               uint32_t len0, uint32_t &bad)
{
    uint32_t *hist = kQual ? s.qh : s.nh;
    const uint8_t *p0 = arr + base_off;
    // 4-cycle groups per read; the last one holds only len0 % 4 cycles when that is not 0 (its load reaches into the next
    // read's first bytes, which are masked: a lane per read walking those tail bytes one by one put 4096 x 2 byte loads
    // and as many 32-way bank conflicts into every chunk -- 22 % of the kernel at 150 bp)
    const uint32_t ngr = (len0 + 3u) >> 2;
    const uint32_t rpr = kHistThreads / ngr;              // reads per round (>= 16: ngr <= 64)
    const uint32_t lr = threadIdx.x / ngr, j = threadIdx.x - lr * ngr;   // the one division per chunk
    const uint32_t nvalid = min(4u, len0 - 4u * j);       // bytes of this lane's group that belong to the read
    const bool lane_on = lr < rpr;                        // the last 1024 - rpr * ngr lanes have no item
    const uint32_t step = rpr * len0;                     // bytes between a lane's items of consecutive rounds
    uint32_t off = lr * len0 + 4u * j;                    // chunk-relative byte offset of this lane's item
    uint32_t r = lr;                                      // ... and its read
    uint32_t col[4];                                      // byte k of every item of this lane goes to word row*256 + col[k]
    uint32_t rmask[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool mine = (uint32_t)k < nvalid;
        rmask[k] = mine ? ~0u : 0u;
        col[k] = mine ? 64u * k + j : (kQual ? kJunkQ : kJunkN) + (threadIdx.x & 63u);
    }
    uint32_t va[kSpanRound], vb[kSpanRound];
    uint32_t na = 0, nb = 0;                              // items of the set that exist (wave-varying only in the last round)
    auto fetch = [&](uint32_t (&v)[kSpanRound], uint32_t &nv) {
        nv = 0;
#pragma unroll
        for (int m = 0; m < kSpanRound; ++m) {
            // no branch around the load (a lane without an item re-reads the chunk's first bytes): the
            // compiler then counts the loads in flight instead of draining them all
            const bool on = lane_on && r < cnt;
            // the partial group of the batch's very last read: the dword is taken `back` bytes earlier, so that it ends
            // with the batch, and shifted down (no branch: the loads of a set stay together)
            const uint32_t back = kPartial && base_off + off + 4u > arr_end ? 4u - nvalid : 0u;
            v[m] = load_unaligned4(p0 + (on ? off - back : 0u)) >> (8u * back);
            nv += on;
            off += step, r += rpr;
        }
    };
    // A quality byte >= 128 has no row: bit 7 of every byte is OR-ed into `seen` (checked once
    // per chunk; the batch is then rejected) and the row index is masked to 7 bits so that the
    // LDS address stays in range: no compare, no exec-mask juggling per byte.
    uint32_t seen = 0;
    auto tally = [&](const uint32_t (&v)[kSpanRound], uint32_t nv) {
#pragma unroll
        for (int m = 0; m < kSpanRound; ++m) {
            if ((uint32_t)m >= nv) break;                 // a lane's items of a set are its first nv
            const uint32_t d = v[m];
            if (kQual) seen |= d;   // (the bytes behind a partial group are the next read's: quality bytes as well)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t byte = (d >> (8 * k)) & (kQual ? 0x7fu : 0xffu);
                const uint32_t row = kQual ? byte : (uint32_t)s.nlut[byte];
                // bytes behind a partial group go to this lane's junk word behind the image: no branch per byte
                atomicAdd(&hist[(kPartial ? row & rmask[k] : row) * kRowWords + col[k]], 1u);
            }
        }
    };
    const uint32_t rounds = (cnt + rpr - 1) / rpr;        // uniform: every lane runs the same number of fetches
    for (uint32_t q = 0; q < rounds; q += 2 * kSpanRound) {
        if (q == 0) fetch(va, na);
        const bool more_b = q + kSpanRound < rounds;
        if (more_b) fetch(vb, nb);
        tally(va, na);
        if (!more_b) break;
        const bool more_a = q + 2 * kSpanRound < rounds;
        if (more_a) fetch(va, na);
        tally(vb, nb);
        if (!more_a) break;
    }
    if (seen & 0x80808080u) bad = 1;
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
            const int r = w / kRowWords, c = w - r * kRowWords;  // c = (cycle & 3) * 64 + (cycle >> 2)
            atomicAdd(&gacc[r * HPN_LEN_BINS + 4 * (c & 63) + (c >> 6)], (u64)v);
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
                                                            u64 *__restrict__ acc)
{
    __shared__ HistLds s;
    constexpr int kSp = (kQualHist && kNucHist) ? kSpanBoth : kSpanOne;
    const int tid = threadIdx.x;
    for (int i = tid; i < HPN_QUAL_ROWS * kRowWords; i += kHistThreads) s.qh[i] = 0;
    for (int i = tid; i < HPN_NUC_CODES * kRowWords; i += kHistThreads) s.nh[i] = 0;
    for (int i = tid; i <= HPN_LEN_BINS; i += kHistThreads) s.lhist[i] = 0;
    if (tid < 256) s.nlut[tid] = (uint8_t)nuc_code((uint32_t)tid);
    __syncthreads();

    uint32_t bad = 0;
    u64 *gq = acc + HPN_TALLY_W_QUAL, *gn = acc + HPN_TALLY_W_NUC;
    HiTot hi;
    const uint64_t arr_end = off[n];          // first byte offset that is not the batch's
    const uint64_t nchunk = (n + kHistRecs - 1) / kHistRecs;
    for (uint64_t ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const uint64_t r0 = ch * kHistRecs;
        const uint32_t cnt = (uint32_t)min((uint64_t)kHistRecs, n - r0);
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
                if (kQualHist) (len0 & 3u) ? stream_uniform<true, true, kSp>(s, qual, base_off, arr_end, cnt, len0, bad) : stream_uniform<true, false, kSp>(s, qual, base_off, arr_end, cnt, len0, bad);
                if (kNucHist) (len0 & 3u) ? stream_uniform<false, true, kSp>(s, base, base_off, arr_end, cnt, len0, bad) : stream_uniform<false, false, kSp>(s, base, base_off, arr_end, cnt, len0, bad);
            } else {
                if (kQualHist) stream_ragged<true>(s, gq, qual, base_off, cnt, bad, hi);
                if (kNucHist) stream_ragged<false>(s, gn, base, base_off, cnt, bad, hi);
            }
        }
        __syncthreads();
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
    if (qual_hist && nuc_hist)
        hipLaunchKernelGGL((k_tally_hist<true, true>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc);
    else if (qual_hist)
        hipLaunchKernelGGL((k_tally_hist<true, false>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc);
    else
        hipLaunchKernelGGL((k_tally_hist<false, true>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc);
    return hipGetLastError();
}

}  // namespace hpn
