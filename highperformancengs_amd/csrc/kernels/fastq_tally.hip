// fastq_tally.hip -- K1L `k_tally_hist`: the full per-(symbol, cycle) histograms.
//
// Replaces AssignQuality (reference fastq_count.c:29-35) when the whole
// Quality[128][512] matrix is wanted (`fastq_count_kthread -L`, printQ :52-64) and
// the per-cycle nucleotide tally of Rgzfastq_uniq.c:50-57 (Nucleotide[5][512]).
// The default report needs neither: that is K1, fastq_scan.hip.
//
// One 1024-thread workgroup per CU owns a histogram image in LDS: 16-bit counters
// packed two per dword (row stride 257 dwords: bank = (sym + cycle/2) % 32), flushed
// with 64-bit global atomics before any counter can wrap (a counter gets at most
// one hit per read -> every <= 65,535 reads).  Records are taken in chunks of 1024
// whose boundaries sit in LDS.
//
//   chunk of equal-length reads (the normal case, len >= 16):
//     work item = (read r, vector v): bytes [16v, 16v+16) OF THE READ, fetched with
//     one unaligned 16-byte load.  The cycle of byte k is 16v+k: even base, no wrap
//     into the next read, so every lane of a wave runs the same 16 straight-line
//     ds_add_u32 (2 VALU + 1 LDS per byte, half-word selected at compile time).
//     A lane's items are (tid + 1024 m); (r, v) advance by a per-chunk constant.
//     Loads run one round of four vectors ahead of the tally.  sum / Q20 / Q30 and
//     the domain check are SWAR on the vector.  The len%16 tail bytes of each read
//     are walked by one lane per read.
//   anything else (ragged lengths, reads shorter than 16):
//     aligned vectors of the chunk's byte range, binary search over the LDS
//     boundaries per vector, per-byte walk.
//
// PMC history (profiles/r01b): memory-aligned vectors made lanes diverge on cycle
// parity and on read boundaries, which doubled the LDS and VALU instruction
// counts (2.05 LDS and 16-30 VALU wave-instructions per 64 bytes).
// LDS bound on MI355X (scripts/lds_atomic_ubench.hip): ds_add_u32 runs at 16
// lane-ops/cycle/CU conflict-free and ~10.5 for this address pattern, i.e. ~5.9e12
// tallied bytes/s for the chip.  No MFMA: there is no contraction here.
#include "tally_util.hpp"

namespace hpn {

constexpr int kHistThreads = 1024;
constexpr int kHistRecs = 4096;                  // records per chunk: 600 KB at 150 bp between barriers
constexpr int kRound = 4;                        // vectors per lane per round
constexpr int kRowWords = HPN_LEN_BINS / 2 + 1;  // 256 dwords of packed u16 pairs + 1 pad
constexpr uint32_t kFlushReads = 65535;          // a counter gets at most one hit per read

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

struct HistLds {
    uint32_t qh[HPN_QUAL_ROWS * kRowWords];
    uint32_t nh[HPN_NUC_CODES * kRowWords];
    uint32_t loff[kHistRecs + 1];
    uint32_t lhist[HPN_LEN_BINS + 1];
    uint32_t red[3][kHistThreads / kWave];
};

template <bool kQual>
__device__ __forceinline__ uint32_t hist_row(uint32_t byte)
{
    return (kQual ? byte : nuc_code(byte)) * kRowWords;
}

__device__ __forceinline__ uint32_t byte_of(const u32 &v, int k) { return (v[k >> 2] >> (8 * (k & 3))) & 0xffu; }

// One symbol at one cycle.
template <bool kQual>
__device__ __forceinline__ void bump(uint32_t *hist, uint32_t byte, uint32_t pos)
{
    atomicAdd(&hist[hist_row<kQual>(byte) + (pos >> 1)], 1u << ((pos & 1) << 4));
}

// 16 bytes of one read starting at the EVEN cycle 2*half.
template <bool kQual>
__device__ __forceinline__ void add16(uint32_t *hist, const u32 &v, uint32_t half)
{
#pragma unroll
    for (int k = 0; k < 16; ++k)
        atomicAdd(&hist[hist_row<kQual>(byte_of(v, k)) + half + (k >> 1)], (k & 1) ? 0x10000u : 1u);
}

struct Tot {
    uint32_t c20, c30, bad;
};

// sum / Q20 / Q30 + domain check of bytes [k0,k1) of a quality vector; false = skip it
template <bool kQual>
__device__ __forceinline__ bool vec_totals(const u32 &v, int k0, int k1, Tot &t)
{
    if (!kQual) return true;
    uint32_t hb = 0;
    swar16((k0 == 0 && k1 == 16) ? v : mask_bytes(v, k0, k1), t.c20, t.c30, hb);
    if (hb & 0x80808080u) {  // quality byte >= 128: batch rejected; keep LDS indices in range
        t.bad = 1;
        return false;
    }
    return true;
}

// 16 bytes from any address (hardware unaligned access; the compiler emits one
// global_load_dwordx4 for the align-1 copy).
__device__ __forceinline__ u32 load_unaligned16(const uint8_t *p)
{
    u32 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// Equal-length chunk (len0 >= 16): cnt reads of len0 bytes starting at arr + base_off.
template <bool kQual>
__device__ __forceinline__ void stream_uniform(HistLds &s, const uint8_t *arr, uint64_t base_off, uint32_t cnt,
                                               uint32_t len0, Tot &t)
{
    uint32_t *hist = kQual ? s.qh : s.nh;
    const uint8_t *p0 = arr + base_off;
    const uint32_t nvr = len0 >> 4;               // whole vectors per read
    const uint32_t items = cnt * nvr;
    // item w = r * nvr + v; a lane's items are w0 + 1024 m: (r, v) advance by (dr, dv) with carry
    const uint32_t dr = kHistThreads / nvr, dv = kHistThreads - dr * nvr;
    uint32_t w = threadIdx.x;
    uint32_t r = w / nvr, v = w - r * nvr;        // the one division per chunk
    auto step = [&]() {
        w += kHistThreads;
        r += dr, v += dv;
        if (v >= nvr) v -= nvr, ++r;
    };
    auto addr = [&](uint32_t rr, uint32_t vv) { return p0 + (size_t)rr * len0 + 16u * vv; };

    u32 va[kRound], vb[kRound];
    uint32_t ha[kRound], hb[kRound];  // cycle/2 of byte 0, or ~0u = no item
    auto fetch = [&](u32 (&vec)[kRound], uint32_t (&half)[kRound]) {
#pragma unroll
        for (int m = 0; m < kRound; ++m) {
            // no branch around the load (an idle lane re-reads the chunk's first bytes): the
            // compiler then counts the loads in flight (vmcnt(4)) instead of draining them all
            const bool on = w < items;
            vec[m] = load_unaligned16(on ? addr(r, v) : p0);
            half[m] = on ? 8u * v : ~0u;
            step();
        }
    };
    auto tally = [&](u32 (&vec)[kRound], uint32_t (&half)[kRound]) {
#pragma unroll
        for (int m = 0; m < kRound; ++m)
            if (half[m] != ~0u && vec_totals<kQual>(vec[m], 0, 16, t)) add16<kQual>(hist, vec[m], half[m]);
    };
    if (items) {
        fetch(va, ha);
        for (;;) {
            const bool more_b = w < items;  // w now points at round B's first item of this lane
            fetch(vb, hb);
            tally(va, ha);
            if (!more_b) break;
            const bool more_a = w < items;
            fetch(va, ha);
            tally(vb, hb);
            if (!more_a) break;
        }
    }
    // tails: the last len0 % 16 bytes of every read, one lane per read
    const uint32_t rem = len0 & 15u;
    if (rem) {
#pragma unroll 1
        for (uint32_t rr = threadIdx.x; rr < cnt; rr += kHistThreads) {
            const uint8_t *q = addr(rr, nvr);
#pragma unroll 1
            for (uint32_t k = 0; k < rem; ++k) {
                const uint32_t byte = q[k];
                if (kQual) {
                    if (byte >= HPN_QUAL_ROWS) {
                        t.bad = 1;
                        continue;
                    }
                    t.c20 += byte >= 53, t.c30 += byte >= 63;
                }
                bump<kQual>(hist, byte, 16u * nvr + k);
            }
        }
    }
}

// Ragged chunk: aligned vectors of the byte range, each located by binary search.
// Records are [loff[lo-1], loff[lo]).
template <bool kQual>
__device__ __forceinline__ void stream_ragged(HistLds &s, const uint8_t *arr, uint64_t base_off, uint32_t cnt, Tot &t)
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
        if (!vec_totals<kQual>(v, k0, k1, t)) continue;
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
            bump<kQual>(hist, (wd >> (8 * (k & 3))) & 0xffu, pos);
            ++b, ++pos;
        }
    }
}

__device__ __forceinline__ void hist_flush(uint32_t *lds, int rows, u64 *__restrict__ gacc)
{
    for (int w = threadIdx.x; w < rows * kRowWords; w += kHistThreads) {
        const uint32_t v = lds[w];
        if (v) {
            const int r = w / kRowWords, c = w - r * kRowWords;
            if (v & 0xffffu) atomicAdd(&gacc[r * HPN_LEN_BINS + 2 * c], (u64)(v & 0xffffu));
            if (v >> 16) atomicAdd(&gacc[r * HPN_LEN_BINS + 2 * c + 1], (u64)(v >> 16));
            lds[w] = 0;
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
    const int tid = threadIdx.x;
    for (int i = tid; i < HPN_QUAL_ROWS * kRowWords; i += kHistThreads) s.qh[i] = 0;
    for (int i = tid; i < HPN_NUC_CODES * kRowWords; i += kHistThreads) s.nh[i] = 0;
    for (int i = tid; i <= HPN_LEN_BINS; i += kHistThreads) s.lhist[i] = 0;
    __syncthreads();

    Tot t{0, 0, 0};
    uint32_t since_flush = 0;
    u64 bytes = 0;
    const uint64_t nchunk = (n + kHistRecs - 1) / kHistRecs;
    for (uint64_t ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const uint64_t r0 = ch * kHistRecs;
        const uint32_t cnt = (uint32_t)min((uint64_t)kHistRecs, n - r0);
        if (since_flush + cnt > kFlushReads) {
            __syncthreads();
            if (kQualHist) hist_flush(s.qh, HPN_QUAL_ROWS, acc + HPN_TALLY_W_QUAL);
            if (kNucHist) hist_flush(s.nh, HPN_NUC_CODES, acc + HPN_TALLY_W_NUC);
            since_flush = 0;
        }
        since_flush += cnt;
        const uint64_t base_off = off[r0];
        // chunk-relative boundaries; a chunk spans < 1024*512 bytes inside the domain
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
                if (len >= HPN_LEN_BINS) len = HPN_LEN_BINS, t.bad = 1;
                all_same = all_same && len == len0;
            }
            hist_len(s.lhist, valid, len);
        }
        // an over-long record poisons position tracking: stop tallying bytes, the batch
        // is rejected as a whole (HPN_E_DOMAIN) once `bad` is seen
        if (!__syncthreads_or((int)t.bad)) {
            const bool uniform = __syncthreads_and((int)all_same) && len0 >= 16;
            if (tid == 0) bytes += s.loff[cnt];
            if (uniform) {
                if (kQualHist) stream_uniform<true>(s, qual, base_off, cnt, len0, t);
                if (kNucHist) stream_uniform<false>(s, base, base_off, cnt, len0, t);
            } else {
                if (kQualHist) stream_ragged<true>(s, qual, base_off, cnt, t);
                if (kNucHist) stream_ragged<false>(s, base, base_off, cnt, t);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (kQualHist) hist_flush(s.qh, HPN_QUAL_ROWS, acc + HPN_TALLY_W_QUAL);
    if (kNucHist) hist_flush(s.nh, HPN_NUC_CODES, acc + HPN_TALLY_W_NUC);

    const uint32_t c20 = wave_sum(t.c20), c30 = wave_sum(t.c30), bad = wave_or(t.bad);
    if (lane_id() == 0) {
        s.red[0][wave_id()] = c20;
        s.red[1][wave_id()] = c30;
        s.red[2][wave_id()] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        u64 s20 = 0, s30 = 0;
        uint32_t h = 0;
        for (int w = 0; w < kHistThreads / kWave; ++w) s20 += s.red[0][w], s30 += s.red[1][w], h |= s.red[2][w];
        if (kQualHist) {
            if (s20) atomicAdd(&acc[HPN_TALLY_W_Q20], s20);
            if (s30) atomicAdd(&acc[HPN_TALLY_W_Q30], s30);
        }
        if (h) atomicAdd(&acc[HPN_TALLY_W_BAD], (u64)1);
        if (kQualHist && bytes) atomicAdd(&acc[HPN_TALLY_W_TOTAL], bytes);
    }
    // SeqLen / sum come from this kernel only when it replaces the flat scan
    // (quality matrix requested); a nucleotide-only launch runs beside k_tally_scan.
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
