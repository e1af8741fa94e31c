// fastq_tally.hip -- K1L `k_tally_hist`: the full per-(symbol, cycle) histograms.
//
// Replaces AssignQuality (reference fastq_count.c:29-35) when the whole
// Quality[128][512] matrix is wanted (`fastq_count_kthread -L`, printQ :52-64) and
// the per-cycle nucleotide tally of Rgzfastq_uniq.c:50-57 (Nucleotide[5][512]).
// The default report needs neither: that is K1, fastq_scan.hip.
//
// One 1024-thread workgroup per CU owns a histogram image in LDS: 16-bit
// counters packed two per dword, flushed with 64-bit global atomics before any
// counter can wrap (a counter gets at most one hit per read).  Records are taken
// in chunks of 1024 whose boundaries sit in LDS; each lane takes one 16-byte
// vector of the chunk, finds its record by binary search and walks 16 bytes.
// Bound: LDS atomic rate.  No MFMA: there is no contraction here.
#include "tally_util.hpp"

namespace hpn {

constexpr int kHistThreads = 1024;
constexpr int kHistRecs = 1024;                  // records per chunk
constexpr int kRowWords = HPN_LEN_BINS / 2 + 1;  // 256 dwords of packed u16 pairs + 1 pad:
                                                 // bank = (sym + pos/2) % 32, so lanes at one
                                                 // cycle but different symbols spread over banks
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
__device__ __forceinline__ void hist_byte(HistLds &s, uint32_t byte, uint32_t pos, uint32_t &c20, uint32_t &c30,
                                          uint32_t &bad)
{
    if (kQual) {
        if (byte >= HPN_QUAL_ROWS) {
            bad = 1;
            return;
        }
        c20 += byte >= 53;
        c30 += byte >= 63;
        atomicAdd(&s.qh[byte * kRowWords + (pos >> 1)], 1u << ((pos & 1) << 4));
    } else {
        atomicAdd(&s.nh[nuc_code(byte) * kRowWords + (pos >> 1)], 1u << ((pos & 1) << 4));
    }
}

// One pass over the chunk's bytes of one array (quality or bases).
template <bool kQual>
__device__ __forceinline__ void hist_stream(HistLds &s, const uint8_t *arr, uint64_t base_off, uint32_t cnt,
                                            uint32_t &c20, uint32_t &c30, uint32_t &bad)
{
    const uint32_t B = s.loff[cnt];
    if (B == 0) return;
    const uint8_t *p0 = arr + base_off;
    const int a0 = (int)((uintptr_t)p0 & 15);
    const u32 *vec = reinterpret_cast<const u32 *>(p0 - a0);
    const uint32_t nvec = (uint32_t)((a0 + B + 15) >> 4);
    for (uint32_t j = threadIdx.x; j < nvec; j += kHistThreads) {
        const u32 v = load_stream16(vec + j);
        const int rel = (int)(16 * j) - a0;  // chunk-relative index of the vector's byte 0
        const int k0 = rel < 0 ? -rel : 0;
        const int k1 = min(16, (int)B - rel);
        uint32_t b = (uint32_t)(rel + k0);
        // first record boundary above b: records are [loff[i-1], loff[i])
        uint32_t lo = 1, hi = cnt;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s.loff[mid] > b) hi = mid;
            else lo = mid + 1;
        }
        uint32_t nxt = s.loff[lo];
        uint32_t pos = b - s.loff[lo - 1];
        if (k0 == 0 && k1 == 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (b == nxt) {  // crossed into the next non-empty record
                    do { ++lo; } while (s.loff[lo] <= b);
                    pos = b - s.loff[lo - 1];
                    nxt = s.loff[lo];
                }
                hist_byte<kQual>(s, (v[k >> 2] >> (8 * (k & 3))) & 0xffu, pos, c20, c30, bad);
                ++b;
                ++pos;
            }
        } else {
            for (int k = k0; k < k1; ++k) {
                if (b == nxt) {
                    do { ++lo; } while (s.loff[lo] <= b);
                    pos = b - s.loff[lo - 1];
                    nxt = s.loff[lo];
                }
                const uint32_t w = k < 4 ? v[0] : k < 8 ? v[1] : k < 12 ? v[2] : v[3];
                hist_byte<kQual>(s, (w >> (8 * (k & 3))) & 0xffu, pos, c20, c30, bad);
                ++b;
                ++pos;
            }
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

    uint32_t c20 = 0, c30 = 0, bad = 0, since_flush = 0;
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
        uint32_t len = 0;
        const bool valid = (uint32_t)tid < cnt;
        if (valid) {
            len = s.loff[tid + 1] - s.loff[tid];
            if (len >= HPN_LEN_BINS) len = HPN_LEN_BINS, bad = 1;
        }
        hist_len(s.lhist, valid, len);
        // an over-long record poisons position tracking: stop tallying bytes, the batch
        // is rejected as a whole (HPN_E_DOMAIN) once `bad` is seen
        if (!__syncthreads_or((int)bad)) {
            if (tid == 0) bytes += s.loff[cnt];
            if (kQualHist) hist_stream<true>(s, qual, base_off, cnt, c20, c30, bad);
            if (kNucHist) hist_stream<false>(s, base, base_off, cnt, c20, c30, bad);
        }
        __syncthreads();
    }
    __syncthreads();
    if (kQualHist) hist_flush(s.qh, HPN_QUAL_ROWS, acc + HPN_TALLY_W_QUAL);
    if (kNucHist) hist_flush(s.nh, HPN_NUC_CODES, acc + HPN_TALLY_W_NUC);

    c20 = wave_sum(c20);
    c30 = wave_sum(c30);
    bad = wave_or(bad);
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
    const unsigned grid = (unsigned)(nchunk < (uint64_t)n_cu ? nchunk : (uint64_t)n_cu);  // LDS: one per CU
    if (qual_hist && nuc_hist)
        hipLaunchKernelGGL((k_tally_hist<true, true>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc);
    else if (qual_hist)
        hipLaunchKernelGGL((k_tally_hist<true, false>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc);
    else
        hipLaunchKernelGGL((k_tally_hist<false, true>), dim3(grid), dim3(kHistThreads), 0, st, d_qual, d_base, d_off, n, d_acc);
    return hipGetLastError();
}

}  // namespace hpn
