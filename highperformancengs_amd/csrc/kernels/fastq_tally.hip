// fastq_tally.hip -- gfx950 kernels behind hpn_fastq_tally (include/hpngs.h).
//
// Replaces the scan loop of count_read (reference fastq_count.c:112-119,
// fastq_count_kthread.c:126-135; AssignQuality :29-35; statQ :37-47).
//
// Input in HBM (structure of arrays): qual[] = the quality bytes of all records
// back to back, off[n+1] = uint64 record boundaries.  Two kernels:
//
//   k_tally_scan  (K1)  what fastq_count prints: SeqLen[512], sum, sum>=53, sum>=63.
//                       Quality[q][pos] is only ever reduced over pos and over
//                       q>=53 / q>=63, so the byte stream is scanned flat with
//                       16-byte loads and SWAR compares; positions are not needed.
//                       Bound: HBM read, 1 B per base + 8 B per record.
//   k_tally_hist  (K1L) the full Quality[128][512] (and Nucleotide[5][512])
//                       matrix of `fastq_count_kthread -L`: per-workgroup
//                       histogram in LDS (16-bit counters packed two per dword,
//                       flushed before they can wrap), flushed with 64-bit
//                       global atomics.  Bound: LDS atomic rate.
//
// No MFMA: there is no contraction here.
#include "common.hpp"

namespace hpn {

// ---------------------------------------------------------------------------
// K1: flat scan
// ---------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanUnroll = 8;  // 16-byte loads in flight per thread
constexpr int kScanTileVec = kScanThreads * kScanUnroll;  // vectors per block-iteration (32 KiB)
constexpr int kLenPerThread = 4;
constexpr int kLenTile = kScanThreads * kLenPerThread;

// Bytes are < 128 inside the domain, so x+75 sets bit 7 exactly when x >= 53
// and x+65 exactly when x >= 63, with no carry between bytes.  `hi` collects
// bit 7 of every input byte: non-zero there = domain violation.
__device__ __forceinline__ void swar16(u32 v, uint32_t &c20, uint32_t &c30, uint32_t &hi)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t w = v[k];
        hi |= w;
        c20 += __builtin_popcount((w + 0x4b4b4b4bu) & 0x80808080u);
        c30 += __builtin_popcount((w + 0x41414141u) & 0x80808080u);
    }
}

// Keep bytes [a, b) of a 16-byte vector, zero the rest (a zero byte counts nowhere).
__device__ __forceinline__ u32 mask_bytes(u32 v, int a, int b)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int lo = min(max(a - 4 * k, 0), 4), hi = min(max(b - 4 * k, 0), 4);
        uint32_t mh = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
        uint32_t ml = lo >= 4 ? 0xffffffffu : ((1u << (8 * lo)) - 1u);
        v[k] &= mh & ~ml;
    }
    return v;
}

// Add one length per active lane to the LDS histogram.  Reads of one run have
// one length almost always: then a single lane adds the whole wave's count
// instead of 64 lanes serialising on one LDS address.
__device__ __forceinline__ void hist_len(uint32_t *s_hist, bool valid, uint32_t len)
{
    const u64 act = __ballot(valid);
    if (act == 0) return;
    const int leader = __builtin_ctzll(act);
    const uint32_t first = __shfl(len, leader, kWave);
    const u64 same = __ballot(valid && len == first);
    if (same == act) {
        if (lane_id() == leader) atomicAdd(&s_hist[first], (uint32_t)__builtin_popcountll(act));
    } else if (valid) {
        atomicAdd(&s_hist[len], 1u);
    }
}

__global__ __launch_bounds__(kScanThreads) void k_tally_scan(const uint8_t *__restrict__ qual,
                                                            const uint64_t *__restrict__ off,
                                                            uint64_t n, u64 *__restrict__ acc)
{
    __shared__ uint32_t s_hist[HPN_LEN_BINS + 1];  // last bin: length out of domain
    __shared__ uint32_t s_red[3][kScanThreads / kWave];
    const int tid = threadIdx.x;
    for (int i = tid; i <= HPN_LEN_BINS; i += kScanThreads) s_hist[i] = 0;
    __syncthreads();

    // ---- bytes: [off[0], off[n]) of qual, as 16-byte vectors from an aligned base ----
    const uint64_t b0 = off[0], b1 = off[n];
    uint32_t c20 = 0, c30 = 0, hi = 0;
    if (b1 > b0) {
        const uint64_t nbytes = b1 - b0;
        const uint8_t *pbeg = qual + b0;
        const int a0 = (int)((uintptr_t)pbeg & 15);
        const u32 *vec = reinterpret_cast<const u32 *>(pbeg - a0);  // vector i = bytes [16i-a0, 16i-a0+16)
        const uint64_t nv = (a0 + nbytes + 15) >> 4;
        // vectors 1 .. nv-2 are whole; the first and last may be partial
        if (nv > 2) {
            const uint64_t nvec = nv - 2;
            const uint64_t ntile = (nvec + kScanTileVec - 1) / kScanTileVec;
            for (uint64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
                const uint64_t base = 1 + t * kScanTileVec + tid;
                u32 v[kScanUnroll];
                if ((t + 1) * kScanTileVec <= nvec) {
#pragma unroll
                    for (int k = 0; k < kScanUnroll; ++k) v[k] = load_stream16(vec + base + (uint64_t)k * kScanThreads);
                } else {
#pragma unroll
                    for (int k = 0; k < kScanUnroll; ++k) {
                        const uint64_t idx = base + (uint64_t)k * kScanThreads;
                        v[k] = idx < nv - 1 ? load_stream16(vec + idx) : u32{0, 0, 0, 0};
                    }
                }
#pragma unroll
                for (int k = 0; k < kScanUnroll; ++k) swar16(v[k], c20, c30, hi);
            }
        }
        if (blockIdx.x == 0 && tid == 0) {
            const int e0 = (int)min((uint64_t)16, (uint64_t)a0 + nbytes);
            swar16(mask_bytes(vec[0], a0, e0), c20, c30, hi);
            if (nv > 1) swar16(mask_bytes(vec[nv - 1], 0, (int)((a0 + nbytes) - ((nv - 1) << 4))), c20, c30, hi);
        }
    }

    // ---- record lengths: off[] read once, as 16-byte pairs {off[e], off[e+1]} ----
    // e0 = first 16-byte aligned element; pair p holds elements e0+2p, e0+2p+1; the
    // boundary after the pair comes from the next lane (lane 63 reads it itself).
    const uint64_t e0 = ((uintptr_t)off >> 3) & 1;
    if (e0 && blockIdx.x == 0 && tid == 0 && n) {  // record 0 sits in front of the first aligned pair
        const uint64_t len = off[1] - off[0];
        atomicAdd(&s_hist[len < HPN_LEN_BINS ? (uint32_t)len : (uint32_t)HPN_LEN_BINS], 1u);
    }
    if (n + 1 > e0) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 *pairs = reinterpret_cast<const u64x2 *>(off + e0);
        const uint64_t npair = (n + 1 - e0 + 1) >> 1;
        const uint64_t ptiles = (npair + kLenTile - 1) / kLenTile;
        for (uint64_t t = blockIdx.x; t < ptiles; t += gridDim.x) {
            u64x2 v[kLenPerThread];
            uint64_t nx[kLenPerThread];
#pragma unroll
            for (int k = 0; k < kLenPerThread; ++k) {
                const uint64_t p = t * kLenTile + (uint64_t)k * kScanThreads + tid;
                v[k] = p < npair ? __builtin_nontemporal_load(pairs + p) : u64x2{0, 0};
                const uint64_t e = e0 + 2 * p + 2;  // element after the pair
                nx[k] = (lane_id() == kWave - 1 && e <= n) ? off[e] : 0;
            }
#pragma unroll
            for (int k = 0; k < kLenPerThread; ++k) {
                const uint64_t p = t * kLenTile + (uint64_t)k * kScanThreads + tid;
                const uint64_t e = e0 + 2 * p;      // record e = [off[e], off[e+1]), record e+1 = [off[e+1], off[e+2])
                const uint64_t from_next = __shfl_down(v[k][0], 1, kWave);
                const uint64_t after = lane_id() == kWave - 1 ? nx[k] : from_next;
                const uint64_t l0 = v[k][1] - v[k][0], l1 = after - v[k][1];
                hist_len(s_hist, p < npair && e < n, l0 < HPN_LEN_BINS ? (uint32_t)l0 : (uint32_t)HPN_LEN_BINS);
                hist_len(s_hist, p < npair && e + 1 < n, l1 < HPN_LEN_BINS ? (uint32_t)l1 : (uint32_t)HPN_LEN_BINS);
            }
        }
    }

    // ---- workgroup reduction, one global atomic per counter per workgroup ----
    c20 = wave_sum(c20);
    c30 = wave_sum(c30);
    hi = wave_or(hi);
    if (lane_id() == 0) {
        s_red[0][wave_id()] = c20;
        s_red[1][wave_id()] = c30;
        s_red[2][wave_id()] = hi;
    }
    __syncthreads();
    if (tid == 0) {
        u64 s20 = 0, s30 = 0;
        uint32_t h = 0;
        for (int w = 0; w < kScanThreads / kWave; ++w) s20 += s_red[0][w], s30 += s_red[1][w], h |= s_red[2][w];
        if (s20) atomicAdd(&acc[HPN_TALLY_W_Q20], s20);
        if (s30) atomicAdd(&acc[HPN_TALLY_W_Q30], s30);
        if (h & 0x80808080u) atomicAdd(&acc[HPN_TALLY_W_BAD], (u64)1);
        if (blockIdx.x == 0) atomicAdd(&acc[HPN_TALLY_W_TOTAL], (u64)(b1 - b0));
    }
    for (int i = tid; i <= HPN_LEN_BINS; i += kScanThreads) {
        const uint32_t h = s_hist[i];
        if (h) atomicAdd(&acc[i < HPN_LEN_BINS ? HPN_TALLY_W_SEQLEN + i : HPN_TALLY_W_BAD], (u64)h);
    }
}

// ---------------------------------------------------------------------------
// K1L: full per-(symbol, cycle) histograms
// ---------------------------------------------------------------------------
constexpr int kHistThreads = 1024;
constexpr int kHistRecs = 1024;           // records per chunk
constexpr int kRowWords = HPN_LEN_BINS / 2 + 1;  // 256 dwords of packed u16 pairs + 1 pad:
                                                 // bank = (sym + pos/2) % 32, so lanes at one
                                                 // cycle but different symbols spread over banks
constexpr uint32_t kFlushReads = 65535;   // a counter gets at most one hit per read

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
__device__ __forceinline__ void hist_byte(HistLds &s, uint32_t byte, uint32_t pos, uint32_t &c20,
                                          uint32_t &c30, uint32_t &bad)
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
__device__ __forceinline__ void hist_stream(HistLds &s, const uint8_t *arr, uint64_t base_off,
                                            uint32_t cnt, uint32_t &c20, uint32_t &c30, uint32_t &bad)
{
    const uint32_t B = s.loff[cnt];
    if (B == 0) return;
    const uintptr_t p0 = (uintptr_t)arr + base_off;
    const int a0 = (int)(p0 & 15);
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
            if (s.loff[mid] > b) hi = mid; else lo = mid + 1;
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
                                                            const uint64_t *__restrict__ off,
                                                            uint64_t n, u64 *__restrict__ acc)
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
        // any over-long record poisons position tracking: skip the chunk's bytes,
        // the batch is rejected as a whole (HPN_E_DOMAIN) once `bad` is seen
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

// ---------------------------------------------------------------------------
// launchers (called from hpn_abi.cpp)
// ---------------------------------------------------------------------------
hipError_t launch_tally_scan(const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint64_t approx_bytes,
                             u64 *d_acc, int n_cu, hipStream_t st)
{
    const uint64_t vec_tiles = approx_bytes / (16ull * kScanTileVec) + 1;
    const uint64_t len_tiles = n / kLenTile + 1;
    uint64_t want = vec_tiles > len_tiles ? vec_tiles : len_tiles;
    const uint64_t cap = (uint64_t)n_cu * 8;  // 8 workgroups of 4 waves = 32 waves per CU
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(k_tally_scan, dim3(grid), dim3(kScanThreads), 0, st, d_qual, d_off, n, d_acc);
    return hipGetLastError();
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
