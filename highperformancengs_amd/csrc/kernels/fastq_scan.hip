// fastq_scan.hip -- K1 `k_tally_scan`: what fastq_count prints, from one flat pass.
//
// Replaces the scan loop of count_read (reference fastq_count.c:112-119,
// fastq_count_kthread.c:126-135; AssignQuality :29-35) for the default report:
// SeqLen[512], sum, sum(q>=53), sum(q>=63) (statSeqLen :63-74, statQ :37-47,124).
// Quality[q][pos] is only ever reduced over pos and over q>=53 / q>=63, so the
// positions are not needed: the byte range [off[0], off[n]) of qual[] is scanned
// flat with 16-byte loads and SWAR compares; off[] is read once, as 16-byte
// pairs, for the length histogram.
//
// Work = byte tiles (256 lanes x U x 16 B) and pair tiles (the pair chunks spread
// evenly among the byte chunks), handed out in chunks of 8 tiles from ONE device
// counter (fetched one chunk ahead, so the atomic's latency hides under the current chunk).  All workgroups therefore
// stream one compact, advancing window of HBM and finish together; a static
// grid-stride split lets workgroups drift apart over a 158 GB launch and
// cost 3 % (scripts/k1_split.py, profiles/r01).
// Bound: HBM read, 1 B per base + 8 B per record.  No MFMA: no contraction here.
#include <stdlib.h>

#include <type_traits>

#include "tally_util.hpp"

namespace hpn {

constexpr int kScanThreads = 256;
constexpr int kPairPerThread = 8;
constexpr int kPairTile = kScanThreads * kPairPerThread;  // 2048 pairs = 4096 records = 32 KiB of off[]
constexpr int kChunkTiles = 8;

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// kSched: 0 = static grid-stride split, 1 = chunks from one counter, pair chunks last, 2 = ... pair chunks spread among the byte chunks
template <int U, bool kNt, int kSched>
__global__ __launch_bounds__(kScanThreads) void k_tally_scan(const uint8_t *__restrict__ qual,
                                                            const uint64_t *__restrict__ off, uint64_t n,
                                                            u64 *__restrict__ acc, u64 *__restrict__ sched)
{
    constexpr int kTileVec = kScanThreads * U;  // 16-byte vectors per byte tile
    __shared__ uint32_t s_hist[HPN_LEN_BINS + 1];  // last bin: length out of domain
    __shared__ uint32_t s_red[3][kScanThreads / kWave];
    __shared__ u64 s_chunk[2];
    const int tid = threadIdx.x;
    for (int i = tid; i <= HPN_LEN_BINS; i += kScanThreads) s_hist[i] = 0;

    // ---- geometry: bytes [off[0], off[n]) as 16-byte vectors from an aligned base ----
    const uint64_t b0 = off[0], b1 = off[n];
    const uint64_t nbytes = b1 > b0 ? b1 - b0 : 0;
    const uint8_t *pbeg = qual + b0;
    const int a0 = (int)((uintptr_t)pbeg & 15);
    const u32 *vec = reinterpret_cast<const u32 *>(pbeg - a0);  // vector i = bytes [16i-a0, 16i-a0+16)
    const uint64_t nv = nbytes ? (a0 + nbytes + 15) >> 4 : 0;
    const uint64_t nvec = nv > 2 ? nv - 2 : 0;  // vectors 1 .. nv-2 are whole; first and last may be partial
    const uint64_t btiles = (nvec + kTileVec - 1) / kTileVec;
    // off[] as pairs {off[e], off[e+1]} from the first 16-byte aligned element e0
    const uint64_t e0 = ((uintptr_t)off >> 3) & 1;
    const u64x2 *pairs = reinterpret_cast<const u64x2 *>(off + e0);
    const uint64_t npair = n + 1 > e0 ? (n + 1 - e0 + 1) >> 1 : 0;
    const uint64_t ptiles = (npair + kPairTile - 1) / kPairTile;
    const uint64_t tiles = btiles + ptiles;

    uint32_t c20 = 0, c30 = 0, hi = 0;
    auto ld = [](const u32 *p) { return kNt ? __builtin_nontemporal_load(p) : *p; };

    auto byte_tile = [&](uint64_t t) {
        const uint64_t base = 1 + t * kTileVec + tid;
        u32 v[U];
        if ((t + 1) * kTileVec <= nvec) {
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = ld(vec + base + (uint64_t)k * kScanThreads);
        } else {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint64_t idx = base + (uint64_t)k * kScanThreads;
                v[k] = idx < nv - 1 ? ld(vec + idx) : u32{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) swar16(v[k], c20, c30, hi);
    };

    // pair p holds elements e0+2p, e0+2p+1: record e = [off[e], off[e+1]), e+1 = [off[e+1], off[e+2]).  A wave owns a
    // contiguous span of 64 x kPairPerThread pairs (row k = its pairs 64k .. 64k+63), so the boundary after a pair comes
    // from the next lane, for lane 63 from lane 0 of the wave's next row, and only after the span's last pair from memory:
    // one single-lane load per wave and tile (it was one per row, as many load instructions again as the rows themselves;
    // with 4 rows in flight the offsets passed at ~4 TB/s while the bytes pass at 6.8).
    auto pair_tile = [&](uint64_t t) {
        const uint64_t wbase = t * kPairTile + (uint64_t)wave_id() * (kWave * kPairPerThread) + lane_id();
        u64x2 v[kPairPerThread];
#pragma unroll
        for (int k = 0; k < kPairPerThread; ++k) {
            const uint64_t p = wbase + (uint64_t)k * kWave;
            v[k] = p < npair ? __builtin_nontemporal_load(pairs + p) : u64x2{0, 0};
        }
        const uint64_t e_tail = e0 + 2 * (wbase + (uint64_t)(kPairPerThread - 1) * kWave) + 2;
        const uint64_t tail = (lane_id() == kWave - 1 && e_tail <= n) ? off[e_tail] : 0;
        uint32_t la[kPairPerThread], lb[kPairPerThread];
#pragma unroll
        for (int k = 0; k < kPairPerThread; ++k) {
            const uint64_t from_next = __shfl_down(v[k][0], 1, kWave);
            const uint64_t next_row = k + 1 < kPairPerThread ? __shfl(v[k + 1 < kPairPerThread ? k + 1 : k][0], 0, kWave) : tail;
            const uint64_t after = lane_id() == kWave - 1 ? next_row : from_next;
            const uint64_t l0 = v[k][1] - v[k][0], l1 = after - v[k][1];
            la[k] = l0 < HPN_LEN_BINS ? (uint32_t)l0 : (uint32_t)HPN_LEN_BINS;
            lb[k] = l1 < HPN_LEN_BINS ? (uint32_t)l1 : (uint32_t)HPN_LEN_BINS;
        }
        // the whole span valid and of one length (fixed-length reads, every tile but the last): ONE add for its 1024 records
        const uint64_t p_last = wbase + (uint64_t)(kPairPerThread - 1) * kWave;
        const uint32_t first = (uint32_t)__shfl((int)la[0], 0, kWave);
        bool same = p_last < npair && e0 + 2 * p_last + 1 < n;
#pragma unroll
        for (int k = 0; k < kPairPerThread; ++k) same = same && la[k] == first && lb[k] == first;
        if (__ballot(!same) == 0) {
            if (lane_id() == 0) atomicAdd(&s_hist[first], (uint32_t)(2 * kWave * kPairPerThread));
            return;
        }
        // Mixed lengths (trimmed reads).  Two lengths are counted with ballots into scalar registers and added once per tile by one
        // lane: the span's LONGEST length (what trimming left untouched: 70 % of the lanes adding to one LDS address serialise) and
        // the span's first; every other length adds by itself.  ~11 instructions per length (round 4's hist_len per row: ~20, and
        // with two waves per SIMD a pair tile of ragged reads was bound by its instructions, not by HBM: 0.82 of peak).
        // (a span that lies wholly inside the batch -- all but the batch's last -- skips the per-length validity tests)
        const bool whole = __ballot(!(p_last < npair && e0 + 2 * p_last + 1 < n)) == 0;
        auto ragged = [&](auto whole_tag) {
            constexpr bool kWhole = decltype(whole_tag)::value;
            auto ok = [&](int k, int second) {
                if (kWhole) return true;
                const uint64_t p = wbase + (uint64_t)k * kWave;
                return p < npair && e0 + 2 * p + (uint64_t)second < n;
            };
            uint32_t mx = 0;
#pragma unroll
            for (int k = 0; k < kPairPerThread; ++k) {
                if (ok(k, 0)) mx = la[k] > mx ? la[k] : mx;
                if (ok(k, 1)) mx = lb[k] > mx ? lb[k] : mx;
            }
            const uint32_t m1 = wave_max(mx), m2 = first;
            uint32_t n1 = 0, n2 = 0;
            auto count = [&](bool valid, uint32_t len) {
                const bool is1 = valid && len == m1, is2 = valid && len == m2;
                n1 += (uint32_t)__builtin_popcountll(__ballot(is1));
                n2 += (uint32_t)__builtin_popcountll(__ballot(is2));
                if (valid && !is1 && !is2) atomicAdd(&s_hist[len], 1u);
            };
#pragma unroll
            for (int k = 0; k < kPairPerThread; ++k) {
                count(ok(k, 0), la[k]);
                count(ok(k, 1), lb[k]);
            }
            if (lane_id() == 0) {
                if (n1) atomicAdd(&s_hist[m1], n1);
                if (n2 && m2 != m1) atomicAdd(&s_hist[m2], n2);
            }
        };
        if (whole) ragged(std::true_type{});
        else ragged(std::false_type{});
    };
    auto do_tile = [&](uint64_t t) {  // (the static split: byte tiles, then pair tiles)
        if (t < btiles) byte_tile(t);
        else pair_tile(t - btiles);
    };

    constexpr bool kDyn = kSched != 0;
    if (kDyn) {
        // kSched 2: chunks of byte tiles with the chunks of pair tiles spread evenly among them (round 5; until then all pair tiles came last).
        // A pair tile of reads of ONE length is a handful of loads and one LDS add; of ragged reads (trimmed data) it is 16 LDS
        // atomics per lane, and with every workgroup in its pair tiles at the same time -- the end of the launch -- that phase ran
        // at the speed of the LDS atomics, not of HBM (0.82 of peak on lengths 100..151 against 0.86 on 150).  Spread out, one
        // workgroup in twenty is in a pair chunk while the others stream bytes.  The window the workgroups stream stays compact:
        // a pair chunk's offsets belong to records whose bytes are being read at about the same time.
        const uint64_t nb = (btiles + kChunkTiles - 1) / kChunkTiles, np = (ptiles + kChunkTiles - 1) / kChunkTiles;
        const uint64_t nchunk = nb + np;
        const uint32_t every = kSched == 2 && np ? (uint32_t)(nchunk / np) : 0u;       // (< 2^32 chunks: 2^32 x 256 KiB is a petabyte; 0: pair chunks last)
        if (tid == 0) s_chunk[0] = atomicAdd(&sched[0], (u64)1);
        __syncthreads();  // also orders the s_hist clear before its first use
        int par = 0;
        uint64_t c = s_chunk[0];
        while (c < nchunk) {
            u64 nxt = 0;
            if (tid == 0) nxt = atomicAdd(&sched[0], (u64)1);  // in flight while this chunk streams
            // which chunk is this: pair chunk g (the last chunk of group g of `every`), or the byte chunk behind g pair chunks
            const uint32_t g = every ? (uint32_t)c / every : (c >= nb ? (uint32_t)(c - nb) : 0u);
            const bool is_pair = every ? (g < np && (uint32_t)c - g * every == every - 1) : c >= nb;
            if (is_pair) {
                const uint64_t t1 = min((uint64_t)(g + 1) * kChunkTiles, ptiles);
                for (uint64_t t = (uint64_t)g * kChunkTiles; t < t1; ++t) pair_tile(t);
            } else {
                const uint64_t cb = every ? c - (g < np ? g : np) : c;
                const uint64_t t1 = min((cb + 1) * kChunkTiles, btiles);
                for (uint64_t t = cb * kChunkTiles; t < t1; ++t) byte_tile(t);
            }
            if (tid == 0) s_chunk[par ^ 1] = nxt;
            __syncthreads();
            par ^= 1;
            c = s_chunk[par];
        }
    } else {
        __syncthreads();
        for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) do_tile(t);
    }

    if (blockIdx.x == 0 && tid == 0) {
        if (nv) {  // the two possibly partial vectors at the ends of the byte range
            const int e = (int)min((uint64_t)16, (uint64_t)a0 + nbytes);
            swar16(mask_bytes(vec[0], a0, e), c20, c30, hi);
            if (nv > 1) swar16(mask_bytes(vec[nv - 1], 0, (int)((a0 + nbytes) - ((nv - 1) << 4))), c20, c30, hi);
        }
        if (e0 && n) {  // record 0 sits in front of the first aligned pair
            const uint64_t len = off[1] - off[0];
            atomicAdd(&s_hist[len < HPN_LEN_BINS ? (uint32_t)len : (uint32_t)HPN_LEN_BINS], 1u);
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
        if (blockIdx.x == 0) atomicAdd(&acc[HPN_TALLY_W_TOTAL], (u64)nbytes);
    }
    for (int i = tid; i <= HPN_LEN_BINS; i += kScanThreads) {
        const uint32_t h = s_hist[i];
        if (h) atomicAdd(&acc[i < HPN_LEN_BINS ? HPN_TALLY_W_SEQLEN + i : HPN_TALLY_W_BAD], (u64)h);
    }
    if (kDyn && tid == 0) {
        // the last workgroup to leave rewinds the chunk counter for the next launch;
        // every other workgroup made its final fetch before it signed off here
        if (atomicAdd(&sched[1], (u64)1) == (u64)gridDim.x - 1) {
            atomicExch(&sched[0], (u64)0);
            atomicExch(&sched[1], (u64)0);
        }
    }
}

// Tuning knobs for A/B runs (test-hooks builds; scripts/k1_sweep.py): HPN_K1_VARIANT = unroll*100 + nt*10 + sched
// (default 811: 8 loads in flight, non-temporal, dynamic chunks with the pair chunks last; ..2: pair chunks spread out), HPN_K1_WG_PER_CU.
hipError_t launch_tally_scan(const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint64_t approx_bytes,
                             u64 *d_acc, u64 *d_sched, int n_cu, hipStream_t st)
{
    const char *ev = test_env("HPN_K1_VARIANT"), *eg = test_env("HPN_K1_WG_PER_CU");
    const int variant = ev ? atoi(ev) : 811;
    const int unroll = variant / 100;
    // workgroups (4 waves each) per CU.  Long launches: 2 -- same-session sweep over 158 GB: 2 -> 24.01 ms, 3 -> 24.35,
    // 4 -> 24.51, 6 -> 24.63, 8/16 slower still (profiles/r01e/k1_sweep_wg.txt; a loads-only kernel reads 6.6-6.9 TB/s
    // with 2 per CU, 6.3-6.4 with 4: profiles/r01e/hbm_read_ubench.txt).  Short launches keep 4 for the latency.
    const uint64_t per_cu = eg ? (uint64_t)atoi(eg) : (approx_bytes >= (1ull << 30) ? 2 : 4);
    if (unroll <= 0) return hipErrorInvalidValue;
    const uint64_t tiles = approx_bytes / (16ull * kScanThreads * unroll) + n / (2 * kPairTile) + 2;
    const uint64_t want = (variant % 10) ? (tiles + kChunkTiles - 1) / kChunkTiles : tiles;
    const uint64_t cap = (uint64_t)n_cu * per_cu;
    const dim3 grid((unsigned)(want < cap ? want : cap)), block(kScanThreads);
    switch (variant) {
#define HPN_K1(U, NT, DYN) \
    case U * 100 + NT * 10 + DYN: \
        hipLaunchKernelGGL((k_tally_scan<U, NT, DYN>), grid, block, 0, st, d_qual, d_off, n, d_acc, d_sched); \
        break;
        HPN_K1(4, 1, 0) HPN_K1(4, 1, 1) HPN_K1(8, 0, 0) HPN_K1(8, 0, 1) HPN_K1(8, 1, 0) HPN_K1(8, 1, 1) HPN_K1(8, 1, 2) HPN_K1(16, 1, 0)
        HPN_K1(16, 1, 1)
#undef HPN_K1
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace hpn
