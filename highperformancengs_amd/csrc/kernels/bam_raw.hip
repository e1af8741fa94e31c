// bam_raw.hip -- BAM records straight from inflated BGZF blocks on the device.
//
// The reference walks a BAM with bam_read1 (samtools-0.1.19 bam.c:191): block_size, 32 bytes of
// core fields, name, CIGAR, packed sequence, qualities -- one record after the other on the
// host.  With the blocks inflated on the GPU (bgzf_inflate.hip) the record chain is walked
// there too.  A chain is serial, but samtools never lets a record straddle a BGZF block
// (bam_write1 calls bgzf_flush_try(4 + block_len), bam.c:238; the header is followed by a
// flush), so every block starts at a record boundary and the blocks can be walked
// independently: one lane per block hops through its ~200 records.  Files written otherwise
// (records packed across blocks) are detected -- a walk that does not end exactly at its
// block's end -- and the tools then decode the file on the host.
//
//   k_raw_count    per block: number of records, smallest / largest refID, chain check
//   k_raw_scan     exclusive scan of the per-block counts (one workgroup; a batch has ~10^4 blocks)
//   k_raw_index    per block: byte offset of every record -> rec_off[]
//   k_raw_fields   per record: refID, pos, flag, l_seq, offset of the packed sequence
//                  (the SoA view k_window_add takes; the sequence stays where it is)
// K3 (bam2depth.c:86-110) reads core fields and CIGAR in place through RawRecs (bam_depth.hip).
#include "common.hpp"

namespace hpn {

struct RawBlock {  // = hpn_bgzf_block
    uint64_t in_off;
    uint32_t in_len, out_len;
    uint64_t out_off;
};

constexpr int kRawThreads = 256;

__device__ __forceinline__ uint32_t ld32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

// info words: [0] flags (1 = chain does not end at the block end, 2 = block failed to inflate),
// [1] min refID, [2] max refID (as int32), [3] unused; total record count goes to bases[n_blocks]
__global__ __launch_bounds__(kRawThreads) void k_raw_count(const uint8_t *__restrict__ raw, const RawBlock *__restrict__ blocks,
                                                           uint32_t n_blocks, uint32_t first_off,
                                                           const uint32_t *__restrict__ status, uint32_t *__restrict__ counts,
                                                           int32_t *__restrict__ info)
{
    const uint32_t b = blockIdx.x * kRawThreads + threadIdx.x;
    if (b >= n_blocks) return;
    if (status[b]) {
        atomicOr((uint32_t *)&info[0], 2u);
        counts[b] = 0;
        return;
    }
    const RawBlock blk = blocks[b];
    const uint8_t *p = raw + blk.out_off;
    uint32_t at = b == 0 ? first_off : 0u, n = 0;
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    bool broken = at > blk.out_len;
    while (!broken && at + 4u <= blk.out_len) {
        const uint32_t bs = ld32(p + at);
        if (bs < 32u || bs > blk.out_len - at - 4u) {  // a record never ends beyond its block here
            broken = true;
            break;
        }
        // the variable-length fields must fit the record (bam_read1 trusts them, bam.c:191; the later kernels
        // read name, CIGAR and sequence in place, so a record that lies about them is caught here)
        const uint32_t l_name = p[at + 12u], n_cigar = ld16(p + at + 16u), l_seq = ld32(p + at + 20u);
        if (l_seq > 0x7fffffffu ||
            32ull + l_name + 4ull * n_cigar + (((uint64_t)l_seq + 1u) >> 1) + (uint64_t)l_seq > (uint64_t)bs) {
            broken = true;
            break;
        }
        const int32_t tid = (int32_t)ld32(p + at + 4u);
        lo = tid < lo ? tid : lo, hi = tid > hi ? tid : hi;
        at += 4u + bs;
        ++n;
    }
    if (broken || at != blk.out_len) atomicOr((uint32_t *)&info[0], 1u);
    counts[b] = n;
    if (n) atomicMin(&info[1], lo), atomicMax(&info[2], hi);
}

__global__ __launch_bounds__(1024) void k_raw_scan(const uint32_t *__restrict__ counts, uint32_t n_blocks, u64 *__restrict__ bases)
{
    __shared__ u64 s_wave[16];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_blocks; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        const u64 v = i < n_blocks ? counts[i] : 0;
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
        if (i < n_blocks) bases[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) bases[n_blocks] = s_carry;
}

__global__ __launch_bounds__(kRawThreads) void k_raw_index(const uint8_t *__restrict__ raw, const RawBlock *__restrict__ blocks,
                                                           uint32_t n_blocks, uint32_t first_off, const uint32_t *__restrict__ counts,
                                                           const u64 *__restrict__ bases, uint64_t *__restrict__ rec_off)
{
    const uint32_t b = blockIdx.x * kRawThreads + threadIdx.x;
    if (b >= n_blocks) return;
    const RawBlock blk = blocks[b];
    const uint8_t *p = raw + blk.out_off;
    uint32_t at = b == 0 ? first_off : 0u;
    u64 r = bases[b];
    for (uint32_t k = 0, n = counts[b]; k < n; ++k) {
        rec_off[r++] = blk.out_off + at;
        at += 4u + ld32(p + at);
    }
}

// bam1_core_t on disk (bam.h:178-187) behind block_size: refID, pos, bin_mq_nl, flag_nc, l_seq, ...
__global__ __launch_bounds__(kRawThreads) void k_raw_fields(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off,
                                                            uint64_t n, int32_t *__restrict__ tid, int32_t *__restrict__ pos,
                                                            uint32_t *__restrict__ flag, int32_t *__restrict__ l_qseq,
                                                            uint64_t *__restrict__ seq_off)
{
    for (uint64_t r = (uint64_t)blockIdx.x * kRawThreads + threadIdx.x; r < n; r += (uint64_t)gridDim.x * kRawThreads) {
        const uint64_t o = rec_off[r];
        const uint8_t *p = raw + o;
        const uint32_t l_name = p[12], n_cigar = ld16(p + 16);
        tid[r] = (int32_t)ld32(p + 4);
        pos[r] = (int32_t)ld32(p + 8);
        flag[r] = ld16(p + 18);
        l_qseq[r] = (int32_t)ld32(p + 20);
        seq_off[r] = o + 36u + l_name + 4u * n_cigar;
    }
}

hipError_t launch_raw_count(const uint8_t *raw, const void *blocks, uint32_t n_blocks, uint32_t first_off, const uint32_t *status,
                            uint32_t *counts, u64 *bases, int32_t *info, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_raw_count, dim3((n_blocks + kRawThreads - 1) / kRawThreads), dim3(kRawThreads), 0, st, raw,
                       (const RawBlock *)blocks, n_blocks, first_off, status, counts, info);
    hipLaunchKernelGGL(k_raw_scan, dim3(1), dim3(1024), 0, st, counts, n_blocks, bases);
    return hipGetLastError();
}

hipError_t launch_raw_index(const uint8_t *raw, const void *blocks, uint32_t n_blocks, uint32_t first_off, const uint32_t *counts,
                            const u64 *bases, uint64_t *rec_off, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_raw_index, dim3((n_blocks + kRawThreads - 1) / kRawThreads), dim3(kRawThreads), 0, st, raw,
                       (const RawBlock *)blocks, n_blocks, first_off, counts, bases, rec_off);
    return hipGetLastError();
}

static unsigned rec_grid(uint64_t n, int n_cu)
{
    const uint64_t want = (n + kRawThreads - 1) / kRawThreads, cap = (uint64_t)n_cu * 8;
    return (unsigned)(want < cap ? want : cap);
}

hipError_t launch_raw_fields(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, int32_t *tid, int32_t *pos, uint32_t *flag,
                             int32_t *l_qseq, uint64_t *seq_off, int n_cu, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_raw_fields, dim3(rec_grid(n, n_cu)), dim3(kRawThreads), 0, st, raw, rec_off, n, tid, pos, flag, l_qseq,
                       seq_off);
    return hipGetLastError();
}

}  // namespace hpn
