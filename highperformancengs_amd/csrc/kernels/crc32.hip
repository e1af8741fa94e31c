// crc32.hip -- CRC-32 (RFC 1952 / ISO 3309, reflected polynomial 0xEDB88320) of byte ranges that lie in HBM.
//
// zlib's gzread -- behind the 4 x gzgets loops of the reference (fastq_count.c:112-118, IO_stream.h:122-136) -- checks the
// CRC-32 of every gzip member and stops handing out bytes when one fails.  A file inflated on the device is checked here:
// a workgroup takes a block of 64 KiB, every thread the CRC of its own 256 bytes (table in LDS, a byte per step), and the
// 256 values are folded pairwise in LDS with CRC(A || B) = CRC(A) * x^(8 |B|) mod P  xor  CRC(B); the host folds the blocks of
// a range the same way (hpn_gz.hip).  Bound: LDS table lookups (a byte each); ~100x faster than the inflate it checks.
#include "common.hpp"

namespace hpn {

constexpr uint32_t kCrcPoly = 0xedb88320u;
constexpr int kCrcThreads = 256;
constexpr uint32_t kCrcChunk = 256;                          // bytes per thread
constexpr uint32_t kCrcBlock = kCrcThreads * kCrcChunk;      // bytes per workgroup: 64 KiB
constexpr uint32_t kCrcStride = kCrcChunk / 4 + 1;           // words between the chunks of neighbouring threads in LDS: odd, no bank conflicts

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
__host__ __device__ inline uint32_t crc_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0) break;
        }
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

// x^(8 n) mod P.  pow2[k] = x^(2^k) mod P.
struct CrcPow {
    uint32_t pow2[64];
};
__host__ __device__ inline uint32_t crc_xpow8(const CrcPow &t, uint64_t n)
{
    uint32_t p = 0x80000000u;   // x^0
    for (int k = 3; n; n >>= 1, ++k)
        if (n & 1u) p = crc_mul(t.pow2[k & 63], p);
    return p;
}
CrcPow crc_pow_table()
{
    CrcPow t;
    uint32_t p = 0x40000000u;   // x^1
    for (int k = 0; k < 64; ++k) {
        t.pow2[k] = p;
        p = crc_mul(p, p);
    }
    return t;
}
// CRC of A || B from CRC(A), CRC(B) and |B|
uint32_t crc_combine(const CrcPow &t, uint32_t crc_a, uint32_t crc_b, uint64_t len_b) { return crc_mul(crc_xpow8(t, len_b), crc_a) ^ crc_b; }

struct CrcBlock {
    uint64_t off;
    uint32_t len, reserved;
};

__global__ __launch_bounds__(kCrcThreads) void k_crc32_blocks(const uint8_t *__restrict__ data, const CrcBlock *__restrict__ blocks,
                                                              uint32_t *__restrict__ out, CrcPow pw)
{
    __shared__ uint32_t s_tab[256];
    __shared__ uint32_t s_d[kCrcThreads * kCrcStride];
    __shared__ uint32_t s_crc[kCrcThreads], s_len[kCrcThreads];
    const int tid = threadIdx.x;
    const CrcBlock b = blocks[blockIdx.x];
    {   // the byte table: tab[i] = CRC register after shifting byte i through
        uint32_t c = (uint32_t)tid;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        s_tab[tid] = c;
    }
    // the block into LDS, coalesced: 16-byte piece q of the block belongs to thread q / 16
    const uint8_t *src = data + b.off;
    for (uint32_t q = (uint32_t)tid; q * 16u < b.len; q += kCrcThreads) {
        u32 v = u32{0, 0, 0, 0};
        if (q * 16u + 16u <= b.len) __builtin_memcpy(&v, src + (size_t)q * 16u, 16);
        else
            for (uint32_t k = q * 16u; k < b.len; ++k) reinterpret_cast<uint8_t *>(&v)[k - q * 16u] = src[k];
        uint32_t *d = s_d + (q >> 4) * kCrcStride + (q & 15u) * 4u;
        d[0] = v[0], d[1] = v[1], d[2] = v[2], d[3] = v[3];
    }
    __syncthreads();
    const uint32_t lo = (uint32_t)tid * kCrcChunk;
    const uint32_t mine = b.len > lo ? (b.len - lo < kCrcChunk ? b.len - lo : kCrcChunk) : 0u;
    uint32_t c = 0xffffffffu;
    const uint32_t *w = s_d + (uint32_t)tid * kCrcStride;
    for (uint32_t k = 0; k < mine; k += 4) {
        uint32_t x = w[k >> 2];
        const uint32_t nb = mine - k < 4u ? mine - k : 4u;
        for (uint32_t j = 0; j < nb; ++j, x >>= 8) c = s_tab[(c ^ x) & 0xffu] ^ (c >> 8);
    }
    s_crc[tid] = ~c;           // (of zero bytes: 0)
    s_len[tid] = mine;
    __syncthreads();
    for (int s = 1; s < kCrcThreads; s <<= 1) {
        if ((tid & (2 * s - 1)) == 0) {
            const uint32_t lb = s_len[tid + s];
            if (lb) {
                // full chunks behind a power-of-two number of threads: x^(8 lb) is a table entry
                const uint32_t xp = (lb & (lb - 1u)) == 0 ? pw.pow2[(__builtin_ctz(lb) + 3) & 63] : crc_xpow8(pw, lb);
                s_crc[tid] = crc_mul(xp, s_crc[tid]) ^ s_crc[tid + s];
                s_len[tid] += lb;
            }
        }
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = s_crc[0];
}

uint32_t crc_block_bytes() { return kCrcBlock; }

hipError_t launch_crc32_blocks(const uint8_t *d_data, const void *d_blocks, uint32_t n_blocks, uint32_t *d_out, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    static const CrcPow pw = crc_pow_table();
    hipLaunchKernelGGL(k_crc32_blocks, dim3(n_blocks), dim3(kCrcThreads), 0, st, d_data, (const CrcBlock *)d_blocks, d_out, pw);
    return hipGetLastError();
}

// host side of the fold: CRC of a range from the CRCs of its blocks (all of kCrcBlock bytes but the last)
uint32_t crc_fold_blocks(const uint32_t *crcs, uint64_t n_blocks, uint64_t total_len)
{
    static const CrcPow pw = crc_pow_table();
    static const uint32_t x_block = crc_xpow8(pw, kCrcBlock);
    uint32_t c = 0;
    uint64_t left = total_len;
    for (uint64_t k = 0; k < n_blocks; ++k) {
        const uint64_t len = left < kCrcBlock ? left : kCrcBlock;
        c = crc_mul(len == kCrcBlock ? x_block : crc_xpow8(pw, len), c) ^ crcs[k];
        left -= len;
    }
    return c;
}
uint32_t crc_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b)
{
    static const CrcPow pw = crc_pow_table();
    return crc_combine(pw, crc_a, crc_b, len_b);
}

}  // namespace hpn
