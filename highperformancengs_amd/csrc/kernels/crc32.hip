// crc32.hip -- CRC-32 (RFC 1952 / ISO 3309, reflected polynomial 0xEDB88320) of byte ranges that lie in HBM.
//
// zlib's gzread -- behind the 4 x gzgets loops of the reference (fastq_count.c:112-118, IO_stream.h:122-136) -- checks the
// CRC-32 of every gzip member and stops handing out bytes when one fails.  A file inflated on the device is checked here:
// a workgroup takes a block of 64 KiB (k_crc32_blocks below: columns folded down rows of 4 KiB with table look-ups that do not
// depend on each other); the host joins the blocks of a range with CRC(A || B) = CRC(A) * x^(8 |B|) mod P  xor  CRC(B)
// (hpn_gz.hip).  Bound: LDS table look-ups (one per byte).
#include "common.hpp"

namespace hpn {

constexpr uint32_t kCrcPoly = 0xedb88320u;
constexpr int kCrcThreads = 256;
constexpr uint32_t kCrcChunk = 256;                          // bytes per thread
constexpr uint32_t kCrcBlock = kCrcThreads * kCrcChunk;      // bytes per workgroup: 64 KiB

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

// ---- round 6: columns instead of chunks --------------------------------------------------------------------------------------
// Round 3's kernel gave every thread 256 consecutive bytes (staged through 66 KB of LDS to keep the loads coalesced) and a
// byte-at-a-time table walk: 256 DEPENDENT LDS look-ups per thread, two workgroups per CU -- 0.82 TB/s, 5.6 % of the gzip
// route's device time (profiles/r05/kernel_stats_gz_tool_final.csv).  The CRC register without its pre- and post-inversion is
// LINEAR in the message (L below), so the sum can be taken in any order:
//   * a workgroup reads its 64 KiB block in rows of 4 KiB, thread t the 16 bytes at 16 t of every row (whole cache lines per wave
//     instruction, no staging): four 32-bit COLUMNS per thread, each folded down the rows by Horner's rule,
//     A = A * x^(8 * 4096) xor w -- a multiplication by a constant = four look-ups in tables made for that constant, all sixteen
//     of a row independent of each other;
//   * behind the last row a thread shifts its four columns together (the ordinary word steps of a CRC over 16 bytes), multiplies
//     by x^(8 (4080 - 16 t)) -- what lies between its 16 bytes and the row's end -- and the 256 values are xor'ed.
// A short block is laid against the END of the 64 KiB (zeros in front of a message do not change L).  The workgroup hands out
// L(block); pre- and post-inversion are K(len) = crc32 of len zero bytes = 0xffffffff * x^(8 len) xor 0xffffffff, added by the host
// (crc_fold_blocks), which then joins the blocks as before.  LDS: 6 KiB of tables.  Bound: LDS look-ups, one per byte, independent.
constexpr uint32_t kCrcRow = kCrcThreads * 16u;               // 4 KiB
constexpr uint32_t kCrcRows = kCrcBlock / kCrcRow;           // 16

struct CrcTabs {
    uint32_t word[4][256];     // word[k][b]: (b << 8k) * x^32            -- the ordinary slicing-by-4 step
    uint32_t row[4][256];      // row[k][b]:  (b << 8k) * x^(8 * kCrcRow)   -- a column's step down one row
    uint32_t tail[kCrcThreads];// tail[t]:    x^(8 * (kCrcRow - 16 - 16 t)) -- from thread t's 16 bytes to the row's end
};
constexpr uint32_t crc_mul_c(uint32_t a, uint32_t b)          // (crc_mul, usable in a constant expression)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
constexpr uint32_t crc_xpow_c(uint64_t bits)                  // x^bits mod P
{
    uint32_t p = 0x80000000u, sq = 0x40000000u;               // x^0, x^1
    for (; bits; bits >>= 1) {
        if (bits & 1u) p = crc_mul_c(sq, p);
        sq = crc_mul_c(sq, sq);
    }
    return p;
}
constexpr CrcTabs crc_make_tabs()
{
    CrcTabs t{};
    const uint32_t x32 = crc_xpow_c(32), xrow = crc_xpow_c(8ull * kCrcRow);
    for (uint32_t k = 0; k < 4; ++k)
        for (uint32_t b = 0; b < 256; ++b) {
            t.word[k][b] = crc_mul_c(b << (8u * k), x32);
            t.row[k][b] = crc_mul_c(b << (8u * k), xrow);
        }
    for (uint32_t i = 0; i < (uint32_t)kCrcThreads; ++i) t.tail[i] = crc_xpow_c(8ull * (kCrcRow - 16u - 16u * i));
    return t;
}
__device__ const CrcTabs g_crc_tabs = crc_make_tabs();

__global__ __launch_bounds__(kCrcThreads) void k_crc32_blocks(const uint8_t *__restrict__ data, const CrcBlock *__restrict__ blocks,
                                                              uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_word[4][256], s_row[4][256];
    __shared__ uint32_t s_part[kCrcThreads / kWave];
    const uint32_t tid = threadIdx.x;
    const CrcBlock b = blocks[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) s_word[k][tid] = g_crc_tabs.word[k][tid], s_row[k][tid] = g_crc_tabs.row[k][tid];
    __syncthreads();
    // the block lies against the end of kCrcBlock virtual bytes: virtual byte v is data[off + v - pad]
    const uint32_t pad = kCrcBlock - b.len;
    const uint8_t *base = data + b.off - pad;                 // (never dereferenced below virtual byte `pad`)
    auto piece = [&](uint32_t v) -> u32 {                     // virtual bytes [v, v + 16)
        u32 w = {0, 0, 0, 0};
        if (v >= pad) {
            __builtin_memcpy(&w, base + v, 16);
        } else if (v + 16u > pad) {                           // the block begins inside this piece (one thread of one row)
#pragma unroll
            for (uint32_t k = 0; k < 16u; ++k) w[k >> 2] |= (v + k >= pad ? (uint32_t)base[v + k] : 0u) << (8u * (k & 3u));
        }
        return w;
    };
    auto down = [&](uint32_t a) {                             // a * x^(8 * kCrcRow)
        return s_row[0][a & 255u] ^ s_row[1][(a >> 8) & 255u] ^ s_row[2][(a >> 16) & 255u] ^ s_row[3][a >> 24];
    };
    auto step = [&](uint32_t a) {                             // a * x^32
        return s_word[0][a & 255u] ^ s_word[1][(a >> 8) & 255u] ^ s_word[2][(a >> 16) & 255u] ^ s_word[3][a >> 24];
    };
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint32_t r = pad / kCrcRow;                               // rows in front of it hold nothing
    u32 w = r < kCrcRows ? piece(r * kCrcRow + tid * 16u) : u32{0, 0, 0, 0};
    for (; r < kCrcRows; ++r) {
        const u32 cur = w;
        if (r + 1u < kCrcRows) w = piece((r + 1u) * kCrcRow + tid * 16u);       // (in flight under this row's look-ups)
        a0 = down(a0) ^ cur[0], a1 = down(a1) ^ cur[1], a2 = down(a2) ^ cur[2], a3 = down(a3) ^ cur[3];
    }
    // the thread's four columns as one: ((((a0) x^32 ^ a1) x^32 ^ a2) x^32 ^ a3) x^32, then on to the row's end
    uint32_t c = step(step(step(step(a0) ^ a1) ^ a2) ^ a3);
    c = crc_mul(g_crc_tabs.tail[tid], c);
    // xor over the workgroup
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c ^= (uint32_t)__shfl_xor((int)c, o, kWave);
    if (lane_id() == 0) s_part[wave_id()] = c;
    __syncthreads();
    if (tid == 0) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < kCrcThreads / kWave; ++k) x ^= s_part[k];
        out[blockIdx.x] = x;                                  // L(block): the host adds K(len)
    }
}

uint32_t crc_block_bytes() { return kCrcBlock; }

hipError_t launch_crc32_blocks(const uint8_t *d_data, const void *d_blocks, uint32_t n_blocks, uint32_t *d_out, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_crc32_blocks, dim3(n_blocks), dim3(kCrcThreads), 0, st, d_data, (const CrcBlock *)d_blocks, d_out);
    return hipGetLastError();
}

// host side of the fold: CRC of a range from what the kernel left for its blocks (all of kCrcBlock bytes but the last).  The
// kernel's values are L(block), the register without its inversions; L(A || B) = L(A) x^(8 |B|) xor L(B), and the CRC-32 of the
// range is L xor K(total_len), K(n) = 0xffffffff x^(8 n) xor 0xffffffff (the CRC-32 of n zero bytes; 0 for n = 0).
uint32_t crc_fold_blocks(const uint32_t *crcs, uint64_t n_blocks, uint64_t total_len)
{
    static const CrcPow pw = crc_pow_table();
    // c * x^(8 kCrcBlock) by four look-ups (round 6: the bit-serial crc_mul per block was 24 ms of one host core per 16 GB batch of
    // the gzip route -- 242,000 blocks -- on the caller's thread, between two batches: profiles/r06/c2_gz_1e9.json's gap)
    struct BlockStep {
        uint32_t t[4][256];
        BlockStep(const CrcPow &p)
        {
            const uint32_t x_block = crc_xpow8(p, kCrcBlock);
            for (uint32_t k = 0; k < 4; ++k)
                for (uint32_t b = 0; b < 256; ++b) t[k][b] = crc_mul(b << (8u * k), x_block);
        }
    };
    static const BlockStep step(pw);
    uint32_t c = 0;
    uint64_t left = total_len;
    for (uint64_t k = 0; k < n_blocks; ++k) {
        const uint64_t len = left < kCrcBlock ? left : kCrcBlock;
        c = (len == kCrcBlock ? step.t[0][c & 255u] ^ step.t[1][(c >> 8) & 255u] ^ step.t[2][(c >> 16) & 255u] ^ step.t[3][c >> 24]
                              : crc_mul(crc_xpow8(pw, len), c)) ^ crcs[k];
        left -= len;
    }
    return total_len ? c ^ crc_mul(crc_xpow8(pw, total_len), 0xffffffffu) ^ 0xffffffffu : 0u;
}
uint32_t crc_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b)
{
    static const CrcPow pw = crc_pow_table();
    return crc_combine(pw, crc_a, crc_b, len_b);
}

}  // namespace hpn
