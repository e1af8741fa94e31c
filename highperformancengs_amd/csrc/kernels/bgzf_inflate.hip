// bgzf_inflate.hip -- DEFLATE (RFC 1951) decoding of independent BGZF blocks on gfx950.
//
// A BAM / bgzip file is a sequence of gzip members of at most 64 KiB each (SAM spec 4.1);
// the reference inflates them one after the other on the host (samtools-0.1.19 bgzf.c:214-307
// inflate_block / bgzf_read_block), which is where bam2depth and bam_sliding_count spend their
// time once the per-record work is on the GPU.  Blocks are independent, and a file has one per
// ~20 KB of compressed data, so here every block is decoded by its own wavefront:
//
//   * a block's headers and code tables are handled by the wave as a whole on wave-uniform state (bit buffer, positions:
//     scalar registers, scalar branches); the lanes cooperate where there is data parallelism: staging the compressed
//     bytes through an LDS ring (coalesced 16-byte loads) and filling the decode tables;
//   * the SYMBOLS are decoded 64 bit offsets at a time: lane k decodes the symbol that would start at bit k, the wave then
//     follows the true chain of symbol starts through the lanes' results and the lanes on it write their literals in one
//     store; a window with LZ77 matches is put together by the lanes, one output position each (a literal byte, or a copy of
//     position - distance out of memory or out of another lane: decode_symbols / assemble, inflate_core.hpp);
//   * the output goes straight to global memory, and a match reads its source back from there:
//     the 32 KiB history window does not have to live in LDS, which leaves two-level decode
//     tables (9-bit root for literal/length, 8-bit for distance) and a 1 KiB input ring =
//     7.1 KiB per wave = six of the CU's 1,280-byte LDS granules, 20 waves per CU, 5,120 blocks in flight on the chip (with the
//     window in LDS it was 3 waves per CU and 2.9x slower).  A match whose source overlaps bytes this
//     wave stored since its last wait first waits for those stores (workgroup-scope fence: the
//     CU's vector cache is coherent for its own waves, so that is a counter wait only).
//
// Bound: instruction issue -- 18 single-wave decoders share the CU's one scalar unit and its four vector units; with the
// per-symbol work in the lanes the two are about evenly loaded (SQ_ACTIVE_INST_SCA 0.54, VALU 0.52 of the busy CU cycles).
// Like the reference's reader, the CRC32 of the trailer is not checked; the ISIZE is (the block
// must produce exactly out_len bytes).
//
// Anything malformed sets the block's status word; the host then inflates that file with zlib.
#include "common.hpp"
#include "inflate_core.hpp"

namespace hpn {

struct BgzfBlock {  // = hpn_bgzf_block
    uint64_t in_off;
    uint32_t in_len, out_len;
    uint64_t out_off;
};

// where the decoded bytes go: straight to global memory; a match reads its source back from there
struct ByteSink {
    static constexpr bool kDry = false;
    uint8_t *out;
    uint32_t out_len, op, safe;   // op: bytes decoded; output bytes below `safe` are known to have reached memory
    __device__ __forceinline__ void pin_state() { op = uni(op), safe = uni(safe); }
    // entry e: [23:16] the literal, [31:24] the second one of a pair
    __device__ __forceinline__ void lits(bool mine, bool two, uint32_t at, uint32_t e)
    {
        if (mine) {
            if (two) {
                const uint16_t v = (uint16_t)(e >> 16);
                __builtin_memcpy(out + at, &v, 2);              // (one store; `at` may be odd)
            } else {
                out[at] = (uint8_t)(e >> 16);
            }
        }
    }
    __device__ __forceinline__ bool in_reach(uint32_t at, uint32_t dist) const { return dist <= at; }
    // output bytes at `from` (< base, the first position of the chunk being assembled) for the lanes that ask
    __device__ __forceinline__ uint32_t fetch(bool ask, int32_t from, uint32_t base, uint32_t val)
    {
        if (__builtin_amdgcn_ballot_w64(ask && (uint32_t)from >= safe)) {
            // the source reaches into bytes this wave stored a moment ago: wait for those stores
            // (workgroup scope = this CU's vector cache: a counter wait, no cache invalidate)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            safe = base;
        }
        return ask ? (uint32_t)out[(uint32_t)from] : val;
    }
    __device__ __forceinline__ void put(bool live, uint32_t at, uint32_t val)
    {
        if (live) out[at] = (uint8_t)val;
    }
};

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(HPN_INF_EU, 8))) void k_bgzf_inflate(const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks,
                                                        uint32_t n_blocks, uint8_t *__restrict__ outbuf,
                                                        uint32_t *__restrict__ status, uint32_t *__restrict__ ticket)
{
    __shared__ InfLds s;
    const int lane = lane_id();
    // blocks are TAKEN, not dealt: a wave that is done with its block takes the next one nobody has (one atomic per block).  Dealt
    // by stride, a launch ended with the wave whose blocks happened to be the slow ones (round 4: in the tools a launch is 3 - 6
    // blocks per wave; same-session A/B in profiles/r04/ab_inflate_ticket.txt)
#ifdef HPN_INF_STRIDE
    for (uint32_t bi = blockIdx.x; bi < n_blocks; bi += gridDim.x) {
#else
    for (uint32_t bi = blockIdx.x; bi < n_blocks;
         bi = gridDim.x + uni(lane == 0 ? atomicAdd(ticket, 1u) : 0u)) {
#endif
        const BgzfBlock blk = blocks[bi];
        const uint8_t *in = comp + blk.in_off;
        uint8_t *out = outbuf + blk.out_off;
        const uint32_t in_len = blk.in_len, out_len = blk.out_len;
        Bits b;
        stage(s, b, in, in_len);
        stage(s, b, in, in_len);
        ByteSink sink{out, out_len, 0u, 0u};
        uint32_t err = 0;
        bool last = false;
        while (!last && !err) {
            pin(b), pin(err), sink.pin_state();
            refill(s, b, in, in_len);
            if (b.in_pos - (b.bc >> 3) > in_len) {  // ran past the payload (the ring would only repeat itself)
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
                uint32_t &op = sink.op;
                if ((len ^ nlen) != 0xffffu || op + len > out_len) {
                    err = 1;
                    break;
                }
                // the stored bytes start with what is still in the bit buffer; copy them from
                // global memory in pieces the window can hold, writing each piece back at once
                const uint32_t src = b.in_pos - (b.bc >> 3);  // compressed offset of the first stored byte
                if (src + len > in_len) {
                    err = 2;
                    break;
                }
                for (uint32_t i = (uint32_t)lane; i < len; i += kWave) out[op + i] = in[src + i];
                op += len;
                // restart the bit reader behind the stored bytes
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
            // ---- symbols of this block: 64 bit offsets at a time (decode_symbols, inflate_core.hpp) --------
            // (every symbol emits at least one byte -- bounded by out_len -- or ends the block, so this ends on any
            // input; running past the payload is caught per block above)
            Pos p = pos_of(b);
            if (!decode_symbols(s, b, p, in, in_len, sink, err)) break;
            seek(s, b, p, in, in_len);
        }
        // the FINAL block must end inside the payload too (bytes behind in_len are read as zeros, and seven zero bits are the
        // end-of-block code of a fixed-Huffman block: a payload cut short inside that code decoded to the stated length --
        // zlib calls such a stream unfinished; scripts/soak_inflate_damaged.py, round 431 of 1,000)
        if (!err && b.in_pos - (b.bc >> 3) > in_len) err = 17;
        if (!err && sink.op != out_len) err = 16;
        if (lane == 0) status[bi] = err;
    }
}

hipError_t launch_bgzf_inflate(const uint8_t *d_comp, const void *d_blocks, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_status,
                               uint32_t *d_ticket, int n_cu, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    const uint32_t cap = (uint32_t)n_cu * kInflateWavesPerCu;  // single-wave workgroups: 24 per CU (inflate_core.hpp)
    hipError_t e = hipMemsetAsync(d_ticket, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(n_blocks < cap ? n_blocks : cap), dim3(kWave), 0, st, d_comp, (const BgzfBlock *)d_blocks,
                       n_blocks, d_out, d_status, d_ticket);
    return hipGetLastError();
}

}  // namespace hpn
