// bam_raw.hip -- BAM records straight from inflated BGZF blocks on the device.
//
// The reference walks a BAM with bam_read1 (samtools-0.1.19 bam.c:191): block_size, 32 bytes of
// core fields, name, CIGAR, packed sequence, qualities -- one record after the other on the
// host, through bgzf_read, which does not care where blocks end (bgzf.c:342).  With the blocks inflated on the GPU
// (bgzf_inflate.hip) the record chain is walked there too.  A chain is serial; what makes it parallel is knowing where the
// first record of every block starts:
//   * samtools never lets a record straddle a BGZF block (bam_write1 calls bgzf_flush_try(4 + block_len), bam.c:238; the
//     header is followed by a flush): every block starts at a record, offset 0;
//   * htsjdk (Picard, GATK -- most production BAMs) packs records across blocks.  Round 4: the start is GUESSED per block
//     (k_raw_starts: the lowest offset at which a plausible chain of records begins -- sizes that fit, a name that ends in NUL,
//     reference ids and positions in range) and then PROVEN for the whole call (k_raw_scan): the chain, followed block by block
//     from the call's first record -- whose place is known --, must arrive exactly at each block's guess; where it does not,
//     the block is walked again from where the chain does arrive.  The guesses make the walk parallel, never the result.
// The inflated bytes of a batch are one contiguous stream, so a record that runs into the next block is whole in memory; the
// record that runs past the batch's END is not: its bytes are reported (hpn_raw_info.tail_bytes) and the host copies them in
// front of the next batch's stream (host/bam_gpu.hpp), which therefore starts at a record again.
//
//   k_raw_starts   per block: where its first record starts (wave per block; block 0: given)
//   k_raw_count    per block: number of records that START in it, smallest / largest refID, where its chain leaves it
//   k_raw_scan     exclusive scan of the per-block counts + the proof (one workgroup; a batch has ~10^4 blocks)
//   k_raw_index    per block: the offsets k_raw_count wrote down, laid one after the other -> rec_off[]
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

constexpr uint32_t kNoStart = 0xffffffffu;

// Could a record start at p?  `room` = bytes of the batch's stream from p on.  0 = no; 1 = yes, *step = bytes to the next record;
// 2 = cannot tell (the stream ends inside its fixed part or its name: the batch's tail).
// bam_read1 trusts the fields (bam.c:191); the later kernels read name, CIGAR and sequence in place, so a record that lies about
// them must not get past here.
__device__ __forceinline__ int record_at(const uint8_t *p, uint64_t room, uint32_t *step)
{
    if (room < 36u) return 2;
    const uint32_t bs = ld32(p);
    if (bs < 32u || bs > (1u << 28)) return 0;
    const int32_t tid = (int32_t)ld32(p + 4), pos = (int32_t)ld32(p + 8), mtid = (int32_t)ld32(p + 24), mpos = (int32_t)ld32(p + 28);
    const uint32_t l_name = p[12], n_cigar = ld16(p + 16), l_seq = ld32(p + 20);
    if (tid < -1 || pos < -1 || mtid < -1 || mpos < -1 || l_name == 0u || l_seq > 0x7fffffffu) return 0;
    if (32ull + l_name + 4ull * n_cigar + (((uint64_t)l_seq + 1u) >> 1) + (uint64_t)l_seq > (uint64_t)bs) return 0;
    if (room < 36u + l_name) return 2;
    if (p[35u + l_name] != 0u) return 0;                   // read_name is NUL-terminated
    *step = 4u + bs;
    return 1;
}

// record_at for a GUESS: also what real records look like beyond what makes them safe to read -- a read name of printable
// characters (the SAM specification's [!-?A-~]{1,254}) and a block_size below 16 MB.  Without these a run of qualities read as a
// header is plausible once per ~100 offsets, and where its block_size (33 .. 268 million) lands on a true record start of a
// 700 MB stream, four records in a row follow: 2 in 1,000 blocks of a packed file guessed wrong, each walked again by one
// lane, 7 ms per launch (scripts/diag_packed.py, HPN_TIMING=2).  A true record that fails here merely gets no guess.
__device__ __forceinline__ int likely_record_at(const uint8_t *p, uint64_t room, uint32_t *step)
{
    const int r = record_at(p, room, step);
    if (r != 1) return r;
    if (*step > (1u << 24)) return 0;
    const uint32_t l_name = p[12];
    for (uint32_t k = 0; k + 1u < l_name; ++k)
        if ((uint32_t)p[36u + k] - 33u > 93u) return 0;       // 33 .. 126
    return 1;
}

// Where does the first record of block b start?  One wave per block (b >= 1; block 0's start is given): lane k tries offset
// base + k and follows the chain four records (one well-formed header is no rarity in BAM bytes -- small integers everywhere,
// and a run of qualities read as block_size is a plausible 10^8; four in a row are).  The lowest offset with four records
// IN the stream is taken.  Only a block that has none (the last few records of a call) falls back on the lowest offset whose
// chain holds as far as the stream goes: such a chain proves little -- a plausible header with a large block_size leaves a
// 350 MB stream in one hop, and on a packed file that is what most blocks would find first (measured: nearly every guess wrong,
// 10 s of serial re-walks for a 10 GB file).  Guesses all: k_raw_scan proves each one or walks the block itself.
__global__ __launch_bounds__(kWave) void k_raw_starts(const uint8_t *__restrict__ raw, const RawBlock *__restrict__ blocks, uint32_t n_blocks,
                                                      const uint32_t *__restrict__ status, uint32_t *__restrict__ starts)
{
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const RawBlock blk = blocks[b], last = blocks[n_blocks - 1];
    const uint64_t stream_len = last.out_off + last.out_len;
    const int lane = lane_id();
    if (b == 0 || status[b] || blk.out_len == 0) {           // (block 0 is walked from first_abs)
        if (lane == 0) starts[b] = kNoStart;
        return;
    }
    uint32_t found = kNoStart, weak = kNoStart;
    for (uint32_t base = 0; base < blk.out_len && found == kNoStart; base += kWave) {
        const uint32_t s0 = base + (uint32_t)lane;
        bool ok = s0 < blk.out_len, whole = true;      // whole: all four records lie in the stream
        uint64_t at = blk.out_off + s0;
        for (int hop = 0; ok && hop < 4; ++hop) {      // four records in a row, into the blocks behind if need be (one stream)
            uint32_t step = 0;
            const int r = at < stream_len ? likely_record_at(raw + at, stream_len - at, &step) : 2;
            if (r == 0 || (r == 2 && hop == 0)) ok = false;      // (a start of which not even the fixed part is there: k_raw_scan's)
            if (r != 1) {
                whole = false;                            // the chain has left the stream (its last record is the unfinished one)
                break;
            }
            at += step;
        }
        const u64 strong = __ballot(ok && whole), any = __ballot(ok);
        if (strong) found = base + (uint32_t)__builtin_ctzll(strong);
        else if (any && weak == kNoStart) weak = base + (uint32_t)__builtin_ctzll(any);
    }
    if (lane == 0) starts[b] = found != kNoStart ? found : weak;
}

constexpr uint32_t kBroken = 0x80000000u;      // counts[b]: the block's walk met an impossible record
constexpr u64 kTailBit = 1ull << 63;           // exits[b]: the walk ended at an unfinished record, which starts at exits[b] & ~kTailBit

// One block's records from `at` on: how many start in it, their smallest / largest refID, where the chain leaves the block.
struct RawWalk {
    uint32_t n;        // | kBroken
    int32_t lo, hi;
    u64 exit;          // | kTailBit
};
// (list: where the records' stream offsets go, one after the other -- see k_raw_count; may be null)
__device__ __forceinline__ RawWalk walk_block(const uint8_t *__restrict__ raw, uint64_t at, uint64_t end, uint64_t stream_len, u64 *__restrict__ list = nullptr,
                                              uint64_t list_room = 0)
{
    RawWalk w = {0u, INT32_MAX, INT32_MIN, 0ull};
    while (at < end) {
        uint32_t step = 0;
        const int r = record_at(raw + at, stream_len - at, &step);
        if (r == 0) {
            w.n |= kBroken;
            break;
        }
        if (r == 2 || at + step > stream_len) {          // the record is not whole in this call: it starts the next one
            at |= kTailBit;
            break;
        }
        const int32_t tid = (int32_t)ld32(raw + at + 4u);
        w.lo = tid < w.lo ? tid : w.lo, w.hi = tid > w.hi ? tid : w.hi;
        if (list) {
            if (w.n >= list_room) {                      // (cannot happen for blocks of BGZF's 64 KiB: the list is sized for them)
                w.n |= kBroken;
                break;
            }
            list[w.n] = at;
        }
        at += step;
        ++w.n;
    }
    w.exit = at;
    return w;
}

// ---- the walk, one round trip and three loads per record (round 6) --------------------------------------------------------------
// walk_block above waits twice per record: for the fixed part (block_size tells where the next record is), then for the byte
// that must be the name's NUL -- and k_raw_index walked every chain a second time to write the offsets down: 0.62 + 0.27 ms per
// 1.2 GB launch (4.4 M records, ~230 to a block), the largest kernels of the BAM tools behind the inflater
// (profiles/r05/kernel_stats_bam2depth_final.csv).  What a record costs here turned out to be its LOAD INSTRUCTIONS, not its round
// trips: a lane per block means 64 lanes reading 64 different lines, ~0.45 us per such instruction and wave whatever is asked
// for.  So: the record's fixed part is two 16-byte loads (round 5: eight narrow ones), the next record's fixed part and this
// record's NUL byte are asked for together (one wait per record), and the offsets are written down as the records are counted
// (list: block b's at raw_list_base(b), one after the other; two records never share a slot because a record is at least 36
// bytes), so the index is a copy, not a second walk.  Measured on the way and dropped, each slower for its extra loads
// (docs/kernels/bam_raw_index.md): betting on records of one size (the record after next asked for early), the first 80 bytes of
// three records in flight, three records to a round trip.  Same results as walk_block, record for record (tests/test_bam_raw_gpu.py).
struct RawHead {           // the first 32 bytes of a record: block_size | refID pos bin_mq_nl | flag_nc l_seq next_refID next_pos
    u32 a, b;
};
__device__ __forceinline__ RawHead head_at(const uint8_t *__restrict__ raw, uint64_t at, uint64_t stream_len)
{
    // (where the fixed part does not fit the stream's first bytes are loaded instead and never looked at: head_check sees the room)
    const uint8_t *p = raw + (at + 36u <= stream_len ? at : 0ull);
    RawHead h;
    __builtin_memcpy(&h.a, p, 16);
    __builtin_memcpy(&h.b, p + 16, 16);
    return h;
}
// record_at on a fixed part that is in registers, up to the name's NUL: 0 = no record, 2 = cannot tell (the stream ends inside it),
// 1 = a record if the byte at *nul_at is 0, *step bytes long
__device__ __forceinline__ int head_check(const RawHead &h, uint64_t room, uint32_t *step, uint32_t *nul_at)
{
    if (room < 36u) return 2;
    const uint32_t bs = h.a[0];
    if (bs < 32u || bs > (1u << 28)) return 0;
    const int32_t tid = (int32_t)h.a[1], pos = (int32_t)h.a[2], mtid = (int32_t)h.b[2], mpos = (int32_t)h.b[3];
    const uint32_t l_name = h.a[3] & 255u, n_cigar = h.b[0] & 0xffffu, l_seq = h.b[1];
    if (tid < -1 || pos < -1 || mtid < -1 || mpos < -1 || l_name == 0u || l_seq > 0x7fffffffu) return 0;
    if (32ull + l_name + 4ull * n_cigar + (((uint64_t)l_seq + 1u) >> 1) + (uint64_t)l_seq > (uint64_t)bs) return 0;
    if (room < 36u + l_name) return 2;
    *nul_at = 35u + l_name;
    *step = 4u + bs;
    return 1;
}
__device__ __forceinline__ RawWalk walk_block_ahead(const uint8_t *__restrict__ raw, uint64_t at, uint64_t end, uint64_t stream_len, u64 *__restrict__ list,
                                                    uint64_t list_room)
{
    RawWalk w = {0u, INT32_MAX, INT32_MIN, 0ull};
    if (at >= end) {
        w.exit = at;
        return w;
    }
    RawHead cur = head_at(raw, at, stream_len);
    while (at < end) {
        uint32_t step = 0, nul_at = 0;
        const int r = head_check(cur, stream_len - at, &step, &nul_at);
        if (r == 0) {
            w.n |= kBroken;
            break;
        }
        if (r == 2) {
            at |= kTailBit;
            break;
        }
        const uint64_t next = at + step;
        uint32_t nul;                                                              // (a whole word, asked for together with ...
        __builtin_memcpy(&nul, raw + at + nul_at, 4);
        const RawHead nx = head_at(raw, next < end ? next : 0ull, stream_len);    // ... the next record's fixed part: one wait)
        // (both asked for before either is looked at: without the two lines below the compiler tests the NUL first and asks for
        // the fixed part behind that test -- two round trips per record again)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::"v"(nul), "v"(nx.a[0]), "v"(nx.b[0]));
        if ((nul & 255u) != 0u || w.n >= list_room) {                              // read_name is NUL-terminated (record_at's last test)
            w.n |= kBroken;
            break;
        }
        if (next > stream_len) {                                                   // the record is not whole in this call: it starts the next one
            at |= kTailBit;
            break;
        }
        const int32_t tid = (int32_t)cur.a[1];
        w.lo = tid < w.lo ? tid : w.lo, w.hi = tid > w.hi ? tid : w.hi;
        list[w.n] = at;
        ++w.n;
        at = next, cur = nx;
    }
    w.exit = at;
    return w;
}

// where block b's records go in the list: a record takes at least 36 bytes of the stream, so the records that start in
// [0, out_off) are at most out_off / 36 (+ one that a block's chain may count across its start, + slack): slots never meet
__host__ __device__ __forceinline__ uint64_t raw_list_base(uint64_t out_off, uint32_t b) { return b ? out_off / 36u + 2ull * b : 0ull; }

// info words: [0] flags (1 = an impossible record on the chain, 2 = a block failed to inflate, 4 = records run across block ends
// (packed BAM)), [1] min refID, [2] max refID (as int32), [3] unused; u64 at info + 4: stream offset of the call's unfinished
// last record (~0: the stream ends on a record).  k_raw_count only sets flag 2; the rest is k_raw_scan's, from the proven chain.
__global__ __launch_bounds__(kRawThreads) void k_raw_count(const uint8_t *__restrict__ raw, const RawBlock *__restrict__ blocks,
                                                           uint32_t n_blocks, uint64_t first_abs,
                                                           const uint32_t *__restrict__ status, const uint32_t *__restrict__ starts,
                                                           uint32_t *__restrict__ counts, u64 *__restrict__ exits, int32_t *__restrict__ lo,
                                                           int32_t *__restrict__ hi, int32_t *__restrict__ info, u64 *__restrict__ list,
                                                           uint64_t list_words)
{
    const uint32_t b = blockIdx.x * kRawThreads + threadIdx.x;
    if (b >= n_blocks) return;
    const RawBlock blk = blocks[b], last = blocks[n_blocks - 1];
    const uint64_t stream_len = last.out_off + last.out_len, end = blk.out_off + blk.out_len;
    if (status[b]) {
        atomicOr((uint32_t *)&info[0], 2u);
        counts[b] = kBroken, exits[b] = end, lo[b] = INT32_MAX, hi[b] = INT32_MIN;
        return;
    }
    // a block without a start (its predecessor's record runs over it whole, or it lies behind the call's last whole record)
    // counts nothing; whether that is right is k_raw_scan's to say
    const uint64_t at = b == 0 ? first_abs : starts[b] == kNoStart ? end : blk.out_off + starts[b];
    const uint64_t lb = raw_list_base(blk.out_off, b);
    const RawWalk w = walk_block_ahead(raw, at, end, stream_len, list + lb, lb < list_words ? list_words - lb : 0ull);
    counts[b] = w.n, exits[b] = w.exit, lo[b] = w.lo, hi[b] = w.hi;
}

// The proof, the exclusive scan of the per-block counts, the refID range.  One workgroup; a call has ~2 x 10^4 blocks at most.
// The proof: walking the blocks in order, the chain enters block i at `from`.  Either it has already run past the block (a long
// record, or the call's unfinished tail): then no record starts in the block, whatever its guess found; or the block's guessed
// start IS `from` and the chain goes on where the block's own walk left it; or the guess was wrong (or there was none) and this
// lane walks the block from `from` itself.  By induction from first_abs every counted record is a true one, and the unfinished
// tail is the chain's, not a guess's.  In chunks of 1,024 blocks through LDS, one lane walking; samtools' blocks and all but a
// few of htsjdk's (near the stream's end, where chains are short) are taken on their guess.
__global__ __launch_bounds__(1024) void k_raw_scan(const uint8_t *__restrict__ raw, uint32_t *__restrict__ counts, uint32_t n_blocks,
                                                   u64 *__restrict__ bases, const RawBlock *__restrict__ blocks, uint64_t first_abs,
                                                   uint32_t *__restrict__ starts, const u64 *__restrict__ exits,
                                                   int32_t *__restrict__ lo, int32_t *__restrict__ hi, int32_t *__restrict__ info,
                                                   u64 *__restrict__ list, uint64_t list_words)
{
    __shared__ u64 s_wave[16];
    __shared__ u64 s_carry, s_from, s_tail;
    __shared__ uint32_t s_flags;
    __shared__ u64 s_off[1024], s_end[1024], s_exit[1024];
    __shared__ uint32_t s_cnt[1024], s_start[1024];
    __shared__ int32_t s_lo[1024], s_hi[1024];
    const RawBlock last = blocks[n_blocks - 1];
    const u64 stream_len = last.out_off + last.out_len;
    __shared__ uint32_t s_walked, s_rewalked;                 // chunks the lane walked, blocks it walked again (info[3]: diagnostics)
    if (threadIdx.x == 0) s_carry = 0, s_from = first_abs, s_tail = ~0ull, s_flags = 0, s_walked = 0, s_rewalked = 0;
    int32_t my_lo = INT32_MAX, my_hi = INT32_MIN;
    for (uint32_t i0 = 0; i0 < n_blocks; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        __syncthreads();
        if (i < n_blocks) {
            const RawBlock blk = blocks[i];
            s_off[threadIdx.x] = blk.out_off, s_end[threadIdx.x] = blk.out_off + blk.out_len;
            s_exit[threadIdx.x] = exits[i], s_cnt[threadIdx.x] = counts[i], s_start[threadIdx.x] = starts[i];
            s_lo[threadIdx.x] = lo[i], s_hi[threadIdx.x] = hi[i];
        } else {
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        // the common chunk needs no walk: every block's chain ends inside the next block, exactly at that block's guess
        bool plain = true, crosses = false;
        if (i < n_blocks) {
            const u64 arrives = threadIdx.x == 0 ? s_from : s_exit[threadIdx.x - 1];     // (an exit with kTailBit equals no guess)
            const u64 own = i == 0 ? first_abs : s_start[threadIdx.x] == kNoStart ? ~0ull : s_off[threadIdx.x] + s_start[threadIdx.x];
            plain = own == arrives && arrives < s_end[threadIdx.x] && !(s_cnt[threadIdx.x] & kBroken) && !(s_exit[threadIdx.x] & kTailBit);
            crosses = s_exit[threadIdx.x] > s_end[threadIdx.x] && i + 1 < n_blocks;
        }
        const bool all_plain = __syncthreads_and(plain), any_cross = __syncthreads_or(crosses);
        if (all_plain) {
            if (i == (n_blocks - i0 < 1024u ? n_blocks - 1 : i0 + 1023u)) {
                s_from = s_exit[threadIdx.x];
                if (any_cross) s_flags |= 4u;
            }
        } else if (threadIdx.x == 0) {
            ++s_walked;
            u64 from = s_from, tail = s_tail;
            uint32_t flags = s_flags;
            const uint32_t m = n_blocks - i0 < 1024u ? n_blocks - i0 : 1024u;
            for (uint32_t k = 0; k < m; ++k) {
                const u64 end = s_end[k];
                if (from >= end) {                                  // the chain never sets foot in this block
                    s_cnt[k] = 0, s_start[k] = kNoStart;
                    continue;
                }
                const u64 own = i0 + k == 0 ? first_abs : s_start[k] == kNoStart ? ~0ull : s_off[k] + s_start[k];
                if (own != from) {                                  // no guess, or not where the chain arrives: walked here
                    ++s_rewalked;
                    const uint64_t lb = raw_list_base(s_off[k], i0 + k);
                    const RawWalk w = walk_block(raw, from, end, stream_len, list + lb, lb < list_words ? list_words - lb : 0ull);   // (its list again, from where the chain arrives)
                    s_cnt[k] = w.n, s_exit[k] = w.exit, s_lo[k] = w.lo, s_hi[k] = w.hi;
                    s_start[k] = (uint32_t)(from - s_off[k]);
                }
                if (s_cnt[k] & kBroken) flags |= 1u, s_cnt[k] &= ~kBroken;
                if (s_exit[k] & kTailBit) tail = s_exit[k] & ~kTailBit, from = stream_len;
                else from = s_exit[k];
                if (flags & 1u) from = stream_len;              // (nothing behind an impossible record is looked at)
                if (from > end && i0 + k + 1 < n_blocks) flags |= 4u;
            }
            s_from = from, s_tail = tail, s_flags = flags;
        }
        __syncthreads();
        const u64 v = s_cnt[threadIdx.x];
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
        if (i < n_blocks) {
            bases[i] = before + inc - v, counts[i] = (uint32_t)v, starts[i] = s_start[threadIdx.x];
            // (the proven chain's refID range per block, too: hosts pick a target's records by it -- hpn_depth_add_raw_dev)
            lo[i] = v ? s_lo[threadIdx.x] : INT32_MAX, hi[i] = v ? s_hi[threadIdx.x] : INT32_MIN;
            if (v) my_lo = s_lo[threadIdx.x] < my_lo ? s_lo[threadIdx.x] : my_lo, my_hi = s_hi[threadIdx.x] > my_hi ? s_hi[threadIdx.x] : my_hi;
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
    }
    __syncthreads();
    if (my_lo <= my_hi) atomicMin(&info[1], my_lo), atomicMax(&info[2], my_hi);
    if (threadIdx.x == 0) {
        bases[n_blocks] = s_carry;
        uint32_t flags = s_flags;
        if (s_from != stream_len) flags |= 1u;
        if (flags) atomicOr((uint32_t *)&info[0], flags);
        *(u64 *)(info + 4) = s_tail;
        info[3] = (int32_t)(s_walked << 20 | (s_rewalked & 0xfffffu));
    }
}

// rec_off[]: the blocks' lists one after the other (bases[] from the scan).  A wave per block, coalesced both ways.
__global__ __launch_bounds__(kRawThreads) void k_raw_index(const RawBlock *__restrict__ blocks, uint32_t n_blocks,
                                                           const uint32_t *__restrict__ counts, const u64 *__restrict__ bases,
                                                           const u64 *__restrict__ list, uint64_t *__restrict__ rec_off)
{
    const uint32_t b = blockIdx.x * (kRawThreads / kWave) + (uint32_t)wave_id();
    if (b >= n_blocks) return;
    const u64 *from = list + raw_list_base(blocks[b].out_off, b);
    uint64_t *to = rec_off + bases[b];
    for (uint32_t k = (uint32_t)lane_id(), n = counts[b]; k < n; k += kWave) to[k] = from[k];
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

size_t raw_list_words(uint64_t stream_len, uint32_t n_blocks) { return (size_t)(stream_len / 36u + 2ull * n_blocks + 4096u); }

hipError_t launch_raw_count(const uint8_t *raw, const void *blocks, uint32_t n_blocks, uint64_t first_abs, const uint32_t *status,
                            uint32_t *starts, uint32_t *counts, u64 *exits, int32_t *lo, int32_t *hi, u64 *bases, int32_t *info, u64 *list,
                            uint64_t list_words, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_raw_starts, dim3(n_blocks), dim3(kWave), 0, st, raw, (const RawBlock *)blocks, n_blocks, status, starts);
    hipLaunchKernelGGL(k_raw_count, dim3((n_blocks + kRawThreads - 1) / kRawThreads), dim3(kRawThreads), 0, st, raw,
                       (const RawBlock *)blocks, n_blocks, first_abs, status, starts, counts, exits, lo, hi, info, list, list_words);
    hipLaunchKernelGGL(k_raw_scan, dim3(1), dim3(1024), 0, st, raw, counts, n_blocks, bases, (const RawBlock *)blocks, first_abs, starts,
                       exits, lo, hi, info, list, list_words);
    return hipGetLastError();
}

hipError_t launch_raw_index(const void *blocks, uint32_t n_blocks, const uint32_t *counts, const u64 *bases, const u64 *list,
                            uint64_t *rec_off, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    constexpr uint32_t per = kRawThreads / kWave;
    hipLaunchKernelGGL(k_raw_index, dim3((n_blocks + per - 1) / per), dim3(kRawThreads), 0, st, (const RawBlock *)blocks, n_blocks, counts,
                       bases, list, rec_off);
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
