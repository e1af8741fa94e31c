// pgz_reader.hpp -- parallel inflate of ONE gzip member (the usual .fastq.gz), two passes.
//
// The reference reads .fastq.gz through gzread on the calling thread (fastq_count.c:112-118 via
// gzgets): one deflate stream, every byte of which may refer to the 32 KiB before it, so it cannot
// simply be cut into pieces.  What can be done (the published "pugz" idea, restated here):
//
//   1. cut the COMPRESSED file into chunks; in each chunk find the first deflate block that starts
//      there -- try every bit position, keep the one where a dynamic-Huffman header parses, the
//      block and the beginning of the next one decode, and everything they produce is text;
//   2. inflate every chunk from its block start with the 32 KiB of history UNKNOWN: the output is
//      16-bit symbols, a byte value or "whatever byte history[i] turns out to be" (256 + i); match
//      copies move the placeholders along like any other symbol (fast_inflate.hpp, T = uint16_t);
//   3. walk the chunks in order: the decoded tail of one chunk, resolved through its own history,
//      IS the next chunk's history (32 K table look-ups per chunk, serial but tiny);
//   4. translate every chunk's symbols to bytes through its history -- in parallel again.
//
// Nothing is taken on faith: the first chunk starts at the member's first block (known), and a
// chunk is accepted only if the previous one, decoded from an accepted start, arrives exactly at
// its start bit on a block boundary.  A start that is not reached that way was a false positive;
// the stretch is decoded again from where the previous chunk did end.  So the bytes delivered are
// those of a serial inflate by construction; the CRC-32 of the trailer is checked as well and
// reported (crc_failed) the way gzread reports it, after the data.  Anything the quick decoder
// rejects (damaged or truncated streams, more than 24x expansion) goes back to zlib: gzdopen
// re-reads the file, the bytes already delivered are skipped, and gzread's output and error
// take over.  Further members after the first are followed (their first stretch serially).
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include "cpus.hpp"
#include "fast_inflate.hpp"

namespace hpn {

// End of the gzip member header that starts at p (RFC 1952), or nullptr.
inline const uint8_t *gzip_header_end(const uint8_t *p, const uint8_t *lim)
{
    if (lim - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return nullptr;
    const uint32_t flg = p[3];
    p += 10;
    if (flg & 4) {  // FEXTRA
        if (lim - p < 2) return nullptr;
        const size_t xl = p[0] | p[1] << 8;
        if ((size_t)(lim - p) < 2 + xl) return nullptr;
        p += 2 + xl;
    }
    for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
        if (flg & f) {
            const void *z = memchr(p, 0, (size_t)(lim - p));
            if (!z) return nullptr;
            p = (const uint8_t *)z + 1;
        }
    if (flg & 2) {  // FHCRC
        if (lim - p < 2) return nullptr;
        p += 2;
    }
    return p;
}

// ---- where does a deflate block start? (shared with the device path, host/gz_gpu.hpp) ---------------
constexpr size_t kGzFindHist = 32768, kGzFindScratch = (size_t)1 << 20;  // history placeholders; two candidate blocks of symbols
constexpr uint64_t kGzNone = ~(uint64_t)0;

inline bool gz_texty(const uint16_t *s, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const uint16_t v = s[i];
        if (v < 256 && !(v >= 32 && v < 127) && v != '\n' && v != '\r' && v != '\t') return false;
    }
    return true;
}
inline uint16_t *gz_find_scratch()  // per thread: kGzFindScratch symbols behind the placeholder history
{
    static thread_local std::vector<uint16_t> store;
    if (store.empty()) {
        store.resize(kGzFindHist + kGzFindScratch + FastInflateT<uint16_t>::kOvershoot);
        for (size_t i = 0; i < kGzFindHist; ++i) store[i] = (uint16_t)(256 + i);
    }
    return store.data() + kGzFindHist;
}
// First bit position in [lo, hi) of data[0..size) where a dynamic-Huffman block and the start of the next decode to text; kGzNone if
// there is none.  Per candidate position, cheapest test first: BFINAL = 0 / BTYPE = 2 / HLIT <= 29 / HDIST <= 29
// (RFC 1951 3.2.7); then the code-length code (HCLEN + 4 three-bit lengths) must be a complete prefix code or zlib's
// inftrees rejects it -- Kraft sum = 1, ~20 operations that turn away ~99 % of what got this far; only then tables are
// built and blocks decoded.
inline uint64_t gz_find_block_start(const uint8_t *data, uint64_t size, uint64_t lo, uint64_t hi, uint16_t *scratch, size_t cap)
{
    const uint8_t *lim = data + size;
    if (size < 8) return kGzNone;
    if (hi > (size - 8) * 8) hi = (size - 8) * 8;  // the trailer is not deflate data
    FastInflateT<uint16_t> fi;
    for (uint64_t byte = lo >> 3; byte * 8 < hi; ++byte) {
        const uint8_t *b = data + byte;
        uint64_t w0 = 0, w1 = 0;
        if (lim - b >= 16) {
            memcpy(&w0, b, 8), memcpy(&w1, b + 8, 8);
        } else {
            memcpy(&w0, b, lim - b >= 8 ? 8 : (size_t)(lim - b));
            if (lim - b > 8) memcpy(&w1, b + 8, (size_t)(lim - b - 8));
        }
        for (uint32_t sh = 0; sh < 8; ++sh) {
            const uint64_t v0 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
            if ((v0 & 7) != 4 || ((v0 >> 3) & 31) > 29 || ((v0 >> 8) & 31) > 29) continue;
            const uint64_t p = byte * 8 + sh;
            if (p < lo || p >= hi) continue;
            const uint32_t hclen = (uint32_t)(v0 >> 13 & 15) + 4;
            uint64_t x = (v0 >> 17) | ((w1 >> sh) << 47);
            uint32_t kraft = 0;
            for (uint32_t i = 0; i < hclen; ++i, x >>= 3) kraft += (128u >> (x & 7)) & 127u;  // length 0: unused
            if (kraft != 128) continue;
            fi.begin(b, lim, sh, true);
            uint16_t *out = scratch;
            int r = fi.run(out, scratch + cap, scratch - kGzFindHist);
            if (r != FastInflateT<uint16_t>::kBlockEnd || out == scratch || !gz_texty(scratch, (size_t)(out - scratch))) continue;
            // ... and the next block must at least begin like one (header parses, tables build, the first symbols decode
            // to text).  Decoding it to its end as well would double the cost of a search for nothing: whoever uses the
            // start proves it anyway (a stretch has to arrive exactly there from a proven start).
            uint16_t *mid = out;
            uint16_t *const stop = out + 256 < scratch + cap ? out + 256 : scratch + cap;
            r = fi.run(out, stop, scratch - kGzFindHist);
            if (r == FastInflateT<uint16_t>::kError || !gz_texty(mid, (size_t)((out < stop ? out : stop) - mid))) continue;
            return p;
        }
    }
    return kGzNone;
}

// First bit position in [lo, hi) that is the first block of a gzip MEMBER: a header (1f 8b 08, reserved flag bits clear,
// optional fields in range) whose first block decodes to text.  For files of many small members -- every block of such a
// file is a final block, which gz_find_block_start does not propose -- on the device route, whose decoder continues from one
// member into the next (kernels/gz_inflate.hip).  Like every proposed start it is proven only by the stretch before it
// arriving there.
inline uint64_t gz_find_member_start(const uint8_t *data, uint64_t size, uint64_t lo, uint64_t hi, uint16_t *scratch, size_t cap)
{
    const uint8_t *lim = data + size;
    if (size < 18) return kGzNone;
    FastInflateT<uint16_t> fi;
    for (uint64_t byte = lo >> 3; byte * 8 < hi && byte + 18 <= size; ++byte) {
        const uint8_t *h = (const uint8_t *)memchr(data + byte, 0x1f, (size_t)((hi + 7) / 8 - byte < size - byte ? (hi + 7) / 8 - byte : size - byte));
        if (!h) break;
        byte = (uint64_t)(h - data);
        const uint8_t *body = gzip_header_end(h, lim);
        if (!body) continue;
        const uint64_t p = (uint64_t)(body - data) * 8;
        if (p < lo || p >= hi || ((body[0] >> 1) & 3) == 3) continue;
        fi.begin(body, lim, 0, true);
        uint16_t *out = scratch;
        const int r = fi.run(out, scratch + cap, scratch - kGzFindHist);
        if ((r != FastInflateT<uint16_t>::kBlockEnd && r != FastInflateT<uint16_t>::kDone) || out == scratch || !gz_texty(scratch, (size_t)(out - scratch))) continue;
        return p;
    }
    return kGzNone;
}

class PgzReader {
public:
    // threads <= 0: HPN_GZ_THREADS, else min(16, usable CPUs).  chunk_bytes 0: HPN_PGZ_CHUNK, else 2 MiB.
    bool open(const char *path, int threads = 0, size_t chunk_bytes = 0)
    {
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) return false;
        struct stat sb;
        if (fstat(fd_, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) return fail_open();
        size_ = (uint64_t)sb.st_size;
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return fail_open();
        data_ = (const uint8_t *)m;
        madvise(m, size_, MADV_SEQUENTIAL);
        const uint8_t *body = gzip_header_end(data_, data_ + size_);
        if (!body) return fail_open();
        if (threads <= 0) {
            const char *e = getenv("HPN_GZ_THREADS");
            long n = e ? atol(e) : usable_cpus();
            threads = (int)(n < 1 ? 1 : n > 16 ? 16 : n);
        }
        if (!chunk_bytes) {
            const char *e = test_env("HPN_PGZ_CHUNK");
            chunk_bytes = e ? (size_t)atoll(e) : (size_t)2 << 20;
        }
        if (chunk_bytes < 1024) chunk_bytes = 1024;
        sym_cap_ = chunk_bytes * 24 < ((size_t)1 << 20) ? (size_t)1 << 20 : chunk_bytes * 24;
        window_ = (size_t)threads * 2;
        // chunk 0 starts at the member's first block; chunk k > 0 looks for a block from byte k * chunk on
        const uint64_t n = (size_ + chunk_bytes - 1) / chunk_bytes;
        chunks_.resize(n);
        for (uint64_t k = 0; k < n; ++k) {
            chunks_[k].lo = k * chunk_bytes * 8;
            chunks_[k].hi = (k + 1 < n ? (k + 1) * chunk_bytes : size_) * 8;
        }
        pos_ = (uint64_t)(body - data_) * 8;
        chunks_[0].start = pos_, chunks_[0].found = true;
        // chunks that begin inside the header have nothing to find
        for (uint64_t k = 1; k < n && chunks_[k].lo < pos_; ++k) chunks_[k].lo = pos_;
        cur_window_.assign(kHist, 0);
        for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work_loop(); });
        return true;
    }
    ~PgzReader()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
        for (auto &s : accepted_) release(*s);
        if (gap_) release(*gap_);
        for (Chunk &c : chunks_) release(c);
        for (uint16_t *p : sym_pool_) free(p - kHist);
        for (uint8_t *p : byte_pool_) free(p);
        if (fallback_) gzclose(fallback_);
        if (data_) munmap((void *)data_, size_);
        if (fd_ >= 0) close(fd_);
    }

    // Up to n bytes of the uncompressed stream; fewer only at its end (or at an error, like gzread).
    size_t read(void *dst, size_t n)
    {
        uint8_t *out = (uint8_t *)dst;
        size_t got = 0;
        while (got < n) {
            if (fallback_) {
                const size_t ask = n - got < ((size_t)1 << 30) ? n - got : (size_t)1 << 30;
                const int k = gzread(fallback_, out + got, (unsigned)ask);
                if (k < 0) damaged_ = true;   // zlib's verdict on the stream: the caller re-reads it the reference's way (InStream::damaged)
                if (k <= 0) break;
                got += (size_t)k;
                continue;
            }
            if (have_ && off_ < have_->nsym) {
                size_t k = have_->nsym - off_;
                if (k > n - got) k = n - got;
                memcpy(out + got, have_->bytes + off_, k);
                off_ += k, got += k, delivered_ += k;
                continue;
            }
            if (!next_segment()) break;
        }
        return got;
    }
    bool crc_failed() const { return crc_failed_; }
    // CRC-32 / ISIZE of a member wrong, or zlib refused the stream: what was delivered is NOT what the reference's gzgets hands out
    bool damaged() const { return crc_failed_ || damaged_; }
    // how the stream was produced (tests, HPN_TIMING)
    uint64_t chunks_accepted() const { return n_accepted_; }
    uint64_t gaps_decoded() const { return n_gaps_; }
    bool fell_back() const { return fallback_ != nullptr; }
    // seconds of thread time spent searching block starts / decoding / translating (+CRC), over all workers
    double seconds_find() const { return t_find_.load() * 1e-9; }
    double seconds_decode() const { return t_decode_.load() * 1e-9; }
    double seconds_translate() const { return t_translate_.load() * 1e-9; }

private:
    static constexpr size_t kHist = 32768;
    static constexpr uint64_t kNone = ~(uint64_t)0;
    enum Status { kOk = 0, kBad = 1, kTooBig = 2 };

    // One stretch of the deflate stream, decoded from a block start to a later block boundary.
    struct Chunk {
        uint64_t lo = 0, hi = 0;    // bit range in which this chunk looks for its start
        uint64_t start = kNone;     // block start found (bit offset in the file), kNone: nothing here
        bool found = false;         // `start` is final
        bool finding = false;       // a thread is looking for `start`
        bool taken = false, decoded = false, translated = false, discard = false;
        uint64_t next = 0;          // first chunk that may hold this stretch's end
        uint16_t *sym = nullptr;    // decoded symbols (kHist placeholders sit in front)
        size_t nsym = 0;
        uint64_t end = 0;           // bit offset of the boundary the decode stopped at; after a final block: byte-aligned
        bool final_block = false;   // stopped after a final block (member end)
        int status = kOk;
        std::vector<uint8_t> window;  // the kHist bytes before `start`
        uint8_t *bytes = nullptr;
        uint32_t crc = 0;
        bool member_end = false;
        uint32_t want_crc = 0, want_size = 0;
    };

    struct Stopwatch {  // adds its lifetime (ns) to a counter
        std::atomic<uint64_t> &acc;
        const double t0;
        explicit Stopwatch(std::atomic<uint64_t> &a) : acc(a), t0(wall_s()) {}
        ~Stopwatch() { acc.fetch_add((uint64_t)((wall_s() - t0) * 1e9), std::memory_order_relaxed); }
    };
    bool fail_open()
    {
        if (data_) munmap((void *)data_, size_);
        data_ = nullptr;
        if (fd_ >= 0) close(fd_);
        fd_ = -1;
        return false;
    }

    // ---- buffers ----------------------------------------------------------------------------------------
    uint16_t *get_sym()  // m_ held
    {
        if (!sym_pool_.empty()) {
            uint16_t *p = sym_pool_.back();
            sym_pool_.pop_back();
            return p;
        }
        uint16_t *raw = (uint16_t *)malloc((kHist + sym_cap_ + FastInflateT<uint16_t>::kOvershoot) * sizeof(uint16_t));
        if (!raw) return nullptr;
        for (size_t i = 0; i < kHist; ++i) raw[i] = (uint16_t)(256 + i);  // "the byte at history[i]"
        return raw + kHist;
    }
    uint8_t *get_bytes()  // m_ held
    {
        if (!byte_pool_.empty()) {
            uint8_t *p = byte_pool_.back();
            byte_pool_.pop_back();
            return p;
        }
        return (uint8_t *)malloc(sym_cap_ + 64);
    }
    void release(Chunk &c)  // m_ held (or no other thread left)
    {
        if (c.sym) sym_pool_.push_back(c.sym);
        if (c.bytes) byte_pool_.push_back(c.bytes);
        c.sym = nullptr, c.bytes = nullptr;
    }

    // ---- pass 1: where does a block start? ----------------------------------------------------------
    static uint16_t *scratch() { return gz_find_scratch(); }
    uint64_t find_start(uint64_t lo, uint64_t hi, uint16_t *scratch, size_t cap)
    {
        const Stopwatch sw(t_find_);
        return gz_find_block_start(data_, size_, lo, hi, scratch, cap);
    }

    // ---- pass 2: decode a stretch with unknown history -------------------------------------------
    // Should the decode that has arrived at block boundary b stop there?  (waits for the finders of
    // the chunks b has reached)
    bool reached(Chunk &c, uint64_t b)
    {
        std::unique_lock<std::mutex> lk(m_);
        while (c.next < chunks_.size()) {
            Chunk &n = chunks_[c.next];
            if (b < n.lo) return false;
            if (!n.found && !n.finding) {  // nobody has looked there yet: do it now rather than wait for a free worker
                n.finding = true;
                lk.unlock();
                const uint64_t s = n.lo < n.hi ? find_start(n.lo, n.hi, scratch(), kScratch) : kNone;
                lk.lock();
                n.start = s, n.found = true;
                cv_.notify_all();
            }
            cv_.wait(lk, [&] { return n.found || stop_; });
            if (stop_) return true;
            if (n.start == kNone || n.start < c.start) {
                ++c.next;
                continue;
            }
            return b >= n.start;
        }
        return false;  // the last stretch runs to the final block
    }
    void decode(Chunk &c)
    {
        const Stopwatch sw(t_decode_);
        FastInflateT<uint16_t> fi;
        fi.begin(data_ + (c.start >> 3), data_ + size_, (uint32_t)(c.start & 7), true);
        uint16_t *out = c.sym;
        for (;;) {
            const int r = fi.run(out, c.sym + sym_cap_, c.sym - kHist);
            if (r == FastInflateT<uint16_t>::kBlockEnd) {
                const uint64_t b = fi.bit_pos(data_);
                if (!reached(c, b)) continue;
                c.end = b;
                break;
            }
            if (r == FastInflateT<uint16_t>::kDone) {
                c.final_block = true;
                c.end = (uint64_t)(fi.in_pos() - data_) * 8;
                break;
            }
            c.status = r == FastInflateT<uint16_t>::kNeedOutput ? kTooBig : kBad;
            break;
        }
        c.nsym = (size_t)(out - c.sym);
    }

    // ---- pass 4: symbols -> bytes ------------------------------------------------------------------
    static void translate(const uint16_t *s, size_t n, const uint8_t *window, uint8_t *out)
    {
        size_t i = 0;
#if defined(__x86_64__)
        // low bytes of 16 symbols packed at once; the (few) placeholders among them are patched from the history
        const __m128i low = _mm_set1_epi16(0x00ff), zero = _mm_setzero_si128();
        for (; i + 16 <= n; i += 16) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 8));
            _mm_storeu_si128((__m128i *)(out + i), _mm_packus_epi16(_mm_and_si128(a, low), _mm_and_si128(b, low)));
            const __m128i high = _mm_packus_epi16(_mm_srli_epi16(a, 8), _mm_srli_epi16(b, 8));
            uint32_t m = ~(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(high, zero)) & 0xffffu;
            while (m) {
                const uint32_t k = (uint32_t)__builtin_ctz(m);
                out[i + k] = window[s[i + k] - 256];
                m &= m - 1;
            }
        }
#endif
        for (; i < n; ++i) out[i] = s[i] < 256 ? (uint8_t)s[i] : window[s[i] - 256];
    }

    // ---- scheduling ----------------------------------------------------------------------------------
    void work_loop()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (stop_) return;
            if (!todo_translate_.empty()) {  // accepted stretches first: the consumer waits for these
                std::shared_ptr<Chunk> c = todo_translate_.front();
                todo_translate_.pop_front();
                c->bytes = get_bytes();
                lk.unlock();
                if (c->bytes) {
                    const Stopwatch sw(t_translate_);
                    translate(c->sym, c->nsym, c->window.data(), c->bytes);
                    c->crc = crc32_fast(0, c->bytes, c->nsym);
                }
                lk.lock();
                if (!c->bytes) failed_ = true;
                sym_pool_.push_back(c->sym);
                c->sym = nullptr;
                c->translated = true;
                cv_.notify_all();
                continue;
            }
            if (gap_ && !gap_->taken) {  // a stretch no chunk covers: the stitcher waits for it
                std::shared_ptr<Chunk> g = gap_;
                g->taken = true;
                g->sym = get_sym();
                lk.unlock();
                if (g->sym) decode(*g);
                else g->status = kBad;
                lk.lock();
                g->decoded = true;
                stitch();
                cv_.notify_all();
                continue;
            }
            if (!failed_ && !done_ && next_take_ < chunks_.size() && next_take_ < stitch_idx_ + window_ && accepted_.size() < window_) {
                Chunk &c = chunks_[next_take_++];
                c.taken = true;
                c.next = (uint64_t)(&c - chunks_.data()) + 1;
                if (!c.found && !c.finding) {
                    c.finding = true;
                    lk.unlock();
                    const uint64_t s = c.lo < c.hi ? find_start(c.lo, c.hi, scratch(), kScratch) : kNone;
                    lk.lock();
                    c.start = s, c.found = true;
                    cv_.notify_all();
                }
                cv_.wait(lk, [&] { return c.found || stop_; });
                if (c.start == kNone || c.discard || stop_) {
                    c.decoded = true;
                    stitch();
                    cv_.notify_all();
                    continue;
                }
                c.sym = get_sym();
                lk.unlock();
                if (c.sym) decode(c);
                else c.status = kBad;
                lk.lock();
                c.decoded = true;
                if (c.discard) release(c);
                stitch();
                cv_.notify_all();
                continue;
            }
            cv_.wait(lk);
        }
    }

    // pass 3 (m_ held): accept, in stream order, the stretches that start where the previous one ended
    void stitch()
    {
        while (!failed_ && !done_) {
            std::shared_ptr<Chunk> seg;
            if (gap_) {
                if (!gap_->decoded) return;
                seg = gap_;
                gap_.reset();
            } else {
                // chunks that found nothing, or whose start the stream has already passed, are dropped
                while (stitch_idx_ < chunks_.size() && chunks_[stitch_idx_].found &&
                       (chunks_[stitch_idx_].start == kNone || chunks_[stitch_idx_].start < pos_)) {
                    Chunk &c = chunks_[stitch_idx_];
                    c.discard = true;
                    if (c.decoded || !c.taken) release(c);
                    ++stitch_idx_;
                }
                if (stitch_idx_ < chunks_.size()) {
                    Chunk &c = chunks_[stitch_idx_];
                    if (!c.found) return;
                    if (c.start == pos_) {
                        if (!c.decoded) return;
                        seg = std::make_shared<Chunk>(std::move(c));
                        c.sym = nullptr, c.bytes = nullptr;
                        ++stitch_idx_;
                        ++n_accepted_;
                    }
                }
                if (!seg) {  // nothing starts at pos_: decode from there to the next start
                    gap_ = std::make_shared<Chunk>();
                    gap_->start = pos_, gap_->found = true;
                    gap_->next = stitch_idx_;
                    ++n_gaps_;
                    cv_.notify_all();
                    return;
                }
            }
            if (seg->status != kOk) {
                release(*seg);
                failed_ = true;
                return;
            }
            // this stretch's history is the current window; the next one's is its resolved tail
            seg->window = cur_window_;
            if (seg->nsym >= kHist) {
                const uint16_t *t = seg->sym + seg->nsym - kHist;
                for (size_t i = 0; i < kHist; ++i) cur_window_[i] = t[i] < 256 ? (uint8_t)t[i] : seg->window[t[i] - 256];
            } else {
                const size_t keep = kHist - seg->nsym;
                memmove(cur_window_.data(), cur_window_.data() + seg->nsym, keep);
                for (size_t i = 0; i < seg->nsym; ++i)
                    cur_window_[keep + i] = seg->sym[i] < 256 ? (uint8_t)seg->sym[i] : seg->window[seg->sym[i] - 256];
            }
            pos_ = seg->end;
            if (seg->final_block) {  // trailer, then another member or the end of the data
                const uint64_t t = pos_ >> 3;
                if (t + 8 > size_) {  // truncated: zlib decides what that means
                    release(*seg);
                    failed_ = true;
                    return;
                }
                seg->member_end = true;
                memcpy(&seg->want_crc, data_ + t, 4), memcpy(&seg->want_size, data_ + t + 4, 4);
                const uint8_t *body = t + 8 < size_ ? gzip_header_end(data_ + t + 8, data_ + size_) : nullptr;
                if (body) {
                    pos_ = (uint64_t)(body - data_) * 8;
                    memset(cur_window_.data(), 0, kHist);
                } else {
                    done_ = true;  // end of file, or trailing bytes that are not a member (gzread ignores those)
                }
            }
            accepted_.push_back(seg);
            todo_translate_.push_back(seg);
            cv_.notify_all();
        }
    }

    // the consumer: next accepted stretch, translated
    bool next_segment()
    {
        std::unique_lock<std::mutex> lk(m_);
        if (have_) {
            release(*have_);
            have_.reset();
            cv_.notify_all();
        }
        off_ = 0;
        for (;;) {
            if (!accepted_.empty()) {
                std::shared_ptr<Chunk> c = accepted_.front();
                cv_.wait(lk, [&] { return c->translated; });
                accepted_.pop_front();
                if (!c->bytes) {
                    failed_ = true;
                    continue;
                }
                member_crc_ = member_bytes_ ? (uint32_t)crc32_combine(member_crc_, c->crc, (z_off_t)c->nsym) : c->crc;
                member_bytes_ += c->nsym;
                if (c->member_end) {
                    if (member_crc_ != c->want_crc || (uint32_t)member_bytes_ != c->want_size) crc_failed_ = true;
                    member_crc_ = 0, member_bytes_ = 0;
                }
                have_ = c;
                cv_.notify_all();
                if (c->nsym == 0) {
                    release(*have_);
                    have_.reset();
                    continue;
                }
                return true;
            }
            if (failed_) return start_fallback(lk);
            if (done_) return false;
            stitch();
            if (!accepted_.empty() || failed_ || done_) continue;
            cv_.wait(lk);
        }
    }

    // zlib takes over: re-read from the first byte, skip what was delivered
    bool start_fallback(std::unique_lock<std::mutex> &lk)
    {
        stop_ = true;
        cv_.notify_all();
        lk.unlock();
        for (auto &t : workers_) t.join();
        workers_.clear();
        lk.lock();
        const int fd = dup(fd_);
        if (fd < 0 || lseek(fd, 0, SEEK_SET) != 0) return false;
        fallback_ = gzdopen(fd, "rb");
        if (!fallback_) {
            close(fd);
            return false;
        }
        gzbuffer(fallback_, 1u << 20);
        std::vector<uint8_t> sink((size_t)1 << 20);
        uint64_t left = delivered_;
        while (left) {
            const unsigned ask = left < sink.size() ? (unsigned)left : (unsigned)sink.size();
            const int k = gzread(fallback_, sink.data(), ask);
            if (k < 0) damaged_ = true;
            if (k <= 0) return false;
            left -= (uint64_t)k;
        }
        return true;
    }

    static constexpr size_t kScratch = kGzFindScratch;

    int fd_ = -1;
    const uint8_t *data_ = nullptr;
    uint64_t size_ = 0;
    size_t sym_cap_ = 0, window_ = 0;

    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::thread> workers_;
    std::vector<Chunk> chunks_;
    std::deque<std::shared_ptr<Chunk>> accepted_, todo_translate_;
    std::shared_ptr<Chunk> gap_;
    std::vector<uint16_t *> sym_pool_;
    std::vector<uint8_t *> byte_pool_;
    std::vector<uint8_t> cur_window_;
    uint64_t pos_ = 0;  // bit offset the accepted stream has reached
    size_t next_take_ = 0, stitch_idx_ = 0;
    bool stop_ = false, failed_ = false, done_ = false;
    uint64_t n_accepted_ = 0, n_gaps_ = 0;
    std::atomic<uint64_t> t_find_{0}, t_decode_{0}, t_translate_{0};

    // consumer side
    std::shared_ptr<Chunk> have_;
    size_t off_ = 0;
    uint64_t delivered_ = 0, member_bytes_ = 0;
    uint32_t member_crc_ = 0;
    bool crc_failed_ = false;
    gzFile fallback_ = nullptr;
    bool damaged_ = false;
};

}  // namespace hpn
