// mgz_reader.hpp -- parallel inflate of concatenated gzip members.
//
// The reference reads .fastq.gz through zlib's gzread, which inflates the members of a
// multi-member file one after the other on the calling thread (fastq_count.c:112-118 via
// gzgets); at ~0.4 GB/s of text per core that is the whole run time of the gzip workloads.
// Members are independent deflate streams, so they can be inflated side by side -- the only
// thing a gzip file lacks is an index of where they start.  This reader speculates:
//
//   * candidate starts = byte patterns 1f 8b 08 <flags with the reserved bits clear>, found by
//     scanning ahead of the consumer in the mmap'ed file;
//   * a pool of threads inflates candidates (zlib inflate with the gzip wrapper, so header,
//     CRC32 and ISIZE are checked exactly as gzread checks them) into recycled 4 MiB blocks;
//   * the consumer only ever takes bytes from the member that starts where the previous one
//     ended (the first at offset 0), so the bytes delivered are exactly gzread's; candidates
//     that turn out to lie inside a member are cancelled and dropped.
//
// The member being consumed is streamed block by block while it is still being inflated, and
// no member may run more than 128 MiB ahead of its consumption: a single-member file of any
// size degenerates to one worker and a bounded buffer, i.e. to zlib's own behaviour.
//
// Whatever zlib would treat specially is handed to zlib itself: when a member fails to inflate
// (corrupt, truncated), gzdopen re-reads it from its first byte, the bytes already delivered
// are skipped, and gzread's remaining output and its error take over; bytes after the last
// member that are not a gzip header are trailing garbage, which gzread ignores.
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

#include "cpus.hpp"
#include "fast_inflate.hpp"

namespace hpn {

class MgzReader {
public:
    // threads <= 0: HPN_GZ_THREADS, else min(16, usable CPUs)
    bool open(const char *path, int threads = 0)
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
        if (!is_header(0)) return fail_open();
        if (threads <= 0) {
            const char *e = getenv("HPN_GZ_THREADS");
            long n = e ? atol(e) : usable_cpus();
            threads = (int)(n < 1 ? 1 : n > 16 ? 16 : n);
        }
        max_jobs_ = (size_t)threads + 2;
        for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work_loop(); });
        return true;
    }
    ~MgzReader()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            for (auto &j : jobs_) j->cancel = true;
            if (cur_) cur_->cancel = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
        if (have_.p) free(have_.p);
        if (cur_) drop_blocks(*cur_);
        for (auto &j : jobs_) drop_blocks(*j);
        for (Block &b : pool_) free(b.p);
        if (fallback_) gzclose(fallback_);
        if (data_) munmap((void *)data_, size_);
        if (fd_ >= 0) close(fd_);
    }

    // zlib refused the stream (a member's CRC-32 / ISIZE / data): what was delivered is NOT what the reference's gzgets hands out
    bool damaged() const { return damaged_; }

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
            if (have_.p && pos_ < have_.n) {
                size_t k = have_.n - pos_;
                if (k > n - got) k = n - got;
                memcpy(out + got, have_.p + pos_, k);
                pos_ += k, got += k, delivered_ += k;
                continue;
            }
            if (!next_block()) break;
        }
        return got;
    }

private:
    enum State { kQueued, kRunning, kDone, kFailed };
    static constexpr size_t kBlock = (size_t)4 << 20;
    static constexpr size_t kMaxAhead = 32;  // blocks a member may hold unconsumed (128 MiB)
    struct Block {
        uint8_t *p = nullptr;
        size_t n = 0;
    };
    struct Job {
        uint64_t start = 0, end = 0;
        State st = kQueued;
        std::atomic<bool> cancel{false};
        std::deque<Block> out;  // inflated, not yet consumed; guarded by m_
    };

    // ---- block pool: after the first members no page is faulted in again ------------------
    Block get_block()
    {
        {
            std::lock_guard<std::mutex> lk(pool_m_);
            if (!pool_.empty()) {
                Block b = pool_.back();
                pool_.pop_back();
                b.n = 0;
                return b;
            }
        }
        Block b;
        void *p = nullptr;
        if (posix_memalign(&p, (size_t)2 << 20, kBlock) != 0) return b;
        madvise(p, kBlock, MADV_HUGEPAGE);
        b.p = (uint8_t *)p;
        return b;
    }
    void put_block(Block b)
    {
        if (!b.p) return;
        std::lock_guard<std::mutex> lk(pool_m_);
        pool_.push_back(b);
    }
    void drop_blocks(Job &j)  // m_ held (or no other thread left)
    {
        for (Block &b : j.out) put_block(b);
        j.out.clear();
    }

    bool fail_open()
    {
        if (data_) munmap((void *)data_, size_);
        data_ = nullptr;
        if (fd_ >= 0) close(fd_);
        fd_ = -1;
        return false;
    }
    bool is_header(uint64_t p) const
    {
        return p + 18 <= size_ && data_[p] == 0x1f && data_[p + 1] == 0x8b && data_[p + 2] == 8 && !(data_[p + 3] & 0xe0);
    }
    // next candidate in [p, limit) (limit if none)
    uint64_t find_candidate(uint64_t p, uint64_t limit) const
    {
        if (limit > size_) limit = size_;
        while (p + 18 <= limit) {
            const uint8_t *q = (const uint8_t *)memchr(data_ + p, 0x1f, limit - 17 - p);
            if (!q) return limit;
            p = (uint64_t)(q - data_);
            if (is_header(p)) return p;
            ++p;
        }
        return limit;
    }
    // Keep the queue of speculative jobs filled, in file order.  Called by the consumer with
    // m_ held; the scan itself (read-only data, consumer-only cursor) runs unlocked, and looks
    // at most 256 MiB ahead per call so that a single huge member is not read twice up front.
    void schedule(std::unique_lock<std::mutex> &lk)
    {
        uint64_t budget = (uint64_t)256 << 20;
        while (jobs_.size() + (cur_ ? 1 : 0) < max_jobs_ && scan_ < size_ && budget) {
            const uint64_t from = scan_ < expect_ ? expect_ : scan_;
            if (from >= size_) {
                scan_ = size_;
                break;
            }
            const uint64_t limit = from + budget < size_ ? from + budget : size_;
            lk.unlock();
            const uint64_t c = find_candidate(from, limit);
            lk.lock();
            const uint64_t reached = c < limit ? c + 1 : limit;
            budget -= reached - from;
            scan_ = reached;
            if (c >= limit) continue;
            bool have = cur_ && cur_->start == c;
            for (auto &j : jobs_) have |= j->start == c;
            if (have) continue;
            auto j = std::make_shared<Job>();
            j->start = c;
            jobs_.push_back(j);
            cv_.notify_all();
        }
    }

    void work_loop()
    {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this, &job] {
                    if (stop_) return true;
                    if (cur_ && cur_->st == kQueued) {
                        job = cur_;
                        return true;
                    }
                    for (auto &j : jobs_)
                        if (j->st == kQueued) {
                            job = j;
                            return true;
                        }
                    return false;
                });
                if (stop_) return;
                job->st = kRunning;
            }
            inflate_member(job);
        }
    }

    // Hand a filled block to the job's queue; wait while the member is too far ahead of its
    // consumption.  false = cancelled / stopping.
    bool publish(const std::shared_ptr<Job> &j, Block b)
    {
        std::unique_lock<std::mutex> lk(m_);
        j->out.push_back(b);
        cv_.notify_all();
        cv_.wait(lk, [&] { return j->out.size() < kMaxAhead || j->cancel || stop_; });
        return !(j->cancel || stop_);
    }

    // The quick decoder first (fast_inflate.hpp), with gzread's own checks of the trailer; zlib when it
    // declines before anything was handed out; the consumer's zlib fallback when it fails later.
    void inflate_member(const std::shared_ptr<Job> &j)
    {
        static const bool use_fast = [] {
            const char *e = test_env("HPN_FAST_INFLATE");
            return !(e && e[0] == '0');
        }();
        if (use_fast) {
            const int r = inflate_member_fast(j);
            if (r != 0) return;
        }
        inflate_member_zlib(j);
    }

    // 1 = member done, -1 = failed after output was published (state set), 0 = declined, nothing published
    int inflate_member_fast(const std::shared_ptr<Job> &j)
    {
        // gzip header (RFC 1952): magic, CM, FLG, MTIME(4), XFL, OS, then optional fields
        const uint8_t *p = data_ + j->start, *lim = data_ + size_;
        if (lim - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return 0;
        const uint32_t flg = p[3];
        p += 10;
        if (flg & 4) {  // FEXTRA
            if (lim - p < 2) return 0;
            const size_t xl = p[0] | p[1] << 8;
            if ((size_t)(lim - p) < 2 + xl) return 0;
            p += 2 + xl;
        }
        for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
            if (flg & f) {
                const void *z = memchr(p, 0, (size_t)(lim - p));
                if (!z) return 0;
                p = (const uint8_t *)z + 1;
            }
        if (flg & 2) {  // FHCRC
            if (lim - p < 2) return 0;
            p += 2;
        }
        constexpr size_t kHist = 32768, kChunk = kBlock - 512;
        static thread_local std::vector<uint8_t> buf;
        buf.resize(kHist + kChunk + FastInflate::kOvershoot);
        FastInflate fi;
        fi.begin(p, lim);
        uint8_t *const base = buf.data() + kHist;
        uint8_t *cur = base;
        const uint8_t *hist = base;
        uint32_t crc = 0;
        uint64_t total = 0;
        bool published = false;
        for (;;) {
            if (j->cancel.load(std::memory_order_relaxed)) break;
            const int r = fi.run(cur, base + kChunk, hist);
            const size_t n = (size_t)(cur - base);
            if (r == FastInflate::kError) break;
            if (n) {
                crc = crc32_fast(crc, base, n);
                total += n;
                if (r == FastInflate::kDone) {  // the trailer decides before the last block goes out
                    const uint8_t *t = fi.in_pos();
                    uint32_t want_crc, want_size;
                    if (lim - t < 8) break;
                    memcpy(&want_crc, t, 4), memcpy(&want_size, t + 4, 4);
                    if (want_crc != crc || want_size != (uint32_t)total) break;
                }
                Block b = get_block();
                if (!b.p) break;
                memcpy(b.p, base, n);
                b.n = n;
                published = true;
                if (!publish(j, b)) break;
            } else if (r == FastInflate::kDone) {  // empty member (or nothing new): still check the trailer
                const uint8_t *t = fi.in_pos();
                uint32_t want_crc, want_size;
                if (lim - t < 8) break;
                memcpy(&want_crc, t, 4), memcpy(&want_size, t + 4, 4);
                if (want_crc != crc || want_size != (uint32_t)total) break;
            }
            if (r == FastInflate::kDone) {
                std::lock_guard<std::mutex> lk(m_);
                j->end = (uint64_t)(fi.in_pos() + 8 - data_);
                j->st = kDone;
                if (j->cancel) drop_blocks(*j);
                cv_.notify_all();
                return 1;
            }
            const size_t keep = n < kHist ? n + (size_t)(base - hist) > kHist ? kHist : n + (size_t)(base - hist) : kHist;
            memmove(base - keep, cur - keep, keep);  // the last `keep` bytes of history + new output stay in front
            hist = base - keep;
            cur = base;
        }
        if (!published && !j->cancel) return 0;
        std::lock_guard<std::mutex> lk(m_);
        j->st = kFailed;
        if (j->cancel) drop_blocks(*j);
        cv_.notify_all();
        return -1;
    }

    void inflate_member_zlib(const std::shared_ptr<Job> &j)
    {
        z_stream s;
        memset(&s, 0, sizeof s);
        bool ok = inflateInit2(&s, 15 + 16) == Z_OK;
        const bool inited = ok;
        uint64_t in_pos = j->start;
        int rc = Z_OK;
        Block b;
        while (ok && rc != Z_STREAM_END) {
            if (j->cancel.load(std::memory_order_relaxed)) {
                ok = false;
                break;
            }
            if (s.avail_in == 0) {
                const uint64_t left = size_ - in_pos;
                if (left == 0) {  // input exhausted inside the member: truncated file
                    ok = false;
                    break;
                }
                const uint32_t take = left > ((uint64_t)1 << 30) ? 1u << 30 : (uint32_t)left;
                s.next_in = (Bytef *)(data_ + in_pos);
                s.avail_in = take;
                in_pos += take;
            }
            if (!b.p) {
                b = get_block();
                if (!b.p) {
                    ok = false;
                    break;
                }
            }
            s.next_out = b.p + b.n;
            s.avail_out = (uInt)(kBlock - b.n);
            rc = inflate(&s, Z_NO_FLUSH);
            b.n = kBlock - s.avail_out;
            if (rc != Z_OK && rc != Z_STREAM_END) {  // Z_DATA_ERROR, Z_BUF_ERROR, ...
                ok = false;
                break;
            }
            if (b.n == kBlock && rc != Z_STREAM_END) {
                const bool go = publish(j, b);
                b = Block();
                if (!go) {
                    ok = false;
                    break;
                }
            }
        }
        if (inited) inflateEnd(&s);
        std::lock_guard<std::mutex> lk(m_);
        if (ok) {
            if (b.p && b.n) j->out.push_back(b);
            else put_block(b);
            j->end = in_pos - s.avail_in;
            j->st = kDone;
        } else {
            put_block(b);  // the partial block of a failed member is never delivered
            j->st = kFailed;
        }
        if (j->cancel) drop_blocks(*j);  // dropped by the consumer meanwhile
        cv_.notify_all();
    }

    // The consumer ran out of bytes: next block of the current member, or the next member.
    // false = end of the stream.
    bool next_block()
    {
        if (have_.p) {
            put_block(have_);
            have_ = Block();
        }
        pos_ = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (cur_) {
                cv_.wait(lk, [&] { return !cur_->out.empty() || cur_->st == kDone || cur_->st == kFailed; });
                if (!cur_->out.empty()) {
                    have_ = cur_->out.front();
                    cur_->out.pop_front();
                    cv_.notify_all();  // the producer may go on
                    return true;
                }
                if (cur_->st == kFailed) return start_fallback(lk);
                expect_ = cur_->end;  // member complete and fully consumed
                cur_.reset();
                delivered_ = 0;
            }
            // everything that starts before expect_ lies inside a consumed member: drop it
            for (auto it = jobs_.begin(); it != jobs_.end();) {
                if ((*it)->start < expect_) {
                    (*it)->cancel = true;
                    if ((*it)->st == kDone || (*it)->st == kFailed) drop_blocks(**it);
                    it = jobs_.erase(it);
                } else {
                    ++it;
                }
            }
            cv_.notify_all();
            if (expect_ >= size_) return false;
            for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
                if ((*it)->start == expect_) {
                    cur_ = *it;
                    jobs_.erase(it);
                    break;
                }
            if (!cur_) {
                if (!is_header(expect_)) {
                    // not something this reader speculates on: a gzip magic goes to zlib as it is
                    // (it will report the bad method / flags), anything else is trailing garbage
                    if (data_[expect_] == 0x1f && expect_ + 1 < size_ && data_[expect_ + 1] == 0x8b) return start_fallback(lk);
                    return false;
                }
                cur_ = std::make_shared<Job>();
                cur_->start = expect_;
            }
            cv_.notify_all();  // a waiting producer of this member may go on; a free worker may take it
            schedule(lk);
        }
    }

    // zlib takes over at the start of the current member (a gzip magic is there): the bytes of it
    // that were already delivered are skipped, the rest -- and the error -- are gzread's.
    bool start_fallback(std::unique_lock<std::mutex> &lk)
    {
        stop_ = true;
        for (auto &j : jobs_) {
            j->cancel = true;
            if (j->st == kDone || j->st == kFailed) drop_blocks(*j);
        }
        jobs_.clear();
        const uint64_t at = cur_ ? cur_->start : expect_;
        if (cur_) {
            cur_->cancel = true;
            drop_blocks(*cur_);
            cur_.reset();
        }
        lk.unlock();
        cv_.notify_all();
        const int fd = dup(fd_);
        bool ok = fd >= 0 && lseek(fd, (off_t)at, SEEK_SET) >= 0;
        if (ok) {
            fallback_ = gzdopen(fd, "rb");
            ok = fallback_ != nullptr;
        }
        if (ok) {
            gzbuffer(fallback_, 1u << 20);
            std::vector<uint8_t> sink((size_t)1 << 20);
            uint64_t skip = delivered_;
            while (skip) {
                const int k = gzread(fallback_, sink.data(), (unsigned)(skip < sink.size() ? skip : sink.size()));
                if (k < 0) damaged_ = true;
                if (k <= 0) break;
                skip -= (uint64_t)k;
            }
        }
        lk.lock();
        return ok;  // read() continues through gzread
    }

    int fd_ = -1;
    const uint8_t *data_ = nullptr;
    uint64_t size_ = 0;
    uint64_t expect_ = 0;     // compressed offset where the next member must start
    uint64_t scan_ = 0;       // candidates before this offset are already queued
    uint64_t delivered_ = 0;  // bytes of the current member handed to the caller
    size_t max_jobs_ = 4;
    std::deque<std::shared_ptr<Job>> jobs_;  // speculative members, in file order
    std::shared_ptr<Job> cur_;               // the member being consumed
    Block have_;                             // the block being read from
    size_t pos_ = 0;
    std::mutex pool_m_;
    std::vector<Block> pool_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
    gzFile fallback_ = nullptr;
    bool damaged_ = false;
};

}  // namespace hpn
