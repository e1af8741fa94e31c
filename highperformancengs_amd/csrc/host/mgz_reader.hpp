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
//     CRC32 and ISIZE are checked exactly as gzread checks them);
//   * the consumer only ever accepts the member that starts where the previous accepted one
//     ended (the first at offset 0), so the bytes delivered are exactly gzread's; candidates
//     that turn out to lie inside a member are cancelled and dropped.
//
// Whatever zlib would treat specially is handed to zlib itself: a member that fails to inflate
// (corrupt, truncated) is re-read with gzdopen from its first byte on, which reproduces gzread's
// partial output and error; bytes after the last member that are not a gzip header are trailing
// garbage, which gzread ignores.  A single-member file simply degenerates to one worker.
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

namespace hpn {

class MgzReader {
public:
    // threads <= 0: HPN_GZ_THREADS, else min(16, online CPUs)
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
        if (!is_header(0)) return fail_open();
        if (threads <= 0) {
            const char *e = getenv("HPN_GZ_THREADS");
            long n = e ? atol(e) : usable_cpus();
            threads = (int)(n < 1 ? 1 : n > 16 ? 16 : n);
        }
        max_jobs_ = (size_t)threads + 2;
        scan_ = 0;
        for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work_loop(); });
        return true;
    }
    ~MgzReader()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            for (auto &j : jobs_) j->cancel = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
        if (cur_) put_blocks(cur_->out);
        for (auto &j : jobs_) put_blocks(j->out);
        for (Block &b : pool_) free(b.p);
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
                if (k <= 0) break;
                got += (size_t)k;
                continue;
            }
            if (cur_ && blk_ < cur_->out.size()) {
                Block &b = cur_->out[blk_];
                size_t k = b.n - pos_;
                if (k > n - got) k = n - got;
                memcpy(out + got, b.p + pos_, k);
                pos_ += k, got += k;
                if (pos_ == b.n) ++blk_, pos_ = 0;
                continue;
            }
            if (!next_member()) break;
        }
        return got;
    }

private:
    enum State { kQueued, kRunning, kDone, kFailed };
    // Inflated bytes live in fixed 4 MiB blocks that are recycled through a pool: after the
    // first few members no page is ever faulted in again (growing one vector per member made
    // the workers fight over the address-space lock and ran slower than one thread).
    static constexpr size_t kBlock = (size_t)4 << 20;
    struct Block {
        uint8_t *p = nullptr;
        size_t n = 0;
    };
    struct Job {
        uint64_t start = 0, end = 0;
        State st = kQueued;
        std::atomic<bool> cancel{false};
        std::vector<Block> out;
    };
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
    void put_blocks(std::vector<Block> &v)
    {
        std::lock_guard<std::mutex> lk(pool_m_);
        for (Block &b : v)
            if (b.p) pool_.push_back(b);
        v.clear();
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
    // next candidate at or after p (size_ if none)
    uint64_t find_candidate(uint64_t p) const
    {
        while (p + 18 <= size_) {
            const uint8_t *q = (const uint8_t *)memchr(data_ + p, 0x1f, size_ - 17 - p);
            if (!q) return size_;
            p = (uint64_t)(q - data_);
            if (is_header(p)) return p;
            ++p;
        }
        return size_;
    }
    // with m_ held: keep the queue of speculative jobs filled, in file order
    void schedule()
    {
        while (jobs_.size() < max_jobs_ && scan_ < size_) {
            uint64_t c = find_candidate(scan_ < expect_ ? expect_ : scan_);
            if (c >= size_) {
                scan_ = size_;
                break;
            }
            scan_ = c + 1;
            bool have = false;
            for (auto &j : jobs_) have |= j->start == c;
            if (have) continue;
            auto j = std::make_shared<Job>();
            j->start = c;
            jobs_.push_back(j);
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
            const bool ok = inflate_member(*job);
            {
                std::lock_guard<std::mutex> lk(m_);
                job->st = ok ? kDone : kFailed;
                if (job->cancel) put_blocks(job->out);  // dropped by the consumer meanwhile
            }
            cv_.notify_all();
        }
    }

    bool inflate_member(Job &j)
    {
        z_stream s;
        memset(&s, 0, sizeof s);
        if (inflateInit2(&s, 15 + 16) != Z_OK) return false;
        uint64_t in_pos = j.start;
        int rc = Z_OK;
        bool ok = true;
        while (rc != Z_STREAM_END) {
            if (j.cancel.load(std::memory_order_relaxed)) break;
            if (s.avail_in == 0) {
                const uint64_t left = size_ - in_pos;
                if (left == 0) break;  // input exhausted inside the member: truncated file
                const uint32_t take = left > ((uint64_t)1 << 30) ? 1u << 30 : (uint32_t)left;
                s.next_in = (Bytef *)(data_ + in_pos);
                s.avail_in = take;
                in_pos += take;
            }
            if (j.out.empty() || j.out.back().n == kBlock) {
                Block b = get_block();
                if (!b.p) {
                    ok = false;
                    break;
                }
                j.out.push_back(b);
            }
            Block &b = j.out.back();
            s.next_out = b.p + b.n;
            s.avail_out = (uInt)(kBlock - b.n);
            rc = inflate(&s, Z_NO_FLUSH);
            b.n = kBlock - s.avail_out;
            if (rc != Z_OK && rc != Z_STREAM_END) break;  // Z_DATA_ERROR, Z_BUF_ERROR, ...
        }
        ok = ok && rc == Z_STREAM_END;
        if (ok) {
            j.end = in_pos - s.avail_in;
            if (!j.out.empty() && j.out.back().n == 0) {  // a block taken right before the stream ended
                std::vector<Block> last(1, j.out.back());
                j.out.pop_back();
                put_blocks(last);
            }
        } else {
            put_blocks(j.out);
        }
        inflateEnd(&s);
        return ok;
    }

    // Advance to the member that starts at expect_.  false = end of the stream.
    bool next_member()
    {
        std::unique_lock<std::mutex> lk(m_);
        if (cur_) put_blocks(cur_->out);
        cur_.reset();
        pos_ = 0, blk_ = 0;
        for (;;) {
            // everything that starts before expect_ lies inside an accepted member: drop it
            for (auto it = jobs_.begin(); it != jobs_.end();) {
                if ((*it)->start < expect_) {
                    (*it)->cancel = true;
                    if ((*it)->st == kDone) put_blocks((*it)->out);
                    it = jobs_.erase(it);
                } else {
                    ++it;
                }
            }
            if (expect_ >= size_) return false;
            std::shared_ptr<Job> mine;
            for (auto &j : jobs_)
                if (j->start == expect_) mine = j;
            if (!mine) {
                if (!is_header(expect_)) {
                    // not something this reader speculates on: a gzip magic goes to zlib as it is
                    // (it will report the bad method / flags), anything else is trailing garbage
                    if (data_[expect_] == 0x1f && expect_ + 1 < size_ && data_[expect_ + 1] == 0x8b) return start_fallback(lk);
                    return false;
                }
                mine = std::make_shared<Job>();
                mine->start = expect_;
                jobs_.push_front(mine);
            }
            schedule();
            cv_.notify_all();
            cv_.wait(lk, [&] { return mine->st == kDone || mine->st == kFailed; });
            if (mine->st == kFailed) return start_fallback(lk);
            cur_ = mine;
            expect_ = mine->end;
            for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
                if (*it == mine) {
                    jobs_.erase(it);
                    break;
                }
            schedule();
            cv_.notify_all();
            if (!cur_->out.empty()) return true;
            cur_.reset();  // empty member: go on to the next one
        }
    }

    // zlib takes over at expect_ (a gzip magic is there): its output and its error are the reference's.
    bool start_fallback(std::unique_lock<std::mutex> &lk)
    {
        stop_ = true;
        for (auto &j : jobs_) {
            j->cancel = true;
            if (j->st == kDone) put_blocks(j->out);
        }
        jobs_.clear();
        lk.unlock();
        cv_.notify_all();
        const int fd = dup(fd_);
        if (fd < 0 || lseek(fd, (off_t)expect_, SEEK_SET) < 0) return false;
        fallback_ = gzdopen(fd, "rb");
        if (!fallback_) return false;
        gzbuffer(fallback_, 1u << 20);
        lk.lock();
        return true;  // read() continues through gzread
    }

    int fd_ = -1;
    const uint8_t *data_ = nullptr;
    uint64_t size_ = 0;
    uint64_t expect_ = 0;  // compressed offset where the next accepted member must start
    uint64_t scan_ = 0;    // candidates before this offset are already queued
    size_t max_jobs_ = 4;
    std::deque<std::shared_ptr<Job>> jobs_;
    std::shared_ptr<Job> cur_;
    size_t blk_ = 0, pos_ = 0;  // read position in cur_
    std::mutex pool_m_;
    std::vector<Block> pool_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
    gzFile fallback_ = nullptr;
};

}  // namespace hpn
