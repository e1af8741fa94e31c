// text_stream.hpp -- raw text of one FASTQ input -> pinned chunks for the device framer.
//
// The reference's tools spend their FASTQ time in gzgets (fastq_count.c:112-118,
// fastq_trim.c:67-89).  With hpn_fastq_text_count / hpn_fastq_text_trim the framing runs
// on the GPU, so the host only has to deliver bytes: a reader thread fills pinned chunks
//   - plain regular files: parallel pread straight into the pinned buffer,
//   - gzip: zlib gzread (inflate is the bound, as in the reference),
//   - BGZF: the block-parallel inflater (bam_reader.hpp),
// while the caller's thread submits the previous chunk.  The chunk that hits the end of
// the data carries eof = true (it may be empty).
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>

#include <atomic>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "fastq_reader.hpp"
#include "hpngs.h"

namespace hpn {

inline bool text_path_enabled()
{
    const char *e = getenv("HPN_TEXT");
    return !(e && e[0] == '0');
}

// default chunk: the pinned footprint of a run stays near 100 MB however many workers there are
inline size_t text_chunk_bytes()
{
    const char *e = test_env("HPN_TEXT_CHUNK");
    const long long v = e ? atoll(e) : 0;
    if (v >= 64) return (size_t)v;
    const size_t c = ((size_t)32 << 20) / (size_t)text_workers_in_flight();
    return c < ((size_t)4 << 20) ? (size_t)4 << 20 : c;
}

// How many of the `requested` workers (-t) to start.  gzip input is inflate-bound at a few
// hundred MB/s per file, so every file gets its worker, as in the reference.  Plain files
// stream at tens of GB/s per worker: a few lanes saturate PCIe, and each further GPU context
// only adds start-up (15-30 ms of hardware-queue creation each, serialised by the driver) --
// one lane per 8 GiB of input, two per device (= per PCIe link) at most.
inline int text_workers(char **files, int n, int requested, int ndev = 1)
{
    if (!text_path_enabled() || test_env("HPN_ALL_WORKERS")) return requested;
    uint64_t plain = 0;
    for (int i = 0; i < n; ++i) {
        struct stat sb;
        if (strncmp(files[i], "-", 1) == 0 || !strcmp(files[i], "") || stat(files[i], &sb) != 0 || !S_ISREG(sb.st_mode))
            return requested;
        uint8_t magic[2] = {0, 0};
        const int fd = open(files[i], O_RDONLY);
        const ssize_t k = fd >= 0 ? pread(fd, magic, 2, 0) : 0;
        if (fd >= 0) close(fd);
        if (k == 2 && magic[0] == 0x1f && magic[1] == 0x8b) return requested;
        plain += (uint64_t)sb.st_size;
    }
    // (round 6: one lane per 8 GiB, two per device.  ONE worker streams a plain file at 49 GB/s of the link's 51 - 53
    // (profiles/r05/plain_a.txt), two reach it; the third and fourth of rounds 3 - 5 only added their contexts' set-up --
    // 30 ms each, one after the other in the driver -- to a run that is otherwise the process's start plus bytes / link:
    // profiles/r06/plain8_timeline.txt)
    int lanes = (int)(plain >> 33) + 1;
    if (lanes > 2 * (ndev < 1 ? 1 : ndev)) lanes = 2 * (ndev < 1 ? 1 : ndev);
    return lanes < requested ? lanes : requested;
}

class TextPump {
public:
    struct Chunk {
        uint8_t *p = nullptr;
        size_t n = 0;
        bool eof = false;
        int idx = -1;
    };

    // raw: deliver the file's own bytes whatever they are (compressed BGZF for the device inflater)
    // start: file offset the stream begins at (plain / raw files only)
    // pad: writable bytes in front of and behind every chunk (text_shard.hpp puts a piece's head byte and tail there)
    // read_threads: pread threads of a plain file (0: HPN_READ_THREADS, else 6 shared among the workers in flight)
    TextPump(hpn_ctx *ctx, const char *path, size_t chunk, int nbuf = 3, bool raw = false, uint64_t start = 0, size_t pad = 0, int read_threads = 0)
        : ctx_(ctx), cap_(chunk), pad_(pad), pos_(start), read_threads_(read_threads)
    {
        struct stat sb;
        uint8_t magic[2] = {0, 0};
        const bool is_stdin = strncmp(path, "-", 1) == 0 || !strcmp(path, "");
        if (!is_stdin && stat(path, &sb) == 0 && S_ISREG(sb.st_mode)) {
            const int fd = open(path, O_RDONLY);
            if (fd >= 0) {
                const ssize_t k = pread(fd, magic, 2, 0);
                if (raw || k < 2 || magic[0] != 0x1f || magic[1] != 0x8b) {  // zlib would copy it through unchanged
                    fd_ = fd;
                } else {
                    close(fd);
                }
            }
        }
        if (fd_ < 0) in_ = open_input_stream(path);
        if (!is_stdin && stat(path, &sb) == 0 && S_ISREG(sb.st_mode)) {
            // small inputs get small buffers (pinning memory costs ~0.3 ms per MB): the whole
            // plain file + 1 byte, or 16 x the compressed size, rounded up to 64 KiB
            // (never nothing: an input that did not exist has just been created empty -- open_input_stream's O_CREAT, the reference's
            // IO_stream.h:127 -- and a reader with buffers of no bytes would hand out empty chunks for ever)
            uint64_t guess = ((fd_ >= 0 ? (uint64_t)sb.st_size + 1 : (uint64_t)sb.st_size * 16) + 65535) & ~(uint64_t)65535;
            if (guess < 65536) guess = 65536;
            if (guess < cap_) cap_ = (size_t)guess;
        }
        // ONE buffer now, the others when the reader first needs them (on its thread, beside the caller's work on the chunks before):
        // pinning memory costs ~0.2 ms per MiB when it is made and as much again when the process ends (scripts/micro/startup_hip.hip,
        // exit_hip.hip) -- three chunks of 88 MiB in a row were 50 ms in front of a BAM's first byte
        nbuf_ = nbuf < 1 ? 1 : nbuf;
        buf_.reserve((size_t)nbuf_);
        {   // (pinned next to the device, like the ones the reader's thread makes: the caller's thread is put back where it was)
            cpu_set_t was;
            const bool have = sched_getaffinity(0, sizeof was, &was) == 0;
            bind_thread_near(ctx_);
            ok_ = grow();
            if (have) (void)sched_setaffinity(0, sizeof was, &was);
        }
        ok_ = ok_ && (fd_ >= 0 || in_.gz || in_.bz || in_.mz || in_.pz);
        if (ok_) th_ = std::thread([this] { loop(); });
    }
    ~TextPump()
    {
        halt();
        for (uint8_t *p : buf_) hpn_host_free(ctx_, p);
        if (fd_ >= 0) close(fd_);
        if (!handed_over_) in_.close();
    }
    bool ok() const { return ok_; }
    size_t chunk_bytes() const { return cap_; }
    bool plain_file() const { return fd_ >= 0; }
    // the reader met a CRC-32 / ISIZE / data error (valid once the eof chunk is out): the caller reads the input again through
    // open_input_stream_exact, whose bytes are the reference's even then
    bool damaged() const { return in_.damaged(); }

    // Start over at file offset `pos` with the same pinned buffers (plain / raw files only): what was read ahead is dropped.
    bool restart(uint64_t pos)
    {
        if (fd_ < 0 || !ok_) return false;
        halt();
        full_.clear(), free_.clear();
        for (size_t i = 0; i < buf_.size(); ++i) free_.push_back((int)i);
        stop_ = false, done_ = false, pos_ = pos;
        th_ = std::thread([this] { loop(); });
        return true;
    }

    // Next filled chunk in stream order; false once the eof chunk has been handed out.
    bool next(Chunk &c)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !full_.empty() || done_; });
        if (full_.empty()) return false;
        c = full_.front();
        full_.pop_front();
        return true;
    }
    void recycle(const Chunk &c)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            free_.push_back(c.idx);
        }
        cv_.notify_all();
    }

    // Stop reading.  Every byte that was read but not handed out is appended to `rest`,
    // and the returned stream continues right behind those bytes.
    InStream stop(std::vector<char> &rest)
    {
        halt();
        for (const Chunk &c : full_) rest.insert(rest.end(), (const char *)c.p, (const char *)c.p + c.n);
        full_.clear();
        if (fd_ >= 0) {
            lseek(fd_, (off_t)pos_, SEEK_SET);
            in_.gz = gzdopen(fd_, "rb");
            if (in_.gz) gzbuffer(in_.gz, 1u << 20);
            fd_ = -1;
        }
        handed_over_ = true;
        return in_;
    }

private:
    void halt()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    bool grow()       // one more pinned buffer (constructor, then the reader's thread; buf_ is only read by that thread while it runs)
    {
        void *p = nullptr;
        if (hpn_host_malloc(ctx_, cap_ + 2 * pad_ + 64, &p) != HPN_OK) return false;
        std::lock_guard<std::mutex> lk(m_);
        buf_.push_back((uint8_t *)p);
        free_.push_back((int)buf_.size() - 1);
        return true;
    }

    size_t fill(uint8_t *dst)
    {
        if (fd_ >= 0) {  // plain file: a few threads pull disjoint pieces out of the page cache
            static const int kDefault = [] {
                const char *e = getenv("HPN_READ_THREADS");
                long n = e ? atol(e) : 6;   // (15.2 GB from the page cache: 4 / 6 / 8 / 12 threads 0.51 / 0.47 / 0.45-0.49 / 0.49-0.51 s)
                if (!e && n > usable_cpus() / text_workers_in_flight()) n = usable_cpus() / text_workers_in_flight();
                return (int)(n < 1 ? 1 : n > 32 ? 32 : n);
            }();
            const int kThreads = read_threads_ > 0 && !getenv("HPN_READ_THREADS") ? (read_threads_ > 32 ? 32 : read_threads_) : kDefault;
            const size_t piece = (cap_ / (size_t)kThreads + 4095) & ~(size_t)4095;
            std::vector<size_t> got((size_t)kThreads, 0);
            auto pull = [&](int t) {
                const size_t lo = (size_t)t * piece, hi = lo + piece < cap_ ? lo + piece : cap_;
                size_t g = 0;
                while (lo + g < hi) {
                    const ssize_t k = pread(fd_, dst + lo + g, hi - lo - g, (off_t)(pos_ + lo + g));
                    if (k <= 0) break;
                    g += (size_t)k;
                }
                got[(size_t)t] = g;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < kThreads && (size_t)t * piece < cap_; ++t) th.emplace_back(pull, t);
            pull(0);
            for (auto &t : th) t.join();
            size_t n = 0;
            for (int t = 0; t < kThreads; ++t) {  // contiguous prefix: a short piece is the end of the file
                const size_t lo = (size_t)t * piece, want = lo >= cap_ ? 0 : (lo + piece < cap_ ? piece : cap_ - lo);
                n += got[(size_t)t];
                if (got[(size_t)t] < want) break;
            }
            pos_ += n;
            return n;
        }
        size_t n = 0;
        while (n < cap_) {
            const size_t ask = cap_ - n < ((size_t)1 << 28) ? cap_ - n : (size_t)1 << 28;
            const int k = in_.read(dst + n, (unsigned)ask);
            if (k <= 0) break;
            n += (size_t)k;
        }
        return n;
    }

    void loop()
    {
        bind_thread_near(ctx_);     // (the pread threads of fill() start from this one: the pinned chunks are filled next to the device)
        for (;;) {
            int idx;
            bool more;
            {
                std::lock_guard<std::mutex> lk(m_);
                more = free_.empty() && !stop_ && (int)buf_.size() < nbuf_;
            }
            if (more && !grow()) nbuf_ = (int)buf_.size();       // (no more pinned memory: the reader goes on with what it has)
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return !free_.empty() || stop_; });
                if (stop_) break;
                idx = free_.front();
                free_.pop_front();
            }
            Chunk c;
            c.idx = idx;
            c.p = buf_[(size_t)idx] + pad_;
            c.n = fill(c.p);
            c.eof = c.n < cap_;
            {
                std::lock_guard<std::mutex> lk(m_);
                full_.push_back(c);
                if (c.eof) done_ = true;
            }
            cv_.notify_all();
            if (c.eof) return;
        }
        std::lock_guard<std::mutex> lk(m_);
        done_ = true;
    }

    hpn_ctx *ctx_;
    size_t cap_, pad_;
    int fd_ = -1;
    uint64_t pos_ = 0;   // (declared after cap_, pad_: the constructor initialises them in this order)
    int read_threads_ = 0, nbuf_ = 1;
    InStream in_;
    bool ok_ = false, handed_over_ = false;
    std::vector<uint8_t *> buf_;
    std::deque<int> free_;
    std::deque<Chunk> full_;
    std::mutex m_;
    std::condition_variable cv_;
    std::thread th_;
    bool stop_ = false, done_ = false;
};

// A slab of output to a FILE, from the ONE thread that writes that FILE.  Slabs of megabytes to a regular file go straight to
// the descriptor at the file's position: the blocks asked for first (fallocate: one writer fills a file at 14 GB/s, at 18
// with its blocks there), then plain write() -- pieces of a slab written by several threads with pwrite were SLOWER on the
// round's boxes (11.5 GB/s on 4 and 8 threads: writes to one file are serialised by the file system; scripts/micro/close_cost.cpp,
// profiles/r05/close_cost.txt) and wrong on a descriptor opened O_APPEND (`fastq_trim -o - >> all.fq`: pwrite ignores its offset
// there), so round 4's threads are gone.  Anything else through fwrite.  false: the bytes are not all in the file (ENOSPC, EIO,
// a closed pipe): the writers below remember it and the tools leave with a non-zero code.
inline bool write_slab(FILE *out, const void *p, size_t n)
{
    if (!n) return true;
    static const bool kDrop = test_env("HPN_TRIM_NOWRITE") != nullptr;       // (test-hooks builds, timing only: what the output file costs)
    if (kDrop) return true;
    struct stat sb;
    const int fd = fileno(out);
    if (n < ((size_t)4 << 20) || fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return fwrite(p, 1, n, out) == n;
    if (fflush(out) != 0) return false;
    const int fl = fcntl(fd, F_GETFL);
    const off_t at = lseek(fd, 0, SEEK_CUR);
    // (fallocate(2), not posix_fallocate: where the file system cannot do it the call fails at once -- glibc's emulation would write
    // every block with zeros first)
    if (fl >= 0 && !(fl & O_APPEND) && at >= 0) (void)fallocate(fd, 0, at, (off_t)n);
    for (size_t done = 0; done < n;) {
        const ssize_t k = write(fd, (const char *)p + done, n - done);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        done += (size_t)k;
    }
    return true;
}

// Output side of fastq_trim's fast path: pinned buffers the GPU result is copied into, written
// to the FILE in submission order by a thread of their own, so that writing chunk k overlaps
// with the copy / framing / trimming of chunk k+1.
class AsyncWriter {
public:
    AsyncWriter(hpn_ctx *ctx, FILE *out, size_t cap, int nbuf = 2) : ctx_(ctx), out_(out)
    {
        for (int i = 0; i < nbuf; ++i) {
            void *p = nullptr;
            if (hpn_host_malloc(ctx_, cap, &p) != HPN_OK) break;
            buf_.push_back(p);
            free_.push_back(i);
        }
        ok_ = (int)buf_.size() == nbuf;
        if (ok_) th_ = std::thread([this] { loop(); });
    }
    ~AsyncWriter()
    {
        finish();
        for (void *p : buf_) hpn_host_free(ctx_, p);
    }
    bool ok() const { return ok_; }
    void *acquire(int *idx)  // a buffer nobody is writing from
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !free_.empty(); });
        *idx = free_.front();
        free_.pop_front();
        return buf_[(size_t)*idx];
    }
    void submit(int idx, size_t n)  // n may be 0: the buffer just goes back
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            todo_.push_back({idx, n});
        }
        cv_.notify_all();
    }
    void finish()  // everything submitted is in the FILE when this returns
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    bool failed() const { return failed_; }     // (after finish(): a write did not go through)

private:
    void loop()
    {
        for (;;) {
            std::pair<int, size_t> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return !todo_.empty() || stop_; });
                if (todo_.empty()) return;
                job = todo_.front();
                todo_.pop_front();
            }
            if (job.second && !failed_ && !write_slab(out_, buf_[(size_t)job.first], job.second)) failed_ = true;
            {
                std::lock_guard<std::mutex> lk(m_);
                free_.push_back(job.first);
            }
            cv_.notify_all();
        }
    }
    hpn_ctx *ctx_;
    FILE *out_;
    bool ok_ = false, stop_ = false;
    std::atomic<bool> failed_{false};
    std::vector<void *> buf_;
    std::deque<int> free_;
    std::deque<std::pair<int, size_t>> todo_;
    std::mutex m_;
    std::condition_variable cv_;
    std::thread th_;
};

}  // namespace hpn
