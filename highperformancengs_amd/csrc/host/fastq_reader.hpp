// fastq_reader.hpp -- host ingest for the FASTQ tools: gzip/plain stream -> SoA batches.
//
// The reference reads EXACTLY four gzgets() lines per record into one reused
// 1024-byte buffer and never validates them (fastq_count.c:107,112-118;
// fastq_trim.c:67-89).  Everything odd it does on odd input (lines longer than
// the buffer, a missing final newline, CRLF, truncated records) follows from
// that.  LineSource::gets reproduces zlib's gzgets contract on top of gzread
// with a large buffer; the framers keep the same persistent 1024-byte buffer, so
// the bytes handed to the GPU are exactly the bytes the reference's loop would tally.
//
// Batches are fixed-capacity buffers that are allocated once (pinned when the
// caller passes hpn_host_malloc) and refilled: no growth, no page-fault churn.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <memory>
#include <string>
#include <vector>

#include "bam_reader.hpp"  // BgzfReader
#include "mgz_reader.hpp"
#include "pgz_reader.hpp"

namespace hpn {

constexpr int kLineBuf = 1024;  // fastq_count.c:107

// Files in flight at once (set by the tools): the inflate thread pools and the text chunk
// size are scaled by it.
inline int &text_workers_in_flight()
{
    static int n = 1;
    return n;
}

// A byte source: zlib's gzFile (plain text, gzip, concatenated members -- what the
// reference reads with), or, for regular files, the threaded inflaters: BGZF blocks
// (bgzip output) and concatenated gzip members.  All deliver the same byte stream; only
// the speed differs.
struct InStream {
    gzFile gz = nullptr;
    std::shared_ptr<BgzfReader> bz;
    std::shared_ptr<MgzReader> mz;
    std::shared_ptr<PgzReader> pz;
    // bytes that were already taken from the stream and are to be served again first
    // (the text front end hands an irregular stream back to the exact framer this way)
    std::shared_ptr<const std::vector<char>> pre;
    size_t pre_pos = 0;
    // exact: the stream is zlib's own gzFile with its default buffers, read in requests small enough to go through them, as
    // the reference's gzgets does (IO_stream.h:122-136): on a damaged file the bytes handed out before the error are then the
    // reference's, buffer for buffer (open_input_stream_exact)
    bool exact = false;
    bool gz_error = false;
    // a reader noticed that the stream is not sound (CRC-32, ISIZE, a data error): what it delivered may differ from what
    // gzgets hands out, and the caller reads the input again through open_input_stream_exact
    bool damaged() const { return gz_error || (pz && pz->damaged()) || (mz && mz->damaged()) || (bz && bz->error() != nullptr); }
    int read(void *dst, unsigned n)
    {
        if (pre && pre_pos < pre->size()) {
            const size_t k = pre->size() - pre_pos < n ? pre->size() - pre_pos : n;
            memcpy(dst, pre->data() + pre_pos, k);
            pre_pos += k;
            return (int)k;
        }
        if (bz) return (int)bz->read(dst, n);
        if (mz) return (int)mz->read(dst, n);
        if (pz) return (int)pz->read(dst, n);
        const int k = gzread(gz, dst, exact && n > 8192u ? 8192u : n);   // (below 2 x 8192 zlib serves a request from its own buffer)
        if (k < 0) gz_error = true;
        return k;
    }
    void close()
    {
        if (gz) gzclose(gz);
        gz = nullptr;
        bz.reset();
        mz.reset();
        pz.reset();
    }
};

// Does a second gzip member start within the first 64 MiB (or the file is small: < 4 MiB)?  Such files go to
// the member-parallel reader.  A member start is the header pattern followed by a deflate block header that
// parses (decoded strictly, with no history to refer to: compressed data that looks like a header does not get far).
inline bool gzip_has_second_member(int fd, bool any_size = false)
{
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return true;
    if (test_env("HPN_PGZ_FORCE")) return false;  // tests: any gzip file through the two-pass reader
    if (sb.st_size < (4 << 20) && !any_size) return true;  // not worth the threads
    const size_t n = (size_t)sb.st_size < ((size_t)64 << 20) ? (size_t)sb.st_size : (size_t)64 << 20;
    void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return true;
    const uint8_t *d = (const uint8_t *)m, *lim = d + n;
    bool second = false;
    for (const uint8_t *p = d + 18; !second && p + 18 < lim;) {
        p = (const uint8_t *)memchr(p, 0x1f, (size_t)(lim - 18 - p));
        if (!p) break;
        const uint8_t *body = gzip_header_end(p, lim);
        // A member start: the header (XFL is 0, 2 or 4 from every deflate writer) and a deflate stream that decodes
        // from there WITHOUT any history -- compressed bytes that merely look like a header fail that within a few
        // symbols (a match reaching in front of the stream's first byte), and what decodes must be text.
        if (body && body + 8 < lim && (p[8] == 0 || p[8] == 2 || p[8] == 4)) {
            FastInflate fi;
            static thread_local std::vector<uint8_t> out(4096 + FastInflate::kOvershoot);
            uint8_t *o = out.data();
            fi.begin(body, lim, 0, true);
            const int r = fi.run(o, out.data() + 4096, out.data());
            bool texty = r != FastInflate::kError;
            for (const uint8_t *q = out.data(); texty && q < o && q < out.data() + 4096; ++q)
                texty = (*q >= 32 && *q < 127) || *q == '\n' || *q == '\r' || *q == '\t';
            second = texty;
        }
        ++p;
    }
    munmap(m, n);
    return second;
}

// open_input_stream (IO_stream.h:122-136): a name starting with '-' (or empty) is
// stdin; anything else is open()+gzdopen(fd,"rb"), which reads plain files, gzip
// and concatenated gzip members alike.  The reference passes O_CREAT and so
// creates a missing input as an empty file; this does too (drop-in behaviour).
inline InStream open_input_stream(const char *name)
{
    InStream in;
    int fd;
    if (strncmp(name, "-", 1) == 0 || !strcmp(name, "")) {
        fd = STDIN_FILENO;
    } else {
        fd = open(name, O_CREAT | O_RDONLY, 0666);
        if (fd == -1) fprintf(stderr, "Failed to create input file (%s)", name);
        uint8_t h[18] = {0};  // BGZF: gzip member with the 'BC' extra subfield first (SAM spec 4.1)
        if (fd != -1 && pread(fd, h, 18, 0) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) &&
            h[12] == 'B' && h[13] == 'C' && !test_env("HPN_NO_BGZF")) {
            auto bz = std::make_shared<BgzfReader>();
            bz->check_crc(true);   // (text: gzread's checks; InStream::damaged)
            const long share = usable_cpus() / text_workers_in_flight();
            if (bz->open(name, getenv("HPN_BGZF_THREADS") ? 0 : (int)(share < 1 ? 1 : share > 16 ? 16 : share))) {
                close(fd);
                in.bz = bz;
                return in;
            }
        }
        long cpus = usable_cpus() / text_workers_in_flight();
        // gzip with no second member in sight: one deflate stream, inflated in parallel by the two-pass reader
        if (fd != -1 && h[0] == 0x1f && h[1] == 0x8b && !test_env("HPN_NO_PGZ") && !test_env("HPN_NO_MGZ") && (cpus >= 3 || test_env("HPN_PGZ_FORCE"))  /* the symbolic decode costs ~1.6x the plain one */ && !gzip_has_second_member(fd)) {
            auto pz = std::make_shared<PgzReader>();
            if (pz->open(name, getenv("HPN_GZ_THREADS") ? 0 : (int)(cpus > 16 ? 16 : cpus))) {
                close(fd);
                in.pz = pz;
                return in;
            }
        }
        if (fd != -1 && h[0] == 0x1f && h[1] == 0x8b && !test_env("HPN_NO_MGZ")) {  // gzip: members inflated in parallel
            auto mz = std::make_shared<MgzReader>();
            if (mz->open(name, getenv("HPN_GZ_THREADS") ? 0 : (int)(cpus < 1 ? 1 : cpus > 16 ? 16 : cpus))) {
                close(fd);
                in.mz = mz;
                return in;
            }
        }
    }
    in.gz = gzdopen(fd, "rb");   // zlib itself: default buffers and small requests, so that even a damaged stream reads as in the reference
    in.exact = true;
    return in;
}

// The reference's own reader and nothing else: gzdopen on the file, default buffers (see InStream::exact).
inline InStream open_input_stream_exact(const char *name)
{
    InStream in;
    const int fd = strncmp(name, "-", 1) == 0 || !strcmp(name, "") ? STDIN_FILENO : open(name, O_CREAT | O_RDONLY, 0666);
    if (fd == -1) fprintf(stderr, "Failed to create input file (%s)", name);
    in.gz = gzdopen(fd, "rb");
    in.exact = true;
    return in;
}

class LineSource {
public:
    explicit LineSource(InStream f, size_t cap = 4u << 20) : f_(f), buf_(cap) {}

    // zlib gzgets(file, dst, len): copy until len-1 chars or through '\n' or to the
    // end of data; NUL-terminate if anything was copied; NULL (dst untouched) if
    // nothing was.  `past` mirrors what gzeof() reports: a read ran beyond the data.
    // *n receives the number of characters copied (strlen of the result unless the
    // data holds NUL bytes).
    char *gets(char *dst, int len, size_t *n = nullptr)
    {
        if (len < 1) return nullptr;
        if (f_.exact && f_.gz && !f_.pre) {
            // zlib's own gzgets on zlib's own buffers: the reference's call, so a damaged stream ends for us where it ends for the
            // reference (gzread in pieces does not: a request that straddles two of zlib's buffers loses the part already copied)
            const z_off_t before = gztell(f_.gz);
            char *r = gzgets(f_.gz, dst, len);
            if (n) *n = (size_t)(gztell(f_.gz) - before);
            past_ = gzeof(f_.gz) != 0;
            return r;
        }
        unsigned left = (unsigned)len - 1;
        char *out = dst;
        bool eol = false;
        while (left && !eol) {
            if (have_ == 0 && !fill()) {
                // gzgets: "if (state->x.have == 0 && gz_fetch(state) == -1) return NULL" -- an ERROR of the stream ends the call at
                // once, whatever part of a line has been copied (it stays in dst, unterminated: the callers' stale-buffer rules see
                // it); a clean end of data ends the line with what there is
                if (error_) {
                    if (n) *n = (size_t)(out - dst);
                    return nullptr;
                }
                past_ = true;
                break;
            }
            size_t k = have_ < left ? have_ : left;
            const char *nl = (const char *)memchr(cur_, '\n', k);
            if (nl) {
                k = (size_t)(nl - cur_) + 1;
                eol = true;
            }
            memcpy(out, cur_, k);
            out += k, cur_ += k, have_ -= k, left -= (unsigned)k;
        }
        if (n) *n = (size_t)(out - dst);
        if (out == dst) return nullptr;
        *out = 0;
        return dst;
    }
    bool eof() const { return past_; }  // gzeof()

private:
    bool fill()
    {
        if (done_) return false;
        int n = f_.read(buf_.data(), (unsigned)buf_.size());
        if (n <= 0) {
            done_ = true;
            error_ = n < 0;   // gzread's -1: Z_DATA_ERROR / Z_BUF_ERROR (damaged or truncated gzip)
            return false;
        }
        cur_ = buf_.data();
        have_ = (size_t)n;
        return true;
    }
    InStream f_;
    std::vector<char> buf_;
    const char *cur_ = nullptr;
    size_t have_ = 0;
    bool done_ = false, past_ = false, error_ = false;
};

// One batch of records as structure of arrays (what hpn_fastq_tally / hpn_fastq_trim take).
// off[i] is relative to the start of seq / qual; off[0] = 0.
struct FastqBatch {
    typedef void *(*alloc_fn)(size_t);
    typedef void (*free_fn)(void *);

    uint8_t *seq = nullptr, *qual = nullptr;
    uint64_t *off = nullptr;
    size_t cap_bytes = 0, cap_recs = 0, nbytes = 0, nrec = 0;
    std::vector<std::string> names;  // fastq_trim only
    free_fn release = nullptr;

    // room for max_bytes bytes per array (+1 KiB: one more record always fits) and max_recs records
    bool init(size_t max_bytes, size_t max_recs, bool with_seq, alloc_fn a = malloc, free_fn f = free)
    {
        release = f;
        cap_bytes = max_bytes, cap_recs = max_recs;
        qual = (uint8_t *)a(max_bytes + kLineBuf);
        if (with_seq) seq = (uint8_t *)a(max_bytes + kLineBuf);
        off = (uint64_t *)a((max_recs + 1) * sizeof(uint64_t));
        if (!qual || !off || (with_seq && !seq)) return false;
        off[0] = 0;
        return true;
    }
    ~FastqBatch()
    {
        if (release) {
            if (qual) release(qual);
            if (seq) release(seq);
            if (off) release(off);
        }
    }
    uint64_t n() const { return nrec; }
    bool full() const { return nrec >= cap_recs || nbytes >= cap_bytes; }
    void clear()
    {
        nbytes = nrec = 0;
        names.clear();
    }
    void push(size_t len)
    {
        nbytes += len;
        off[++nrec] = nbytes;
    }
};

// count_read's framing (fastq_count.c:112-119): per record, line 2 gives
// seqLen = (uint16_t)(strlen - 1) and the first seqLen bytes of whatever the
// buffer holds after the 4th gzgets are "the quality".  With b.seq allocated it
// additionally keeps the first seqLen bytes of the buffer after the 2nd gzgets.
class CountFramer {
public:
    explicit CountFramer(const InStream &f) : src_(f) { memset(buf_, 0, sizeof buf_); }

    // Appends records until the batch is full; returns false once the stream is
    // exhausted (the batch may still hold the last records).
    // *domain_err is set when the reference would index out of its arrays.
    bool fill(FastqBatch &b, bool *domain_err)
    {
        while (!b.full()) {
            if (!src_.gets(buf_, kLineBuf)) return false;   // name line; NULL ends the loop (:112)
            src_.gets(buf_, kLineBuf);                      // sequence line (return value ignored, :113)
            const uint16_t len = (uint16_t)(strlen(buf_) - 1);  // :114, including the uint16 wrap
            if (len >= 512) {                               // SeqLen[512] overrun in the reference
                *domain_err = true;
                return false;
            }
            if (b.seq) memcpy(b.seq + b.nbytes, buf_, len);
            src_.gets(buf_, kLineBuf);                      // '+' line
            src_.gets(buf_, kLineBuf);                      // quality line
            memcpy(b.qual + b.nbytes, buf_, len);
            b.push(len);
        }
        return true;
    }

private:
    LineSource src_;
    char buf_[kLineBuf];
};

// readNextNode's framing (fastq_trim.c:67-89): buffer zeroed per record, EOF is
// tested with gzeof() AFTER the name line was read, every line loses its last
// character (the '\n', or a real character when the final newline is missing).
class TrimFramer {
public:
    // S, E: the cut the batch will get.  Only needed for reads shorter than S, see take().
    explicit TrimFramer(const InStream &f, int S = 0, int E = 0) : src_(f), S_(S < 0 ? 0 : (size_t)S), E_(E < S ? S_ : (size_t)E) {}

    bool fill(FastqBatch &b)
    {
        char buf[kLineBuf];
        while (!b.full()) {
            memset(buf, 0, sizeof buf);                     // :97
            char *p = src_.gets(buf, kLineBuf);
            if (src_.eof() || !p) return false;             // :69-70
            chop(buf);
            b.names.emplace_back(buf);
            src_.gets(buf, kLineBuf);
            chop(buf);
            const size_t ls = take(b.seq + b.nbytes, buf);
            src_.gets(buf, kLineBuf);
            src_.gets(buf, kLineBuf);
            chop(buf);
            // One offset array serves both lines.  The reference cuts each line with
            // strncpy, i.e. up to that line's own NUL; when the two lines differ in
            // length the shorter is NUL-padded to the longer and the writer prints
            // each cut as a C string, which gives the same bytes.
            const size_t lq = take(b.qual + b.nbytes, buf), lr = ls > lq ? ls : lq;
            memset(b.seq + b.nbytes + ls, 0, lr - ls);
            memset(b.qual + b.nbytes + lq, 0, lr - lq);
            b.push(lr);
        }
        return true;
    }

private:
    static void chop(char *s)
    {
        const size_t l = strlen(s);
        if (l) s[l - 1] = 0;  // (the reference writes s[-1] on an empty string)
    }
    // The line in buf -> dst; returns its length.  strncpy(dst, buf + S, E - S) of the reference
    // (:76-77, :83-84) starts S bytes into the buffer whether or not the line is that long: past
    // the line's NUL it finds what the record's earlier, longer lines left there (the buffer
    // is zeroed per record, so nothing older).  Such a read is handed on as S filler bytes
    // followed by exactly those bytes, so that the cut [S, E) of the batch yields them.
    size_t take(uint8_t *dst, const char *buf) const
    {
        const size_t l = strlen(buf);
        if (S_ <= l || S_ >= (size_t)kLineBuf) {  // (S beyond the buffer: the reference reads out of bounds)
            memcpy(dst, buf, l);
            return l;
        }
        const size_t room = (size_t)kLineBuf - 1 - S_;
        const size_t k = strnlen(buf + S_, E_ - S_ < room ? E_ - S_ : room);
        memset(dst, 'x', S_);
        memcpy(dst + S_, buf + S_, k);
        return S_ + k;
    }
    LineSource src_;
    size_t S_, E_;
};

}  // namespace hpn
