// bam_reader.hpp -- host ingest for the BAM tools: BGZF file -> header + SoA record batches.
//
// Stands where samopen / samread / bam_read1 / bgzf_read stand in the reference
// (samtools-0.1.19: sam.c:46,132; bam.c:81,191; bgzf.c:214,307,342).  Records are
// decoded into the structure-of-arrays layout of hpn_bam_batch (include/hpngs.h):
// exactly the fields the two fetch_func callbacks read (bam2depth.c:86-110,
// bam_sliding_count.c:93-124): tid, pos, flag, l_qseq, CIGAR words, packed sequence.
//
// BGZF blocks are independent deflate streams (<= 64 KiB each), so they are inflated
// by a small pool of threads and handed to the decoder in file order;
// the reference inflates them one after the other on the calling thread.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cpus.hpp"
#include "fast_inflate.hpp"
#include "hpngs.h"

namespace hpn {

class BgzfReader {
public:
    // threads: inflate workers (0 = the usable CPUs minus three -- file reader, decoder, GPU runtime --
    // overridable with HPN_BGZF_THREADS)
    // start: file offset of the BGZF block to begin with (a virtual offset >> 16)
    bool open(const char *path, int threads = 0, uint64_t start = 0)
    {
        fp_ = fopen(path, "rb");
        if (!fp_) return false;
        if (start && fseeko(fp_, (off_t)start, SEEK_SET) != 0) return false;
        if (threads <= 0) {
            const char *e = getenv("HPN_BGZF_THREADS");
            long n = e ? atol(e) : usable_cpus() - 3;
            threads = (int)(n < 1 ? 1 : n > 64 ? 64 : n);
        }
        slots_.resize((size_t)threads * 4);
        io_ = std::thread([this] { io_loop(); });
        for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work_loop(); });
        return true;
    }
    ~BgzfReader()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (io_.joinable()) io_.join();
        for (auto &t : workers_) t.join();
        if (fp_) fclose(fp_);
    }
    // Read exactly n bytes of the uncompressed stream; returns bytes read (< n at EOF).
    size_t read(void *dst, size_t n)
    {
        uint8_t *out = (uint8_t *)dst;
        size_t got = 0;
        while (got < n) {
            if (!cur_ || pos_ == cur_->data.size()) {
                if (!next_block()) break;
                continue;
            }
            size_t k = cur_->data.size() - pos_;
            if (k > n - got) k = n - got;
            memcpy(out + got, cur_->data.data() + pos_, k);
            pos_ += k, got += k;
        }
        return got;
    }
    // The next n bytes in place (no copy), or nullptr when they straddle a block boundary or
    // the stream ends first.  The pointer stays valid until the next read / peek call.
    const uint8_t *peek(size_t n)
    {
        if ((!cur_ || pos_ == cur_->data.size()) && !next_block()) return nullptr;
        return cur_->data.size() - pos_ >= n ? cur_->data.data() + pos_ : nullptr;
    }
    void skip(size_t n) { pos_ += n; }  // only what peek() just showed
    const char *error() const { return err_; }
    // before open(): also verify every block's CRC-32 (a block that fails ends the stream with error() set, like a data error)
    void check_crc(bool on) { check_crc_ = on; }

private:
    enum State { kFree, kRaw, kBusy, kDone };
    struct Slot {
        State st = kFree;
        std::vector<uint8_t> raw, data;
        uint32_t isize = 0;
        bool bad = false;
    };
    bool check_crc_ = false;

    // file order = slot order (round robin): the reader fills slot w_, the consumer takes slot r_
    void io_loop()
    {
        for (;;) {
            Slot *s;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || slots_[w_ % slots_.size()].st == kFree; });
                if (stop_) return;
                s = &slots_[w_ % slots_.size()];
            }
            uint8_t h[18];
            const size_t k = fread(h, 1, 18, fp_);
            bool eof = k == 0, bad = false;
            if (!eof) {
                if (k != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') {
                    bad = true;
                } else {
                    const unsigned bsize = (h[16] | (h[17] << 8)) + 1u;
                    const unsigned xlen = h[10] | (h[11] << 8);
                    if (bsize < 26u || xlen < 6u || bsize < xlen + 20u) {
                        bad = true;
                    } else if (s->raw.resize(bsize - 18), fread(s->raw.data(), 1, s->raw.size(), fp_) != s->raw.size()) {
                        bad = true;
                    } else {
                        memcpy(&s->isize, s->raw.data() + s->raw.size() - 4, 4);
                        // drop the rest of the extra field: raw = deflate data + crc32 + isize
                        s->raw.erase(s->raw.begin(), s->raw.begin() + (xlen - 6));
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                if (eof || bad) {
                    end_at_ = w_;  // no block at index w_: end of stream (or error)
                    if (bad) err_ = "not a BGZF block / truncated file";
                } else {
                    s->st = kRaw;
                    ++w_;
                }
            }
            cv_.notify_all();
            if (eof || bad) return;
        }
    }
    void work_loop()
    {
        z_stream zs;  // one inflate state per worker, reset per block
        memset(&zs, 0, sizeof zs);
        const bool zs_ok = inflateInit2(&zs, -15) == Z_OK;
        FastInflate fi;
        const bool fast_ok = !(test_env("HPN_FAST_INFLATE") && test_env("HPN_FAST_INFLATE")[0] == '0');
        for (;;) {
            Slot *s = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] {
                    if (stop_) return true;
                    for (auto &x : slots_)
                        if (x.st == kRaw) return true;
                    return false;
                });
                if (stop_) break;
                for (auto &x : slots_)
                    if (x.st == kRaw) {
                        s = &x;
                        break;
                    }
                s->st = kBusy;
            }
            s->bad = false;
            bool done = false;
            if (s->isize && fast_ok) {  // the quick decoder first; zlib below if it declines
                s->data.resize((size_t)s->isize + 1 + FastInflate::kOvershoot);
                uint8_t *d = s->data.data();
                fi.begin(s->raw.data(), s->raw.data() + s->raw.size() - 8);
                done = fi.run(d, s->data.data() + s->isize + 1, s->data.data()) == FastInflate::kDone && d == s->data.data() + s->isize;
            }
            s->data.resize(s->isize);
            if (s->isize && !done) {
                if (!zs_ok || inflateReset(&zs) != Z_OK) s->bad = true;
                else {
                    zs.next_in = s->raw.data();
                    zs.avail_in = (uInt)(s->raw.size() - 8);
                    zs.next_out = s->data.data();
                    zs.avail_out = s->isize;
                    // (a stream that ends short of ISIZE is a damaged block too: its last bytes would be whatever the buffer held)
                    if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) s->bad = true;
                }
            }
            if (!s->isize && s->raw.size() > 8 + 2) {   // an empty block is the two bytes 03 00; anything longer must not hide data
                uint8_t one;
                if (!zs_ok || inflateReset(&zs) != Z_OK) s->bad = true;
                else {
                    zs.next_in = s->raw.data(), zs.avail_in = (uInt)(s->raw.size() - 8), zs.next_out = &one, zs.avail_out = 1;
                    if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 1) s->bad = true;
                }
            }
            // Where the bytes are TEXT the reference reads them through gzread, which checks every member's CRC-32 and hands out
            // nothing of a member that fails (the BAM tools' reader, samtools' bgzf.c, checks none: check_crc_ stays off there)
            if (!s->bad && check_crc_) {
                uint32_t want;
                memcpy(&want, s->raw.data() + s->raw.size() - 8, 4);
                if (crc32_fast(0, s->data.data(), s->data.size()) != want) s->bad = true;
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                s->st = kDone;
            }
            cv_.notify_all();
        }
        if (zs_ok) inflateEnd(&zs);
    }
    bool next_block()
    {
        std::unique_lock<std::mutex> lk(m_);
        if (cur_) {  // give the consumed slot back
            cur_->st = kFree;
            cur_ = nullptr;
            ++r_;
            cv_.notify_all();
        }
        for (;;) {
            Slot &s = slots_[r_ % slots_.size()];
            cv_.wait(lk, [&] { return s.st == kDone || (end_at_ != ~0ull && r_ >= end_at_); });
            if (s.st != kDone) return false;  // end of stream
            if (s.bad) {
                err_ = "inflate failed";
                return false;
            }
            cur_ = &s;
            pos_ = 0;
            if (!s.data.empty()) return true;
            s.st = kFree;  // empty block (the EOF marker): skip it
            cur_ = nullptr;
            ++r_;
            cv_.notify_all();
        }
    }

    FILE *fp_ = nullptr;
    std::vector<Slot> slots_;
    std::thread io_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    uint64_t w_ = 0, r_ = 0, end_at_ = ~0ull;
    bool stop_ = false;
    Slot *cur_ = nullptr;
    size_t pos_ = 0;
    const char *err_ = nullptr;
};

struct BamHeader {
    std::vector<std::string> target_name;
    std::vector<uint32_t> target_len;
    int32_t n_targets() const { return (int32_t)target_name.size(); }
};

struct BamBatch {
    std::vector<int32_t> tid, pos, l_qseq;
    std::vector<uint32_t> flag, cigar_off{0}, cigar;
    std::vector<uint64_t> seq_off{0};
    std::vector<uint8_t> seq4;
    uint64_t n() const { return tid.size(); }
    void clear()
    {
        tid.clear(), pos.clear(), l_qseq.clear(), flag.clear(), cigar.clear(), seq4.clear();
        cigar_off.assign(1, 0), seq_off.assign(1, 0);
    }
    hpn_bam_batch view()
    {
        seq4.reserve(seq4.size() + 16);  // aligned 16-byte reads may look past the end
        hpn_bam_batch b;
        b.n = n();
        b.tid = tid.data(), b.pos = pos.data(), b.flag = flag.data(), b.l_qseq = l_qseq.data();
        b.cigar_off = cigar_off.data(), b.cigar = cigar.data();
        b.seq_off = seq_off.data(), b.seq4 = seq4.data();
        return b;
    }
};

class BamReader {
public:
    // Records from a BGZF virtual offset on (block offset << 16 | offset inside the inflated block), as a .bai gives it:
    // what bam_fetch's iterator does with the index (bam_index.c:682) -- no header is read here.
    bool open_at(const char *path, uint64_t voffset)
    {
        if (!z_.open(path, 0, voffset >> 16)) return false;
        size_t skip = (size_t)(voffset & 0xffff);
        while (skip) {
            uint8_t tmp[4096];
            const size_t k = z_.read(tmp, skip < sizeof tmp ? skip : sizeof tmp);
            if (!k) return false;
            skip -= k;
        }
        return true;
    }

    // samopen(fn, "rb") + bam_header_read (bam.c:81)
    bool open(const char *path, BamHeader &h)
    {
        if (!z_.open(path)) return false;
        char magic[4];
        int32_t l_text, n_ref;
        if (z_.read(magic, 4) != 4 || memcmp(magic, "BAM\1", 4)) return false;
        if (z_.read(&l_text, 4) != 4) return false;
        std::vector<char> text((size_t)l_text);
        if (z_.read(text.data(), text.size()) != text.size()) return false;
        if (z_.read(&n_ref, 4) != 4) return false;
        for (int32_t i = 0; i < n_ref; ++i) {
            int32_t l_name, l_ref;
            if (z_.read(&l_name, 4) != 4) return false;
            std::vector<char> nm((size_t)l_name);
            if (z_.read(nm.data(), nm.size()) != nm.size()) return false;
            if (z_.read(&l_ref, 4) != 4) return false;
            h.target_name.emplace_back(nm.data());
            h.target_len.push_back((uint32_t)l_ref);
        }
        return true;
    }

    // bam_read1 (bam.c:191): append one record; false at end of file.
    // want_seq: keep the 4-bit sequence (bam_sliding_count); depth needs CIGAR only.
    bool next(BamBatch &b, bool want_seq)
    {
        if (pending_) {  // record looked at by peek_tid
            pending_ = false;
            append(b, want_seq);
            return true;
        }
        if (!fetch()) return false;
        append(b, want_seq);
        return true;
    }
    // tid of the record `next` would deliver, without consuming it (INT32_MIN at EOF)
    int32_t peek_tid()
    {
        if (!pending_) {
            if (!fetch()) return INT32_MIN;
            pending_ = true;
        }
        int32_t t;
        memcpy(&t, p_, 4);
        return t;
    }

private:
    // Next record -> p_ (the bytes after block_size).  A record that lies inside one BGZF block
    // (all but one per 64 KiB) is parsed where the inflater put it; only a straddling record
    // is assembled in rec_.
    bool fetch()
    {
        int32_t block_size;
        bool have = false;
        if (corrupt_) return false;
        if (const uint8_t *h = z_.peek(4)) {
            memcpy(&block_size, h, 4);
            if (block_size < 32) return corrupt();
            if (const uint8_t *q = z_.peek(4 + (size_t)block_size)) {
                z_.skip(4 + (size_t)block_size);
                p_ = q + 4;
                have = true;
            }
        }
        if (!have) {
            if (z_.read(&block_size, 4) != 4) return false;
            if (block_size < 32) return corrupt();
            rec_.resize((size_t)block_size);
            if (z_.read(rec_.data(), rec_.size()) != rec_.size()) return false;
            p_ = rec_.data();
        }
        // bam_read1 (bam.c:191) trusts l_read_name / n_cigar_op / l_seq and the reference's callbacks then read
        // past the record; here a record whose fields do not fit its block_size ends the file with a message
        uint32_t bin_mq_nl, flag_nc, l_seq;
        memcpy(&bin_mq_nl, p_ + 8, 4), memcpy(&flag_nc, p_ + 12, 4), memcpy(&l_seq, p_ + 16, 4);
        if (l_seq > 0x7fffffffu || 32ull + (bin_mq_nl & 0xff) + 4ull * (flag_nc & 0xffff) + (((uint64_t)l_seq + 1) >> 1) + l_seq >
                                       (uint64_t)block_size)
            return corrupt();
        return true;
    }
    bool corrupt()
    {
        if (!corrupt_) fprintf(stderr, "[hpn] corrupt BAM record (fields do not fit block_size): reading stops here\n");
        corrupt_ = true;
        return false;
    }
    void append(BamBatch &b, bool want_seq)
    {
        // bam1_core_t on disk (bam.h:178-187): refID, pos, bin_mq_nl, flag_nc, l_seq, ...
        const uint8_t *p = p_;
        int32_t tid, pos, l_seq;
        uint32_t bin_mq_nl, flag_nc;
        memcpy(&tid, p, 4), memcpy(&pos, p + 4, 4), memcpy(&bin_mq_nl, p + 8, 4), memcpy(&flag_nc, p + 12, 4);
        memcpy(&l_seq, p + 16, 4);
        const uint32_t l_name = bin_mq_nl & 0xff, n_cigar = flag_nc & 0xffff, flag = flag_nc >> 16;
        const uint8_t *cig = p + 32 + l_name;
        b.tid.push_back(tid), b.pos.push_back(pos), b.flag.push_back(flag), b.l_qseq.push_back(l_seq);
        const size_t c0 = b.cigar.size();
        b.cigar.resize(c0 + n_cigar);
        if (n_cigar) memcpy(b.cigar.data() + c0, cig, 4 * (size_t)n_cigar);
        b.cigar_off.push_back((uint32_t)b.cigar.size());
        if (want_seq) {
            const uint8_t *sq = cig + 4 * (size_t)n_cigar;
            b.seq4.insert(b.seq4.end(), sq, sq + (size_t)((l_seq + 1) >> 1));
        }
        b.seq_off.push_back(b.seq4.size());
    }
    BgzfReader z_;
    std::vector<uint8_t> rec_;
    const uint8_t *p_ = nullptr;
    bool pending_ = false;
    bool corrupt_ = false;
};

}  // namespace hpn
