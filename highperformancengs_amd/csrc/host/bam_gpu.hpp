// bam_gpu.hpp -- BGZF ingest (BAM, bgzip-compressed FASTQ) with the inflate and the record walk on the GPU.
//
// The host-side path (bam_reader.hpp) inflates BGZF blocks on a thread pool and decodes records on
// one thread: 84 % of bam2depth's run time once the per-record work is on the GPU, and capped by
// the host's cores.  Here the host only moves bytes: a reader thread preads the compressed file
// into pinned chunks, the 18-byte block headers are walked to build the block table (one hop
// per ~20 KB), and the device inflates the blocks (hpn_bgzf_inflate_dev) and indexes the records
// where they lie (hpn_bam_raw_index_dev).  The BAM header is parsed on the host from the first
// blocks (zlib), which also tells where the first record starts.
//
// Works for files whose BGZF blocks start at record boundaries (everything samtools writes,
// bam.c:238); otherwise, and on any damaged block, next() reports failure and the tool runs
// the host path instead.
#pragma once
#include <zlib.h>

#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "bam_reader.hpp"
#include "text_stream.hpp"

namespace hpn {

// One batch of whole BGZF blocks as the host sees it: the block table, the block carried over from the previous
// chunk (pageable copy) and the span of the pinned chunk that follows it.
struct BgzfParsed {
    std::vector<hpn_bgzf_block> blocks;
    std::vector<uint32_t> crcs;  // the blocks' CRC-32 fields (checked where the stream is TEXT: gzread, which the reference reads FASTQ through, checks them)
    std::vector<uint8_t> carry;
    const uint8_t *body = nullptr;
    size_t body_len = 0;
    uint64_t out_bytes = 0;      // inflated size of the batch
    uint32_t first_off = 0;      // offset of the first record inside block 0
};

// The compressed side of ONE inflate launch, prepared ahead of time: the bytes of up to `rounds` chunks on the device (copied
// through a context of their own, so the copies run beside the kernels of the launch before) and the launch's block table.
struct BgzfStage {
    void *d_comp = nullptr;
    size_t cap = 0, at_comp = 0;
    std::vector<hpn_bgzf_block> pieces;
    std::vector<uint32_t> crcs;
    uint64_t at_out = 0;
    uint32_t first_off = 0;
    bool have_first = false, eof = false;
    int result = 1;          // 1 = blocks (or an empty batch), 0 = the file's end, -1 = not decodable here
    int state = 0;           // 0 free (the producer may fill it), 1 ready (next() may take it)
};

// One context's side of the ingest: device buffers, copies, inflate, record index.  Several of these can take the
// batches of one file in turn (bam_multi.hpp); a BgzfGpuStream owns one.
class BgzfDevice {
public:
    explicit BgzfDevice(hpn_ctx *ctx = nullptr) : ctx_(ctx) {}
    BgzfDevice(const BgzfDevice &) = delete;
    ~BgzfDevice()
    {
        if (ctx_) {
            hpn_dev_free(ctx_, d_comp_), hpn_dev_free(ctx_, d_blocks_), hpn_dev_free(ctx_, d_out_), hpn_dev_free(ctx_, d_status_);
            if (d_carry_) hpn_dev_free(ctx_, d_carry_);
            hpn_host_free(ctx_, h_blocks_);
        }
    }
    void bind(hpn_ctx *ctx) { ctx_ = ctx; }
    hpn_ctx *ctx() const { return ctx_; }
    const uint8_t *d_raw() const { return d_out_ ? (const uint8_t *)d_out_ + pad_front_ : nullptr; }
    // writable room in front of and behind the inflated bytes (text batches framed as pieces: the byte before, the 4 KiB after)
    void set_out_pad(size_t front, size_t back) { pad_front_ = front, pad_back_ = back; }
    // Records may run across block ends (htsjdk-written BAMs; bgzf_read hides the blocks from bam_read1, bgzf.c:342) and so
    // across the END of a launch: the bytes of the unfinished record (hpn_raw_info.tail_bytes) are kept and laid in front of
    // the next launch's stream, whose blocks move up by as much -- the stream starts at a record again (kernels/bam_raw.hip).
    // Switched on by the stream that owns this device side and feeds it IN FILE ORDER; batches dealt to several devices
    // (BgzfFanout) cannot carry and give such a file back to the one-stream route.
    void allow_carry(bool on) { carry_ok_ = on, carry_n_ = 0; }
    void drop_carry() { carry_n_ = 0; }                  // (behind a seek: the next launch starts at a record of its own)
    size_t carried() const { return carry_n_; }          // != 0 at the end of the file: it ends inside a record
    static constexpr size_t kMaxCarry = (size_t)4 << 20; // the longest unfinished record kept from launch to launch

    // Bytes and table to the device, inflate, then (records) index or (text) check every block.
    // 1 = ok, -1 = not decodable here.  On return the pinned chunk `pb.body` points into is free again.
    int run(const BgzfParsed &pb, bool text_mode, hpn_raw_info *info)
    {
        memset(info, 0, sizeof *info);
        const size_t nb = pb.blocks.size();
        if (!nb) return 1;
        const size_t comp_bytes = pb.carry.size() + pb.body_len;
        if (!reserve<uint8_t>(d_comp_, cap_comp_, comp_bytes + 64)) return -1;
        if (!pb.carry.empty()) {
            if (hpn_memcpy_h2d(ctx_, d_comp_, pb.carry.data(), pb.carry.size()) != HPN_OK) return -1;
            if (hpn_ctx_sync(ctx_) != HPN_OK) return -1;  // pageable source
        }
        if (pb.body_len && hpn_memcpy_h2d(ctx_, (uint8_t *)d_comp_ + pb.carry.size(), pb.body, pb.body_len) != HPN_OK) return -1;
        // (the sync inside the index call / the status read-back also covers the copies out of the pinned chunk)
        return launch(d_comp_, pb.blocks.data(), pb.crcs.data(), nb, pb.out_bytes, pb.first_off, text_mode, false, info);
    }

    // The same in pieces: several chunks of the file under ONE inflate launch.  A launch of one round of the chip's ~5,000
    // decoder waves (an 88 MB chunk holds ~4,500 blocks) ends with its slowest block -- on a 10 GB BAM the 116 launches ran at
    // 23.6 GB/s of inflated bytes where one launch over all blocks reaches 39.6 (rocprofv3: profiles/r03/kernel_stats_c4_*.csv) --;
    // with four rounds under a launch the waves that finish early take the next blocks.  The pieces' compressed bytes are
    // copied as they come (the pinned chunk is free again when add() returns), so the pinned memory stays three chunks.
    bool begin(size_t comp_room)                     // comp_room: compressed bytes the launch may hold
    {
        pieces_.clear(), crcs_.clear();
        at_comp_ = 0, at_out_ = 0, first_off_ = 0, have_first_ = false;
        return reserve<uint8_t>(d_comp_, cap_comp_, comp_room + 64);
    }
    bool add(const BgzfParsed &pb)
    {
        const size_t comp_bytes = pb.carry.size() + pb.body_len;
        if (!pb.blocks.empty() && !have_first_) first_off_ = pb.first_off, have_first_ = true;
        if (at_comp_ + comp_bytes + 64 > cap_comp_) return false;
        if (!pb.carry.empty() && hpn_memcpy_h2d(ctx_, (uint8_t *)d_comp_ + at_comp_, pb.carry.data(), pb.carry.size()) != HPN_OK) return false;
        if (pb.body_len && hpn_memcpy_h2d(ctx_, (uint8_t *)d_comp_ + at_comp_ + pb.carry.size(), pb.body, pb.body_len) != HPN_OK) return false;
        for (hpn_bgzf_block b : pb.blocks) {
            b.in_off += at_comp_, b.out_off += at_out_;
            pieces_.push_back(b);
        }
        crcs_.insert(crcs_.end(), pb.crcs.begin(), pb.crcs.end());
        at_comp_ += comp_bytes, at_out_ += pb.out_bytes;
        return hpn_ctx_sync(ctx_) == HPN_OK;           // (pageable carry, and the pinned chunk goes back to the reader)
    }
    size_t blocks_added() const { return pieces_.size(); }
    int finish(bool text_mode, hpn_raw_info *info)
    {
        return launch(d_comp_, pieces_.data(), crcs_.data(), pieces_.size(), at_out_, first_off_, text_mode, carry_ok_, info);
    }

    // The launch a BgzfStage holds: block table to the device, inflate, then (records) index or (text) check every block.
    int finish_stage(const BgzfStage &st, bool text_mode, hpn_raw_info *info)
    {
        return launch(st.d_comp, st.pieces.data(), st.crcs.data(), st.pieces.size(), st.at_out, st.first_off, text_mode, carry_ok_, info);
    }

private:
    // Table to the device, inflate behind the carried bytes, index (records) or check (text); then keep the new tail.
    int launch(const void *d_comp, const hpn_bgzf_block *table, const uint32_t *crcs, size_t nb, uint64_t out_bytes, uint32_t first_off,
               bool text_mode, bool may_carry, hpn_raw_info *info)
    {
        memset(info, 0, sizeof *info);
        if (!nb) return 1;
        const size_t h = may_carry && !text_mode ? carry_n_ : 0;      // bytes of the record the launch before ended in
        if (!reserve<hpn_bgzf_block>(d_blocks_, cap_blocks_, nb) ||
            !reserve<uint8_t>(d_out_, cap_out_, h + out_bytes + 64 + pad_front_ + pad_back_) || !reserve<uint32_t>(d_status_, cap_status_, nb))
            return -1;
        if (nb > h_blocks_cap_) {
            if (h_blocks_) hpn_host_free(ctx_, h_blocks_);
            h_blocks_cap_ = nb + nb / 2 + 1024;
            if (hpn_host_malloc(ctx_, h_blocks_cap_ * sizeof(hpn_bgzf_block), &h_blocks_) != HPN_OK) return -1;
        }
        hpn_bgzf_block *hb = (hpn_bgzf_block *)h_blocks_;
        memcpy(hb, table, nb * sizeof(hpn_bgzf_block));
        if (h)
            for (size_t i = 0; i < nb; ++i) hb[i].out_off += h;
        uint8_t *stream = (uint8_t *)d_out_ + pad_front_;
        if (hpn_memcpy_h2d(ctx_, d_blocks_, h_blocks_, nb * sizeof(hpn_bgzf_block)) != HPN_OK) return -1;
        if (h && hpn_memcpy_d2d(ctx_, stream, d_carry_, h) != HPN_OK) return -1;
        if (hpn_bgzf_inflate_dev(ctx_, (const uint8_t *)d_comp, (const hpn_bgzf_block *)d_blocks_, nb, stream, (uint32_t *)d_status_) != HPN_OK)
            return -1;
        if (text_mode) {  // no records: wait, check every block's status
            status_.resize(nb);
            if (hpn_memcpy_d2h(ctx_, status_.data(), d_status_, nb * sizeof(uint32_t)) != HPN_OK || hpn_ctx_sync(ctx_) != HPN_OK) return -1;
            for (uint32_t st : status_)
                if (st) return -1;
            // Text (a bgzip-compressed FASTQ) is what the reference reads through gzread, and gzread checks every member's CRC-32:
            // a block whose bytes are not what its trailer says ends the stream there for the reference, so this route hands the
            // file back (the host readers then stop where zlib stops).  samtools' BGZF reader checks no CRC: records are not
            // checked here either.  (Round 6: found by scripts/soak_fastq_tools.py -- a flipped bit inside a literal passed.)
            spans_.clear();
            for (size_t i = 0; i < nb; ++i) spans_.push_back(hpn_span{hb[i].out_off, hb[i].out_len});
            got_crcs_.resize(nb);
            if (hpn_crc32_dev(ctx_, stream, spans_.data(), (uint32_t)nb, got_crcs_.data()) != HPN_OK) return -1;
            for (size_t i = 0; i < nb; ++i)
                if (got_crcs_[i] != crcs[i]) return -1;
            info->n_records = out_bytes;
            return 1;
        }
        if (hpn_bam_raw_index_dev(ctx_, stream, (const hpn_bgzf_block *)d_blocks_, nb, h ? 0u : first_off, (const uint32_t *)d_status_,
                                  info) != HPN_OK)
            return -1;
        if (info->flags & 3u) return -1;
        carry_n_ = 0;
        if (info->tail_bytes) {
            if (!may_carry || info->tail_bytes > kMaxCarry) return -1;
            if (!d_carry_ && hpn_dev_malloc(ctx_, kMaxCarry, &d_carry_) != HPN_OK) return -1;
            // (in stream order: before the next launch's inflate writes over the stream, after this launch's index)
            if (hpn_memcpy_d2d(ctx_, d_carry_, stream + h + out_bytes - info->tail_bytes, info->tail_bytes) != HPN_OK) return -1;
            carry_n_ = info->tail_bytes;
        }
        return 1;
    }
    std::vector<hpn_bgzf_block> pieces_;
    std::vector<uint32_t> crcs_, got_crcs_;
    std::vector<hpn_span> spans_;
    size_t at_comp_ = 0;
    uint64_t at_out_ = 0;
    uint32_t first_off_ = 0;
    bool have_first_ = false;
    template <typename T>
    bool reserve(void *&p, size_t &cap, size_t n)
    {
        if (n * sizeof(T) <= cap) return true;
        if (p) hpn_dev_free(ctx_, p);
        p = nullptr, cap = 0;
        const size_t want = n * sizeof(T) + n * sizeof(T) / 4 + 4096;
        if (hpn_dev_malloc(ctx_, want, &p) != HPN_OK) return false;
        cap = want;
        return true;
    }
    hpn_ctx *ctx_;
    size_t pad_front_ = 0, pad_back_ = 0;
    std::vector<uint32_t> status_;
    void *d_comp_ = nullptr, *d_blocks_ = nullptr, *d_out_ = nullptr, *d_status_ = nullptr, *h_blocks_ = nullptr, *d_carry_ = nullptr;
    size_t carry_n_ = 0;
    bool carry_ok_ = false;
    size_t cap_comp_ = 0, cap_blocks_ = 0, cap_out_ = 0, cap_status_ = 0, h_blocks_cap_ = 0;
};

class BgzfGpuStream {
public:
    ~BgzfGpuStream()
    {
        halt_producer();
        pump_.reset();
        for (BgzfStage &st : stage_)
            if (st.d_comp && up_ctx_) hpn_dev_free(up_ctx_, st.d_comp);
        if (up_ctx_) hpn_ctx_destroy(up_ctx_);
    }

    // Parses the BAM header (host, zlib) and positions the stream at the first record.
    // nbuf: pinned chunks of read-ahead (one more than the contexts that take batches)
    // chunk: bytes per pinned chunk (0: default_chunk())
    bool open(hpn_ctx *ctx, const char *path, BamHeader &hdr, int nbuf = 3, size_t chunk = 0)
    {
        ctx_ = ctx;
        dev_.bind(ctx);
        FILE *f = fopen(path, "rb");
        if (!f) return false;
        std::vector<uint8_t> text, raw;
        uint64_t file_off = 0;
        bool ok = false;
        size_t need = 12;
        for (;;) {
            uint8_t h[18];
            if (fread(h, 1, 18, f) != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') break;
            const uint32_t bsize = (h[16] | (h[17] << 8)) + 1u, xlen = h[10] | (h[11] << 8);
            if (bsize < xlen + 20u || xlen < 6u) break;
            raw.resize(bsize - 18);
            if (fread(raw.data(), 1, raw.size(), f) != raw.size()) break;
            uint32_t isize;
            memcpy(&isize, raw.data() + raw.size() - 4, 4);
            const size_t before = text.size();
            text.resize(before + isize);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) break;
            zs.next_in = raw.data() + (xlen - 6), zs.avail_in = (uInt)(raw.size() - (xlen - 6) - 8);
            zs.next_out = text.data() + before, zs.avail_out = isize;
            const int rc = isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || (isize && zs.avail_out != 0)) break;
            const int st = parse_header(text, hdr, &need);
            if (st < 0) break;
            if (st > 0) {  // complete: the first record is at text[need]
                ok = true;
                first_off_ = (uint32_t)(need - before);  // inside this block ...
                start_ = file_off;                        // ... which starts here in the file
                if (need == text.size()) first_off_ = 0, start_ = file_off + bsize;  // the header ends with its block
                break;
            }
            file_off += bsize;
        }
        fclose(f);
        if (!ok) return false;
        chunk_ = default_chunk(chunk);
        pump_.reset(new TextPump(ctx, path, chunk_, nbuf, true));
        if (!pump_->ok()) return false;
        skip_ = start_;
        dev_.allow_carry(true);                 // one stream, batches in file order: records may run from one launch into the next
        {   // launches of 44 chunks only where the file is long enough to pay for them: a launch twice as long also ends the file
            // with twice the work behind the last upload and starts it with a longer ramp (+ 20 - 50 ms of wall on a 10.6 GB
            // file, measured: profiles/r06/tools_c.txt), while the decoders' better rate is worth 4 % of the device time
            struct stat sb;
            if (!rounds_env() && stat(path, &sb) == 0 && (uint64_t)sb.st_size < ((uint64_t)32 << 30)) rounds_ = 22;
        }
        return true;
    }

    // bgzip-compressed text (a .fastq.gz written by bgzip): batches of inflated bytes on the device,
    // no records to index.  next() then fills only info->n_records with the batch's byte count.
    bool open_text(hpn_ctx *ctx, const char *path, int nbuf = 3, size_t chunk = 0)
    {
        ctx_ = ctx;
        dev_.bind(ctx);
        text_mode_ = true;
        chunk_ = default_chunk(chunk);
        pump_.reset(new TextPump(ctx, path, chunk_, nbuf, true));
        return pump_->ok();
    }
    bool at_eof() const { return ahead_ ? last_eof_ : eof_ && carry_.empty(); }   // the batch next() returned last ends the file

    // Continue at a BGZF virtual offset (compressed block offset << 16 | offset inside the inflated block), as the
    // .bai gives it for the first record of a target: the same buffers, a fresh read-ahead.
    bool seek(uint64_t voffset)
    {
        halt_producer();
        if (!pump_ || !pump_->restart(voffset >> 16)) return false;
        first_off_ = (uint32_t)(voffset & 0xffff), skip_ = 0, eof_ = false;
        carry_.clear();
        dev_.drop_carry();
        if (!rounds_env()) rounds_ = 1;
        return true;
    }

    // Read ahead from now on (optional; next() does it otherwise): the upload context, the first chunks and their copies run
    // beside whatever the caller does between open() and its first next() -- output files, hpn_depth_begin's buffers (round 5:
    // bam2depth spent 76 ms there with the file untouched).  Not for a stream that will seek() first.
    void start()
    {
        if (ahead_enabled() && !producer_.joinable() && !ended_) (void)start_producer();
    }

    const uint8_t *d_raw() const { return dev_.d_raw(); }
    // chunks under one inflate launch, for a caller that knows better than the default (HPN_BAM_ROUNDS overrides both)
    void prefer_rounds(int n) { if (!rounds_env() && n > 0 && !producer_.joinable()) rounds_ = n; }   // (before start())

    // Next batch of records, inflated and indexed on the device: 1 = ok (info filled in; a batch may
    // be empty), 0 = end of file, -1 = not decodable here (the caller switches to the host path).
    // The compressed bytes of the batch AFTER this one are read, parsed and copied to the device by a producer thread through a
    // context of its own while this batch is inflated and used (round 4: the copies were a fifth of the ingest's device time,
    // in line with the kernels on one stream).  HPN_BAM_AHEAD=0: everything on the caller's thread and context, as before.
    int next(hpn_raw_info *info)
    {
        memset(info, 0, sizeof *info);
        if (!ahead_enabled()) {
            ahead_ = false;
            if (!dev_.begin((size_t)rounds_ * (chunk_ + 65536 + 64))) return -1;
            BgzfStage none;
            const int r = gather(nullptr, none);
            if (r != 1) return r;
            if (!dev_.blocks_added()) return eof_ ? (dev_.carried() ? -1 : 0) : 1;   // (a batch may be empty; a file may not end inside a record)
            return dev_.finish(text_mode_, info);
        }
        if (ended_) return final_;                       // (the end, or a failure, has been handed out: it stays)
        if (!producer_.joinable() && !start_producer()) return -1;
        ahead_ = true;
        BgzfStage &st = stage_[turn_ & 1];
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return st.state == 1; });
        }
        int r = st.result;
        last_eof_ = st.eof;
        if (r == 1 && !st.pieces.empty()) r = dev_.finish_stage(st, text_mode_, info);
        stamp("BGZF: launch inflated and indexed");
        if (r == 0 && dev_.carried()) r = -1;            // the file ends inside a record: the host route reports it as the reference does
        {   // the launch has read the stage's bytes (finish_stage waits for its kernels): the producer may fill it again
            std::lock_guard<std::mutex> lk(mu_);
            st.state = 0;
        }
        cv_.notify_all();
        ++turn_;
        if (r != 1) ended_ = true, final_ = r;
        return r;
    }

private:
    static bool ahead_enabled()
    {
        const char *e = test_env("HPN_BAM_AHEAD");
        return !(e && e[0] == '0');
    }
    // Up to rounds_ chunks of the file: parsed, their bytes on the device.  st == nullptr: into dev_ through the caller's context
    // (BgzfDevice::add); else into the stage through up_ctx_.  1 = ok, -1 = not decodable / truncated.
    int gather(BgzfStage *st, BgzfStage &)
    {
        // (the FIRST launch of a file read front to back is six chunks: what waits for the first records -- bam2depth's first
        // target, the writer behind it -- starts that much earlier; the launches behind it take rounds_)
        // Round 6: and, for files of 32 GiB and more, they GROW -- 6, then 22, then rounds_ = 44 chunks (1.4 GB compressed, ~2.5 GB
        // inflated: six blocks per decoder wave instead of three).  A launch ends with its slowest blocks while the other decoders idle: by the
        // profiles 0.33 GB launches run at 81 GB/s of inflated bytes, 1.2 GB launches at 86, one launch of 3.85 GB at 98
        // (profiles/r05/kernel_stats_bam2depth_final.csv, kernel_stats_bgzf_inflate.csv); the upload of 44 chunks takes as long as
        // their inflate, so the pipeline stays balanced.
        const int limit = rounds_env() ? rounds_ : launches_ == 0 && rounds_ > 6 ? 6 : launches_ == 1 && rounds_ > 22 ? rounds_ / 2 : rounds_;
        ++launches_;
        for (int taken = 0; taken < limit;) {         // several chunks under one inflate launch
            TextPump::Chunk c;
            const double t_r = wall_s();
            const bool got = !eof_ && pump_->next(c);
            t_read_ += wall_s() - t_r;
            if (!got) {
                if (!carry_.empty()) return -1;              // a partial block at the end: truncated file
                break;
            }
            if (c.eof) eof_ = true;
            size_t at = 0;
            if (skip_) {  // chunks before the first record's block
                const size_t k = skip_ < c.n ? (size_t)skip_ : c.n;
                skip_ -= k, at = k;
            }
            if (at == c.n) {
                pump_->recycle(c);
                if (eof_ && !carry_.empty()) return -1;
                continue;
            }
            BgzfParsed pb;
            int r = parse(c, at, pb);
            const double t_c = wall_s();
            if (r == 1 && !(st ? stage_add(*st, pb) : dev_.add(pb))) r = -1;
            t_copy_ += wall_s() - t_c;
            pump_->recycle(c);
            if (r != 1) return r;
            ++taken;
        }
        return 1;
    }
    bool stage_add(BgzfStage &st, const BgzfParsed &pb)
    {
        const size_t comp_bytes = pb.carry.size() + pb.body_len;
        if (!pb.blocks.empty() && !st.have_first) st.first_off = pb.first_off, st.have_first = true;
        if (st.at_comp + comp_bytes + 64 > st.cap) return false;
        if (!pb.carry.empty() && hpn_memcpy_h2d(up_ctx_, (uint8_t *)st.d_comp + st.at_comp, pb.carry.data(), pb.carry.size()) != HPN_OK) return false;
        if (pb.body_len && hpn_memcpy_h2d(up_ctx_, (uint8_t *)st.d_comp + st.at_comp + pb.carry.size(), pb.body, pb.body_len) != HPN_OK) return false;
        for (hpn_bgzf_block b : pb.blocks) {
            b.in_off += st.at_comp, b.out_off += st.at_out;
            st.pieces.push_back(b);
        }
        st.crcs.insert(st.crcs.end(), pb.crcs.begin(), pb.crcs.end());
        st.at_comp += comp_bytes, st.at_out += pb.out_bytes;
        return hpn_ctx_sync(up_ctx_) == HPN_OK;        // (pageable carry, and the pinned chunk goes back to the reader)
    }
    bool start_producer()
    {
        stop_ = false, turn_ = 0;
        for (BgzfStage &st : stage_) st.state = 0;
        producer_ = std::thread([this] {
            bind_thread_near(ctx_);
            if (!up_ctx_) {    // (on this thread: ~40 ms of queue creation beside the caller.  The tools leave through _exit (report.hpp:
                               // leave) -- the runtime's own exit handlers crash under a thread that is inside the runtime)
                int device = 0;
                if (hpn_ctx_device(ctx_, &device) != HPN_OK || hpn_ctx_create(device, &up_ctx_) != HPN_OK) {
                    up_ctx_ = nullptr;
                    BgzfStage &st = stage_[0];
                    st.result = -1;
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        st.state = 1;
                    }
                    cv_.notify_all();
                    return;
                }
                stamp("BGZF: context for the uploads made");
            }
            for (uint32_t k = 0;; ++k) {
                BgzfStage &st = stage_[k & 1];
                {
                    const double t_w = wall_s();
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return st.state == 0 || stop_; });
                    t_stage_ += wall_s() - t_w;
                    if (stop_) return;
                }
                const size_t room = (size_t)rounds_ * (chunk_ + 65536 + 64) + 64;
                st.pieces.clear(), st.crcs.clear(), st.at_comp = 0, st.at_out = 0, st.first_off = 0, st.have_first = false;
                int r = 1;
                if (room > st.cap) {
                    if (st.d_comp) hpn_dev_free(up_ctx_, st.d_comp);
                    st.d_comp = nullptr, st.cap = 0;
                    if (hpn_dev_malloc(up_ctx_, room + room / 8, &st.d_comp) == HPN_OK) st.cap = room + room / 8;
                    else r = -1;
                }
                if (r == 1) r = gather(&st, st);
                if (r == 1 && st.pieces.empty() && eof_) r = 0;
                st.result = r, st.eof = eof_ && carry_.empty();
                stamp("BGZF: launch prepared (compressed bytes on the device), blocks:", (double)st.pieces.size());
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    st.state = 1;
                }
                cv_.notify_all();
                if (r != 1) {                // the end (or a failure) has been handed over
                    if (getenv("HPN_TIMING"))
                        fprintf(stderr, "[hpn] BGZF read-ahead: waited %.3f s for the file reader, %.3f s in copies to the device, %.3f s for a free stage\n",
                                t_read_, t_copy_, t_stage_);
                    return;
                }
            }
        });
        return true;
    }
    void halt_producer()
    {
        if (!producer_.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        producer_.join();
        for (BgzfStage &st : stage_) st.state = 0;
        stop_ = false, ended_ = false;
    }
    hpn_ctx *up_ctx_ = nullptr;
    double t_read_ = 0, t_copy_ = 0, t_stage_ = 0;   // HPN_TIMING: where the read-ahead thread waits
    BgzfStage stage_[2];
    std::thread producer_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false, ahead_ = false, last_eof_ = false, ended_ = false;
    int final_ = 0;
    uint32_t turn_ = 0;

private:
    // 0 = need more bytes, 1 = complete (*need = header length), -1 = not BAM
    static int parse_header(const std::vector<uint8_t> &t, BamHeader &h, size_t *need)
    {
        if (t.size() < 8) return 0;
        if (memcmp(t.data(), "BAM\1", 4)) return -1;
        int32_t l_text, n_ref;
        memcpy(&l_text, t.data() + 4, 4);
        size_t p = 8 + (size_t)l_text;
        if (t.size() < p + 4) return 0;
        memcpy(&n_ref, t.data() + p, 4);
        p += 4;
        h.target_name.clear(), h.target_len.clear();
        for (int32_t i = 0; i < n_ref; ++i) {
            if (t.size() < p + 4) return 0;
            int32_t l_name, l_ref;
            memcpy(&l_name, t.data() + p, 4);
            if (t.size() < p + 4 + (size_t)l_name + 4) return 0;
            h.target_name.emplace_back((const char *)t.data() + p + 4);
            memcpy(&l_ref, t.data() + p + 4 + l_name, 4);
            h.target_len.push_back((uint32_t)l_ref);
            p += 8 + (size_t)l_name;
        }
        *need = p;
        return 1;
    }

    // Blocks of carry_ + chunk[at..): the table, on the host (stream state: carry_ moves on to the chunk's partial tail).
    // 1 = ok (`pb` describes the batch; it may hold no block), -1 = not BGZF / truncated.
    int parse(const TextPump::Chunk &c, size_t at, BgzfParsed &pb)
    {
        const uint8_t *p = c.p + at;
        const size_t n = c.n - at;
        size_t done = 0;
        pb = BgzfParsed();
        pb.first_off = first_off_;
        uint64_t in = 0;
        if (!carry_.empty()) {  // complete the block the previous chunk ended in
            while (carry_.size() < 18 && done < n) carry_.push_back(p[done++]);
            if (carry_.size() < 18) return eof_ ? -1 : 1;
            const uint32_t bsize = (carry_[16] | (carry_[17] << 8)) + 1u;
            const size_t more = bsize > carry_.size() ? bsize - carry_.size() : 0;
            if (more > n - done) {
                carry_.insert(carry_.end(), p + done, p + n);
                return eof_ ? -1 : 1;
            }
            carry_.insert(carry_.end(), p + done, p + done + more);
            done += more;
            if (!add_block(pb, carry_.data(), 0, bsize)) return -1;
            pb.carry.assign(carry_.begin(), carry_.begin() + bsize);
            in = bsize;
            carry_.clear();
        }
        pb.body = p + done;     // chunk bytes [body, body + body_len) go to the device behind the carried block
        size_t q = done;
        while (q + 18 <= n) {
            const uint32_t bsize = (p[q + 16] | (p[q + 17] << 8)) + 1u;
            if (q + bsize > n) break;
            if (!add_block(pb, p + q, in + (q - done), bsize)) return -1;
            q += bsize;
        }
        pb.body_len = q - done;
        carry_.assign(p + q, p + n);  // a partial block: kept for the next chunk
        if (!pb.blocks.empty()) first_off_ = 0;
        if (eof_ && !carry_.empty()) return -1;
        return 1;
    }

    static bool add_block(BgzfParsed &pb, const uint8_t *h, uint64_t in_off, uint32_t bsize)
    {
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') return false;
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (bsize < xlen + 20u) return false;
        hpn_bgzf_block b;
        b.in_off = in_off + 12u + xlen;
        b.in_len = bsize - xlen - 20u;
        memcpy(&b.out_len, h + bsize - 4, 4);
        if (b.out_len > 65536u) return false;
        uint32_t crc;
        memcpy(&crc, h + bsize - 8, 4);
        pb.crcs.push_back(crc);
        b.out_off = pb.out_bytes;
        pb.out_bytes += b.out_len;
        pb.blocks.push_back(b);
        return true;
    }

    hpn_ctx *ctx_ = nullptr;
    std::unique_ptr<TextPump> pump_;
    size_t chunk_ = 0;
    uint64_t start_ = 0, skip_ = 0;  // file offset of the block holding the first record
    uint32_t first_off_ = 0;         // ... and the record's offset inside it
    bool eof_ = false, text_mode_ = false;
    // Pinned chunks of 32 MiB (round 4: 88 MiB -- three of them cost 50 ms to pin in front of the file's first byte and as much
    // again when the process ends), 22 of them = ~700 MiB of compressed bytes under one inflate launch when the file is read
    // front to back (six in the first launch; 6,144 decoder waves hold ~130 MiB at once and a launch ends with its slowest
    // block: profiles/r04/e2e_rounds.txt); one behind a seek() -- a worker that reads ONE target (bam_multi.hpp) would inflate
    // the chunks of its neighbours behind the target's last record (HPN_BAM_CHUNK, HPN_BAM_ROUNDS: both)
    // Streams whose every chunk is a launch of its own (one target behind a seek, batches dealt to several devices) ask for
    // kLaunchChunk: ~4,500 blocks, most of one round of the chip's 6,144 decoder waves.
    static size_t default_chunk(size_t asked = 0)
    {
        if (const char *e = test_env("HPN_BAM_CHUNK")) return (size_t)atoll(e) < 65536 + 64 ? 65536 + 64 : (size_t)atoll(e);
        return asked ? asked : (size_t)32 << 20;
    }
public:
    static constexpr size_t kLaunchChunk = (size_t)88 << 20;
private:
    static int rounds_env()
    {
        const char *e = test_env("HPN_BAM_ROUNDS");
        const int v = e ? atoi(e) : 0;
        return v < 0 ? 0 : v > 64 ? 64 : v;
    }
    int rounds_ = rounds_env() ? rounds_env() : 44;
    uint32_t launches_ = 0;
    std::vector<uint8_t> carry_;
    BgzfDevice dev_;

    friend class BgzfFanout;
    friend class BgzfTextFanout;
};

inline bool bam_gpu_enabled()
{
    const char *e = getenv("HPN_BAM_GPU");
    return !(e && e[0] == '0');
}

// All records of one target into the open depth accumulation, from whichever ingest is active.
class DepthFeeder {
public:
    // try_gpu: device inflate + record walk (feed() returns 1 when the file turns out not to be decodable there)
    bool open(hpn_ctx *ctx, const char *path, BamHeader &hdr, bool try_gpu)
    {
        ctx_ = ctx;
        if (try_gpu) {
            gpu_.reset(new BgzfGpuStream());
            if (gpu_->open(ctx, path, hdr)) {
                gpu_->start();
                return true;
            }
            gpu_.reset();
            hdr = BamHeader();
        }
        return host_.open(path, hdr);
    }
    bool on_gpu() const { return (bool)gpu_; }
    // The tool's last input is done: nothing is taken apart (pinned chunks, the upload context and its buffers: ~0.1 s of
    // unpinning and queue destruction in front of an _exit that hands all of it back anyway).
    // (HPN_FULL_EXIT: the process will run exit handlers -- a profiler's -- and a read-ahead thread still inside the runtime at
    // that moment crashes its teardown: the stream is stopped and taken apart the ordinary way)
    void abandon()
    {
        if (getenv("HPN_FULL_EXIT")) gpu_.reset();
        else (void)gpu_.release();
    }

    // HPN_OK, an hpn_status, or 1 = the GPU ingest gave up (re-run the file with try_gpu = false)
    int feed(int32_t j)
    {
        if (gpu_) {
            for (;;) {
                if (!have_) {
                    const int r = gpu_->next(&info_);
                    if (r == 0) return HPN_OK;
                    if (r < 0) return 1;
                    have_ = true;
                }
                if (info_.n_records && info_.tid_min <= j && j <= info_.tid_max) {
                    const int rc = hpn_depth_add_raw_dev(ctx_, gpu_->d_raw());
                    if (rc != HPN_OK) return rc;
                }
                if (info_.n_records && info_.tid_max > j) return HPN_OK;  // the batch also holds later targets
                have_ = false;
            }
        }
        for (;;) {  // the target's records are contiguous in a coordinate-sorted file
            int32_t t = host_.peek_tid();
            while (t != INT32_MIN && t >= 0 && t < j) {  // out of order: not reachable through the index either
                batch_.clear();
                host_.next(batch_, false);
                t = host_.peek_tid();
            }
            batch_.clear();
            while (t == j && batch_.n() < (4u << 20)) {
                host_.next(batch_, false);
                t = host_.peek_tid();
            }
            if (batch_.n()) {
                hpn_bam_batch v = batch_.view();
                const int rc = hpn_depth_add(ctx_, &v);
                if (rc != HPN_OK) return rc;
            }
            if (t != j) return HPN_OK;
        }
    }

private:
    hpn_ctx *ctx_ = nullptr;
    std::unique_ptr<BgzfGpuStream> gpu_;
    BamReader host_;
    BamBatch batch_;
    hpn_raw_info info_;
    bool have_ = false;
};

}  // namespace hpn
