// gz_gpu.hpp -- a single-member .fastq.gz inflated on the GPU (hpn_gz_inflate_dev), host side.
//
// host/pgz_reader.hpp inflates ONE deflate stream in parallel on the host cores (two passes:
// symbolic decode with unknown history, histories resolved in order); the decode is 3/4 of
// that thread time.  Here the host keeps only what is cheap per megabyte: finding where
// deflate blocks start (gz_find_block_start: ~1 ms per stretch on one core) and moving the
// compressed bytes into pinned chunks and on to the device.  The device inflates every stretch
// with one wavefront, resolves the histories and writes one contiguous text per batch, which the
// caller frames where it lies (hpn_fastq_text_count / _trim take device text).
//
// Two halves per batch, on two threads: PREPARE (block-start search on the pool, the compressed bytes through pinned chunks to
// the device, the stretch table) needs nothing of the batch before it but where that one's last stretch ends, which the search
// itself tells; RUN (inflate, histories, translation, member checks) needs the device.  A producer thread prepares batch k + 1
// into the second set of buffers -- through a context of its own, so that its copies run beside the kernels -- while the caller
// runs batch k and frames its text: the search and the upload (half of the route's wall on a 2.4 GB file) leave the path.
//
// Exactness is by construction as in the host reader: stretch 0 starts at the member's first
// block, and the device accepts a stretch only if it ends on the next one's first bit at a block
// boundary.  The last member must end the file (trailer + nothing else); every member's ISIZE and CRC-32 (hpn_crc32_dev over
// the inflated text) must match, as gzread demands.  Anything else -- trailing bytes, a block start the trial decoder got wrong,
// more than 8x expansion, damage -- makes next() return -1 and the caller reads the file
// through the host readers from its first byte.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "pgz_reader.hpp"
#include "text_stream.hpp"

namespace hpn {

inline bool gz_gpu_enabled()
{
    const char *e = getenv("HPN_GZ_GPU");
    return !(e && e[0] == '0');
}

constexpr size_t kTextPad = 8192;     // room in front of a batch's text for the framer's carried bytes
constexpr uint64_t kGzOversub = 1;    // stretches per decoder slot and batch (HPN_GZ_OVERSUB in test-hooks builds)

class GzGpuStream {
public:
    ~GzGpuStream()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (producer_.joinable()) producer_.join();
        if (ctx2_maker_.joinable()) ctx2_maker_.join();
        pump_.reset();
        if (ctx_) {
            for (Slot &s : slot_) {
                hpn_dev_free(ctx_, s.d_comp), hpn_dev_free(ctx_, s.d_chunks);
                if (s.h_chunks) hpn_host_free(ctx_, s.h_chunks);
            }
            hpn_dev_free(ctx_, d_text_), hpn_dev_free(ctx_, d_win_[0]), hpn_dev_free(ctx_, d_win_[1]);
        }
        if (up_ctx_) hpn_ctx_destroy(up_ctx_);
        if (ctx2_) hpn_ctx_destroy(ctx2_);
        if (data_) munmap((void *)data_, size_);
        if (fd_ >= 0) close(fd_);
    }

    // threads: block-start searchers.  max_stretches: per device call.  stretch_bytes 0: HPN_GZ_STRETCH, else the file
    // size / max_stretches within 256 KiB .. 1.5 MiB (the search costs ~0.4 ms per stretch).
    bool open(hpn_ctx *ctx, const char *path, int threads, uint32_t max_stretches, size_t stretch_bytes = 0)
    {
        ctx_ = ctx;
        // (the upload context is made beside the probe below: ~25 ms each)
        std::thread up_maker([this] {
            int device = 0;
            if (hpn_ctx_device(ctx_, &device) != HPN_OK || hpn_ctx_create(device, &up_ctx_) != HPN_OK) up_ctx_ = nullptr;
        });
        struct Joiner {
            std::thread &t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } up_join{up_maker};
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) return give_up("cannot open");
        struct stat sb;
        if (fstat(fd_, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) return give_up("not a regular file");
        size_ = (uint64_t)sb.st_size;
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return give_up("mmap failed");
        data_ = (const uint8_t *)m;
        const uint8_t *body = gzip_header_end(data_, data_ + size_);
        if (!body) return give_up("no gzip header");
        threads_ = threads < 1 ? 1 : threads;
        {
            const char *e = test_env("HPN_GZ_FIND");
            search_on_device_ = e ? !strcmp(e, "device") : threads_ < 6;
        }
        max_stretches_ = max_stretches < 1 ? 1 : max_stretches > 65535u ? 65535u : max_stretches;
        // A file that fills the chip twice or more with stretches of 256 KiB is taken in several batches, and then the DEVICE looks
        // for the block starts whatever the core count (four times the stretches are four times the cores' search: 0.65 s for
        // 24 K stretches on 15 cores).  One batch of 6,144 stretches of 1.2 MB holds 38 GB of symbol scratch + 12 GB of text at once:
        // measured on a 7.2 GB file 0.34 s of device time on a fresh box and 1.8 - 2.4 s on a box whose memory had been in use
        // (profiles/r04/e2e_tools_b.txt); four batches of a quarter of that run 0.10 s each on both.
        if (!test_env("HPN_GZ_FIND") && size_ / ((uint64_t)max_stretches_ * ((uint64_t)256 << 10)) >= 2) search_on_device_ = true;
        // symbols of scratch per stretch: from the expansion of the member's first megabytes (FASTQ is homogeneous; a
        // stretch that needs more is decoded again with twice the room)
        {
            std::vector<uint8_t> probe(((size_t)4 << 20) + FastInflate::kOvershoot);
            FastInflate fi;
            fi.begin(body, data_ + size_);
            uint8_t *o = probe.data();
            const int r = fi.run(o, probe.data() + ((size_t)4 << 20), probe.data());
            if (r == FastInflate::kError) return give_up("the stream does not decode");
            const double in = (double)(fi.in_pos() - body) + 1, out = (double)(o - probe.data()) + 1;
            ratio_ = out / in < 1.0 ? 1.0 : out / in;
        }
        if (!stretch_bytes) {
            // as many stretches as the device calls may hold, so that every call fills the chip: a wavefront inflates ~4.5 MB
            // of text per second whatever the stretch size, so only the number of waves in flight matters
            const char *e = test_env("HPN_GZ_STRETCH");
            // at most 1.5 MiB (~0.5 s of one wavefront; the search is per stretch, so few and long), less for text that
            // expands a lot: the symbol scratch is ~3 bytes per byte of text in flight
            uint64_t kMaxStretch = (uint64_t)((double)((uint64_t)3 << 19) * (ratio_ > 2.0 ? 2.0 / ratio_ : 1.0));
            if (kMaxStretch < ((uint64_t)512 << 10)) kMaxStretch = (uint64_t)512 << 10;
            uint64_t calls = (size_ + max_stretches_ * kMaxStretch - 1) / (max_stretches_ * kMaxStretch);
            // ... and where the DEVICE looks for the block starts (cheap per stretch), a file that can fill the chip more than once
            // with stretches of 256 KiB is taken in up to four batches: the next batch's upload runs beside this one's device work
            const uint64_t fills = size_ / ((uint64_t)max_stretches_ * ((uint64_t)256 << 10));
            // ... as FEW batches as keep a batch's symbol scratch near 10 GB (2.8 bytes x the expansion per compressed byte in
            // flight): every batch pays the histories' serial walk over its stretches once (23 ms per 6,000 stretches, whatever
            // their size: profiles/r04/kernel_stats_gz_tool.csv) and ends with its longest stretch -- but large allocations are
            // what the runtime sometimes stalls in for seconds (see HPN_GZ_OVERLAP above), so the scratch stays modest
            uint64_t want = (uint64_t)((double)size_ * ratio_ * 2.8 / 10e9) + 1;
            if (want < 2) want = 2;
            if (want > 4) want = 4;
            if (want > fills) want = fills;
            if (search_on_device_ && calls < want) calls = want;
            // (rounded up to 4 KiB, not more: a stretch 12 % longer than the share leaves 12 % of the wave slots empty AND makes
            // the one launch 12 % longer -- 2.4 GB file, 4,608 slots: 4,097 stretches of 576 KiB where 4,590 of 516 fit)
            stretch_bytes = e ? (size_t)atoll(e) : (size_t)((size_ / (calls * max_stretches_) + 4096) & ~(uint64_t)4095);
            if (!e && stretch_bytes < ((size_t)256 << 10)) stretch_bytes = (size_t)256 << 10;
            if (!e && stretch_bytes > kMaxStretch) stretch_bytes = (size_t)kMaxStretch;  // (symbol scratch: ~12 bytes per compressed byte in flight)
        }
        if (stretch_bytes < 4096) stretch_bytes = 4096;
        // More stretches than decoder slots (round 6): the device TAKES stretches one by one (k_gz_sym_inflate's counter), so a
        // batch of `over` shorter stretches per slot -- the same bytes -- does not end with the slot whose one stretch was the
        // slow one.  What it costs: the search and the histories are per stretch.
        if (!test_env("HPN_GZ_STRETCH")) {
            const char *o = test_env("HPN_GZ_OVERSUB");
            uint64_t over = o ? (uint64_t)atoi(o) : kGzOversub;
            if (over < 1) over = 1;
            uint64_t cut = (stretch_bytes / over + 4095) & ~(uint64_t)4095;
            if (cut < ((uint64_t)128 << 10)) cut = (uint64_t)128 << 10;
            if (cut < stretch_bytes) {
                const uint64_t per_slot = stretch_bytes / cut;
                uint64_t ms = (uint64_t)max_stretches_ * per_slot;
                max_stretches_ = ms > 65535u ? 65535u : (uint32_t)ms;
                stretch_bytes = (size_t)cut;
            }
        }
        stretch_ = stretch_bytes;
        sym_cap_ = cap_for(ratio_ * 1.4);
        {   // what a device call holds, at its largest (symbol scratch grown once, compressed bytes, text): within half of the
            // memory that is free now -- other processes may share the device
            uint64_t free_b = 0, total_b = 0;
            if (hpn_dev_mem_info(ctx_, &free_b, &total_b) == HPN_OK && free_b) {
                const double per_stretch = 2.0 * cap_for(ratio_ * 3.0) + (double)stretch_ * (2.0 + ratio_ * 1.25 * 1.125);   // (two sets of compressed bytes)
                const uint64_t fit = (uint64_t)((double)(free_b / 2) / per_stretch);
                if (fit < max_stretches_) max_stretches_ = fit < 64 ? 64u : (uint32_t)fit;
            }
        }
        first_bit_ = (uint64_t)(body - data_) * 8;
        next_start_ = first_bit_;
        // the compressed bytes reach the device through pinned chunks read in parallel (the page cache is not pinned)
        int device = 0;
        (void)hpn_ctx_device(ctx_, &device);
        up_maker.join();
        if (!up_ctx_) return give_up("no context for the uploads");
        stamp("gzip: probed, second context");
        pump_.reset(new TextPump(up_ctx_, path, (size_t)32 << 20, 3, true));
        if (!pump_->ok()) return give_up("reader not available");
        if (hpn_dev_malloc(ctx_, 32768, &d_win_[0]) != HPN_OK || hpn_dev_malloc(ctx_, 32768, &d_win_[1]) != HPN_OK) return give_up("device memory");
        producer_ = std::thread([this] { produce(); });
        // HPN_GZ_OVERLAP=1: a second context, so that batch k + 1 decodes while batch k's histories are resolved (next()).  Off by
        // default: measured -0.02 s of 0.57 on the 7.2 GB file (the one-workgroup history walk runs three times slower beside
        // 6,144 decoder waves), for a second set of symbol scratch -- and allocations of this size are what sometimes takes the
        // runtime seconds (profiles/r04/gz_stamps.txt: one run in six stalls 3.6 s in a 14 GB hipMalloc)
        if (test_env("HPN_GZ_OVERLAP") && test_env("HPN_GZ_OVERLAP")[0] == '1')
            ctx2_maker_ = std::thread([this, device] {            // (~30 ms, beside the first batch's upload)
                if (hpn_ctx_create(device, &ctx2_) != HPN_OK) ctx2_ = nullptr;
            });
        return true;
    }
    const uint8_t *d_text() const { return (const uint8_t *)d_text_ + kTextPad; }   // (kTextPad writable bytes in front: hpn_fastq_text_count_inplace)
    const char *why() const { return why_; }  // what made open() / next() give up
    double ratio() const { return ratio_; }     // text bytes per compressed byte over the member's first megabytes
    uint64_t members() const { return n_members_; }   // gzip members met so far
    uint64_t file_bytes() const { return size_; }
    double seconds_find() const { return t_find_; }
    double seconds_upload() const { return t_upload_; }
    double seconds_device() const { return t_device_; }
    bool at_end() const { return done_; }

    // Next batch of text on the device: 1 = ok (*n_bytes of it at d_text()), 0 = end of the member (checked), -1 = not
    // decodable here.
    int next(uint64_t *n_bytes)
    {
        *n_bytes = 0;
        if (done_) return 0;
        Slot &sl = slot_[batch_ & 1];
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return sl.state != 0; });
        }
        if (sl.state < 0) return give_up(sl.why) - 1;
        stamp("batch prepared (block starts, compressed bytes on the device), stretches:", (double)sl.n);
        const uint32_t n = sl.n;
        const uint64_t comp_bytes = sl.comp_bytes, end_bit = sl.end_bit;
        const bool last_batch = sl.last;
        void *const d_comp_ = sl.d_comp, *const d_chunks_ = sl.d_chunks;
        const double t2 = wall_s();
        // ---- inflate, resolve, translate ----
        // The batches alternate between two contexts (two streams): the symbolic decode of batch k + 1 is started BEFORE batch k's
        // histories are resolved (one workgroup, a serial walk over the stretches: 23 ms per batch), its text translated, checked
        // and tallied -- none of which needs the chip -- and its first waves take the decoder slots batch k's last stretches
        // leave (round 4: all of that was in a row, 130 ms per batch of which 100 decode).
        hpn_ctx *const cx = inflate_ctx(batch_);
        auto begin = [&](hpn_ctx *c, Slot &b) {
            const int rc = hpn_gz_inflate_begin_dev(c, (const uint8_t *)b.d_comp, (const hpn_gz_chunk *)b.d_chunks, b.n, sym_cap_);
            b.begun = rc == HPN_OK, b.begun_cap = sym_cap_;
            return rc;
        };
        if (!sl.begun) {
            if (begin(cx, sl) != HPN_OK) return give_up(hpn_ctx_last_error(cx)) - 1;
            stamp("symbolic decode started (a context's first: its scratch allocated)");
        }
        if (!last_batch && inflate_ctx(batch_ + 1) != cx) {
            Slot &nx = slot_[(batch_ + 1) & 1];
            {   // (prepared while this batch decodes: the wait is in that shadow)
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return nx.state != 0; });
            }
            if (nx.state > 0 && !nx.begun) (void)begin(inflate_ctx(batch_ + 1), nx);     // (a failure shows when the batch's turn comes)
            stamp("the next batch's decode started");
        }
        uint64_t want = (uint64_t)((double)comp_bytes * ratio_ * 1.25) + ((uint64_t)8 << 20);
        hpn_gz_info info;
        for (int attempt = 0;; ++attempt) {
            if (!reserve(ctx_, d_text_, cap_text_, want + 64 + kTextPad)) return give_up("device memory (text)") - 1;
            const int rc = hpn_gz_inflate_finish_dev(cx, batch_ ? (const uint8_t *)d_win_[batch_ & 1] : nullptr, (uint8_t *)d_text_ + kTextPad, cap_text_ - 64 - kTextPad,
                                                     (uint8_t *)d_win_[(batch_ + 1) & 1], &info);
            if (rc == HPN_E_CAPACITY && attempt == 0) {  // more text than guessed: once more with room for it (the symbols stay)
                want = info.n_bytes;
                continue;
            }
            if (rc != HPN_OK) return give_up(hpn_ctx_last_error(cx)) - 1;
            break;
        }
        t_device_ += wall_s() - t2;
        stamp("inflate returned");
        if ((info.status == 12 || info.status == 14 || info.status == 1) && (!grown_ || sl.begun_cap < sym_cap_)) {  // out of room: once more with twice as much
            if (!grown_) grown_ = true, sym_cap_ = cap_for(ratio_ * 3.0);
            const int rc = hpn_gz_inflate_dev(cx, (const uint8_t *)d_comp_, (const hpn_gz_chunk *)d_chunks_, n, sym_cap_,
                                              batch_ ? (const uint8_t *)d_win_[batch_ & 1] : nullptr, (uint8_t *)d_text_ + kTextPad, cap_text_ - 64 - kTextPad,
                                              (uint8_t *)d_win_[(batch_ + 1) & 1], &info);
            if (rc != HPN_OK) return give_up(hpn_ctx_last_error(cx)) - 1;
        }
        if (info.status) {
            snprintf(why_buf_, sizeof why_buf_, "stretch %u of %u: decoder status %u", info.bad_chunk, n, info.status);
            return give_up(why_buf_) - 1;
        }
        const uint64_t last_start = sl.starts[n - 1];
        {   // the compressed bytes are done with: the producer may fill this set again
            std::lock_guard<std::mutex> g(mu_);
            sl.state = 0;
        }
        cv_.notify_all();
        ++batch_;
        {   // members that ended inside this batch (cat a.gz b.gz): every one's ISIZE against the bytes it produced, and its
            // CRC-32 against the text (gzread checks both, and the reference counts nothing of a member's last buffers if they
            // fail: such a file is read again through zlib itself, tally_file).  The text of a member may span batches: the CRC
            // of each piece is taken while its text is on the device and folded into the member's (hpn_crc32_join).
            uint32_t nm = 0;
            int rc = hpn_gz_members(cx, nullptr, 0, &nm);
            if (rc == HPN_E_CAPACITY) {
                members_.resize(nm);
                rc = hpn_gz_members(cx, members_.data(), nm, &nm);
            }
            if (rc != HPN_OK) return give_up("member list") - 1;
            spans_.clear();
            uint64_t at = 0;
            for (uint32_t k = 0; k < nm; ++k) spans_.push_back(hpn_span{at, members_[k].text_end - at}), at = members_[k].text_end;
            spans_.push_back(hpn_span{at, info.n_bytes - at});          // the piece of the member that goes on in the next batch
            crcs_.resize(spans_.size());
            if (check_crc_ && hpn_crc32_dev(ctx_, (const uint8_t *)d_text_ + kTextPad, spans_.data(), (uint32_t)spans_.size(), crcs_.data()) != HPN_OK)
                return give_up("CRC-32 kernel") - 1;
            for (uint32_t k = 0; k < nm; ++k) {
                const uint64_t end = total_ + members_[k].text_end;
                if (members_[k].isize != (uint32_t)(end - member_start_)) return give_up("ISIZE mismatch in a member") - 1;
                member_crc_ = hpn_crc32_join(member_crc_, crcs_[k], spans_[k].len);
                if (check_crc_ && member_crc_ != members_[k].crc32) return give_up("CRC-32 mismatch in a member") - 1;
                member_crc_ = 0;
                member_start_ = end;
                ++n_members_;
            }
            member_crc_ = hpn_crc32_join(member_crc_, crcs_.back(), spans_.back().len);
        }
        total_ += info.n_bytes;
        *n_bytes = info.n_bytes;
        if (last_batch) {
            // the last member must end the file: final block in the last stretch, 8-byte trailer, nothing behind, ISIZE right
            if (info.final_chunk != n) return give_up("the member ends before the file does") - 1;
            const uint64_t trailer = ((last_start & ~(uint64_t)7) + info.end_bit) >> 3;
            if (trailer + 8 != size_) return give_up("bytes behind the member") - 1;
            uint32_t isize, crc;
            memcpy(&crc, data_ + trailer, 4);
            memcpy(&isize, data_ + trailer + 4, 4);
            if (isize != (uint32_t)(total_ - member_start_)) return give_up("ISIZE mismatch") - 1;
            if (check_crc_ && crc != member_crc_) return give_up("CRC-32 mismatch") - 1;
            ++n_members_;
            done_ = true;
        } else {
            if (info.final_chunk) return give_up("the member ends inside the file") - 1;  // more members or garbage follow
            (void)end_bit;
        }
        return 1;
    }

private:
    // one batch's compressed side: filled by the producer (state 0 -> 1, or -1 with `why`), released by next() (-> 0)
    struct Slot {
        void *d_comp = nullptr, *d_chunks = nullptr;
        size_t cap_comp = 0, cap_chunks = 0, h_cap = 0;
        hpn_gz_chunk *h_chunks = nullptr;
        std::vector<uint64_t> starts;
        uint32_t n = 0;
        uint64_t comp_bytes = 0, end_bit = kGzNone;
        bool last = false;
        bool begun = false;          // its symbolic decode has been started (next(): beside the batch before)
        uint32_t begun_cap = 0;      // ... with this much symbol scratch per stretch
        int state = 0;
        const char *why = "";
    };
    bool fail(Slot &sl, const char *why)
    {
        sl.why = why;
        return false;
    }
    void produce()
    {
        bind_thread_near(ctx_);
        for (uint32_t k = 0;; ++k) {
            Slot &sl = slot_[k & 1];
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return sl.state == 0 || stop_; });
                if (stop_) return;
            }
            const bool ok = prepare(sl);
            {
                std::lock_guard<std::mutex> g(mu_);
                sl.state = ok ? 1 : -1;
            }
            cv_.notify_all();
            if (!ok || sl.last) return;
        }
    }
    // search + upload + stretch table of the batch that starts at next_start_ (producer thread, its own context)
    bool prepare(Slot &sl)
    {
        // ---- the batch's stretches: starts found in [lo, hi) slices of the compressed file, by the pool, while this
        // thread moves the batch's compressed bytes to the device ----
        const uint64_t base_byte = next_start_ >> 3;
        const uint64_t left = (size_ - base_byte + stretch_ - 1) / stretch_;
        const uint64_t calls = (left + max_stretches_ - 1) / max_stretches_;
        uint64_t slices = (left + calls - 1) / calls;  // even shares: no call is left with a sliver
        bool last_batch = calls <= 1;
        if (last_batch) slices = left;
        const double t0 = wall_s();
        std::vector<uint64_t> found((size_t)slices + 1, kGzNone);
        found[0] = next_start_;
        const uint64_t n_search = last_batch ? slices - 1 : slices;              // slices 1 .. n_search are searched
        auto slice_bits = [&](uint64_t k, uint64_t &lo, uint64_t &hi) {
            lo = (base_byte + k * stretch_) * 8, hi = (base_byte + (k + 1) * stretch_) * 8;
            if (hi > size_ * 8) hi = size_ * 8;
        };
        bool up_ok = reserve(up_ctx_, sl.d_comp, sl.cap_comp, (size_t)((slices + 3) * stretch_ + 8192 + 256));
        uint64_t body_end;
        if (search_on_device_) {
            // the compressed bytes first -- the batch and two slices behind it: the slice that tells where the batch ends, and
            // what a trial decode at its end reads --, then the search where they lie (k_gz_find_starts); what the device does not
            // find in a slice (a member's first block is a final block: files of many small members) the cores look for
            body_end = base_byte + (slices + 2) * stretch_;
            if (last_batch || body_end > size_) body_end = size_;
            if (up_ok) up_ok = upload(sl, base_byte, body_end, base_byte);
            t_upload_ += wall_s() - t0;
            if (!up_ok) return fail(sl, "upload failed");
            std::vector<hpn_span> spans;
            for (uint64_t k = 1; k <= n_search; ++k) {
                uint64_t lo, hi;
                slice_bits(k, lo, hi);
                const uint64_t cap = (size_ - 8) * 8;                              // (the trailer is not deflate data)
                if (hi > cap) hi = cap;
                spans.push_back(hpn_span{lo - base_byte * 8, hi > lo ? hi - lo : 0});
            }
            if (!spans.empty()) {
                if (hpn_gz_find_starts_dev(up_ctx_, (const uint8_t *)sl.d_comp, body_end - base_byte, spans.data(), (uint32_t)spans.size(), found.data() + 1) != HPN_OK)
                    return fail(sl, "block-start search on the device");
                for (uint64_t k = 1; k <= n_search; ++k)
                    if (found[(size_t)k] != kGzNone) found[(size_t)k] += base_byte * 8;
            }
            std::atomic<uint64_t> take{1};
            auto rest = [&] {
                for (;;) {
                    const uint64_t k = take.fetch_add(1);
                    if (k > n_search) return;
                    if (found[(size_t)k] != kGzNone) continue;
                    uint64_t lo, hi;
                    slice_bits(k, lo, hi);
                    found[(size_t)k] = find_start(lo, hi);
                }
            };
            uint64_t missing = 0;
            for (uint64_t k = 1; k <= n_search; ++k) missing += found[(size_t)k] == kGzNone;
            if (missing) {
                std::vector<std::thread> pool;
                for (int t = 0; t < threads_ && (uint64_t)t < missing; ++t) pool.emplace_back(rest);
                for (auto &t : pool) t.join();
            }
        } else {
            std::atomic<uint64_t> take{1};
            auto work = [&] {
                for (;;) {
                    const uint64_t k = take.fetch_add(1);
                    if (k > n_search) return;
                    uint64_t lo, hi;
                    slice_bits(k, lo, hi);
                    found[(size_t)k] = find_start(lo, hi);
                }
            };
            std::vector<std::thread> pool;
            for (int t = 0; t < threads_; ++t) pool.emplace_back(work);
            body_end = last_batch || base_byte + slices * stretch_ > size_ ? size_ : base_byte + slices * stretch_;
            if (up_ok) up_ok = upload(sl, base_byte, body_end, base_byte);
            t_upload_ += wall_s() - t0;
            for (auto &t : pool) t.join();
            if (!up_ok) return fail(sl, "upload failed");
        }
        // the slice after the batch tells where its last stretch ends; nothing found there: one more slice ...
        uint64_t end_bit = kGzNone;
        if (!last_batch) {
            uint64_t k = slices;
            end_bit = found[(size_t)slices];
            while (end_bit == kGzNone && k < slices + 1 && (base_byte + (k + 1) * stretch_) < size_) {  // (a slice without a dynamic block)
                ++k;
                const uint64_t lo = (base_byte + k * stretch_) * 8, hi = (base_byte + (k + 1) * stretch_) * 8;
                end_bit = find_start(lo, hi < size_ * 8 ? hi : size_ * 8);
            }
            if (end_bit == kGzNone) {
                if ((base_byte + (k + 1) * stretch_) < size_) return fail(sl, "no block start where one is expected");
                last_batch = true;  // ... or, at the end of the file, run to the final block
            }
        }
        t_find_ += wall_s() - t0;
        sl.starts.clear();
        for (uint64_t k = 0; k < slices; ++k)
            if (found[(size_t)k] != kGzNone) sl.starts.push_back(found[(size_t)k]);
        const uint32_t n = (uint32_t)sl.starts.size();
        // text compressed the usual way has a block start in every slice; a file where most slices show none is not
        // what this route is for (one wavefront would crawl through megabytes)
        if (slices >= 8 && (uint64_t)n * 2 < slices) return fail(sl, "hardly any block starts found: not gzip'ed text");
        // ---- the rest of the compressed bytes, up to a little beyond the batch's end ----
        const uint64_t stop_byte = last_batch ? size_ : ((end_bit >> 3) + 4096 < size_ ? (end_bit >> 3) + 4096 : size_);
        const uint64_t comp_bytes = stop_byte - base_byte;
        if (comp_bytes + 256 > sl.cap_comp) return fail(sl, "a stretch without a block start");
        if (stop_byte > body_end && !upload(sl, body_end, stop_byte, base_byte)) return fail(sl, "upload failed");
        // ---- stretch table ----
        if (n > sl.h_cap) {
            if (sl.h_chunks) hpn_host_free(up_ctx_, sl.h_chunks);
            sl.h_cap = n + n / 2 + 64;
            void *p = nullptr;
            if (hpn_host_malloc(up_ctx_, sl.h_cap * sizeof(hpn_gz_chunk), &p) != HPN_OK) return fail(sl, "pinned memory");
            sl.h_chunks = (hpn_gz_chunk *)p;
        }
        for (uint32_t k = 0; k < n; ++k) {
            hpn_gz_chunk &c = sl.h_chunks[k];
            const uint64_t s = sl.starts[k], e = k + 1 < n ? sl.starts[k + 1] : end_bit;
            c.in_off = (s >> 3) - base_byte;
            c.start_bit = (uint32_t)(s & 7);
            c.end_bit = e == kGzNone ? kGzNone : e - (s & ~(uint64_t)7);
            const uint64_t room = stop_byte - (s >> 3);
            c.in_len = room > 0x7fffff00ull ? 0x7fffff00u : (uint32_t)room;  // (the kernel adds small constants to it in 32 bits)
        }
        if (!reserve(up_ctx_, sl.d_chunks, sl.cap_chunks, (size_t)n * sizeof(hpn_gz_chunk))) return fail(sl, "device memory");
        if (hpn_memcpy_h2d(up_ctx_, sl.d_chunks, sl.h_chunks, (size_t)n * sizeof(hpn_gz_chunk)) != HPN_OK || hpn_ctx_sync(up_ctx_) != HPN_OK) return fail(sl, "copy failed");
        sl.n = n, sl.comp_bytes = comp_bytes, sl.end_bit = end_bit, sl.last = last_batch, sl.begun = false;
        if (!last_batch) next_start_ = end_bit;
        return true;
    }
    // a block that is not its member's last, or -- files of many one-block members -- a member's first block
    uint64_t find_start(uint64_t lo, uint64_t hi) const
    {
        const uint64_t b = gz_find_block_start(data_, size_, lo, hi, gz_find_scratch(), kGzFindScratch);
        return b != kGzNone ? b : gz_find_member_start(data_, size_, lo, hi, gz_find_scratch(), kGzFindScratch);
    }
    uint32_t cap_for(double ratio) const  // symbols per stretch: its compressed bytes + a block or two, expanded
    {
        const double c = ((double)stretch_ + 131072.0) * ratio + 65536.0;
        return (uint32_t)(((uint64_t)c + 7) & ~(uint64_t)7);
    }
    bool give_up(const char *why)  // false (callers returning int subtract 1)
    {
        why_ = why;
        return false;
    }
    bool reserve(hpn_ctx *ctx, void *&p, size_t &cap, size_t bytes)    // (ctx: the calling thread's)
    {
        if (bytes <= cap) return true;
        if (p) hpn_dev_free(ctx, p);
        p = nullptr, cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        if (hpn_dev_malloc(ctx, want, &p) != HPN_OK) return false;
        cap = want;
        return true;
    }
    // file bytes [from, to) -> d_comp_[0 ..).  The pump delivers the file once, in order, in pinned chunks; a chunk that
    // reaches beyond `to` stays current for the next batch.  Batches overlap by a few KiB at their seams (a stretch may be
    // read a little past its end): bytes the pump has already let go of are taken from the mapped file.
    bool upload(Slot &sl, uint64_t from, uint64_t to, uint64_t origin)
    {
        void *const d_comp_ = sl.d_comp;
        hpn_ctx *const ctx_ = up_ctx_;      // (the producer's context: its copies run beside the caller's kernels)
        uint64_t at = from;
        if (at < pump_off_) {
            const uint64_t e = to < pump_off_ ? to : pump_off_;
            if (hpn_memcpy_h2d(ctx_, (uint8_t *)d_comp_ + (at - origin), data_ + at, e - at) != HPN_OK || hpn_ctx_sync(ctx_) != HPN_OK) return false;
            at = e;
        }
        while (at < to) {
            if (!have_chunk_) {
                if (!pump_->next(cur_)) return false;
                have_chunk_ = true;
            }
            const uint64_t c0 = pump_off_, c1 = pump_off_ + cur_.n;
            const uint64_t b = at >= c1 ? at : (to < c1 ? to : c1);
            if (b > at && hpn_memcpy_h2d(ctx_, (uint8_t *)d_comp_ + (at - origin), cur_.p + (at - c0), b - at) != HPN_OK) return false;
            at = b;
            if (at >= c1) {  // used up: back to the reader once the copy out of it is done
                if (hpn_ctx_sync(ctx_) != HPN_OK) return false;
                const bool eof = cur_.eof;
                pump_->recycle(cur_);
                have_chunk_ = false;
                pump_off_ = c1;
                if (eof && at < to) return false;  // the file is shorter than its size said
            }
        }
        return hpn_ctx_sync(ctx_) == HPN_OK;
    }

    hpn_ctx *ctx_ = nullptr, *up_ctx_ = nullptr, *ctx2_ = nullptr;
    std::thread ctx2_maker_;
    // the context batch b is decoded through: the caller's and a second one in turn (no second one: all through the caller's)
    hpn_ctx *inflate_ctx(uint32_t b)
    {
        if (ctx2_maker_.joinable()) ctx2_maker_.join();
        return (b & 1u) && ctx2_ ? ctx2_ : ctx_;
    }
    // who looks for the block starts: the cores (0.4 ms per stretch and core, beside the upload and the batch before: with 16
    // cores that keeps up with the device), or the device (HPN_GZ_FIND=device; ~15 ms per batch, but behind the upload and not
    // beside an inflate kernel, which fills the CUs' LDS) -- the default where the process has few cores
    bool search_on_device_ = false;
    Slot slot_[2];
    std::thread producer_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    int fd_ = -1;
    const uint8_t *data_ = nullptr;
    uint64_t size_ = 0;
    size_t stretch_ = 0;
    int threads_ = 1;
    uint32_t max_stretches_ = 0, sym_cap_ = 0, batch_ = 0;
    uint64_t first_bit_ = 0, next_start_ = 0, total_ = 0;
    uint64_t member_start_ = 0, n_members_ = 0;   // text offset where the current member began; members finished
    std::vector<hpn_gz_member> members_;
    std::vector<hpn_span> spans_;
    std::vector<uint32_t> crcs_;
    uint32_t member_crc_ = 0;                       // CRC-32 of the current member's text so far
    const bool check_crc_ = !(test_env("HPN_GZ_CRC") && test_env("HPN_GZ_CRC")[0] == '0');   // (timing experiments only)
    bool done_ = false, grown_ = false;
    double ratio_ = 4.0;
    const char *why_ = "";
    char why_buf_[96];
    double t_find_ = 0, t_upload_ = 0, t_device_ = 0;
    std::unique_ptr<TextPump> pump_;
    TextPump::Chunk cur_;
    bool have_chunk_ = false;
    uint64_t pump_off_ = 0;  // file offset of cur_'s first byte
    void *d_text_ = nullptr, *d_win_[2] = {nullptr, nullptr};
    size_t cap_text_ = 0;
};

}  // namespace hpn
