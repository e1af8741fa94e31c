// gz_shard.hpp -- ONE .fastq.gz inflated, framed and tallied on several GPU contexts ("lanes", one per device).
//
// The reference reads gzip input through zlib's gzread behind gzgets (IO_stream.h:122-136, fastq_count.c:112-118): one core,
// one stream.  host/gz_gpu.hpp moved the inflate to ONE device; configs[1] / configs[4] of BASELINE.json are gzipped FASTQ and the
// device inflate is their whole end-to-end wall, so this file spreads it over the node ("records shard trivially
// by block across the GPUs").  What makes that possible:
//
//   * the compressed file is cut into a FIXED grid of slices (stretch_ bytes each, from the member's first block on); batch b =
//     slices [b S, (b + 1) S) goes to lane b mod L.  The block start found in a slice is a function of the slice alone, so every
//     lane finds its own batch's stretches AND the start that ends its last stretch (it lies in the next lane's first slices,
//     which that lane finds too: the same search, the same answer) -- no lane waits for another one's search or upload;
//   * the symbolic decode of a batch (hpn_gz_inflate_begin_dev: 4/5 of the device time) needs nothing of the text before it;
//   * three short chains run through the batches in order, each a few values handed over on the host:
//       W  the 32 KiB window a batch ends with -> hpn_gz_inflate_finish_dev of the next (histories, translation: ~1/5),
//       M  the running member state (text offset, CRC-32 so far, member start) -> gzread's per-member ISIZE / CRC-32 checks,
//       L  the number of text lines before a batch -> which lines start records (hpn_fastq_text_piece_count);
//   * a batch's text is framed where it lies, in slices, with the piece calls of include/hpngs.h: the byte in front of a batch and
//     the 4 KiB behind it (the next batch's first bytes) cross between devices through the host.
//
// Exactness is by construction as in gz_gpu.hpp: stretch 0 starts at the member's first block, a stretch is accepted only if it
// stops on the next one's first bit at a block boundary (the last stretch of a batch: on the first stretch of the next batch),
// every member's ISIZE and CRC-32 are checked, the last member must end the file.  Anything else abandons the route: nothing is
// added and the caller reads the file another way.
#pragma once
#include "gz_gpu.hpp"
#include "text_relay.hpp"
#include "text_shard.hpp"

namespace hpn {

class GzSharded {
public:
    GzSharded(LaneGroup &g, const char *path, int threads, uint32_t per_call) : g_(g), path_(path), threads_(threads < 1 ? 1 : threads), S_(per_call < 1 ? 1 : per_call) {}
    ~GzSharded()
    {
        if (data_) munmap((void *)data_, size_);
        if (fd_ >= 0) close(fd_);
    }
    const char *why() const { return why_; }
    uint64_t batches() const { return n_batches_; }
    uint64_t text_bytes() const { return total_text_; }
    double ratio() const { return ratio_; }
    uint64_t file_bytes() const { return size_; }

    // false: not a file this route takes (why() says); nothing has been touched
    bool open()
    {
        fd_ = ::open(path_, O_RDONLY);
        if (fd_ < 0) return give_up("cannot open");
        struct stat sb;
        if (fstat(fd_, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) return give_up("not a regular file");
        size_ = (uint64_t)sb.st_size;
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return give_up("mmap failed");
        data_ = (const uint8_t *)m;
        const uint8_t *body = gzip_header_end(data_, data_ + size_);
        if (!body) return give_up("no gzip header");
        body_byte_ = (uint64_t)(body - data_);
        {   // expansion of the member's first megabytes: sizes the symbol scratch and the text buffers
            std::vector<uint8_t> probe(((size_t)4 << 20) + FastInflate::kOvershoot);
            FastInflate fi;
            fi.begin(body, data_ + size_);
            uint8_t *o = probe.data();
            if (fi.run(o, probe.data() + ((size_t)4 << 20), probe.data()) == FastInflate::kError) return give_up("the stream does not decode");
            const double in = (double)(fi.in_pos() - body) + 1, out = (double)(o - probe.data()) + 1;
            ratio_ = out / in < 1.0 ? 1.0 : out / in;
        }
        {
            const char *e = test_env("HPN_GZ_FIND");
            search_on_device_ = e ? !strcmp(e, "device") : threads_ / g_.lanes() < 6;
        }
        // stretches as in GzGpuStream::open: every call fills the chip, 256 KiB .. 1.5 MiB of compressed bytes each
        const char *e = test_env("HPN_GZ_STRETCH");
        uint64_t max_stretch = (uint64_t)((double)((uint64_t)3 << 19) * (ratio_ > 2.0 ? 2.0 / ratio_ : 1.0));
        if (max_stretch < ((uint64_t)512 << 10)) max_stretch = (uint64_t)512 << 10;
        const uint64_t calls = (size_ + S_ * max_stretch - 1) / (S_ * max_stretch);
        uint64_t want_calls = calls < (uint64_t)g_.lanes() ? (uint64_t)g_.lanes() : calls;   // at least one batch per lane
        {   // as in GzGpuStream::open: the DEVICE looks for the block starts of a file that fills the lanes' chips twice or more, and
            // a device's share goes in as few batches as keep a batch's symbol scratch near 10 GB (lanes that share a device --
            // HPN_NGPU on a one-GPU box -- share its memory: their batches are that much smaller)
            const uint64_t fills = size_ / (S_ * (uint64_t)g_.lanes() * ((uint64_t)256 << 10));
            if (!test_env("HPN_GZ_FIND") && fills >= 2) search_on_device_ = true;
            const double per_device = (double)size_ / (g_.distinct() ? (double)g_.lanes() : 1.0);
            uint64_t want = (uint64_t)(per_device * ratio_ * 2.8 / 10e9) + 1;
            // (per lane: the lanes of a device run side by side, so 1 / want of the device's share is in flight at a time)
            if (want > 4) want = 4;
            if (search_on_device_ && fills >= 2 && want_calls < want * (uint64_t)g_.lanes()) want_calls = want * (uint64_t)g_.lanes();
        }
        stretch_ = e ? (size_t)atoll(e) : (size_t)((size_ / (want_calls * S_) + 4096) & ~(uint64_t)4095);
        if (!e && stretch_ < ((size_t)256 << 10)) stretch_ = (size_t)256 << 10;
        if (!e && stretch_ > max_stretch) stretch_ = (size_t)max_stretch;
        if (stretch_ < 4096) stretch_ = 4096;
        n_slices_ = (size_ - body_byte_ + stretch_ - 1) / stretch_;
        // a file that does not fill every lane's chip once is spread evenly all the same: several devices half full beat one full one
        if (!test_env("HPN_GZ_BATCH") && n_slices_ < S_ * (uint64_t)g_.lanes()) {
            S_ = (n_slices_ + (uint64_t)g_.lanes() - 1) / (uint64_t)g_.lanes();
            if (S_ < 64) S_ = 64;
        }
        n_batches_ = (n_slices_ + S_ - 1) / S_;
        sym_cap_ = cap_for(ratio_ * 1.4);
        return true;
    }

    // Inflates, frames and tallies the whole file into the lanes' accumulators.  HPN_OK with *unusable: the route gave up
    // (the accumulators may hold partial counts: the caller drops them).
    int run(uint32_t tally_flags, bool *unusable)
    {
        *unusable = false;
        const int L = g_.lanes();
        relay_.reset(new TextRelay(tally_flags, (uint32_t)L));
        window_.assign(32768, 0);
        lanes_.clear();
        for (int l = 0; l < L; ++l) lanes_.emplace_back(new Lane());
        std::vector<std::thread> th;
        for (int l = 0; l < L; ++l) {
            th.emplace_back([this, l] { bind_thread_near(g_.ctx(l)); produce(l); });
            th.emplace_back([this, l] { bind_thread_near(g_.ctx(l)); consume(l); });
        }
        for (auto &t : th) t.join();
        for (auto &ln : lanes_) release(*ln);
        if (rc_ != HPN_OK) return rc_;
        if (abort_) *unusable = true;
        return HPN_OK;
    }

private:
    struct Slot {   // one batch's compressed side: filled by the lane's producer (state 0 -> 1), released by its consumer (-> 0)
        void *d_comp = nullptr, *d_chunks = nullptr;
        size_t cap_comp = 0, cap_chunks = 0, h_cap = 0;
        hpn_gz_chunk *h_chunks = nullptr;
        uint64_t last_start = 0, end_bit = kGzNone;
        uint32_t n = 0;
        bool final_data = false;    // the batch holds the file's last stretch (it runs to the final block and the trailer)
        int state = 0;
    };
    struct Lane {
        hpn_ctx *ctx = nullptr, *up = nullptr;
        std::unique_ptr<TextPump> pump;
        Slot slot[2];
        void *d_text = nullptr, *d_win_in = nullptr, *d_win_out = nullptr, *h_win = nullptr, *h_edge = nullptr;
        size_t cap_text = 0;
    };
    static constexpr uint64_t kLookAhead = 16;
    static constexpr size_t kFront = TextRelay::kFront;   // room in front of a batch's text for the byte before it

    bool give_up(const char *why)
    {
        why_ = why;
        return false;
    }
    void stop(const char *why, int rc = HPN_OK)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!abort_) {
                abort_ = true, rc_ = rc;
                snprintf(why_store_, sizeof why_store_, "%s", why);
                why_ = why_store_;
            }
        }
        cv_.notify_all();
        if (relay_) relay_->abort(why, rc);
    }
    bool stopped()
    {
        std::lock_guard<std::mutex> lk(m_);
        return abort_;
    }
    uint32_t cap_for(double ratio) const
    {
        const double c = ((double)stretch_ + 131072.0) * ratio + 65536.0;
        return (uint32_t)(((uint64_t)c + 7) & ~(uint64_t)7);
    }
    static bool reserve(hpn_ctx *ctx, void *&p, size_t &cap, size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) hpn_dev_free(ctx, p);
        p = nullptr, cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        if (hpn_dev_malloc(ctx, want, &p) != HPN_OK) return false;
        cap = want;
        return true;
    }
    void release(Lane &ln)
    {
        ln.pump.reset();
        if (ln.up) {
            for (Slot &s : ln.slot) {
                if (s.d_comp) hpn_dev_free(ln.up, s.d_comp);
                if (s.d_chunks) hpn_dev_free(ln.up, s.d_chunks);
                if (s.h_chunks) hpn_host_free(ln.up, s.h_chunks);
            }
            hpn_ctx_destroy(ln.up);
            ln.up = nullptr;
        }
        if (ln.ctx) {
            if (ln.d_text) hpn_dev_free(ln.ctx, ln.d_text);
            if (ln.d_win_in) hpn_dev_free(ln.ctx, ln.d_win_in);
            if (ln.d_win_out) hpn_dev_free(ln.ctx, ln.d_win_out);
            if (ln.h_win) hpn_host_free(ln.ctx, ln.h_win);
            if (ln.h_edge) hpn_host_free(ln.ctx, ln.h_edge);
            ln.d_text = ln.d_win_in = ln.d_win_out = ln.h_win = ln.h_edge = nullptr;
        }
    }
    void slice_bits(uint64_t k, uint64_t &lo, uint64_t &hi) const
    {
        lo = (body_byte_ + k * stretch_) * 8, hi = (body_byte_ + (k + 1) * stretch_) * 8;
        const uint64_t cap = (size_ - 8) * 8;            // (the trailer is not deflate data)
        if (hi > cap) hi = cap;
        if (lo > hi) lo = hi;
    }
    uint64_t find_start(uint64_t lo, uint64_t hi) const
    {
        if (hi <= lo) return kGzNone;
        const uint64_t b = gz_find_block_start(data_, size_, lo, hi, gz_find_scratch(), kGzFindScratch);
        return b != kGzNone ? b : gz_find_member_start(data_, size_, lo, hi, gz_find_scratch(), kGzFindScratch);
    }

    // ---- producer of lane l: search + upload + stretch table of its batches, one ahead of the consumer ----
    void produce(int l)
    {
        Lane &ln = *lanes_[(size_t)l];
        ln.ctx = g_.ctx(l);
        int device = 0;
        if (hpn_ctx_device(ln.ctx, &device) != HPN_OK || hpn_ctx_create(device, &ln.up) != HPN_OK) return stop("no context for the uploads");
        ln.pump.reset(new TextPump(ln.up, path_, (size_t)16 << 20, 3, true, body_byte_));
        if (!ln.pump->ok()) return stop("reader not available");
        uint32_t turn = 0;
        for (uint64_t b = (uint64_t)l; b < n_batches_; b += (uint64_t)g_.lanes(), ++turn) {
            Slot &sl = ln.slot[turn & 1];
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return sl.state == 0 || abort_; });
                if (abort_) return;
            }
            if (!prepare(ln, sl, b)) return;     // (prepare has said why)
            {
                std::lock_guard<std::mutex> lk(m_);
                sl.state = 1;
            }
            cv_.notify_all();
        }
    }
    bool prepare(Lane &ln, Slot &sl, uint64_t b)
    {
        const uint64_t k0 = b * S_, k1 = (b + 1) * S_ < n_slices_ ? (b + 1) * S_ : n_slices_;
        // the slices behind the batch tell where its last stretch ends: the first block start in them (looked for one slice after the
        // other; a deflate block of well-compressed text can be longer than a small slice)
        const uint64_t look = k1 + kLookAhead < n_slices_ ? k1 + kLookAhead : n_slices_;
        const uint64_t base_byte = body_byte_ + k0 * stretch_;
        uint64_t up_end = body_byte_ + look * stretch_ + 8192;
        if (up_end > size_ || look == n_slices_) up_end = size_;
        if (!reserve(ln.up, sl.d_comp, sl.cap_comp, (size_t)(up_end - base_byte) + 8192 + 256)) return fail("device memory (compressed bytes)");
        std::vector<uint64_t> found((size_t)(look - k0), kGzNone);
        auto host_search = [&](int nthreads) {
            std::atomic<uint64_t> take{0};
            const uint64_t own = k1 - k0;                 // the batch's own slices side by side, then the slices behind it until one has a start
            auto work = [&] {
                for (;;) {
                    const uint64_t i = take.fetch_add(1);
                    if (i >= own) return;
                    if (found[(size_t)i] != kGzNone) continue;
                    if (k0 + i == 0) {
                        found[0] = body_byte_ * 8;        // the member's first block
                        continue;
                    }
                    uint64_t lo, hi;
                    slice_bits(k0 + i, lo, hi);
                    found[(size_t)i] = find_start(lo, hi);
                }
            };
            std::vector<std::thread> pool;
            for (int t = 0; t < nthreads; ++t) pool.emplace_back(work);
            for (auto &t : pool) t.join();
            for (uint64_t i = own; i < found.size(); ++i) {
                if (found[(size_t)i] == kGzNone) {
                    uint64_t lo, hi;
                    slice_bits(k0 + i, lo, hi);
                    found[(size_t)i] = find_start(lo, hi);
                }
                if (found[(size_t)i] != kGzNone) break;
            }
        };
        const int share = threads_ / g_.lanes() < 1 ? 1 : threads_ / g_.lanes();
        if (search_on_device_) {
            if (!upload(ln, sl, base_byte, up_end)) return fail("upload failed");
            std::vector<hpn_span> spans;
            std::vector<size_t> which;
            for (size_t i = 0; i < found.size(); ++i) {
                if (k0 + i == 0) {
                    found[0] = body_byte_ * 8;
                    continue;
                }
                uint64_t lo, hi;
                slice_bits(k0 + i, lo, hi);
                if (hi > lo) spans.push_back(hpn_span{lo - base_byte * 8, hi - lo}), which.push_back(i);
            }
            if (!spans.empty()) {
                std::vector<uint64_t> got(spans.size(), kGzNone);
                if (hpn_gz_find_starts_dev(ln.up, (const uint8_t *)sl.d_comp, up_end - base_byte, spans.data(), (uint32_t)spans.size(), got.data()) != HPN_OK)
                    return fail("block-start search on the device");
                for (size_t j = 0; j < which.size(); ++j)
                    if (got[j] != kGzNone) found[which[j]] = got[j] + base_byte * 8;
            }
            host_search(share);          // what the device did not find (a member's first block that is a final block)
        } else {
            std::thread up([&] { up_ok_local(ln, sl, base_byte, up_end); });
            host_search(share);
            up.join();
            if (!ln_up_ok_) return fail("upload failed");
        }
        // the batch's stretches, and where its last one must stop
        std::vector<uint64_t> starts;
        for (uint64_t k = k0; k < k1; ++k)
            if (found[(size_t)(k - k0)] != kGzNone) starts.push_back(found[(size_t)(k - k0)]);
        uint64_t end_bit = kGzNone;
        for (uint64_t k = k1; k < look && end_bit == kGzNone; ++k) end_bit = found[(size_t)(k - k0)];
        const bool to_the_end = end_bit == kGzNone;
        if (to_the_end && look < n_slices_) return fail("no block start where one is expected");
        const uint32_t n = (uint32_t)starts.size();
        if (k1 - k0 >= 8 && (uint64_t)n * 2 < k1 - k0 && !(to_the_end && n == 0)) return fail("hardly any block starts found: not gzip'ed text");
        const uint64_t stop_byte = to_the_end ? size_ : ((end_bit >> 3) + 4096 < size_ ? (end_bit >> 3) + 4096 : size_);
        if (stop_byte > up_end) return fail("a stretch without a block start");
        if (n > sl.h_cap) {
            if (sl.h_chunks) hpn_host_free(ln.up, sl.h_chunks);
            sl.h_cap = n + n / 2 + 64;
            void *p = nullptr;
            if (hpn_host_malloc(ln.up, sl.h_cap * sizeof(hpn_gz_chunk), &p) != HPN_OK) return fail("pinned memory");
            sl.h_chunks = (hpn_gz_chunk *)p;
        }
        for (uint32_t k = 0; k < n; ++k) {
            hpn_gz_chunk &c = sl.h_chunks[k];
            const uint64_t s = starts[k], e = k + 1 < n ? starts[k + 1] : end_bit;
            c.in_off = (s >> 3) - base_byte;
            c.start_bit = (uint32_t)(s & 7);
            c.end_bit = e == kGzNone ? kGzNone : e - (s & ~(uint64_t)7);
            const uint64_t room = stop_byte - (s >> 3);
            c.in_len = room > 0x7fffff00ull ? 0x7fffff00u : (uint32_t)room;
        }
        if (n) {
            if (!reserve(ln.up, sl.d_chunks, sl.cap_chunks, (size_t)n * sizeof(hpn_gz_chunk))) return fail("device memory");
            if (hpn_memcpy_h2d(ln.up, sl.d_chunks, sl.h_chunks, (size_t)n * sizeof(hpn_gz_chunk)) != HPN_OK || hpn_ctx_sync(ln.up) != HPN_OK) return fail("copy failed");
        }
        sl.n = n, sl.end_bit = end_bit, sl.last_start = n ? starts[n - 1] : 0, sl.final_data = to_the_end && n > 0;
        return true;
    }
    bool fail(const char *why)
    {
        stop(why);
        return false;
    }
    std::atomic<bool> ln_up_ok_{true};     // (written by upload threads: any failure stops the route)
    void up_ok_local(Lane &ln, Slot &sl, uint64_t from, uint64_t to)
    {
        if (!upload(ln, sl, from, to)) ln_up_ok_ = false;
    }
    // file bytes [from, to) -> sl.d_comp[0 ..): the lane's pump restarted at `from` (the batches of a lane are not neighbours)
    bool upload(Lane &ln, Slot &sl, uint64_t from, uint64_t to)
    {
        if (!ln.pump->restart(from)) return false;
        uint64_t at = from;
        TextPump::Chunk c;
        while (at < to) {
            if (!ln.pump->next(c)) return false;
            const uint64_t take = c.n < to - at ? c.n : to - at;
            const bool ok = take == 0 || (hpn_memcpy_h2d(ln.up, (uint8_t *)sl.d_comp + (at - from), c.p, take) == HPN_OK && hpn_ctx_sync(ln.up) == HPN_OK);
            const bool eof = c.eof;
            ln.pump->recycle(c);
            if (!ok) return false;
            at += take;
            if (eof && at < to) return false;      // the file is shorter than its size said
        }
        return true;
    }

    // ---- consumer of lane l: decode, the three chains, framing ----
    void consume(int l)
    {
        Lane &ln = *lanes_[(size_t)l];
        {   // the producer makes the lane's contexts
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return ln.slot[0].state != 0 || abort_ || n_batches_ <= (uint64_t)l; });
            if (abort_) return;
        }
        hpn_ctx *ctx = g_.ctx(l);
        void *p = nullptr;
        if (hpn_dev_malloc(ctx, 32768, &ln.d_win_in) != HPN_OK || hpn_dev_malloc(ctx, 32768, &ln.d_win_out) != HPN_OK || hpn_host_malloc(ctx, 32768, &ln.h_win) != HPN_OK ||
            hpn_host_malloc(ctx, TextRelay::kEdgeBytes, &p) != HPN_OK)
            return stop("memory for the hand-overs");
        ln.h_edge = p;
        uint32_t turn = 0;
        for (uint64_t b = (uint64_t)l; b < n_batches_; b += (uint64_t)g_.lanes(), ++turn) {
            Slot &sl = ln.slot[turn & 1];
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return sl.state == 1 || abort_; });
                if (abort_) return;
            }
            if (!one_batch(ln, sl, b)) return;
        }
    }
    bool one_batch(Lane &ln, Slot &sl, uint64_t b)
    {
        hpn_ctx *ctx = ln.ctx;
        const uint32_t n = sl.n;
        uint32_t cap = sym_cap_now();
        if (hpn_gz_inflate_begin_dev(ctx, (const uint8_t *)sl.d_comp, (const hpn_gz_chunk *)sl.d_chunks, n, cap) != HPN_OK) return fail_ctx(ctx);
        // ---- W: the window in front of this batch ----
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return w_next_ == b || abort_; });
            if (abort_) return false;
            memcpy(ln.h_win, window_.data(), 32768);
        }
        if (b && (hpn_memcpy_h2d(ctx, ln.d_win_in, ln.h_win, 32768) != HPN_OK || hpn_ctx_sync(ctx) != HPN_OK)) return fail_ctx(ctx);
        const uint64_t comp_bytes = (uint64_t)(sl.n ? stretch_ * (uint64_t)(S_) : 0);
        uint64_t want = (uint64_t)((double)comp_bytes * ratio_ * 1.25) + ((uint64_t)8 << 20);
        hpn_gz_info info;
        for (bool resized = false, grew = false;;) {
            if (!reserve(ctx, ln.d_text, ln.cap_text, (size_t)want + kFront + HPN_TEXT_PIECE_TAIL + 128)) return fail("device memory (text)");
            const int rc = hpn_gz_inflate_finish_dev(ctx, b ? (const uint8_t *)ln.d_win_in : nullptr, (uint8_t *)ln.d_text + kFront,
                                                     ln.cap_text - kFront - HPN_TEXT_PIECE_TAIL - 128, (uint8_t *)ln.d_win_out, &info);
            if (rc == HPN_E_CAPACITY && !resized) {      // more text than guessed: the symbols are still there
                want = info.n_bytes, resized = true;
                continue;
            }
            if (rc != HPN_OK) return fail_ctx(ctx);
            if ((info.status == 12 || info.status == 14 || info.status == 1) && !grew) {
                // out of symbol scratch: this batch once more with twice the room (later batches start with it)
                grew = true;
                cap = grow_sym_cap();
                if (hpn_gz_inflate_begin_dev(ctx, (const uint8_t *)sl.d_comp, (const hpn_gz_chunk *)sl.d_chunks, n, cap) != HPN_OK) return fail_ctx(ctx);
                continue;
            }
            break;
        }
        if (info.status) {
            snprintf(why_buf_, sizeof why_buf_, "batch %llu, stretch %u of %u: decoder status %u", (unsigned long long)b, info.bad_chunk, n, info.status);
            return fail(why_buf_);
        }
        if (!sl.final_data && info.final_chunk) return fail("the member ends inside the file");
        if (sl.final_data && info.final_chunk != n) return fail("the member ends before the file does");
        if (hpn_memcpy_d2h(ctx, ln.h_win, ln.d_win_out, 32768) != HPN_OK || hpn_ctx_sync(ctx) != HPN_OK) return fail_ctx(ctx);
        const uint64_t nb = info.n_bytes;
        const bool ends_stream = sl.final_data || (n == 0 && after_final_.load());
        if (sl.final_data) after_final_ = true;
        {
            std::lock_guard<std::mutex> lk(m_);
            memcpy(window_.data(), ln.h_win, 32768);
            w_next_ = b + 1;
        }
        cv_.notify_all();
        // what the neighbours need of this text: its first bytes (the tail of the batch before), its last byte (the head of the next)
        if (!relay_->publish(ctx, b, (const uint8_t *)ln.d_text + kFront, nb, ends_stream, ln.h_edge)) return fail(relay_->why());
        // ---- M: gzread's checks of every member that ended in this batch ----
        {
            uint32_t nm = 0;
            std::vector<hpn_gz_member> members;
            int rc = hpn_gz_members(ctx, nullptr, 0, &nm);
            if (rc == HPN_E_CAPACITY) {
                members.resize(nm);
                rc = hpn_gz_members(ctx, members.data(), nm, &nm);
            }
            if (rc != HPN_OK) return fail("member list");
            std::vector<hpn_span> spans;
            uint64_t at = 0;
            for (uint32_t k = 0; k < nm; ++k) spans.push_back(hpn_span{kFront + at, members[k].text_end - at}), at = members[k].text_end;
            spans.push_back(hpn_span{kFront + at, nb - at});
            std::vector<uint32_t> crcs(spans.size(), 0);
            if (check_crc_ && hpn_crc32_dev(ctx, (const uint8_t *)ln.d_text, spans.data(), (uint32_t)spans.size(), crcs.data()) != HPN_OK) return fail("CRC-32 kernel");
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return m_next_ == b || abort_; });
            if (abort_) return false;
            const char *bad = nullptr;
            for (uint32_t k = 0; k < nm && !bad; ++k) {
                const uint64_t end = total_text_ + members[k].text_end;
                if (members[k].isize != (uint32_t)(end - member_start_)) bad = "ISIZE mismatch in a member";
                member_crc_ = hpn_crc32_join(member_crc_, crcs[k], spans[k].len);
                if (!bad && check_crc_ && member_crc_ != members[k].crc32) bad = "CRC-32 mismatch in a member";
                member_crc_ = 0, member_start_ = end, ++n_members_;
            }
            member_crc_ = hpn_crc32_join(member_crc_, crcs.back(), spans.back().len);
            total_text_ += nb;
            if (!bad && sl.final_data) {     // the last member must end the file: trailer right behind the final block, nothing after it
                const uint64_t trailer = ((sl.last_start & ~(uint64_t)7) + info.end_bit) >> 3;
                uint32_t isize = 0, crc = 0;
                if (trailer + 8 != size_) bad = "bytes behind the member";
                else {
                    memcpy(&crc, data_ + trailer, 4), memcpy(&isize, data_ + trailer + 4, 4);
                    if (isize != (uint32_t)(total_text_ - member_start_)) bad = "ISIZE mismatch";
                    else if (check_crc_ && crc != member_crc_) bad = "CRC-32 mismatch";
                }
                ++n_members_;
            }
            m_next_ = b + 1;
            lk.unlock();
            cv_.notify_all();
            if (bad) return fail(bad);
        }
        {   // the compressed bytes are done with: the producer may fill this set again
            std::lock_guard<std::mutex> lk(m_);
            sl.state = 0;
        }
        cv_.notify_all();
        // ---- L: the batch's text in slices through the piece calls; the lines in front of it come down the relay ----
        if (!relay_->frame(ctx, b, (uint8_t *)ln.d_text + kFront, nb, ends_stream, ln.h_edge)) return fail(relay_->why());
        return true;
    }
    // A library call failed in a lane (device memory for the symbol scratch -- ~2.8 x the expansion per compressed byte, and
    // lanes may share a device --, a second HPN_E_CAPACITY, a HIP error): the ROUTE is abandoned, not the file -- the caller
    // drops the lanes' partial counts and takes the one-context route, which gives up and falls back for the same conditions
    // and reports a device that really is broken itself.
    bool fail_ctx(hpn_ctx *ctx)
    {
        snprintf(why_buf_, sizeof why_buf_, "%s", hpn_ctx_last_error(ctx));
        stop(why_buf_);
        return false;
    }
    uint32_t sym_cap_now()
    {
        std::lock_guard<std::mutex> lk(m_);
        return sym_cap_;
    }
    uint32_t grow_sym_cap()
    {
        std::lock_guard<std::mutex> lk(m_);
        sym_cap_ = cap_for(ratio_ * 3.0);
        return sym_cap_;
    }

    LaneGroup &g_;
    const char *path_;
    int threads_;
    uint64_t S_;                        // slices (= stretches, at most) per batch
    int fd_ = -1;
    const uint8_t *data_ = nullptr;
    uint64_t size_ = 0, body_byte_ = 0, n_slices_ = 0, n_batches_ = 0;
    size_t stretch_ = 0;
    double ratio_ = 4.0;
    uint32_t sym_cap_ = 0;
    bool search_on_device_ = false;
    const bool check_crc_ = !(test_env("HPN_GZ_CRC") && test_env("HPN_GZ_CRC")[0] == '0');
    std::vector<std::unique_ptr<Lane>> lanes_;
    // the chains (all under m_)
    std::mutex m_;
    std::condition_variable cv_;
    bool abort_ = false;
    int rc_ = HPN_OK;
    const char *why_ = "";
    char why_buf_[160];
    uint64_t w_next_ = 0, m_next_ = 0;
    std::vector<uint8_t> window_;
    std::unique_ptr<TextRelay> relay_;
    char why_store_[200];
    uint64_t total_text_ = 0, member_start_ = 0, n_members_ = 0;
    uint32_t member_crc_ = 0;
    std::atomic<bool> after_final_{false};
};

// fastq_count / fastq_count_kthread: one gzip input over the group's lanes.  *unusable: nothing was added, the caller takes
// the one-context routes.
inline int tally_gz_sharded(LaneGroup &g, const char *path, hpn_tally *acc, bool *unusable)
{
    *unusable = false;
    const double t0 = wall_s();
    if (!g.ensure()) {
        *unusable = true;
        return HPN_OK;
    }
    const long cpus = usable_cpus() / text_workers_in_flight();
    uint32_t slots = 5120;
    (void)hpn_inflate_slots(g.ctx(0), &slots);                        // stretches a chip decodes at once
    uint32_t per_call = (uint32_t)(slots / (uint32_t)text_workers_in_flight());
    if (!g.distinct()) per_call /= (uint32_t)g.lanes();              // lanes on ONE device (HPN_NGPU on a one-GPU box) share its decoder slots and its memory
    if (const char *e = test_env("HPN_GZ_BATCH")) per_call = (uint32_t)atol(e);
    GzSharded gs(g, path, (int)(cpus < 1 ? 1 : cpus > 32 ? 32 : cpus), per_call < 1 ? 1 : per_call);
    if (!gs.open()) {
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] %s: gzip over %d lanes not taken: %s\n", path, g.lanes(), gs.why());
        *unusable = true;
        return HPN_OK;
    }
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    bool gave_up = false;
    const int rc = gs.run(flags, &gave_up);
    if (rc != HPN_OK || gave_up) {
        g.drop_all();
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] %s: gzip over %d lanes abandoned after %.3f s: %s\n", path, g.lanes(), wall_s() - t0, gs.why());
        if (rc != HPN_OK && !gave_up) return rc;
        *unusable = true;
        return HPN_OK;
    }
    const double t1 = wall_s();
    const char *how = "host";
    const int rs = g.sum_into(acc, &how);
    if (getenv("HPN_TIMING"))
        fprintf(stderr, "[hpn] %s: one gzip input over %d lanes: %llu batches, %.1f MB -> %.1f MB of text, inflate + frame + tally %.3f s, sum by %s %.3f s%s\n", path,
                g.lanes(), (unsigned long long)gs.batches(), gs.file_bytes() / 1e6, gs.text_bytes() / 1e6, t1 - t0, how, wall_s() - t1,
                g.distinct() ? "" : " (lanes share a device)");
    return rs;
}

}  // namespace hpn
