// text_relay.hpp -- the text of ONE FASTQ stream arriving in batches on different GPU contexts, framed where it lies.
//
// host/gz_shard.hpp (a gzip member's stretches) and bam_multi.hpp's block fan-out (BGZF) inflate the batches of one compressed
// input on several devices; each batch's text stays on the device that made it.  The four gzgets of count_read
// (fastq_count.c:112-118) do not care where a batch ends, so the records that straddle batches need what the piece calls of
// include/hpngs.h ask for: the byte in front of a batch, the 4 KiB behind it, and the number of lines the stream has before it.
// This class is that hand-over, through the host:
//   publish(b)   a batch's first bytes and last byte, as soon as its text exists (its neighbours wait for them);
//   frame(b)     the batch in slices through hpn_fastq_text_piece_lines / _count; the line counts run down the batches in order
//                (a batch's first slice is framed once every earlier batch has indexed all of its text).
// Batches are numbered in stream order and a lane takes its batches in increasing order; a batch may be short or empty (its
// neighbours then look further).  Irregular text or any failure stops the relay (aborted(), why()): the caller drops what the
// lanes have counted and reads the input another way.
#pragma once
#include <condition_variable>
#include <mutex>
#include <vector>

#include "text_stream.hpp"

namespace hpn {

class TextRelay {
public:
    static constexpr size_t kFront = 64;                          // writable bytes a batch's buffer has in front of its text
    static constexpr size_t kEdgeBytes = HPN_TEXT_PIECE_TAIL + 64;   // pinned scratch a lane brings (h_edge)
    // lanes: batch b is framed by lane b % lanes, which takes its batches one after the other (0: unknown, nothing is checked)
    explicit TextRelay(uint32_t tally_flags = 0, uint32_t lanes = 0) : flags_(tally_flags), lanes_(lanes) {}

    bool aborted()
    {
        std::lock_guard<std::mutex> lk(m_);
        return abort_;
    }
    const char *why() const { return why_; }
    int status() const { return rc_; }
    void abort(const char *why, int rc = HPN_OK)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!abort_) {
                abort_ = true, rc_ = rc;
                snprintf(why_, sizeof why_, "%s", why);
            }
        }
        cv_.notify_all();
    }

    // d_text[0, nb): batch b's text on ctx's device.  ends_stream: no text follows it.
    bool publish(hpn_ctx *ctx, uint64_t b, const uint8_t *d_text, uint64_t nb, bool ends_stream, void *h_edge)
    {
        uint8_t *edge = (uint8_t *)h_edge;
        const uint32_t head_n = (uint32_t)(nb < HPN_TEXT_PIECE_TAIL ? nb : HPN_TEXT_PIECE_TAIL);
        if (nb) {
            if (hpn_memcpy_d2h(ctx, edge, d_text, head_n) != HPN_OK || hpn_memcpy_d2h(ctx, edge + HPN_TEXT_PIECE_TAIL, d_text + nb - 1, 1) != HPN_OK ||
                hpn_ctx_sync(ctx) != HPN_OK)
                return fail_ctx(ctx);
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            Pub &pb = at(b);
            pb.n_bytes = nb, pb.head_n = head_n, pb.ends_stream = ends_stream;
            pb.head.assign(edge, edge + head_n);
            pb.last_byte = nb ? edge[HPN_TEXT_PIECE_TAIL] : 0;
            pb.ready = true;
        }
        cv_.notify_all();
        return true;
    }

    // Frames and tallies batch b (published before).  d_text: kFront writable bytes in front, HPN_TEXT_PIECE_TAIL behind nb.
    bool frame(hpn_ctx *ctx, uint64_t b, uint8_t *text, uint64_t nb, bool ends_stream, void *h_edge)
    {
        uint64_t lines_here = 0, before = 0;
        bool have_board = false;
        auto board = [&]() -> bool {       // lines in front of this batch (all earlier batches have indexed all their text)
            if (have_board) return true;
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return l_next_ >= b || abort_; });
            if (abort_) return false;
            before = board_at(b), have_board = true;
            return true;
        };
        auto done = [&](uint64_t total) {
            {
                std::lock_guard<std::mutex> lk(m_);
                board_at(b + 1) = total;
                l_next_ = b + 1;
            }
            cv_.notify_all();
        };
        if (!nb) {
            if (!board()) return false;
            done(before);
            return true;
        }
        uint32_t head = 0;
        if (b) {    // the byte in front of this batch: the last byte of the nearest batch with text before it
            uint8_t prev = 0;
            bool any = false;
            {
                std::unique_lock<std::mutex> lk(m_);
                for (uint64_t i = b; i-- > 0 && !any;) {
                    cv_.wait(lk, [&] { return at(i).ready || abort_; });
                    if (abort_) return false;
                    if (at(i).n_bytes) prev = at(i).last_byte, any = true;
                }
            }
            if (any) {
                *(uint8_t *)h_edge = prev;
                if (hpn_memcpy_h2d(ctx, text - 1, h_edge, 1) != HPN_OK || hpn_ctx_sync(ctx) != HPN_OK) return fail_ctx(ctx);
                head = 1;
            }
        }
        const uint64_t slice = slice_bytes();
        for (uint64_t pos = 0; pos < nb;) {
            const uint64_t own = nb - pos < slice ? nb - pos : slice;
            const bool last_slice = pos + own == nb;
            uint64_t tail = last_slice ? 0 : (nb - pos - own < HPN_TEXT_PIECE_TAIL ? nb - pos - own : HPN_TEXT_PIECE_TAIL);
            bool last_piece = last_slice && ends_stream;
            if (last_slice && !ends_stream) {     // the tail lies in the text of the batches that follow (one, unless it is short or empty)
                std::unique_lock<std::mutex> lk(m_);
                bool ended = false;
                for (uint64_t i = b + 1; tail < HPN_TEXT_PIECE_TAIL && !ended; ++i) {
                    if (lanes_ && i >= b + lanes_ && !at(i).ready) {
                        // the tail would have to come from a batch of THIS lane, which is published behind this very call (the lanes
                        // between hold less than 4 KiB of text together: tiny BGZF blocks or gzip members, test-sized batches): give
                        // the route up instead of waiting for ever
                        lk.unlock();
                        return fail("batches too short to frame across lanes");
                    }
                    cv_.wait(lk, [&] { return at(i).ready || abort_; });
                    if (abort_) return false;
                    const Pub &nx = at(i);
                    const uint64_t k = nx.head_n < HPN_TEXT_PIECE_TAIL - tail ? nx.head_n : HPN_TEXT_PIECE_TAIL - tail;
                    if (k) memcpy((uint8_t *)h_edge + tail, nx.head.data(), k);
                    tail += k;
                    ended = nx.ends_stream && k == nx.n_bytes;       // (a batch's head is all of its text when it is that short)
                }
                lk.unlock();
                last_piece = tail == 0;                             // nothing follows: this slice ends the stream after all
                if (tail && (hpn_memcpy_h2d(ctx, text + nb, h_edge, tail) != HPN_OK || hpn_ctx_sync(ctx) != HPN_OK)) return fail_ctx(ctx);
            }
            const uint32_t h = pos ? 1u : head;
            hpn_text_piece pl;
            int rc = hpn_fastq_text_piece_lines(ctx, text + pos - h, h + own + tail, h, own, last_piece ? 1 : 0, &pl);
            if (rc != HPN_OK) return fail_ctx(ctx);
            if (pl.irregular) return fail("irregular text");
            if (!board()) return false;
            if (last_slice) done(before + lines_here + pl.n_lines);     // this batch's lines are all counted: the next batch may frame
            hpn_text_info info;
            rc = hpn_fastq_text_piece_count(ctx, before + lines_here, flags_, &info);
            if (rc != HPN_OK) return fail_ctx(ctx);
            if (info.irregular) return fail("irregular text");
            lines_here += pl.n_lines;
            pos += own;
            std::lock_guard<std::mutex> lk(m_);
            n_records_ += info.n_records, n_pieces_ += 1;
        }
        return true;
    }
    uint64_t records() const { return n_records_; }
    uint64_t pieces() const { return n_pieces_; }

private:
    struct Pub {    // what a batch tells its neighbours once its text exists
        bool ready = false;
        uint64_t n_bytes = 0;
        uint32_t head_n = 0;
        uint8_t last_byte = 0;
        bool ends_stream = false;      // no text follows this batch's
        std::vector<uint8_t> head;     // its first min(4096, n_bytes) bytes
    };
    Pub &at(uint64_t b)                // (under m_)
    {
        if (pub_.size() <= b) pub_.resize((size_t)b + 64);
        return pub_[(size_t)b];
    }
    uint64_t &board_at(uint64_t b)     // (under m_)
    {
        if (board_.size() <= b) board_.resize((size_t)b + 64, 0);
        return board_[(size_t)b];
    }
    static uint64_t slice_bytes()   // HPN_TEXT_SLICE: tests cut small texts into several pieces
    {
        const char *e = test_env("HPN_TEXT_SLICE");
        return e && atoll(e) >= 2 * (long long)HPN_TEXT_PIECE_TAIL ? (uint64_t)atoll(e) : (uint64_t)256 << 20;
    }
    bool fail(const char *why)
    {
        abort(why);
        return false;
    }
    bool fail_ctx(hpn_ctx *ctx)
    {
        abort(hpn_ctx_last_error(ctx), HPN_E_HIP);
        return false;
    }
    uint32_t flags_, lanes_ = 0;
    std::mutex m_;
    std::condition_variable cv_;
    bool abort_ = false;
    int rc_ = HPN_OK;
    char why_[200] = "";
    uint64_t l_next_ = 0, n_records_ = 0, n_pieces_ = 0;
    std::vector<Pub> pub_;
    std::vector<uint64_t> board_;
};

}  // namespace hpn
