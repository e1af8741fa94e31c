// bgzf_shard.hpp -- ONE bgzip-compressed FASTQ over several GPU contexts ("lanes", one per device).
//
// BGZF blocks are independent deflate streams, so any run of whole blocks can be inflated anywhere: ONE reader walks the
// compressed file (pinned chunks, block headers: BgzfGpuStream's host side) and hands the chunks to the lanes in turn; every
// lane copies its chunk over its own PCIe link, inflates the blocks (hpn_bgzf_inflate_dev) and frames the text where it lands.
// The records that straddle two chunks are what host/text_relay.hpp is for: the byte in front of a batch, the 4 KiB behind it
// and the number of lines before it travel between the lanes through the host.  The lanes' count vectors are summed where
// reduceStats sums files (fastq_count_kthread.c:180-210): LaneGroup::sum_into.  The reference reads such a file through zlib's
// gzread like any gzip input (IO_stream.h:122-136, fastq_count.c:112-118).
// Anything the route cannot take (a damaged block, irregular text, a lane that cannot be made) abandons it: nothing is added.
#pragma once
#include "bam_gpu.hpp"
#include "text_relay.hpp"
#include "text_shard.hpp"

namespace hpn {

class BgzfTextFanout {
public:
    // counts into the lanes' accumulators; false: abandoned (why() says), the accumulators may hold partial counts
    static bool run(LaneGroup &g, const char *path, uint32_t tally_flags, char *why, size_t why_cap, uint64_t *n_batches, uint64_t *text_bytes)
    {
        const int L = g.lanes();
        struct Box {
            std::mutex m;
            std::condition_variable cv;
            bool has = false, quit = false, ends = false;
            uint64_t seq = 0;
            TextPump::Chunk c;
            BgzfParsed pb;
        };
        std::vector<std::unique_ptr<Box>> box;
        for (int l = 0; l < L; ++l) box.emplace_back(new Box());
        BgzfGpuStream gs;                       // the reader: read-ahead, block tables
        if (!gs.open_text(g.ctx(0), path, L + 2, BgzfGpuStream::kLaunchChunk)) {
            snprintf(why, why_cap, "reader not available");
            return false;
        }
        TextRelay relay(tally_flags, (uint32_t)L);
        std::atomic<bool> failed{false};
        std::atomic<uint64_t> bytes{0};
        std::vector<std::thread> th;
        for (int l = 0; l < L; ++l)
            th.emplace_back([&, l] {
                hpn_ctx *ctx = g.ctx(l);
                bind_thread_near(ctx);
                BgzfDevice dev(ctx);
                dev.set_out_pad(TextRelay::kFront, HPN_TEXT_PIECE_TAIL + 64);
                void *h_edge = nullptr;
                bool ok = hpn_host_malloc(ctx, TextRelay::kEdgeBytes, &h_edge) == HPN_OK;
                if (!ok) relay.abort("pinned memory for the hand-overs");
                Box &b = *box[(size_t)l];
                for (;;) {
                    std::unique_lock<std::mutex> lk(b.m);
                    b.cv.wait(lk, [&] { return b.has || b.quit; });
                    if (!b.has) break;
                    lk.unlock();
                    if (ok && !relay.aborted()) {
                        hpn_raw_info info;
                        ok = dev.run(b.pb, true, &info) == 1;
                        if (!ok) relay.abort("a BGZF block that does not inflate");
                    }
                    gs.pump_->recycle(b.c);       // (run() has waited for its copies out of the pinned chunk)
                    const uint64_t nb = b.pb.out_bytes;
                    if (ok && !relay.aborted()) {
                        bytes += nb;
                        uint8_t *text = const_cast<uint8_t *>(dev.d_raw());
                        ok = relay.publish(ctx, b.seq, text, nb, b.ends, h_edge) && relay.frame(ctx, b.seq, text, nb, b.ends, h_edge);
                    }
                    lk.lock();
                    b.has = false;
                    lk.unlock();
                    b.cv.notify_all();
                }
                if (h_edge) hpn_host_free(ctx, h_edge);
            });
        uint64_t seq = 0;
        bool sent_end = false;
        for (int turn = 0; !relay.aborted();) {
            TextPump::Chunk c;
            if (gs.eof_ || !gs.pump_->next(c)) break;
            if (c.eof) gs.eof_ = true;
            BgzfParsed pb;
            if (gs.parse(c, 0, pb) != 1 || (gs.eof_ && !gs.carry_.empty())) {     // not BGZF, or the file ends inside a block
                gs.pump_->recycle(c);
                relay.abort("not BGZF all the way, or truncated");
                break;
            }
            if (pb.blocks.empty() && !gs.eof_) {       // (a chunk smaller than a block: its bytes are carried)
                gs.pump_->recycle(c);
                continue;
            }
            Box &b = *box[(size_t)turn];
            {
                std::unique_lock<std::mutex> lk(b.m);
                b.cv.wait(lk, [&] { return !b.has; });
                b.c = c, b.pb = std::move(pb), b.has = true, b.seq = seq++, b.ends = gs.eof_;
            }
            b.cv.notify_all();
            sent_end = gs.eof_;
            turn = (turn + 1) % L;
        }
        if (!sent_end && !relay.aborted()) relay.abort("the stream's end was not reached");
        for (auto &b : box) {
            {
                std::unique_lock<std::mutex> lk(b->m);
                b->cv.wait(lk, [&] { return !b->has; });
                b->quit = true;
            }
            b->cv.notify_all();
        }
        for (auto &t : th) t.join();
        *n_batches = seq, *text_bytes = bytes;
        (void)failed;
        if (relay.aborted()) {
            snprintf(why, why_cap, "%s", relay.why());
            return false;
        }
        return true;
    }
};

// fastq_count / fastq_count_kthread: one bgzip'ed input over the group's lanes.  *unusable: nothing was added.
inline int tally_bgzf_sharded(LaneGroup &g, const char *path, hpn_tally *acc, bool *unusable)
{
    *unusable = false;
    const double t0 = wall_s();
    if (!g.ensure()) {
        *unusable = true;
        return HPN_OK;
    }
    char why[200] = "";
    uint64_t batches = 0, text = 0;
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    if (!BgzfTextFanout::run(g, path, flags, why, sizeof why, &batches, &text)) {
        g.drop_all();
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] %s: BGZF over %d lanes abandoned after %.3f s: %s\n", path, g.lanes(), wall_s() - t0, why);
        *unusable = true;
        return HPN_OK;
    }
    const double t1 = wall_s();
    const char *how = "host";
    const int rc = g.sum_into(acc, &how);
    if (getenv("HPN_TIMING"))
        fprintf(stderr, "[hpn] %s: one BGZF input over %d lanes: %llu batches, %.1f MB of text, inflate + frame + tally %.3f s, sum by %s %.3f s%s\n", path, g.lanes(),
                (unsigned long long)batches, text / 1e6, t1 - t0, how, wall_s() - t1, g.distinct() ? "" : " (lanes share a device)");
    return rc;
}

}  // namespace hpn
