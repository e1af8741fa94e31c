// text_shard.hpp -- ONE FASTQ input sharded by record block over several GPU contexts ("lanes").
//
// The reference's parallelism stops at whole files (fastq_count.c:213-230, fastq_count_kthread.c:270 ->
// klib/kthread.c:34-60), and fastq_trim is single-threaded (fastq_trim.c:91-108): a single 300 GB FASTQ
// uses one core there and would use one GPU and one PCIe link here.  SURVEY.md 8e / BASELINE north_star:
// "records shard trivially by block across the GPUs of one node with a final all-reduce of the small count
// vectors".  This file is that, inside the C tools:
//
//   reader (TextPump)      the stream's bytes in pieces cut ANYWHERE (no host pass over the text);
//   dispatcher             gives piece j the byte in front of it and a 4 KiB tail of piece j+1 (two small
//                          host copies) and queues it; any idle lane takes it;
//   lane = thread + ctx    hpn_fastq_text_piece_lines (copy over the lane's own PCIe link, line index) ->
//                          publishes the piece's line count on the board -> waits for the pieces before it
//                          (a chain of integers, nothing else is exchanged) -> hpn_fastq_text_piece_count /
//                          _trim frames the records the piece owns and tallies / cuts them;
//   sum                    fastq_count: ONE sum of the lanes' count vectors, where reduceStats' element-wise
//                          sum stands (fastq_count_kthread.c:180-210): a grouped RCCL all-reduce over xGMI
//                          when the lanes sit on distinct devices (hpn_comm_init_all, started in the
//                          background while the file streams), else added on the host;
//                          fastq_trim: no collective, the lanes' output slabs are written in piece order.
//
// Anything irregular anywhere (include/hpngs.h: HPN_TEXT_*) abandons the route: nothing is added / the
// output is rewound, and the caller frames the whole input on one context as before.
// HPN_NGPU=n forces n lanes on whatever devices exist (lane % devices) -- how the route runs on a one-GPU
// box; HPN_ALLREDUCE=host|rccl overrides the choice of the sum.
#pragma once
#include <sys/stat.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "text_stream.hpp"

namespace hpn {

// The contexts one input is sharded over.  Lane 0 is the caller's own context; the others are made on
// first use (in parallel: the driver serialises queue creation at 15-30 ms each) and kept for later files.
class LaneGroup {
public:
    // own = the worker's context on device base + rel; lane k on device base + (rel + k) % ndev
    LaneGroup(hpn_ctx *own, int base, int rel, int ndev, int lanes) : base_(base), rel_(rel), ndev_(ndev < 1 ? 1 : ndev)
    {
        ctx_.assign((size_t)(lanes < 1 ? 1 : lanes), nullptr);
        ctx_[0] = own;
        distinct_ = (int)ctx_.size() <= ndev_;
    }
    ~LaneGroup()
    {
        if (comm_thread_.joinable()) comm_thread_.join();
    }
    int lanes() const { return (int)ctx_.size(); }
    hpn_ctx *ctx(int lane) const { return ctx_[(size_t)lane]; }
    bool distinct() const { return distinct_; }

    // Makes the missing contexts; false: some device refused (the caller stays on one context).
    bool ensure()
    {
        if (ready_) return true;
        std::vector<std::thread> th;
        std::atomic<bool> ok{true};
        for (size_t k = 1; k < ctx_.size(); ++k)
            th.emplace_back([&, k] {
                if (!ctx_[k] && hpn_ctx_create(base_ + (rel_ + (int)k) % ndev_, &ctx_[k]) != HPN_OK) ok = false;
            });
        for (auto &t : th) t.join();
        ready_ = ok;
        if (ready_ && want_rccl() && !comm_started_) {  // communicator set-up (seconds on 8 devices) hides behind the streaming
            comm_started_ = true;
            comm_thread_ = std::thread([this] {
                const double t0 = wall_s();
                comm_rc_ = hpn_comm_init_all(ctx_.data(), (int)ctx_.size());
                comm_s_ = wall_s() - t0;
            });
        }
        return ready_;
    }

    // Sum of the lanes' count vectors into `acc` (their device accumulators are zero afterwards).
    // how: what did it ("rccl" / "host"), for the route's report on stderr.
    int sum_into(hpn_tally *acc, const char **how)
    {
        if (comm_thread_.joinable()) comm_thread_.join();
        *how = "host";
        if (comm_started_ && comm_rc_ == HPN_OK) {
            std::vector<uint64_t *> vec(ctx_.size(), nullptr);
            int rc = HPN_OK;
            for (size_t k = 0; k < ctx_.size() && rc == HPN_OK; ++k) rc = hpn_fastq_tally_devptr(ctx_[k], &vec[k]);
            if (rc == HPN_OK) rc = hpn_allreduce_u64_all(ctx_.data(), vec.data(), (int)ctx_.size(), HPN_TALLY_WORDS);
            if (rc == HPN_OK) {  // every lane now holds the sum: lane 0's is fetched, the others' are dropped
                *how = "rccl";
                say_once("rccl");
                rc = hpn_fastq_tally_fetch(ctx_[0], acc);
                for (size_t k = 1; k < ctx_.size(); ++k) drop(ctx_[k]);
                return rc;
            }
            if (rc == HPN_E_PARTIAL) {   // some lanes may hold sums already: adding the vectors again would count records twice
                fprintf(stderr, "[hpn] RCCL all-reduce failed half-way (%s): this input is abandoned\n", hpn_ctx_last_error(ctx_[0]));
                drop_all();
                return rc;
            }
            fprintf(stderr, "[hpn] RCCL all-reduce failed (%s): the lanes' vectors are added on the host\n", hpn_ctx_last_error(ctx_[0]));
        }
        say_once("host");
        int rc = HPN_OK;
        for (size_t k = 0; k < ctx_.size(); ++k) {  // hpn_fastq_tally_fetch ADDS: the host sum is the fetch itself
            const int r = hpn_fastq_tally_fetch(ctx_[k], acc);
            if (r != HPN_OK && rc == HPN_OK) rc = r, first_bad_ = (int)k;
        }
        return rc;
    }
    void drop_all()
    {
        for (hpn_ctx *c : ctx_)
            if (c) drop(c);
    }
    hpn_ctx *failing_ctx() const { return ctx_[(size_t)first_bad_]; }
    double comm_seconds() const { return comm_s_; }
    int comm_status() const { return comm_started_ ? comm_rc_ : 1; }

private:
    // One line per process on stderr when an input went over several lanes: which devices, and who added the counts.
    void say_once(const char *how)
    {
        static std::atomic<bool> said{false};
        if (ctx_.size() < 2 || said.exchange(true)) return;
        char where[1024];
        describe_devices(ctx_.data(), (int)ctx_.size(), where, sizeof where);
        int ranks = 0;
        if (!strcmp(how, "rccl")) (void)hpn_comm_count(ctx_[0], &ranks);
        if (ranks) fprintf(stderr, "[hpn] %d lanes on devices %s; counts summed by RCCL (%d ranks)\n", (int)ctx_.size(), where, ranks);
        else fprintf(stderr, "[hpn] %d lanes on devices %s; counts summed on the host%s\n", (int)ctx_.size(), where, distinct_ ? "" : " (lanes share a device)");
    }
    static void drop(hpn_ctx *c)
    {
        hpn_tally scratch;
        memset(&scratch, 0, sizeof scratch);
        (void)hpn_fastq_tally_fetch(c, &scratch);
    }
    bool want_rccl() const
    {
        const char *e = getenv("HPN_ALLREDUCE");
        if (e && !strcmp(e, "host")) return false;
        // one RCCL rank per device: lanes that share a device add on the host (HPN_COMM_SHARED_DEVICE=1, tests: hpn_comm_init_all
        // is asked all the same -- the RCCL stand-in of tests/stub accepts them, the real library refuses and the host adds)
        const char *sh = test_env("HPN_COMM_SHARED_DEVICE");
        return (distinct_ || (sh && sh[0] == '1')) && ctx_.size() > 1;
    }
    std::vector<hpn_ctx *> ctx_;
    int base_, rel_, ndev_;
    bool distinct_ = false, ready_ = false, comm_started_ = false;
    int comm_rc_ = HPN_E_RCCL, first_bad_ = 0;
    double comm_s_ = 0;
    std::thread comm_thread_;
};

// How many lanes for this input.  HPN_NGPU=n: n, whatever the input.  Otherwise: a plain regular file, or -- for callers that
// have a route which inflates on the lanes' own devices (gz_lanes: the count tools, host/gz_shard.hpp / bgzf_shard.hpp) -- a
// gzip file of 256 MiB or more, as many lanes as the file is worth (cpus.hpp: lanes_worth); for every other caller compressed input is ONE lane (fastq_trim: a
// single host zlib reader would feed N lanes, which adds N contexts and nothing else).  Plain: as many lanes as the file is worth (lanes_worth) -- a context costs 15-30 ms to make
// and one lane already streams at the PCIe rate of its link --, at most `devices_for_me`.
// pair_on_one_device (the count tools): with a single device, a plain file of 4 GiB or more still gets TWO lanes on it -- one lane's
// copy over PCIe then runs beside the other's framing and tally (15.2 GB: 0.42 s against 0.49-0.53 s on one context; three or
// four lanes only add contexts: scripts/e2e_lanes.sh).
inline int shard_lanes_for(const char *path, int devices_for_me, bool pair_on_one_device = false, bool gz_lanes = false)
{
    if (!text_path_enabled()) return 1;
    if (const char *e = getenv("HPN_NGPU")) return atoi(e) > 1 ? atoi(e) : 1;
    if (devices_for_me < 2 && !pair_on_one_device) return 1;
    struct stat sb;
    if (strncmp(path, "-", 1) == 0 || !strcmp(path, "") || stat(path, &sb) != 0 || !S_ISREG(sb.st_mode)) return 1;
    uint8_t magic[2] = {0, 0};
    const int fd = open(path, O_RDONLY);
    const ssize_t k = fd >= 0 ? pread(fd, magic, 2, 0) : 0;
    if (fd >= 0) close(fd);
    if (k == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        // gzip: the device inflate is the wall, and its batches spread over the devices (host/gz_shard.hpp, bam_gpu.hpp's
        // block fan-out for BGZF) -- from 256 MiB on, as many lanes as the file is worth
        if (!gz_lanes || devices_for_me < 2 || sb.st_size < ((off_t)256 << 20)) return 1;
        // (a lane here = two contexts and the symbol scratch, ~0.1 s, while one device takes ~20 GB/s of compressed bytes)
        return lanes_worth((uint64_t)sb.st_size, (uint64_t)2 << 30, devices_for_me);
    }
    if (devices_for_me < 2) return sb.st_size >= ((off_t)4 << 30) ? 2 : 1;
    // (a lane = a context, a reader and its pinned chunks, ~70 ms, while one device streams ~45 GB/s of plain text: 3.2 GB)
    return lanes_worth((uint64_t)sb.st_size, (uint64_t)3200 << 20, devices_for_me);
}

// What one worker (one kt_for worker of the reference = one thread + its own context) may spread an input over: the
// node's devices divided among the workers in flight.  for_file: the group to shard `path` over, nullptr = one context.
class WorkerLanes {
public:
    static int cap(int ndev, int workers)
    {
        if (const char *e = getenv("HPN_NGPU")) return atoi(e) > 1 ? atoi(e) : 1;
        const int c = ndev / (workers < 1 ? 1 : workers);
        return c < 1 ? 1 : c;
    }
    // device of worker w's own context, relative to the first device: the workers' device ranges lie side by side
    static int device_of(int ndev, int workers, int w) { return (w * (getenv("HPN_NGPU") ? 1 : cap(ndev, workers))) % ndev; }

    // own sits on device base + rel of the ndev devices base .. base + ndev - 1
    // gz_lanes: the caller can inflate one gzip input on several devices (shard_lanes_for)
    WorkerLanes(hpn_ctx *own, int base, int rel, int ndev, int cap, bool pair_on_one_device = false, bool gz_lanes = false)
        : own_(own), base_(base), rel_(rel), ndev_(ndev), cap_(cap), pair_(pair_on_one_device), gz_lanes_(gz_lanes)
    {
    }
    LaneGroup *for_file(const char *path)
    {
        const int want = shard_lanes_for(path, cap_, pair_, gz_lanes_);
        if (want < 2) return nullptr;
        if (!group_) group_.reset(new LaneGroup(own_, base_, rel_, ndev_, want));   // sized by the first input that is sharded, kept for the others
        return group_.get();
    }

private:
    hpn_ctx *own_;
    int base_, rel_, ndev_, cap_;
    bool pair_, gz_lanes_;
    std::unique_ptr<LaneGroup> group_;
};

// Piece size of the sharded route: the chunk size of the plain route, never below two tails.
inline size_t shard_piece_bytes(int lanes)
{
    const char *e = test_env("HPN_TEXT_CHUNK");
    size_t c = e && atoll(e) >= 64 ? (size_t)atoll(e) : (size_t)32 << 20;
    if (!e && lanes > 4) c = (size_t)16 << 20;   // (lanes + 2) pinned buffers: keep the footprint near 200 MB
    return c < 2 * HPN_TEXT_PIECE_TAIL ? 2 * HPN_TEXT_PIECE_TAIL : c;
}

// The machinery both tools share: reader -> dispatcher -> lanes, with the board of line counts.
// Work: int lines_done(lane, ctx, piece_seq, lines_before, own_bytes + tail, &info) -- runs the second half of the piece.
class PieceRun {
public:
    struct Piece {
        TextPump::Chunk c;
        uint64_t seq = 0;
        uint32_t head = 0;
        size_t tail = 0;
        bool last = false;
    };

    PieceRun(LaneGroup &g, const char *path, size_t piece_bytes)
        : g_(g), pump_(g.ctx(0), path, piece_bytes, g.lanes() + 2, false, 0, HPN_TEXT_PIECE_TAIL + 64, reader_threads(g.lanes()))
    {
    }
    // ONE reader feeds every lane: four pread threads per lane (a lane's PCIe link takes ~25 GB/s, a thread copies ~7 GB/s out
    // of the page cache), within this worker's share of the host's CPUs
    static int reader_threads(int lanes)
    {
        const long share = usable_cpus() / text_workers_in_flight();
        long n = 4L * lanes;
        n = n < 6 ? 6 : n;
        n = n > share ? share : n;
        return (int)(n < 1 ? 1 : n);
    }
    bool ok() const { return pump_.ok(); }
    size_t piece_bytes() const { return pump_.chunk_bytes(); }
    bool irregular() const { return irregular_ || pump_.damaged(); }   // (a damaged gzip stream is for zlib's own reader: tally_file / fastq_trim)
    int status() const { return rc_; }
    hpn_ctx *failing_ctx() const { return bad_ctx_; }
    uint64_t pieces() const { return n_pieces_; }
    uint64_t bytes() const { return n_bytes_; }
    uint64_t records() const { return n_records_; }
    // Called once, by the lane that stops the route, after the reason is recorded: whatever ELSE a lane may be blocked in
    // (fastq_trim: OrderedWriter::acquire behind a slab that will now never come) must be woken from here.
    void on_stop(std::function<void()> f) { on_stop_ = std::move(f); }

    // second(lane, ctx, piece, lines_before, info): the piece's second half (count or trim + hand-over of its output);
    // returns an hpn status.  Runs until the stream ends or something stops the route.
    template <class Second>
    void run(Second second)
    {
        std::vector<std::thread> th;
        for (int l = 0; l < g_.lanes(); ++l)
            th.emplace_back([this, l, &second] {
                bind_thread_near(g_.ctx(l));      // (a lane's thread and the reader it starts: next to the lane's device)
                lane(l, second);
            });
        dispatch();
        for (auto &t : th) t.join();
    }

private:
    void dispatch()
    {
        Piece prev;
        bool have = false;
        uint64_t seq = 0;
        TextPump::Chunk c;
        while (!stopped() && pump_.next(c)) {
            if (have) {
                if (c.n == 0) {  // the stream ended exactly at prev's end
                    pump_.recycle(c);
                    break;
                }
                const size_t t = c.n < HPN_TEXT_PIECE_TAIL ? c.n : (size_t)HPN_TEXT_PIECE_TAIL;
                memcpy(prev.c.p + prev.c.n, c.p, t);   // prev's tail
                c.p[-1] = prev.c.p[prev.c.n - 1];      // this piece's head byte
                prev.tail = t;
                push(prev);
            }
            prev = Piece();
            prev.c = c, prev.seq = seq++, prev.head = have ? 1u : 0u;
            have = true;
            if (c.eof) break;
        }
        if (have && !stopped()) {
            prev.last = true;
            push(prev);
        } else if (have) {
            pump_.recycle(prev.c);
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            closed_ = true;
        }
        cv_.notify_all();
    }
    void push(const Piece &p)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return (int)q_.size() < g_.lanes() || stop_; });
        if (stop_) {
            lk.unlock();
            pump_.recycle(p.c);
            return;
        }
        q_.push_back(p);
        lk.unlock();
        cv_.notify_all();
    }
    bool stopped()
    {
        std::lock_guard<std::mutex> lk(m_);
        return stop_;
    }
    void stop(int rc, hpn_ctx *ctx, bool irregular)
    {
        bool first = false;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!stop_) {   // the first reason stands: a lane woken by the stop reports HPN_E_STATE afterwards, which is not why
                stop_ = first = true;
                irregular_ = irregular;
                rc_ = rc;
                bad_ctx_ = ctx;
            }
        }
        cv_.notify_all();
        if (first && on_stop_) on_stop_();
    }

    template <class Second>
    void lane(int l, Second &second)
    {
        hpn_ctx *ctx = g_.ctx(l);
        for (;;) {
            Piece p;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !q_.empty() || closed_ || stop_; });
                if (stop_) {
                    for (const Piece &x : q_) pump_.recycle(x.c);
                    q_.clear();
                    return;
                }
                if (q_.empty()) return;   // closed and drained
                p = q_.front();
                q_.pop_front();
            }
            cv_.notify_all();
            hpn_text_piece pl;
            int rc = hpn_fastq_text_piece_lines(ctx, p.c.p - p.head, p.head + p.c.n + p.tail, p.head, p.c.n, p.last, &pl);
            pump_.recycle(p.c);   // the text is on the device
            if (rc != HPN_OK || pl.irregular) {
                stop(rc, ctx, rc == HPN_OK);
                return;
            }
            uint64_t before = 0;
            {   // the board: lines in front of piece j's text = sum of the pieces before it
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return board_.size() > p.seq || stop_; });
                if (stop_) return;
                before = board_[p.seq];
                board_.push_back(before + pl.n_lines);
            }
            cv_.notify_all();
            hpn_text_info info;
            rc = second(l, ctx, p, before, &info);
            if (rc != HPN_OK || info.irregular) {
                stop(rc, ctx, rc == HPN_OK);
                return;
            }
            std::lock_guard<std::mutex> lk(m_);
            ++n_pieces_, n_bytes_ += p.c.n, n_records_ += info.n_records;
        }
    }

    LaneGroup &g_;
    TextPump pump_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Piece> q_;
    std::vector<uint64_t> board_{0};
    bool closed_ = false, stop_ = false, irregular_ = false;
    int rc_ = HPN_OK;
    hpn_ctx *bad_ctx_ = nullptr;
    std::function<void()> on_stop_;
    uint64_t n_pieces_ = 0, n_bytes_ = 0, n_records_ = 0;
};

// fastq_count / fastq_count_kthread: one input over the group's lanes.  *irregular (or a lane that could not be made):
// nothing was added to acc and the caller takes the one-context routes.
inline int tally_text_sharded(LaneGroup &g, const char *path, hpn_tally *acc, bool *irregular)
{
    *irregular = false;
    const double t0 = wall_s();
    if (!g.ensure()) {
        *irregular = true;
        return HPN_OK;
    }
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    PieceRun run(g, path, shard_piece_bytes(g.lanes()));
    if (!run.ok()) return HPN_E_NOMEM;
    const double t1 = wall_s();
    run.run([&](int, hpn_ctx *ctx, const PieceRun::Piece &, uint64_t before, hpn_text_info *info) {
        return hpn_fastq_text_piece_count(ctx, before, flags, info);
    });
    const double t2 = wall_s();
    if (run.irregular() || run.status() != HPN_OK) {
        g.drop_all();
        if (run.status() != HPN_OK) {
            fprintf(stderr, "[hpn] %s: a lane failed: %s\n", path, run.failing_ctx() ? hpn_ctx_last_error(run.failing_ctx()) : "?");
            return run.status();
        }
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] %s: %d lanes  (abandoned: irregular text)\n", path, g.lanes());
        *irregular = true;
        return HPN_OK;
    }
    const char *how = "host";
    const int rc = g.sum_into(acc, &how);
    if (getenv("HPN_TIMING"))
        fprintf(stderr, "[hpn] %s: one input over %d lanes: %llu pieces, %.1f MB, setup %.3f s, stream %.3f s, sum by %s %.3f s%s\n", path,
                g.lanes(), (unsigned long long)run.pieces(), run.bytes() / 1e6, t1 - t0, t2 - t1, how, wall_s() - t2,
                g.distinct() ? "" : " (lanes share a device)");
    return rc;
}

// Output slabs of several lanes written to one FILE in piece order (fastq_trim; no collective).
class OrderedWriter {
public:
    OrderedWriter(hpn_ctx *ctx, FILE *out, size_t cap, int lanes) : ctx_(ctx), out_(out)
    {
        lane_free_.resize((size_t)lanes);
        for (int l = 0; l < lanes; ++l)
            for (int k = 0; k < 2; ++k) {   // two slabs per lane: one being written while the next is filled
                void *p = nullptr;
                if (hpn_host_malloc(ctx_, cap, &p) != HPN_OK) return;
                buf_.push_back(p);
                lane_free_[(size_t)l].push_back((int)buf_.size() - 1);
            }
        ok_ = true;
        th_ = std::thread([this] { loop(); });
    }
    ~OrderedWriter()
    {
        finish();
        for (void *p : buf_) hpn_host_free(ctx_, p);
    }
    bool ok() const { return ok_; }
    void *acquire(int lane, int *idx)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !lane_free_[(size_t)lane].empty() || stop_ || cancel_; });
        if (cancel_ || lane_free_[(size_t)lane].empty()) return nullptr;
        *idx = lane_free_[(size_t)lane].front();
        lane_free_[(size_t)lane].pop_front();
        return buf_[(size_t)*idx];
    }
    void submit(uint64_t seq, int lane, int idx, size_t n)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            todo_[seq] = Job{lane, idx, n};
        }
        cv_.notify_all();
    }
    void give_back(int lane, int idx)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            lane_free_[(size_t)lane].push_back(idx);
        }
        cv_.notify_all();
    }
    // The route is being abandoned: a lane waiting for a slab (both of its own queued behind a sequence number that
    // will never be submitted) gets nullptr instead of waiting for ever; what is already in sequence is still written.
    void cancel()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            cancel_ = true;
        }
        cv_.notify_all();
    }
    void finish()   // every slab submitted in sequence is in the FILE when this returns
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
    struct Job {
        int lane, idx;
        size_t n;
    };
    void loop()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return todo_.count(next_) || stop_; });
                auto it = todo_.find(next_);
                if (it == todo_.end()) return;   // stopped, and the next slab in sequence never came
                j = it->second;
                todo_.erase(it);
                ++next_;
            }
            if (j.n && !failed_ && !write_slab(out_, buf_[(size_t)j.idx], j.n)) failed_ = true;
            give_back(j.lane, j.idx);
        }
    }
    hpn_ctx *ctx_;
    FILE *out_;
    bool ok_ = false, stop_ = false, cancel_ = false;
    std::atomic<bool> failed_{false};
    std::vector<void *> buf_;
    std::vector<std::deque<int>> lane_free_;
    std::map<uint64_t, Job> todo_;
    uint64_t next_ = 0;
    std::mutex m_;
    std::condition_variable cv_;
    std::thread th_;
};

// fastq_trim: one input over the group's lanes, the trimmed text written to `out` in stream order.
// *irregular: the route was abandoned -- `out` may hold a prefix of the result: the caller rewinds it.
inline int trim_text_sharded(LaneGroup &g, const char *path, int32_t S, int32_t E, FILE *out, unsigned long *reads, bool *irregular)
{
    *irregular = false;
    const double t0 = wall_s();
    if (!g.ensure()) {
        *irregular = true;
        return HPN_OK;
    }
    PieceRun run(g, path, shard_piece_bytes(g.lanes()));
    if (!run.ok()) return HPN_E_NOMEM;
    const size_t ocap = run.piece_bytes() + HPN_TEXT_PIECE_TAIL + 8192 + 64;
    OrderedWriter writer(g.ctx(0), out, ocap, g.lanes());
    if (!writer.ok()) return HPN_E_NOMEM;
    run.on_stop([&writer] { writer.cancel(); });
    const double t1 = wall_s();
    run.run([&](int lane, hpn_ctx *ctx, const PieceRun::Piece &p, uint64_t before, hpn_text_info *info) {
        int oi = -1;
        void *obuf = writer.acquire(lane, &oi);
        if (!obuf) return (int)HPN_E_STATE;
        const int rc = hpn_fastq_text_piece_trim(ctx, before, S, E, obuf, ocap, info);
        if (rc != HPN_OK || info->irregular) writer.give_back(lane, oi);
        else writer.submit(p.seq, lane, oi, info->n_bytes);
        return rc;
    });
    const double t2 = wall_s();
    writer.finish();
    if (writer.failed()) {          // (ENOSPC, EIO, a closed pipe: the output is incomplete -- not something to start over from)
        fprintf(stderr, "fastq_trim: writing the output failed\n");
        return HPN_E_STATE;
    }
    if (run.status() != HPN_OK) {
        fprintf(stderr, "[hpn] %s: a lane failed: %s\n", path, run.failing_ctx() ? hpn_ctx_last_error(run.failing_ctx()) : "?");
        return run.status();
    }
    if (run.irregular()) {
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] %s: %d lanes  (abandoned: irregular text)\n", path, g.lanes());
        *irregular = true;
        return HPN_OK;
    }
    *reads += run.records();
    if (getenv("HPN_TIMING"))
        fprintf(stderr, "[hpn] %s: one input over %d lanes: %llu pieces, %.1f MB, setup %.3f s, stream %.3f s, final drain %.3f s%s\n", path, g.lanes(),
                (unsigned long long)run.pieces(), run.bytes() / 1e6, t1 - t0, t2 - t1, wall_s() - t2, g.distinct() ? "" : " (lanes share a device)");
    return HPN_OK;
}

}  // namespace hpn
