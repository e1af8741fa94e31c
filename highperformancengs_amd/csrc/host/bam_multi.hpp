// bam_multi.hpp -- bam2depth / bam2wig over several GPUs: the per-target loop of the reference (bam2depth.c:325-339)
// is independent per target, so targets are handed out largest first to one worker per device (own context, own
// read-ahead of the compressed file, started at the target's first record as the .bai gives it) and their results
// are written by the caller in target order.  No collective: a chromosome's 1 GB difference array never leaves its
// GPU.  HPN_NGPU=n forces n workers (device = worker % devices): the way this path is exercised on a
// single-GPU box.
#pragma once
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "bam_gpu.hpp"

namespace hpn {

// First record of every target from a .bai (SAM spec 5.2; samtools-0.1.19 bam_index.c:348 bam_index_load_core):
// the smallest chunk start over the target's bins (the metadata pseudo-bin 37450 aside).  ~0 = no record.
inline bool bai_first_offsets(const char *bam, int32_t n_targets, std::vector<uint64_t> &first)
{
    std::string a = std::string(bam) + ".bai", b = bam;
    FILE *f = fopen(a.c_str(), "rb");
    if (!f && b.size() > 3 && b.compare(b.size() - 3, 3, "bam") == 0) {
        b.replace(b.size() - 3, 3, "bai");
        f = fopen(b.c_str(), "rb");
    }
    if (!f) return false;
    char magic[4];
    int32_t n_ref = 0;
    bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "BAI\1", 4) && fread(&n_ref, 4, 1, f) == 1 && n_ref == n_targets;
    first.assign((size_t)(n_targets > 0 ? n_targets : 0), ~0ull);
    for (int32_t t = 0; ok && t < n_ref; ++t) {
        int32_t n_bin = 0;
        ok = fread(&n_bin, 4, 1, f) == 1 && n_bin >= 0;
        for (int32_t k = 0; ok && k < n_bin; ++k) {
            uint32_t bin;
            int32_t n_chunk = 0;
            ok = fread(&bin, 4, 1, f) == 1 && fread(&n_chunk, 4, 1, f) == 1 && n_chunk >= 0;
            for (int32_t c = 0; ok && c < n_chunk; ++c) {
                uint64_t be[2];
                ok = fread(be, 8, 2, f) == 2;
                if (ok && bin != 37450u && be[0] < first[(size_t)t]) first[(size_t)t] = be[0];
            }
        }
        int32_t n_intv = 0;
        ok = ok && fread(&n_intv, 4, 1, f) == 1 && n_intv >= 0 && fseek(f, (long)n_intv * 8, SEEK_CUR) == 0;
    }
    fclose(f);
    return ok;
}

// Where bam_fetch(tid, beg, ...) starts reading: the 16 kb linear index entry of `beg` (every record that overlaps that
// window starts at or behind it; samtools-0.1.19 bam_index.c:608-640), else the target's first record.
// 1: *voffset set; 0: the index is sound and the target holds no record; -1: no usable index (missing, bad magic, other
// target count, truncated) -- the caller then reads the file from its first record.
inline int bai_region_start(const char *bam, int32_t n_targets, int32_t tid, uint32_t beg, uint64_t *voffset)
{
    std::string a = std::string(bam) + ".bai", b = bam;
    FILE *f = fopen(a.c_str(), "rb");
    if (!f && b.size() > 3 && b.compare(b.size() - 3, 3, "bam") == 0) {
        b.replace(b.size() - 3, 3, "bai");
        f = fopen(b.c_str(), "rb");
    }
    if (!f) return -1;
    char magic[4];
    int32_t n_ref = 0;
    bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "BAI\1", 4) && fread(&n_ref, 4, 1, f) == 1 && n_ref == n_targets && tid < n_ref;
    uint64_t first = ~0ull, lin = 0;
    for (int32_t t = 0; ok && t <= tid; ++t) {
        int32_t n_bin = 0;
        ok = fread(&n_bin, 4, 1, f) == 1 && n_bin >= 0;
        for (int32_t k = 0; ok && k < n_bin; ++k) {
            uint32_t bin;
            int32_t n_chunk = 0;
            ok = fread(&bin, 4, 1, f) == 1 && fread(&n_chunk, 4, 1, f) == 1 && n_chunk >= 0;
            for (int32_t c = 0; ok && c < n_chunk; ++c) {
                uint64_t be[2];
                ok = fread(be, 8, 2, f) == 2;
                if (ok && t == tid && bin != 37450u && be[0] < first) first = be[0];
            }
        }
        int32_t n_intv = 0;
        ok = ok && fread(&n_intv, 4, 1, f) == 1 && n_intv >= 0;
        if (ok && t == tid) {
            const int64_t w = (int64_t)(beg >> 14);
            if (w < n_intv && fseek(f, (long)w * 8, SEEK_CUR) == 0 && fread(&lin, 8, 1, f) != 1) lin = 0;
        } else if (ok) {
            ok = fseek(f, (long)n_intv * 8, SEEK_CUR) == 0;
        }
    }
    fclose(f);
    if (!ok) return -1;
    if (first == ~0ull) return 0;
    *voffset = lin > first ? lin : first;
    return 1;
}

inline int multi_gpu_workers()
{
    if (const char *e = getenv("HPN_NGPU")) return atoi(e) > 1 ? atoi(e) : 1;
    int n = 1;
    if (hpn_device_count(&n) != HPN_OK || n < 1) n = 1;
    return n;
}

// ... and for one file: as many workers as the file is worth (cpus.hpp: lanes_worth -- a worker costs ~80 ms to set up, the
// driver makes hardware queues one after the other; a 10 GB BAM is worth two devices, a 93 GB one six).  With ONE device a file of 2 GiB or more still gets three workers on it: one's upload and
// block-table walk run beside another's inflate and a third's record kernels (hg38-shaped 10.6 GB BAM: bam_sliding_count 1.80 ->
// 1.24-1.31 s, bam2depth 1.88 -> 1.77 s; profiles/r03/bench.json).
// by_target: the tool hands whole TARGETS to its workers (bam2depth, bam2wig: each worker seeks to its targets and reads them with
// a stream of its own).  On one device that no longer pays: a stream that reads the file front to back puts four chunks under
// one inflate launch (BgzfGpuStream::next), which a per-target reader cannot (it would inflate its neighbours' blocks behind the
// target's end) -- 10.6 GB BAM, bam2depth: one worker 1.42 s, three 1.56 s; bam_sliding_count's workers take record batches of
// ONE reader in turn and stay at three (1.04 s against 1.19; scripts/e2e_bam_chunk.py).
inline int multi_gpu_workers_for(const char *path, bool /*by_target*/ = false)
{
    int n = multi_gpu_workers();
    if (!getenv("HPN_NGPU")) {
        struct stat sb;
        if (stat(path, &sb) == 0) {
            // (round 3 gave a large file three workers on ONE device here: batches in turn hid the copies behind the kernels.  The
            // one-stream route now reads ahead on a context of its own (host/bam_gpu.hpp) and is the faster one on one device:
            // 0.83 - 0.91 s against 0.96 - 1.04 s on the 10.6 GB file, 2.5 against 3.5 s on the 47 GB one: profiles/r04)
            // (a worker = two contexts, its chunks and stage buffers, ~80 ms, while one device ingests ~32 GB/s of BAM: 2.5 GB)
            n = lanes_worth((uint64_t)sb.st_size, (uint64_t)2560 << 20, n);
        }
    }
    return n;
}

// Worker contexts live for the whole process and are reused across input files: a depth context keeps several GB of
// scratch for a chr1-sized target, and bam2depth / bam2wig / bam_sliding_count come here once per file.
inline hpn_ctx *pooled_worker_ctx(int worker, int device)
{
    static std::mutex m;
    static std::vector<hpn_ctx *> pool;
    std::lock_guard<std::mutex> lk(m);
    if ((size_t)worker >= pool.size()) pool.resize((size_t)worker + 1, nullptr);
    if (!pool[(size_t)worker] && hpn_ctx_create(device, &pool[(size_t)worker]) != HPN_OK) pool[(size_t)worker] = nullptr;
    return pool[(size_t)worker];
}

// One line per process on stderr when a BAM went over several workers: which devices did the work (no collective on this
// path: whole targets per worker / per-window vectors added on the host).
inline void say_workers_once(const char *what, int workers)
{
    static std::atomic<bool> said{false};
    if (workers < 2 || said.exchange(true)) return;
    std::vector<hpn_ctx *> c;
    int devices = 1;
    if (hpn_device_count(&devices) != HPN_OK || devices < 1) devices = 1;
    for (int w = 0; w < workers; ++w) c.push_back(pooled_worker_ctx(w, w % devices));
    char where[1024];
    describe_devices(c.data(), workers, where, sizeof where);
    fprintf(stderr, "[hpn] %s: %d workers on devices %s; results joined on the host%s\n", what, workers, where, workers > devices ? " (workers share a device)" : "");
}

struct TargetOut {
    std::vector<hpn_run> runs;
    std::vector<char> text;      // bedGraph lines formatted on the device (text_name given), instead of runs
    uint64_t n_runs = 0;
    std::vector<uint64_t> win;
    int state = 0;   // 0 pending, 1 ready, -1 failed (the file goes back to the single-stream route)
};

// Runs every target of `path` through hpn_depth_begin / add / finish on `workers` contexts and calls
// emit(j, out) for j = 0, 1, ... in order from the calling thread.  false: not usable for this file (no index
// offsets, a block not decodable on the GPU, no second context) -- nothing has been emitted for targets >= the
// returned *emitted, and the caller runs the remaining work on the single-stream route from scratch.
// bedgraph_text: the workers return the bedGraph lines of a target (hpn_depth_bedgraph_format) instead of its runs.
template <class Emit>
bool depth_targets_multi(const char *path, const BamHeader &hdr, uint32_t mask, uint32_t window, bool want_win, int workers,
                         Emit emit, bool bedgraph_text = false)
{
    const int32_t nt = hdr.n_targets();
    std::vector<uint64_t> first;
    if (nt < 2 || !bai_first_offsets(path, nt, first)) return false;
    int with_records = 0;
    for (uint64_t f : first) with_records += f != ~0ull;
    if (workers > with_records) workers = with_records;      // a context per target that holds records at most
    if (workers < 2) return false;
    std::vector<int32_t> order((size_t)nt);
    for (int32_t j = 0; j < nt; ++j) order[(size_t)j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return hdr.target_len[a] > hdr.target_len[b]; });
    std::vector<TargetOut> out((size_t)nt);
    std::mutex m;
    std::condition_variable cv;
    size_t next = 0;
    bool failed = false;
    int devices = 1;
    if (hpn_device_count(&devices) != HPN_OK || devices < 1) devices = 1;
    // Finished targets wait in host memory until every earlier one is written (a chr1 at 30x is ~2.6 GB of bedGraph
    // text): a worker takes a target only while the estimated bytes of everything claimed and not yet emitted stay
    // within a budget -- except the target the writer is waiting for, and anything at all when nothing is held back.
    const uint64_t budget = getenv("HPN_DEPTH_LOOKAHEAD") ? strtoull(getenv("HPN_DEPTH_LOOKAHEAD"), nullptr, 10) : (uint64_t)12 << 30;
    auto estimate = [&](int32_t j) { return (uint64_t)hdr.target_len[j] * 12u; };
    std::vector<char> claimed((size_t)nt, 0);
    uint64_t outstanding = 0, peak_outstanding = 0, budget_waits = 0;
    int32_t emit_at = 0;   // the target the writer will emit next
    auto work = [&](int w) {
        hpn_ctx *ctx = pooled_worker_ctx(w, w % devices);
        bool ok = ctx != nullptr;
        BgzfGpuStream gs;
        BamHeader h2;
        ok = ok && gs.open(ctx, path, h2, 3, BgzfGpuStream::kLaunchChunk) && h2.n_targets() == nt;
        for (;;) {
            int32_t j = -1;
            {
                std::unique_lock<std::mutex> lk(m);
                if (!ok) failed = true;
                for (;;) {
                    if (failed || next >= order.size()) break;
                    // the target the writer is waiting for must be worked on whatever the budget says; so may anything when nothing
                    // is held back; every other target only while it fits -- else wait until emit() has given bytes back
                    if (emit_at < nt && !claimed[(size_t)emit_at]) j = emit_at;
                    for (size_t k = 0; j < 0 && k < order.size(); ++k) {   // largest first
                        const int32_t t = order[k];
                        if (!claimed[(size_t)t] && (outstanding == 0 || outstanding + estimate(t) <= budget)) j = t;
                    }
                    if (j >= 0) break;
                    ++budget_waits;
                    cv.wait(lk);
                }
                if (j < 0) break;
                claimed[(size_t)j] = 1, ++next, outstanding += estimate(j);
                if (outstanding > peak_outstanding) peak_outstanding = outstanding;
            }
            TargetOut &o = out[(size_t)j];
            int rc = hpn_depth_begin_w(ctx, j, hdr.target_len[j], mask, want_win ? window : 0u);
            if (rc == HPN_OK && first[(size_t)j] != ~0ull) {
                ok = gs.seek(first[(size_t)j]);
                hpn_raw_info info;
                int r;
                while (ok && rc == HPN_OK && (r = gs.next(&info)) != 0) {
                    if (r < 0) {
                        ok = false;
                        break;
                    }
                    if (!info.n_records) continue;
                    if (info.tid_min <= j && j <= info.tid_max) rc = hpn_depth_add_raw_dev(ctx, gs.d_raw());
                    if (info.tid_min > j || info.tid_max > j || info.tid_min < 0) break;   // past the target (or into the unmapped tail)
                }
            }
            if (ok && rc == HPN_OK && bedgraph_text) {
                o.win.assign(want_win ? (size_t)hdr.target_len[j] / window + 1 : 0, 0);
                uint64_t nbytes = 0;
                rc = hpn_depth_finish(ctx, window, nullptr, 0, &o.n_runs, want_win ? o.win.data() : nullptr);
                if (rc == HPN_OK) rc = hpn_depth_bedgraph_format(ctx, hdr.target_name[j].c_str(), &nbytes);
                if (rc == HPN_OK) {
                    o.text.resize(nbytes);
                    rc = hpn_depth_bedgraph_read(ctx, 0, o.text.data(), nbytes);
                }
            } else if (ok && rc == HPN_OK) {
                o.win.assign(want_win ? (size_t)hdr.target_len[j] / window + 1 : 0, 0);
                if (o.runs.empty()) o.runs.resize(1u << 20);
                rc = hpn_depth_finish(ctx, window, o.runs.data(), o.runs.size(), &o.n_runs, want_win ? o.win.data() : nullptr);
                if (rc == HPN_E_CAPACITY) {
                    o.runs.resize(o.n_runs);
                    rc = hpn_depth_finish(ctx, window, o.runs.data(), o.runs.size(), &o.n_runs, want_win ? o.win.data() : nullptr);
                }
            }
            if (rc != HPN_OK && rc != 1) {
                fprintf(stderr, "[hpn] target %d on worker %d: %s\n", j, w, ctx ? hpn_ctx_last_error(ctx) : "no context");
                ok = false;
            }
            {
                std::lock_guard<std::mutex> lk(m);
                o.state = ok ? 1 : -1;
                if (!ok) failed = true;
            }
            cv.notify_all();
        }
        cv.notify_all();
        // the pooled context stays for the next file (and is left to process exit: tearing one down costs as much as making it)
    };
    std::vector<std::thread> th;
    for (int w = 0; w < workers; ++w) th.emplace_back(work, w);
    bool good = true;
    for (int32_t j = 0; j < nt && good; ++j) {
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return out[(size_t)j].state != 0 || failed; });
            good = out[(size_t)j].state == 1;
        }
        if (good) {
            emit(j, out[(size_t)j]);
            std::vector<hpn_run>().swap(out[(size_t)j].runs);   // a chromosome's runs are ~1 GB: give them back
            std::vector<char>().swap(out[(size_t)j].text);
            {
                std::lock_guard<std::mutex> lk(m);
                outstanding -= estimate(j);
                emit_at = j + 1;
            }
            cv.notify_all();
        }
    }
    if (!good) {
        std::lock_guard<std::mutex> lk(m);
        failed = true;
    }
    cv.notify_all();
    for (auto &t : th) t.join();
    if (getenv("HPN_TIMING"))
        fprintf(stderr, "[hpn] look-ahead: at most %.1f MB of estimated output claimed and not yet written (budget %.1f MB), %llu waits for the writer\n",
                peak_outstanding / 1e6, budget / 1e6, (unsigned long long)budget_waits);
    return good;
}

// One file, batches of whole BGZF blocks handed to `workers` contexts in turn (bam_sliding_count: records are
// independent): ONE reader walks the block headers (BgzfGpuStream's host side), every worker owns a
// context, device buffers and the inflate + record index of the batches it is given.
//   setup(w, ctx) once per worker, batch(w, ctx, d_raw, info) per indexed batch on the worker's thread,
//   finish(w, ctx) at the end; each returns false to abandon.  false: the file is not usable this way (a block that
//   does not start at a record, damage, a context that cannot be made) -- the caller starts over on one stream.
class BgzfFanout {
public:
    template <class Setup, class Batch, class Finish>
    static bool run(const char *path, int workers, Setup setup, Batch batch, Finish finish)
    {
        int devices = 1;
        if (hpn_device_count(&devices) != HPN_OK || devices < 1) devices = 1;
        struct Box {
            std::mutex m;
            std::condition_variable cv;
            bool has = false, quit = false;
            TextPump::Chunk c;
            BgzfParsed pb;
        };
        std::vector<std::unique_ptr<Box>> box;
        for (int w = 0; w < workers; ++w) box.emplace_back(new Box());
        std::vector<hpn_ctx *> ctxs((size_t)workers, nullptr);
        for (int w = 0; w < workers; ++w)
            if (!(ctxs[(size_t)w] = pooled_worker_ctx(w, w % devices))) return false;
        BgzfGpuStream gs;                       // the reader: header, read-ahead, block tables
        BamHeader hdr;
        if (!gs.open(ctxs[0], path, hdr, workers + 2, BgzfGpuStream::kLaunchChunk)) return false;
        std::atomic<bool> failed{false};
        std::vector<std::thread> th;
        for (int w = 0; w < workers; ++w)
            th.emplace_back([&, w] {
                hpn_ctx *ctx = ctxs[(size_t)w];
                bind_thread_near(ctx);
                BgzfDevice dev(ctx);
                bool ok = setup(w, ctx);
                Box &b = *box[(size_t)w];
                for (;;) {
                    std::unique_lock<std::mutex> lk(b.m);
                    b.cv.wait(lk, [&] { return b.has || b.quit; });
                    if (!b.has) break;
                    lk.unlock();
                    hpn_raw_info info;
                    if (ok && !failed) {
                        ok = dev.run(b.pb, false, &info) == 1;
                        if (ok && info.n_records) ok = batch(w, ctx, dev.d_raw(), info);
                    }
                    gs.pump_->recycle(b.c);
                    if (!ok) failed = true;
                    lk.lock();
                    b.has = false;
                    lk.unlock();
                    b.cv.notify_all();
                }
                if (ok && !failed) ok = finish(w, ctx);
                if (!ok) failed = true;
            });
        int turn = 0;
        for (;;) {                              // BgzfGpuStream::next, with the device side handed out
            TextPump::Chunk c;
            if (failed || gs.eof_ || !gs.pump_->next(c)) {
                if (!gs.carry_.empty()) failed = true;   // a partial block at the end: truncated file
                break;
            }
            if (c.eof) gs.eof_ = true;
            size_t at = 0;
            if (gs.skip_) {
                const size_t k = gs.skip_ < c.n ? (size_t)gs.skip_ : c.n;
                gs.skip_ -= k, at = k;
            }
            if (at == c.n) {
                gs.pump_->recycle(c);
                continue;
            }
            BgzfParsed pb;
            if (gs.parse(c, at, pb) != 1) {     // not BGZF, or the file ends inside a block
                gs.pump_->recycle(c);
                failed = true;
                break;
            }
            if (pb.blocks.empty()) {
                gs.pump_->recycle(c);
                continue;
            }
            Box &b = *box[(size_t)turn];
            {
                std::unique_lock<std::mutex> lk(b.m);
                b.cv.wait(lk, [&] { return !b.has; });
                b.c = c, b.pb = std::move(pb), b.has = true;
            }
            b.cv.notify_all();
            turn = (turn + 1) % workers;
        }
        for (auto &b : box) {
            {
                std::unique_lock<std::mutex> lk(b->m);
                b->cv.wait(lk, [&] { return !b->has; });
                b->quit = true;
            }
            b->cv.notify_all();
        }
        for (auto &t : th) t.join();
        return !failed;
    }
};

}  // namespace hpn
