// bam2depth -- drop-in for the reference tool of the same name (bam2depth.c): per-position
// coverage of CIGAR-M blocks as bedGraph + per-window mean depth, the record loop and the
// breakpoint sweep running on MI355X through libhpngs.
//
//   bam2depth [-o OUT] [-w W] [-W] [-r REGION] [-s INT] [-h] a.bam b.bam ...
//
// Writes ./<basename(bam)>.<n>.bedGraph, OUT.<n>.depth and with -W OUT.<n>.wig +
// OUT.<n>.chromSize.txt (bam2depth.c:312-321).  Like the reference it insists on
// <bam>.bai (:112-119) although it reads the coordinate-sorted file front to back:
// bam_fetch(tid, 0, 1<<29) per target visits exactly the target's records.
// -r / -s are parsed and unused, as in the reference (:281-285).
#include <err.h>
#include <getopt.h>
#include <fcntl.h>
#include <libgen.h>
#include <sys/stat.h>

#include <thread>

#include "../host/bam_gpu.hpp"
#include "../host/bam_multi.hpp"
#include "../host/bam_reader.hpp"
#include "../host/report.hpp"

using namespace hpn;

#define BAM_DEF_MASK (4 | 256 | 512 | 1024) /* bam.h:124 */

// The bedGraph text of a finished target, from the device to the file, on the writer's thread: through a context (= a stream)
// of its own, so that the copies run beside the kernels that already ingest the next target.  First a device-to-device copy
// into the fetcher's own buffer (a millisecond per GB) -- after it the library's text buffer is free for the next target's
// hpn_depth_bedgraph_format (taken() tells) --, then pinned slices of 16 MiB to the file, slice k written while k + 1 is copied.
struct TextFetcher {
    hpn_ctx *cx = nullptr;
    void *pin[2] = {nullptr, nullptr}, *d_own = nullptr;
    size_t own_cap = 0;
    static constexpr size_t kSlice = (size_t)16 << 20;    // (pinned memory costs ~0.4 ms per MiB over a process's life)
    bool tried = false;
    std::mutex m;
    std::condition_variable cv;
    bool busy = false;               // a fetch has been handed the library's buffer and has not copied it yet
    std::thread maker;
    // The context and the pinned slices take ~40 ms to make: started on a thread when the first target begins (not earlier: a
    // tool that leaves through exit() -- no index, a file that is no BAM -- while a thread is inside the runtime crashes in the
    // runtime's teardown), joined when the first target's text is there.  false: not available, the caller reads the text in line.
    void start(hpn_ctx *main_ctx)
    {
        if (tried) return;
        tried = true;
        maker = std::thread([this, main_ctx] {
            int device = 0;
            if (hpn_ctx_device(main_ctx, &device) != HPN_OK || hpn_ctx_create(device, &cx) != HPN_OK) {
                cx = nullptr;
                return;
            }
            if (hpn_host_malloc(cx, kSlice, &pin[0]) != HPN_OK || hpn_host_malloc(cx, kSlice, &pin[1]) != HPN_OK) release();
        });
    }
    bool ready(hpn_ctx *main_ctx)
    {
        start(main_ctx);
        if (maker.joinable()) maker.join();
        return cx != nullptr;
    }
    void hand_over()                 // (main thread, before the writer thread is started with the buffer)
    {
        std::lock_guard<std::mutex> lk(m);
        busy = true;
    }
    void taken()                     // (main thread, before the library's buffer is written again)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
    }
    // 0: written; 1: the copy from the device failed; 2: the file could not be written
    int fetch(const uint8_t *d_text, uint64_t n, FILE *out)
    {
        bool ok = true;
        if (n > own_cap) {
            if (d_own) hpn_dev_free(cx, d_own);
            d_own = nullptr, own_cap = 0;
            if (hpn_dev_malloc(cx, (size_t)n + (size_t)(n / 4) + 4096, &d_own) == HPN_OK) own_cap = (size_t)n + (size_t)(n / 4) + 4096;
            else ok = false;
        }
        if (ok && n) ok = hpn_memcpy_d2d(cx, d_own, d_text, (size_t)n) == HPN_OK && hpn_ctx_sync(cx) == HPN_OK;
        {
            std::lock_guard<std::mutex> lk(m);
            busy = false;
        }
        cv.notify_all();
        if (!ok) return 1;
        const uint8_t *src = (const uint8_t *)d_own;
        uint64_t at = 0, prev = 0;
        int k = 0;
        if (fflush(out) != 0) return 2;
        const int fd = fileno(out);
        {   // the target's text has a known size: the file's blocks are asked for at once (one writer thread fills a file at 14 GB/s,
            // at 18 with its blocks there: scripts/micro/close_cost.cpp -- the writer is what the last target waits for).  The
            // file's SIZE is left alone (FALLOC_FL_KEEP_SIZE): a write that fails later leaves a short file, not a full-size one
            // with zeros at its end; nothing is asked for on a descriptor that appends (its position is not where it writes).
            struct stat sb;
            const off_t here = fd >= 0 ? lseek(fd, 0, SEEK_CUR) : -1;
            const int fl = fd >= 0 ? fcntl(fd, F_GETFL) : -1;
            if (here >= 0 && fl >= 0 && !(fl & O_APPEND) && n >= ((uint64_t)64 << 20) && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode))
                (void)fallocate(fd, FALLOC_FL_KEEP_SIZE, here, (off_t)n);   // (fallocate(2): fails at once where the file system cannot; no zero-writing emulation)
        }
        if (n && hpn_memcpy_d2h(cx, pin[0], src, n < kSlice ? n : kSlice) != HPN_OK) return 1;
        while (at < n) {
            const uint64_t len = n - at < kSlice ? n - at : kSlice;
            if (hpn_ctx_sync(cx) != HPN_OK) return 1;                  // slice k is here
            prev = at, at += len;
            if (at < n && hpn_memcpy_d2h(cx, pin[(k + 1) & 1], src + at, n - at < kSlice ? n - at : kSlice) != HPN_OK) return 1;
            const char *p = (const char *)pin[k & 1];
            for (size_t done = 0, want = (size_t)(at - prev); done < want;) {      // (the stream is flushed: straight to the descriptor)
                const ssize_t w = fd >= 0 ? write(fd, p + done, want - done) : -1;
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) return 2;
                done += (size_t)w;
            }
            ++k;
        }
        return 0;
    }
    void release()
    {
        if (maker.joinable() && std::this_thread::get_id() != maker.get_id()) maker.join();
        if (cx) {
            if (pin[0]) hpn_host_free(cx, pin[0]);
            if (pin[1]) hpn_host_free(cx, pin[1]);
            if (d_own) hpn_dev_free(cx, d_own);
            hpn_ctx_destroy(cx);
        }
        cx = nullptr, pin[0] = pin[1] = nullptr, d_own = nullptr, own_cap = 0;
    }
};

static void usage(const char *prog)
{
    fprintf(stderr,
            "\nUsage: %s [-o OUTFILE] [-w WINDOW_SIZE] [-r chr1:1-2000000] [-W] [-s 0] [-h] bamFile1 bamFile2 ..\n"
            "  Converts indexed BAM files to bedGraph and reports the mean depth per window\n"
            "  (MI355X build of HighPerformanceNGS bam2depth).\n\n"
            "   [-o OUTPUT_FILE]  output prefix\n"
            "   [-w WINDOW_SIZE]  window size, default 20000\n"
            "   [-W]              also write wig + chromSize files\n"
            "   [-r], [-s]        accepted, unused\n"
            "   [-h]              this help\n\n",
            prog);
    exit(1);
}

static bool index_exists(const char *bam)
{
    std::string a = std::string(bam) + ".bai", b = bam;
    if (access(a.c_str(), R_OK) == 0) return true;
    if (b.size() > 3 && b.compare(b.size() - 3, 3, "bam") == 0) {  // x.bam -> x.bai (bam_index.c: bam_index_load_local)
        b.replace(b.size() - 3, 3, "bai");
        if (access(b.c_str(), R_OK) == 0) return true;
    }
    return false;
}

int main(int argc, char *argv[])
{
    bind_before_runtime();     // (host/cpus.hpp: next to the device before the runtime starts)
    const char *outfile = "-";
    uint32_t window = 20000;
    int wig = 0;
    stamp("main");
    if (argc < 2) usage(argv[0]);
    int opt;
    while ((opt = getopt(argc, argv, "o:w:r:s:Wh?")) != -1) {
        switch (opt) {
        case 'o': outfile = optarg; break;
        case 'w': window = (uint32_t)atoi(optarg); break;
        case 'W': wig++; break;
        case 'r':  // falls through into -s in the reference
        case 's': break;
        case '?':
        case 'h': usage(argv[0]); break;
        default: fprintf(stderr, "error parameter!\n"); break;
        }
    }
    if (window == 0) {
        fprintf(stderr, "bam2depth: window size must be positive\n");
        return 2;
    }
    char **infiles = argv + optind;
    const int n_in = argc - optind;
    const long long begin = usec();

    hpn_ctx *ctx = nullptr;
    int rc = hpn_ctx_create(getenv("HPN_DEVICE") ? atoi(getenv("HPN_DEVICE")) : 0, &ctx);
    if (rc != HPN_OK) die_hpn(nullptr, rc, "hpn_ctx_create");
    bind_for_device(ctx);
    const bool timing = getenv("HPN_TIMING") != nullptr;
    if (timing) fprintf(stderr, "[hpn] context at %.3f s\n", (double)(usec() - begin) / CLOCKS_PER_SEC);
    stamp("context created");

    char suffix[64];
    // the bedGraph lines are formatted on the device (hpn_depth_bedgraph_format); HPN_BEDGRAPH_HOST=1: from the runs, on the host
    const bool dev_text = !(getenv("HPN_BEDGRAPH_HOST") && getenv("HPN_BEDGRAPH_HOST")[0] == '1');
    for (int i = 0; i < n_in; ++i) {
      const int workers = multi_gpu_workers_for(infiles[i], true);   // > 1: targets are spread over the GPUs (host/bam_multi.hpp)
      bool try_multi = workers > 1 && bam_gpu_enabled();
      // first with the BGZF inflate and the record walk on the GPU; a file that cannot be decoded
      // there (a damaged block, an impossible record, a file that ends inside a record) is done again with the host reader
      for (int pass = bam_gpu_enabled() ? 0 : 1; pass < 2; ++pass) {
        DepthFeeder bam;
        BamHeader hdr;
        if (try_multi) {
            BamReader r;
            if (!r.open(infiles[i], hdr)) err(1, "bam2bed: Fail to open BAM file %s\n", infiles[i]);
        } else if (!bam.open(ctx, infiles[i], hdr, pass == 0)) {
            err(1, "bam2bed: Fail to open BAM file %s\n", infiles[i]);
        }
        if (timing) fprintf(stderr, "[hpn] %s open at %.3f s\n", infiles[i], (double)(usec() - begin) / CLOCKS_PER_SEC);
        stamp("input open");
        std::string nm = infiles[i];
        snprintf(suffix, sizeof suffix, ".%u.bedGraph", i + 1);
        FILE *bedGraph = fcreat_outfile(basename(&nm[0]), suffix);
        snprintf(suffix, sizeof suffix, ".%u.depth", i + 1);
        FILE *depth = fcreat_outfile(outfile, suffix);
        FILE *WIG = nullptr, *chrSize = nullptr;
        if (wig) {
            snprintf(suffix, sizeof suffix, ".%u.wig", i + 1);
            WIG = fcreat_outfile(outfile, suffix);
            snprintf(suffix, sizeof suffix, ".%u.chromSize.txt", i + 1);
            chrSize = fcreat_outfile(outfile, suffix);
        }
        if (!index_exists(infiles[i])) {
            fprintf(stderr, "bam2bed: BAM indexing file is not available.\n");
            leave(1);
        }
        if (try_multi) {   // one worker per GPU, targets largest first, results written here in target order
            try_multi = false;
            const bool done = depth_targets_multi(infiles[i], hdr, BAM_DEF_MASK, window, true, workers, [&](int32_t j, TargetOut &o) {
                const char *name = hdr.target_name[j].c_str();
                const uint32_t tlen = hdr.target_len[j];
                if (dev_text) fwrite(o.text.data(), 1, o.text.size(), bedGraph);
                else print_bedgraph(bedGraph, name, o.runs.data(), o.n_runs);
                print_depth_bins(depth, name, tlen, window, o.win.data());
                if (wig) {
                    print_wig_bins(WIG, name, tlen, window, o.win.data());
                    fprintf(chrSize, "%s\t%d\n", name, (int)tlen);
                }
                fprintf(stderr, "%s at %.3f s\n", name, (double)(usec() - begin) / CLOCKS_PER_SEC);
            }, dev_text);
            if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] GPU ingest on %d workers%s\n", workers, done ? "" : "  (abandoned)");
            if (done) say_workers_once("targets by worker", workers);
            fclose(bedGraph);
            fclose(depth);
            if (wig) {
                fclose(WIG);
                fclose(chrSize);
            }
            if (done) break;
            --pass;        // the single-stream route starts the file over (its outputs are re-created)
            continue;
        }
        // two result buffers: target j is formatted and written by a thread of its own while the GPU
        // already ingests target j + 1 (the writers run one after the other, so the files stay in order)
        std::vector<hpn_run> runs_buf[2] = {std::vector<hpn_run>(1u << 20), std::vector<hpn_run>(1u << 20)};
        std::vector<char> text_buf[2];
        std::vector<uint64_t> win_buf[2];
        std::thread printer;
        // (one for the process, made for the first large input and never taken apart: the tool leaves through _exit, and taking a
        // context apart costs as much as making it -- on a 600 MB BAM the fetcher's 40 + 30 ms were a quarter of the run, so
        // files under 2 GiB read their text in line as before)
        static TextFetcher &copy = *new TextFetcher;     // (never destroyed: no destructor runs on whatever path the tool leaves)
        struct stat fsb;
        const bool fetch_aside = stat(infiles[i], &fsb) == 0 && fsb.st_size >= ((off_t)2 << 30);
        double t_feed = 0, t_finish = 0, t_print = 0, t0;  // HPN_TIMING diagnostics
        bool redo = false;
        for (int32_t j = 0; j < hdr.n_targets() && !redo; ++j) {
            const uint32_t tlen = hdr.target_len[j];
            const char *name = hdr.target_name[j].c_str();
            if ((rc = hpn_depth_begin_w(ctx, j, tlen, BAM_DEF_MASK, (uint32_t)window)) != HPN_OK) die_hpn(ctx, rc, "hpn_depth_begin");   // sorted input: swept while it streams
            if (timing && j == 0) fprintf(stderr, "[hpn] first target begun at %.3f s\n", (double)(usec() - begin) / CLOCKS_PER_SEC);
            if (j == 0) stamp("first target begun");
            if (dev_text && fetch_aside && j == 0) copy.start(ctx);
            t0 = wall_s();
            rc = bam.feed(j);
            t_feed += wall_s() - t0;
            if (rc == 1) {
                redo = true;
                break;
            }
            if (rc != HPN_OK) die_hpn(ctx, rc, "hpn_depth_add");
            t0 = wall_s();
            std::vector<hpn_run> &runs = runs_buf[j & 1];
            std::vector<uint64_t> &win = win_buf[j & 1];
            win.assign((size_t)tlen / window + 1, 0);
            uint64_t n_runs = 0;
            std::vector<char> &text = text_buf[j & 1];
            const uint8_t *d_text = nullptr;
            uint64_t text_bytes = 0;
            bool by_fetcher = false;
            if (dev_text) {
                uint64_t nbytes = 0;
                rc = hpn_depth_finish(ctx, window, nullptr, 0, &n_runs, win.data());
                copy.taken();                                 // (the writer of the target before has the text it was handed)
                if (rc == HPN_OK) rc = hpn_depth_bedgraph_format(ctx, name, &nbytes);
                // (the text stays on the device: the writer thread copies it out through a context of its own, in pinned slices,
                // while this thread already ingests the next target -- 3 GB of text per 10 GB of BAM were read back in line, into
                // pageable memory, 0.10 of the tool's 0.87 s)
                if (rc == HPN_OK && fetch_aside && copy.ready(ctx)) {
                    rc = hpn_depth_bedgraph_dev(ctx, &d_text, &text_bytes);
                    by_fetcher = rc == HPN_OK;
                } else if (rc == HPN_OK) {                    // (no second context: read in line, as before)
                    text.resize(nbytes);
                    rc = hpn_depth_bedgraph_read(ctx, 0, text.data(), nbytes);
                }
            } else {
                rc = hpn_depth_finish(ctx, window, runs.data(), runs.size(), &n_runs, win.data());
                if (rc == HPN_E_CAPACITY) {
                    runs.resize(n_runs);
                    rc = hpn_depth_finish(ctx, window, runs.data(), runs.size(), &n_runs, win.data());
                }
            }
            if (rc != HPN_OK) die_hpn(ctx, rc, name);
            t_finish += wall_s() - t0;
            t0 = wall_s();
            if (printer.joinable()) printer.join();
            if (by_fetcher) copy.hand_over();
            printer = std::thread([=, &runs, &win, &text] {
                if (by_fetcher) {
                    const int fr = copy.fetch(d_text, text_bytes, bedGraph);
                    if (fr) {
                        fprintf(stderr, fr == 2 ? "bam2depth: writing the bedGraph text of %s failed\n" : "bam2depth: copying the bedGraph text of %s from the device failed\n", name);
                        _exit(fr);
                    }
                } else if (dev_text) {
                    fwrite(text.data(), 1, text.size(), bedGraph);
                } else {
                    print_bedgraph(bedGraph, name, runs.data(), n_runs);
                }
                print_depth_bins(depth, name, tlen, window, win.data());
                if (wig) {
                    print_wig_bins(WIG, name, tlen, window, win.data());
                    fprintf(chrSize, "%s\t%d\n", name, (int)tlen);
                }
            });
            t_print += wall_s() - t0;
            fprintf(stderr, "%s at %.3f s\n", name, (double)(usec() - begin) / CLOCKS_PER_SEC);
        }
        stamp("last target handed to the writer");
        t0 = wall_s();
        if (printer.joinable()) printer.join();
        t_print += wall_s() - t0;
        stamp("writer done");
        if (copy.maker.joinable()) copy.maker.join();
        if (getenv("HPN_TIMING"))
            fprintf(stderr, "[hpn] %s ingest + scatter %.3f s  scan+fetch runs %.3f s  waiting for the writer %.3f s%s\n",
                    bam.on_gpu() ? "GPU" : "host", t_feed, t_finish, t_print, redo ? "  (abandoned: not decodable on the GPU)" : "");
        stamp("writer joined, timing line out");
        // (a stream that met ENOSPC / EIO says so here at the latest: a short bedGraph must not leave with exit code 0)
        bool wrote = !ferror(bedGraph) && fclose(bedGraph) == 0;
        stamp("bedGraph closed");
        wrote = (!ferror(depth) && fclose(depth) == 0) && wrote;
        if (wig) {
            wrote = (!ferror(WIG) && fclose(WIG) == 0) && wrote;
            wrote = (!ferror(chrSize) && fclose(chrSize) == 0) && wrote;
        }
        stamp("outputs closed");
        if (!wrote) {
            fprintf(stderr, "bam2depth: writing the outputs of %s failed\n", infiles[i]);
            leave(2);
        }
        if (!redo && i + 1 == n_in) bam.abandon();   // (the last input: its read-ahead is not taken apart, the process ends)
        if (!redo) break;  // else: the outputs are re-created (truncated) by the host pass
      }
      fprintf(stderr, "Converted %s to wig format at %.3f s\n", infiles[i], (double)(usec() - begin) / CLOCKS_PER_SEC);
    }
    stamp("finished");
    quick_exit_ok();
}
