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
#include <libgen.h>

#include <thread>

#include "../host/bam_gpu.hpp"
#include "../host/bam_multi.hpp"
#include "../host/bam_reader.hpp"
#include "../host/report.hpp"

using namespace hpn;

#define BAM_DEF_MASK (4 | 256 | 512 | 1024) /* bam.h:124 */

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
    const char *outfile = "-";
    uint32_t window = 20000;
    int wig = 0;
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
    const bool timing = getenv("HPN_TIMING") != nullptr;
    if (timing) fprintf(stderr, "[hpn] context at %.3f s\n", (double)(usec() - begin) / CLOCKS_PER_SEC);

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
            exit(1);
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
        double t_feed = 0, t_finish = 0, t_print = 0, t0;  // HPN_TIMING diagnostics
        bool redo = false;
        for (int32_t j = 0; j < hdr.n_targets() && !redo; ++j) {
            const uint32_t tlen = hdr.target_len[j];
            const char *name = hdr.target_name[j].c_str();
            if ((rc = hpn_depth_begin_w(ctx, j, tlen, BAM_DEF_MASK, (uint32_t)window)) != HPN_OK) die_hpn(ctx, rc, "hpn_depth_begin");   // sorted input: swept while it streams
            if (timing && j == 0) fprintf(stderr, "[hpn] first target begun at %.3f s\n", (double)(usec() - begin) / CLOCKS_PER_SEC);
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
            if (dev_text) {
                uint64_t nbytes = 0;
                rc = hpn_depth_finish(ctx, window, nullptr, 0, &n_runs, win.data());
                if (rc == HPN_OK) rc = hpn_depth_bedgraph_format(ctx, name, &nbytes);
                if (rc == HPN_OK) {
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
            printer = std::thread([=, &runs, &win, &text] {
                if (dev_text) fwrite(text.data(), 1, text.size(), bedGraph);
                else print_bedgraph(bedGraph, name, runs.data(), n_runs);
                print_depth_bins(depth, name, tlen, window, win.data());
                if (wig) {
                    print_wig_bins(WIG, name, tlen, window, win.data());
                    fprintf(chrSize, "%s\t%d\n", name, (int)tlen);
                }
            });
            t_print += wall_s() - t0;
            fprintf(stderr, "%s at %.3f s\n", name, (double)(usec() - begin) / CLOCKS_PER_SEC);
        }
        t0 = wall_s();
        if (printer.joinable()) printer.join();
        t_print += wall_s() - t0;
        if (getenv("HPN_TIMING"))
            fprintf(stderr, "[hpn] %s ingest + scatter %.3f s  scan+fetch runs %.3f s  waiting for the writer %.3f s%s\n",
                    bam.on_gpu() ? "GPU" : "host", t_feed, t_finish, t_print, redo ? "  (abandoned: not decodable on the GPU)" : "");
        fclose(bedGraph);
        fclose(depth);
        if (wig) {
            fclose(WIG);
            fclose(chrSize);
        }
        if (!redo) break;  // else: the outputs are re-created (truncated) by the host pass
      }
      fprintf(stderr, "Converted %s to wig format at %.3f s\n", infiles[i], (double)(usec() - begin) / CLOCKS_PER_SEC);
    }
    quick_exit_ok();
}
