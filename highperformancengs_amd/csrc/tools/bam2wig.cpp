// bam2wig -- drop-in for the reference tool of the same name (bam2wig.c): bam2depth's
// breakpoint sweep with the filter "unmapped only" (bam2wig.c:88) and only the wig and
// chromSize outputs.  Record loop and sweep run on MI355X through libhpngs (the same
// kernels as bam2depth, flag mask 0x4); the window bins follow bam2wig's own overlap()
// arithmetic (inclusive window ends, report.hpp::wig_bins_from_runs).
//
//   bam2wig [-o OUT] [-w W] [-r REGION] [-s INT] [-h] a.bam b.bam ...
//
// Writes OUT.<n>.wig and OUT.<n>.chromSize.txt (bam2wig.c:291-294); needs <bam>.bai.
#include <err.h>
#include <getopt.h>

#include "../host/bam_gpu.hpp"
#include "../host/bam_multi.hpp"
#include "../host/bam_reader.hpp"
#include "../host/report.hpp"

using namespace hpn;

#define BAM_FUNMAP 4

static void usage(const char *prog)
{
    fprintf(stderr,
            "\nUsage: %s [-o OUTFILE] [-w WINDOW_SIZE] [-r chr1:1-2000000] [-s 0] [-h] bamFile1 bamFile2 ..\n"
            "  Converts indexed BAM files to wig (mean depth per window)\n"
            "  (MI355X build of HighPerformanceNGS bam2wig).\n\n"
            "   [-o OUTPUT_FILE]  output prefix\n"
            "   [-w WINDOW_SIZE]  window size, default 20000\n"
            "   [-r], [-s]        accepted, unused\n"
            "   [-h]              this help\n\n",
            prog);
    exit(1);
}

static bool index_exists(const char *bam)
{
    std::string a = std::string(bam) + ".bai", b = bam;
    if (access(a.c_str(), R_OK) == 0) return true;
    if (b.size() > 3 && b.compare(b.size() - 3, 3, "bam") == 0) {
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
    if (argc < 2) usage(argv[0]);
    int opt;
    while ((opt = getopt(argc, argv, "o:w:r:s:h?")) != -1) {
        switch (opt) {
        case 'o': outfile = optarg; break;
        case 'w': window = (uint32_t)atoi(optarg); break;
        case 'r':
        case 's': break;
        case '?':
        case 'h': usage(argv[0]); break;
        default: fprintf(stderr, "error parameter!\n"); break;
        }
    }
    if (window == 0) {
        fprintf(stderr, "bam2wig: window size must be positive\n");
        return 2;
    }
    char **infiles = argv + optind;
    const int n_in = argc - optind;
    const long long begin = usec();
    hpn_ctx *ctx = nullptr;
    int rc = hpn_ctx_create(getenv("HPN_DEVICE") ? atoi(getenv("HPN_DEVICE")) : 0, &ctx);
    if (rc != HPN_OK) die_hpn(nullptr, rc, "hpn_ctx_create");
    bind_for_device(ctx);

    char suffix[64];
    for (int i = 0; i < n_in; ++i) {
      const int workers = multi_gpu_workers_for(infiles[i], true);   // > 1: targets are spread over the GPUs (host/bam_multi.hpp)
      bool try_multi = workers > 1 && bam_gpu_enabled();
      for (int pass = bam_gpu_enabled() ? 0 : 1; pass < 2; ++pass) {  // GPU ingest first, host reader if the file needs it
        DepthFeeder bam;
        BamHeader hdr;
        if (try_multi) {
            BamReader r;
            if (!r.open(infiles[i], hdr)) err(1, "bam2bed: Fail to open BAM file %s\n", infiles[i]);
        } else if (!bam.open(ctx, infiles[i], hdr, pass == 0)) {
            err(1, "bam2bed: Fail to open BAM file %s\n", infiles[i]);
        }
        snprintf(suffix, sizeof suffix, ".%u.wig", i + 1);
        FILE *wig = fcreat_outfile(outfile, suffix);
        snprintf(suffix, sizeof suffix, ".%u.chromSize.txt", i + 1);
        FILE *chrSize = fcreat_outfile(outfile, suffix);
        if (!index_exists(infiles[i])) {
            fprintf(stderr, "bam2bed: BAM indexing file is not available.\n");
            leave(1);
        }
        std::vector<hpn_run> runs(1u << 20);
        std::vector<double> bins;
        bool redo = false;
        if (try_multi) {   // one worker per GPU, targets largest first, results written here in target order
            try_multi = false;
            const bool done = depth_targets_multi(infiles[i], hdr, BAM_FUNMAP, window, false, workers, [&](int32_t j, TargetOut &o) {
                const char *name = hdr.target_name[j].c_str();
                const uint32_t tlen = hdr.target_len[j];
                bins.assign((size_t)tlen / window + 2, 0.0);
                wig_bins_from_runs(o.runs.data(), o.n_runs, tlen, window, bins.data());
                print_wig_bins_d(wig, name, tlen, window, bins.data());
                fprintf(chrSize, "%s\t%d\n", name, (int)tlen);
                fprintf(stderr, "%s at %.3f s\n", name, (double)(usec() - begin) / CLOCKS_PER_SEC);
            });
            if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] GPU ingest on %d workers%s\n", workers, done ? "" : "  (abandoned)");
            if (done) say_workers_once("targets by worker", workers);
            fclose(wig);
            fclose(chrSize);
            if (done) break;
            --pass;
            continue;
        }
        for (int32_t j = 0; j < hdr.n_targets(); ++j) {
            const uint32_t tlen = hdr.target_len[j];
            const char *name = hdr.target_name[j].c_str();
            if ((rc = hpn_depth_begin(ctx, j, tlen, BAM_FUNMAP)) != HPN_OK) die_hpn(ctx, rc, "hpn_depth_begin");
            rc = bam.feed(j);
            if (rc == 1) {
                redo = true;
                break;
            }
            if (rc != HPN_OK) die_hpn(ctx, rc, "hpn_depth_add");
            uint64_t n_runs = 0;
            rc = hpn_depth_finish(ctx, window, runs.data(), runs.size(), &n_runs, nullptr);
            if (rc == HPN_E_CAPACITY) {
                runs.resize(n_runs);
                rc = hpn_depth_finish(ctx, window, runs.data(), runs.size(), &n_runs, nullptr);
            }
            if (rc != HPN_OK) die_hpn(ctx, rc, name);
            bins.assign((size_t)tlen / window + 2, 0.0);
            wig_bins_from_runs(runs.data(), n_runs, tlen, window, bins.data());
            print_wig_bins_d(wig, name, tlen, window, bins.data());
            fprintf(chrSize, "%s\t%d\n", name, (int)tlen);
            fprintf(stderr, "%s at %.3f s\n", name, (double)(usec() - begin) / CLOCKS_PER_SEC);
        }
        fclose(wig);
        fclose(chrSize);
        if (!redo) break;
      }
      fprintf(stderr, "Converted %s to wig format at %.3f s\n", infiles[i], (double)(usec() - begin) / CLOCKS_PER_SEC);
    }
    quick_exit_ok();
}
