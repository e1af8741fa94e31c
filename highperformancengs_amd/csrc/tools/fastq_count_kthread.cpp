// fastq_count_kthread -- drop-in for the reference tool of the same name
// (fastq_count_kthread.c), with count_read's loop (:126-135) on MI355X through libhpngs.
//
//   fastq_count_kthread [-o OUT] [-t N] [-H] [-L] [-h] file1.fq[.gz] ...
//
// Per input i a report ./<basename>.<i>.tsv, then the merged row of reduceStats
// (:180-210) on -o.  -L adds the length histogram and the 128 x maxLen quality
// matrix (printQ :52-64), which is when the full Quality[128][512] histogram kernel
// runs; without -L only sum / Q20 / Q30 are needed and the flat scan kernel is used.
// Files are handed to N workers like kt_for does (klib/kthread.c:48-60); the output
// does not depend on the schedule.
#include <getopt.h>
#include <libgen.h>

#include <atomic>
#include <thread>

#include "../host/report.hpp"
#include "../host/tally_stream.hpp"

using namespace hpn;

static struct {
    char **infiles;
    const char *outfile;
    int numInfiles, thread, header, LengthDetail;
} g;
static int g_ndev = 1, g_dev0 = 0;

struct FileAcc {
    hpn_tally t;
    std::vector<uint64_t> qual;  // Quality[128][512] when -L
    CountSummary s;
    FILE *out = nullptr;
};

static void usage(const char *prog)
{
    fprintf(stderr,
            "\nUsage: %s file1.fq file2.fq ... [-o outfile] [-t thread] [-H] [-L] [-h]\n"
            "  Per-file and merged read / base / length / Q20 / Q30 statistics of plain or gzip FASTQ files\n"
            "  (MI355X build of HighPerformanceNGS fastq_count_kthread).\n\n"
            "   [-o OUTPUT] merged report, default stdout; per-file reports go to ./<basename>.<i>.tsv\n"
            "   [-H]        print header lines\n"
            "   [-L]        also print length histograms and the quality-by-cycle matrix\n"
            "   [-t N]      worker threads, default min(#files, #cpus)\n"
            "   [-h]        this help\n\n",
            prog);
    exit(1);
}

static void count_file(hpn_ctx *ctx, WorkerLanes &lanes, FileAcc &fa, const char *infile)
{
    int rc;
    bool too_long = false;
    rc = tally_file(ctx, infile, &fa.t, &too_long, lanes.for_file(infile));  // count_read's loop (:126-135), tally on the GPU(s)
    if (too_long) {
        fprintf(stderr, "%s: read longer than 511 bases (outside SeqLen[512])\n", infile);
        leave(2);     // (not exit(): other workers are inside the runtime, whose exit handlers crash under them -- a SIGSEGV instead of code 2; scripts/soak_fastq_tools.py)
    }
    if (rc != HPN_OK) die_hpn(ctx, rc, infile);
    fa.s = summarise(fa.t);
    if (g.header) print_count_header(fa.out);            // :140
    print_kthread_file_row(fa.out, infile, fa.t, fa.s);  // :141
    if (g.LengthDetail) {
        print_len_detail(fa.out, fa.t.seqlen, fa.s.min_len, fa.s.max_len);
        print_quality_matrix(fa.out, fa.qual.data(), fa.s.max_len);
    }
}

int main(int argc, char *argv[])
{
    bind_before_runtime();     // (host/cpus.hpp: next to the device before the runtime starts)
    g.outfile = "-";
    g.thread = (int)sysconf(_SC_NPROCESSORS_ONLN);
    int opt;
    while ((opt = getopt(argc, argv, "o:t:HLh?")) != -1) {
        switch (opt) {
        case 'o': g.outfile = optarg; break;
        case 't': g.thread = atoi(optarg); break;
        case 'H': g.header++; break;
        case 'L': g.LengthDetail++; break;
        case '?':
        case 'h': usage(argv[0]); break;
        default: fprintf(stderr, "error parameter!\n"); break;
        }
    }
    g.infiles = argv + optind;
    g.numInfiles = argc - optind;
    if (g.numInfiles < g.thread) g.thread = g.numInfiles;
    if (const char *d = getenv("HPN_DEVICE")) g_dev0 = atoi(d), g_ndev = 1;
    else if (hpn_device_count(&g_ndev) != HPN_OK || g_ndev < 1) die_hpn(nullptr, HPN_E_NODEVICE, "fastq_count_kthread");

    const long long begin = usec();
    if (g.numInfiles) {
        std::vector<FileAcc> acc((size_t)g.numInfiles);
        char suffix[32];
        for (int i = 0; i < g.numInfiles; ++i) {
            memset(&acc[i].t, 0, sizeof(hpn_tally));
            if (g.LengthDetail) {
                acc[i].qual.assign((size_t)HPN_QUAL_ROWS * HPN_LEN_BINS, 0);
                acc[i].t.qual_hist = acc[i].qual.data();
            }
            snprintf(suffix, sizeof suffix, ".%d.tsv", i);
            std::string name = g.infiles[i];  // basename() may modify its argument
            acc[i].out = fcreat_outfile(basename(&name[0]), suffix);  // :265-266
        }
        std::atomic<long> next{0};
        std::vector<std::thread> th;
        const int workers = text_workers(g.infiles, g.numInfiles, g.thread < 1 ? 1 : g.thread, g_ndev);
        text_workers_in_flight() = workers;
        for (int t = 0; t < workers; ++t)
            th.emplace_back([&, t] {  // one kt_for worker (klib/kthread.c:34-60) = one thread + one GPU context
                // ... and, with fewer files in flight than devices, its share of the devices to spread each input over
                // by record block (host/text_shard.hpp); the lanes' count vectors are summed where reduceStats sums files
                hpn_ctx *ctx = nullptr;
                const int nw = workers < g.numInfiles ? workers : g.numInfiles;
                const int rel = WorkerLanes::device_of(g_ndev, nw, t);
                const int rc = hpn_ctx_create(g_dev0 + rel, &ctx);
                if (rc != HPN_OK) die_hpn(nullptr, rc, "hpn_ctx_create");
                bind_for_device(ctx);
                WorkerLanes lanes(ctx, g_dev0, rel, g_ndev, WorkerLanes::cap(g_ndev, nw), nw == 1, true);
                for (long i; (i = next.fetch_add(1)) < g.numInfiles;) count_file(ctx, lanes, acc[(size_t)i], g.infiles[i]);
            });
        for (auto &t : th) t.join();

        // reduceStats (:180-210)
        FILE *Out = fopen_output_stream(g.outfile);
        uint32_t sumRC = 0, TotalMinLen = 10000, TotalMaxLen = 0;
        double sumBC = 0;
        hpn_tally m;
        memset(&m, 0, sizeof m);
        std::vector<uint64_t> mq((size_t)HPN_QUAL_ROWS * HPN_LEN_BINS, 0);
        for (int i = 0; i < g.numInfiles; ++i) {
            sumRC += (uint32_t)acc[i].s.reads;  // uint32 accumulation (:186)
            sumBC += acc[i].s.bases;
            if (acc[i].s.min_len < TotalMinLen) TotalMinLen = acc[i].s.min_len;
            if (acc[i].s.max_len > TotalMaxLen) TotalMaxLen = acc[i].s.max_len;
            for (int l = 0; l < HPN_LEN_BINS; ++l) m.seqlen[l] += acc[i].t.seqlen[l];
            m.total += acc[i].t.total, m.q20 += acc[i].t.q20, m.q30 += acc[i].t.q30;
            if (g.LengthDetail)
                for (size_t k = 0; k < mq.size(); ++k) mq[k] += acc[i].qual[k];
        }
        if (g.header) fprintf(Out, "#ReadCount\tBaseCount\tMeanLen\tMinLen\tMaxLen\tQ20(%%)\tQ30(%%)\n");
        fprintf(Out, "%u\t%.0f\t%.0f\t%u\t%u\t%.3f\t%.3f\n", sumRC, sumBC, sumBC / sumRC, TotalMinLen, TotalMaxLen,
                1.0 * m.q20 / m.total * 100, 1.0 * m.q30 / m.total * 100);
        if (g.LengthDetail) {
            print_len_detail(Out, m.seqlen, TotalMinLen, TotalMaxLen);
            print_quality_matrix(Out, mq.data(), TotalMaxLen);
        }
        fclose(Out);
        for (int i = 0; i < g.numInfiles; ++i) fclose(acc[i].out);
    }
    fprintf(stderr, "Finished at %.3f s\n", (double)(usec() - begin) / CLOCKS_PER_SEC);
    quick_exit_ok();
}
