// fastq_count -- drop-in for the reference tool of the same name (fastq_count.c), with the
// per-record tally (count_read's loop, :112-119) running on MI355X through libhpngs.
//
//   fastq_count [-o OUT] [-t N] [-H] [-L] [-h] file1.fq[.gz] file2.fq[.gz] ...
//
// Same flags, same report bytes: one row per file, printed in completion order under
// a lock, one host thread per file in batches of -t (fastq_count.c:213-230).
// Differences, all outside the report: a read longer than 511 or a quality byte
// >= 128 is an error here (the reference overruns its arrays); HPN_DEVICE picks the GPU.
#include <getopt.h>
#include <pthread.h>

#include <atomic>
#include <mutex>
#include <thread>

#include "../host/report.hpp"
#include "../host/tally_stream.hpp"

using namespace hpn;

static struct {
    char **infiles;
    const char *outfile;
    int numInfiles, thread, header, LengthDetail;
} g;
static std::mutex g_lock;
static std::atomic<int> g_next{0};
static int g_ndev = 1, g_dev0 = 0;

static void usage(const char *prog)
{
    fprintf(stderr,
            "\nUsage: %s file1.fq file2.fq ... [-o outfile] [-t thread] [-H] [-L] [-h]\n"
            "  Counts reads, bases, mean/min/max length and %%Q20 / %%Q30 of plain or gzip FASTQ files\n"
            "  (MI355X build of HighPerformanceNGS fastq_count).\n\n"
            "   [-o OUTPUT] output file, default stdout\n"
            "   [-H]        print the header line\n"
            "   [-L]        also print the read length histogram\n"
            "   [-t N]      files processed concurrently, default min(#files, #cpus)\n"
            "   [-h]        this help\n\n",
            prog);
    exit(1);
}

static void count_file(hpn_ctx *ctx, WorkerLanes &lanes, const char *infile, FILE *out)
{
    int rc;
    hpn_tally acc;
    memset(&acc, 0, sizeof acc);
    bool too_long = false;
    rc = tally_file(ctx, infile, &acc, &too_long, lanes.for_file(infile));  // count_read's loop (:112-119), tally on the GPU(s)
    if (too_long) {
        fprintf(stderr, "%s: read longer than 511 bases (outside fastq_count's SeqLen[512])\n", infile);
        leave(2);     // (not exit(): other workers are inside the runtime; found by scripts/soak_fastq_tools.py as a SIGSEGV of the kthread tool)
    }
    if (rc != HPN_OK) die_hpn(ctx, rc, infile);
    const CountSummary s = summarise(acc);
    {
        std::lock_guard<std::mutex> lk(g_lock);  // pthread_mutex_lock (fastq_count.c:126)
        print_count_row(out, infile, acc, s);
        if (g.LengthDetail) print_len_detail(out, acc.seqlen, s.min_len, s.max_len);
    }
}

// One worker = one host thread + one GPU context, like one kt_for worker of the reference;
// it takes the next unprocessed file (fastq_count.c:213-230 runs them in waves of T).
// With fewer files in flight than devices, a worker spreads each of its inputs over its share of the
// devices by record block (host/text_shard.hpp; the reference stops at one file per thread).
static void worker(int slot, int workers, FILE *out)
{
    hpn_ctx *ctx = nullptr;
    const double t0 = wall_s();
    const int rel = WorkerLanes::device_of(g_ndev, workers, slot);
    const int rc = hpn_ctx_create(g_dev0 + rel, &ctx);
    if (rc != HPN_OK) die_hpn(nullptr, rc, "hpn_ctx_create");
    bind_for_device(ctx);
    if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] worker %d: context %.3f s\n", slot, wall_s() - t0);
    stamp("context created");
    WorkerLanes lanes(ctx, g_dev0, rel, g_ndev, WorkerLanes::cap(g_ndev, workers), workers == 1, true);
    for (int i; (i = g_next.fetch_add(1)) < g.numInfiles;) count_file(ctx, lanes, g.infiles[i], out);
}

int main(int argc, char *argv[])
{
    bind_before_runtime();     // (host/cpus.hpp: next to the device before the runtime starts)
    stamp("main");
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
    if (g.thread < 1) g.thread = 1;
    if (const char *d = getenv("HPN_DEVICE")) g_dev0 = atoi(d), g_ndev = 1;
    else if (hpn_device_count(&g_ndev) != HPN_OK || g_ndev < 1) die_hpn(nullptr, HPN_E_NODEVICE, "fastq_count");

    stamp("devices counted");
    const long long begin = usec();
    FILE *out = fopen_output_stream(g.outfile);
    if (g.header) print_count_header(out);
    {
        const int workers = text_workers(g.infiles, g.numInfiles, g.thread, g_ndev);
        text_workers_in_flight() = workers;
        std::vector<std::thread> th;
        for (int j = 0; j < workers && j < g.numInfiles; ++j) th.emplace_back(worker, j, workers < g.numInfiles ? workers : g.numInfiles, out);
        for (auto &t : th) t.join();
    }
    fprintf(stderr, "Finished at %.3f s\n", (double)(usec() - begin) / CLOCKS_PER_SEC);
    stamp("finished");
    fclose(out);
    quick_exit_ok();
}
