// bam_sliding_count -- drop-in for the counting path of the reference tool of the same
// name (bam_sliding_count.c): per fixed window (keyed by read start) read count, GC bases
// and read length, the record loop running on MI355X through libhpngs.
//
//   bam_sliding_count [-o OUT] [-w W] [-r REGION] [-s INT] [-h] a.bam b.bam ...
//
// Writes OUT.txt (default out.txt) for the FIRST input only, like the reference
// (output_count_GC(databuf,...) :416).  The PNG of draw_hits (:274-329, needs libgd)
// is not produced: that is plotting, not part of the scan path.
#include <err.h>
#include <getopt.h>

#include <cctype>

#include "../host/bam_gpu.hpp"
#include "../host/bam_multi.hpp"
#include "../host/bam_reader.hpp"
#include "../host/report.hpp"

using namespace hpn;

static void usage(const char *prog)
{
    fprintf(stderr,
            "\nUsage: %s [-o OUTFILE] [-w WINDOW_SIZE] [-r chr1:1-2000000] [-s 0] [-h] bamFile1 bamFile2 ..\n"
            "  Read count, GC content and read length per window of BAM files\n"
            "  (MI355X build of HighPerformanceNGS bam_sliding_count; no PNG plot).\n\n"
            "   [-o OUTPUT_FILE]  output prefix, default \"out\" (writes OUTPUT_FILE.txt)\n"
            "   [-w WINDOW_SIZE]  window size, default 20000\n"
            "   [-r REGION]       only reads overlapping chr[:beg-end]\n"
            "   [-s]              accepted, unused\n"
            "   [-h]              this help\n\n",
            prog);
    exit(1);
}

// bam_parse_region (samtools-0.1.19 bam_aux.c:107): name[:beg[-end]], commas ignored,
// beg 1-based -> 0-based, end defaults to 1<<29; a malformed interval makes the whole
// string the name.
static bool parse_region(const BamHeader &h, const char *str, int *ref, int *beg, int *end)
{
    std::string s;
    for (const char *p = str; *p; ++p)
        if (!isspace((unsigned char)*p)) s += *p;
    size_t name_end = s.rfind(':');
    auto find = [&](const std::string &n) {
        for (int32_t t = 0; t < h.n_targets(); ++t)
            if (h.target_name[t] == n) return t;
        return -1;
    };
    *beg = 0, *end = 1 << 29;
    if (name_end != std::string::npos) {
        int hyphens = 0;
        bool ok = true;
        for (size_t i = name_end + 1; i < s.size(); ++i) {
            if (s[i] == '-') ++hyphens;
            else if (!isdigit((unsigned char)s[i]) && s[i] != ',') ok = false;
        }
        if (ok && hyphens <= 1 && (*ref = find(s.substr(0, name_end))) >= 0) {
            std::string iv;
            for (size_t i = name_end + 1; i < s.size(); ++i)
                if (s[i] != ',') iv += s[i];
            *beg = atoi(iv.c_str());
            size_t dash = iv.find('-');
            *end = dash != std::string::npos ? atoi(iv.c_str() + dash + 1) : 1 << 29;
            if (*beg > 0) --*beg;
            return *beg <= *end;
        }
    }
    *ref = find(s);
    return *ref >= 0;
}

int main(int argc, char *argv[])
{
    bind_before_runtime();     // (host/cpus.hpp: next to the device before the runtime starts)
    const char *outfile = "out", *region = "-";
    int window = 20000;
    if (argc < 2) usage(argv[0]);
    int opt;
    while ((opt = getopt(argc, argv, "o:w:r:s:h?")) != -1) {
        switch (opt) {
        case 'o': outfile = optarg; break;
        case 'w': window = atoi(optarg); break;
        case 'r': region = optarg; break;  // (falls through into -s in the reference; -s is unused)
        case 's': break;
        case '?':
        case 'h': usage(argv[0]); break;
        default: fprintf(stderr, "error parameter!\n"); break;
        }
    }
    if (window <= 0) {
        fprintf(stderr, "bam_sliding_count: window size must be positive\n");
        return 2;
    }
    char **infiles = argv + optind;
    const int n_in = argc - optind;
    const long long begin = usec();
    hpn_ctx *ctx = nullptr;
    stamp("main (options read)");
    int rc = hpn_ctx_create(getenv("HPN_DEVICE") ? atoi(getenv("HPN_DEVICE")) : 0, &ctx);
    if (rc != HPN_OK) die_hpn(nullptr, rc, "hpn_ctx_create");
    bind_for_device(ctx);
    stamp("context created");

    // results of the first input: the only one the report uses
    BamHeader hdr0;
    std::vector<uint64_t> off0, gc0;
    std::vector<uint32_t> bins0, len0;
    std::vector<uint8_t> touched0;
    uint64_t n_count0 = 0;

    for (int i = 0; i < n_in; ++i) {
        BamReader bam;
        BamHeader hdr;
        if (!bam.open(infiles[i], hdr)) err(1, "bam2bed: Fail to open BAM file %s\n", infiles[i]);
        stamp("host reader open (header)");
        std::vector<uint64_t> off((size_t)hdr.n_targets() + 1, 0);
        for (int32_t t = 0; t < hdr.n_targets(); ++t) off[t + 1] = off[t] + (uint64_t)(hdr.target_len[t] / (uint32_t)window + 1);
        if (hdr.n_targets() == 0) off.push_back(0);
        if ((rc = hpn_window_begin(ctx, hdr.n_targets() > 0 ? hdr.n_targets() : 1, off.data(), (uint32_t)window)) != HPN_OK)
            die_hpn(ctx, rc, "hpn_window_begin");
        int ref = -1, beg = 0, end = 1 << 29;
        const bool whole = strncmp(region, "-", 6) == 0;  // :388
        if (!whole) {
            std::string idx = std::string(infiles[i]) + ".bai";
            if (access(idx.c_str(), R_OK) != 0) {
                fprintf(stderr, "bam2bed: BAM indexing file is not available.\n");
                return 1;
            }
            if (!parse_region(hdr, region, &ref, &beg, &end) || ref < 0) {
                fprintf(stderr, "bam2bed: Invalid region %s\n", region);
                return 1;
            }
            fprintf(stdout, "%s\t%d\t%d\n", hdr.target_name[ref].c_str(), beg, end);
        }
        // Whole-file mode: BGZF inflate and record walk on the GPU when the file allows it (every block
        // starts at a record boundary, as samtools writes them); else, and for -r, the host reader.
        bool on_gpu = false;
        const int workers = multi_gpu_workers_for(infiles[i]);
        std::vector<uint32_t> m_bins, m_len;          // several GPUs: the workers' per-window vectors, summed here
        std::vector<uint64_t> m_gc;
        std::vector<uint8_t> m_touched;
        uint64_t m_count = 0;
        bool multi_done = false;
        if (whole && bam_gpu_enabled() && workers > 1) {
            // record batches go to the GPUs in turn; bins / GC / length are sums, so each GPU keeps private vectors and the
            // host adds them (3 x ~155 K entries for hg38 at W = 20000) before the float32 replay of calc_winGC
            const int32_t nt = hdr.n_targets() > 0 ? hdr.n_targets() : 1;
            const size_t total = off.back() ? off.back() : 1;
            m_bins.assign(total, 0), m_len.assign(total, 0), m_gc.assign(total, 0), m_touched.assign((size_t)nt, 0);
            std::mutex sum_m;
            multi_done = BgzfFanout::run(
                infiles[i], workers,
                [&](int, hpn_ctx *c) { return hpn_window_begin(c, nt, off.data(), (uint32_t)window) == HPN_OK; },
                [&](int, hpn_ctx *c, const uint8_t *d_raw, const hpn_raw_info &) { return hpn_window_add_raw_dev(c, d_raw) == HPN_OK; },
                [&](int w, hpn_ctx *c) {
                    std::vector<uint32_t> b(total), l(total);
                    std::vector<uint64_t> g(total);
                    std::vector<uint8_t> t((size_t)nt);
                    uint64_t n = 0;
                    const int r = hpn_window_finish(c, b.data(), g.data(), l.data(), t.data(), &n);
                    if (r != HPN_OK) {
                        fprintf(stderr, "[hpn] worker %d: %s\n", w, hpn_ctx_last_error(c));
                        return false;
                    }
                    std::lock_guard<std::mutex> lk(sum_m);
                    for (size_t k = 0; k < total; ++k) m_bins[k] += b[k], m_len[k] += l[k], m_gc[k] += g[k];   // unsigned: wraps like the reference's
                    for (size_t k = 0; k < (size_t)nt; ++k) m_touched[k] |= t[k];
                    m_count += n;
                    return true;
                });
            if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] GPU ingest on %d workers%s\n", workers, multi_done ? "" : "  (abandoned)");
            if (multi_done) say_workers_once("record batches by worker", workers);
        }
        if (!multi_done && whole && bam_gpu_enabled()) {
            // (on the heap: after the tool's last input the stream is not taken apart -- pinned chunks, the upload context and its
            // buffers are ~0.1 s of unpinning and queue destruction in front of an _exit that hands all of it back anyway)
            BgzfGpuStream *gsp = new BgzfGpuStream;
            struct Drop {
                BgzfGpuStream *p;
                bool keep;
                ~Drop() { if (!keep) delete p; }
            } drop{gsp, i + 1 == n_in && !getenv("HPN_FULL_EXIT")};   // (a full exit runs handlers: no thread may be left in the runtime)
            BgzfGpuStream &gs = *gsp;
            BamHeader h2;
            if (gs.open(ctx, infiles[i], h2)) {
                gs.start();
                stamp("GPU stream open, reading ahead");
                hpn_raw_info info;
                int r;
                while ((r = gs.next(&info)) == 1)
                    if (info.n_records && (rc = hpn_window_add_raw_dev(ctx, gs.d_raw())) != HPN_OK) die_hpn(ctx, rc, "hpn_window_add");
                on_gpu = r == 0;
                if (!on_gpu) {  // start over on the host: forget what was added
                    if ((rc = hpn_window_begin(ctx, hdr.n_targets() > 0 ? hdr.n_targets() : 1, off.data(), (uint32_t)window)) != HPN_OK)
                        die_hpn(ctx, rc, "hpn_window_begin");
                }
            }
        }
        if (getenv("HPN_TIMING") && !multi_done) fprintf(stderr, "[hpn] %s ingest\n", on_gpu ? "GPU" : "host");
        stamp("ingest done");
        BamBatch batch, one;
        bool more = !on_gpu && !multi_done;
        // -r: bam_fetch reads from where the index says the region's first record can be, up to the first record that
        // starts at or behind the region's end (a coordinate-sorted file; reading every record from the top of the file,
        // as round 1 did, is O(file) for a region the reference answers in milliseconds)
        BamReader at;
        BamReader *rd = &bam;
        if (!whole) {
            uint64_t vo = 0;
            const int have = bai_region_start(infiles[i], hdr.n_targets(), ref, (uint32_t)beg, &vo);
            if (have == 0) more = false;   // the target holds no record
            else if (have < 0) rd = &bam;  // index present (checked above) but not usable: every record from the top, filtered below
            else if (!at.open_at(infiles[i], vo)) err(1, "bam2bed: Fail to open BAM file %s\n", infiles[i]);
            else rd = &at;
            if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] region: reading from virtual offset %llu\n", (unsigned long long)vo);
        }
        while (more) {
            batch.clear();
            while (batch.n() < (2u << 20)) {
                if (whole) {
                    if (!(more = bam.next(batch, true))) break;
                } else {  // bam_fetch(ref, beg, end): records of `ref` overlapping [beg, end) (bam_index.c:571,682)
                    one.clear();
                    if (!(more = rd->next(one, true))) break;
                    if (one.tid[0] != ref || (uint32_t)one.pos[0] >= (uint32_t)end) {   // past the region: nothing further overlaps it
                        if (one.tid[0] > ref || one.tid[0] < 0 || (one.tid[0] == ref && (uint32_t)one.pos[0] >= (uint32_t)end)) more = false;
                        if (!more) break;
                        continue;
                    }
                    uint32_t rend = (uint32_t)one.pos[0] + 1;
                    if (!one.cigar.empty()) {
                        rend = (uint32_t)one.pos[0];
                        for (uint32_t w : one.cigar) {
                            const uint32_t op = w & 0xf;
                            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rend += w >> 4;
                        }
                    }
                    if (!(rend > (uint32_t)beg && (uint32_t)one.pos[0] < (uint32_t)end)) continue;
                    batch.tid.push_back(one.tid[0]), batch.pos.push_back(one.pos[0]), batch.flag.push_back(one.flag[0]);
                    batch.l_qseq.push_back(one.l_qseq[0]);
                    batch.cigar.insert(batch.cigar.end(), one.cigar.begin(), one.cigar.end());
                    batch.cigar_off.push_back((uint32_t)batch.cigar.size());
                    batch.seq4.insert(batch.seq4.end(), one.seq4.begin(), one.seq4.end());
                    batch.seq_off.push_back(batch.seq4.size());
                }
            }
            if (batch.n()) {
                hpn_bam_batch v = batch.view();
                if ((rc = hpn_window_add(ctx, &v)) != HPN_OK) die_hpn(ctx, rc, "hpn_window_add");
            }
        }
        std::vector<uint32_t> bins(off.back() ? off.back() : 1), len(bins.size());
        std::vector<uint64_t> gc(bins.size());
        std::vector<uint8_t> touched((size_t)(hdr.n_targets() > 0 ? hdr.n_targets() : 1));
        uint64_t n_count = 0;
        if (multi_done) {
            hpn_window_finish(ctx, bins.data(), gc.data(), len.data(), touched.data(), &n_count);   // closes this context's (unused) accumulation
            bins = m_bins, len = m_len, gc = m_gc, touched = m_touched, n_count = m_count;
        } else if ((rc = hpn_window_finish(ctx, bins.data(), gc.data(), len.data(), touched.data(), &n_count)) != HPN_OK) {
            die_hpn(ctx, rc, infiles[i]);
        }
        fprintf(stderr, "Done load bam file %s at %.3f s\n", infiles[i], (double)(usec() - begin) / CLOCKS_PER_SEC);
        if (i == 0) {
            hdr0 = hdr, off0 = off, bins0 = bins, len0 = len, gc0 = gc, touched0 = touched, n_count0 = n_count;
        }
        // draw_hits (PNG) is out of scope; the reference's progress line is kept
        fprintf(stderr, "Done draw hit %s_hits.png at %.3f s\n", infiles[i], (double)(usec() - begin) / CLOCKS_PER_SEC);
        fprintf(stderr, "%lu\n", (unsigned long)n_count0);  // databuf->n_count: always the first file's (:414)
    }
    if (n_in > 0) {
        size_t bt = 0;
        uint64_t bw = 0;
        if (!window_gc_in_domain(hdr0.target_len, off0.data(), gc0.data(), touched0.data(), &bt, &bw)) {
            // GC[tid][w] += gc is a float32 sum in the reference (bam_sliding_count.c:121): from 2^24 on its value depends on
            // the order of the additions, which this build does not keep -- refuse rather than print other digits
            fprintf(stderr, "bam_sliding_count: window %llu of %s holds %llu G/C bases (2^24 or more): the reference's float32 sum is "
                    "order-dependent there; use a smaller -w\n", (unsigned long long)(bw + 1), hdr0.target_name[bt].c_str(),
                    (unsigned long long)gc0[off0[bt] + bw]);
            return 2;
        }
        std::string path = std::string(outfile) + ".txt";
        FILE *out = fopen(path.c_str(), "wb");
        if (!out) err(1, "Failed to open file (%s)", path.c_str());
        print_window_report(out, hdr0.target_name, hdr0.target_len, (uint32_t)window, off0.data(), bins0.data(), gc0.data(),
                            len0.data(), touched0.data());
        fclose(out);
    }
    fprintf(stderr, "Done output %s.txt at %.3f s\n", outfile, (double)(usec() - begin) / CLOCKS_PER_SEC);
    stamp("finished");
    quick_exit_ok();
}
