// fastq_trim -- drop-in for the reference tool of the same name (fastq_trim.c): cut every
// read to cycles [S, E), the substring gather running on MI355X through libhpngs.
//
//   fastq_trim [-i IN] [-o OUT] [-s S] [-e E] [-h]        (-v and -z are accepted and ignored)
//
// Output OUT.trim.fastq (stdout when OUT starts with '-'): name, seq[S:min(E,len)], "+",
// qual[S:min(E,len)] (fastq_trim.c:101).  Domain: 0 <= S <= E and S <= every read's
// length (beyond a read's end the reference copies stale buffer bytes; this prints an
// empty line there).
#include <getopt.h>

#include "../host/bam_gpu.hpp"
#include "../host/fastq_reader.hpp"
#include "../host/tally_stream.hpp"
#include "../host/text_stream.hpp"
#include "../host/report.hpp"

using namespace hpn;

static void usage(const char *prog)
{
    fprintf(stderr,
            "\nUsage: %s [-i Infile] [-o OUTFILE] [-s start] [-e end] [-h]\n"
            "  Cuts the reads of a plain or gzip FASTQ file to the given cycles\n"
            "  (MI355X build of HighPerformanceNGS fastq_trim).\n\n"
            "   [-i Infile]  input, default stdin\n"
            "   [-o OUTPUT]  output prefix (writes OUTPUT.trim.fastq), default stdout\n"
            "   [-s Start]   0-based start position, default 0\n"
            "   [-e End]     1-based end position, default 400\n"
            "   [-h]         this help\n\n",
            prog);
    exit(1);
}

int main(int argc, char *argv[])
{
    bind_before_runtime();     // (host/cpus.hpp: next to the device before the runtime starts)
    const char *infile = "-", *outfile = "-";
    int start = 0, end = 400;
    if (argc < 2) usage(argv[0]);
    int opt;
    while ((opt = getopt(argc, argv, "i:o:s:e:vzh?")) != -1) {
        switch (opt) {
        case 'i': infile = optarg; break;
        case 'o': outfile = optarg; break;
        case 'v': break;
        case 'z': break;
        case 's': start = atoi(optarg); break;
        case 'e': end = atoi(optarg); break;
        case '?':
        case 'h': usage(argv[0]); break;
        default: fprintf(stderr, "error parameter!\n"); break;
        }
    }
    if (start < 0 || end < start) {
        fprintf(stderr, "fastq_trim: need 0 <= start <= end (got -s %d -e %d)\n", start, end);
        return 2;
    }
    hpn_ctx *ctx = nullptr;
    int dev0 = 0, ndev = 1;
    if (const char *d = getenv("HPN_DEVICE")) dev0 = atoi(d);
    else if (hpn_device_count(&ndev) != HPN_OK || ndev < 1) ndev = 1;
    int rc = hpn_ctx_create(dev0, &ctx);
    if (rc != HPN_OK) die_hpn(nullptr, rc, "hpn_ctx_create");
    bind_for_device(ctx);

    FILE *out = fcreat_outfile(outfile, ".trim.fastq");
    unsigned long reads = 0;
    const long long begin = usec();
    InStream in;
    bool exact = !text_path_enabled();
    bool done = false;
    // bgzip-compressed input, output to a file: the BGZF blocks are inflated on the GPU (as for BAM) and the text
    // is framed and cut where it lands; the host moves compressed bytes in and trimmed text out.  A damaged block
    // or irregular text: the output is emptied and the file goes through the paths below from its first byte.
    const bool to_file = !(strncmp(outfile, "-", 1) == 0 || !strcmp(outfile, ""));
    const bool from_file = !(strncmp(infile, "-", 1) == 0 || !strcmp(infile, ""));
    // device text -> trimmed text, in 64 MB slices (each with its own pinned output buffer); false: irregular text
    const size_t slice = (size_t)64 << 20, ocap = slice + 8192 + 64;
    auto cut_device_text = [&](AsyncWriter &writer, const uint8_t *d_text, uint64_t total, bool fin) {
        for (uint64_t at = 0; at < total || (fin && total == 0);) {
            const uint64_t k = total - at < slice ? total - at : slice;
            hpn_text_info info;
            int oi;
            void *obuf = writer.acquire(&oi);
            rc = hpn_fastq_text_trim(ctx, d_text + at, k, fin && at + k == total, start, end, obuf, ocap, &info);
            if (rc != HPN_OK) die_hpn(ctx, rc, "hpn_fastq_text_trim");
            if (info.irregular) {
                writer.submit(oi, 0);
                return false;
            }
            writer.submit(oi, info.n_bytes);
            reads += info.n_records;
            at += k;
            if (total == 0) break;
        }
        return true;
    };
    auto write_failed = [&]() {
        fprintf(stderr, "fastq_trim: writing the output failed (%s)\n", errno ? strerror(errno) : "short write");
        leave(2);
    };
    auto start_over = [&] {
        fflush(out);
        if (ftruncate(fileno(out), 0) != 0 || fseek(out, 0, SEEK_SET) != 0) die_hpn(ctx, HPN_E_STATE, "fastq_trim: cannot rewind the output");
        reads = 0;
    };
    if (!exact && to_file && from_file && bam_gpu_enabled() && !test_env("HPN_NO_BGZF") && is_bgzf_file(infile)) {
        BgzfGpuStream gs;
        bool usable = gs.open_text(ctx, infile);
        if (usable) {
            AsyncWriter writer(ctx, out, ocap);
            if (!writer.ok()) die_hpn(ctx, HPN_E_NOMEM, "fastq_trim");
            if ((rc = hpn_fastq_text_begin(ctx)) != HPN_OK) die_hpn(ctx, rc, "fastq_trim");
            for (bool fin = false; usable && !fin;) {
                hpn_raw_info bi;
                const int r = gs.next(&bi);
                if (r < 0) {
                    usable = false;
                    break;
                }
                fin = r == 0 || gs.at_eof();
                usable = cut_device_text(writer, gs.d_raw(), r == 0 ? 0 : bi.n_records, fin);  // text mode: n_records = bytes inflated
            }
            writer.finish();
            if (writer.failed()) write_failed();
        }
        if (usable) done = true;
        else start_over();
    }
    // a plain .fastq.gz (one member): block starts found here, the stretches inflated on the GPU (host/gz_gpu.hpp)
    // Writing the trimmed text is as slow as the host's own two-pass inflate on 16 cores (1.1 s vs 1.25 s on 3 GB), so
    // this route is taken when the host has few cores (or when asked for: HPN_GZ_GPU=1)
    const char *want_gz_gpu = getenv("HPN_GZ_GPU");
    const bool gz_on_gpu = gz_gpu_enabled() && (usable_cpus() <= 8 || (want_gz_gpu && want_gz_gpu[0] == '1') || test_env("HPN_GZ_GPU_FORCE"));
    if (!done && !exact && to_file && from_file && gz_on_gpu && !test_env("HPN_NO_MGZ") && !test_env("HPN_NO_PGZ") &&
        is_plain_gzip_file(infile)) {
        GzGpuStream gs;
        const long cpus = usable_cpus();
        uint32_t per_call = 5120;
        (void)hpn_inflate_slots(ctx, &per_call);                          // stretches the chip decodes at once
        const uint32_t slots = per_call;
        if (const char *e = test_env("HPN_GZ_BATCH")) per_call = (uint32_t)atol(e);
        // several device calls per file, so that the writer thread has text to write while the next part is inflated:
        // a quarter of the file per call, in stretches small enough to fill the chip each time
        size_t stretch = 0;
        struct stat sb;
        if (!test_env("HPN_GZ_STRETCH") && stat(infile, &sb) == 0) {
            stretch = ((size_t)sb.st_size / 4 / slots + 65536) & ~(size_t)65535;
            stretch = stretch < ((size_t)256 << 10) ? (size_t)256 << 10 : stretch > ((size_t)1 << 20) ? (size_t)1 << 20 : stretch;
        }
        bool usable = gs.open(ctx, infile, (int)(cpus < 1 ? 1 : cpus > 16 ? 16 : cpus), per_call < 1 ? 1 : per_call, stretch);
        if (usable) {
            AsyncWriter writer(ctx, out, ocap);
            if (!writer.ok()) die_hpn(ctx, HPN_E_NOMEM, "fastq_trim");
            if ((rc = hpn_fastq_text_begin(ctx)) != HPN_OK) die_hpn(ctx, rc, "fastq_trim");
            for (bool fin = false; usable && !fin;) {
                uint64_t n = 0;
                const int r = gs.next(&n);
                if (r < 0) {
                    usable = false;
                    break;
                }
                fin = r == 0 || gs.at_end();
                usable = cut_device_text(writer, gs.d_text(), n, fin);
            }
            writer.finish();
            if (writer.failed()) write_failed();
            if (getenv("HPN_TIMING"))
                fprintf(stderr, usable ? "[hpn] gzip on the GPU: block starts %.3f s, upload %.3f s, device inflate %.3f s\n"
                                       : "[hpn] gzip route on the GPU abandoned (%.3f / %.3f / %.3f s)\n",
                        gs.seconds_find(), gs.seconds_upload(), gs.seconds_device());
        }
        if (usable) done = true;
        else start_over();
    }
    // Plain text over several GPUs: record blocks of the one input go to one context per device, every lane cuts its
    // pieces, the output slabs are written in piece order (host/text_shard.hpp; no collective).  Irregular text: the
    // output is emptied and the input goes through the one-context path below from its first byte.
    if (!done && !exact && to_file && from_file) {
        WorkerLanes lanes(ctx, dev0, 0, ndev, WorkerLanes::cap(ndev, 1));
        if (LaneGroup *group = lanes.for_file(infile)) {
            bool irregular = false;
            rc = trim_text_sharded(*group, infile, start, end, out, &reads, &irregular);
            if (rc != HPN_OK) die_hpn(ctx, rc, "fastq_trim");
            if (irregular) start_over();
            else done = true;
        }
    }
    if (done) {
    } else if (!exact) {
        // Fast path: raw text to the GPU, trimmed text back (framing, cut and formatting on the
        // device).  At the first chunk that is not regular FASTQ the bytes not yet written
        // out -- the carried-over tail, this chunk, whatever the reader has queued -- go to
        // the exact framer below; records before them are unaffected (readNextNode zeroes its
        // buffer per record, fastq_trim.c:97, so framing has no memory across records).
        TextPump pump(ctx, infile, text_chunk_bytes());
        if (!pump.ok()) die_hpn(ctx, HPN_E_NOMEM, "fastq_trim");
        const size_t ocap = pump.chunk_bytes() + 8192 + 64;
        AsyncWriter writer(ctx, out, ocap);  // chunk k is written while chunk k+1 is on the GPU
        if (!writer.ok()) die_hpn(ctx, HPN_E_NOMEM, "fastq_trim");
        if ((rc = hpn_fastq_text_begin(ctx)) != HPN_OK) die_hpn(ctx, rc, "fastq_trim");
        std::vector<char> carry;  // host copy of the bytes the device carries over
        TextPump::Chunk c;
        double t_wait = 0, t_gpu = 0, t1 = wall_s();
        while (pump.next(c)) {
            hpn_text_info info;
            int oi;
            void *obuf = writer.acquire(&oi);
            const double t2 = wall_s();
            t_wait += t2 - t1;
            rc = hpn_fastq_text_trim(ctx, c.p, c.n, c.eof, start, end, obuf, ocap, &info);
            const double t3 = wall_s();
            t_gpu += t3 - t2;
            if (rc != HPN_OK) die_hpn(ctx, rc, "hpn_fastq_text_trim");
            if (info.irregular) {
                writer.submit(oi, 0);
                auto rest = std::make_shared<std::vector<char>>(std::move(carry));
                rest->insert(rest->end(), (const char *)c.p, (const char *)c.p + c.n);
                pump.recycle(c);
                in = pump.stop(*rest);
                in.pre = rest;
                exact = true;
                break;
            }
            writer.submit(oi, info.n_bytes);
            reads += info.n_records;
            if (info.carry_bytes <= c.n) {  // the unfinished record lies within this chunk
                carry.assign((const char *)c.p + c.n - info.carry_bytes, (const char *)c.p + c.n);
            } else {  // ... or began in an earlier one
                carry.erase(carry.begin(), carry.end() - (ptrdiff_t)(info.carry_bytes - c.n));
                carry.insert(carry.end(), (const char *)c.p, (const char *)c.p + c.n);
            }
            pump.recycle(c);
            t1 = wall_s();
        }
        const double t4 = wall_s();
        writer.finish();
        if (writer.failed()) write_failed();
        if (getenv("HPN_TIMING"))
            fprintf(stderr, "[hpn] waiting for reader / writer %.3f s  copy+frame+trim+copy back %.3f s  final drain %.3f s\n", t_wait,
                    t_gpu, wall_s() - t4);
        if (!exact && pump.damaged()) {
            // A gzip member failed its CRC-32 / ISIZE / a data error: zlib -- hence the reference -- does not hand out the bytes of
            // the buffer it was filling when it noticed; the threaded inflaters delivered them.  Output to a file is done again
            // through zlib's own reader; output that cannot be taken back is refused rather than left different.
            if (!to_file) {
                fprintf(stderr, "fastq_trim: %s is a damaged gzip file (CRC-32 / ISIZE / data error) and the output cannot be rewound: "
                        "write to a file (-o) to get the reference's bytes\n", infile);
                return 2;
            }
            start_over();
            in = open_input_stream_exact(infile);
            exact = true;
        }
    } else {
        in = open_input_stream_exact(infile);
    }
    if (exact && !done) {
        for (int pass = 0; pass < 2; ++pass) {
            TrimFramer framer(in, start, end);
            FastqBatch b;
            const size_t kBytes = 64u << 20, kRecs = 1u << 20;
            if (!b.init(kBytes, kRecs, true)) die_hpn(ctx, HPN_E_NOMEM, "fastq_trim");
            std::vector<uint8_t> oseq(kBytes + kLineBuf), oqual(kBytes + kLineBuf);
            std::vector<uint64_t> ooff(kRecs + 1);
            bool more = true;
            while (more) {
                b.clear();
                more = framer.fill(b);
                const uint64_t n = b.n();
                if (!n) continue;
                rc = hpn_fastq_trim(ctx, b.seq, b.qual, b.off, n, start, end, oseq.data(), oqual.data(), ooff.data());
                if (rc != HPN_OK) die_hpn(ctx, rc, "hpn_fastq_trim");
                for (uint64_t i = 0; i < n; ++i) {  // fprintf("%s\n%s\n+\n%s\n") (:101): each cut ends at its first NUL
                    const uint64_t a = ooff[i], len = ooff[i + 1] - a;
                    fputs(b.names[i].c_str(), out);
                    fputc('\n', out);
                    fwrite(oseq.data() + a, 1, strnlen((const char *)oseq.data() + a, len), out);
                    fputs("\n+\n", out);
                    fwrite(oqual.data() + a, 1, strnlen((const char *)oqual.data() + a, len), out);
                    fputc('\n', out);
                }
                reads += n;
            }

            // the stream came from a threaded inflater that met a damaged gzip member: only zlib's own reader stops where the
            // reference's gzgets stops (see above)
            if (pass == 1 || in.exact || !in.damaged()) break;
            if (!to_file) {
                fprintf(stderr, "fastq_trim: %s is a damaged gzip file (CRC-32 / ISIZE / data error) and the output cannot be rewound: "
                        "write to a file (-o) to get the reference's bytes\n", infile);
                return 2;
            }
            start_over();
            in.close();
            in = open_input_stream_exact(infile);
        }
    }
    fprintf(stderr, "Total_reads: %lu\nFinished in %.3f s\n", reads, (double)(usec() - begin) / CLOCKS_PER_SEC);
    in.close();
    if (ferror(out) | fclose(out)) write_failed();     // (the reference's fprintf never looks; a short output with exit code 0 helps nobody)
    quick_exit_ok();
}
