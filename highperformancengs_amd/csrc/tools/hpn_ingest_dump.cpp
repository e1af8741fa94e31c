// hpn_ingest_dump -- developer/test utility: run the host ingest (no GPU) and dump the
// structure-of-arrays batch the tools would hand to libhpngs.  Used by the CPU test
// suite to check the gzgets-framing emulation and the BAM decoder.
//
//   hpn_ingest_dump count FILE   -> u64 n, u64 off[n+1], u8 qual[off[n]], u8 seq[off[n]]
//   hpn_ingest_dump trim  FILE   -> u64 n, u64 off[n+1], u8 seq[..], u8 qual[..], then n NUL-terminated names
//   hpn_ingest_dump cat   FILE   -> the decompressed byte stream as the tools' readers deliver it
//   hpn_ingest_dump crc   FILE   -> crc32_fast and zlib's crc32 of the file's bytes in odd-sized pieces (hex, must agree)
//   hpn_ingest_dump bam   FILE   -> u64 n, i32 tid[n], i32 pos[n], u32 flag[n], i32 l_qseq[n],
//                                   u32 cigar_off[n+1], u32 cigar[..], u64 seq_off[n+1], u8 seq4[..]
#include <stdio.h>

#include "../host/bam_reader.hpp"
#include "../host/fastq_reader.hpp"

using namespace hpn;

template <typename T>
static void put(const std::vector<T> &v)
{
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), stdout);
}

int main(int argc, char **argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: %s count|trim|cat|crc|bam FILE   (count-exact / trim-exact / cat-exact: through zlib's own reader)\n", argv[0]);
        return 1;
    }
    std::string mode = argv[1];
    const bool exact = mode.size() > 6 && mode.compare(mode.size() - 6, 6, "-exact") == 0;   // the stream the tools read a damaged gzip file with
    if (exact) mode.resize(mode.size() - 6);
    if (mode == "count" || mode == "trim") {
        InStream f = exact ? open_input_stream_exact(argv[2]) : open_input_stream(argv[2]);
        // small batches on purpose: the dump is the concatenation of many refills
        FastqBatch b;
        if (!b.init(1u << 16, 1u << 10, true)) return 4;
        std::vector<uint64_t> off{0};
        std::vector<uint8_t> seq, qual;
        std::vector<std::string> names;
        bool bad = false, more = true;
        CountFramer cf(f);
        TrimFramer tf(f);
        while (more) {
            b.clear();
            more = mode == "count" ? cf.fill(b, &bad) : tf.fill(b);
            if (bad) return 3;
            const uint64_t base = off.back();
            for (uint64_t i = 0; i < b.n(); ++i) off.push_back(base + b.off[i + 1]);
            seq.insert(seq.end(), b.seq, b.seq + b.nbytes);
            qual.insert(qual.end(), b.qual, b.qual + b.nbytes);
            names.insert(names.end(), b.names.begin(), b.names.end());
        }
        f.close();
        const uint64_t n = off.size() - 1;
        fwrite(&n, 8, 1, stdout);
        put(off);
        if (mode == "count") put(qual), put(seq);
        else {
            put(seq), put(qual);
            for (auto &s : names) fwrite(s.c_str(), 1, s.size() + 1, stdout);
        }
        return 0;
    }
    if (mode == "cat") {
        InStream f = exact ? open_input_stream_exact(argv[2]) : open_input_stream(argv[2]);
        std::vector<uint8_t> buf(1u << 20);
        for (;;) {
            const int n = f.read(buf.data(), (unsigned)buf.size());
            if (n <= 0) break;
            fwrite(buf.data(), 1, (size_t)n, stdout);
        }
        if (test_env("HPN_READER_STATS"))  // which reader produced the stream, and how
            fprintf(stderr, "reader=%s accepted=%lu gaps=%lu fallback=%d crc_failed=%d find_s=%.3f decode_s=%.3f translate_s=%.3f\n",
                    f.pz ? "pgz" : f.mz ? "mgz" : f.bz ? "bgzf" : "zlib", f.pz ? (unsigned long)f.pz->chunks_accepted() : 0ul,
                    f.pz ? (unsigned long)f.pz->gaps_decoded() : 0ul, f.pz ? (int)f.pz->fell_back() : 0, f.pz ? (int)f.pz->crc_failed() : 0,
                    f.pz ? f.pz->seconds_find() : 0.0, f.pz ? f.pz->seconds_decode() : 0.0, f.pz ? f.pz->seconds_translate() : 0.0);
        if (f.damaged()) fprintf(stderr, "damaged\n");   // (what makes the tools read the file again, the reference's way)
        f.close();
        return 0;
    }
    if (mode == "crc") {
        FILE *fp = fopen(argv[2], "rb");
        if (!fp) return 2;
        std::vector<uint8_t> all;
        uint8_t tmp[65536];
        size_t k;
        while ((k = fread(tmp, 1, sizeof tmp, fp)) > 0) all.insert(all.end(), tmp, tmp + k);
        fclose(fp);
        uint32_t a = 0, b = 0;
        size_t o = 0, step = 1;
        while (o < all.size()) {  // pieces of 1, 3, 7, 15 ... bytes, capped: every alignment and the short-tail path
            const size_t n = step < all.size() - o ? step : all.size() - o;
            a = crc32_fast(a, all.data() + o, n);
            b = (uint32_t)crc32(b, all.data() + o, (uInt)n);
            o += n;
            step = step * 2 + 1 > 300000 ? 1 : step * 2 + 1;
        }
        printf("%08x %08x %08x\n", a, b, crc32_fast(0, all.data(), all.size()));
        return a == b ? 0 : 5;
    }
    if (mode == "bam") {
        BamReader r;
        BamHeader h;
        if (!r.open(argv[2], h)) return 2;
        BamBatch b;
        while (r.next(b, true)) {}
        const uint64_t n = b.n();
        fwrite(&n, 8, 1, stdout);
        put(b.tid), put(b.pos), put(b.flag), put(b.l_qseq), put(b.cigar_off), put(b.cigar), put(b.seq_off), put(b.seq4);
        fprintf(stderr, "%d targets\n", h.n_targets());
        return 0;
    }
    return 1;
}
