// bam_synth.cpp -- synthetic coordinate-sorted BAM of the shape SURVEY.md §8(d) asks for, fast
// enough to make multi-GB inputs on the GPU box (bench / profiling input only, not the product):
//   g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/bam_synth -lz -lpthread
//   /tmp/bam_synth out.bam <reads> <contigs> <contig_len> [threads] [packed]
// 150 bp reads, starts uniform per contig, CIGAR mix 85 % 150M, 5 % 40M2I108M, 5 % 60M5D90M,
// 5 % 10S140M; flags 90 % {0,16}, 10 % from {4,256,512,1024}; bases ACGT + 1 % N.
// Writes out.bam and out.bam.bai (bins + 16 kb linear index, as samtools index would: the reference
// tools load it with bam_index_load and fetch through it).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <map>
#include <string>
#include <thread>
#include <vector>

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 30, x *= 0xBF58476D1CE4E5B9ull, x ^= x >> 27, x *= 0x94D049BB133111EBull, x ^= x >> 31;
    return x;
}
static void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
static void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x), v.push_back((uint8_t)(x >> 8)); }

static std::vector<uint8_t> bgzf_block(const uint8_t *src, size_t n)
{
    std::vector<uint8_t> out(18 + compressBound(n) + 8);
    z_stream s;
    memset(&s, 0, sizeof s);
    deflateInit2(&s, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    s.next_in = (Bytef *)src, s.avail_in = (uInt)n;
    s.next_out = out.data() + 18, s.avail_out = (uInt)(out.size() - 18);
    deflate(&s, Z_FINISH);
    const size_t clen = s.total_out;
    deflateEnd(&s);
    const uint8_t hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
    memcpy(out.data(), hdr, 12);
    out[12] = 'B', out[13] = 'C', out[14] = 2, out[15] = 0;
    const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
    out[16] = (uint8_t)bsize, out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), src, (uInt)n);
    for (int i = 0; i < 4; ++i) out[18 + clen + i] = (uint8_t)(crc >> (8 * i)), out[18 + clen + 4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
    out.resize(18 + clen + 8);
    return out;
}

// records = true: `raw` is a run of whole BAM records and no record may straddle two blocks
// (what samtools' bam_write1 guarantees through bgzf_flush_try, bam.c:238)
struct BlockAt {
    uint64_t file_off;     // where the compressed block starts in the file
    size_t raw_beg, raw_end;
};

// .bai under construction: bin -> chunks of virtual offsets, and the 16 kb linear index, per contig
struct Index {
    std::vector<std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins;
    std::vector<std::vector<uint64_t>> lin;
    static uint32_t reg2bin(uint32_t beg, uint32_t end)
    {
        --end;
        if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
        if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
        if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
        if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
        if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
        return 0;
    }
    void add(int tid, uint32_t pos, uint32_t rend, uint64_t vbeg, uint64_t vend)
    {
        auto &ch = bins[(size_t)tid][reg2bin(pos, rend)];
        if (!ch.empty() && ch.back().second == vbeg) ch.back().second = vend;
        else ch.emplace_back(vbeg, vend);
        auto &l = lin[(size_t)tid];
        for (uint32_t w = pos >> 14; w <= (rend - 1) >> 14; ++w) {
            if (l.size() <= w) l.resize(w + 1, 0);
            if (l[w] == 0 || vbeg < l[w]) l[w] = vbeg;
        }
    }
    void write(const std::string &path) const
    {
        FILE *f = fopen(path.c_str(), "wb");
        auto w32 = [&](uint32_t x) { fwrite(&x, 4, 1, f); };
        auto w64 = [&](uint64_t x) { fwrite(&x, 8, 1, f); };
        fwrite("BAI\1", 1, 4, f);
        w32((uint32_t)bins.size());
        for (size_t t = 0; t < bins.size(); ++t) {
            w32((uint32_t)bins[t].size());
            for (auto &kv : bins[t]) {
                w32(kv.first), w32((uint32_t)kv.second.size());
                for (auto &c : kv.second) w64(c.first), w64(c.second);
            }
            w32((uint32_t)lin[t].size());
            uint64_t last = 0;
            for (uint64_t v : lin[t]) {   // windows without a record carry the previous offset (bam_index.c fill_missing)
                if (v == 0) v = last;
                last = v;
                w64(v);
            }
        }
        fclose(f);
    }
};

static std::vector<BlockAt> write_blocks(FILE *f, const std::vector<uint8_t> &raw, int threads, bool records)
{
    const size_t kIn = 0xff00;
    std::vector<size_t> cut{0};
    if (records) {
        size_t p = 0, start = 0;
        while (p < raw.size()) {
            uint32_t bs;
            memcpy(&bs, raw.data() + p, 4);
            if (p + 4 + bs - start > kIn && p > start) cut.push_back(p), start = p;
            p += 4 + (size_t)bs;
        }
    } else {
        for (size_t p = kIn; p < raw.size(); p += kIn) cut.push_back(p);
    }
    cut.push_back(raw.size());
    const size_t nb = cut.size() - 1;
    std::vector<std::vector<uint8_t>> out(nb);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            for (size_t b = (size_t)t; b < nb; b += (size_t)threads)
                if (cut[b + 1] > cut[b]) out[b] = bgzf_block(raw.data() + cut[b], cut[b + 1] - cut[b]);
        });
    for (auto &t : th) t.join();
    std::vector<BlockAt> at;
    uint64_t off = (uint64_t)ftello(f);
    for (size_t b = 0; b < nb; ++b) {
        if (out[b].empty()) continue;
        at.push_back(BlockAt{off, cut[b], cut[b + 1]});
        fwrite(out[b].data(), 1, out[b].size(), f);
        off += out[b].size();
    }
    return at;
}

int main(int argc, char **argv)
{
    if (argc >= 4 && !strcmp(argv[1], "--bgzip")) {  // bam_synth --bgzip in out [threads]: what `bgzip -c in > out` does
        FILE *in = fopen(argv[2], "rb"), *out = fopen(argv[3], "wb");
        if (!in || !out) return perror("bgzip"), 1;
        const int threads = argc > 4 ? atoi(argv[4]) : 8;
        std::vector<uint8_t> buf((size_t)0xff00 * 4096);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), in)) > 0) {
            std::vector<uint8_t> piece(buf.begin(), buf.begin() + n);
            write_blocks(out, piece, threads, false);
        }
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        fwrite(eof, 1, 28, out);
        fclose(in), fclose(out);
        return 0;
    }
    if (argc < 5) return fprintf(stderr, "usage: %s out.bam reads contigs contig_len [threads]\n", argv[0]), 1;
    const uint64_t reads = (uint64_t)atof(argv[2]);
    const int contigs = atoi(argv[3]);
    const uint32_t clen = (uint32_t)atof(argv[4]);
    const int threads = argc > 5 ? atoi(argv[5]) : 8;
    FILE *f = fopen(argv[1], "wb");
    if (!f) return perror(argv[1]), 1;
    std::vector<uint8_t> raw;
    {  // header
        std::string text = "@HD\tVN:1.0\tSO:coordinate\n";
        for (int c = 0; c < contigs; ++c) text += "@SQ\tSN:chr" + std::to_string(c + 1) + "\tLN:" + std::to_string(clen) + "\n";
        raw.insert(raw.end(), {'B', 'A', 'M', 1});
        put32(raw, (uint32_t)text.size());
        raw.insert(raw.end(), text.begin(), text.end());
        put32(raw, (uint32_t)contigs);
        for (int c = 0; c < contigs; ++c) {
            const std::string nm = "chr" + std::to_string(c + 1);
            put32(raw, (uint32_t)nm.size() + 1);
            raw.insert(raw.end(), nm.begin(), nm.end());
            raw.push_back(0);
            put32(raw, clen);
        }
        write_blocks(f, raw, threads, false);
    }
    Index idx;
    idx.bins.resize((size_t)contigs), idx.lin.resize((size_t)contigs);
    const uint64_t per = reads / (uint64_t)contigs, kBatch = 1u << 20;
    static const uint32_t cig[4][3] = {{150u << 4, 0, 0}, {40u << 4, (2u << 4) | 1, 108u << 4}, {60u << 4, (5u << 4) | 2, 90u << 4}, {(10u << 4) | 4, 140u << 4, 0}};
    static const int ncig[4] = {1, 3, 3, 2};
    static const uint32_t odd[4] = {4, 256, 512, 1024};
    for (int c = 0; c < contigs; ++c)
        for (uint64_t i0 = 0; i0 < per; i0 += kBatch) {
            const uint64_t n = per - i0 < kBatch ? per - i0 : kBatch;
            std::vector<std::vector<uint8_t>> part((size_t)threads);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t] {
                    std::vector<uint8_t> &v = part[(size_t)t];
                    const uint64_t lo = i0 + n * (uint64_t)t / (uint64_t)threads, hi = i0 + n * (uint64_t)(t + 1) / (uint64_t)threads;
                    v.reserve((hi - lo) * 300);
                    for (uint64_t i = lo; i < hi; ++i) {
                        const uint64_t h = mix64(((uint64_t)c << 40) ^ i ^ 0x9E3779B97F4A7C15ull);
                        const uint32_t pos = (uint32_t)((double)i * (double)(clen - 160) / (double)per);
                        const int k = (h % 100) < 85 ? 0 : (int)((h % 100 - 85) / 5) + 1;
                        const uint32_t flag = ((h >> 8) % 10) ? (((h >> 16) & 1) ? 16u : 0u) : odd[(h >> 20) & 3];
                        char name[32];
                        const int ln = snprintf(name, sizeof name, "r%llu", (unsigned long long)(c * per + i)) + 1;
                        const uint32_t bs = 32 + (uint32_t)ln + 4u * (uint32_t)ncig[k > 3 ? 3 : k] + 75 + 150;
                        const int kk = k > 3 ? 3 : k;
                        put32(v, bs), put32(v, (uint32_t)c), put32(v, pos);
                        v.push_back((uint8_t)ln), v.push_back(30), put16(v, 4680);
                        put16(v, (uint32_t)ncig[kk]), put16(v, flag), put32(v, 150);
                        put32(v, 0xffffffffu), put32(v, 0xffffffffu), put32(v, 0);
                        v.insert(v.end(), name, name + ln);
                        for (int q = 0; q < ncig[kk]; ++q) put32(v, cig[kk][q]);
                        uint64_t g = h;
                        for (int b = 0; b < 75; ++b) {
                            g = mix64(g + (uint64_t)b);
                            static const uint8_t code[4] = {1, 2, 4, 8};
                            uint8_t hi4 = code[g & 3], lo4 = code[(g >> 2) & 3];
                            if ((g >> 8) % 100 == 0) hi4 = 15;
                            if ((g >> 20) % 100 == 0) lo4 = 15;
                            v.push_back((uint8_t)(hi4 << 4 | lo4));
                        }
                        for (int b = 0; b < 150; ++b) v.push_back((uint8_t)(2 + ((g >> (b & 31)) + (uint64_t)b * 7) % 40));
                    }
                });
            for (auto &t : th) t.join();
            raw.clear();
            for (auto &p : part) raw.insert(raw.end(), p.begin(), p.end());
            const std::vector<BlockAt> at = write_blocks(f, raw, threads, argc <= 6);  // a 7th argument: pack records across blocks instead
            if (argc <= 6) {                                   // index the batch: every record lies inside one block
                const uint64_t after = (uint64_t)ftello(f);
                for (size_t b = 0; b < at.size(); ++b) {
                    const uint64_t next = b + 1 < at.size() ? at[b + 1].file_off : after;
                    for (size_t p = at[b].raw_beg; p < at[b].raw_end;) {
                        uint32_t bs, pos, flag_nc;
                        memcpy(&bs, raw.data() + p, 4), memcpy(&pos, raw.data() + p + 8, 4), memcpy(&flag_nc, raw.data() + p + 16, 4);
                        const uint8_t *cg = raw.data() + p + 36 + raw[p + 12];
                        uint32_t rl = 0;
                        for (uint32_t q = 0; q < (flag_nc & 0xffffu); ++q) {
                            uint32_t w;
                            memcpy(&w, cg + 4 * q, 4);
                            if ((w & 15) == 0 || (w & 15) == 2 || (w & 15) == 3 || (w & 15) == 7 || (w & 15) == 8) rl += w >> 4;
                        }
                        const size_t e = p + 4 + bs;
                        const uint64_t vbeg = at[b].file_off << 16 | (uint64_t)(p - at[b].raw_beg);
                        const uint64_t vend = e < at[b].raw_end ? (at[b].file_off << 16 | (uint64_t)(e - at[b].raw_beg)) : next << 16;
                        idx.add(c, pos, pos + (rl ? rl : 1), vbeg, vend);
                        p = e;
                    }
                }
            }
        }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, f);
    fclose(f);
    if (argc <= 6) idx.write(std::string(argv[1]) + ".bai");
    else fclose(fopen((std::string(argv[1]) + ".bai").c_str(), "wb"));   // packed records: our tools only (they need the file to exist)
    return 0;
}
