// bam_synth.cpp -- synthetic coordinate-sorted BAM of the shape SURVEY.md §8(d) asks for, fast
// enough to make multi-GB inputs on the GPU box (bench / profiling input only, not the product):
//   g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/bam_synth -lz -lpthread
//   /tmp/bam_synth out.bam <reads> <contigs> <contig_len> [threads] [packed]
//   /tmp/bam_synth out.bam --targets name:len:reads,name:len:reads,... [threads] [soa_prefix]
//       targets of their own names / lengths / read counts (e.g. the 25 hg38 primary contigs); soa_prefix: the records
//       also as raw SoA files <prefix>.tid/.pos (int32) .flag (uint32) .kind (uint8, CIGAR of the mix below) .seq4
//       (75 bytes per record) -- what a checker needs to run the oracle without decoding the BAM again
// 150 bp reads, starts uniform per contig, CIGAR mix 85 % 150M, 5 % 40M2I108M, 5 % 60M5D90M,
// 5 % 10S140M; flags 90 % {0,16}, 10 % from {4,256,512,1024}; bases ACGT + 1 % N.
// Writes out.bam and out.bam.bai (bins + 16 kb linear index, as samtools index would: the reference
// tools load it with bam_index_load and fetch through it).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <future>
#include <map>
#include <memory>
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
    // BAM_SYNTH_HUFFMAN=1: Huffman-only blocks (no match search; ~4x faster to write, a larger file) for very large inputs
    static const int strategy = getenv("BAM_SYNTH_HUFFMAN") ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY;
    deflateInit2(&s, 1, Z_DEFLATED, -15, 8, strategy);
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

struct Compressed {
    std::vector<size_t> cut;
    std::vector<std::vector<uint8_t>> out;
};

static Compressed compress_blocks(const std::vector<uint8_t> &raw, int threads, bool records)
{
    const size_t kIn = 0xff00;
    Compressed c;
    c.cut.push_back(0);
    if (records) {
        size_t p = 0, start = 0;
        while (p < raw.size()) {
            uint32_t bs;
            memcpy(&bs, raw.data() + p, 4);
            if (p + 4 + bs - start > kIn && p > start) c.cut.push_back(p), start = p;
            p += 4 + (size_t)bs;
        }
    } else {
        for (size_t p = kIn; p < raw.size(); p += kIn) c.cut.push_back(p);
    }
    c.cut.push_back(raw.size());
    const size_t nb = c.cut.size() - 1;
    c.out.resize(nb);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            for (size_t b = (size_t)t; b < nb; b += (size_t)threads)
                if (c.cut[b + 1] > c.cut[b]) c.out[b] = bgzf_block(raw.data() + c.cut[b], c.cut[b + 1] - c.cut[b]);
        });
    for (auto &t : th) t.join();
    return c;
}

static std::vector<BlockAt> write_compressed(FILE *f, const Compressed &c)
{
    std::vector<BlockAt> at;
    uint64_t off = (uint64_t)ftello(f);
    for (size_t b = 0; b + 1 < c.cut.size(); ++b) {
        if (c.out[b].empty()) continue;
        at.push_back(BlockAt{off, c.cut[b], c.cut[b + 1]});
        fwrite(c.out[b].data(), 1, c.out[b].size(), f);
        off += c.out[b].size();
    }
    return at;
}

static std::vector<BlockAt> write_blocks(FILE *f, const std::vector<uint8_t> &raw, int threads, bool records)
{
    return write_compressed(f, compress_blocks(raw, threads, records));
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
    if (argc < 4) return fprintf(stderr, "usage: %s out.bam reads contigs contig_len [threads]  |  out.bam --targets name:len:reads,... [threads] [soa_prefix]\n", argv[0]), 1;
    std::vector<std::string> tname;
    std::vector<uint32_t> tlen;
    std::vector<uint64_t> treads;
    int threads = 8;
    bool per_block = true;           // no record straddles two BGZF blocks (what samtools writes)
    std::string soa;
    if (!strcmp(argv[2], "--targets")) {
        for (char *tok = strtok(argv[3], ","); tok; tok = strtok(nullptr, ",")) {
            char nm[128];
            double len = 0, rd = 0;
            if (sscanf(tok, "%127[^:]:%lf:%lf", nm, &len, &rd) != 3) return fprintf(stderr, "bad target %s\n", tok), 1;
            tname.push_back(nm), tlen.push_back((uint32_t)len), treads.push_back((uint64_t)rd);
        }
        if (argc > 4) threads = atoi(argv[4]);
        if (argc > 5) soa = argv[5];
    } else {
        if (argc < 5) return fprintf(stderr, "usage: %s out.bam reads contigs contig_len [threads]\n", argv[0]), 1;
        const uint64_t reads = (uint64_t)atof(argv[2]);
        const int contigs = atoi(argv[3]);
        for (int c = 0; c < contigs; ++c) tname.push_back("chr" + std::to_string(c + 1)), tlen.push_back((uint32_t)atof(argv[4])), treads.push_back(reads / (uint64_t)contigs);
        if (argc > 5) threads = atoi(argv[5]);
        per_block = argc <= 6;       // a 7th argument: pack records across blocks instead
    }
    if (const char *e = getenv("BAM_SYNTH_PACKED"))      // records packed across BGZF blocks (batches of 2^20 records still end with a block)
        if (e[0] == '1') per_block = false;
    const int contigs = (int)tname.size();
    FILE *f = fopen(argv[1], "wb");
    if (!f) return perror(argv[1]), 1;
    FILE *soa_f[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!soa.empty()) {
        static const char *ext[5] = {".tid", ".pos", ".flag", ".kind", ".seq4"};
        for (int k = 0; k < 5; ++k)
            if (!(soa_f[k] = fopen((soa + ext[k]).c_str(), "wb"))) return perror(soa.c_str()), 1;
    }
    std::vector<uint8_t> raw;
    {  // header
        std::string text = "@HD\tVN:1.0\tSO:coordinate\n";
        for (int c = 0; c < contigs; ++c) text += "@SQ\tSN:" + tname[(size_t)c] + "\tLN:" + std::to_string(tlen[(size_t)c]) + "\n";
        raw.insert(raw.end(), {'B', 'A', 'M', 1});
        put32(raw, (uint32_t)text.size());
        raw.insert(raw.end(), text.begin(), text.end());
        put32(raw, (uint32_t)contigs);
        for (int c = 0; c < contigs; ++c) {
            const std::string &nm = tname[(size_t)c];
            put32(raw, (uint32_t)nm.size() + 1);
            raw.insert(raw.end(), nm.begin(), nm.end());
            raw.push_back(0);
            put32(raw, tlen[(size_t)c]);
        }
        write_blocks(f, raw, threads, false);
    }
    Index idx;
    idx.bins.resize((size_t)contigs), idx.lin.resize((size_t)contigs);
    const uint64_t kBatch = 1u << 20;
    uint64_t name_base = 0;
    static const uint32_t cig[4][3] = {{150u << 4, 0, 0}, {40u << 4, (2u << 4) | 1, 108u << 4}, {60u << 4, (5u << 4) | 2, 90u << 4}, {(10u << 4) | 4, 140u << 4, 0}};
    static const int ncig[4] = {1, 3, 3, 2};
    static const uint32_t odd[4] = {4, 256, 512, 1024};
    // batches of <= 2^20 records of one target; batch k+1 is generated and compressed (all threads) while batch k is
    // written and indexed (this thread)
    struct Soa { std::vector<int32_t> pos; std::vector<uint32_t> flag; std::vector<uint8_t> kind, seq4; };
    struct Batch {
        int c = 0;
        std::vector<std::vector<uint8_t>> part;
        std::vector<Soa> sp;
        std::vector<uint8_t> raw;
        Compressed z;
    };
    struct Job { int c; uint64_t i0, n, per, clen, name_base; };
    std::vector<Job> jobs;
    for (int c = 0; c < contigs; name_base += treads[(size_t)c], ++c)
        for (uint64_t i0 = 0, per = treads[(size_t)c]; i0 < per; i0 += kBatch)
            jobs.push_back(Job{c, i0, per - i0 < kBatch ? per - i0 : kBatch, per, tlen[(size_t)c] > 400 ? tlen[(size_t)c] : 400u, name_base});
    auto prepare = [&](const Job &J) {
        auto Bp = std::make_shared<Batch>();
        Batch &B = *Bp;
        B.c = J.c;
        B.part.resize((size_t)threads), B.sp.resize((size_t)threads);
        const int c = J.c;
        const uint64_t i0 = J.i0, n = J.n, per = J.per, clen = J.clen, name_base = J.name_base;
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] {
                    std::vector<uint8_t> &v = B.part[(size_t)t];
                    const uint64_t lo = i0 + n * (uint64_t)t / (uint64_t)threads, hi = i0 + n * (uint64_t)(t + 1) / (uint64_t)threads;
                    v.resize((hi - lo) * 320);     // 32 + name (<= 22) + 12 + 75 + 150 + the 4-byte size
                    uint8_t *w = v.data();
                    Soa &so = B.sp[(size_t)t];
                    if (soa_f[0]) so.pos.resize(hi - lo), so.flag.resize(hi - lo), so.kind.resize(hi - lo), so.seq4.resize((hi - lo) * 75);
                    auto w32 = [&](uint32_t x) { memcpy(w, &x, 4), w += 4; };
                    auto w16 = [&](uint32_t x) { const uint16_t y = (uint16_t)x; memcpy(w, &y, 2), w += 2; };
                    for (uint64_t i = lo; i < hi; ++i) {
                        const uint64_t h = mix64(((uint64_t)c << 40) ^ i ^ 0x9E3779B97F4A7C15ull);
                        const uint32_t pos = (uint32_t)((double)i * (double)(clen - 160) / (double)per);
                        const int k = (h % 100) < 85 ? 0 : (int)((h % 100 - 85) / 5) + 1;
                        const uint32_t flag = ((h >> 8) % 10) ? (((h >> 16) & 1) ? 16u : 0u) : odd[(h >> 20) & 3];
                        char name[32];
                        int ln = 0;
                        {   // "r<decimal>" + NUL
                            char tmp[24];
                            int d = 0;
                            unsigned long long x = name_base + i;
                            do tmp[d++] = (char)('0' + x % 10), x /= 10; while (x);
                            name[ln++] = 'r';
                            while (d) name[ln++] = tmp[--d];
                            name[ln++] = 0;
                        }
                        const int kk = k > 3 ? 3 : k;
                        const uint32_t bs = 32 + (uint32_t)ln + 4u * (uint32_t)ncig[kk] + 75 + 150;
                        w32(bs), w32((uint32_t)c), w32(pos);
                        *w++ = (uint8_t)ln, *w++ = 30, w16(4680);
                        w16((uint32_t)ncig[kk]), w16(flag), w32(150);
                        w32(0xffffffffu), w32(0xffffffffu), w32(0);
                        memcpy(w, name, (size_t)ln), w += ln;
                        for (int q = 0; q < ncig[kk]; ++q) w32(cig[kk][q]);
                        uint64_t g = h;
                        uint8_t *sq = w;
                        for (int b = 0; b < 75; ++b) {
                            g = mix64(g + (uint64_t)b);
                            static const uint8_t code[4] = {1, 2, 4, 8};
                            uint8_t hi4 = code[g & 3], lo4 = code[(g >> 2) & 3];
                            if ((g >> 8) % 100 == 0) hi4 = 15;
                            if ((g >> 20) % 100 == 0) lo4 = 15;
                            *w++ = (uint8_t)(hi4 << 4 | lo4);
                        }
                        if (soa_f[0]) {
                            const size_t r = (size_t)(i - lo);
                            so.pos[r] = (int32_t)pos, so.flag[r] = flag, so.kind[r] = (uint8_t)kk;
                            memcpy(so.seq4.data() + r * 75, sq, 75);
                        }
                        for (int b = 0; b < 150; ++b) *w++ = (uint8_t)(2 + ((g >> (b & 31)) + (uint64_t)b * 7) % 40);
                    }
                    v.resize((size_t)(w - v.data()));
                });
        for (auto &t : th) t.join();
        size_t total = 0;
        for (auto &p : B.part) total += p.size();
        B.raw.reserve(total);
        for (auto &p : B.part) B.raw.insert(B.raw.end(), p.begin(), p.end()), std::vector<uint8_t>().swap(p);
        B.z = compress_blocks(B.raw, threads, per_block);
        return Bp;
    };
    std::future<std::shared_ptr<Batch>> next;
    if (!jobs.empty()) next = std::async(std::launch::async, prepare, jobs[0]);
    for (size_t j = 0; j < jobs.size(); ++j) {
        std::shared_ptr<Batch> Bp = next.get();
        if (j + 1 < jobs.size()) next = std::async(std::launch::async, prepare, jobs[j + 1]);
        const Batch &B = *Bp;
        const int c = B.c;
        const std::vector<uint8_t> &raw = B.raw;
        if (soa_f[0])
            for (const Soa &q : B.sp) {
                const std::vector<int32_t> tids(q.pos.size(), (int32_t)c);
                fwrite(tids.data(), 4, tids.size(), soa_f[0]), fwrite(q.pos.data(), 4, q.pos.size(), soa_f[1]);
                fwrite(q.flag.data(), 4, q.flag.size(), soa_f[2]), fwrite(q.kind.data(), 1, q.kind.size(), soa_f[3]);
                fwrite(q.seq4.data(), 1, q.seq4.size(), soa_f[4]);
            }
        const std::vector<BlockAt> at = write_compressed(f, B.z);
        if (!per_block) {                                  // packed (htsjdk's way): a record starts in block b and may end blocks later
            const uint64_t after = (uint64_t)ftello(f);
            auto voff = [&](size_t p, size_t &b) {         // bgzf_tell with p bytes of the batch consumed (b: cursor, moves forward)
                while (b < at.size() && p >= at[b].raw_end) ++b;
                return b < at.size() ? (at[b].file_off << 16 | (uint64_t)(p - at[b].raw_beg)) : after << 16;
            };
            size_t b0 = 0, b1 = 0;
            for (size_t p = 0; p < raw.size();) {
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
                const uint64_t vbeg = voff(p, b0), vend = voff(e, b1);
                idx.add(c, pos, pos + (rl ? rl : 1), vbeg, vend);
                p = e;
            }
        } else {                                           // index the batch: every record lies inside one block
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
    for (FILE *q : soa_f)
        if (q) fclose(q);
    idx.write(std::string(argv[1]) + ".bai");
    return 0;
}
