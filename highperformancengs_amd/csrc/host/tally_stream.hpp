// tally_stream.hpp -- one FASTQ stream -> hpn_tally, the body both fastq_count tools share.
//
// Stands where count_read stands (fastq_count.c:106-133): frame the stream exactly like
// the reference's 4 x gzgets loop (CountFramer) and add the per-record tally into the
// caller's accumulators -- but the tally runs on the GPU.  Two pinned batches alternate:
// while the GPU copies and scans batch k, the host frames batch k+1.
#pragma once
#include "../host/fastq_reader.hpp"
#include "../host/bam_gpu.hpp"
#include "../host/gz_gpu.hpp"
#include "../host/text_stream.hpp"
#include "../host/text_shard.hpp"
#include "../host/gz_shard.hpp"
#include "../host/bgzf_shard.hpp"
#include "hpngs.h"

namespace hpn {

constexpr size_t kBatchBytes = 64u << 20;  // quality bytes per batch
constexpr size_t kBatchRecs = 2u << 20;    // records per batch (short reads)

// Returns HPN_OK, an hpn_status, or HPN_E_DOMAIN with *too_long set when a read of 512+
// bases shows up (the reference would overrun SeqLen[512]).
inline int tally_stream(hpn_ctx *ctx, const InStream &fq, hpn_tally *acc, bool *too_long)
{
    struct Slot {
        FastqBatch b;
        void *d_qual = nullptr, *d_off = nullptr;
    } slot[2];
    int rc = HPN_OK;
    for (Slot &s : slot) {  // pinned host batches + device staging, allocated once (a failure frees what exists, below)
        void *q = nullptr, *o = nullptr;
        if (rc == HPN_OK && (rc = hpn_host_malloc(ctx, kBatchBytes + kLineBuf, &q)) == HPN_OK) s.b.qual = (uint8_t *)q;
        if (rc == HPN_OK && (rc = hpn_host_malloc(ctx, (kBatchRecs + 1) * sizeof(uint64_t), &o)) == HPN_OK) s.b.off = (uint64_t *)o, s.b.off[0] = 0;
        s.b.cap_bytes = kBatchBytes, s.b.cap_recs = kBatchRecs;
        if (rc == HPN_OK) rc = hpn_dev_malloc(ctx, kBatchBytes + kLineBuf + 64, &s.d_qual);
        if (rc == HPN_OK) rc = hpn_dev_malloc(ctx, (kBatchRecs + 1) * sizeof(uint64_t), &s.d_off);
    }
    const bool allocated = rc == HPN_OK;
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    CountFramer framer(fq);
    bool more = true, bad = false;
    int cur = 0;
    while (more && rc == HPN_OK) {
        FastqBatch &b = slot[cur].b;
        b.clear();
        more = framer.fill(b, &bad);  // the GPU is busy with the other slot meanwhile
        if (bad) {
            *too_long = true;
            rc = HPN_E_DOMAIN;
            break;
        }
        if (!b.n()) continue;
        if ((rc = hpn_ctx_sync(ctx)) != HPN_OK) break;  // the other slot's work is done: it may be refilled next
        if ((rc = hpn_memcpy_h2d(ctx, slot[cur].d_qual, b.qual, b.nbytes)) != HPN_OK) break;
        if ((rc = hpn_memcpy_h2d(ctx, slot[cur].d_off, b.off, (b.n() + 1) * sizeof(uint64_t))) != HPN_OK) break;
        rc = hpn_fastq_tally_dev(ctx, (const uint8_t *)slot[cur].d_qual, nullptr, (const uint64_t *)slot[cur].d_off, b.n(), flags);
        cur ^= 1;
    }
    if (rc == HPN_OK) rc = hpn_fastq_tally_fetch(ctx, acc);
    else if (allocated) hpn_ctx_sync(ctx);
    for (Slot &s : slot) {
        if (s.b.qual) hpn_host_free(ctx, s.b.qual);
        if (s.b.off) hpn_host_free(ctx, s.b.off);
        s.b.qual = nullptr, s.b.off = nullptr;
        if (s.d_qual) hpn_dev_free(ctx, s.d_qual);
        if (s.d_off) hpn_dev_free(ctx, s.d_off);
    }
    return rc;
}

// The same through the raw-text front end: the reader thread delivers bytes, the GPU frames
// and tallies them.  *irregular is set (and nothing is added to acc) when the text is not
// regular FASTQ in the sense of include/hpngs.h; the caller then runs tally_stream.
inline int tally_text_stream(hpn_ctx *ctx, const char *path, hpn_tally *acc, bool *irregular)
{
    *irregular = false;
    const bool timing = getenv("HPN_TIMING") != nullptr;  // phase times on stderr (diagnostics only)
    const double t0 = wall_s();
    TextPump pump(ctx, path, text_chunk_bytes());
    if (!pump.ok()) return HPN_E_NOMEM;
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    int rc = hpn_fastq_text_begin(ctx);
    TextPump::Chunk c;
    double t_wait = 0, t_gpu = 0, t1 = wall_s();
    const double t_setup = t1 - t0;
    uint64_t bytes = 0;
    while (rc == HPN_OK && pump.next(c)) {
        hpn_text_info info;
        const double t2 = wall_s();
        t_wait += t2 - t1;
        rc = hpn_fastq_text_count(ctx, c.p, c.n, c.eof, flags, &info);
        bytes += c.n;
        pump.recycle(c);
        t1 = wall_s();
        t_gpu += t1 - t2;
        if (rc == HPN_OK && info.irregular) {
            *irregular = true;
            break;
        }
    }
    if (timing)
        fprintf(stderr, "[hpn] %s: %.1f MB  setup %.3f s  waiting for the reader %.3f s  copy+frame+tally %.3f s\n", path,
                bytes / 1e6, t_setup, t_wait, t_gpu);
    if (rc == HPN_OK && !*irregular && pump.damaged()) {   // a damaged gzip stream: only zlib's own reader hands out the reference's bytes
        if (timing) fprintf(stderr, "[hpn] %s: the gzip stream is damaged: read again the reference's way\n", path);
        *irregular = true;
    }
    if (*irregular || rc != HPN_OK) {  // drop whatever earlier chunks added on the device
        hpn_tally scratch;
        memset(&scratch, 0, sizeof scratch);
        (void)hpn_fastq_tally_fetch(ctx, &scratch);
        return rc;
    }
    return hpn_fastq_tally_fetch(ctx, acc);
}

// bgzip-compressed FASTQ: compressed bytes to the GPU, BGZF blocks inflated there, the text framed
// and tallied where it lands -- the host never sees the text.  *unusable: a damaged block or a
// stream that is not regular FASTQ; nothing is added to acc and the caller takes another route.
inline bool is_bgzf_file(const char *path)
{
    uint8_t h[18] = {0};
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    const bool ok = pread(fd, h, 18, 0) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
    close(fd);
    return ok;
}
inline uint64_t bgzf_text_slice()   // HPN_BGZF_SLICE: tests cut small files into several calls
{
    const char *e = test_env("HPN_BGZF_SLICE");
    return e && atoll(e) >= 4096 ? (uint64_t)atoll(e) : (uint64_t)256 << 20;
}
inline int tally_bgzf_on_gpu(hpn_ctx *ctx, const char *path, hpn_tally *acc, bool *unusable)
{
    *unusable = false;
    BgzfGpuStream gs;
    if (!gs.open_text(ctx, path)) {
        *unusable = true;
        return HPN_OK;
    }
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    int rc = hpn_fastq_text_begin(ctx);
    while (rc == HPN_OK && !*unusable) {
        hpn_raw_info bi;
        const int r = gs.next(&bi);
        if (r < 0) {
            *unusable = true;
            break;
        }
        const bool last = r == 0 || gs.at_eof();
        // One inflate launch covers up to eight chunks of compressed bytes: at a ratio of 6 (binned qualities) that is far past the
        // 2 GiB one hpn_fastq_text_count call frames.  The text is cut anywhere, as on the gzip route below.
        const uint64_t n = r == 0 ? 0 : bi.n_records, slice = bgzf_text_slice();
        for (uint64_t at = 0; rc == HPN_OK && !*unusable && (at < n || (last && n == 0));) {
            const uint64_t k = n - at < slice ? n - at : slice;
            hpn_text_info info;
            rc = hpn_fastq_text_count(ctx, (const uint8_t *)gs.d_raw() + at, k, last && at + k == n, flags, &info);
            if (rc == HPN_OK && info.irregular) *unusable = true;
            at += k;
            if (n == 0) break;
        }
        if (last) break;
    }
    if (*unusable || rc != HPN_OK) {
        hpn_tally scratch;
        memset(&scratch, 0, sizeof scratch);
        (void)hpn_fastq_tally_fetch(ctx, &scratch);
        return rc;
    }
    return hpn_fastq_tally_fetch(ctx, acc);
}

// A single-member .fastq.gz: block starts found on the host, the stretches inflated on the device (host/gz_gpu.hpp),
// the text framed and tallied where it lands.  *unusable as above.
inline bool is_plain_gzip_file(const char *path)  // gzip, not BGZF
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    uint8_t h[18] = {0};
    struct stat sb;
    const bool gz = pread(fd, h, 18, 0) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 &&
                    !((h[3] & 4) && h[12] == 'B' && h[13] == 'C') && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    const bool force = test_env("HPN_GZ_GPU_FORCE") != nullptr;  // tests: small files too
    // several members (cat a.gz b.gz) are decoded in one go (kernels/gz_inflate.hip); HPN_GZ_MEMBERS=0: such files go to the
    // host's member-parallel reader as before
    const bool members_too = !(test_env("HPN_GZ_MEMBERS") && test_env("HPN_GZ_MEMBERS")[0] == '0');
    const bool ok = gz && (force || sb.st_size >= (8 << 20)) && (members_too || !gzip_has_second_member(fd, force));
    close(fd);
    return ok;
}
inline int tally_gz_on_gpu(hpn_ctx *ctx, const char *path, hpn_tally *acc, bool *unusable)
{
    *unusable = false;
    GzGpuStream gs;
    const long cpus = usable_cpus() / text_workers_in_flight();
    uint32_t slots = 5120;
    (void)hpn_inflate_slots(ctx, &slots);                             // stretches the chip decodes at once (24 decoder waves per CU: 6,144)
    uint32_t per_call = (uint32_t)(slots / (uint32_t)text_workers_in_flight());
    if (const char *e = test_env("HPN_GZ_BATCH")) per_call = (uint32_t)atol(e);  // (tests: several device calls per file)
    const double t0 = wall_s();
    if (!gs.open(ctx, path, (int)(cpus < 1 ? 1 : cpus > 16 ? 16 : cpus), per_call < 1 ? 1 : per_call)) {
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] gzip route on the GPU not taken: %s\n", gs.why());
        *unusable = true;
        return HPN_OK;
    }
    // Highly compressible text (long matches) is cheap for the host's own two-pass reader: measured on a 308 MB file of
    // ratio 4.7 with 16 cores, 0.39 s there against 0.48 s here (the search and one round of wavefronts are ~0.2 s whatever
    // the size); at ratio 1.7 this route is twice as fast from 700 MB on.  Small, compressible, plenty of cores: the host.
    if (!test_env("HPN_GZ_GPU_FORCE") && gs.ratio() > 3.5 && gs.file_bytes() < ((uint64_t)1 << 30) && cpus >= 12) {
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] gzip route on the GPU not taken: small and compressible (x%.1f), the host cores do it\n", gs.ratio());
        *unusable = true;
        return HPN_OK;
    }
    const double t_open = wall_s() - t0;
    stamp("gzip route open");
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    int rc = hpn_fastq_text_begin(ctx);
    // the batch's text is framed where it lies (hpn_fastq_text_count_inplace: no copy into a slot), 1 GiB at a time (round 5:
    // 256 MiB slices copied first -- 60 calls a batch, each with its launches, its wait and its 2 x 256 MiB of copy traffic)
    uint64_t slice = (uint64_t)1 << 30;
    if (const char *e = test_env("HPN_TEXT_SLICE")) slice = (uint64_t)atoll(e) < 64 ? 64 : (uint64_t)atoll(e);   // (tests: several slices in a small batch)
    while (rc == HPN_OK && !*unusable) {
        uint64_t n = 0;
        const int r = gs.next(&n);
        if (r < 0) {
            *unusable = true;
            break;
        }
        stamp("batch inflated, GB of text:", (double)n / 1e9);
        const bool fin = r == 0 || gs.at_end();
        for (uint64_t at = 0; rc == HPN_OK && !*unusable && (at < n || (fin && n == 0));) {
            const uint64_t k = n - at < slice ? n - at : slice;
            hpn_text_info info;
            rc = hpn_fastq_text_count_inplace(ctx, gs.d_text() + at, k, fin && at + k == n, flags, &info);
            if (rc == HPN_OK && info.irregular) *unusable = true;
            at += k;
            if (n == 0) break;
        }
        stamp("batch framed and tallied");
        if (fin) break;
    }
    if (*unusable || rc != HPN_OK) {
        hpn_tally scratch;
        memset(&scratch, 0, sizeof scratch);
        (void)hpn_fastq_tally_fetch(ctx, &scratch);
        if (getenv("HPN_TIMING")) fprintf(stderr, "[hpn] gzip route on the GPU abandoned after %.3f s: %s\n", wall_s() - t0, gs.why());
        return rc;
    }
    if (getenv("HPN_TIMING"))
        fprintf(stderr, "[hpn] gzip on the GPU: inflate + frame + tally %.3f s (open %.3f s, block starts %.3f s, upload %.3f s, device inflate %.3f s)\n",
                wall_s() - t0, t_open, gs.seconds_find(), gs.seconds_upload(), gs.seconds_device());
    return hpn_fastq_tally_fetch(ctx, acc);
}

// One input file of fastq_count / fastq_count_kthread (count_read, fastq_count.c:106-133).
// group: the lanes to shard a text input over (host/text_shard.hpp), nullptr = this context alone.
inline int tally_file(hpn_ctx *ctx, const char *path, hpn_tally *acc, bool *too_long, LaneGroup *group = nullptr)
{
    const bool is_stdin = strncmp(path, "-", 1) == 0 || !strcmp(path, "");
    if (text_path_enabled() && !is_stdin && bam_gpu_enabled() && !test_env("HPN_NO_BGZF") && is_bgzf_file(path)) {
        if (group && group->lanes() > 1) {      // chunks of whole blocks over one context per device (host/bgzf_shard.hpp)
            bool unusable = false;
            const int rc = tally_bgzf_sharded(*group, path, acc, &unusable);
            if (!unusable) return rc;
        }
        bool unusable = false;
        const int rc = tally_bgzf_on_gpu(ctx, path, acc, &unusable);
        if (!unusable) return rc;
    }
    if (text_path_enabled() && !is_stdin && gz_gpu_enabled() && !test_env("HPN_NO_MGZ") && !test_env("HPN_NO_PGZ") && is_plain_gzip_file(path)) {
        if (group && group->lanes() > 1) {      // batches of its stretches over one context per device (host/gz_shard.hpp)
            bool unusable = false;
            const int rc = tally_gz_sharded(*group, path, acc, &unusable);
            if (!unusable) return rc;
        }
        bool unusable = false;
        const int rc = tally_gz_on_gpu(ctx, path, acc, &unusable);
        if (!unusable) return rc;
    }
    if (text_path_enabled() && !is_stdin && group && group->lanes() > 1) {  // record blocks of this one input over several GPUs
        bool irregular = false;
        const int rc = tally_text_sharded(*group, path, acc, &irregular);
        if (!irregular) return rc;
    }
    if (text_path_enabled() && !is_stdin) {  // an irregular stream is framed again from its first byte
        bool irregular = false;
        const int rc = tally_text_stream(ctx, path, acc, &irregular);
        if (!irregular) return rc;
    }
    // the exact route: the 4 x gzgets framer on the host, over zlib's own reader (what the reference reads with).  With the text
    // front end switched off (HPN_TEXT=0) the threaded inflaters are tried first; a stream they find damaged is read again.
    if (!text_path_enabled() && !is_stdin) {
        InStream fq = open_input_stream(path);
        hpn_tally tmp;
        memset(&tmp, 0, sizeof tmp);
        std::vector<uint64_t> qh;
        if (acc->qual_hist) qh.assign((size_t)HPN_QUAL_ROWS * HPN_LEN_BINS, 0), tmp.qual_hist = qh.data();
        const int rc = tally_stream(ctx, fq, &tmp, too_long);
        const bool damaged = fq.damaged();
        fq.close();
        if (!damaged || rc != HPN_OK) {
            for (int l = 0; l < HPN_LEN_BINS; ++l) acc->seqlen[l] += tmp.seqlen[l];
            acc->total += tmp.total, acc->q20 += tmp.q20, acc->q30 += tmp.q30;
            if (acc->qual_hist)
                for (size_t k = 0; k < qh.size(); ++k) acc->qual_hist[k] += qh[k];
            return rc;
        }
        *too_long = false;
    }
    InStream fq = open_input_stream_exact(path);
    const int rc = tally_stream(ctx, fq, acc, too_long);
    fq.close();
    return rc;
}

}  // namespace hpn
