// report.hpp -- text writers of the drop-in tools: byte-for-byte the reference's formats.
// Each function cites the printf it reproduces.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "hpngs.h"

namespace hpn {

// End of main: everything the tool wrote is flushed and closed by now.  Skipping the HIP
// runtime's teardown (unpinning buffers, freeing device memory, unloading code objects: 80-160 ms)
// changes nothing observable; the driver reclaims all of it with the process.
[[noreturn]] inline void quick_exit_ok()
{
    fflush(NULL);
    if (getenv("HPN_FULL_EXIT")) exit(0);      // (profilers write their output from exit handlers: rocprofv3 -- tool ...)
    _exit(0);
}

// Any other way out of a tool once the device is in use (an error, a missing index): the same code the reference exits with,
// without the runtime's exit handlers -- a helper thread may be inside the runtime at this moment (a context being made, the
// reader ahead of the caller), and the runtime's own teardown crashes under it.
[[noreturn]] inline void leave(int code)
{
    fflush(NULL);
    if (getenv("HPN_FULL_EXIT")) exit(code);
    _exit(code);
}

inline long long usec()
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (((long long)tv.tv_sec) * 1000000) + tv.tv_usec;
}

// Where a tool's output goes (the rule of IO_stream.h:69-83): a name that is empty or begins with '-' means standard output,
// anything else is created or truncated, mode 0666.  A file that cannot be made is reported on stderr with the reference's
// words (no newline) and the caller gets whatever fdopen makes of -1, as there.
inline FILE *fopen_output_stream(const char *filename)
{
    const std::string name(filename ? filename : "");
    if (name.empty() || name[0] == '-') return fdopen(STDOUT_FILENO, "wb");
    const int made = ::open(name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (made < 0) fprintf(stderr, "Failed to create output file (%s)", name.c_str());
    return fdopen(made, "wb");
}

// fcreat_outfile (IO_stream.h:92-97): prefix + suffix, through the rule above.
inline FILE *fcreat_outfile(const char *outfile, const char *suffix)
{
    std::string s = std::string(outfile) + suffix;
    return fopen_output_stream(s.c_str());
}

// ---- fastq_count ---------------------------------------------------------------------

struct CountSummary {
    uint64_t reads = 0;      // sumFreq
    double bases = 0;        // sum of 1.0*h[l]*l as double
    uint32_t min_len = 0, max_len = 0;
};

// statSeqLen (fastq_count.c:63-74): min is the first non-empty bin seen while min is
// still 0, so bin 0 never becomes the minimum.
inline CountSummary summarise(const hpn_tally &t)
{
    CountSummary s;
    for (uint32_t l = 0; l < HPN_LEN_BINS; ++l) {
        if (!t.seqlen[l]) continue;
        s.reads += t.seqlen[l];
        s.bases += 1.0 * (double)t.seqlen[l] * l;
        if (!s.min_len) s.min_len = l;
        if (s.max_len < l) s.max_len = l;
    }
    return s;
}

inline void print_count_header(FILE *out)  // fastq_count.c:212
{
    fprintf(out, "#Filename\tReadCount\tBaseCount\tMeanLen\tMinLen\tMaxLen\tQ20(%%)\tQ30(%%)\n");
}

inline void print_count_row(FILE *out, const char *name, const hpn_tally &t, const CountSummary &s)  // :127
{
    fprintf(out, "%s\t%lu\t%.0f\t%.0f\t%u\t%u\t%.3f\t%.3f\n", name, (unsigned long)s.reads, s.bases,
            s.bases / (double)s.reads, s.min_len, s.max_len, 1.0 * t.q20 / t.total * 100, 1.0 * t.q30 / t.total * 100);
}

inline void print_kthread_file_row(FILE *out, const char *name, const hpn_tally &t, const CountSummary &s)
{   // fastq_count_kthread.c:141 -- the per-file read count is a uint32_t (:106,258)
    const uint32_t rc32 = (uint32_t)s.reads;
    fprintf(out, "%s\t%u\t%.0f\t%.0f\t%u\t%u\t%.3f\t%.3f\n", name, rc32, s.bases, s.bases / rc32, s.min_len, s.max_len,
            1.0 * t.q20 / t.total * 100, 1.0 * t.q30 / t.total * 100);
}

inline void print_len_detail(FILE *out, const uint64_t *seqlen, uint32_t min_len, uint32_t max_len)  // :49-61
{
    fprintf(out, "#Len:");
    for (uint32_t l = min_len; l <= max_len; ++l) fprintf(out, "\t%u", l);
    fprintf(out, "\n#Freq:");
    for (uint32_t l = min_len; l <= max_len; ++l) fprintf(out, "\t%lu", (unsigned long)seqlen[l]);
    fprintf(out, "\n");
}

inline void print_quality_matrix(FILE *out, const uint64_t *qual_hist, uint32_t ncol)  // kthread :52-64
{
    for (uint32_t q = 0; q < HPN_QUAL_ROWS; ++q)
        for (uint32_t p = 0; p < ncol; ++p)
            fprintf(out, p == ncol - 1 ? "%lu\n" : "%lu\t", (unsigned long)qual_hist[q * HPN_LEN_BINS + p]);
}

// ---- bam2depth -------------------------------------------------------------------------

// "%d" of a non-negative int, written backwards from `end`; returns the first character
inline char *put_int(char *end, int32_t v)
{
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    do {
        *--end = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (v < 0) *--end = '-';
    return end;
}

// "%s\t%d\t%d\t%d\n" per run (bam2depth.c:217).  hg38 at 30x is ~1e9 such lines: they are
// formatted by hand into a 4 MiB buffer instead of one fprintf each (same bytes).
inline void print_bedgraph(FILE *out, const char *chr, const hpn_run *runs, uint64_t n)
{
    const size_t lc = strlen(chr);
    std::vector<char> buf((4u << 20) + lc + 64);
    char *p = buf.data(), *const lim = buf.data() + (4u << 20);
    char num[16];
    for (uint64_t i = 0; i < n; ++i) {
        memcpy(p, chr, lc);
        p += lc;
        const int32_t f[3] = {runs[i].start, runs[i].end, runs[i].depth};
        for (int k = 0; k < 3; ++k) {
            *p++ = '\t';
            char *s = put_int(num + 16, f[k]);
            const size_t l = (size_t)(num + 16 - s);
            memcpy(p, s, l);
            p += l;
        }
        *p++ = '\n';
        if (p >= lim) {
            fwrite(buf.data(), 1, (size_t)(p - buf.data()), out);
            p = buf.data();
        }
    }
    if (p != buf.data()) fwrite(buf.data(), 1, (size_t)(p - buf.data()), out);
}

inline void print_depth_bins(FILE *out, const char *chr, uint32_t target_len, uint32_t W, const uint64_t *win_sum)
{   // output_bins (:238-246): the bins are doubles holding exact integer sums; divides by the full W
    const uint32_t windows = target_len / W + 1;
    for (uint32_t k = 0; k < windows; ++k) {
        const int ws = (int)(W * k);
        const int we = (W * (k + 1) > target_len) ? (int)target_len : (int)(W * (k + 1));
        fprintf(out, "%s\t%d\t%d\t%.2f\n", chr, ws, we, (double)win_sum[k] / W);
    }
}

inline void print_wig_bins(FILE *out, const char *chr, uint32_t target_len, uint32_t W, const uint64_t *win_sum)
{   // output_bins_wig (:248-255)
    const uint32_t windows = target_len / W + 1;
    fprintf(out, "variableStep chrom=%s span=%d\n", chr, (int)W);
    for (uint32_t k = 0; k < windows; ++k)
        if (win_sum[k]) fprintf(out, "%d\t%.2f\n", (int)(W * k), (double)win_sum[k] / W);
}

// ---- bam2wig -----------------------------------------------------------------------------------

// bam2wig's window bins.  Its overlap() (bam2wig.c:131-175) closes windows at (k+1)W-1
// INCLUSIVE and carries its window cursor from one run to the next, so a run that ends on
// a window's last base moves the cursor past a window the next run may still start in.
// The bins are therefore not a per-window sum of coverage; they are reproduced by feeding
// the runs (ascending, as the device scan emits them) through the same cursor.
// bins: target_len / W + 2 doubles (the reference may touch one past its own array).
inline void wig_bins_from_runs(const hpn_run *runs, uint64_t n_runs, uint32_t target_len, uint32_t W, double *bins)
{
    const int windows = (int)(target_len / W + 1);
    for (int k = 0; k <= windows; ++k) bins[k] = 0.0;
    int cur = 0, hit = 0;  // window cursor; windows the previous run added to
    for (uint64_t i = 0; i < n_runs; ++i) {
        const uint32_t rs = (uint32_t)runs[i].start, re = (uint32_t)runs[i].end;
        const double depth = (double)runs[i].depth;
        if (hit > 1) cur = cur - hit >= 0 ? cur - hit : 0;
        hit = 0;
        for (; cur <= windows;) {
            const uint32_t lo = W * (uint32_t)cur;
            uint32_t hi = ((uint32_t)cur + 1) * W - 1;
            if (hi > target_len) hi = target_len;
            if (re < lo) break;
            if (rs < lo) {          // run began in an earlier window
                ++hit;
                if (re < hi) {
                    bins[cur] += (re - lo) * depth;
                    break;
                }
                bins[cur++] += (hi - lo + 1) * depth;
            } else if (rs <= hi) {  // run begins in this window
                ++hit;
                if (re <= hi) {
                    bins[cur] += (re - rs) * depth;
                    break;
                }
                bins[cur++] += (hi - rs) * depth;
            } else {
                ++cur;
            }
        }
    }
}

inline void print_wig_bins_d(FILE *out, const char *chr, uint32_t target_len, uint32_t W, const double *bins)
{   // bam2wig.c output_bins_wig (:245-252): same text as bam2depth's wig writer
    const uint32_t windows = target_len / W + 1;
    fprintf(out, "variableStep chrom=%s span=%d\n", chr, (int)W);
    for (uint32_t k = 0; k < windows; ++k)
        if (bins[k]) fprintf(out, "%d\t%.2f\n", (int)(W * k), bins[k] / W);
}

// ---- bam_sliding_count ---------------------------------------------------------------------

// calc_winGC + output_count_GC (bam_sliding_count.c:126-164).  The reference keeps GC[k]
// as float32 and adds an unsigned short per record (:119-121); an integer sum converts to the
// same float as long as it stays below 2^24 (always, for windows up to ~100 kb at WGS
// depth).  From 2^24 on the reference's value depends on the ORDER of its additions (every
// += rounds), which integer sums taken in parallel -- and, over several GPUs, out of file
// order -- cannot give back: such a window is outside the domain, and window_gc_in_domain()
// says so before anything is printed (the tool then stops with HPN_E_DOMAIN's exit code
// instead of printing other digits than the reference).
// The per-chromosome running sums are replayed here in float32, window order.
inline bool window_gc_in_domain(const std::vector<uint32_t> &tlen, const uint64_t *win_off, const uint64_t *gc, const uint8_t *touched,
                                size_t *bad_target, uint64_t *bad_window)
{
    for (size_t t = 0; t < tlen.size(); ++t) {
        if (!touched[t]) continue;
        for (uint64_t s = win_off[t]; s < win_off[t + 1]; ++s)
            if (gc[s] >= (1ull << 24)) {
                *bad_target = t, *bad_window = s - win_off[t];
                return false;
            }
    }
    return true;
}

inline void print_window_report(FILE *out, const std::vector<std::string> &names, const std::vector<uint32_t> &tlen,
                                uint32_t W, const uint64_t *win_off, const uint32_t *bins, const uint64_t *gc,
                                const uint32_t *len, const uint8_t *touched)
{
    fprintf(out, "#chr\tchr_len\tchr_sum_read_count\tchr_sum_base\tchr_mean_cov\tchr_mean_GC%%");
    uint32_t max_len = 0;
    for (uint32_t l : tlen)
        if (l > max_len) max_len = l;
    const uint32_t max_windows = max_len / W + 1;
    for (uint32_t k = 0; k < max_windows; ++k) fprintf(out, "\t%u\tcount\tGC%%", k + 1);
    fprintf(out, "\n");
    for (size_t t = 0; t < names.size(); ++t) {
        if (!touched[t]) continue;  // windows[j]==0 (:155)
        const uint64_t slots = win_off[t + 1] - win_off[t];
        unsigned int sum_count = 0;
        float sum_gc = 0.0f;
        unsigned long sum_base = 0;
        std::vector<float> pct(slots);
        for (uint64_t k = 0; k < slots; ++k) {
            const uint64_t s = win_off[t] + k;
            const float g = (float)gc[s];
            sum_count += bins[s];
            sum_gc += g;
            sum_base += len[s];
            pct[k] = g != 0 ? g / len[s] * 100 : g;
        }
        sum_gc = sum_gc / sum_base * 100;
        fprintf(out, "%s\t%d\t%u\t%lu\t%f\t%f", names[t].c_str(), (int)tlen[t], sum_count, sum_base,
                (double)sum_base / tlen[t], sum_gc);
        for (uint64_t k = 0; k < slots; ++k) fprintf(out, "\t%d\t%u\t%f", (int)(k + 1), bins[win_off[t] + k], pct[k]);
        fprintf(out, "\n");
    }
}

// ---- shared CLI helpers ----------------------------------------------------------------------

inline void die_hpn(hpn_ctx *ctx, int rc, const char *what)
{
    fprintf(stderr, "%s: %s (%s)\n", what, hpn_strerror(rc), ctx ? hpn_ctx_last_error(ctx) : "");
    leave(2);
}

}  // namespace hpn
