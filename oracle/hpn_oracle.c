/*
 * hpn_oracle.c -- CPU restatement of the HighPerformanceNGS scan loops.
 *
 * TEST INFRASTRUCTURE ONLY (see hpn_oracle.h).  Written from the behaviour of
 * the reference tools, not from their text; each function names the reference
 * lines it restates.  Pinned against the compiled reference by
 * tests/test_oracle_golden.py (tests/golden/).
 */
#define _GNU_SOURCE
#include "hpn_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <zlib.h>

/* ------------------------------------------------------------------------- */
/* accumulators                                                              */
/* ------------------------------------------------------------------------- */

orc_counts *orc_counts_new(void)
{
    orc_counts *c = (orc_counts *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    for (int q = 0; q < ORC_QUAL_ROWS; ++q) {
        c->quality[q] = (uint64_t *)calloc(ORC_LEN_BINS, sizeof(uint64_t));
        if (!c->quality[q]) {
            orc_counts_free(c);
            return NULL;
        }
    }
    return c;
}

void orc_counts_free(orc_counts *c)
{
    if (!c) return;
    for (int q = 0; q < ORC_QUAL_ROWS; ++q) free(c->quality[q]);
    free(c);
}

void orc_counts_add(orc_counts *dst, const orc_counts *src)
{
    for (int l = 0; l < ORC_LEN_BINS; ++l) dst->seqlen[l] += src->seqlen[l];
    for (int q = 0; q < ORC_QUAL_ROWS; ++q)
        for (int p = 0; p < ORC_LEN_BINS; ++p) dst->quality[q][p] += src->quality[q][p];
}

void orc_counts_flat_quality(const orc_counts *c, uint64_t *out)
{
    for (int q = 0; q < ORC_QUAL_ROWS; ++q)
        memcpy(out + (size_t)q * ORC_LEN_BINS, c->quality[q], ORC_LEN_BINS * sizeof(uint64_t));
}

/* ------------------------------------------------------------------------- */
/* fastq_count                                                               */
/* ------------------------------------------------------------------------- */

/* One record's contribution (fastq_count.c:114-118, macro :29-35): length
 * histogram bump, then Quality[byte][cycle]++ for the first `len` bytes of the
 * quality line.  The reference indexes without bounds checks; where it would
 * leave its arrays (len >= 512, quality byte >= 128) the oracle reports
 * ORC_E_DOMAIN instead of reproducing undefined behaviour. */
static inline int tally_one(orc_counts *c, const uint8_t *q, uint32_t len)
{
    if (len >= ORC_LEN_BINS) return ORC_E_DOMAIN;
    c->seqlen[len]++;
    for (uint32_t i = 0; i < len; ++i) {
        uint8_t v = q[i];
        if (v >= ORC_QUAL_ROWS) return ORC_E_DOMAIN;
        c->quality[v][i]++;
    }
    return ORC_OK;
}

/* open() + gzdopen(fd,"rb") like open_input_stream (IO_stream.h:122-136);
 * "-" prefix or empty name = stdin.  The reference passes O_CREAT; the oracle
 * does not create missing inputs. */
static gzFile open_in(const char *path)
{
    if (path[0] == '-' || path[0] == '\0') return gzdopen(0, "rb");
    return gzopen(path, "rb");
}

int orc_count_stream(const char *path, orc_counts *c)
{
    gzFile fq = open_in(path);
    if (!fq) return ORC_E_IO;
    /* One 1024-byte line buffer reused for all four lines of every record
     * (fastq_count.c:107,112-118).  It is never cleared, so a short or
     * missing line leaves older bytes visible exactly as in the reference. */
    char *buf = (char *)calloc(ORC_LINE_BUF, 1);
    int rc = ORC_OK;
    while (gzgets(fq, buf, ORC_LINE_BUF) != NULL) { /* name line, ignored */
        gzgets(fq, buf, ORC_LINE_BUF);              /* sequence line */
        uint16_t len = (uint16_t)(strlen(buf) - 1); /* :114, uint16 truncation */
        gzgets(fq, buf, ORC_LINE_BUF);              /* '+' line, ignored */
        gzgets(fq, buf, ORC_LINE_BUF);              /* quality line */
        rc = tally_one(c, (const uint8_t *)buf, len);
        if (rc != ORC_OK) break;
    }
    free(buf);
    gzclose(fq);
    return rc;
}

int orc_count_soa(const uint8_t *qual, const uint64_t *off, uint64_t n, orc_counts *c)
{
    for (uint64_t r = 0; r < n; ++r) {
        uint64_t len = off[r + 1] - off[r];
        if (len >= ORC_LEN_BINS) return ORC_E_DOMAIN;
        int rc = tally_one(c, qual + off[r], (uint32_t)len);
        if (rc != ORC_OK) return rc;
    }
    return ORC_OK;
}

void orc_summarise(const orc_counts *c, orc_summary *s)
{
    memset(s, 0, sizeof(*s));
    /* statSeqLen (fastq_count.c:63-74): min is "first non-zero index while min
     * is still 0", so a non-empty bin 0 never becomes the minimum. */
    for (uint32_t l = 0; l < ORC_LEN_BINS; ++l) {
        uint64_t h = c->seqlen[l];
        if (!h) continue;
        s->reads += h;
        s->bases += 1.0 * (double)h * l;
        if (!s->min_len) s->min_len = l;
        if (s->max_len < l) s->max_len = l;
    }
    /* statQ(Quality,128,512,sum,53,sumQ1,63,sumQ2) (fastq_count.c:37-47,124) */
    for (uint32_t q = 0; q < ORC_QUAL_ROWS; ++q)
        for (uint32_t p = 0; p < ORC_LEN_BINS; ++p) {
            uint64_t v = c->quality[q][p];
            s->sum += v;
            if (q >= 53) s->q20 += v;
            if (q >= 63) s->q30 += v;
        }
}

int orc_fmt_count_header(char *dst, size_t cap)
{
    return snprintf(dst, cap, "#Filename\tReadCount\tBaseCount\tMeanLen\tMinLen\tMaxLen\tQ20(%%)\tQ30(%%)\n");
}

int orc_fmt_count_row(char *dst, size_t cap, const char *name, const orc_summary *s)
{
    double mean = s->bases / (double)s->reads; /* 0/0 -> -nan on empty input */
    return snprintf(dst, cap, "%s\t%lu\t%.0f\t%.0f\t%u\t%u\t%.3f\t%.3f\n", name,
                    (unsigned long)s->reads, s->bases, mean, s->min_len, s->max_len,
                    1.0 * s->q20 / s->sum * 100, 1.0 * s->q30 / s->sum * 100);
}

int orc_fmt_kthread_file_row(char *dst, size_t cap, const char *name, const orc_summary *s)
{
    /* per-file read count is a uint32_t (fastq_count_kthread.c:106,258) and the
     * mean divides by that truncated value (:137). */
    uint32_t rc32 = (uint32_t)s->reads;
    double mean = s->bases / rc32;
    return snprintf(dst, cap, "%s\t%u\t%.0f\t%.0f\t%u\t%u\t%.3f\t%.3f\n", name, rc32, s->bases,
                    mean, s->min_len, s->max_len, 1.0 * s->q20 / s->sum * 100,
                    1.0 * s->q30 / s->sum * 100);
}

int orc_fmt_len_detail(char *dst, size_t cap, const orc_counts *c, uint32_t min_len,
                       uint32_t max_len)
{
    size_t o = 0;
#define EMIT(...)                                                              \
    do {                                                                       \
        int k_ = snprintf(dst ? dst + (o < cap ? o : cap) : NULL,              \
                          dst && o < cap ? cap - o : 0, __VA_ARGS__);          \
        o += (size_t)k_;                                                       \
    } while (0)
    EMIT("#Len:");
    for (uint32_t l = min_len; l <= max_len; ++l) EMIT("\t%u", l);
    EMIT("\n#Freq:");
    for (uint32_t l = min_len; l <= max_len; ++l) EMIT("\t%lu", (unsigned long)c->seqlen[l]);
    EMIT("\n");
    return (int)o;
}

int orc_fmt_quality_matrix(char *dst, size_t cap, const orc_counts *c, uint32_t ncol)
{
    size_t o = 0;
    /* printQ(Quality,128,maxLen) (fastq_count_kthread.c:52-64): with ncol == 0
     * the loops print nothing. */
    for (uint32_t q = 0; q < ORC_QUAL_ROWS; ++q)
        for (uint32_t p = 0; p < ncol; ++p)
            EMIT(p == ncol - 1 ? "%lu\n" : "%lu\t", (unsigned long)c->quality[q][p]);
    return (int)o;
#undef EMIT
}

void orc_reduce_stats(const orc_counts *const *per_file, const orc_summary *per_file_sum, int n,
                      orc_counts *merged_counts, orc_merged *m)
{
    memset(m, 0, sizeof(*m));
    m->min_len = 10000; /* fastq_count_kthread.c:182 */
    for (int i = 0; i < n; ++i) {
        m->reads_u32 += (uint32_t)per_file_sum[i].reads; /* uint32 accumulate, :186 */
        m->bases += per_file_sum[i].bases;
        if (per_file_sum[i].min_len < m->min_len) m->min_len = per_file_sum[i].min_len;
        if (per_file_sum[i].max_len > m->max_len) m->max_len = per_file_sum[i].max_len;
        orc_counts_add(merged_counts, per_file[i]);
    }
    orc_summary s;
    orc_summarise(merged_counts, &s);
    m->sum = s.sum;
    m->q20 = s.q20;
    m->q30 = s.q30;
}

int orc_fmt_kthread_merged_header(char *dst, size_t cap)
{
    return snprintf(dst, cap, "#ReadCount\tBaseCount\tMeanLen\tMinLen\tMaxLen\tQ20(%%)\tQ30(%%)\n");
}

int orc_fmt_kthread_merged_row(char *dst, size_t cap, const orc_merged *m)
{
    return snprintf(dst, cap, "%u\t%.0f\t%.0f\t%u\t%u\t%.3f\t%.3f\n", m->reads_u32, m->bases,
                    m->bases / m->reads_u32, m->min_len, m->max_len, 1.0 * m->q20 / m->sum * 100,
                    1.0 * m->q30 / m->sum * 100);
}

/* ---- threaded baseline (kt_for scheduling, klib/kthread.c:24-60) -------- */

typedef struct pf_shared {
    int n_threads;
    long n;
    long *next; /* per-worker cursor, stride n_threads */
    const char *const *paths;
    orc_counts **per_file;
    int *rc;
} pf_shared;

typedef struct pf_worker {
    pf_shared *sh;
    int tid;
} pf_worker;

static void *pf_main(void *arg)
{
    pf_worker *w = (pf_worker *)arg;
    pf_shared *sh = w->sh;
    for (;;) { /* own stripe first: tid, tid+T, ... */
        long i = __sync_fetch_and_add(&sh->next[w->tid], sh->n_threads);
        if (i >= sh->n) break;
        sh->rc[i] = orc_count_stream(sh->paths[i], sh->per_file[i]);
    }
    for (;;) { /* then steal from the worker that is furthest behind */
        int victim = -1;
        long lo = 0x7fffffffffffffffL;
        for (int t = 0; t < sh->n_threads; ++t)
            if (sh->next[t] < lo) lo = sh->next[t], victim = t;
        long i = __sync_fetch_and_add(&sh->next[victim], sh->n_threads);
        if (i >= sh->n) break;
        sh->rc[i] = orc_count_stream(sh->paths[i], sh->per_file[i]);
    }
    return NULL;
}

static double now_s(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

int orc_count_files_threaded(const char *const *paths, int n_files, int n_threads,
                             orc_counts *merged, double *seconds)
{
    if (n_files <= 0 || n_threads <= 0) return ORC_E_ARG;
    if (n_threads > n_files) n_threads = n_files; /* fastq_count_kthread.c:245 */
    pf_shared sh;
    sh.n_threads = n_threads;
    sh.n = n_files;
    sh.paths = paths;
    sh.next = (long *)calloc((size_t)n_threads, sizeof(long));
    sh.per_file = (orc_counts **)calloc((size_t)n_files, sizeof(orc_counts *));
    sh.rc = (int *)calloc((size_t)n_files, sizeof(int));
    pf_worker *ws = (pf_worker *)calloc((size_t)n_threads, sizeof(pf_worker));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    double t0 = now_s();
    for (int i = 0; i < n_files; ++i) sh.per_file[i] = orc_counts_new(); /* :255-269 */
    for (int t = 0; t < n_threads; ++t) {
        sh.next[t] = t;
        ws[t].sh = &sh;
        ws[t].tid = t;
    }
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, pf_main, &ws[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    int rc = ORC_OK;
    for (int i = 0; i < n_files; ++i) { /* reduceStats, element-wise sums */
        if (sh.rc[i] != ORC_OK) rc = sh.rc[i];
        orc_counts_add(merged, sh.per_file[i]);
    }
    if (seconds) *seconds = now_s() - t0;
    for (int i = 0; i < n_files; ++i) orc_counts_free(sh.per_file[i]);
    free(sh.next);
    free(sh.per_file);
    free(sh.rc);
    free(ws);
    free(th);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* fastq_trim                                                                */
/* ------------------------------------------------------------------------- */

/* Replace the last character of the C string in buf by NUL (fastq_trim.c:71,75,82). */
static inline void chop_last(char *buf)
{
    buf[strlen(buf) - 1] = '\0';
}

/* strncpy(dst, buf+S, E-S) into a zeroed E-S+1 buffer (fastq_trim.c:76-77):
 * copies until NUL or E-S chars, whichever first. */
static char *cut(const char *buf, int S, int E)
{
    char *d = (char *)calloc((size_t)(E - S + 1), 1);
    strncpy(d, buf + S, (size_t)(E - S));
    return d;
}

/* ---- Rgzfastq_uniq.c (PARITY UNPINNED: restated from the source text, R is not in the image) ---- */

static int rqc_ntval(unsigned char b) /* initNtVal, Rgzfastq_uniq.c:97-108 */
{
    switch (b) {
    case 'c': case 'C': return 1;
    case 'a': case 'A': return 2;
    case 'g': case 'G': return 3;
    case '.': case 'N': return 4;
    default: return 0; /* t T u U and every other byte */
    }
}

static int rqc_one(const uint8_t *seq, uint64_t ls, const uint8_t *qual, uint64_t lq, int32_t *quality,
                   int32_t *nucleotide, int32_t *length, double *gc)
{
    if (ls < 1 || ls > ORC_RQC_MAXLEN || lq > ORC_RQC_MAXLEN) return ORC_E_DOMAIN;
    double GC = 0; /* STATSEQ :50-57 */
    for (uint64_t L = 0; L < ls; ++L) {
        if (seq[L] == 'G' || seq[L] == 'C') GC++;
        if (nucleotide) nucleotide[5 * L + rqc_ntval(seq[L])]++;
    }
    GC /= (double)ls;
    if (gc) *gc = GC;
    for (uint64_t i = 0; i < lq; ++i) { /* AssignQuality :42-48 */
        if (qual[i] >= 128) return ORC_E_DOMAIN;
        if (quality) quality[qual[i] + 128 * i]++;
    }
    if (length) length[ls - 1]++; /* :174 */
    return ORC_OK;
}

int orc_rqc_soa(const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n, int32_t *quality,
                int32_t *nucleotide, int32_t *length, double *gc)
{
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t len = off[r + 1] - off[r];
        int rc = rqc_one(seq + off[r], len, qual + off[r], len, quality, nucleotide, length, gc ? gc + r : NULL);
        if (rc != ORC_OK) return rc;
    }
    return ORC_OK;
}

int orc_rqc_stream(const char *path, int32_t *quality, int32_t *nucleotide, int32_t *length, double *gc,
                   uint64_t gc_cap, uint64_t *n_reads)
{
    gzFile fq = open_in(path);
    if (!fq) return ORC_E_IO;
    char *buf = (char *)malloc(ORC_LINE_BUF);
    uint8_t seq[ORC_LINE_BUF];
    uint64_t n = 0;
    int rc = ORC_OK;
    for (;;) { /* readNextNode :118-137 */
        char *p = gzgets(fq, buf, ORC_LINE_BUF);
        if (gzeof(fq) || !p) break;
        gzgets(fq, buf, ORC_LINE_BUF);
        chop_last(buf);
        const size_t ls = strlen(buf);
        memcpy(seq, buf, ls + 1);
        gzgets(fq, buf, ORC_LINE_BUF);
        gzgets(fq, buf, ORC_LINE_BUF);
        chop_last(buf);
        rc = rqc_one(seq, ls, (const uint8_t *)buf, strlen(buf), quality, nucleotide, length,
                     gc && n < gc_cap ? gc + n : NULL);
        if (rc != ORC_OK) break;
        ++n;
    }
    free(buf);
    gzclose(fq);
    if (n_reads) *n_reads = n;
    return rc;
}

int orc_trim_stream(const char *path, int S, int E, FILE *out, uint64_t *n_reads)
{
    if (S < 0 || E < S || S >= ORC_LINE_BUF) return ORC_E_DOMAIN;
    gzFile fq = open_in(path);
    if (!fq) return ORC_E_IO;
    char *buf = (char *)malloc(ORC_LINE_BUF);
    uint64_t n = 0;
    for (;;) {
        memset(buf, 0, ORC_LINE_BUF);                 /* fastq_trim.c:97 */
        char *p = gzgets(fq, buf, ORC_LINE_BUF);      /* name line */
        if (gzeof(fq)) break;                         /* :69-70: EOF is tested after the read */
        if (!p) break;                                /* read error: the reference would crash */
        chop_last(buf);
        char *name = strdup(buf);
        gzgets(fq, buf, ORC_LINE_BUF);                /* sequence */
        chop_last(buf);
        char *seq = cut(buf, S, E);
        gzgets(fq, buf, ORC_LINE_BUF);                /* '+' line, dropped */
        gzgets(fq, buf, ORC_LINE_BUF);                /* quality */
        chop_last(buf);
        char *qual = cut(buf, S, E);
        ++n;
        fprintf(out, "%s\n%s\n+\n%s\n", name, seq, qual); /* :101 */
        free(name);
        free(seq);
        free(qual);
    }
    free(buf);
    gzclose(fq);
    if (n_reads) *n_reads = n;
    return ORC_OK;
}

int orc_trim_soa(const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n, int S,
                 int E, uint8_t *out_seq, uint8_t *out_qual, uint64_t *out_off)
{
    if (S < 0 || E < S) return ORC_E_DOMAIN;
    uint64_t o = 0;
    out_off[0] = 0;
    for (uint64_t r = 0; r < n; ++r) {
        uint64_t len = off[r + 1] - off[r];
        uint64_t b = (uint64_t)S < len ? (uint64_t)S : len;
        uint64_t e = (uint64_t)E < len ? (uint64_t)E : len;
        memcpy(out_seq + o, seq + off[r] + b, e - b);
        memcpy(out_qual + o, qual + off[r] + b, e - b);
        o += e - b;
        out_off[r + 1] = o;
    }
    return ORC_OK;
}

/* extension: quality-threshold trim points (no reference counterpart) */
void orc_qtrim_points(const uint8_t *qual, const uint64_t *off, uint64_t n, uint32_t threshold,
                      uint32_t *beg, uint32_t *end)
{
    for (uint64_t r = 0; r < n; ++r) {
        const uint8_t *q = qual + off[r];
        uint64_t len = off[r + 1] - off[r], b = 0, e = 0;
        int seen = 0;
        for (uint64_t i = 0; i < len; ++i)
            if (q[i] >= threshold) {
                if (!seen) b = i, seen = 1;
                e = i + 1;
            }
        beg[r] = (uint32_t)b;
        end[r] = (uint32_t)e;
    }
}

int orc_trim_points_soa(const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n,
                        const uint32_t *beg, const uint32_t *end, uint8_t *out_seq, uint8_t *out_qual,
                        uint64_t *out_off)
{
    uint64_t o = 0;
    out_off[0] = 0;
    for (uint64_t r = 0; r < n; ++r) {
        uint64_t len = off[r + 1] - off[r];
        uint64_t b = beg[r] < len ? beg[r] : len, e = end[r] < len ? end[r] : len;
        if (e < b) e = b;
        memcpy(out_seq + o, seq + off[r] + b, e - b);
        memcpy(out_qual + o, qual + off[r] + b, e - b);
        o += e - b;
        out_off[r + 1] = o;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* bam2depth: dense model of fetch_func + hash2BedGraph + overlap            */
/* ------------------------------------------------------------------------- */

int orc_depth_target(const int32_t *rec_tid, const int32_t *rec_pos, const uint32_t *rec_flag,
                     const uint32_t *cigar_off, const uint32_t *cigar, uint64_t n, int32_t tid,
                     uint32_t target_len, uint32_t W, uint32_t flag_mask, orc_run **runs_out,
                     uint64_t *n_runs_out, double *bins)
{
    if (W == 0) return ORC_E_ARG;
    /* Breakpoints may lie past target_len (reads overhanging the contig end are
     * not clipped); size the difference array to the furthest one. */
    uint64_t hi = target_len, hi_bp = 0;
    for (uint64_t r = 0; r < n; ++r) {
        if (rec_tid[r] != tid || rec_tid[r] < 0 || (rec_flag[r] & flag_mask)) continue;
        uint64_t p = (uint32_t)rec_pos[r];
        for (uint32_t k = cigar_off[r]; k < cigar_off[r + 1]; ++k) {
            uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
            if (op == 2 || op == 3) p += len;
            else if (op == 0) {
                p += len;
                if (p > hi_bp) hi_bp = p;      /* the breakpoint one past an M block */
            }
        }
    }
    /* int2char keeps 28 bits of a position (hashtbl.c:243-249): a BREAKPOINT at 2^28 or beyond
     * aliases another key in the reference; outside the restated domain.  (A longer contig whose
     * reads all end below 2^28 is fine: only breakpoints become keys.) */
    if (hi_bp >= (1ull << 28)) return ORC_E_DOMAIN;
    if (hi_bp > hi) hi = hi_bp;
    int32_t *diff = (int32_t *)calloc(hi + 2, sizeof(int32_t));
    if (!diff) return ORC_E_NOMEM;
    for (uint64_t r = 0; r < n; ++r) {
        /* bam2depth.c:90: BAM_DEF_MASK or tid<0 -> skipped. */
        if (rec_tid[r] != tid || rec_tid[r] < 0 || (rec_flag[r] & flag_mask)) continue;
        uint32_t p = (uint32_t)rec_pos[r]; /* unsigned int temp_start = c->pos (:93) */
        for (uint32_t k = cigar_off[r]; k < cigar_off[r + 1]; ++k) {
            uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
            if (op == 1) continue;             /* I: nothing */
            if (op == 2 || op == 3) p += len;  /* D, N: advance only */
            else if (op == 0) {                /* M: +1 at start, -1 at end */
                diff[p] += 1;
                p += len;
                diff[p] -= 1;
            }                                  /* S,H,P,=,X: ignored, no advance */
        }
    }
    /* prefix sum -> maximal runs of equal depth > 0 (hash2BedGraph :203-236);
     * window sums clipped at target_len (overlap :132-176). */
    uint32_t windows = target_len / W + 1;
    for (uint32_t k = 0; k < windows; ++k) bins[k] = 0.0;
    uint64_t cap = 1024, nr = 0;
    orc_run *runs = (orc_run *)malloc(cap * sizeof(orc_run));
    int32_t cov = 0, cur = 0;
    int64_t start = 0;
    for (uint64_t i = 0; i <= hi; ++i) {
        cov += diff[i];
        if (cov != cur) {
            if (cur > 0) {
                if (nr == cap) runs = (orc_run *)realloc(runs, (cap *= 2) * sizeof(orc_run));
                runs[nr].start = (int32_t)start;
                runs[nr].end = (int32_t)i;
                runs[nr].depth = cur;
                ++nr;
            }
            cur = cov;
            start = (int64_t)i;
        }
        if (i < target_len && cov > 0) bins[i / W] += (double)cov;
    }
    free(diff);
    *runs_out = runs;
    *n_runs_out = nr;
    return ORC_OK;
}

int orc_fmt_bedgraph(FILE *out, const char *chr, const orc_run *runs, uint64_t n_runs)
{
    for (uint64_t i = 0; i < n_runs; ++i)
        fprintf(out, "%s\t%d\t%d\t%d\n", chr, runs[i].start, runs[i].end, runs[i].depth);
    return ORC_OK;
}

int orc_fmt_depth_bins(FILE *out, const char *chr, uint32_t target_len, uint32_t W,
                       const double *bins)
{
    uint32_t windows = target_len / W + 1; /* bam2depth.c:326 */
    for (uint32_t k = 0; k < windows; ++k) {
        /* int arithmetic as in output_bins (:241-242) */
        int ws = (int)(W * k);
        int we = (W * (k + 1) > target_len) ? (int)target_len : (int)(W * (k + 1));
        fprintf(out, "%s\t%d\t%d\t%.2f\n", chr, ws, we, bins[k] / W);
    }
    return ORC_OK;
}

int orc_fmt_wig_bins(FILE *out, const char *chr, uint32_t target_len, uint32_t W,
                     const double *bins)
{
    uint32_t windows = target_len / W + 1;
    fprintf(out, "variableStep chrom=%s span=%d\n", chr, (int)W);
    for (uint32_t k = 0; k < windows; ++k)
        if (bins[k]) fprintf(out, "%d\t%.2f\n", (int)(W * k), bins[k] / W);
    return ORC_OK;
}

/* bam2wig.c:131-175, called once per emitted run in ascending order (:213-231). */
void orc_wig_bins(const orc_run *runs, uint64_t n_runs, uint32_t target_len, uint32_t W, double *bins)
{
    const int windows = (int)(target_len / W + 1);
    for (int k = 0; k <= windows; ++k) bins[k] = 0.0;
    int j = 0, touched = 0; /* j, subject_count of hash2BedGraph */
    for (uint64_t i = 0; i < n_runs; ++i) {
        const uint32_t s = (uint32_t)runs[i].start, e = (uint32_t)runs[i].end;
        const double d = (double)runs[i].depth;
        if (touched > 1) j = (j - touched >= 0) ? j - touched : 0; /* step back over the windows of the previous run */
        touched = 0;
        while (j <= windows) {
            const uint32_t ws = W * (uint32_t)j;
            uint32_t we = ((uint32_t)j + 1) * W - 1; /* inclusive end */
            if (we > target_len) we = target_len;
            if (e < ws) break;
            if (s < ws) {
                if (e < we) {
                    bins[j] += (e - ws) * d;
                    touched++;
                    break;
                }
                bins[j++] += (we - ws + 1) * d;
                touched++;
            } else if (s <= we) {
                if (e <= we) {
                    bins[j] += (e - s) * d;
                    touched++;
                    break;
                }
                bins[j++] += (we - s) * d;
                touched++;
            } else {
                j++;
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* bam_sliding_count                                                         */
/* ------------------------------------------------------------------------- */

/* cal_GC (bam_sliding_count.c:84-91): count 4-bit codes 2 (C) and 4 (G);
 * base i is the high nibble of byte i/2 for even i (bam1_seqi, bam.h:260).
 * The count lives in an unsigned short. */
static inline uint16_t gc_nibbles(const uint8_t *seq4, int32_t l_qseq)
{
    uint16_t gc = 0;
    for (int32_t i = 0; i < l_qseq; ++i) {
        int code = (seq4[i >> 1] >> ((~i & 1) << 2)) & 0xf;
        if (code == 2 || code == 4) gc++;
    }
    return gc;
}

int orc_window_add(const int32_t *rec_tid, const int32_t *rec_pos, const uint32_t *rec_flag,
                   const int32_t *l_qseq, const uint64_t *seq_off, const uint8_t *seq4, uint64_t n,
                   uint32_t W, int32_t n_targets, const uint64_t *win_off, uint32_t *bins,
                   uint64_t *gc, uint32_t *len, uint8_t *touched, uint64_t *n_count)
{
    if (W == 0) return ORC_E_ARG;
    for (uint64_t r = 0; r < n; ++r) {
        int32_t t = rec_tid[r];
        if (t < 0) continue;              /* :96 */
        if (rec_flag[r] & 4) continue;    /* :97 BAM_FUNMAP */
        if (t >= n_targets) return ORC_E_DOMAIN;
        if (touched) touched[t] = 1;      /* lazily allocated arrays :98-103 */
        if (n_count) (*n_count)++;
        /* c->pos / window in int, then (unsigned short) (:117) */
        uint16_t w = (uint16_t)(rec_pos[r] / (int32_t)W);
        uint64_t slots = win_off[t + 1] - win_off[t];
        if (w >= slots) return ORC_E_DOMAIN; /* the reference would write out of bounds */
        uint64_t s = win_off[t] + w;
        bins[s] += 1;
        gc[s] += gc_nibbles(seq4 + seq_off[r], l_qseq[r]);
        len[s] += (uint32_t)l_qseq[r];
    }
    return ORC_OK;
}

int orc_window_gc_f32(const int32_t *rec_tid, const int32_t *rec_pos, const uint32_t *rec_flag,
                      const int32_t *l_qseq, const uint64_t *seq_off, const uint8_t *seq4,
                      uint64_t n, uint32_t W, int32_t n_targets, const uint64_t *win_off,
                      float *gc_f32)
{
    if (W == 0) return ORC_E_ARG;
    for (uint64_t r = 0; r < n; ++r) {
        int32_t t = rec_tid[r];
        if (t < 0 || (rec_flag[r] & 4)) continue;
        if (t >= n_targets) return ORC_E_DOMAIN;
        uint16_t w = (uint16_t)(rec_pos[r] / (int32_t)W);
        if (w >= win_off[t + 1] - win_off[t]) return ORC_E_DOMAIN;
        gc_f32[win_off[t] + w] += gc_nibbles(seq4 + seq_off[r], l_qseq[r]); /* float += ushort, :121 */
    }
    return ORC_OK;
}

int orc_fmt_window_report(FILE *out, int32_t n_targets, const char *const *names,
                          const uint32_t *target_len, uint32_t W, const uint64_t *win_off,
                          const uint32_t *bins, const float *gc_in, const uint32_t *len,
                          const uint8_t *touched)
{
    /* header (bam_sliding_count.c:148-153) */
    fprintf(out, "#chr\tchr_len\tchr_sum_read_count\tchr_sum_base\tchr_mean_cov\tchr_mean_GC%%");
    uint32_t max_len = 0;
    for (int32_t t = 0; t < n_targets; ++t)
        if (target_len[t] > max_len) max_len = target_len[t];
    uint32_t max_windows = max_len / W + 1;
    for (uint32_t k = 0; k < max_windows; ++k) fprintf(out, "\t%u\tcount\tGC%%", k + 1);
    fprintf(out, "\n");
    for (int32_t t = 0; t < n_targets; ++t) {
        if (!touched[t]) continue; /* windows[j]==0 (:155) */
        uint64_t slots = win_off[t + 1] - win_off[t];
        /* calc_winGC (:126-138): float32 running sums in window order. */
        unsigned int sum_count = 0;
        float sum_gc = 0.0f;
        unsigned long sum_base = 0;
        float *gcp = (float *)malloc(slots * sizeof(float));
        for (uint64_t k = 0; k < slots; ++k) {
            uint64_t s = win_off[t] + k;
            sum_count += bins[s];
            sum_gc += gc_in[s];
            sum_base += len[s];
            gcp[k] = gc_in[s] != 0 ? gc_in[s] / len[s] * 100 : gc_in[s];
        }
        sum_gc = sum_gc / sum_base * 100;
        fprintf(out, "%s\t%d\t%u\t%lu\t%f\t%f", names[t], (int)target_len[t], sum_count, sum_base,
                (double)sum_base / target_len[t], sum_gc);
        for (uint64_t k = 0; k < slots; ++k)
            fprintf(out, "\t%d\t%u\t%f", (int)(k + 1), bins[win_off[t] + k], gcp[k]);
        fprintf(out, "\n");
        free(gcp);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* synthetic input                                                           */
/* ------------------------------------------------------------------------- */

#define ORC_GOLD 0x9E3779B97F4A7C15ull
#define ORC_STEP 0xD1B54A32D192ED03ull

uint64_t orc_mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

static inline uint64_t rec_key(uint64_t seed, uint64_t rec)
{
    return orc_mix64(seed + rec * ORC_GOLD);
}

uint32_t orc_synth_len(uint64_t seed, uint64_t rec, uint32_t len_lo, uint32_t len_hi)
{
    if (len_hi <= len_lo) return len_lo;
    return len_lo + (uint32_t)(orc_mix64(rec_key(seed, rec) ^ 0xA5A5A5A5A5A5A5A5ull) %
                               (uint64_t)(len_hi - len_lo + 1));
}

void orc_synth_record(uint64_t seed, uint64_t rec, uint32_t len, uint8_t *seq, uint8_t *qual)
{
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    uint64_t key = rec_key(seed, rec);
    for (uint32_t k0 = 0; k0 < len; k0 += 4) { /* one 64-bit word feeds 4 bytes */
        uint64_t j = k0 >> 2;
        uint64_t wq = qual ? orc_mix64(key + (2 * j + 1) * ORC_STEP) : 0;
        uint64_t wb = seq ? orc_mix64(key + (2 * j + 2) * ORC_STEP) : 0;
        for (uint32_t k = k0; k < k0 + 4 && k < len; ++k) {
            uint32_t sh = 16 * (k & 3);
            if (qual) {
                uint32_t u = (uint32_t)(wq >> sh) & 0xffff;
                qual[k] = (uint8_t)(35 + ((u * 40u) >> 16)); /* Phred 2..41, +33 */
            }
            if (seq) {
                uint32_t u = (uint32_t)(wb >> sh) & 0xffff;
                uint32_t x = (u * 10000u) >> 16; /* 0..9999 */
                seq[k] = x < 100 ? 'N' : (uint8_t)ACGT[(x - 100) / 2475];
            }
        }
    }
}

int orc_synth_soa(uint64_t seed, uint64_t first, uint64_t n, uint32_t len_lo, uint32_t len_hi,
                  uint8_t *seq, uint8_t *qual, uint64_t *off)
{
    uint64_t o = 0;
    off[0] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t len = orc_synth_len(seed, first + i, len_lo, len_hi);
        orc_synth_record(seed, first + i, len, seq ? seq + o : NULL, qual ? qual + o : NULL);
        o += len;
        off[i + 1] = o;
    }
    return ORC_OK;
}

int orc_synth_write_fastq(const char *path, uint64_t seed, uint64_t first, uint64_t n,
                          uint32_t len_lo, uint32_t len_hi, int gz_members)
{
    if (len_hi >= ORC_LEN_BINS) return ORC_E_ARG;
    uint8_t seq[ORC_LEN_BINS], qual[ORC_LEN_BINS];
    char line[2 * ORC_LEN_BINS + 64];
    if (gz_members <= 0) {
        FILE *f = fopen(path, "wb");
        if (!f) return ORC_E_IO;
        for (uint64_t r = first; r < first + n; ++r) {
            uint32_t len = orc_synth_len(seed, r, len_lo, len_hi);
            orc_synth_record(seed, r, len, seq, qual);
            int k = snprintf(line, sizeof line, "@r%lu\n%.*s\n+\n%.*s\n", (unsigned long)r,
                             (int)len, (const char *)seq, (int)len, (const char *)qual);
            fwrite(line, 1, (size_t)k, f);
        }
        fclose(f);
        return ORC_OK;
    }
    /* multi-member gzip: members are written by appending independent gzip
     * streams; zlib's gzread (hence the reference) reads them as one. */
    uint64_t per = (n + (uint64_t)gz_members - 1) / (uint64_t)gz_members;
    if (per == 0) per = 1;
    uint64_t r = first, end = first + n;
    int first_member = 1;
    while (r < end || first_member) {
        gzFile g = gzopen(path, first_member ? "wb1" : "ab1");
        if (!g) return ORC_E_IO;
        uint64_t stop = r + per < end ? r + per : end;
        for (; r < stop; ++r) {
            uint32_t len = orc_synth_len(seed, r, len_lo, len_hi);
            orc_synth_record(seed, r, len, seq, qual);
            int k = snprintf(line, sizeof line, "@r%lu\n%.*s\n+\n%.*s\n", (unsigned long)r,
                             (int)len, (const char *)seq, (int)len, (const char *)qual);
            gzwrite(g, line, (unsigned)k);
        }
        gzclose(g);
        first_member = 0;
    }
    return ORC_OK;
}
