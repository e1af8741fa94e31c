/* seam3_fetch_func.c -- the INTEGRATION.md "Seam 3" stub as a C99 program.
 *
 * fetch_func() has the shape of the reference's bam_fetch_f callback (bam.h:627; bam2depth.c:86): it is called once
 * per record and only appends the record to a structure-of-arrays batch; the hash tables, the key sort and the
 * sweep of bam2depth.c:203-236 are three library calls per target.  The records come from a small BAM walk in plain C
 * (BGZF is multi-member gzip, so zlib's gzread delivers the uncompressed stream; bam_read1's layout, bam.c:191).
 * Output: the bedGraph lines (bam2depth.c:217 format) of every target, then "#win <name> <k> <sum>" lines.
 * tests/test_abi_c.py compiles this with  gcc -std=c99 -pedantic -Wall -Werror  and compares the bedGraph bytes with the
 * reference's (tests/golden/expected/depth_a3, depth_rand).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <zlib.h>

#include "hpngs.h"

typedef struct {
    int32_t *tid, *pos;
    uint32_t *flag, *cigar_off, *cigar;
    uint64_t n, cap, ccap;
} Batch;

typedef struct {                 /* what the callback sees of a record (bam1_t's core + CIGAR) */
    int32_t tid, pos;
    uint32_t flag, n_cigar;
    const uint32_t *cigar;
} Rec;

static int fetch_func(const Rec *b, void *data)
{
    Batch *B = (Batch *)data;
    if (B->n == B->cap) {
        B->cap = B->cap ? 2 * B->cap : 1024;
        B->tid = (int32_t *)realloc(B->tid, B->cap * 4), B->pos = (int32_t *)realloc(B->pos, B->cap * 4);
        B->flag = (uint32_t *)realloc(B->flag, B->cap * 4);
        B->cigar_off = (uint32_t *)realloc(B->cigar_off, (B->cap + 1) * 4);
        if (B->n == 0) B->cigar_off[0] = 0;
    }
    if (B->cigar_off[B->n] + b->n_cigar > B->ccap) {
        B->ccap = 2 * (B->cigar_off[B->n] + b->n_cigar) + 1024;
        B->cigar = (uint32_t *)realloc(B->cigar, B->ccap * 4);
    }
    B->tid[B->n] = b->tid, B->pos[B->n] = b->pos, B->flag[B->n] = b->flag;
    memcpy(B->cigar + B->cigar_off[B->n], b->cigar, 4 * (size_t)b->n_cigar);
    B->cigar_off[B->n + 1] = B->cigar_off[B->n] + b->n_cigar;
    ++B->n;
    return 0;
}

static int32_t rd32(gzFile f)
{
    int32_t v = 0;
    if (gzread(f, &v, 4) != 4) return -1;
    return v;
}

int main(int argc, char **argv)
{
    gzFile f;
    char magic[4], **names;
    int32_t l_text, n_ref, i, *lens;
    int W, rc;
    Batch B;
    hpn_ctx *ctx;
    hpn_bam_batch view;
    uint8_t *rec = NULL;
    size_t rec_cap = 0;
    if (argc < 3) return 1;
    W = atoi(argv[2]);
    memset(&B, 0, sizeof B);
    f = gzopen(argv[1], "rb");
    if (!f || gzread(f, magic, 4) != 4 || memcmp(magic, "BAM\1", 4)) return 1;
    l_text = rd32(f);
    gzseek(f, l_text, SEEK_CUR);
    n_ref = rd32(f);
    names = (char **)calloc((size_t)n_ref, sizeof *names), lens = (int32_t *)calloc((size_t)n_ref, 4);
    for (i = 0; i < n_ref; ++i) {
        int32_t l = rd32(f);
        names[i] = (char *)malloc((size_t)l);
        gzread(f, names[i], (unsigned)l);
        lens[i] = rd32(f);
    }
    for (;;) {                                          /* samread loop: bam_read1 + callback */
        int32_t bs = rd32(f);
        Rec r;
        uint32_t flag_nc;
        if (bs < 32) break;
        if ((size_t)bs > rec_cap) rec = (uint8_t *)realloc(rec, rec_cap = (size_t)bs);
        if (gzread(f, rec, (unsigned)bs) != bs) break;
        memcpy(&r.tid, rec, 4), memcpy(&r.pos, rec + 4, 4), memcpy(&flag_nc, rec + 12, 4);
        r.flag = flag_nc >> 16, r.n_cigar = flag_nc & 0xffffu;
        r.cigar = (const uint32_t *)(const void *)(rec + 32 + rec[8]);
        fetch_func(&r, &B);
    }
    gzclose(f);
    if ((rc = hpn_ctx_create(0, &ctx)) != HPN_OK) {
        fprintf(stderr, "hpn_ctx_create: %d\n", rc);
        return 2;
    }
    memset(&view, 0, sizeof view);
    view.n = B.n, view.tid = B.tid, view.pos = B.pos, view.flag = B.flag, view.cigar_off = B.cigar_off, view.cigar = B.cigar;
    for (i = 0; i < n_ref; ++i) {                       /* main(), per target -- replaces bam2depth.c:327-334 */
        uint64_t n_runs = 0, cap = 2 * B.n + 16, k, windows = (uint64_t)lens[i] / (uint64_t)W + 1;
        hpn_run *runs = (hpn_run *)malloc(cap * sizeof *runs);
        uint64_t *win_sum = (uint64_t *)calloc(windows, 8);
        if (hpn_depth_begin(ctx, i, (uint32_t)lens[i], 0x704) != HPN_OK) return 2;           /* BAM_DEF_MASK (bam.h:124) */
        if (B.n && hpn_depth_add(ctx, &view) != HPN_OK) return 2;
        if ((rc = hpn_depth_finish(ctx, (uint32_t)W, runs, cap, &n_runs, win_sum)) != HPN_OK) {
            fprintf(stderr, "hpn_depth_finish: %d %s\n", rc, hpn_ctx_last_error(ctx));
            return 2;
        }
        for (k = 0; k < n_runs; ++k) printf("%s\t%d\t%d\t%d\n", names[i], runs[k].start, runs[k].end, runs[k].depth);
        for (k = 0; k < windows; ++k) printf("#win %s %llu %llu\n", names[i], (unsigned long long)k, (unsigned long long)win_sum[k]);
        free(runs), free(win_sum);
    }
    hpn_ctx_destroy(ctx);
    return 0;
}
