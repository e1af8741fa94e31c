/* seam1_count_read.c -- the INTEGRATION.md "Seam 1" stub as a C99 program.
 *
 * count_read() keeps the reference's signature (fastq_count_kthread.c:116,126): the caller owns and zeroes
 * sumFreq / minLen / maxLen / SeqLen[512] / Quality[128][512], the callee reads the stream with the same
 * 4 x gzgets framing and only ADDS -- but the tally (AssignQuality, fastq_count.c:29-35) runs in libhpngs.
 * main() plays the reference's load_fq + the numbers its report row is made of, printed plainly:
 *   line 1: reads min_len max_len sum q20 q30
 *   line 2: SeqLen[0..511]
 *   128 lines: Quality[q][0..511]
 * tests/test_abi_c.py compiles this with  gcc -std=c99 -pedantic -Wall -Werror  against include/hpngs.h and
 * compares the numbers with the reference's own per-file .tsv (tests/golden/expected/kthread_*).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <zlib.h>

#include "hpngs.h"

static hpn_ctx *ctx;

static void count_read(gzFile fq, uint32_t *sumFreq, uint32_t *minLen, uint32_t *maxLen, uint64_t *SeqLen,
                       uint64_t **Quality, const char *infile, FILE *out, double *mean_length)
{
    enum { BATCH = 1000 };                              /* small on purpose: several GPU calls per file */
    uint8_t *qual = (uint8_t *)malloc((size_t)BATCH * 1024);
    uint64_t *off = (uint64_t *)malloc((BATCH + 1) * sizeof *off);
    uint64_t *flatQ = (uint64_t *)calloc(128 * 512, sizeof *flatQ);
    hpn_tally acc;
    char *buf = (char *)calloc(1024, 1);
    uint64_t n = 0, sum_len = 0;
    int l, q, p;
    (void)infile, (void)out;
    memset(&acc, 0, sizeof acc);
    acc.qual_hist = flatQ;
    off[0] = 0;
    while (gzgets(fq, buf, 1024) != NULL) {             /* unchanged 4 x gzgets framing (:128-133) */
        uint16_t seqLen;
        gzgets(fq, buf, 1024);
        seqLen = (uint16_t)(strlen(buf) - 1);
        gzgets(fq, buf, 1024);
        gzgets(fq, buf, 1024);
        memcpy(qual + off[n], buf, seqLen);             /* was: AssignQuality(Quality, buf, seqLen) */
        off[n + 1] = off[n] + seqLen;
        if (++n == BATCH) {
            if (hpn_fastq_tally(ctx, qual, NULL, off, n, &acc) != HPN_OK) exit(2);
            n = 0;
        }
    }
    if (hpn_fastq_tally(ctx, qual, NULL, off, n, &acc) != HPN_OK) exit(2);
    for (l = 0; l < 512; ++l) SeqLen[l] += acc.seqlen[l];   /* add into the caller's arrays */
    for (q = 0; q < 128; ++q)
        for (p = 0; p < 512; ++p) Quality[q][p] += flatQ[q * 512 + p];
    /* statSeqLen (fastq_count.c:63-74) */
    *minLen = 0, *maxLen = 0, *sumFreq = 0;
    for (l = 0; l < 512; ++l) {
        if (!SeqLen[l]) continue;
        if (*minLen == 0) *minLen = (uint32_t)l;
        *maxLen = (uint32_t)l;
        *sumFreq += (uint32_t)SeqLen[l];
        sum_len += SeqLen[l] * (uint64_t)l;
    }
    *mean_length = (double)sum_len;
    free(qual), free(off), free(flatQ), free(buf);
}

int main(int argc, char **argv)
{
    uint32_t sumFreq = 0, minLen = 0, maxLen = 0;
    uint64_t *SeqLen = (uint64_t *)calloc(512, sizeof *SeqLen);
    uint64_t **Quality = (uint64_t **)calloc(128, sizeof *Quality);
    uint64_t sum = 0, q20 = 0, q30 = 0;
    double mean_length = 0;
    gzFile fq;
    int q, p, rc;
    if (argc < 2) return 1;
    for (q = 0; q < 128; ++q) Quality[q] = (uint64_t *)calloc(512, sizeof **Quality);
    if ((rc = hpn_ctx_create(0, &ctx)) != HPN_OK) {
        fprintf(stderr, "hpn_ctx_create: %d\n", rc);
        return 2;
    }
    fq = gzopen(argv[1], "rb");
    if (!fq) return 1;
    count_read(fq, &sumFreq, &minLen, &maxLen, SeqLen, Quality, argv[1], stdout, &mean_length);
    gzclose(fq);
    for (q = 0; q < 128; ++q)                           /* statQ(Quality,128,512,sum,53,sumQ1,63,sumQ2) (:37-47) */
        for (p = 0; p < 512; ++p) {
            sum += Quality[q][p];
            if (q >= 53) q20 += Quality[q][p];
            if (q >= 63) q30 += Quality[q][p];
        }
    printf("%u %u %u %llu %llu %llu\n", sumFreq, minLen, maxLen, (unsigned long long)sum, (unsigned long long)q20,
           (unsigned long long)q30);
    for (p = 0; p < 512; ++p) printf("%llu%c", (unsigned long long)SeqLen[p], p == 511 ? '\n' : ' ');
    for (q = 0; q < 128; ++q)
        for (p = 0; p < 512; ++p) printf("%llu%c", (unsigned long long)Quality[q][p], p == 511 ? '\n' : ' ');
    hpn_ctx_destroy(ctx);
    return 0;
}
