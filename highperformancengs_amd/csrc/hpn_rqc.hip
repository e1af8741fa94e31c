// hpn_rqc.hip -- C ABI of the R plugin's per-read tally (Rgzfastq_uniq.c:42-57,174,250):
// Quality / Nucleotide / Length in the plugin's transposed int layouts + per-read GC fraction.
// The matrices come from the same histogram kernel as fastq_count_kthread -L (k_tally_hist);
// the GC fraction from k_read_gc (kernels/fastq_gc.hip).
#include <string.h>

#include <vector>

#include "hpn_ctx.hpp"

namespace hpn {
hipError_t launch_read_gc(const uint8_t *d_seq, const uint64_t *d_off, uint64_t n, double *d_gc, int n_cu, hipStream_t st);
}

using namespace hpn;

extern "C" {

int hpn_fastq_read_gc_dev(hpn_ctx *c, const uint8_t *d_seq, const uint64_t *d_off, uint64_t n, double *d_gc)
{
    if (!c || !d_off || (n && (!d_seq || !d_gc))) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    HPN_HIP(c, launch_read_gc(d_seq, d_off, n, d_gc, c->n_cu, c->stream));
    return HPN_OK;
}

int hpn_fastq_rqc(hpn_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n, hpn_rqc *out)
{
    if (!c || !off || !out) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t b0 = off[0], b1 = off[n];
    if (b1 < b0) return fail(c, HPN_E_ARG, "offsets decrease");
    const uint64_t nbytes = b1 - b0;
    if (nbytes && (!seq || !qual)) return fail(c, HPN_E_ARG, "NULL array");
    for (uint64_t i = 0; i < n; ++i) {  // Length[len-1] and the MaxLen-wide matrices (:174, :37-40)
        const uint64_t len = off[i + 1] - off[i];
        if (len < 1 || len > HPN_RQC_MAXLEN)
            return fail(c, HPN_E_DOMAIN, "read %llu has length %llu, outside 1..%d (the plugin writes out of bounds)",
                        (unsigned long long)i, (unsigned long long)len, HPN_RQC_MAXLEN);
    }
    std::vector<uint64_t> qh((size_t)HPN_QUAL_ROWS * HPN_LEN_BINS, 0), nh((size_t)HPN_NUC_CODES * HPN_LEN_BINS, 0);
    hpn_tally acc;
    memset(&acc, 0, sizeof acc);
    acc.qual_hist = qh.data();
    acc.nuc_hist = nh.data();
    int rc = hpn_fastq_tally(c, qual, seq, off, n, &acc);  // stages seq in s_b, qual in s_a, off in s_c
    if (rc != HPN_OK) return rc;
    if (out->gc && n) {
        if ((rc = scratch_reserve(c, c->s_d, n * sizeof(double))) != HPN_OK) return rc;
        const size_t pad = 16 + (b0 & 15);  // where hpn_fastq_tally put byte b0
        const uint8_t *d_seq = (const uint8_t *)c->s_b.p + pad - b0;
        HPN_HIP(c, launch_read_gc(d_seq, (const uint64_t *)c->s_c.p, n, (double *)c->s_d.p, c->n_cu, c->stream));
        HPN_HIP(c, hipMemcpyAsync(out->gc, c->s_d.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (out->quality)
        for (int q = 0; q < HPN_QUAL_ROWS; ++q)
            for (int p = 0; p < HPN_RQC_MAXLEN; ++p) out->quality[q + 128 * p] += (int32_t)qh[(size_t)q * HPN_LEN_BINS + p];
    if (out->nucleotide)
        for (int k = 0; k < HPN_NUC_CODES; ++k)
            for (int p = 0; p < HPN_RQC_MAXLEN; ++p) out->nucleotide[5 * p + k] += (int32_t)nh[(size_t)k * HPN_LEN_BINS + p];
    if (out->length)
        for (int l = 1; l <= HPN_RQC_MAXLEN; ++l) out->length[l - 1] += (int32_t)acc.seqlen[l];
    return HPN_OK;
}

}  // extern "C"
