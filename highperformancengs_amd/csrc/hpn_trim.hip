// hpn_trim.hip -- C ABI of the fastq_trim cut (reference fastq_trim.c:67-89) and of the
// quality-threshold trim-point extension.
#include "hpn_ctx.hpp"

namespace hpn {
hipError_t launch_trim(const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint64_t S,
                       uint64_t E, const uint32_t *d_beg, const uint32_t *d_end, uint8_t *d_out_seq,
                       uint8_t *d_out_qual, uint64_t *d_out_off, u64 *d_status, uint32_t *d_ticket_err, int n_cu,
                       hipStream_t st);
hipError_t launch_qtrim_points(const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint32_t T, uint32_t *d_beg,
                               uint32_t *d_end, int n_cu, hipStream_t st);
uint64_t trim_status_words(uint64_t n);
}  // namespace hpn

using namespace hpn;

// Cut on device-resident arrays: fixed cycles [S,E) when d_beg/d_end are NULL, else per-record points.
static int trim_dev(hpn_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_off, uint64_t n,
                    int32_t S, int32_t E, const uint32_t *d_beg, const uint32_t *d_end, uint8_t *d_out_seq,
                    uint8_t *d_out_qual, uint64_t *d_out_off)
{
    if (S < 0 || E < S) return fail(c, HPN_E_DOMAIN, "need 0 <= S <= E (got S=%d E=%d)", S, E);
    int rc = scratch_reserve(c, c->s_g, (trim_status_words(n) + 2) * sizeof(u64));
    if (rc != HPN_OK) return rc;
    u64 *status = (u64 *)c->s_g.p + 2;
    uint32_t *ticket_err = (uint32_t *)c->s_g.p;
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamTrim], c->stream));
    HPN_HIP(c, launch_trim(d_seq, d_qual, d_off, n, (uint64_t)S, (uint64_t)E, d_beg, d_end, d_out_seq, d_out_qual,
                           d_out_off, status, ticket_err, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamTrim], c->stream));
    c->ev_valid[kFamTrim] = true;
    return HPN_OK;
}

// Host batch: stage, cut, copy back.  beg/end NULL = fixed cycles [S,E).
static int trim_host(hpn_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n, int32_t S,
                     int32_t E, const uint32_t *beg, const uint32_t *end, uint8_t *out_seq, uint8_t *out_qual,
                     uint64_t *out_off)
{
    if (!c || !off || !out_off || (!beg != !end)) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t b0 = off[0], b1 = off[n];
    if (b1 < b0) return fail(c, HPN_E_ARG, "offsets decrease");
    const uint64_t nbytes = b1 - b0;
    if (nbytes && (!seq || !qual || !out_seq || !out_qual)) return fail(c, HPN_E_ARG, "NULL array");
    int rc;
    if ((rc = scratch_reserve(c, c->s_a, nbytes + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_b, nbytes + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_c, (n + 1) * sizeof(uint64_t))) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_d, nbytes + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_e, nbytes + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_f, (n + 1) * sizeof(uint64_t))) != HPN_OK) return rc;
    const uint32_t *d_beg = nullptr, *d_end = nullptr;
    if (beg) {
        if ((rc = scratch_reserve(c, c->s_h, 2 * (n + 1) * sizeof(uint32_t))) != HPN_OK) return rc;
        uint32_t *p = (uint32_t *)c->s_h.p;
        HPN_HIP(c, hipMemcpyAsync(p, beg, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HPN_HIP(c, hipMemcpyAsync(p + n + 1, end, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        d_beg = p, d_end = p + n + 1;
    }
    uint8_t *ds = (uint8_t *)c->s_a.p, *dq = (uint8_t *)c->s_b.p;
    if (nbytes) {
        HPN_HIP(c, hipMemcpyAsync(ds, seq + b0, nbytes, hipMemcpyHostToDevice, c->stream));
        HPN_HIP(c, hipMemcpyAsync(dq, qual + b0, nbytes, hipMemcpyHostToDevice, c->stream));
    }
    HPN_HIP(c, hipMemcpyAsync(c->s_c.p, off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    // host offsets stay valid on the device: array base shifted by -b0
    rc = trim_dev(c, ds - b0, dq - b0, (const uint64_t *)c->s_c.p, n, S, E, d_beg, d_end, (uint8_t *)c->s_d.p,
                  (uint8_t *)c->s_e.p, (uint64_t *)c->s_f.p);
    if (rc != HPN_OK) return rc;
    uint32_t te[2] = {0, 0};
    HPN_HIP(c, hipMemcpyAsync(out_off, c->s_f.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(te, c->s_g.p, sizeof te, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    if (te[1]) return fail(c, HPN_E_HIP, "prefix-scan hand-off timed out");
    const uint64_t total = out_off[n];
    if (total > nbytes) return fail(c, HPN_E_HIP, "scan produced %llu > %llu bytes", (unsigned long long)total, (unsigned long long)nbytes);
    if (total) {
        HPN_HIP(c, hipMemcpyAsync(out_seq, c->s_d.p, total, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipMemcpyAsync(out_qual, c->s_e.p, total, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipStreamSynchronize(c->stream));
    }
    return HPN_OK;
}

extern "C" {

int hpn_fastq_trim_dev(hpn_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_off, uint64_t n,
                       int32_t S, int32_t E, uint8_t *d_out_seq, uint8_t *d_out_qual, uint64_t *d_out_off)
{
    if (!c || !d_off || !d_out_off || (n && (!d_seq || !d_qual || !d_out_seq || !d_out_qual))) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    return trim_dev(c, d_seq, d_qual, d_off, n, S, E, nullptr, nullptr, d_out_seq, d_out_qual, d_out_off);
}

int hpn_fastq_trim(hpn_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n, int32_t S,
                   int32_t E, uint8_t *out_seq, uint8_t *out_qual, uint64_t *out_off)
{
    return trim_host(c, seq, qual, off, n, S, E, nullptr, nullptr, out_seq, out_qual, out_off);
}

int hpn_fastq_trim_points_dev(hpn_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_off,
                              uint64_t n, const uint32_t *d_beg, const uint32_t *d_end, uint8_t *d_out_seq,
                              uint8_t *d_out_qual, uint64_t *d_out_off)
{
    if (!c || !d_off || !d_out_off || (n && (!d_seq || !d_qual || !d_out_seq || !d_out_qual || !d_beg || !d_end)))
        return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    return trim_dev(c, d_seq, d_qual, d_off, n, 0, 0, d_beg, d_end, d_out_seq, d_out_qual, d_out_off);
}

int hpn_fastq_trim_points(hpn_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n,
                          const uint32_t *beg, const uint32_t *end, uint8_t *out_seq, uint8_t *out_qual,
                          uint64_t *out_off)
{
    if (n && (!beg || !end)) return HPN_E_ARG;
    static const uint32_t none = 0;
    return trim_host(c, seq, qual, off, n, 0, 0, beg ? beg : &none, end ? end : &none, out_seq, out_qual, out_off);
}

int hpn_fastq_qtrim_points_dev(hpn_ctx *c, const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint32_t threshold,
                               uint32_t *d_beg, uint32_t *d_end)
{
    if (!c || !d_off || (n && (!d_qual || !d_beg || !d_end))) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamTrim], c->stream));
    HPN_HIP(c, launch_qtrim_points(d_qual, d_off, n, threshold, d_beg, d_end, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamTrim], c->stream));
    c->ev_valid[kFamTrim] = true;
    return HPN_OK;
}

int hpn_fastq_qtrim_points(hpn_ctx *c, const uint8_t *qual, const uint64_t *off, uint64_t n, uint32_t threshold,
                           uint32_t *beg, uint32_t *end)
{
    if (!c || !off || (n && (!beg || !end))) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t b0 = off[0], b1 = off[n];
    if (b1 < b0) return fail(c, HPN_E_ARG, "offsets decrease");
    if (b1 > b0 && !qual) return fail(c, HPN_E_ARG, "qual is NULL");
    int rc;
    if ((rc = scratch_reserve(c, c->s_a, (b1 - b0) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_c, (n + 1) * sizeof(uint64_t))) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_h, 2 * (n + 1) * sizeof(uint32_t))) != HPN_OK) return rc;
    uint8_t *dq = (uint8_t *)c->s_a.p;
    uint32_t *p = (uint32_t *)c->s_h.p;
    if (b1 > b0) HPN_HIP(c, hipMemcpyAsync(dq, qual + b0, b1 - b0, hipMemcpyHostToDevice, c->stream));
    HPN_HIP(c, hipMemcpyAsync(c->s_c.p, off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    rc = hpn_fastq_qtrim_points_dev(c, dq - b0, (const uint64_t *)c->s_c.p, n, threshold, p, p + n + 1);
    if (rc != HPN_OK) return rc;
    if (n) {
        HPN_HIP(c, hipMemcpyAsync(beg, p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipMemcpyAsync(end, p + n + 1, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    }
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    return HPN_OK;
}

}  // extern "C"
