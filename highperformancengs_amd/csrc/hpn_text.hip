// hpn_text.hip -- C ABI of the raw-text front end: device-side framing of FASTQ text
// (the 4 x gzgets loops of fastq_count.c:112-118 and fastq_trim.c:67-89), feeding the
// tally kernels or writing the trimmed text.  Kernels: kernels/fastq_text.hip.
#include <string.h>

#include "hpn_ctx.hpp"

namespace hpn {
hipError_t launch_text_frame(const uint8_t *d_slot, uint32_t begin, uint32_t end, int last, bool trim, uint32_t S,
                             uint32_t E, uint32_t carry_cap, uint32_t *d_nl, uint32_t nl_cap, uint64_t *d_off,
                             u64 *d_status, uint32_t *d_state, hipStream_t st);
hipError_t launch_text_gather(const uint8_t *d_slot, const uint32_t *d_nl, const uint64_t *d_off, uint32_t n,
                              uint8_t *d_out_qual, uint8_t *d_out_seq, int n_cu, hipStream_t st);
hipError_t launch_text_trim(const uint8_t *d_slot, const uint32_t *d_nl, uint32_t begin, int at_begin, const uint64_t *d_off,
                            uint32_t n, uint32_t S, uint32_t E, uint8_t *d_out, int n_cu, hipStream_t st);
hipError_t launch_text_lines(const uint8_t *d_slot, uint32_t begin, uint32_t end, int last, uint32_t own_end, uint32_t *d_nl,
                             uint32_t nl_cap, u64 *d_status, uint32_t *d_state, hipStream_t st);
hipError_t launch_text_records(const uint32_t *d_nl, uint32_t begin, uint32_t end, int last, bool trim, uint32_t S, uint32_t E,
                               uint32_t carry_cap, int32_t first, uint32_t limit, uint32_t nl_cap, uint64_t *d_off,
                               u64 *d_status, uint32_t *d_state, hipStream_t st);
uint64_t text_tiles1(uint32_t begin, uint32_t end);
uint64_t text_tiles2(uint32_t nl_cap);
}  // namespace hpn

using namespace hpn;

namespace {

constexpr uint32_t kCarryCap = 8192;  // room in front of a chunk for the unfinished record of the previous one
constexpr int kStateWords = 16;       // kernels/fastq_text.hip: kTs*
enum { kTsLines = 0, kTsRecs, kTsFlags, kTsUnterminated, kTsConsumed, kTsTotalLo, kTsTotalHi, kTsErr, kTsTicket1, kTsTicket2, kTsOwnLines };

struct Framed {
    uint32_t begin = 0, n = 0;
    uint64_t total = 0;
    const uint8_t *slot = nullptr;
    const uint32_t *nlp = nullptr;   // pieces: the first record's four line ends
};

// Copy the chunk behind the carried-over bytes, index the lines, validate and scan the
// records, wait for the verdict.  On return the caller's `text` buffer is free again.
// in_place (round 6): `text` is DEVICE memory with kCarryCap writable bytes in front of it, and is framed where it lies -- only
// the carried bytes (< 8 KiB) are copied, in front of it; they are kept in a buffer of their own between calls, so the caller may
// overwrite the text as soon as the call returns.  (The gzip route's text is on the device already: the copy into a slot was
// 2 x 15.9 GB of traffic per 9.3 GB batch, 5.7 ms of its ~250: profiles/r06/kernel_stats_gz_tool_members.csv, __amd_rocclr_copyBuffer.)
int text_frame(hpn_ctx *c, const void *text, uint64_t nbytes, int last, bool trim, uint32_t S, uint32_t E,
               hpn_text_info *info, Framed *f, bool in_place = false)
{
    if (!c->t_open) return fail(c, HPN_E_STATE, "hpn_fastq_text_begin first (or the stream was closed by an irregular chunk)");
    if (nbytes >= (1ull << 31) - 2 * kCarryCap) return fail(c, HPN_E_ARG, "chunk of %llu bytes (limit 2^31 - 16 KiB)", (unsigned long long)nbytes);
    if (nbytes && !text) return fail(c, HPN_E_ARG, "text is NULL");
    memset(info, 0, sizeof *info);
    int rc;
    if (!c->t_state) {
        HPN_HIP(c, hipMalloc((void **)&c->t_state, kStateWords * sizeof(uint32_t)));
        HPN_HIP(c, hipHostMalloc((void **)&c->h_tstate, kStateWords * sizeof(uint32_t), hipHostMallocDefault));
    }
    const int cur = c->t_cur;
    const uint32_t carry = c->t_carry;
    const uint32_t begin = kCarryCap - carry, end = kCarryCap + (uint32_t)nbytes;
    if (in_place && !text) return fail(c, HPN_E_ARG, "text is NULL");
    if (!in_place && (rc = scratch_reserve(c, c->t_slot[cur], (size_t)end + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->t_carrybuf, kCarryCap)) != HPN_OK) return rc;
    uint8_t *slot = in_place ? (uint8_t *)const_cast<void *>(text) - kCarryCap : (uint8_t *)c->t_slot[cur].p;
    if (carry)
        HPN_HIP(c, hipMemcpyAsync(slot + begin, c->t_carry_saved ? (const uint8_t *)c->t_carrybuf.p : (const uint8_t *)c->t_slot[cur ^ 1].p + c->t_tail, carry,
                                  hipMemcpyDeviceToDevice, c->stream));
    if (nbytes && !in_place) HPN_HIP(c, hipMemcpyAsync(slot + kCarryCap, text, nbytes, hipMemcpyDefault, c->stream));
    // one line per 4 bytes is the most the index is sized for (HPN_TEXT_DENSE beyond)
    const uint32_t nl_cap = (((end - begin) / 4u) + 16u) & ~3u;
    if ((rc = scratch_reserve(c, c->t_nl, (size_t)nl_cap * sizeof(uint32_t) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->t_off, ((size_t)nl_cap / 4 + 2) * sizeof(uint64_t))) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->t_status, (text_tiles1(begin, end) + text_tiles2(nl_cap)) * sizeof(u64))) != HPN_OK) return rc;
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamText], c->stream));
    HPN_HIP(c, launch_text_frame(slot, begin, end, last, trim, S, E, kCarryCap - 64, (uint32_t *)c->t_nl.p, nl_cap,
                                 (uint64_t *)c->t_off.p, (u64 *)c->t_status.p, c->t_state, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamText], c->stream));
    c->ev_valid[kFamText] = true;
    HPN_HIP(c, hipMemcpyAsync(c->h_tstate, c->t_state, kStateWords * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t *h = c->h_tstate;
    if (h[kTsErr]) {
        c->t_open = false;
        return fail(c, HPN_E_HIP, "prefix-scan hand-off timed out");
    }
    if (h[kTsFlags]) {  // nothing of this chunk was used; the caller re-frames the stream exactly
        info->irregular = h[kTsFlags];
        c->t_open = false;
        return HPN_OK;
    }
    f->begin = begin;
    f->n = h[kTsRecs];
    f->total = ((uint64_t)h[kTsTotalHi] << 32) | h[kTsTotalLo];
    f->slot = slot;
    info->n_records = f->n;
    info->n_bytes = f->total;
    info->carry_bytes = last ? 0 : end - h[kTsConsumed];
    c->t_carry = (uint32_t)info->carry_bytes;
    c->t_tail = h[kTsConsumed];
    c->t_cur = cur ^ 1;
    c->t_carry_saved = false;
    if (in_place && c->t_carry) {   // (on the stream: in front of whatever the caller lets write over the text next)
        HPN_HIP(c, hipMemcpyAsync(c->t_carrybuf.p, slot + c->t_tail, c->t_carry, hipMemcpyDeviceToDevice, c->stream));
        c->t_carry_saved = true;
    }
    if (last) c->t_open = false;
    return HPN_OK;
}

}  // namespace

extern "C" {

int hpn_fastq_text_begin(hpn_ctx *c)
{
    if (!c) return HPN_E_ARG;
    c->t_open = true;
    c->t_cur = 0;
    c->t_carry = 0;
    c->t_tail = 0;
    c->t_carry_saved = false;
    return HPN_OK;
}

int hpn_fastq_text_count(hpn_ctx *c, const void *text, uint64_t nbytes, int last, uint32_t flags, hpn_text_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    if (flags & ~(HPN_TALLY_QUAL_HIST | HPN_TALLY_NUC_HIST)) return fail(c, HPN_E_ARG, "unknown flags 0x%x", flags);
    HPN_HIP(c, hipSetDevice(c->device));
    Framed f;
    int rc = text_frame(c, text, nbytes, last, false, 0, 0, info, &f);
    if (rc != HPN_OK || info->irregular || f.n == 0) return rc;
    const bool nuc = flags & HPN_TALLY_NUC_HIST;
    if ((rc = scratch_reserve(c, c->t_pq, f.total + 64)) != HPN_OK) return rc;
    if (nuc && (rc = scratch_reserve(c, c->t_ps, f.total + 64)) != HPN_OK) return rc;
    HPN_HIP(c, launch_text_gather(f.slot, (const uint32_t *)c->t_nl.p, (const uint64_t *)c->t_off.p, f.n, (uint8_t *)c->t_pq.p,
                                  nuc ? (uint8_t *)c->t_ps.p : nullptr, c->n_cu, c->stream));
    return tally_launch(c, (const uint8_t *)c->t_pq.p, nuc ? (const uint8_t *)c->t_ps.p : nullptr, (const uint64_t *)c->t_off.p,
                        f.n, f.total, flags);
}

int hpn_fastq_text_count_inplace(hpn_ctx *c, const uint8_t *d_text, uint64_t nbytes, int last, uint32_t flags, hpn_text_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    if (flags & ~(HPN_TALLY_QUAL_HIST | HPN_TALLY_NUC_HIST)) return fail(c, HPN_E_ARG, "unknown flags 0x%x", flags);
    HPN_HIP(c, hipSetDevice(c->device));
    Framed f;
    int rc = text_frame(c, d_text, nbytes, last, false, 0, 0, info, &f, true);
    if (rc != HPN_OK || info->irregular || f.n == 0) return rc;
    const bool nuc = flags & HPN_TALLY_NUC_HIST;
    if ((rc = scratch_reserve(c, c->t_pq, f.total + 64)) != HPN_OK) return rc;
    if (nuc && (rc = scratch_reserve(c, c->t_ps, f.total + 64)) != HPN_OK) return rc;
    HPN_HIP(c, launch_text_gather(f.slot, (const uint32_t *)c->t_nl.p, (const uint64_t *)c->t_off.p, f.n, (uint8_t *)c->t_pq.p,
                                  nuc ? (uint8_t *)c->t_ps.p : nullptr, c->n_cu, c->stream));
    rc = tally_launch(c, (const uint8_t *)c->t_pq.p, nuc ? (const uint8_t *)c->t_ps.p : nullptr, (const uint64_t *)c->t_off.p, f.n, f.total, flags);
    // (the gather reads the text: it has to be through before the caller may write over it)
    if (rc == HPN_OK) HPN_HIP(c, hipStreamSynchronize(c->stream));
    return rc;
}

int hpn_fastq_text_records(hpn_ctx *c, const void *text, uint64_t nbytes, int last, hpn_text_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    Framed f;
    return text_frame(c, text, nbytes, last, false, 0, 0, info, &f);   // lines + records only: nothing is gathered or tallied
}

int hpn_fastq_text_trim(hpn_ctx *c, const void *text, uint64_t nbytes, int last, int32_t S, int32_t E, void *out_text,
                        uint64_t out_cap, hpn_text_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    if (S < 0 || E < S) return fail(c, HPN_E_DOMAIN, "need 0 <= S <= E (got S=%d E=%d)", S, E);
    HPN_HIP(c, hipSetDevice(c->device));
    Framed f;
    int rc = text_frame(c, text, nbytes, last, true, (uint32_t)S, (uint32_t)E, info, &f);
    if (rc != HPN_OK || info->irregular || f.n == 0) return rc;
    if (f.total > out_cap || !out_text) {
        c->t_open = false;
        return fail(c, HPN_E_CAPACITY, "trimmed text needs %llu bytes, out_cap is %llu", (unsigned long long)f.total,
                    (unsigned long long)out_cap);
    }
    if ((rc = scratch_reserve(c, c->t_out, f.total + 64)) != HPN_OK) return rc;
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamTrim], c->stream));
    HPN_HIP(c, launch_text_trim(f.slot, (const uint32_t *)c->t_nl.p, f.begin, 1, (const uint64_t *)c->t_off.p, f.n, (uint32_t)S,
                                (uint32_t)E, (uint8_t *)c->t_out.p, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamTrim], c->stream));
    c->ev_valid[kFamTrim] = true;
    HPN_HIP(c, hipMemcpyAsync(out_text, c->t_out.p, f.total, hipMemcpyDefault, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    return HPN_OK;
}

}  // extern "C"

// ---- one stream, several contexts: pieces (include/hpngs.h) ------------------------------------------------

namespace {

// The first half: copy, line index, the piece's contribution to the stream's line count.
int piece_lines(hpn_ctx *c, const void *text, uint64_t nbytes, uint32_t head, uint64_t own_bytes, int last, hpn_text_piece *out)
{
    if (nbytes >= (1ull << 31) - 2 * kCarryCap) return fail(c, HPN_E_ARG, "piece of %llu bytes (limit 2^31 - 16 KiB)", (unsigned long long)nbytes);
    if (nbytes && !text) return fail(c, HPN_E_ARG, "text is NULL");
    if (head > 1 || head + own_bytes > nbytes || (last && head + own_bytes != nbytes))
        return fail(c, HPN_E_ARG, "piece: head %u + own %llu bytes of %llu%s", head, (unsigned long long)own_bytes, (unsigned long long)nbytes, last ? " (last)" : "");
    memset(out, 0, sizeof *out);
    c->p_state = 0;
    c->t_open = false;   // a chunked stream of this context ends here
    int rc;
    if (!c->t_state) {
        HPN_HIP(c, hipMalloc((void **)&c->t_state, kStateWords * sizeof(uint32_t)));
        HPN_HIP(c, hipHostMalloc((void **)&c->h_tstate, kStateWords * sizeof(uint32_t), hipHostMallocDefault));
    }
    const uint32_t begin = kCarryCap, end = kCarryCap + (uint32_t)nbytes;
    // records starting at or behind `limit` are the next piece's; the line ends in front of limit - 1 are this piece's to count
    const uint32_t limit = last ? end : begin + head + (uint32_t)own_bytes;
    if ((rc = scratch_reserve(c, c->t_slot[0], (size_t)end + 64)) != HPN_OK) return rc;
    uint8_t *slot = (uint8_t *)c->t_slot[0].p;
    if (nbytes) HPN_HIP(c, hipMemcpyAsync(slot + kCarryCap, text, nbytes, hipMemcpyDefault, c->stream));
    const uint32_t nl_cap = (((end - begin) / 4u) + 16u) & ~3u;
    if ((rc = scratch_reserve(c, c->t_nl, (size_t)nl_cap * sizeof(uint32_t) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->t_off, ((size_t)nl_cap / 4 + 2) * sizeof(uint64_t))) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->t_status, (text_tiles1(begin, end) + text_tiles2(nl_cap)) * sizeof(u64))) != HPN_OK) return rc;
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamText], c->stream));
    // limit - 1 >= begin except for an empty first piece (then nothing is counted: own_end 0 would also mean "not a piece", same thing)
    HPN_HIP(c, launch_text_lines(slot, begin, end, last, limit > begin ? limit - 1u : 0u, (uint32_t *)c->t_nl.p, nl_cap, (u64 *)c->t_status.p,
                                 c->t_state, c->stream));
    HPN_HIP(c, hipMemcpyAsync(c->h_tstate, c->t_state, kStateWords * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t *h = c->h_tstate;
    if (h[kTsErr]) return fail(c, HPN_E_HIP, "prefix-scan hand-off timed out");
    out->irregular = h[kTsFlags];   // NUL / DENSE show up here already
    out->n_lines = h[kTsOwnLines];
    if (!out->irregular) {
        c->p_state = 1;
        c->p_begin = begin, c->p_end = end, c->p_limit = limit ? limit : 1u, c->p_head = head, c->p_last = last, c->p_nl_cap = nl_cap;
    }
    return HPN_OK;
}

// The second half: the records of the piece, given the number of lines the stream has in front of the text handed over.
int piece_records(hpn_ctx *c, uint64_t lines_before, bool trim, uint32_t S, uint32_t E, hpn_text_info *info, Framed *f, int *at_begin)
{
    if (c->p_state != 1) return fail(c, HPN_E_STATE, "hpn_fastq_text_piece_lines first (and the piece must have been regular)");
    c->p_state = 0;
    memset(info, 0, sizeof *info);
    if (!c->p_head && lines_before) return fail(c, HPN_E_ARG, "a piece without the byte in front of it is the stream's first: lines_before must be 0");
    // a record starts behind local line end i iff (lines_before + i + 1) % 4 == 0; i = -1 is the stream's first byte
    const int32_t first = c->p_head ? (int32_t)((4u - (uint32_t)((lines_before + 1) & 3u)) & 3u) : -1;
    *at_begin = first < 0;
    HPN_HIP(c, launch_text_records((const uint32_t *)c->t_nl.p, c->p_begin, c->p_end, c->p_last, trim, S, E, kCarryCap - 64, first, c->p_limit,
                                   c->p_nl_cap, (uint64_t *)c->t_off.p, (u64 *)c->t_status.p, c->t_state, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamText], c->stream));
    c->ev_valid[kFamText] = true;
    HPN_HIP(c, hipMemcpyAsync(c->h_tstate, c->t_state, kStateWords * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t *h = c->h_tstate;
    if (h[kTsErr]) return fail(c, HPN_E_HIP, "prefix-scan hand-off timed out");
    if (h[kTsFlags]) {
        info->irregular = h[kTsFlags];
        return HPN_OK;
    }
    f->begin = c->p_begin;
    f->n = h[kTsRecs];
    f->total = ((uint64_t)h[kTsTotalHi] << 32) | h[kTsTotalLo];
    f->slot = (const uint8_t *)c->t_slot[0].p;
    f->nlp = (const uint32_t *)c->t_nl.p + (first + 1);
    info->n_records = f->n;
    info->n_bytes = f->total;
    return HPN_OK;
}

}  // namespace

extern "C" {

int hpn_fastq_text_piece_lines(hpn_ctx *c, const void *text, uint64_t nbytes, uint32_t head, uint64_t own_bytes, int last,
                               hpn_text_piece *out)
{
    if (!c || !out) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    return piece_lines(c, text, nbytes, head, own_bytes, last, out);
}

int hpn_fastq_text_piece_count(hpn_ctx *c, uint64_t lines_before, uint32_t flags, hpn_text_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    if (flags & ~(HPN_TALLY_QUAL_HIST | HPN_TALLY_NUC_HIST)) return fail(c, HPN_E_ARG, "unknown flags 0x%x", flags);
    HPN_HIP(c, hipSetDevice(c->device));
    Framed f;
    int at_begin = 0;
    int rc = piece_records(c, lines_before, false, 0, 0, info, &f, &at_begin);
    if (rc != HPN_OK || info->irregular || f.n == 0) return rc;
    const bool nuc = flags & HPN_TALLY_NUC_HIST;
    if ((rc = scratch_reserve(c, c->t_pq, f.total + 64)) != HPN_OK) return rc;
    if (nuc && (rc = scratch_reserve(c, c->t_ps, f.total + 64)) != HPN_OK) return rc;
    HPN_HIP(c, launch_text_gather(f.slot, f.nlp, (const uint64_t *)c->t_off.p, f.n, (uint8_t *)c->t_pq.p, nuc ? (uint8_t *)c->t_ps.p : nullptr,
                                  c->n_cu, c->stream));
    return tally_launch(c, (const uint8_t *)c->t_pq.p, nuc ? (const uint8_t *)c->t_ps.p : nullptr, (const uint64_t *)c->t_off.p, f.n, f.total, flags);
}

int hpn_fastq_text_piece_trim(hpn_ctx *c, uint64_t lines_before, int32_t S, int32_t E, void *out_text, uint64_t out_cap, hpn_text_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    if (S < 0 || E < S) return fail(c, HPN_E_DOMAIN, "need 0 <= S <= E (got S=%d E=%d)", S, E);
    HPN_HIP(c, hipSetDevice(c->device));
    Framed f;
    int at_begin = 0;
    int rc = piece_records(c, lines_before, true, (uint32_t)S, (uint32_t)E, info, &f, &at_begin);
    if (rc != HPN_OK || info->irregular || f.n == 0) return rc;
    if (f.total > out_cap || !out_text)
        return fail(c, HPN_E_CAPACITY, "trimmed text needs %llu bytes, out_cap is %llu", (unsigned long long)f.total, (unsigned long long)out_cap);
    if ((rc = scratch_reserve(c, c->t_out, f.total + 64)) != HPN_OK) return rc;
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamTrim], c->stream));
    HPN_HIP(c, launch_text_trim(f.slot, f.nlp, f.begin, at_begin, (const uint64_t *)c->t_off.p, f.n, (uint32_t)S, (uint32_t)E, (uint8_t *)c->t_out.p,
                                c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamTrim], c->stream));
    c->ev_valid[kFamTrim] = true;
    HPN_HIP(c, hipMemcpyAsync(out_text, c->t_out.p, f.total, hipMemcpyDefault, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    return HPN_OK;
}

}  // extern "C"
