// hpn_abi.hip -- C ABI: context, memory helpers, fastq tally, synthetic input.
// (trim: hpn_trim.hip, bam2depth / bam_sliding_count: hpn_bam.hip, RCCL: hpn_comm.hip)
#include <string.h>

#include "hpn_ctx.hpp"

using namespace hpn;

extern "C" {

int hpn_abi_version(void) { return HPN_ABI_VERSION; }

const char *hpn_strerror(int status)
{
    switch (status) {
    case HPN_OK: return "ok";
    case HPN_E_NODEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case HPN_E_HIP: return "HIP runtime call failed";
    case HPN_E_ARG: return "bad argument";
    case HPN_E_DOMAIN: return "input outside the reference's defined domain";
    case HPN_E_NOMEM: return "out of memory";
    case HPN_E_STATE: return "call sequence error";
    case HPN_E_RCCL: return "RCCL call failed";
    case HPN_E_CAPACITY: return "output buffer too small";
    default: return "unknown status";
    }
}

int hpn_device_count(int *n)
{
    if (!n) return HPN_E_ARG;
    hpn::warn_unread_env();
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) {
        *n = 0;
        (void)hipGetLastError();
        return HPN_E_NODEVICE;
    }
    *n = k;
    return HPN_OK;
}

int hpn_ctx_create(int device, hpn_ctx **out)
{
    if (!out) return HPN_E_ARG;
    *out = nullptr;
    hpn::warn_unread_env();
    int k = 0;
    if (hipGetDeviceCount(&k) != hipSuccess || k <= 0) {
        (void)hipGetLastError();
        return HPN_E_NODEVICE;
    }
    if (device < 0 || device >= k) return HPN_E_ARG;
    hpn_ctx *c = new (std::nothrow) hpn_ctx();
    if (!c) return HPN_E_NOMEM;
    c->device = device;
    int rc = HPN_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { rc = HPN_E_HIP; break; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { rc = HPN_E_HIP; break; }
        c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { rc = HPN_E_HIP; break; }
        c->stream = c->own_stream;
        if (hipMalloc((void **)&c->d_acc, (HPN_TALLY_WORDS + 16) * sizeof(u64)) != hipSuccess) { rc = HPN_E_NOMEM; break; }
        if (hipHostMalloc((void **)&c->h_acc, HPN_TALLY_WORDS * sizeof(u64), hipHostMallocDefault) != hipSuccess) { rc = HPN_E_NOMEM; break; }
        if (hipMemsetAsync(c->d_acc, 0, (HPN_TALLY_WORDS + 16) * sizeof(u64), c->stream) != hipSuccess) { rc = HPN_E_HIP; break; }
        for (int f = 0; f < kFamCount; ++f) {
            if (hipEventCreate(&c->ev_beg[f]) != hipSuccess || hipEventCreate(&c->ev_end[f]) != hipSuccess) { rc = HPN_E_HIP; break; }
        }
        if (rc != HPN_OK) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = HPN_E_HIP; break; }
    } while (0);
    if (rc != HPN_OK) {
        (void)hipGetLastError();
        hpn_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return HPN_OK;
}

int hpn_comm_destroy(hpn_ctx *ctx);

int hpn_ctx_destroy(hpn_ctx *c)
{
    if (!c) return HPN_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) hpn_comm_destroy(c);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    Scratch *ss[] = {&c->s_a, &c->s_b, &c->s_c, &c->s_d, &c->s_e, &c->s_f, &c->s_g, &c->s_h, &c->d_diff, &c->d_runs,
                     &c->d_win, &c->d_ws, &c->d_tidx, &c->d_text, &c->w_off, &c->w_bins, &c->w_len, &c->w_gc, &c->w_misc, &c->w_todo, &c->t_slot[0], &c->t_slot[1],
                     &c->t_nl, &c->t_off, &c->t_status, &c->t_pq, &c->t_ps, &c->t_out, &c->t_carrybuf, &c->r_counts, &c->r_bases, &c->r_off,
                     &c->r_tid, &c->r_pos, &c->r_flag, &c->r_lq, &c->r_soff, &c->r_info, &c->r_list, &c->g_sym, &c->g_meta, &c->g_windows, &c->g_summary, &c->g_bounds,
                     &c->g_crc, &c->b_ticket, &c->d_sw, &c->d_win_sw, &c->g_groups};
    for (Scratch *s : ss)
        if (s->p) (void)hipFree(s->p);
    if (c->d_acc) (void)hipFree(c->d_acc);
    if (c->h_acc) (void)hipHostFree(c->h_acc);
    if (c->t_state) (void)hipFree(c->t_state);
    if (c->h_tstate) (void)hipHostFree(c->h_tstate);
    for (int f = 0; f < kFamCount; ++f) {
        if (c->ev_beg[f]) (void)hipEventDestroy(c->ev_beg[f]);
        if (c->ev_end[f]) (void)hipEventDestroy(c->ev_end[f]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return HPN_OK;
}

int hpn_ctx_set_stream(hpn_ctx *c, void *hip_stream)
{
    if (!c) return HPN_E_ARG;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return HPN_OK;
}

int hpn_ctx_sync(hpn_ctx *c)
{
    if (!c) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    return HPN_OK;
}

int hpn_ctx_device(const hpn_ctx *c, int *device)
{
    if (!c || !device) return HPN_E_ARG;
    *device = c->device;
    return HPN_OK;
}

int hpn_ctx_pci_address(const hpn_ctx *c, char *buf, int len)
{
    if (!c || !buf || len < 16) return HPN_E_ARG;
    buf[0] = 0;
    if (hipDeviceGetPCIBusId(buf, len, c->device) != hipSuccess) {
        (void)hipGetLastError();
        buf[0] = 0;
        return HPN_E_HIP;
    }
    for (char *p = buf; *p; ++p)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');   // (sysfs names are lower case)
    return HPN_OK;
}

const char *hpn_ctx_last_error(const hpn_ctx *c) { return c ? c->err : "null context"; }

int hpn_ctx_last_kernel_ms(hpn_ctx *c, int family, float *ms)
{
    if (!c || !ms || family < 0 || family >= kFamCount) return HPN_E_ARG;
    if (!c->ev_valid[family]) return fail(c, HPN_E_STATE, "no launch recorded for family %d", family);
    HPN_HIP(c, hipEventSynchronize(c->ev_end[family]));
    HPN_HIP(c, hipEventElapsedTime(ms, c->ev_beg[family], c->ev_end[family]));
    return HPN_OK;
}

// ---- memory -------------------------------------------------------------------

int hpn_dev_malloc(hpn_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, HPN_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    return HPN_OK;
}

int hpn_dev_mem_info(hpn_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
{
    if (!c || !free_bytes || !total_bytes) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HPN_HIP(c, hipMemGetInfo(&f, &t));
    *free_bytes = f, *total_bytes = t;
    return HPN_OK;
}

int hpn_dev_free(hpn_ctx *c, void *dptr)
{
    if (!c) return HPN_E_ARG;
    if (dptr) HPN_HIP(c, hipFree(dptr));
    return HPN_OK;
}

int hpn_host_malloc(hpn_ctx *c, size_t bytes, void **hptr)
{
    if (!c || !hptr) return HPN_E_ARG;
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocPortable);   // pinned for every device: one reader may feed several contexts
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, HPN_E_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    return HPN_OK;
}

int hpn_host_free(hpn_ctx *c, void *hptr)
{
    if (!c) return HPN_E_ARG;
    if (hptr) HPN_HIP(c, hipHostFree(hptr));
    return HPN_OK;
}

int hpn_memcpy_h2d(hpn_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return HPN_E_ARG;
    if (bytes) HPN_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return HPN_OK;
}

int hpn_memcpy_d2h(hpn_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return HPN_E_ARG;
    if (bytes) HPN_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return HPN_OK;
}

int hpn_memcpy_d2d(hpn_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return HPN_E_ARG;
    if (bytes) HPN_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return HPN_OK;
}

// ---- fastq tally -----------------------------------------------------------------

}  // extern "C"

// shared with the raw-text front end (hpn_text.hip)
int hpn::tally_launch(hpn_ctx *c, const uint8_t *d_qual, const uint8_t *d_base, const uint64_t *d_off, uint64_t n,
                      uint64_t approx_bytes, uint32_t flags)
{
    const bool qh = flags & HPN_TALLY_QUAL_HIST, nh = flags & HPN_TALLY_NUC_HIST;
    if (nh && !d_base) return fail(c, HPN_E_ARG, "HPN_TALLY_NUC_HIST needs the base array");
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamTally], c->stream));
    // Without the Quality matrix the flat scan gives everything fastq_count prints;
    // with it, the histogram kernel also produces SeqLen / sum / Q20 / Q30 in its pass.
    if (!qh) HPN_HIP(c, launch_tally_scan(d_qual, d_off, n, approx_bytes, c->d_acc, c->d_sched(), c->n_cu, c->stream));
    if (qh || nh) HPN_HIP(c, launch_tally_hist(d_qual, d_base, d_off, n, qh, nh, c->d_acc, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamTally], c->stream));
    c->ev_valid[kFamTally] = true;
    return HPN_OK;
}

extern "C" {

int hpn_fastq_tally_dev(hpn_ctx *c, const uint8_t *d_qual, const uint8_t *d_base, const uint64_t *d_off,
                        uint64_t n, uint32_t flags)
{
    if (!c || !d_qual || !d_off) return HPN_E_ARG;
    if (flags & ~(HPN_TALLY_QUAL_HIST | HPN_TALLY_NUC_HIST)) return fail(c, HPN_E_ARG, "unknown flags 0x%x", flags);
    HPN_HIP(c, hipSetDevice(c->device));
    // byte count is only known on the device: size the grid for a typical short read
    return tally_launch(c, d_qual, d_base, d_off, n, n * 160, flags);
}

int hpn_fastq_tally_fetch(hpn_ctx *c, hpn_tally *acc)
{
    if (!c || !acc) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    HPN_HIP(c, hipMemcpyAsync(c->h_acc, c->d_acc, HPN_TALLY_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemsetAsync(c->d_acc, 0, HPN_TALLY_WORDS * sizeof(u64), c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    const u64 *h = c->h_acc;
    if (h[HPN_TALLY_W_BAD])
        return fail(c, HPN_E_DOMAIN,
                    "batch holds a read of length >= %d or a quality byte >= %d (reference: out-of-bounds write)",
                    HPN_LEN_BINS, HPN_QUAL_ROWS);
    for (int i = 0; i < HPN_LEN_BINS; ++i) acc->seqlen[i] += h[HPN_TALLY_W_SEQLEN + i];
    acc->total += h[HPN_TALLY_W_TOTAL];
    acc->q20 += h[HPN_TALLY_W_Q20];
    acc->q30 += h[HPN_TALLY_W_Q30];
    if (acc->qual_hist)
        for (int i = 0; i < HPN_QUAL_ROWS * HPN_LEN_BINS; ++i) acc->qual_hist[i] += h[HPN_TALLY_W_QUAL + i];
    if (acc->nuc_hist)
        for (int i = 0; i < HPN_NUC_CODES * HPN_LEN_BINS; ++i) acc->nuc_hist[i] += h[HPN_TALLY_W_NUC + i];
    return HPN_OK;
}

int hpn_fastq_tally_devptr(hpn_ctx *c, uint64_t **d_acc)
{
    if (!c || !d_acc) return HPN_E_ARG;
    *d_acc = (uint64_t *)c->d_acc;
    return HPN_OK;
}

int hpn_fastq_tally(hpn_ctx *c, const uint8_t *qual, const uint8_t *base, const uint64_t *off, uint64_t n,
                    hpn_tally *acc)
{
    if (!c || !off || !acc) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t b0 = off[0], b1 = off[n];
    if (b1 < b0) return fail(c, HPN_E_ARG, "offsets decrease");
    const uint64_t nbytes = b1 - b0;
    if (nbytes && !qual) return fail(c, HPN_E_ARG, "qual is NULL");
    const uint32_t flags = (acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0) | (acc->nuc_hist && base ? HPN_TALLY_NUC_HIST : 0);
    // Stage [b0, b1) so that device address == scratch + 16 + (host offset - b0) keeps
    // the host offsets valid unchanged: the kernels index qual[off[i]].
    const size_t pad = 16 + (b0 & 15);
    int rc;
    if ((rc = scratch_reserve(c, c->s_a, nbytes + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->s_c, (n + 1) * sizeof(uint64_t))) != HPN_OK) return rc;
    if (flags & HPN_TALLY_NUC_HIST)
        if ((rc = scratch_reserve(c, c->s_b, nbytes + 64)) != HPN_OK) return rc;
    uint8_t *dq = (uint8_t *)c->s_a.p + pad, *db = nullptr;
    if (nbytes) HPN_HIP(c, hipMemcpyAsync(dq, qual + b0, nbytes, hipMemcpyHostToDevice, c->stream));
    if (flags & HPN_TALLY_NUC_HIST) {
        db = (uint8_t *)c->s_b.p + pad;
        if (nbytes) HPN_HIP(c, hipMemcpyAsync(db, base + b0, nbytes, hipMemcpyHostToDevice, c->stream));
    }
    HPN_HIP(c, hipMemcpyAsync(c->s_c.p, off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    rc = tally_launch(c, dq - b0, db ? db - b0 : nullptr, (const uint64_t *)c->s_c.p, n, nbytes, flags);
    if (rc != HPN_OK) return rc;
    return hpn_fastq_tally_fetch(c, acc);
}

// ---- synthetic input ----------------------------------------------------------------

int hpn_synth_fastq_dev(hpn_ctx *c, uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint8_t *d_qual,
                        uint8_t *d_base, uint64_t *d_off)
{
    if (!c || !d_qual || !d_off) return HPN_E_ARG;
    if (len == 0 || len >= HPN_LEN_BINS) return fail(c, HPN_E_ARG, "read length %u outside 1..511", len);
    if (((uintptr_t)d_qual & 15) || ((uintptr_t)d_base & 15)) return fail(c, HPN_E_ARG, "output arrays must be 16-byte aligned");
    HPN_HIP(c, hipSetDevice(c->device));
    HPN_HIP(c, launch_synth_fastq(seed, first, n, len, d_qual, d_base, d_off, c->n_cu, c->stream));
    return HPN_OK;
}

}  // extern "C"
