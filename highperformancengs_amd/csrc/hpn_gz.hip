// hpn_gz.hip -- C ABI of the device-side single-member gzip inflater (kernels/gz_inflate.hip).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <vector>

#include "hpn_ctx.hpp"

namespace hpn {
hipError_t launch_gz_sym_inflate(const uint8_t *d_comp, const void *d_chunks, uint32_t n_chunks, uint16_t *d_sym, uint32_t sym_cap,
                                 void *d_meta, void *d_bounds, uint32_t bounds_cap, int n_cu, hipStream_t st);
hipError_t launch_gz_windows(const uint16_t *d_sym, uint32_t sym_cap, void *d_meta, uint32_t n_chunks, const uint8_t *d_window_in,
                             uint8_t *d_windows, uint8_t *d_window_out, u64 *d_summary, void *d_groups, int n_cu, hipStream_t st);
size_t gz_groups_bytes(uint32_t n_chunks);
hipError_t launch_gz_translate(const uint16_t *d_sym, uint32_t sym_cap, const void *d_meta, uint32_t n_chunks, const uint8_t *d_windows,
                               uint8_t *d_text, hipStream_t st);
hipError_t launch_gz_find_starts(const uint8_t *d_comp, uint64_t comp_len, const void *d_slices, uint32_t n, uint64_t *d_found, int n_cu,
                                 hipStream_t st);
uint32_t crc_block_bytes();
uint32_t inflate_waves_per_cu();
hipError_t launch_crc32_blocks(const uint8_t *d_data, const void *d_blocks, uint32_t n_blocks, uint32_t *d_out, hipStream_t st);
uint32_t crc_fold_blocks(const uint32_t *crcs, uint64_t n_blocks, uint64_t total_len);
uint32_t crc_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);
}

using namespace hpn;

namespace {
constexpr uint32_t kGzBounds = 65536;   // members that may end inside one call (beyond: status 23, the caller takes another route)
}

extern "C" {

// The symbolic decode of a call's stretches needs nothing of the text in front of them; resolving the histories does (the 32 KiB
// window in front of the first stretch).  hpn_gz_inflate_begin_dev starts the first on the context's stream and returns at once;
// hpn_gz_inflate_finish_dev takes the window, resolves, translates and reports.  Between the two the caller may wait for whoever
// inflates the stretches in front of these -- on another device (host/gz_gpu.hpp: batches of one file over several contexts).
int hpn_gz_inflate_begin_dev(hpn_ctx *c, const uint8_t *d_comp, const hpn_gz_chunk *d_chunks, uint32_t n_chunks, uint32_t sym_cap)
{
    if (!c || (n_chunks && (!d_comp || !d_chunks))) return HPN_E_ARG;
    if (n_chunks > 65535u || (sym_cap & 7u) || (n_chunks && !sym_cap)) return fail(c, HPN_E_ARG, "hpn_gz_inflate_dev: n_chunks <= 65535, sym_cap a multiple of 8");
    HPN_HIP(c, hipSetDevice(c->device));
    c->gz_pending = false;
    c->gz_members.clear();
    c->gz_n_chunks = n_chunks, c->gz_sym_cap = sym_cap;
    if (n_chunks) {
        int rc;
        if ((rc = scratch_reserve(c, c->g_sym, (size_t)n_chunks * sym_cap * sizeof(uint16_t) + 64)) != HPN_OK) return rc;
        if ((rc = scratch_reserve(c, c->g_meta, (size_t)n_chunks * 32)) != HPN_OK) return rc;
        if ((rc = scratch_reserve(c, c->g_windows, ((size_t)n_chunks + 1) * 32768)) != HPN_OK) return rc;
        if ((rc = scratch_reserve(c, c->g_groups, gz_groups_bytes(n_chunks))) != HPN_OK) return rc;     // the histories in three steps: group maps + histories
        if ((rc = scratch_reserve(c, c->g_summary, 64)) != HPN_OK) return rc;
        if ((rc = scratch_reserve(c, c->g_bounds, 16 + (size_t)kGzBounds * 16)) != HPN_OK) return rc;
        HPN_HIP(c, hipEventRecord(c->ev_beg[kFamInflate], c->stream));
        HPN_HIP(c, launch_gz_sym_inflate(d_comp, d_chunks, n_chunks, (uint16_t *)c->g_sym.p, sym_cap, c->g_meta.p, c->g_bounds.p, kGzBounds, c->n_cu, c->stream));
        HPN_HIP(c, hipEventRecord(c->ev_end[kFamInflate], c->stream));
        c->ev_valid[kFamInflate] = true;
    }
    c->gz_pending = true;
    return HPN_OK;
}

int hpn_gz_inflate_finish_dev(hpn_ctx *c, const uint8_t *d_window_in, uint8_t *d_text, uint64_t text_cap, uint8_t *d_window_out, hpn_gz_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    memset(info, 0, sizeof *info);
    if (!c->gz_pending) return fail(c, HPN_E_STATE, "hpn_gz_inflate_finish_dev without hpn_gz_inflate_begin_dev");
    HPN_HIP(c, hipSetDevice(c->device));
    const uint32_t n_chunks = c->gz_n_chunks, sym_cap = c->gz_sym_cap;
    if (n_chunks == 0) {
        c->gz_pending = false;
        if (d_window_out && d_window_in) HPN_HIP(c, hipMemcpyAsync(d_window_out, d_window_in, 32768, hipMemcpyDeviceToDevice, c->stream));
        else if (d_window_out) HPN_HIP(c, hipMemsetAsync(d_window_out, 0, 32768, c->stream));
        HPN_HIP(c, hipStreamSynchronize(c->stream));
        return HPN_OK;
    }
    const bool dbg = test_env("HPN_GZ_DEBUG") != nullptr;
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    const double t1 = now();
    uint16_t *sym = (uint16_t *)c->g_sym.p;
    HPN_HIP(c, launch_gz_windows(sym, sym_cap, c->g_meta.p, n_chunks, d_window_in, (uint8_t *)c->g_windows.p, d_window_out,
                                 (u64 *)c->g_summary.p, c->g_groups.p, c->n_cu, c->stream));
    u64 summary[4] = {0, 0, 0, 0};
    uint32_t n_bounds = 0;
    HPN_HIP(c, hipMemcpyAsync(summary, c->g_summary.p, sizeof summary, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(&n_bounds, c->g_bounds.p, 4, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    if (n_bounds && !summary[1]) {   // members that ended inside stretches: where in this call's text, and their ISIZE
        if (n_bounds > kGzBounds) n_bounds = kGzBounds;
        struct Meta { uint32_t n_out, status, final_block, reserved; uint64_t end_bit, text_off; };
        struct Bound { uint32_t chunk, n_out, isize, crc; };
        std::vector<Meta> metas(n_chunks);
        std::vector<Bound> bounds(n_bounds);
        HPN_HIP(c, hipMemcpyAsync(metas.data(), c->g_meta.p, (size_t)n_chunks * 32, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipMemcpyAsync(bounds.data(), (const uint8_t *)c->g_bounds.p + 16, (size_t)n_bounds * 16, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipStreamSynchronize(c->stream));
        // stream order = (stretch, symbols the stretch had written at the member's end); an EMPTY member (ISIZE 0,
        // `cat a.gz empty.gz b.gz`) ends at the same text offset as its predecessor, so text_end alone does not order them
        std::stable_sort(bounds.begin(), bounds.end(), [](const Bound &a, const Bound &b) { return a.chunk != b.chunk ? a.chunk < b.chunk : a.n_out < b.n_out; });
        for (const Bound &b : bounds)
            if (b.chunk < n_chunks) c->gz_members.push_back(hpn_gz_member{metas[b.chunk].text_off + b.n_out, b.isize, b.crc});
    }
    const double t2 = now();
    info->n_bytes = summary[0];
    info->status = (uint32_t)summary[1], info->bad_chunk = (uint32_t)summary[2], info->final_chunk = (uint32_t)summary[3];
    if (info->final_chunk) {  // where that stretch stopped: the member's trailer
        struct { uint32_t n_out, status, final_block, reserved; uint64_t end_bit, text_off; } m;
        HPN_HIP(c, hipMemcpyAsync(&m, (const uint8_t *)c->g_meta.p + (size_t)(info->final_chunk - 1) * 32, 32, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipStreamSynchronize(c->stream));
        info->end_bit = m.end_bit;
    }
    // (the symbols stay: a call that reports HPN_E_CAPACITY may be finished again with a larger text buffer)
    if (info->status) {
        c->gz_pending = false;
        return HPN_OK;  // reported, not an API failure: the caller takes another route
    }
    if (info->n_bytes > text_cap) return fail(c, HPN_E_CAPACITY, "hpn_gz_inflate_dev: %llu bytes of text, capacity %llu", (unsigned long long)info->n_bytes, (unsigned long long)text_cap);
    if (info->n_bytes && !d_text) return HPN_E_ARG;
    HPN_HIP(c, launch_gz_translate(sym, sym_cap, c->g_meta.p, n_chunks, (const uint8_t *)c->g_windows.p, d_text, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    c->gz_pending = false;
    if (dbg) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev_beg[kFamInflate], c->ev_end[kFamInflate]);
        fprintf(stderr, "[hpn_gz] %u stretches: histories %.3f s (inflate kernel %.1f ms), translate %.3f s, %.1f MB of text\n",
                n_chunks, t2 - t1, ms, now() - t2, info->n_bytes / 1e6);
    }
    return HPN_OK;
}

int hpn_gz_inflate_dev(hpn_ctx *c, const uint8_t *d_comp, const hpn_gz_chunk *d_chunks, uint32_t n_chunks, uint32_t sym_cap,
                       const uint8_t *d_window_in, uint8_t *d_text, uint64_t text_cap, uint8_t *d_window_out, hpn_gz_info *info)
{
    if (!c || !info) return HPN_E_ARG;
    memset(info, 0, sizeof *info);
    const int rc = hpn_gz_inflate_begin_dev(c, d_comp, d_chunks, n_chunks, sym_cap);
    if (rc != HPN_OK) return rc;
    return hpn_gz_inflate_finish_dev(c, d_window_in, d_text, text_cap, d_window_out, info);
}

int hpn_gz_find_starts_dev(hpn_ctx *c, const uint8_t *d_comp, uint64_t comp_bytes, const hpn_span *slices, uint32_t n, uint64_t *found)
{
    if (!c || (n && (!d_comp || !slices || !found))) return HPN_E_ARG;
    if (n == 0) return HPN_OK;
    HPN_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = scratch_reserve(c, c->g_meta, (size_t)n * 24 + 64)) != HPN_OK) return rc;
    uint64_t *d_sl = (uint64_t *)c->g_meta.p, *d_found = d_sl + (size_t)2 * n;
    std::vector<uint64_t> lohi((size_t)2 * n);
    for (uint32_t k = 0; k < n; ++k) lohi[2 * k] = slices[k].off, lohi[2 * k + 1] = slices[k].off + slices[k].len;
    HPN_HIP(c, hipMemcpyAsync(d_sl, lohi.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HPN_HIP(c, launch_gz_find_starts(d_comp, comp_bytes, d_sl, n, d_found, c->n_cu, c->stream));
    HPN_HIP(c, hipMemcpyAsync(found, d_found, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    return HPN_OK;
}

int hpn_gz_members(hpn_ctx *c, hpn_gz_member *out, uint32_t cap, uint32_t *n)
{
    if (!c || !n || (cap && !out)) return HPN_E_ARG;
    *n = (uint32_t)c->gz_members.size();
    if (*n > cap) return HPN_E_CAPACITY;
    if (*n) memcpy(out, c->gz_members.data(), (size_t)*n * sizeof(hpn_gz_member));
    return HPN_OK;
}

int hpn_crc32_dev(hpn_ctx *c, const uint8_t *d_data, const hpn_span *spans, uint32_t n_spans, uint32_t *crc)
{
    if (!c || (n_spans && (!spans || !crc))) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    struct Block { uint64_t off; uint32_t len, reserved; };
    const uint64_t B = crc_block_bytes();
    std::vector<Block> blocks;
    for (uint32_t k = 0; k < n_spans; ++k) {
        if (spans[k].len && !d_data) return HPN_E_ARG;
        for (uint64_t at = 0; at < spans[k].len; at += B)
            blocks.push_back(Block{spans[k].off + at, (uint32_t)(spans[k].len - at < B ? spans[k].len - at : B), 0u});
    }
    if (blocks.size() > 0x7fffffffull) return fail(c, HPN_E_ARG, "hpn_crc32_dev: more than 2^31 blocks of 64 KiB");
    int rc;
    if ((rc = scratch_reserve(c, c->g_crc, blocks.size() * (sizeof(Block) + sizeof(uint32_t)) + 64)) != HPN_OK) return rc;
    uint32_t *d_out = (uint32_t *)((uint8_t *)c->g_crc.p + blocks.size() * sizeof(Block));
    std::vector<uint32_t> out(blocks.size());
    if (!blocks.empty()) {
        HPN_HIP(c, hipMemcpyAsync(c->g_crc.p, blocks.data(), blocks.size() * sizeof(Block), hipMemcpyHostToDevice, c->stream));
        HPN_HIP(c, launch_crc32_blocks(d_data, c->g_crc.p, (uint32_t)blocks.size(), d_out, c->stream));
        HPN_HIP(c, hipMemcpyAsync(out.data(), d_out, blocks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipStreamSynchronize(c->stream));
    }
    size_t at = 0;
    for (uint32_t k = 0; k < n_spans; ++k) {
        const uint64_t nb = (spans[k].len + B - 1) / B;
        crc[k] = crc_fold_blocks(out.data() + at, nb, spans[k].len);
        at += nb;
    }
    return HPN_OK;
}

int hpn_inflate_slots(hpn_ctx *c, uint32_t *n_slots)
{
    if (!c || !n_slots) return HPN_E_ARG;
    *n_slots = (uint32_t)c->n_cu * inflate_waves_per_cu();
    return HPN_OK;
}

uint32_t hpn_crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) { return crc_join(crc_a, crc_b, len_b); }

}  // extern "C"
