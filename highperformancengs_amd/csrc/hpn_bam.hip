// hpn_bam.hip -- C ABI of the bam2depth and bam_sliding_count record loops.
#include <string.h>

#include <vector>

#include "hpn_ctx.hpp"

namespace hpn {
hipError_t launch_depth_add(const int32_t *tid_a, const int32_t *pos, const uint32_t *flag, const uint32_t *cigar_off,
                            const uint32_t *cigar, uint64_t n, int32_t tid, uint32_t flag_mask, int32_t *diff, uint64_t slots,
                            void *ws, void *sws, hpn_run *runs, uint64_t runs_cap, u64 *win_sw, uint32_t target_len, uint32_t W,
                            uint32_t *bad, int n_cu, hipStream_t st);
hipError_t launch_depth_add_raw(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, int32_t tid, uint32_t flag_mask, int32_t *diff,
                                uint64_t slots, void *ws, void *sws, hpn_run *runs, uint64_t runs_cap, u64 *win_sw, uint32_t target_len,
                                uint32_t W, uint32_t *bad, int n_cu, hipStream_t st);
size_t depth_sweep_bytes(uint64_t slots);
hipError_t depth_sweep_reset(void *sws, uint64_t slots, bool enabled, hipStream_t st);
hipError_t launch_win_from_runs(const hpn_run *runs, uint64_t n_runs, uint32_t target_len, uint32_t W, u64 *win_sum, int n_cu, hipStream_t st);
size_t depth_index_bytes(uint64_t slots);
hipError_t depth_index_reset(void *ws, uint64_t slots, hipStream_t st);
const uint32_t *depth_written(void *ws, uint64_t slots);
size_t depth_scan_bytes(uint64_t slots);
hipError_t launch_depth_scan(const int32_t *diff, const uint32_t *written, uint64_t slots, uint32_t target_len, uint32_t W, hpn_run *runs,
                             uint64_t runs_cap, u64 *win_sum, void *ws, void *sws, hipStream_t st);
uint64_t bedgraph_text_bound(uint64_t n_runs, int name_len);
size_t bedgraph_ws_bytes(uint64_t n_runs);
hipError_t launch_bedgraph_text(const hpn_run *runs, uint64_t n_runs, const char *name, int name_len, const uint8_t *d_long_name,
                                uint8_t *out, void *ws, int n_cu, hipStream_t st);
hipError_t launch_window_add(const int32_t *tid_a, const int32_t *pos, const uint32_t *flag, const int32_t *l_qseq,
                             const uint64_t *seq_off, const uint8_t *seq4, uint64_t n, uint64_t seq_end, uint32_t W, int32_t n_targets,
                             const uint64_t *win_off, uint32_t *bins, u64 *gc, uint32_t *len, uint32_t *touched,
                             u64 *n_count, uint32_t *bad, uint32_t *todo, int n_cu, hipStream_t st);
size_t window_todo_words(uint64_t n);
uint32_t depth_tile_size();
size_t raw_list_words(uint64_t stream_len, uint32_t n_blocks);
hipError_t launch_raw_count(const uint8_t *raw, const void *blocks, uint32_t n_blocks, uint64_t first_abs, const uint32_t *status,
                            uint32_t *starts, uint32_t *counts, u64 *exits, int32_t *lo, int32_t *hi, u64 *bases, int32_t *info, u64 *list,
                            uint64_t list_words, hipStream_t st);
hipError_t launch_raw_index(const void *blocks, uint32_t n_blocks, const uint32_t *counts, const u64 *bases, const u64 *list,
                            uint64_t *rec_off, hipStream_t st);
hipError_t launch_raw_fields(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, int32_t *tid, int32_t *pos, uint32_t *flag,
                             int32_t *l_qseq, uint64_t *seq_off, int n_cu, hipStream_t st);
}  // namespace hpn

using namespace hpn;

namespace {
constexpr uint64_t kPosLimit = 1ull << 28;   // int2char keeps 28 bits of a position (hashtbl.c:243-249)
constexpr uint64_t kOverhang = 1ull << 21;   // room past target_len for reads hanging over the contig end

// Copy one host array to a staging buffer on the context's stream.
template <typename T>
int stage(hpn_ctx *c, Scratch &s, const T *src, uint64_t count, const T **dev)
{
    int rc = scratch_reserve(c, s, count * sizeof(T) + 64);
    if (rc != HPN_OK) return rc;
    if (count) HPN_HIP(c, hipMemcpyAsync(s.p, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dev = (const T *)s.p;
    return HPN_OK;
}
}  // namespace

extern "C" {

// ---- bam2depth -------------------------------------------------------------------------

int hpn_depth_begin_w(hpn_ctx *c, int32_t tid, uint32_t target_len, uint32_t flag_mask, uint32_t W)
{
    if (!c || tid < 0) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    uint64_t slots = (uint64_t)target_len + 1 + kOverhang;
    if (slots > kPosLimit) slots = kPosLimit;
    const bool sweeping = !(flag_mask & HPN_DEPTH_ANY_ORDER);
    int rc = scratch_reserve(c, c->d_diff, slots * sizeof(int32_t) + 64);
    if (rc != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->d_ws, depth_scan_bytes(slots) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->d_tidx, depth_index_bytes(slots) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->d_sw, depth_sweep_bytes(slots) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->w_misc, 64)) != HPN_OK) return rc;
    // The sweep writes runs while the records come and cannot be run again: the buffer holds the most a target can have
    // (a run needs a position; 12 B x 2^28 = 3 GB for chr1, of 288 GB).  The two-pass route grows its buffer on demand.
    if (sweeping && (rc = scratch_reserve(c, c->d_runs, slots * sizeof(hpn_run) + 64)) != HPN_OK) return rc;
    if (sweeping && W) {
        const uint64_t windows = (uint64_t)target_len / W + 1;
        if ((rc = scratch_reserve(c, c->d_win_sw, windows * sizeof(u64) + 64)) != HPN_OK) return rc;
        HPN_HIP(c, hipMemsetAsync(c->d_win_sw.p, 0, windows * sizeof(u64), c->stream));
    }
    // the 1 GB array is NOT cleared: K3 writes whole tiles and a per-tile word says which ones hold data
    HPN_HIP(c, depth_index_reset(c->d_tidx.p, slots, c->stream));
    HPN_HIP(c, depth_sweep_reset(c->d_sw.p, slots, sweeping, c->stream));
    HPN_HIP(c, hipMemsetAsync(c->w_misc.p, 0, 64, c->stream));  // word 0: domain flag of the scatter
    c->depth_open = true;
    c->depth_tid = tid;
    c->depth_len = target_len;
    c->depth_mask = flag_mask & ~HPN_DEPTH_ANY_ORDER;
    c->depth_slots = slots;
    c->depth_scanned = false;
    c->depth_text_bytes = 0;
    c->depth_W = sweeping ? W : 0;
    c->depth_sweeping = sweeping;
    return HPN_OK;
}

int hpn_depth_begin(hpn_ctx *c, int32_t tid, uint32_t target_len, uint32_t flag_mask) { return hpn_depth_begin_w(c, tid, target_len, flag_mask, 0); }

int hpn_depth_progress(hpn_ctx *c, uint64_t *swept_positions)
{
    if (!c || !swept_positions) return HPN_E_ARG;
    if (!c->depth_open) return fail(c, HPN_E_STATE, "hpn_depth_progress before hpn_depth_begin");
    HPN_HIP(c, hipSetDevice(c->device));
    uint32_t frontier = 0;
    HPN_HIP(c, hipMemcpyAsync(&frontier, c->d_sw.p, sizeof frontier, hipMemcpyDeviceToHost, c->stream));   // ctl[kSwFrontier]
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    *swept_positions = (uint64_t)frontier * depth_tile_size();
    return HPN_OK;
}

static int depth_add_common(hpn_ctx *c, const hpn_bam_batch *b)
{
    if (b->n > 0xfffffff0ull) return fail(c, HPN_E_ARG, "more than 2^32 records in one hpn_depth_add call");
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamDepth], c->stream));
    HPN_HIP(c, launch_depth_add(b->tid, b->pos, b->flag, b->cigar_off, b->cigar, b->n, c->depth_tid, c->depth_mask,
                                (int32_t *)c->d_diff.p, c->depth_slots, c->d_tidx.p, c->d_sw.p, (hpn_run *)c->d_runs.p,
                                c->d_runs.cap / sizeof(hpn_run), (u64 *)c->d_win_sw.p, c->depth_len, c->depth_W, (uint32_t *)c->w_misc.p,
                                c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamDepth], c->stream));
    c->ev_valid[kFamDepth] = true;
    c->depth_scanned = false;
    return HPN_OK;
}

int hpn_depth_add_dev(hpn_ctx *c, const hpn_bam_batch *b)
{
    if (!c || !b) return HPN_E_ARG;
    if (!c->depth_open) return fail(c, HPN_E_STATE, "hpn_depth_add before hpn_depth_begin");
    if (b->n && (!b->tid || !b->pos || !b->flag || !b->cigar_off || !b->cigar)) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    return depth_add_common(c, b);
}

int hpn_depth_add(hpn_ctx *c, const hpn_bam_batch *b)
{
    if (!c || !b) return HPN_E_ARG;
    if (!c->depth_open) return fail(c, HPN_E_STATE, "hpn_depth_add before hpn_depth_begin");
    if (b->n == 0) return HPN_OK;
    if (!b->tid || !b->pos || !b->flag || !b->cigar_off || !b->cigar) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    hpn_bam_batch d = *b;
    const uint32_t c0 = b->cigar_off[0], c1 = b->cigar_off[b->n];
    int rc;
    if ((rc = stage(c, c->s_a, b->tid, b->n, &d.tid)) != HPN_OK) return rc;
    if ((rc = stage(c, c->s_b, b->pos, b->n, &d.pos)) != HPN_OK) return rc;
    if ((rc = stage(c, c->s_c, b->flag, b->n, &d.flag)) != HPN_OK) return rc;
    if ((rc = stage(c, c->s_d, b->cigar_off, b->n + 1, &d.cigar_off)) != HPN_OK) return rc;
    const uint32_t *dc = nullptr;
    if ((rc = stage(c, c->s_e, b->cigar + c0, (uint64_t)(c1 - c0), &dc)) != HPN_OK) return rc;
    d.cigar = dc - c0;  // host indices stay valid
    return depth_add_common(c, &d);
}

int hpn_depth_finish(hpn_ctx *c, uint32_t W, hpn_run *runs, uint64_t runs_cap, uint64_t *n_runs, uint64_t *win_sum)
{
    if (!c || !n_runs || W == 0) return HPN_E_ARG;
    if (!c->depth_open) return fail(c, HPN_E_STATE, "hpn_depth_finish before hpn_depth_begin");
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t windows = (uint64_t)c->depth_len / W + 1;  // bam2depth.c:326
    int rc = scratch_reserve(c, c->d_win, windows * sizeof(u64) + 64);
    if (rc != HPN_OK) return rc;
    if (c->d_runs.cap == 0 && (rc = scratch_reserve(c, c->d_runs, (1u << 20) * sizeof(hpn_run))) != HPN_OK) return rc;
    struct { uint32_t ticket, err; u64 n_runs; } head;
    uint32_t bad = 0, sw_ctl[16] = {0};
    // What the sweep did while the records came (k_depth_sweep) stands: runs and, for window size depth_W, window sums of
    // everything in front of its frontier.  k_depth_scan does the rest from the frontier on; with another window size than
    // the sweep was told, the window sums of the whole target are taken from the runs instead.
    const bool sums_ride = c->depth_sweeping && W == c->depth_W;            // the sweep's sums are for this W: k_depth_scan adds the rest
    const bool sums_from_runs = c->depth_sweeping && !sums_ride && win_sum;  // (bam2wig takes no window sums at all)
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint64_t dev_cap = c->d_runs.cap / sizeof(hpn_run);
        if (sums_ride) HPN_HIP(c, hipMemcpyAsync(c->d_win.p, c->d_win_sw.p, windows * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        else HPN_HIP(c, hipMemsetAsync(c->d_win.p, 0, windows * sizeof(u64), c->stream));
        HPN_HIP(c, hipEventRecord(c->ev_beg[kFamDepth], c->stream));
        HPN_HIP(c, launch_depth_scan((const int32_t *)c->d_diff.p, depth_written(c->d_tidx.p, c->depth_slots), c->depth_slots,
                                     c->depth_len, sums_from_runs ? 0u : W, (hpn_run *)c->d_runs.p, dev_cap, (u64 *)c->d_win.p,
                                     c->d_ws.p, c->d_sw.p, c->stream));
        HPN_HIP(c, hipEventRecord(c->ev_end[kFamDepth], c->stream));
        c->ev_valid[kFamDepth] = true;
        HPN_HIP(c, hipMemcpyAsync(&head, c->d_ws.p, sizeof head, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipMemcpyAsync(&bad, c->w_misc.p, sizeof bad, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipMemcpyAsync(sw_ctl, c->d_sw.p, sizeof sw_ctl, hipMemcpyDeviceToHost, c->stream));
        HPN_HIP(c, hipStreamSynchronize(c->stream));
        if (test_env("HPN_SWEEP_DIAG") && sw_ctl[0]) {   // (DIAG_SWEEP_STAMPS builds leave eight words per swept tile in its stretch of d_diff)
            const uint32_t nt = sw_ctl[0];
            std::vector<uint32_t> st((size_t)nt * 8);
            if (hipMemcpy2D(st.data(), 32, c->d_diff.p, (size_t)depth_tile_size() * 4, 32, nt, hipMemcpyDeviceToHost) == hipSuccess) {
                double sum[8] = {0}, mx[8] = {0};
                uint32_t n = 0, late = 0;
                for (uint32_t t = 0; t < nt; ++t) {
                    const uint32_t *o = &st[(size_t)t * 8];
                    if (o[0] != 0x5354414du) continue;
                    ++n, late += o[4] > 2000u;
                    for (int k = 1; k < 8; ++k) sum[k] += o[k], mx[k] = o[k] > mx[k] ? o[k] : mx[k];
                }
                if (n)
                    fprintf(stderr, "[sweep] %u tiles, mean (max) in us: setup %.2f (%.1f), gather %.2f (%.1f), scan %.2f, look-back %.2f (%.1f; %u tiles > 20 us; %.2f polls), "
                            "emit %.2f, flush %.2f (%.1f)\n", n, sum[1] / n / 100, mx[1] / 100, sum[2] / n / 100, mx[2] / 100, sum[3] / n / 100, sum[4] / n / 100, mx[4] / 100, late,
                            sum[7] / n, sum[5] / n / 100, sum[6] / n / 100, mx[6] / 100);
            }
        }
        if (bad)
            return fail(c, HPN_E_DOMAIN, "a CIGAR M block ends at or beyond position %llu (2^28 key limit of the reference, "
                        "or more than %llu bases past the contig end)", (unsigned long long)c->depth_slots,
                        (unsigned long long)kOverhang);
        if (sw_ctl[1])   // kSwLate
            return fail(c, HPN_E_STATE, "records of the target arrived behind positions already swept (tile %u): hpn_depth_add expects "
                        "coordinate order across calls; begin again with HPN_DEPTH_ANY_ORDER in flag_mask for input in any order", sw_ctl[0]);
        if ((head.err | sw_ctl[3]) & 1u) return fail(c, HPN_E_HIP, "prefix-scan hand-off timed out");
        if ((head.err | sw_ctl[3]) & 2u) return fail(c, HPN_E_DOMAIN, "coverage of 2^30 or more on this target");
        if (head.n_runs <= dev_cap) break;
        // device buffer too small for this chromosome (two-pass route only: the sweep's buffer holds any target): grow it and
        // redo the pass (diff is read-only)
        if (sw_ctl[0]) return fail(c, HPN_E_NOMEM, "runs buffer of the sweep too small");
        if ((rc = scratch_reserve(c, c->d_runs, head.n_runs * sizeof(hpn_run))) != HPN_OK) return rc;
    }
    if (sums_from_runs) {
        HPN_HIP(c, launch_win_from_runs((const hpn_run *)c->d_runs.p, head.n_runs, c->depth_len, W, (u64 *)c->d_win.p, c->n_cu, c->stream));
    }
    c->depth_scanned = true;
    c->depth_nruns = head.n_runs;
    *n_runs = head.n_runs;
    if (win_sum) HPN_HIP(c, hipMemcpyAsync(win_sum, c->d_win.p, windows * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    if (!runs && runs_cap == 0) {   // the caller wants the count only (it takes the text: hpn_depth_bedgraph_format)
        HPN_HIP(c, hipStreamSynchronize(c->stream));
        return HPN_OK;
    }
    if (head.n_runs > runs_cap || (!runs && head.n_runs)) {
        HPN_HIP(c, hipStreamSynchronize(c->stream));
        return fail(c, HPN_E_CAPACITY, "%llu runs, caller buffer holds %llu", (unsigned long long)head.n_runs,
                    (unsigned long long)runs_cap);
    }
    if (head.n_runs)
        HPN_HIP(c, hipMemcpyAsync(runs, c->d_runs.p, head.n_runs * sizeof(hpn_run), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    return HPN_OK;
}


int hpn_depth_bedgraph_format(hpn_ctx *c, const char *name, uint64_t *n_bytes)
{
    if (!c || !name || !n_bytes) return HPN_E_ARG;
    if (!c->depth_open || !c->depth_scanned) return fail(c, HPN_E_STATE, "hpn_depth_bedgraph_format needs a finished scan (hpn_depth_finish)");
    HPN_HIP(c, hipSetDevice(c->device));
    const size_t name_len = strlen(name);
    if (name_len > 4096) return fail(c, HPN_E_ARG, "target name of %zu characters", name_len);
    const uint64_t n = c->depth_nruns;
    c->depth_text_bytes = 0, c->depth_text_formatted = 0;
    *n_bytes = 0;
    if (n == 0) return HPN_OK;
    int rc;
    if ((rc = scratch_reserve(c, c->d_text, bedgraph_text_bound(n, (int)name_len))) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->d_ws, bedgraph_ws_bytes(n) + 4096 + 64)) != HPN_OK) return rc;
    uint8_t *d_name = (uint8_t *)c->d_ws.p + bedgraph_ws_bytes(n) + 32;   // names beyond 64 characters ride behind the workspace
    if (name_len > 64) HPN_HIP(c, hipMemcpyAsync(d_name, name, name_len, hipMemcpyHostToDevice, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamDepth], c->stream));
    HPN_HIP(c, launch_bedgraph_text((const hpn_run *)c->d_runs.p, n, name, (int)name_len, d_name, (uint8_t *)c->d_text.p, c->d_ws.p, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamDepth], c->stream));
    c->ev_valid[kFamDepth] = true;
    struct { uint32_t ticket, err; u64 total; } head;
    HPN_HIP(c, hipMemcpyAsync(&head, c->d_ws.p, sizeof head, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    if (head.err) return fail(c, HPN_E_HIP, "prefix-scan hand-off timed out");
    c->depth_text_bytes = head.total, c->depth_text_formatted = head.total;
    *n_bytes = head.total;
    return HPN_OK;
}

int hpn_depth_bedgraph_dev(hpn_ctx *c, const uint8_t **d_text, uint64_t *n_bytes)
{
    if (!c || !d_text || !n_bytes) return HPN_E_ARG;
    *d_text = (const uint8_t *)c->d_text.p, *n_bytes = c->depth_text_formatted;
    return HPN_OK;
}

int hpn_depth_bedgraph_read(hpn_ctx *c, uint64_t offset, void *dst, uint64_t nbytes)
{
    if (!c || (nbytes && !dst)) return HPN_E_ARG;
    if (offset > c->depth_text_bytes || nbytes > c->depth_text_bytes - offset)   // (offset + nbytes may wrap)
        return fail(c, HPN_E_ARG, "%llu bytes at %llu of a text of %llu", (unsigned long long)nbytes, (unsigned long long)offset,
                    (unsigned long long)c->depth_text_bytes);
    HPN_HIP(c, hipSetDevice(c->device));
    if (nbytes) HPN_HIP(c, hipMemcpyAsync(dst, (const uint8_t *)c->d_text.p + offset, nbytes, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    return HPN_OK;
}

// ---- records in place in inflated BGZF blocks -------------------------------------------------

int hpn_bam_raw_index_dev(hpn_ctx *c, const uint8_t *d_raw, const hpn_bgzf_block *d_blocks, uint64_t n_blocks, uint32_t first_off,
                          const uint32_t *d_status, hpn_raw_info *info)
{
    if (!c || !info || (n_blocks && (!d_raw || !d_blocks || !d_status)) || n_blocks > 0xffffffffull) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    memset(info, 0, sizeof *info);
    c->r_n = 0, c->r_fields = false;
    c->r_h_lo.clear(), c->r_h_hi.clear(), c->r_h_bases.clear();
    if (n_blocks == 0) return HPN_OK;
    int rc;
    // per block: (u64) where its chain leaves it | records counted | guessed start | smallest, largest refID
    if ((rc = scratch_reserve(c, c->r_counts, n_blocks * (sizeof(u64) + 4 * sizeof(uint32_t)) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->r_bases, (n_blocks + 1) * sizeof(u64))) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->r_info, 64)) != HPN_OK) return rc;
    // the records' offsets as the walk meets them (k_raw_count), block by block: sized for BGZF's blocks (64 KiB inflated at most)
    // behind a carried record of up to 4 MiB -- a walk that would write beyond it flags the call instead (the host reader's then)
    const size_t list_words = raw_list_words(n_blocks * 65536ull + ((uint64_t)4 << 20) + first_off, (uint32_t)n_blocks);
    if ((rc = scratch_reserve(c, c->r_list, list_words * sizeof(u64))) != HPN_OK) return rc;
    u64 *d_exits = (u64 *)c->r_counts.p;
    uint32_t *d_counts = (uint32_t *)(d_exits + n_blocks), *d_starts = d_counts + n_blocks;
    int32_t *d_lo = (int32_t *)(d_starts + n_blocks), *d_hi = d_lo + n_blocks;
    const int32_t init[6] = {0, INT32_MAX, INT32_MIN, 0, -1, -1};      // [4..5]: u64 ~0 = no unfinished record at the call's end
    HPN_HIP(c, hipMemcpyAsync(c->r_info.p, init, sizeof init, hipMemcpyHostToDevice, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamRaw], c->stream));
    c->ev_valid[kFamRaw] = false;
    HPN_HIP(c, launch_raw_count(d_raw, d_blocks, (uint32_t)n_blocks, first_off, d_status, d_starts, d_counts, d_exits, d_lo, d_hi,
                                (u64 *)c->r_bases.p, (int32_t *)c->r_info.p, (u64 *)c->r_list.p, list_words, c->stream));
    int32_t h[6];
    u64 total = 0;
    hpn_bgzf_block last;
    HPN_HIP(c, hipMemcpyAsync(h, c->r_info.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(&total, (const u64 *)c->r_bases.p + n_blocks, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(&last, d_blocks + (n_blocks - 1), sizeof last, hipMemcpyDeviceToHost, c->stream));
    // the blocks' refID ranges and first-record numbers (~16 bytes a block): which records a target's kernels have to look at
    c->r_h_lo.resize(n_blocks), c->r_h_hi.resize(n_blocks), c->r_h_bases.resize(n_blocks + 1);
    HPN_HIP(c, hipMemcpyAsync(c->r_h_lo.data(), d_lo, n_blocks * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(c->r_h_hi.data(), d_hi, n_blocks * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(c->r_h_bases.data(), c->r_bases.p, (n_blocks + 1) * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    u64 tail;
    memcpy(&tail, h + 4, 8);
    const u64 stream_len = last.out_off + last.out_len;
    info->flags = (uint32_t)h[0];
    info->n_records = total;
    {
        static const bool diag = [] { const char *e = getenv("HPN_TIMING"); return e && e[0] == '2'; }();
        if (diag)
            fprintf(stderr, "[hpn] record index: %llu blocks, %llu records, %u chunks of 1024 blocks walked by one lane, %u blocks walked again, flags %u\n",
                    (unsigned long long)n_blocks, (unsigned long long)total, (uint32_t)h[3] >> 20, (uint32_t)h[3] & 0xfffffu, info->flags);
    }
    info->tid_min = total ? h[1] : 0, info->tid_max = total ? h[2] : -1;
    if (tail != ~0ull && !(info->flags & 3u)) {
        if (stream_len - tail > 0xffffffffull) info->flags |= 1u;
        else info->tail_bytes = (uint32_t)(stream_len - tail);
    }
    if ((info->flags & 3u) || total == 0) return HPN_OK;  // nothing indexed: the caller decodes this file on the host (flags), or waits for more bytes
    if ((rc = scratch_reserve(c, c->r_off, total * sizeof(uint64_t))) != HPN_OK) return rc;
    HPN_HIP(c, launch_raw_index(d_blocks, (uint32_t)n_blocks, d_counts, (const u64 *)c->r_bases.p, (const u64 *)c->r_list.p,
                                (uint64_t *)c->r_off.p, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamRaw], c->stream));       // (family 6: the four index kernels and the host's look at the counts between them)
    c->ev_valid[kFamRaw] = true;
    c->r_n = total;
    return HPN_OK;
}

int hpn_depth_add_raw_dev(hpn_ctx *c, const uint8_t *d_raw)
{
    if (!c || (c->r_n && !d_raw)) return HPN_E_ARG;
    if (!c->depth_open) return fail(c, HPN_E_STATE, "hpn_depth_add before hpn_depth_begin");
    if (c->r_n > 0xfffffff0ull) return fail(c, HPN_E_ARG, "more than 2^32 records in one batch");
    HPN_HIP(c, hipSetDevice(c->device));
    // The target's records only (round 6): a batch of several targets was walked whole by k_depth_index for each of them (33 passes
    // over 8.8 M records for the 25 targets of a 10.6 GB file: 19.5 ms, the tool's largest kernel behind the inflater).  The
    // blocks whose refID range holds the target bound its records from both sides -- whatever the order of the file, every
    // record of the target lies in such a block; the kernels see that stretch of rec_off[] as their batch.
    uint64_t r0 = 0, r1 = c->r_n;
    if (!c->r_h_lo.empty() && c->r_h_bases.size() == c->r_h_lo.size() + 1) {
        const size_t nb = c->r_h_lo.size();
        size_t b0 = nb, b1 = 0;
        for (size_t b = 0; b < nb; ++b)
            if (c->r_h_lo[b] <= c->depth_tid && c->depth_tid <= c->r_h_hi[b]) {
                if (b0 == nb) b0 = b;
                b1 = b;
            }
        if (b0 == nb) return HPN_OK;                       // no record of the target in this batch
        r0 = c->r_h_bases[b0], r1 = c->r_h_bases[b1 + 1];
        if (r1 > c->r_n) r1 = c->r_n;
        if (r0 >= r1) return HPN_OK;
    }
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamDepth], c->stream));
    HPN_HIP(c, launch_depth_add_raw(d_raw, (const uint64_t *)c->r_off.p + r0, r1 - r0, c->depth_tid, c->depth_mask, (int32_t *)c->d_diff.p,
                                    c->depth_slots, c->d_tidx.p, c->d_sw.p, (hpn_run *)c->d_runs.p, c->d_runs.cap / sizeof(hpn_run),
                                    (u64 *)c->d_win_sw.p, c->depth_len, c->depth_W, (uint32_t *)c->w_misc.p, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamDepth], c->stream));
    c->ev_valid[kFamDepth] = true;
    c->depth_scanned = false;
    return HPN_OK;
}

static int window_add_common(hpn_ctx *c, const hpn_bam_batch *b, uint64_t seq_end);

int hpn_window_add_raw_dev(hpn_ctx *c, const uint8_t *d_raw)
{
    if (!c || (c->r_n && !d_raw)) return HPN_E_ARG;
    if (!c->win_open) return fail(c, HPN_E_STATE, "hpn_window_add before hpn_window_begin");
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t n = c->r_n;
    if (n == 0) return HPN_OK;
    int rc;
    if (!c->r_fields) {
        if ((rc = scratch_reserve(c, c->r_tid, n * 4)) != HPN_OK || (rc = scratch_reserve(c, c->r_pos, n * 4)) != HPN_OK ||
            (rc = scratch_reserve(c, c->r_flag, n * 4)) != HPN_OK || (rc = scratch_reserve(c, c->r_lq, n * 4)) != HPN_OK ||
            (rc = scratch_reserve(c, c->r_soff, n * 8)) != HPN_OK)
            return rc;
        HPN_HIP(c, hipEventRecord(c->ev_beg[kFamRawFields], c->stream));
        HPN_HIP(c, launch_raw_fields(d_raw, (const uint64_t *)c->r_off.p, n, (int32_t *)c->r_tid.p, (int32_t *)c->r_pos.p,
                                     (uint32_t *)c->r_flag.p, (int32_t *)c->r_lq.p, (uint64_t *)c->r_soff.p, c->n_cu, c->stream));
        HPN_HIP(c, hipEventRecord(c->ev_end[kFamRawFields], c->stream));
        c->ev_valid[kFamRawFields] = true;
        c->r_fields = true;
    }
    hpn_bam_batch b;
    memset(&b, 0, sizeof b);
    b.n = n;
    b.tid = (const int32_t *)c->r_tid.p, b.pos = (const int32_t *)c->r_pos.p, b.flag = (const uint32_t *)c->r_flag.p;
    b.l_qseq = (const int32_t *)c->r_lq.p, b.seq_off = (const uint64_t *)c->r_soff.p, b.seq4 = d_raw;
    return window_add_common(c, &b, ~0ull);   // the inflated stream is padded (hpn_bgzf_inflate_dev's contract): no limit
}

// ---- bam_sliding_count -------------------------------------------------------------------

// w_misc layout (bytes): [0] depth bad flag (u32) | [8] n_count (u64) | [16] window bad (u32) | [64..] touched (u32 x n_targets)

int hpn_window_begin(hpn_ctx *c, int32_t n_targets, const uint64_t *win_off, uint32_t W)
{
    if (!c || !win_off || n_targets <= 0 || W == 0 || W > 0x7fffffffu) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t total = win_off[n_targets];
    for (int32_t t = 0; t < n_targets; ++t)
        if (win_off[t + 1] < win_off[t]) return fail(c, HPN_E_ARG, "win_off decreases at target %d", t);
    int rc;
    if ((rc = scratch_reserve(c, c->w_off, (n_targets + 1) * sizeof(uint64_t))) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->w_bins, total * sizeof(uint32_t) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->w_len, total * sizeof(uint32_t) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->w_gc, total * sizeof(u64) + 64)) != HPN_OK) return rc;
    if ((rc = scratch_reserve(c, c->w_misc, 64 + (uint64_t)n_targets * sizeof(uint32_t))) != HPN_OK) return rc;
    HPN_HIP(c, hipMemcpyAsync(c->w_off.p, win_off, (n_targets + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HPN_HIP(c, hipMemsetAsync(c->w_bins.p, 0, total * sizeof(uint32_t), c->stream));
    HPN_HIP(c, hipMemsetAsync(c->w_len.p, 0, total * sizeof(uint32_t), c->stream));
    HPN_HIP(c, hipMemsetAsync(c->w_gc.p, 0, total * sizeof(u64), c->stream));
    HPN_HIP(c, hipMemsetAsync(c->w_misc.p, 0, 64 + (uint64_t)n_targets * sizeof(uint32_t), c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));  // win_off is caller memory
    c->win_open = true;
    c->win_targets = n_targets;
    c->win_W = W;
    c->win_total = total;
    c->depth_open = false;  // w_misc is shared with the depth state
    return HPN_OK;
}

// seq_end: first byte offset of b->seq4 that must not be read; 0 = seq_off[n] (read on the device)
static int window_add_common(hpn_ctx *c, const hpn_bam_batch *b, uint64_t seq_end)
{
    uint8_t *m = (uint8_t *)c->w_misc.p;
    const int rc = scratch_reserve(c, c->w_todo, window_todo_words(b->n) * sizeof(uint32_t));   // passes the fast kernel leaves to k_window_rest
    if (rc != HPN_OK) return rc;
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamWindow], c->stream));
    HPN_HIP(c, launch_window_add(b->tid, b->pos, b->flag, b->l_qseq, b->seq_off, b->seq4, b->n, seq_end, c->win_W, c->win_targets,
                                 (const uint64_t *)c->w_off.p, (uint32_t *)c->w_bins.p, (u64 *)c->w_gc.p,
                                 (uint32_t *)c->w_len.p, (uint32_t *)(m + 64), (u64 *)(m + 8), (uint32_t *)(m + 16),
                                 (uint32_t *)c->w_todo.p, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamWindow], c->stream));
    c->ev_valid[kFamWindow] = true;
    return HPN_OK;
}

int hpn_window_add_dev(hpn_ctx *c, const hpn_bam_batch *b)
{
    if (!c || !b) return HPN_E_ARG;
    if (!c->win_open) return fail(c, HPN_E_STATE, "hpn_window_add before hpn_window_begin");
    if (b->n && (!b->tid || !b->pos || !b->flag || !b->l_qseq || !b->seq_off || !b->seq4)) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    return window_add_common(c, b, 0);
}

int hpn_window_add(hpn_ctx *c, const hpn_bam_batch *b)
{
    if (!c || !b) return HPN_E_ARG;
    if (!c->win_open) return fail(c, HPN_E_STATE, "hpn_window_add before hpn_window_begin");
    if (b->n == 0) return HPN_OK;
    if (!b->tid || !b->pos || !b->flag || !b->l_qseq || !b->seq_off || !b->seq4) return HPN_E_ARG;
    HPN_HIP(c, hipSetDevice(c->device));
    hpn_bam_batch d = *b;
    const uint64_t s0 = b->seq_off[0], s1 = b->seq_off[b->n];
    int rc;
    if ((rc = stage(c, c->s_a, b->tid, b->n, &d.tid)) != HPN_OK) return rc;
    if ((rc = stage(c, c->s_b, b->pos, b->n, &d.pos)) != HPN_OK) return rc;
    if ((rc = stage(c, c->s_c, b->flag, b->n, &d.flag)) != HPN_OK) return rc;
    if ((rc = stage(c, c->s_d, b->l_qseq, b->n, &d.l_qseq)) != HPN_OK) return rc;
    if ((rc = stage(c, c->s_e, b->seq_off, b->n + 1, &d.seq_off)) != HPN_OK) return rc;
    // keep the host's byte alignment pattern and 16 bytes of slack on both sides (aligned vector loads)
    const size_t pad = 16 + (s0 & 15);
    if ((rc = scratch_reserve(c, c->s_f, (s1 - s0) + 64)) != HPN_OK) return rc;
    uint8_t *ds = (uint8_t *)c->s_f.p + pad;
    if (s1 > s0) HPN_HIP(c, hipMemcpyAsync(ds, b->seq4 + s0, s1 - s0, hipMemcpyHostToDevice, c->stream));
    d.seq4 = ds - s0;
    return window_add_common(c, &d, s1 + 16);   // the staging buffer has slack behind the last byte
}

int hpn_window_finish(hpn_ctx *c, uint32_t *bins, uint64_t *gc, uint32_t *len, uint8_t *touched, uint64_t *n_count)
{
    if (!c || !bins || !gc || !len) return HPN_E_ARG;
    if (!c->win_open) return fail(c, HPN_E_STATE, "hpn_window_finish before hpn_window_begin");
    HPN_HIP(c, hipSetDevice(c->device));
    const uint64_t total = c->win_total;
    std::vector<uint8_t> misc(64 + (size_t)c->win_targets * sizeof(uint32_t));
    HPN_HIP(c, hipMemcpyAsync(misc.data(), c->w_misc.p, misc.size(), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(bins, c->w_bins.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(len, c->w_len.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipMemcpyAsync(gc, c->w_gc.p, total * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HPN_HIP(c, hipStreamSynchronize(c->stream));
    c->win_open = false;
    uint32_t bad;
    memcpy(&bad, misc.data() + 16, 4);
    if (bad)
        return fail(c, HPN_E_DOMAIN, "%s", (bad & 1) ? "record with tid >= n_targets"
                                                     : "window index (unsigned short)(pos/W) beyond the target's windows "
                                                       "(the reference would write out of bounds)");
    if (n_count) memcpy(n_count, misc.data() + 8, 8);
    if (touched) {
        const uint32_t *t32 = (const uint32_t *)(misc.data() + 64);
        for (int32_t t = 0; t < c->win_targets; ++t) touched[t] = t32[t] ? 1 : 0;
    }
    return HPN_OK;
}

}  // extern "C"
