// hpn_ctx.hpp -- the per-GPU context behind the C ABI (include/hpngs.h).
#pragma once
#include "host/knobs.hpp"
#include <vector>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "hpngs.h"

struct hpn_ctx;
namespace hpn {
typedef unsigned long long u64;

// device launchers (kernels/*.hip)
hipError_t launch_tally_scan(const uint8_t *d_qual, const uint64_t *d_off, uint64_t n, uint64_t approx_bytes,
                             u64 *d_acc, u64 *d_sched, int n_cu, hipStream_t st);
hipError_t launch_tally_hist(const uint8_t *d_qual, const uint8_t *d_base, const uint64_t *d_off, uint64_t n,
                             bool qual_hist, bool nuc_hist, u64 *d_acc, int n_cu, hipStream_t st);
int tally_launch(hpn_ctx *c, const uint8_t *d_qual, const uint8_t *d_base, const uint64_t *d_off, uint64_t n,
                 uint64_t approx_bytes, uint32_t flags);
hipError_t launch_synth_fastq(uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint8_t *d_qual,
                              uint8_t *d_base, uint64_t *d_off, int n_cu, hipStream_t st);

// A device buffer that only ever grows (staging for the host-buffer entry points).
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};

enum { kFamTally = 0, kFamTrim = 1, kFamDepth = 2, kFamWindow = 3, kFamText = 4, kFamInflate = 5, kFamRaw = 6, kFamRawFields = 7, kFamCount = 8 };

}  // namespace hpn

struct hpn_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hpn::u64 *d_acc = nullptr;  // HPN_TALLY_WORDS, followed by 16 words of kernel scheduling state
    hpn::u64 *d_sched() const { return d_acc + HPN_TALLY_WORDS; }
    hpn::u64 *h_acc = nullptr;  // pinned mirror
    hpn::Scratch s_a, s_b, s_c, s_d, s_e, s_f, s_g, s_h;  // staging of host batches
    hpn::Scratch d_diff, d_runs, d_win, d_ws, d_tidx, d_text;     // bam2depth: difference array, runs, window sums, scan workspace, tile index
    hpn::Scratch d_sw, d_win_sw;                                  // the sweep's state (frontier, chain) and the window sums of what it has swept
    hpn::Scratch w_off, w_bins, w_len, w_gc, w_misc, w_todo;      // bam_sliding_count accumulators; passes left for k_window_rest
    hipEvent_t ev_beg[hpn::kFamCount] = {};
    hipEvent_t ev_end[hpn::kFamCount] = {};
    bool ev_valid[hpn::kFamCount] = {};
    // bam2depth state
    bool depth_open = false;
    int32_t depth_tid = -1;
    uint32_t depth_len = 0, depth_mask = 0;
    uint64_t depth_slots = 0;  // int32 entries in the difference array
    bool depth_scanned = false;
    uint64_t depth_nruns = 0;
    uint64_t depth_runs_cap = 0;  // entries the device runs buffer holds
    uint64_t depth_text_bytes = 0;  // bedGraph text formatted on the device (hpn_depth_bedgraph_format)
    uint64_t depth_text_formatted = 0;   // ... of the last hpn_depth_bedgraph_format (hpn_depth_bedgraph_dev: survives hpn_depth_begin)
    bool depth_sweeping = false;    // batches in coordinate order are swept as they come (k_depth_sweep)
    uint32_t depth_W = 0;           // window size the sweep takes its window sums for (0: none; hpn_depth_begin_w)
    // bam_sliding_count state
    bool win_open = false;
    int32_t win_targets = 0;
    uint32_t win_W = 0;
    uint64_t win_total = 0;
    // raw-text front end: two device slots (carry area + chunk), line index, offsets, packed lines
    hpn::Scratch t_slot[2], t_nl, t_off, t_status, t_pq, t_ps, t_out, t_carrybuf;
    uint32_t *t_state = nullptr;    // device state block of the framing kernels
    uint32_t *h_tstate = nullptr;   // pinned mirror
    bool t_open = false;
    int t_cur = 0;
    uint32_t t_carry = 0, t_tail = 0;  // carry bytes and where they start in slot[t_cur ^ 1]
    bool t_carry_saved = false;        // ... or, behind a chunk framed in place, in t_carrybuf
    // a piece between hpn_fastq_text_piece_lines and _count / _trim
    int p_state = 0, p_last = 0;
    uint32_t p_begin = 0, p_end = 0, p_limit = 0, p_head = 0, p_nl_cap = 0;
    // records indexed in place in inflated BGZF blocks (hpn_bam_raw_*)
    hpn::Scratch r_counts, r_bases, r_off, r_tid, r_pos, r_flag, r_lq, r_soff, r_info, r_list;
    std::vector<int32_t> r_h_lo, r_h_hi;      // the indexed batch's blocks on the host: smallest / largest refID, ...
    std::vector<unsigned long long> r_h_bases; // ... number of their first record (and one behind the last block's)
    hpn::Scratch g_crc;   // hpn_crc32_dev: block table + block CRCs
    hpn::Scratch b_ticket;   // hpn_bgzf_inflate_dev: the kernel's block counter
    hpn::Scratch g_sym, g_meta, g_windows, g_summary, g_bounds, g_groups;  // gzip: symbols, per-stretch results, histories, member ends
    std::vector<hpn_gz_member> gz_members;              // members that ended inside the last hpn_gz_inflate_dev call
    bool gz_pending = false;                            // between hpn_gz_inflate_begin_dev and _finish_dev
    uint32_t gz_n_chunks = 0, gz_sym_cap = 0;
    uint64_t r_n = 0;
    bool r_fields = false;  // the SoA view of the current index has been gathered
    // RCCL
    void *comm = nullptr;
    char err[512] = {0};
};

namespace hpn {

inline int fail(hpn_ctx *c, int status, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return status;
}

#define HPN_HIP(c, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return hpn::fail((c), HPN_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                             \
    } while (0)

inline int scratch_reserve(hpn_ctx *c, Scratch &s, size_t bytes)
{
    if (bytes <= s.cap) return HPN_OK;
    if (s.p) HPN_HIP(c, hipFree(s.p));
    s.p = nullptr;
    s.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&s.p, want);
    if (e != hipSuccess) {
        s.p = nullptr;
        return fail(c, HPN_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    s.cap = want;
    return HPN_OK;
}

}  // namespace hpn
