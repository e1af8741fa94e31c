// tally_stream.hpp -- one FASTQ stream -> hpn_tally, the body both fastq_count tools share.
//
// Stands where count_read stands (fastq_count.c:106-133): frame the stream exactly like
// the reference's 4 x gzgets loop (CountFramer) and add the per-record tally into the
// caller's accumulators -- but the tally runs on the GPU.  Two pinned batches alternate:
// while the GPU copies and scans batch k, the host frames batch k+1.
#pragma once
#include "../host/fastq_reader.hpp"
#include "hpngs.h"

namespace hpn {

constexpr size_t kBatchBytes = 64u << 20;  // quality bytes per batch
constexpr size_t kBatchRecs = 2u << 20;    // records per batch (short reads)

// Returns HPN_OK, an hpn_status, or HPN_E_DOMAIN with *too_long set when a read of 512+
// bases shows up (the reference would overrun SeqLen[512]).
inline int tally_stream(hpn_ctx *ctx, const InStream &fq, hpn_tally *acc, bool *too_long)
{
    struct Slot {
        FastqBatch b;
        void *d_qual = nullptr, *d_off = nullptr;
    } slot[2];
    int rc = HPN_OK;
    for (Slot &s : slot) {  // pinned host batches + device staging, allocated once
        void *q = nullptr, *o = nullptr;
        if ((rc = hpn_host_malloc(ctx, kBatchBytes + kLineBuf, &q)) != HPN_OK) return rc;
        if ((rc = hpn_host_malloc(ctx, (kBatchRecs + 1) * sizeof(uint64_t), &o)) != HPN_OK) return rc;
        s.b.qual = (uint8_t *)q, s.b.off = (uint64_t *)o, s.b.off[0] = 0;
        s.b.cap_bytes = kBatchBytes, s.b.cap_recs = kBatchRecs;
        if ((rc = hpn_dev_malloc(ctx, kBatchBytes + kLineBuf + 64, &s.d_qual)) != HPN_OK) return rc;
        if ((rc = hpn_dev_malloc(ctx, (kBatchRecs + 1) * sizeof(uint64_t), &s.d_off)) != HPN_OK) return rc;
    }
    const uint32_t flags = acc->qual_hist ? HPN_TALLY_QUAL_HIST : 0;
    CountFramer framer(fq);
    bool more = true, bad = false;
    int cur = 0;
    while (more && rc == HPN_OK) {
        FastqBatch &b = slot[cur].b;
        b.clear();
        more = framer.fill(b, &bad);  // the GPU is busy with the other slot meanwhile
        if (bad) {
            *too_long = true;
            rc = HPN_E_DOMAIN;
            break;
        }
        if (!b.n()) continue;
        if ((rc = hpn_ctx_sync(ctx)) != HPN_OK) break;  // the other slot's work is done: it may be refilled next
        if ((rc = hpn_memcpy_h2d(ctx, slot[cur].d_qual, b.qual, b.nbytes)) != HPN_OK) break;
        if ((rc = hpn_memcpy_h2d(ctx, slot[cur].d_off, b.off, (b.n() + 1) * sizeof(uint64_t))) != HPN_OK) break;
        rc = hpn_fastq_tally_dev(ctx, (const uint8_t *)slot[cur].d_qual, nullptr, (const uint64_t *)slot[cur].d_off, b.n(), flags);
        cur ^= 1;
    }
    if (rc == HPN_OK) rc = hpn_fastq_tally_fetch(ctx, acc);
    else hpn_ctx_sync(ctx);
    for (Slot &s : slot) {
        hpn_host_free(ctx, s.b.qual), hpn_host_free(ctx, s.b.off);
        s.b.qual = nullptr, s.b.off = nullptr;
        hpn_dev_free(ctx, s.d_qual), hpn_dev_free(ctx, s.d_off);
    }
    return rc;
}

}  // namespace hpn
