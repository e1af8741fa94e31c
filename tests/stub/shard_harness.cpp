// shard_harness.cpp -- the host side of "one FASTQ over several lanes" (csrc/host/text_shard.hpp: reader, dispatcher, lanes, the
// board of line counts, the ordered writer) without a GPU, for ThreadSanitizer / AddressSanitizer (scripts/sanitize_shard.sh).
//
// The handful of ABI calls the route makes are stood in for by plain CPU code in this file -- a context is a struct, the two
// halves of a piece are evaluated by the rule of include/hpngs.h (a record starts behind local line end i iff
// (lines_before + i + 1) % 4 == 0; a piece owns the records that start in it) -- so the threads, queues and hand-overs are the
// product's own and the sanitizer sees them.  Test infrastructure: nothing of this is shipped or linked into libhpngs.
//   shard_harness count FILE LANES   -> "reads bases"            (must equal a serial pass over the file)
//   shard_harness trim  FILE LANES S E OUT  -> trimmed text in OUT (must equal the serial cut)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hpngs.h"

struct hpn_ctx {
    int device = 0;
    // the piece between its two halves
    std::vector<uint8_t> text;
    uint32_t head = 0;
    uint64_t own = 0;
    int last = 0;
    bool pending = false;
    // what the lane has counted (the "device accumulators")
    uint64_t reads = 0, bases = 0;
    char err[64] = "";
};

extern "C" {
int hpn_ctx_create(int device, hpn_ctx **ctx)
{
    *ctx = new hpn_ctx;
    (*ctx)->device = device;
    return HPN_OK;
}
const char *hpn_ctx_last_error(const hpn_ctx *c) { return c->err; }
int hpn_ctx_device(const hpn_ctx *, int *device) { return *device = 0, HPN_OK; }
int hpn_ctx_pci_address(const hpn_ctx *, char *, int) { return HPN_E_NODEVICE; }      // (no device: bind_thread_near leaves the threads where they are)
int hpn_host_malloc(hpn_ctx *, size_t bytes, void **p)
{
    *p = malloc(bytes ? bytes : 1);
    return *p ? HPN_OK : HPN_E_NOMEM;
}
int hpn_host_free(hpn_ctx *, void *p)
{
    free(p);
    return HPN_OK;
}
int hpn_fastq_text_piece_lines(hpn_ctx *c, const void *text, uint64_t nbytes, uint32_t head, uint64_t own_bytes, int last, hpn_text_piece *out)
{
    c->text.assign((const uint8_t *)text, (const uint8_t *)text + nbytes);
    c->head = head, c->own = own_bytes, c->last = last, c->pending = true;
    memset(out, 0, sizeof *out);
    const uint64_t lim = last ? nbytes : head + own_bytes;     // line ends in front of lim - 1 are this piece's to count
    for (uint64_t k = 0; k + 1 < lim; ++k) out->n_lines += c->text[k] == '\n';
    return HPN_OK;
}
// HPN_STUB_IRREGULAR_SLEEP_MS: the lane that finds its piece irregular reports it that much later, so the lanes beside it are
// already blocked behind its sequence number when the route stops (the hang the round-3 advisor reproduced in fastq_trim).
static void slow_irregular()
{
    if (const char *e = getenv("HPN_STUB_IRREGULAR_SLEEP_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(e)));
}
// records owned by the pending piece: (start, line ends e0..e3) by the rule of include/hpngs.h; false: a record does not end in the text
static bool piece_records(hpn_ctx *c, uint64_t lines_before, std::vector<uint64_t> &starts, std::vector<uint64_t> &ends)
{
    const std::vector<uint8_t> &t = c->text;
    const uint64_t n = t.size(), lim = c->last ? n : c->head + c->own;
    std::vector<uint64_t> nl;
    for (uint64_t k = 0; k < n; ++k)
        if (t[k] == '\n') nl.push_back(k);
    if (c->last && n && t[n - 1] != '\n') nl.push_back(n);   // the virtual final newline
    long first = c->head ? (long)((4 - ((lines_before + 1) & 3)) & 3) : -1;
    for (long i = first;; i += 4) {
        const uint64_t start = i < 0 ? 0 : nl.size() > (size_t)i ? nl[(size_t)i] + 1 : n + 1;
        if (start >= lim || (i >= 0 && (size_t)i >= nl.size())) break;
        if ((size_t)(i + 4) >= nl.size()) return false;
        starts.push_back(start);
        for (int k = 1; k <= 4; ++k) ends.push_back(nl[(size_t)(i + k)]);
    }
    return true;
}
int hpn_fastq_text_piece_count(hpn_ctx *c, uint64_t lines_before, uint32_t, hpn_text_info *info)
{
    memset(info, 0, sizeof *info);
    if (!c->pending) return HPN_E_STATE;
    c->pending = false;
    std::vector<uint64_t> st, en;
    if (!piece_records(c, lines_before, st, en)) {
        slow_irregular();
        info->irregular = HPN_TEXT_PARTIAL;
        return HPN_OK;
    }
    for (size_t r = 0; r < st.size(); ++r) {
        const uint64_t len = en[4 * r + 1] - en[4 * r] - 1;
        c->reads += 1, c->bases += len;
        info->n_bytes += len;
    }
    info->n_records = st.size();
    return HPN_OK;
}
int hpn_fastq_text_piece_trim(hpn_ctx *c, uint64_t lines_before, int32_t S, int32_t E, void *out_text, uint64_t out_cap, hpn_text_info *info)
{
    memset(info, 0, sizeof *info);
    if (!c->pending) return HPN_E_STATE;
    c->pending = false;
    std::vector<uint64_t> st, en;
    if (!piece_records(c, lines_before, st, en)) {
        slow_irregular();
        info->irregular = HPN_TEXT_PARTIAL;
        return HPN_OK;
    }
    std::string o;
    const uint8_t *t = c->text.data();
    for (size_t r = 0; r < st.size(); ++r) {
        const uint64_t e0 = en[4 * r], e1 = en[4 * r + 1], e2 = en[4 * r + 2];
        const uint64_t len = e1 - e0 - 1, b = (uint64_t)S < len ? (uint64_t)S : len, e = (uint64_t)E < len ? (uint64_t)E : len;
        o.append((const char *)t + st[r], e0 + 1 - st[r]);
        o.append((const char *)t + e0 + 1 + b, e > b ? e - b : 0);
        o += "\n+\n";
        o.append((const char *)t + e2 + 1 + b, e > b ? e - b : 0);
        o += "\n";
    }
    if (o.size() > out_cap) return HPN_E_CAPACITY;
    memcpy(out_text, o.data(), o.size());
    info->n_records = st.size(), info->n_bytes = o.size();
    return HPN_OK;
}
int hpn_fastq_tally_fetch(hpn_ctx *c, hpn_tally *acc)
{
    acc->seqlen[0] += c->reads, acc->total += c->bases;     // (reads ride in seqlen[0]: this harness only compares the two sums)
    c->reads = c->bases = 0;
    return HPN_OK;
}
int hpn_fastq_tally_devptr(hpn_ctx *, uint64_t **p)
{
    *p = nullptr;
    return HPN_OK;
}
int hpn_comm_init_all(hpn_ctx **, int) { return HPN_E_RCCL; }     // no RCCL here: the lanes' sums are added on the host
int hpn_allreduce_u64_all(hpn_ctx **, uint64_t **, int, size_t) { return HPN_E_RCCL; }
int hpn_comm_count(hpn_ctx *, int *) { return HPN_E_STATE; }
}

#include "../../highperformancengs_amd/csrc/host/text_shard.hpp"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    const int lanes = atoi(argv[3]);
    hpn_ctx *own = nullptr;
    hpn_ctx_create(0, &own);
    hpn::LaneGroup g(own, 0, 0, 1, lanes);      // one "device": the lanes share it, the sum is the host's
    if (mode == "count") {
        hpn_tally acc;
        memset(&acc, 0, sizeof acc);
        bool irregular = false;
        const int rc = hpn::tally_text_sharded(g, argv[2], &acc, &irregular);
        printf("%d %d %llu %llu\n", rc, (int)irregular, (unsigned long long)acc.seqlen[0], (unsigned long long)acc.total);
        return 0;
    }
    if (mode == "trim" && argc >= 7) {
        FILE *out = fopen(argv[6], "wb");
        unsigned long reads = 0;
        bool irregular = false;
        const int rc = hpn::trim_text_sharded(g, argv[2], atoi(argv[4]), atoi(argv[5]), out, &reads, &irregular);
        fclose(out);
        printf("%d %d %lu\n", rc, (int)irregular, reads);
        return 0;
    }
    return 2;
}
