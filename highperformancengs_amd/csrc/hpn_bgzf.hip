// hpn_bgzf.hip -- C ABI of the device-side BGZF inflater (kernels/bgzf_inflate.hip).
#include "hpn_ctx.hpp"

namespace hpn {
hipError_t launch_bgzf_inflate(const uint8_t *d_comp, const void *d_blocks, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_status,
                               uint32_t *d_ticket, int n_cu, hipStream_t st);
}

using namespace hpn;

extern "C" {

int hpn_bgzf_inflate_dev(hpn_ctx *c, const uint8_t *d_comp, const hpn_bgzf_block *d_blocks, uint64_t n_blocks, uint8_t *d_out,
                         uint32_t *d_status)
{
    if (!c || (n_blocks && (!d_comp || !d_blocks || !d_out || !d_status))) return HPN_E_ARG;
    if (n_blocks > 0xffffffffull) return fail(c, HPN_E_ARG, "too many blocks");
    HPN_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = scratch_reserve(c, c->b_ticket, 64)) != HPN_OK) return rc;      // the kernel's block counter
    HPN_HIP(c, hipEventRecord(c->ev_beg[kFamInflate], c->stream));
    HPN_HIP(c, launch_bgzf_inflate(d_comp, d_blocks, (uint32_t)n_blocks, d_out, d_status, (uint32_t *)c->b_ticket.p, c->n_cu, c->stream));
    HPN_HIP(c, hipEventRecord(c->ev_end[kFamInflate], c->stream));
    c->ev_valid[kFamInflate] = true;
    return HPN_OK;
}

}  // extern "C"
