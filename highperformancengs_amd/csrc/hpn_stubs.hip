// hpn_stubs.hip -- entry points whose kernels are not written yet (temporary).
#include "hpn_ctx.hpp"
using namespace hpn;
#define TODO(c) return fail((c), HPN_E_STATE, "%s: not implemented in this build", __func__)
extern "C" {
int hpn_fastq_trim(hpn_ctx *c, const uint8_t *, const uint8_t *, const uint64_t *, uint64_t, int32_t, int32_t, uint8_t *, uint8_t *, uint64_t *) { TODO(c); }
int hpn_fastq_trim_dev(hpn_ctx *c, const uint8_t *, const uint8_t *, const uint64_t *, uint64_t, int32_t, int32_t, uint8_t *, uint8_t *, uint64_t *) { TODO(c); }
int hpn_depth_begin(hpn_ctx *c, int32_t, uint32_t, uint32_t) { TODO(c); }
int hpn_depth_add(hpn_ctx *c, const hpn_bam_batch *) { TODO(c); }
int hpn_depth_add_dev(hpn_ctx *c, const hpn_bam_batch *) { TODO(c); }
int hpn_depth_finish(hpn_ctx *c, uint32_t, hpn_run *, uint64_t, uint64_t *, uint64_t *) { TODO(c); }
int hpn_window_begin(hpn_ctx *c, int32_t, const uint64_t *, uint32_t) { TODO(c); }
int hpn_window_add(hpn_ctx *c, const hpn_bam_batch *) { TODO(c); }
int hpn_window_add_dev(hpn_ctx *c, const hpn_bam_batch *) { TODO(c); }
int hpn_window_finish(hpn_ctx *c, uint32_t *, uint64_t *, uint32_t *, uint8_t *, uint64_t *) { TODO(c); }
}
