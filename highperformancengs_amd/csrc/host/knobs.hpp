// knobs.hpp -- the environment names this code reads, in two classes.
//
// USER knobs are read with getenv() wherever they apply and are documented in README.md ("Environment"): HPN_DEVICE, HPN_NGPU,
// HPN_TIMING, HPN_FULL_EXIT, HPN_NUMA, HPN_READ_THREADS, HPN_GZ_THREADS, HPN_BGZF_THREADS, HPN_TEXT, HPN_BAM_GPU, HPN_GZ_GPU,
// HPN_BEDGRAPH_HOST, HPN_DEPTH_LOOKAHEAD, HPN_ALLREDUCE.
//
// TEST / TIMING switches -- a stand-in collective library, lanes sharing a device, routes forced on inputs too small to take
// them, chunk sizes that cut test files into many pieces, outputs dropped to time the rest -- are read with test_env(), which
// is getenv() only in a build with -DHPN_TEST_HOOKS (highperformancengs_amd/testhooks/: what tests/ and the A/B scripts run
// when they set one) and a constant nullptr in the shipped library and tools: no environment variable can swap the shipped
// library's RCCL, silence a check or drop an output.
#pragma once
#include <stdlib.h>

namespace hpn {

inline const char *test_env(const char *name)
{
#ifdef HPN_TEST_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace hpn
