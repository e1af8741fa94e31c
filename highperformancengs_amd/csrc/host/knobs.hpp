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
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" char **environ;

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

// The shipped build says ONCE per process which HPN_* variables of its environment it does not read (round 6: A/B scripts that
// set a test switch against the shipped binaries measured the default N times and called it a sweep).  The names it knows are the
// user knobs' -- kept as suffixes, so that no test switch's name occurs in what ships -- and LIB (the Python loader's).
inline void warn_unread_env()
{
#ifndef HPN_TEST_HOOKS
    static std::atomic<bool> done{false};
    if (done.exchange(true)) return;
    static const char *const kUser[] = {"DEVICE", "NGPU", "TIMING", "FULL_EXIT", "NUMA", "READ_THREADS", "GZ_THREADS", "BGZF_THREADS", "TEXT",
                                        "BAM_GPU", "GZ_GPU", "BEDGRAPH_HOST", "DEPTH_LOOKAHEAD", "ALLREDUCE", "LIB"};
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "HPN_", 4) != 0) continue;
        const char *name = *e + 4, *eq = strchr(name, '=');
        if (!eq) continue;
        bool known = false;
        for (const char *u : kUser) known = known || (strlen(u) == (size_t)(eq - name) && !strncmp(u, name, (size_t)(eq - name)));
        if (!known)
            fprintf(stderr, "[hpn] %.*s is set but this build does not read it (test and timing switches exist only in the test-hooks build)\n",
                    (int)(eq - *e), *e);
    }
#endif
}

}  // namespace hpn
