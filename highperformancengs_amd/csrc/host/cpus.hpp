// cpus.hpp -- how many CPUs this process can actually keep busy: the affinity mask, cut down
// to the cgroup CPU quota (a container may show 256 online CPUs and be throttled to 16).
// Thread pools sized by the online-CPU count alone oversubscribe such a box and run slower.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

namespace hpn {

inline double wall_s()  // monotonic seconds, for the HPN_TIMING diagnostics
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// HPN_TIMING=2: where the wall time of a run goes, as stamps since the process began (/proc/self/stat's start time is in clock
// ticks, too coarse: the first stamp -- the first line of main -- stands for "loaded and linked")
inline void stamp(const char *what, double arg = -1)
{
    static const bool on = [] {
        const char *e = getenv("HPN_TIMING");
        return e && e[0] == '2';
    }();
    if (!on) return;
    static const double t0 = wall_s();
    if (arg >= 0) fprintf(stderr, "[hpn t=%.3f] %s %.3f\n", wall_s() - t0, what, arg);
    else fprintf(stderr, "[hpn t=%.3f] %s\n", wall_s() - t0, what);
}

inline int usable_cpus()
{
    static const int n = [] {
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < cpus) cpus = CPU_COUNT(&set);
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[32];
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%lld", &period) != 1) period = 0;
                fclose(h);
            }
        }
        if (quota > 0 && period > 0) {
            const long lim = (long)((quota + period - 1) / period);
            if (lim < cpus) cpus = lim;
        }
        return (int)(cpus < 1 ? 1 : cpus);
    }();
    return n;
}

}  // namespace hpn
