// cpus.hpp -- how many CPUs this process can actually keep busy: the affinity mask, cut down
// to the cgroup CPU quota (a container may show 256 online CPUs and be throttled to 16).
// Thread pools sized by the online-CPU count alone oversubscribe such a box and run slower.
#pragma once
#include <dirent.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <mutex>

#include "hpngs.h"
#include "knobs.hpp"

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

// How many devices an input of `bytes` is worth.  A lane (or worker) is not free: its context's hardware queue, its upload context
// and its pinned chunks cost 60 - 100 ms, and the driver makes queues one after the other whatever thread asks
// (profiles/r05/startup_hip.txt, exit_hip.txt) -- while ONE device streams `unit` bytes in that time.  With k lanes the stream
// takes T1 / k and the set-up k x c: the sum is least at k = sqrt(T1 / c) = sqrt(bytes / unit).  (Rounds 3 - 4 gave a lane per
// 2 GiB: eight lanes for a 16 GB file that one device streams in 0.33 s.)
inline int lanes_worth(uint64_t bytes, uint64_t unit, int devices)
{
    int k = 1;
    while (k < devices && (uint64_t)(k + 1) * (uint64_t)k * unit <= bytes) ++k;      // k (k + 1) <= bytes / unit: k ~ sqrt
    return k;
}

// "64-127,192-255" -> the CPUs of that list this thread may run on
inline int parse_cpulist(const char *list, const cpu_set_t &allowed, cpu_set_t *out)
{
    CPU_ZERO(out);
    for (const char *p = list; *p && *p != '\n';) {
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, out);
        if (*e != ',') break;
        p = e + 1;
    }
    return CPU_COUNT(out);
}

// The CPUs the process could use when it started (asked for the first time on the first line of main(), before anything is bound):
// what every later "next to device k" is cut from -- a lane for a device on the OTHER socket must not be cut from the mask its
// maker's thread has meanwhile been narrowed to.
inline const cpu_set_t &initial_cpus()
{
    static const cpu_set_t m = [] {
        cpu_set_t s;
        if (sched_getaffinity(0, sizeof s, &s) != 0) {
            CPU_ZERO(&s);
            for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &s);
        }
        return s;
    }();
    return m;
}

// First line of a tool's main(), BEFORE the first runtime call: where the process can see the render nodes of devices that all sit
// next to ONE set of CPUs (/dev/dri/renderD* -> /sys/class/drm/<node>/device/local_cpulist; a one-GPU container, or GPUs of one
// socket), the main thread -- the only thread so far -- moves there, so that the runtime's own threads and the memory it
// allocates while it starts (staging buffers, signals, kernel arguments) are next to the device as well.  Binding only after the
// first context exists left a 16.5 GB file streaming in 0.33 s or 0.65 s from run to run, depending on where the process had
// happened to start; a process started on the device's socket streamed it in 0.33 s every time
// (profiles/r05/numa_pagecache_probe.txt).  Devices on both sockets, no /dev/dri, HPN_NUMA=0: nothing happens.
inline void bind_before_runtime()
{
    const char *off = getenv("HPN_NUMA");
    if (off && off[0] == '0') return;
    const cpu_set_t allowed = initial_cpus();
    cpu_set_t all_near;
    CPU_ZERO(&all_near);
    int nodes = 0;
    if (DIR *d = opendir("/dev/dri")) {
        while (struct dirent *e = readdir(d)) {
            if (strncmp(e->d_name, "renderD", 7) != 0) continue;
            char path[320], list[4096];
            snprintf(path, sizeof path, "/sys/class/drm/%s/device/local_cpulist", e->d_name);
            FILE *f = fopen(path, "r");
            if (!f) continue;
            cpu_set_t one;
            if (fgets(list, sizeof list, f) && parse_cpulist(list, allowed, &one) > 0) {
                CPU_OR(&all_near, &all_near, &one);
                ++nodes;
            }
            fclose(f);
        }
        closedir(d);
    }
    const int k = CPU_COUNT(&all_near);
    if (nodes > 0 && k >= 4 && k < CPU_COUNT(&allowed)) (void)sched_setaffinity(0, sizeof all_near, &all_near);
}

// The calling thread (and the threads it starts: they inherit its mask) onto the CPUs next to the context's device --
// /sys/bus/pci/devices/<address>/local_cpulist, within what the process may use.  For the threads that FEED the device: file
// readers filling pinned chunks, uploaders.  On a two-socket MI355X box a far-socket feeder moved 38 GB/s, a near-socket one
// 51 GB/s (profiles/r05/numa_probe.txt: bam_sliding_count's ingest of a 10.6 GB BAM 0.40 -> 0.31 s).  Nothing happens where
// the list is missing, covers every usable CPU, or leaves none; HPN_NUMA=0 switches it off.
inline bool near_cpus(hpn_ctx *ctx, cpu_set_t *out)
{
    static const bool off = [] { const char *e = getenv("HPN_NUMA"); return e && e[0] == '0'; }();
    if (off || !ctx) return false;
    struct Near {
        bool known = false, use = false;
        cpu_set_t set;
    };
    static std::mutex m;
    static Near near[64];
    int device = 0;
    if (hpn_ctx_device(ctx, &device) != HPN_OK || device < 0 || device >= 64) return false;
    Near n;
    {
        std::lock_guard<std::mutex> lk(m);
        Near &slot = near[device];
        if (!slot.known) {
            slot.known = true;
            char addr[32], path[96], list[4096];
            const cpu_set_t allowed = initial_cpus();
            if (hpn_ctx_pci_address(ctx, addr, (int)sizeof addr) == HPN_OK) {
                snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", addr);
                if (FILE *f = fopen(path, "r")) {
                    if (fgets(list, sizeof list, f)) {       // "64-127,192-255"
                        const int k = parse_cpulist(list, allowed, &slot.set);
                        slot.use = k >= 4 && k < CPU_COUNT(&allowed);   // (a handful of near CPUs is less than any CPU)
                    }
                    fclose(f);
                }
            }
        }
        n = slot;
    }
    if (n.use) *out = n.set;
    return n.use;
}
inline void bind_thread_near(hpn_ctx *ctx)
{
    cpu_set_t near;
    if (near_cpus(ctx, &near)) (void)sched_setaffinity(0, sizeof near, &near);
}
// ... and EVERY thread the process has now (the runtime's own helpers among them; threads made later inherit from their makers):
// for a process whose work is on ONE device.  Measured on a two-socket box (profiles/r05/numa_pagecache_probe.txt): with only the
// feeders bound a 16.5 GB file streamed in 0.32 - 0.65 s from run to run, with the whole process on the device's socket in
// 0.324 - 0.332 s, wherever the file's page cache lay.
inline void bind_process_near(hpn_ctx *ctx)
{
    cpu_set_t near;
    if (!near_cpus(ctx, &near)) return;
    if (DIR *d = opendir("/proc/self/task")) {
        while (struct dirent *e = readdir(d)) {
            const long tid = atol(e->d_name);
            if (tid > 0) (void)sched_setaffinity((pid_t)tid, sizeof near, &near);
        }
        closedir(d);
    }
    (void)sched_setaffinity(0, sizeof near, &near);
}
// What a tool calls once it has its (first) context: the whole process when the node has one device or the run is held to one
// (HPN_DEVICE), else just this thread -- the other devices' lanes and workers bind their own threads.
inline void bind_for_device(hpn_ctx *ctx)
{
    int n = 0;
    if (getenv("HPN_DEVICE") || (hpn_device_count(&n) == HPN_OK && n == 1)) bind_process_near(ctx);
    else bind_thread_near(ctx);
}

// "0 (0000:05:00.0), 1 (0000:15:00.0)": the devices a set of lanes / workers is bound to, for the one line a multi-device
// run leaves on stderr -- so that a first run on an 8-GPU node shows at a glance that eight DEVICES did the work.
inline void describe_devices(hpn_ctx *const *ctxs, int n, char *buf, size_t cap)
{
    size_t at = 0;
    buf[0] = 0;
    for (int k = 0; k < n && at + 48 < cap; ++k) {
        int device = -1;
        char addr[32] = "?";
        if (ctxs[k]) {
            (void)hpn_ctx_device(ctxs[k], &device);
            if (hpn_ctx_pci_address(ctxs[k], addr, (int)sizeof addr) != HPN_OK) snprintf(addr, sizeof addr, "?");
        }
        at += (size_t)snprintf(buf + at, cap - at, "%s%d (%s)", k ? ", " : "", device, addr);
    }
}

}  // namespace hpn
