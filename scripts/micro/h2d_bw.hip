// What feeds the tools' plain-text legs: pinned host -> device copies over this box's PCIe link, and pread() out of the page cache
// into pinned memory -- the two ceilings every `extra.end_to_end` leg of bench.py is priced against (link_frac).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/h2d_bw.hip -o /tmp/h2d_bw -lpthread && /tmp/h2d_bw [scratch-file-dir]
// Prints one JSON object: h2d GB/s on 1 / 2 / 4 streams (64 MiB and 256 MiB copies), pread GB/s on 1 / 4 / 8 / 16 threads, and
// the two pipelined (pread on T threads while the previous buffer is copied) -- what text_stream.hpp's TextPump does.
#include <ctype.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

static double now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

static double pread_rate(int fd, uint8_t *dst, size_t total, int threads)
{
    const size_t piece = (total / (size_t)threads + 4095) & ~(size_t)4095;
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=] {
            const size_t lo = (size_t)t * piece, hi = lo + piece < total ? lo + piece : total;
            for (size_t at = lo; at < hi;) {
                const ssize_t k = pread(fd, dst + at, hi - at < ((size_t)8 << 20) ? hi - at : (size_t)8 << 20, (off_t)at);
                if (k <= 0) break;
                at += (size_t)k;
            }
        });
    for (auto &t : th) t.join();
    return total / (now() - t0) / 1e9;
}

// this thread (and those it starts) onto the CPUs next to device 0: /sys/bus/pci/devices/<address>/local_cpulist -- what the tools'
// feeding threads do (host/cpus.hpp: bind_thread_near).  false: no such list, or it covers every CPU.
static bool bind_near()
{
    char addr[32], path[96], list[4096];
    if (hipDeviceGetPCIBusId(addr, sizeof addr, 0) != hipSuccess) return false;
    for (char *p = addr; *p; ++p) *p = (char)tolower(*p);
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", addr);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed, near;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    CPU_ZERO(&near);
    for (char *p = list; *p && *p != '\n';) {
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, &near);
        if (*e != ',') break;
        p = e + 1;
    }
    if (CPU_COUNT(&near) < 4 || CPU_COUNT(&near) >= CPU_COUNT(&allowed)) return false;
    return sched_setaffinity(0, sizeof near, &near) == 0;
}

int main(int argc, char **argv)
{
    const size_t total = (size_t)2 << 30;      // 2 GiB moved per measurement
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    uint8_t *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void **)&h, total, hipHostMallocDefault));
    CK(hipMalloc((void **)&d, total));
    memset(h, 7, total);
    hipStream_t st[4];
    for (auto &s : st) CK(hipStreamCreate(&s));
    printf("{");
    // ---- pinned H2D ----
    for (size_t chunk : {(size_t)64 << 20, (size_t)256 << 20}) {
        for (int ns : {1, 2, 4}) {
            double best = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                int k = 0;
                for (size_t at = 0; at < total; at += chunk, ++k) CK(hipMemcpyAsync(d + at, h + at, chunk, hipMemcpyHostToDevice, st[k % ns]));
                CK(hipDeviceSynchronize());
                const double r = total / (now() - t0) / 1e9;
                best = r > best ? r : best;
            }
            printf("\"h2d_%zuMiB_x%d_GBps\": %.2f, ", chunk >> 20, ns, best);
        }
    }
    // ---- pread out of the page cache into pinned memory ----
    const std::string path = dir + "/h2d_bw.scratch";
    {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return 1;
        for (size_t at = 0; at < total; at += (size_t)64 << 20) fwrite(h, 1, (size_t)64 << 20, f);
        fclose(f);
    }
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return 1;
    pread_rate(fd, h, total, 8);      // warm the page cache
    for (int pass = 0; pass < 2; ++pass) {          // pass 1: the same with the threads on the CPUs next to the device
    const char *sfx = pass ? "_near" : "";
    if (pass && !bind_near()) break;
    for (int t : {1, 4, 8, 16}) {
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            const double r = pread_rate(fd, h, total, t);
            best = r > best ? r : best;
        }
        printf("\"pread_pagecache_x%d%s_GBps\": %.2f, ", t, sfx, best);
    }
    // ---- both at once: buffer k is copied while buffer k + 1 is read (two pinned halves of 256 MiB) ----
    for (int t : {4, 8, 16}) {
        const size_t half = (size_t)256 << 20;
        const double t0 = now();
        size_t done = 0;
        pread_rate(fd, h, half, t);
        for (size_t at = 0; at < total; at += half, done += half) {
            uint8_t *cur = h + (at / half % 2) * half, *nxt = h + ((at / half + 1) % 2) * half;
            CK(hipMemcpyAsync(d + at, cur, half, hipMemcpyHostToDevice, st[0]));
            if (at + half < total) {
                const size_t off = at + half;
                const size_t piece = half / (size_t)t;
                std::vector<std::thread> th;
                for (int k = 0; k < t; ++k)
                    th.emplace_back([=] {
                        for (size_t a = (size_t)k * piece; a < (size_t)(k + 1) * piece;) {
                            const ssize_t g = pread(fd, nxt + a, (size_t)(k + 1) * piece - a, (off_t)(off + a));
                            if (g <= 0) break;
                            a += (size_t)g;
                        }
                    });
                for (auto &x : th) x.join();
            }
            CK(hipStreamSynchronize(st[0]));
        }
        printf("\"pread_x%d_and_h2d_pipelined%s_GBps\": %.2f, ", t, sfx, done / (now() - t0) / 1e9);
    }
    }
    close(fd);
    unlink(path.c_str());
    printf("\"bytes_per_measurement\": %zu, \"cpus_online\": %ld}\n", total, sysconf(_SC_NPROCESSORS_ONLN));
    return 0;
}
