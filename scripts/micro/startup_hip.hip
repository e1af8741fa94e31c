// startup_hip.hip -- where a process's fixed HIP cost goes: stamps around the first runtime calls.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/startup_hip.hip -o /tmp/startup_hip && /tmp/startup_hip
#include <hip/hip_runtime.h>
#include <sys/prctl.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
__global__ void k(int *p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv)
{
    if (getenv("NO_THP")) prctl(PR_SET_THP_DISABLE, 1, 0, 0, 0);     // (does the runtime's start wait for huge pages to be made?)
    const double t0 = now();
    double t = t0;
    auto stamp = [&](const char *w) { const double n = now(); printf("%-28s %7.1f ms  (at %7.1f)\n", w, (n - t) * 1e3, (n - t0) * 1e3); t = n; };
    (void)hipInit(0); stamp("hipInit");
    int n = 0; (void)hipGetDeviceCount(&n); stamp("hipGetDeviceCount");
    (void)hipSetDevice(0); stamp("hipSetDevice");
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0); stamp("hipGetDeviceProperties");
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); stamp("hipStreamCreate");
    int *d; (void)hipMalloc((void **)&d, 1 << 20); stamp("hipMalloc 1 MiB");
    void *h; (void)hipHostMalloc(&h, 64 << 20, hipHostMallocDefault); stamp("hipHostMalloc 64 MiB");
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); (void)hipStreamSynchronize(s); stamp("first kernel + sync");
    void *big; (void)hipMalloc(&big, (size_t)12 << 30); stamp("hipMalloc 12 GiB");
    (void)hipMemsetAsync(big, 0, (size_t)12 << 30, s); (void)hipStreamSynchronize(s); stamp("memset 12 GiB");
    void *h2; (void)hipHostMalloc(&h2, (size_t)264 << 20, hipHostMallocDefault); stamp("hipHostMalloc 264 MiB");
    hipStream_t s2; (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); stamp("second stream");
    fflush(stdout);
    if (argc > 1) { (void)hipFree(big); stamp("hipFree 12 GiB"); fflush(stdout); }
    _exit(0);
}
