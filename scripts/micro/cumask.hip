// cumask.hip -- do CU masks keep a bandwidth-bound kernel alive beside a kernel that fills every wave slot it may use?
// (round 6: the gzip route's histories / translation / CRC / framing beside the NEXT batch's decoders.)
//   hog:   one 64-thread workgroup per decoder slot, 6,400 bytes of LDS each, 80 VGPRs-ish of ALU work for ~40 ms (what k_gz_sym_inflate looks like to the dispatcher)
//   copy:  4 GiB read + 4 GiB written, 256 threads per workgroup, no LDS
// prints the copy's time alone, on a stream masked to the last `b` CUs alone, and beside the hog with / without masks.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/cumask.hip -o /tmp/cumask && /tmp/cumask [cus_for_the_copy=32]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void hog(uint32_t *out, uint32_t iters)
{
    __shared__ uint32_t s[1600];
    uint32_t v = threadIdx.x + blockIdx.x, w = 1;
    for (uint32_t i = 0; i < iters; ++i) {
        v = v * 1664525u + 1013904223u;
        w ^= v >> 7;
        s[(v >> 20) % 1600] = w;
        w += s[(w >> 3) % 1600];
    }
    if (w == 0x12345u) out[0] = v;
}
__global__ __launch_bounds__(256) void copy(const uint4 *a, uint4 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
static double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }

int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 32;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    uint32_t ma[16] = {0}, mb[16] = {0};
    for (int i = 0; i < ncu; ++i) (i < ncu - nb ? ma : mb)[i / 32] |= 1u << (i % 32);
    hipStream_t sa, sb, ua, ub;
    CK(hipExtStreamCreateWithCUMask(&sa, (ncu + 31) / 32, ma));
    CK(hipExtStreamCreateWithCUMask(&sb, (ncu + 31) / 32, mb));
    CK(hipStreamCreateWithFlags(&ua, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ub, hipStreamNonBlocking));
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    uint4 *a, *b;
    uint32_t *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    const uint32_t iters = 300000;
    auto run_copy = [&](hipStream_t s) { hipLaunchKernelGGL(copy, dim3(ncu * 8), dim3(256), 0, s, a, b, n); };
    auto run_hog = [&](hipStream_t s) { hipLaunchKernelGGL(hog, dim3(ncu * 24), dim3(64), 0, s, o, iters); };
    auto timed = [&](const char *what, hipStream_t hs, hipStream_t cs, bool with_hog) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            if (with_hog) run_hog(hs);
            run_copy(cs);
            CK(hipStreamSynchronize(cs));
            const double tc = ms_since(t0);
            CK(hipDeviceSynchronize());
            const double all = ms_since(t0);
            if (rep) printf("%-58s copy done after %8.2f ms (%6.0f GB/s), everything after %8.2f ms\n", what, tc, 2.0 * bytes / tc / 1e6, all);
        }
        return 0;
    };
    printf("%d CUs; mask A = the first %d, mask B = the last %d\n", ncu, ncu - nb, nb);
    if (timed("copy alone, unmasked stream", ua, ub, false)) return 1;
    if (timed("copy alone, stream masked to B", ua, sb, false)) return 1;
    {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        run_hog(ua);
        CK(hipDeviceSynchronize());
        printf("%-58s %8.2f ms\n", "hog alone, unmasked", ms_since(t0));
        t0 = std::chrono::steady_clock::now();
        run_hog(sa);
        CK(hipDeviceSynchronize());
        printf("%-58s %8.2f ms\n", "hog alone, masked to A", ms_since(t0));
    }
    if (timed("hog unmasked + copy unmasked (two streams)", ua, ub, true)) return 1;
    if (timed("hog masked to A + copy masked to B", sa, sb, true)) return 1;
    if (timed("hog unmasked + copy masked to B", ua, sb, true)) return 1;
    return 0;
}
