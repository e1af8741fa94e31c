// How many single-wave workgroups with S bytes of LDS does a CU of gfx950 hold?  Every workgroup spins ~1 ms; a launch of
// N x 256 of them takes ~1 ms if they are all resident and ~2 ms if some have to wait for a slot.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void spin(unsigned long long ticks, unsigned *sink)
{
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 63] == 12345u) *sink = 1;
}
int main()
{
    unsigned *d;
    hipMalloc(&d, 4);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int sizes[] = {6144, 7168, 7680, 8192, 8704, 8960, 9216};
    for (int s : sizes) {
        printf("LDS %5d B:", s);
        for (int n = 16; n <= 24; ++n) {
            hipLaunchKernelGGL(spin, dim3(n * 256), dim3(64), s, 0, 100000ull, d);   // warm
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(n * 256), dim3(64), s, 0, 100000ull, d);   // 100 MHz clock: 1 ms
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("  %d/CU %.2f", n, ms);
        }
        printf("\n");
    }
    return 0;
}
