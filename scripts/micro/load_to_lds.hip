// Does global_load_lds_dwordx4 (gfx950) put lane i's 16 bytes at LDS base + 16 i, from unaligned global addresses?  And how many
// clocks until the data can be read (vmcnt)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
__global__ void k(const unsigned char *g, unsigned char *o)
{
    __shared__ __attribute__((aligned(16))) unsigned char s[4][1024];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (w * 64 + lane) * 75 + 3),
                                     (__attribute__((address_space(3))) void *)&s[w][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int b = 0; b < 16; ++b) o[threadIdx.x * 16 + b] = s[w][lane * 16 + b];
}
int main()
{
    unsigned char *g, *o, h[256 * 75 + 64], r[256 * 16];
    for (size_t i = 0; i < sizeof h; ++i) h[i] = (unsigned char)(i * 7 + (i >> 8));
    if (hipMalloc(&g, sizeof h) != hipSuccess || hipMalloc(&o, sizeof r) != hipSuccess) return 1;
    (void)hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o);
    (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t)
        if (memcmp(r + t * 16, h + t * 75 + 3, 16)) ++bad;
    printf("global_load_lds_dwordx4: lane i's 16 bytes at base + 16 i, unaligned sources: %s (%d of 256 lanes differ)\n", bad ? "NO" : "yes", bad);
    return 0;
}
