// Does LDS take unaligned 4- / 8-byte stores on gfx950?  (k_bedgraph_text would build lines with them.)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_unaligned.hip -o /tmp/lds_unaligned && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

__global__ void k(uint8_t *out, int stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0xee;
    __syncthreads();
    const int at = 3 + threadIdx.x * stride;            // odd addresses
    const uint32_t v = 0x03020100u + 0x04040404u * threadIdx.x;
    __builtin_memcpy(s + at, &v, 4);                    // unaligned ds_write_b32 (or byte stores, if the compiler splits it)
    const unsigned long long w = 0x8877665544332211ull;
    __builtin_memcpy(s + 4096 + 5 + threadIdx.x * 9, &w, 8);
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) out[i] = s[i];
}

__global__ void k_asm(uint8_t *out, int stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0xee;
    __syncthreads();
    const uint32_t at = (uint32_t)(uintptr_t)(s) + 3u + threadIdx.x * stride;
    const uint32_t v = 0x03020100u + 0x04040404u * threadIdx.x;
    asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(at), "v"(v) : "memory");
    const uint32_t at2 = (uint32_t)(uintptr_t)(s) + 4096u + 5u + threadIdx.x * 9u;
    const unsigned long long w = 0x8877665544332211ull;
    asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(at2), "v"(w) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) out[i] = s[i];
}

int main()
{
    uint8_t *d;
    hipMalloc(&d, 8192);
    std::vector<uint8_t> h(8192), want(8192, 0xee);
    const int stride = 7;
    for (int t = 0; t < 64; ++t) {
        const uint32_t v = 0x03020100u + 0x04040404u * t;
        memcpy(&want[3 + t * stride], &v, 4);
        const unsigned long long w = 0x8877665544332211ull;
        memcpy(&want[4096 + 5 + t * 9], &w, 8);
    }
    for (int variant = 0; variant < 2; ++variant) {
        if (variant == 0) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        else hipLaunchKernelGGL(k_asm, dim3(1), dim3(64), 0, 0, d, stride);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 8192; ++i) bad += h[i] != want[i];
        printf("%s: %s, %d bytes differ\n", variant ? "ds_write_b32/b64 at odd addresses (asm)" : "memcpy (compiler's choice)", hipGetErrorString(e), bad);
    }
    return 0;
}
