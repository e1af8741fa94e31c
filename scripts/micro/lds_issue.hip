// Microbenchmark: K1L's inner step -- extract a byte, form the LDS address, ds_add -- from registers, no memory loads:
// how many CU clocks per LDS instruction when each is fed by its own 2 (or more) dependent VALU instructions?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

template <int kMode, int kThreads>
__global__ __launch_bounds__(kThreads) void k(uint32_t *out, const uint32_t *sym, int iters)
{
    extern __shared__ uint32_t s[];
    const int words = 128 * 256 + 64;
    for (int i = threadIdx.x; i < words; i += blockDim.x) s[i] = 0;
    __syncthreads();
    const uint32_t tid = threadIdx.x, j = tid % 38;
    uint32_t d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = sym[tid * 8 + u];
    const uint32_t b0 = 4 * j, b1 = 256 + 4 * j, b2 = 512 + 4 * j, b3 = 768 + 4 * j;
    uint8_t *base = reinterpret_cast<uint8_t *>(s);
    uint32_t seen = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t w = d[u];
            if (kMode == 0) {          // as K1L: per byte (and, shift-add), then the four adds
                const uint32_t a0 = ((w & 0x7fu) << 10) + b0, a1 = (((w >> 8) & 0x7fu) << 10) + b1, a2 = (((w >> 16) & 0x7fu) << 10) + b2,
                               a3 = (((w >> 24) & 0x7fu) << 10) + b3;
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (kMode == 1) {   // + the `seen` OR and a mask per byte (partial groups)
                seen |= w;
                const uint32_t a0 = ((w & 0x7fu & b0) << 10) + b0, a1 = (((w >> 8) & 0x7fu & b1) << 10) + b1, a2 = (((w >> 16) & 0x7fu & b2) << 10) + b2,
                               a3 = (((w >> 24) & 0x7fu & b3) << 10) + b3;
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {                   // addresses only from one op per byte (and-or on a pre-shifted word): 1.25 VALU per byte
                const uint32_t w2 = w << 2;
                const uint32_t a1 = (w2 & 0x1fc00u) | b1, a0 = ((w2 << 8) & 0x1fc00u) | b0, a2 = ((w2 >> 8) & 0x1fc00u) | b2, a3 = ((w2 >> 16) & 0x1fc00u) | b3;
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            d[u] = (w >> 8) | (w << 24);   // the same symbols at other cycles: nothing to hoist (one more VALU per dword)
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    uint32_t t = seen;
    for (int i = threadIdx.x; i < words; i += blockDim.x) t += s[i];
    if (t == 12345) out[0] = t;
}

template <int kMode, int kThreads>
static void run(const char *name, const uint32_t *sym, uint32_t *d)
{
    const int iters = 2000, blocks = 256;
    const int lds = (128 * 256 + 64) * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k<kMode, kThreads>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((k<kMode, kThreads>), dim3(blocks), dim3(kThreads), lds, 0, d, sym, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<kMode, kThreads>), dim3(blocks), dim3(kThreads), lds, 0, d, sym, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (kThreads / 64.0) * iters * 32;
    printf("%-72s %8.3f ms   %6.2f clocks per LDS instruction\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main()
{
    uint32_t *d, *sym;
    if (hipMalloc(&d, 4) != hipSuccess || hipMalloc(&sym, 4 << 20) != hipSuccess) return 1;
    uint32_t *h = (uint32_t *)malloc(4 << 20);
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < (1 << 20); ++i) {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            w |= (33 + (uint32_t)(x >> 33) % 41) << (8 * b);
        }
        h[i] = w;
    }
    (void)hipMemcpy(sym, h, 4 << 20, hipMemcpyHostToDevice);
    run<0, 1024>("2 VALU per byte (and, shift-add), 16 waves", sym, d);
    run<1, 1024>("~3 VALU per byte (+ mask, + seen), 16 waves", sym, d);
    run<2, 1024>("1.25 VALU per byte (pre-shifted word, and-or), 16 waves", sym, d);
    run<0, 512>("2 VALU per byte, 8 waves", sym, d);
    run<2, 512>("1.25 VALU per byte, 8 waves", sym, d);
    return 0;
}
