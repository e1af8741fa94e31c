// Microbenchmark: LDS atomic adds with K1L's address pattern -- word = row*256 + k*64 + j, lanes = consecutive (read, group j)
// pairs with ngr groups per read, rows taken from a table of pseudo-random symbols -- against variations of the layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

// mode 0: K1L today.  1: the same, every lane row 0 (no row spread).  2: rows spread, j = lane (64 distinct columns).
// 3: K1L with the second read's lanes moved so that equal j are 32 lanes apart (ngr 32).  4: rows from only 4 values (binned qualities).
// 5: like 0 but row stride 257 words (rows shift the bank).  6: like 4 with row stride 257.
__global__ __launch_bounds__(1024) void k(uint32_t *out, const uint32_t *sym, int iters, int mode, int ngr)
{
    extern __shared__ uint32_t s[];
    const int words = 128 * 260;
    for (int i = threadIdx.x; i < words; i += blockDim.x) s[i] = 0;
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    uint32_t j = tid % ngr;
    if (mode == 2) j = tid & 63;
    const uint32_t stride = (mode == 5 || mode == 6) ? 257u : 256u;
    // eight addresses per lane, fixed for the run (pseudo-random rows, k = 0..3 twice): no arithmetic inside the loop
    uint32_t acc = 0, h = sym[tid] * 2654435761u + 12345u;
    uint32_t *a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        h ^= h << 13, h ^= h >> 17, h ^= h << 5;
        uint32_t row = 33 + (h >> 8) % 41;
        if (mode == 1) row = 0;
        if (mode == 4 || mode == 6) row = 33 + (row & 3) * 11;
        a[u] = &s[row * stride + (u & 3) * 64 + j];
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) __hip_atomic_fetch_add(a[u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    uint32_t t = acc;
    for (int i = threadIdx.x; i < words; i += blockDim.x) t += s[i];
    if (t == 12345) out[0] = t;
}

static void run(int mode, int ngr, const char *name, const uint32_t *sym, uint32_t *d)
{
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 260 * 4));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 128 * 260 * 4, 0, d, sym, 10, mode, ngr);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 128 * 260 * 4, 0, d, sym, iters, mode, ngr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = 16.0 * iters * 8;
    printf("%-64s %8.3f ms   %6.2f clocks per wave instruction\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
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
    run(0, 38, "K1L layout, 150 bp (38 groups), 41 symbols", sym, d);
    run(1, 38, "the same, one row", sym, d);
    run(2, 38, "41 symbols, 64 distinct columns per wave", sym, d);
    run(0, 32, "K1L layout, 32 groups per read (equal j 32 lanes apart)", sym, d);
    run(0, 25, "K1L layout, 100 bp (25 groups)", sym, d);
    run(0, 63, "K1L layout, 250 bp (63 groups)", sym, d);
    run(4, 38, "150 bp, four symbols only", sym, d);
    run(5, 38, "150 bp, 41 symbols, row stride 257 words", sym, d);
    run(6, 38, "150 bp, four symbols, row stride 257 words", sym, d);
    return 0;
}
