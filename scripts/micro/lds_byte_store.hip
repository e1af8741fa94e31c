// Microbenchmark: LDS store throughput by width, lanes 54 bytes apart (one text line per lane, as k_bedgraph_text writes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

template <int kMode>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[40960];
    const uint32_t tid = threadIdx.x;
    uint8_t *p = s + tid * (kMode == 3 ? 64u : 54u);
    uint32_t v = tid;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (kMode == 0) p[u] = (uint8_t)v;                                                   // byte stores, consecutive bytes of a line
            else if (kMode == 1) *reinterpret_cast<uint16_t *>(p + 2 * u) = (uint16_t)v;         // 2-byte stores (lines are 2-byte aligned)
            else if (kMode == 2) *reinterpret_cast<uint32_t *>(s + tid * 56u + 4 * u) = v;       // dword stores, lines 56 bytes apart
            else *reinterpret_cast<uint32_t *>(p + 4 * u) = v;                                    // dword stores, lines 64 bytes apart (one bank!)
        }
        v += 1;
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (s[tid] == 77 && iters < 0) out[0] = s[tid * 3];
}

template <int kMode>
static void run(const char *name)
{
    uint32_t *d;
    CK(hipMalloc(&d, 4));
    const int iters = 4000, blocks = 256 * 3;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks / 256 * 4 * iters * 16;
    printf("%-52s %8.3f ms   %6.2f clocks per wave instruction\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main()
{
    run<0>("ds_write_b8, lanes 54 bytes apart");
    run<1>("ds_write_b16, lanes 54 bytes apart");
    run<2>("ds_write_b32, lanes 56 bytes apart");
    run<3>("ds_write_b32, lanes 64 bytes apart");
    return 0;
}
