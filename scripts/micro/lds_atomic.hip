// Microbenchmark: LDS atomic-add throughput of one MI355X CU for the address patterns K1L meets: clocks per wave instruction.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_atomic.hip -o .scratch/lds_atomic && .scratch/lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

__device__ __forceinline__ uint32_t pattern(int mode, uint32_t lane)
{
    switch (mode) {
    case 0: return lane;                                  // 64 consecutive words
    case 1: return (lane & 31) + (lane >> 5) * 1024;      // halves collide bank for bank, different addresses
    case 2: return (lane & 15) + (lane >> 4) * 1024;      // quarters collide
    case 3: return (lane & 7) + (lane >> 3) * 1024;       // eighths collide
    case 4: return (lane >> 1) + (lane & 1) * 1024;       // neighbours share a bank
    case 5: return lane >> 1;                             // neighbours share an ADDRESS
    case 6: return lane >> 2;                             // four lanes one address
    case 7: return (lane & 31) ;                          // lane and lane+32 share an address
    case 8: return lane * 32;                             // one bank, 64 addresses
    case 9: return 5;                                     // one address
    case 10: return (lane * 2654435761u >> 20) & 4095;    // scattered
    case 11: return lane * 2;                             // stride 2
    default: return (lane & 15) * 2 + (lane >> 4) * 1024; // quarters collide on even banks
    }
}

template <bool kAtomic>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, int mode)
{
    __shared__ uint32_t s[16 * 2048];
    for (int i = threadIdx.x; i < 16 * 2048; i += blockDim.x) s[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *p = s + ((wave * 2048 + pattern(mode, lane)) & (16 * 2048 - 1));   // offsets below stay inside the wave's 2048 words... mostly
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (kAtomic) __hip_atomic_fetch_add(p + (u & 1) * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else p[(u & 1) * 64] += 1;
        }
    }
    __syncthreads();
    uint32_t t = 0;
    for (int i = threadIdx.x; i < 16 * 2048; i += blockDim.x) t += s[i];
    if (t == 12345) out[0] = t;
}

static void run(int mode, const char *name)
{
    uint32_t *d;
    CK(hipMalloc(&d, 4));
    const int iters = 4000, blocks = 256 * 2;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(1024), 0, 0, d, 10, mode);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(1024), 0, 0, d, iters, mode);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks / 256 * 16 * iters * 8;      // wave instructions per CU
    printf("%-52s %8.3f ms   %6.2f clocks per wave instruction   %5.1f lanes per clock per CU\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu,
           64 * instr_per_cu / (ms * 1e-3 * 2.4e9));
    CK(hipFree(d));
}

int main()
{
    run(0, "64 consecutive words");
    run(11, "stride 2 words");
    run(1, "halves collide bank for bank (2 addresses per bank)");
    run(2, "quarters collide (4 per bank)");
    run(12, "quarters collide, even banks only");
    run(3, "eighths collide (8 per bank)");
    run(4, "neighbouring lanes share a bank");
    run(5, "neighbouring lanes share an ADDRESS");
    run(6, "four lanes one address");
    run(7, "lane and lane+32 one address");
    run(8, "one bank, 64 addresses");
    run(9, "one address");
    run(10, "scattered over 4096 words");
    return 0;
}
