// Microbenchmark: do VALU work and LDS atomics of the same waves overlap on an MI355X CU, or add up?
// Each iteration: kValu dependent-free integer operations per lane + 1 LDS atomic add (conflict-free).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

template <int kValu, bool kLds>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters)
{
    __shared__ uint32_t s[16 * 2048];
    for (int i = threadIdx.x; i < 16 * 2048; i += blockDim.x) s[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *p = s + wave * 2048 + lane;
    uint32_t a = lane, b = wave + 1, c = 3, d = 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int v = 0; v < kValu; ++v) {            // four independent chains: no dependency stalls
                if ((v & 3) == 0) a = a * 5 + b;         // v_mad_u32_u24-like / v_mul_lo + add: count the instructions in the ISA
                else if ((v & 3) == 1) b ^= a >> 3;
                else if ((v & 3) == 2) c += b & 0xff;
                else d = (d << 1) | (c & 1);
            }
            if (kLds) __hip_atomic_fetch_add(p + (u & 1) * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    uint32_t t = a + b + c + d;
    for (int i = threadIdx.x; i < 16 * 2048; i += blockDim.x) t += s[i];
    if (t == 12345) out[0] = t;
}

template <int kValu, bool kLds>
static void run()
{
    uint32_t *d;
    CK(hipMalloc(&d, 4));
    const int iters = 2000, blocks = 256 * 2;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<kValu, kLds>), dim3(blocks), dim3(1024), 0, 0, d, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<kValu, kLds>), dim3(blocks), dim3(1024), 0, 0, d, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double steps_per_cu = (double)blocks / 256 * 16 * iters * 8;      // (wave, step) pairs per CU
    printf("valu ops per step %2d  lds atomic %d   %8.3f ms   %6.2f CU clocks per wave step\n", kValu, (int)kLds, ms, ms * 1e-3 * 2.4e9 / steps_per_cu);
    CK(hipFree(d));
}

int main()
{
    run<0, true>();
    run<4, false>();
    run<4, true>();
    run<8, false>();
    run<8, true>();
    run<16, false>();
    run<16, true>();
    return 0;
}
