// Microbenchmark: LDS atomic (ds_add_u32) rate per CU for the address patterns of K1L.
// Indices are precomputed (16 per lane, like the 16 bytes of one vector); the timed loop is only ds_add_u32.
// hipcc --offload-arch=gfx950 -O3 scripts/lds_atomic_ubench.hip -o /tmp/lds_ubench && /tmp/lds_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kThreads = 1024, kWords = 128 * 288, kIters = 512;

__device__ uint32_t pattern(int mode, uint32_t tid, int k, uint32_t &x)
{
    const uint32_t lane = tid & 63;
    x = x * 1664525u + 1013904223u;
    const uint32_t q = 35 + ((x >> 10) % 40), pos = ((tid * 16 + k) % 150), p2 = pos / 2;
    switch (mode) {
    case 0: return tid;                              // conflict-free: lane -> own bank
    case 1: return (x >> 8) % kWords;                // uniformly random word
    case 2: return q * 257 + p2;                     // K1L layout: row stride 257
    case 3: return q * 256 + p2;                     // row stride 256
    case 4: return 40 * 257 + p2;                    // constant quality
    case 5: return q * 257 + (k % 150) / 2;          // every lane at the same cycle (read-per-lane mapping)
    case 6: return p2 * 128 + q;                     // cycle-major
    case 7: return q * 257 + ((lane + k * 64) % 150) / 2;  // byte-interleaved lanes (lane = consecutive bytes)
    case 8: return q * 256 + ((tid + k * 1024) % 150);     // u32 counters, row stride 256: bank = cycle % 32, lanes = consecutive cycles
    case 9: return q * 288 + ((tid + k * 1024) % 150);     // same, rows of 288 words (cycles < 288)
    default: return 0;
    }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k(uint32_t *out, uint64_t *cyc)
{
    __shared__ uint32_t h[kWords];
    for (int i = threadIdx.x; i < kWords; i += kThreads) h[i] = 0;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
    uint32_t idx[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) idx[kk] = pattern(MODE, threadIdx.x, kk, x);
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) atomicAdd(&h[idx[kk]], 1u);
    }
    __syncthreads();
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < kWords; i += kThreads) s += h[i];
    out[blockIdx.x * kThreads + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name)
{
    uint32_t *out;
    uint64_t *cyc;
    hipMalloc(&out, 256 * kThreads * 4);
    hipMalloc(&cyc, 256 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k<MODE><<<256, kThreads>>>(out, cyc);
    hipEventRecord(a);
    k<MODE><<<256, kThreads>>>(out, cyc);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    uint64_t h[256];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += h[i];
    c /= 256;
    const double ops = (double)kThreads * kIters * 16;
    printf("%-52s %8.3f ms  %9.0f cyc/WG  %6.2f lane-ops/cycle/CU  %8.1f Gops/s chip\n", name, ms, c, ops / c,
           ops * 256 / ms / 1e6);
    hipFree(out), hipFree(cyc);
}

int main()
{
    run<0>("conflict-free (lane -> own bank)");
    run<1>("uniformly random word");
    run<2>("K1L: random q x lane-strided cycle, row stride 257");
    run<3>("K1L: row stride 256");
    run<4>("constant q, lane-strided cycle");
    run<5>("random q, all lanes at the same cycle");
    run<6>("cycle-major [p2][128]");
    run<7>("byte-interleaved lanes");
    run<8>("u32 counters, stride 256, lanes = consecutive cycles");
    run<9>("u32 counters, stride 288, lanes = consecutive cycles");
    return 0;
}
