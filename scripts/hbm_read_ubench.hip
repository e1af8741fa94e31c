// hbm_read_ubench.hip -- what a read-only streaming kernel can get out of the MI355X's HBM: the ceiling K1 (k_tally_scan,
// 6.4 TB/s = 0.80 of the 8 TB/s peak) is to be held against.  Nothing but loads: every lane keeps U 16-byte non-temporal
// loads in flight over a grid-stride sweep of the buffer and ORs them together (one store per lane at the end, so the
// loads cannot be dropped).  Swept over workgroups per CU and loads in flight; plain (cached) loads for comparison.
//   hipcc --offload-arch=gfx950 -O3 scripts/hbm_read_ubench.hip -o hbm_read_ubench && ./hbm_read_ubench [GB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ p, size_t n_vec, u32x4 *__restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n_vec; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(p + i + k * stride) : p[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) acc |= v[k];
    }
    for (; i < n_vec; i += stride) acc |= NT ? __builtin_nontemporal_load(p + i) : p[i];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// the same loads, but a workgroup's U rows of 4 KiB are ADJACENT (one 4*U KiB tile per step, tiles grid-strided): K1's layout
template <int U>
__global__ __launch_bounds__(256) void k_read_tile(const u32x4 *__restrict__ p, size_t n_vec, u32x4 *__restrict__ sink)
{
    const size_t tile = (size_t)U * 256, n_tiles = n_vec / tile;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const u32x4 *q = p + t * tile + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(q + k * 256);
#pragma unroll
        for (int k = 0; k < U; ++k) acc |= v[k];
    }
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int U>
static double run_tile(const u32x4 *d, size_t n_vec, u32x4 *sink, int grid, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL((k_read_tile<U>), dim3(grid), dim3(256), 0, 0, d, n_vec, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_read_tile<U>), dim3(grid), dim3(256), 0, 0, d, n_vec, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

template <int U, bool NT>
static double run(const u32x4 *d, size_t n_vec, u32x4 *sink, int grid, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, d, n_vec, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, d, n_vec, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 158.0;
    const size_t bytes = (size_t)(gb * 1e9) & ~(size_t)4095, n_vec = bytes / 16;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    u32x4 *d, *sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, (size_t)n_cu * 32 * 256 * 16) != hipSuccess) return 1;
    hipMemset(d, 0x5a, bytes);
    hipDeviceSynchronize();
    printf("%s, %d CUs, %.1f GB read per launch (best of 5)\n", prop.name, n_cu, bytes / 1e9);
    for (int wg = 1; wg <= 16; wg = wg < 4 ? wg + 1 : wg * 2) {
        const int grid = n_cu * wg;
        const double t4 = run<4, true>(d, n_vec, sink, grid, 5), t8 = run<8, true>(d, n_vec, sink, grid, 5), t16 = run<16, true>(d, n_vec, sink, grid, 5),
                     p8 = run<8, false>(d, n_vec, sink, grid, 5);
        printf("wg/cu=%2d  nt x4 %7.3f ms %7.1f GB/s | nt x8 %7.3f ms %7.1f GB/s | nt x16 %7.3f ms %7.1f GB/s | plain x8 %7.3f ms %7.1f GB/s\n", wg, t4,
               bytes / t4 / 1e6, t8, bytes / t8 / 1e6, t16, bytes / t16 / 1e6, p8, bytes / p8 / 1e6);
        const double c4 = run_tile<4>(d, n_vec, sink, grid, 5), c8 = run_tile<8>(d, n_vec, sink, grid, 5), c16 = run_tile<16>(d, n_vec, sink, grid, 5);
        printf("          adjacent rows (tile per workgroup): x4 %7.3f ms %7.1f GB/s | x8 %7.3f ms %7.1f GB/s | x16 %7.3f ms %7.1f GB/s\n", c4,
               bytes / c4 / 1e6, c8, bytes / c8 / 1e6, c16, bytes / c16 / 1e6);
    }
    return 0;
}
