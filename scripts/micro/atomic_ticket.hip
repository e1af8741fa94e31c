// atomic_ticket.hip -- what does ONE counter that every workgroup of a launch increments cost?
// (round 6: k_text_lines takes 44 ns per 64 KiB tile, k_depth_sweep 48 ns per tile -- both start with
//  `tile = atomicAdd(&ticket, 1)`.)
//   empty    workgroups of 256 threads that do nothing
//   same     thread 0: t = atomicAdd(ctr, 1), broadcast through LDS, out[t] = 1
//   noret    atomicAdd(ctr, 1) whose value nobody uses
//   spread   atomicAdd on one of 64 counters 256 bytes apart
// Measured (profiles/r06/atomic_ticket.txt): 11.4 ns per increment of ONE address whatever the launch size (88 M tickets/s:
// 64 KiB tiles could stream at 5.8 TB/s at most), 0.3 ns when the counters are spread -- the ticket is a floor for kernels of
// many small tiles, not a cost per tile that adds to the rest (k_depth_sweep: 15,195 tiles = 0.17 ms of its 0.74).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/atomic_ticket.hip -o /tmp/atomic_ticket && /tmp/atomic_ticket
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_empty(uint32_t *out) { if (blockIdx.x == 0xffffffffu) out[threadIdx.x] = 1; }
__global__ __launch_bounds__(256) void k_same(uint32_t *ctr, uint32_t *out)
{
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = atomicAdd(ctr, 1u);
    __syncthreads();
    if (threadIdx.x == 1) out[s] = 1;
}
__global__ __launch_bounds__(256) void k_noret(uint32_t *ctr) { if (threadIdx.x == 0) atomicAdd(ctr, 1u); }
__global__ __launch_bounds__(256) void k_spread(uint32_t *ctr, uint32_t *out)
{
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = atomicAdd(ctr + (blockIdx.x & 63u) * 64u, 1u);
    __syncthreads();
    if (threadIdx.x == 1) out[blockIdx.x] = s;
}
int main()
{
    uint32_t *ctr, *out;
    const uint32_t nmax = 1u << 20;
    CK(hipMalloc(&ctr, 64 * 256)); CK(hipMalloc(&out, nmax * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *what, uint32_t n, auto launch) -> int {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(ctr, 0, 64 * 256));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-9s %8u workgroups: %8.3f ms = %6.1f ns per workgroup\n", what, n, best, best * 1e6 / n);
        return 0;
    };
    for (uint32_t n : {16384u, 65536u, 1u << 20}) {
        timed("empty", n, [&] { hipLaunchKernelGGL(k_empty, dim3(n), dim3(256), 0, 0, out); });
        timed("same", n, [&] { hipLaunchKernelGGL(k_same, dim3(n), dim3(256), 0, 0, ctr, out); });
        timed("noret", n, [&] { hipLaunchKernelGGL(k_noret, dim3(n), dim3(256), 0, 0, ctr); });
        timed("spread", n, [&] { hipLaunchKernelGGL(k_spread, dim3(n), dim3(256), 0, 0, ctr, out); });
    }
    return 0;
}
