// Microbenchmark: the memory side of K1L alone.  One 1024-thread workgroup per CU streams chunks of reads; a lane's item is
// kW bytes of one read at a read-relative offset (so items start at any byte phase when len % kW != 0), kSpan items per set,
// two sets (one in flight while the other is consumed by a few VALU operations per byte).  Which item width streams fastest?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

template <int kW> struct Item;
template <> struct Item<4> { typedef uint32_t t; static __device__ uint32_t fold(t v) { return v; } };
template <> struct Item<8> { typedef uint32_t t __attribute__((ext_vector_type(2))); static __device__ uint32_t fold(t v) { return v[0] ^ v[1]; } };
template <> struct Item<16> { typedef uint32_t t __attribute__((ext_vector_type(4))); static __device__ uint32_t fold(t v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; } };

template <int kW>
__device__ typename Item<kW>::t ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    if constexpr (kW == 4) return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    else if constexpr (kW == 8) return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    else return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

template <int kW, int kSpan>
__global__ __launch_bounds__(1024) void k(const uint8_t *data, uint64_t n_reads, uint32_t len, uint32_t chunk_reads, uint32_t *out)
{
    const uint32_t ngr = (len + kW - 1) / kW, rpr = 1024 / ngr, lr = threadIdx.x / ngr, j = threadIdx.x - lr * ngr;
    const uint32_t step = __builtin_amdgcn_readfirstlane(rpr * len);
    uint32_t acc = 0;
    const uint64_t nchunk = n_reads / chunk_reads;
    for (uint64_t ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const uint64_t b0 = (uint64_t)(uintptr_t)data + ch * (uint64_t)chunk_reads * len;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b0)),
            0, (int)(chunk_reads * len), 0x00020000);
        const uint32_t rounds = chunk_reads / rpr, nsets = rounds / kSpan;
        uint32_t voff = lr < rpr ? lr * len + kW * j : 0x7ff00000u;
        typename Item<kW>::t va[kSpan], vb[kSpan];
#pragma unroll
        for (int m = 0; m < kSpan; ++m) va[m] = ld<kW>(rsrc, voff, m * step);
        voff += kSpan * step;
        for (uint32_t s = 0; s + 1 < nsets; s += 2) {
#pragma unroll
            for (int m = 0; m < kSpan; ++m) vb[m] = ld<kW>(rsrc, voff, m * step);
            voff += kSpan * step;
#pragma unroll
            for (int m = 0; m < kSpan; ++m) acc += Item<kW>::fold(va[m]) * 3u;
#pragma unroll
            for (int m = 0; m < kSpan; ++m) va[m] = ld<kW>(rsrc, voff, m * step);
            voff += kSpan * step;
#pragma unroll
            for (int m = 0; m < kSpan; ++m) acc += Item<kW>::fold(vb[m]) * 5u;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int kW, int kSpan>
static void run(const uint8_t *d, uint64_t bytes, uint32_t len, uint32_t *out)
{
    const uint64_t n_reads = bytes / len;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<kW, kSpan>), dim3(256), dim3(1024), 0, 0, d, n_reads, len, 16384u, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<kW, kSpan>), dim3(256), dim3(1024), 0, 0, d, n_reads, len, 16384u, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    printf("read length %3u  item %2d bytes x %2d per set   %7.3f ms   %5.0f GB/s\n", len, kW, kSpan, ms, (double)(n_reads / 16384 * 16384) * len / ms / 1e6);
}

int main()
{
    const uint64_t bytes = 30ull << 30;
    uint8_t *d;
    uint32_t *out;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes);
    for (uint32_t len : {128u, 150u, 100u}) {
        run<4, 16>(d, bytes, len, out);
        run<8, 8>(d, bytes, len, out);
        run<8, 16>(d, bytes, len, out);
        run<16, 4>(d, bytes, len, out);
        run<16, 8>(d, bytes, len, out);
    }
    return 0;
}
