// Does the range check of a raw buffer load (stride 0) on gfx950 include the SGPR offset?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const unsigned char *p, unsigned *o, unsigned soff)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 1024, 0x00020000);
    o[0] = __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0);            // in range
    o[1] = __builtin_amdgcn_raw_buffer_load_b32(r, 1020, 0, 0);         // last dword in range
    o[2] = __builtin_amdgcn_raw_buffer_load_b32(r, 1022, 0, 0);         // straddles the end
    o[3] = __builtin_amdgcn_raw_buffer_load_b32(r, 1024, 0, 0);         // voffset out of range
    o[4] = __builtin_amdgcn_raw_buffer_load_b32(r, 0, soff, 0);         // soffset (4096) out of range
    o[5] = __builtin_amdgcn_raw_buffer_load_b32(r, 512, soff / 8, 0);   // voffset 512 + soffset 512 = 1024: out of range only as a sum
    o[6] = __builtin_amdgcn_raw_buffer_load_b32(r, 1021, 0, 0);         // unaligned, straddles
    o[7] = __builtin_amdgcn_raw_buffer_load_b32(r, 1017, 0, 0);         // unaligned, inside
}
int main()
{
    unsigned char *d;
    unsigned *o, h[8];
    if (hipMalloc(&d, 1 << 20) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    (void)hipMemset(d, 0xAB, 1 << 20);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o, 4096u);
    (void)hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    const char *n[8] = {"offset 0", "offset 1020 (last dword)", "offset 1022 (straddles the end)", "voffset 1024", "soffset 4096", "voffset 512 + soffset 512",
                        "offset 1021 (unaligned, straddles)", "offset 1017 (unaligned, inside)"};
    for (int i = 0; i < 8; ++i) printf("%-40s -> %08x\n", n[i], h[i]);
    return 0;
}
