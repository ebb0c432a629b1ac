// buffer_load_dwordx4 ... lds from a source address that is only 2-byte aligned (the compact dBD matrix of the attention backward
// is the flat dS sequence shifted by (row + 1) elements: read as a VIEW of dS its rows start at odd element offsets): does the
// 16-byte LDS-DMA deliver the right bytes, and what does the range check do with a chunk that starts in front of the buffer
// (voffset wrapped negative) or straddles its end?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
typedef unsigned short u16;
__global__ void k(const u16* A, u16* out, int nbytes, int mis, int base_elems) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 256; i += 64) ((unsigned*)smem)[i] = 0xABABABABu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(A + base_elems), 0, nbytes, 0x00020000);
    const unsigned voff = (unsigned)((int)threadIdx.x * 32 + mis);      // (negative for lane 0 when mis < 0: wraps)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_AS(smem), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = ((u16*)smem)[i];
}
int main() {
    u16 *dA, *dO, h[4096], o[512];
    hipMalloc(&dA, 8192), hipMalloc(&dO, 1024);
    for (int i = 0; i < 4096; ++i) h[i] = (u16)(0x1000 + i);
    hipMemcpy(dA, h, 8192, hipMemcpyHostToDevice);
    const int base = 64;      // the descriptor's base sits 64 elements into the allocation: "in front of the buffer" is mapped memory
    for (int mis = -14; mis <= 14; mis += 2) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, dA, dO, 64 * 32, mis, base);
        hipMemcpy(o, dO, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 1; l < 63; ++l)
            for (int e = 0; e < 8; ++e) bad += o[l * 8 + e] != (u16)(0x1000 + base + l * 16 + mis / 2 + e);
        printf("mis %3d bytes: lanes 1..62 wrong elements %d | lane 0:", mis, bad);
        for (int e = 0; e < 8; ++e) printf(" %04x", o[e]);
        printf(" (in-range would be %04x..) | lane 63:", (u16)(0x1000 + base + mis / 2));
        for (int e = 0; e < 8; ++e) printf(" %04x", o[63 * 8 + e]);
        printf(" (records end after element %04x)\n", (u16)(0x1000 + base + 64 * 16 - 1));
    }
    return 0;
}
