// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS, or leave the old bytes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void k(const unsigned* A, unsigned* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 256; i += 64) ((unsigned*)smem)[i] = 0xABABABABu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x == 5) voff = 0x80000000u;          // far out of range
    if (threadIdx.x == 6) voff = nbytes;               // just out of range
    if (threadIdx.x == 7) voff = 0x80000000u + 64;     // out of range, soffset added below
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_AS(smem), 16, voff, threadIdx.x == 7 ? 0 : 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    unsigned *dA, *dO, h[256];
    hipMalloc(&dA, 4096), hipMalloc(&dO, 1024);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    hipMemcpy(dA, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, dA, dO, 1024);
    hipMemcpy(h, dO, 1024, hipMemcpyDeviceToHost);
    for (int l = 3; l < 9; ++l) printf("lane %d: %08x %08x %08x %08x\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
