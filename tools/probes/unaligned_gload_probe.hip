// global_load_lds_dwordx4 (the 128-row GEMM kernel's operand path) from a source address that is only 2-byte aligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_AS(p) ((__attribute__((address_space(1))) void*)(p))
typedef unsigned short u16;
__global__ void k(const u16* A, u16* out, int mis_elems) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 256; i += 64) ((unsigned*)smem)[i] = 0xABABABABu;
    __syncthreads();
    const u16* src = A + 64 + threadIdx.x * 16 + mis_elems;
    __builtin_amdgcn_global_load_lds(GLB_AS(src), LDS_AS(smem), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = ((u16*)smem)[i];
}
int main() {
    u16 *dA, *dO, h[4096], o[512];
    (void)hipMalloc(&dA, 8192), (void)hipMalloc(&dO, 1024);
    for (int i = 0; i < 4096; ++i) h[i] = (u16)(0x1000 + i);
    (void)hipMemcpy(dA, h, 8192, hipMemcpyHostToDevice);
    for (int mis = -7; mis <= 7; ++mis) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, dA, dO, mis);
        (void)hipMemcpy(o, dO, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) bad += o[l * 8 + e] != (u16)(0x1000 + 64 + l * 16 + mis + e);
        printf("mis %2d elements: wrong elements %d\n", mis, bad);
    }
    return 0;
}
