// Does the vector L1 (TCP) keep lines fetched by global_load_lds (LDS DMA)?  And which CU does workgroup i of a
// 1024 x 256-thread launch land on?  hipcc --offload-arch=gfx950 -O3 -o tcp_dma_probe tcp_dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))

// mode 0: every workgroup streams its own 32 KiB per iteration (no reuse possible)
// mode 1: 16 KiB shared by ALL workgroups (same addresses every iteration) + 16 KiB private
// mode 2: 16 KiB shared by the 4 workgroups with the same (blockIdx.x & ~3) + 16 KiB private, addresses advance per iteration
// mode 4: L2-resident working set (2 MiB): the L2 -> LDS DMA ceiling
// mode 3: 16 KiB shared by the 4 workgroups of one CU (breadth-first dispatch: same XCD, same local index % 32)
__global__ __launch_bounds__(256, 4) void probe(const unsigned char* src, size_t span, int iters, int mode, unsigned int* hw) {
    extern __shared__ unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        unsigned int xcc, id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hw[blockIdx.x * 2] = xcc, hw[blockIdx.x * 2 + 1] = id;
    }
    for (int it = 0; it < iters; ++it) {
        size_t a_off, b_off;
        if (mode == 0) {
            a_off = ((size_t)blockIdx.x * iters + it) * 32768 % span;
            b_off = a_off + 16384;
        } else if (mode == 1) {
            a_off = 0;
            b_off = (16384 + ((size_t)blockIdx.x * iters + it) * 16384) % span;
        } else if (mode == 2) {
            a_off = ((size_t)(blockIdx.x >> 2) * iters + it) * 16384 % (span / 2);
            b_off = span / 2 + ((size_t)blockIdx.x * iters + it) * 16384 % (span / 2);
        } else if (mode == 4) {   // mode 4: everything hits the L2: 32 KiB per iteration out of a 2 MiB window
            a_off = (((size_t)blockIdx.x * 7 + it) * 32768) % ((size_t)2 << 20);
            b_off = a_off + 16384;
        } else {   // mode 3: shared by the workgroups that sit on the same CU: same XCD (blockIdx % 8), same (local index % 32)
            const int xcd = blockIdx.x & 7, cu = (blockIdx.x >> 3) & 31;
            a_off = ((size_t)(xcd * 32 + cu) * iters + it) * 16384 % (span / 2);
            b_off = span / 2 + ((size_t)blockIdx.x * iters + it) * 16384 % (span / 2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds(GLB_AS(src + a_off + (w * 4 + q) * 1024 + lane * 16), LDS_AS(smem + (w * 4 + q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds(GLB_AS(src + b_off + (w * 4 + q) * 1024 + lane * 16), LDS_AS(smem + 16384 + (w * 4 + q) * 1024), 16, 0, 0);
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = 200, nwg = 1024;
    const size_t span = (size_t)1 << 30;
    unsigned char* src;
    unsigned int* hw;
    hipMalloc(&src, span + 65536);
    hipMemset(src, 1, span + 65536);
    hipMalloc(&hw, nwg * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 32768, 0, src, span, iters, mode, hw);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.1f us, %.2f TB/s into LDS\n", mode, ms * 1e3, (double)nwg * iters * 32768 / (ms * 1e-3) / 1e12);
    }
    std::vector<unsigned int> h(nwg * 2);
    hipMemcpy(h.data(), hw, nwg * 8, hipMemcpyDeviceToHost);
    if (mode == 0) {
        // HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
        printf("first 40 workgroups: (wg: xcc se cu)\n");
        for (int i = 0; i < 40; ++i) printf(" %d:%u/%u/%u", i, h[i * 2] & 15, (h[i * 2 + 1] >> 13) & 7, (h[i * 2 + 1] >> 8) & 15);
        printf("\nworkgroups 8k (XCD 0), k = 0..47: (se cu)\n");
        for (int k = 0; k < 48; ++k) printf(" %u/%u", (h[k * 16 + 1] >> 13) & 7, (h[k * 16 + 1] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
