// Which SIMD does wave i of a 512-thread workgroup land on?  (HW_REG_HW_ID bits [5:4] = SIMD id on gfx9.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    extern __shared__ unsigned char smem[];
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hwid;
    if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 16 * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int lds : {0, 131072}) {
        hipLaunchKernelGGL(probe, dim3(16), dim3(512), lds, 0, d);
        unsigned h[128];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("lds %d\n", lds);
        for (int b = 0; b < 4; ++b) {
            printf(" wg %d:", b);
            for (int w = 0; w < 8; ++w) printf("  w%d simd %u cu %u waveslot %u", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
            printf("\n");
        }
    }
    return 0;
}
