// Raw per-CU rates on gfx950: v_mfma_f32_32x32x16_bf16 issue rate, ds_read_b128 bandwidth, and how they overlap
// (a) inside one wave, (b) between the two waves of a SIMD.  512 threads per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// mode bit 0: MFMA work, bit 1: LDS reads; split = 1: waves 0-3 do MFMA only, waves 4-7 LDS only
template <int MODE, int SPLIT>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 32768; i += 512) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 fa[6];
    for (int i = 0; i < 6; ++i) fa[i] = *(const bf16x8*)(smem + ((lane * 16 + i * 1024 + w * 8192) & 131071));
    const bool do_m = (MODE & 1) && (!SPLIT || w < 4), do_l = (MODE & 2) && (!SPLIT || w >= 4);
    const unsigned char* base = smem + w * 12288 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (do_l) {
#pragma unroll
            for (int i = 0; i < 6; ++i) fa[i] = *(const bf16x8*)(base + i * 1024 + (it & 3) * 64);
        }
        if (do_m) {
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a % 4], fa[4 + (a & 1)], acc[a], 0, 0, 0);
        }
        if (do_l && !do_m) {   // consume the loads
            asm volatile("" :: "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fa[4]), "v"(fa[5]));
        }
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 6; ++i) s += (float)fa[i][0];
    if (s == 123.456f) out[0] = s;
}
template <int MODE, int SPLIT>
static void run(const char* name, float* d) {
    const int iters = 20000;
    hipFuncSetAttribute((const void*)probe<MODE, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, SPLIT>), dim3(256), dim3(512), 131072, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, SPLIT>), dim3(256), dim3(512), 131072, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double nm = (MODE & 1) ? (SPLIT ? 4.0 : 8.0) * 8 * iters : 0, nl = (MODE & 2) ? (SPLIT ? 4.0 : 8.0) * 6 * iters : 0;
    printf("%-34s %8.3f ms | per CU: %7.1f ns per MFMA per SIMD (%6.1f TFLOP/s chip) | LDS read %6.1f B/ns per CU\n", name, ms,
           nm ? ms * 1e6 / (nm / 4) : 0.0, nm * 32768.0 * 256 / (ms * 1e-3) / 1e12, nl * 1024.0 / (ms * 1e6));
}
int main() {
    float* d;
    (void)hipMalloc(&d, 1024);
    run<1, 0>("mfma only, 8 waves", d);
    run<2, 0>("ds_read_b128 only, 8 waves", d);
    run<3, 0>("both in every wave", d);
    run<3, 1>("waves 0-3 mfma | waves 4-7 lds", d);
    run<1, 1>("waves 0-3 mfma only", d);
    run<2, 1>("waves 4-7 lds only", d);
    return 0;
}
