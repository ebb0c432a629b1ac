// Probe: what does the split-K epilogue of the 256x256 token-reduction GEMM cost, and does it matter which XCD the splits of a tile run on?
//   grid = tiles * splits workgroups of 512 threads; each adds a 256x256 fp32 tile (64 values per lane, the 8-phase kernel's lane pattern)
//   mode 0: splits of a tile on DIFFERENT XCDs (slice-major: wi -> tile = wi % tiles after the XCD-contiguous remap)
//   mode 1: splits of a tile on the SAME XCD
//   op 0: atomicAdd   1: plain store (wrong sums, traffic only)   2: nothing (launch floor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(512) void epi(float* C, int* xcc_seen, int tiles, int tiles_n, int splits, int ldc, int mode, int op) {
    int wi = blockIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15;
    if (threadIdx.x == 0) xcc_seen[blockIdx.x] = (int)xcc;
    int tile, ks;
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = wi & 7;
    const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wi >> 3);     // XCD-contiguous index
    if (mode == 0) { tile = lin % tiles; ks = lin / tiles; }
    else { tile = lin / splits; ks = lin % splits; }
    if (tile >= tiles) return;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3, g = lane >> 4, pp = lane & 15;
    float v = 1.0f + ks;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = tm * 256 + a * 128 + wr * 64 + i * 16 + g * 4 + rr;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = tn * 256 + b * 128 + wc * 32 + j * 16 + pp;
                        float* c = C + (size_t)m * ldc + n;
                        if (op == 0) atomicAdd(c, v);
                        else if (op == 1) *c = v;
                    }
            }
}
int main() {
    const int tm = 6, tn = 5, tiles = tm * tn, ldc = tn * 256;
    float* C; int* seen;
    hipMalloc(&C, (size_t)tm * 256 * ldc * 4);
    hipMalloc(&seen, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int splits : {8, 4}) for (int mode : {0, 1}) for (int op : {0, 1, 2}) {
        const int grid = tiles * splits;
        hipMemset(C, 0, (size_t)tm * 256 * ldc * 4);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(epi, dim3(grid), dim3(512), 0, 0, C, seen, tiles, tn, splits, ldc, mode, op);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int N = 20;
        for (int it = 0; it < N; ++it) hipLaunchKernelGGL(epi, dim3(grid), dim3(512), 0, 0, C, seen, tiles, tn, splits, ldc, mode, op);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<int> h(grid); hipMemcpy(h.data(), seen, grid * 4, hipMemcpyDeviceToHost);
        int agree = 0; for (int i = 0; i < grid; ++i) agree += (h[i] == (i & 7));
        std::vector<float> c(16); hipMemcpy(c.data(), C, 64, hipMemcpyDeviceToHost);
        printf("splits %d mode %d (%s) op %d (%s): %.1f us/launch   xcc_id==blockIdx%%8 for %d/%d   C[0]=%.0f (expect %d)\n", splits, mode,
               mode ? "same XCD" : "different XCDs", op, op == 0 ? "atomicAdd" : op == 1 ? "store" : "none", ms * 1e3 / N, agree, grid, c[0],
               op == 0 ? (N + 3) * (splits * (splits + 1) / 2) : 0);
    }
    return 0;
}
