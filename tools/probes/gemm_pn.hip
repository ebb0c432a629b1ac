// gemm_pn.hip -- probe (round 3): "row panel" bf16 GEMM for N = 384 outputs on the 8-phase discipline of gemm8p.hip.
// C[M][384] = A[M][K] . B[384][K]^T, bf16 in, fp32 accumulate, bf16 out.  One workgroup = one 160-row panel x ALL 384 columns:
// M = 35840 is 224 panels = ONE round of a 256-CU chip (the 256^2 tiling needs 1.5 column tiles and 1.09 rounds).
//   * 8 waves = 2 (wr) x 4 (wc); wave tile 80 x 96 = 5 x 6 blocks of v_mfma_f32_16x16x32_bf16 (120 accumulator registers);
//   * LDS = 2 K-tile buffers x {A: 160 rows, B lo: 192 rows, B hi: 192 rows} x 64 k (68 KiB per buffer);
//   * a K-tile is four phases of 15 MFMA: (k-step 0, B lo) (k-step 1, B lo) (k-step 0, B hi) (k-step 1, B hi); the A
//     fragments of both k-steps stay in registers for the K-tile, every LDS region is read in two ADJACENT phases;
//   * DMA: B hi of K-tile t+1 in phase 2, A and B lo of K-tile t+2 in phase 4 (2 phases after their buffer's last read),
//     9 instructions per wave and K-tile, two counted waits (vmcnt(9)) -- 4 phases of lead;
//   * B rows are permuted in the LDS image: lane group g of wave wc owns columns wc*96 + jl*32 + g*8 + hb*4 + r.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gemm_pn.hip -o tools/probes/gemm_pn
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define SB() __builtin_amdgcn_sched_barrier(0)
#define BAR()                                   \
    do {                                        \
        SB();                                   \
        asm volatile("s_barrier" ::: "memory"); \
        SB();                                   \
    } while (0)
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

constexpr int A_BYTES = 160 * 128, BH_BYTES = 192 * 128, BUF_BYTES = A_BYTES + 2 * BH_BYTES;   // 20 + 24 + 24 KiB
constexpr unsigned OOB = 0x80000000u;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}

#ifdef TIMING
__device__ unsigned long long g_stamps[256 * 2 * 16];
#define STAMP(k) do { if (lane == 0 && (w & 3) == 0) g_stamps[(blockIdx.x * 2 + wr) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define STAMP(k)
#endif

__global__ __launch_bounds__(512, 2) void gemm_pn_kernel(const u16* __restrict__ A, const u16* __restrict__ B, u16* __restrict__ C, int M,
                                                         int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int tile = blockIdx.x;
    const int nk = K >> 6;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 384 * K * 2, 0x00020000);

    // ---- DMA lane geometry (1 KiB per wave instruction = 8 LDS rows of 128 B, lane -> row +(lane>>3), chunk position lane&7
    // holding source chunk (lane&7) ^ (row&7))
    const int srow = lane >> 3;
    const unsigned schunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
    const unsigned rsb = (unsigned)K * 2u;
    unsigned voffA[3], voffB[6];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ra = (q * 8 + (q == 2 ? (w & 3) : w)) * 8 + srow;      // q = 2: waves 4-7 repeat waves 0-3 (same data, same address)
        const int m = tile * 160 + ra;
        voffA[q] = m < M ? (unsigned)m * rsb + schunk16 : OOB;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int R = (q % 3) * 64 + w * 8 + srow, hb = q / 3;           // row inside the half image
        const int wcr = R / 48, r2 = R % 48, jl = r2 >> 4, rho = r2 & 15;
        const int n = wcr * 96 + jl * 32 + (rho >> 2) * 8 + hb * 4 + (rho & 3);
        voffB[q] = (unsigned)n * rsb + schunk16;
    }
    auto issueA = [&](const int kt, const int buf) __attribute__((always_inline)) {
        const bool live = kt < nk;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            unsigned char* dst = smem + buf * BUF_BYTES + (q * 8 + (q == 2 ? (w & 3) : w)) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_AS(dst), 16, live ? voffA[q] : OOB, (unsigned)kt * 128u, 0, 0);
        }
    };
    auto issueB = [&](const int hb, const int kt, const int buf) __attribute__((always_inline)) {
        const bool live = kt < nk;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            unsigned char* dst = smem + buf * BUF_BYTES + A_BYTES + hb * BH_BYTES + (q * 8 + w) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LDS_AS(dst), 16, live ? voffB[hb * 3 + q] : OOB, (unsigned)kt * 128u, 0, 0);
        }
    };

    // ---- fragment reads: row (lane&15) of a 16-row block, chunk (s*4 + (lane>>4)) ^ (row & 7)
    const int fr = lane & 15, g = lane >> 4;
    const unsigned fch = (unsigned)((g ^ (fr & 7)) << 4);
    const unsigned aoff = (unsigned)((wr * 80 + fr) * 128) + fch;                  // + i*2048; ^64: second k-step
    const unsigned boff = (unsigned)(A_BYTES + (wc * 48 + fr) * 128) + fch;        // + hb*BH_BYTES + jl*2048

    f32x4 acc[5][6];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[5][2], fbx[3], fby[3];
    auto readA = [&](const unsigned char* buf, const int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 5; ++i) fa[i][s] = *(const bf16x8*)(buf + i * 2048 + (s ? (aoff ^ 64u) : aoff));
    };
    auto readB = [&](const unsigned char* buf, const int hb, const int s, bf16x8(&fb)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) fb[j] = *(const bf16x8*)(buf + hb * BH_BYTES + j * 2048 + (s ? (boff ^ 64u) : boff));
    };
    auto mm = [&](const int hb, const int s, const bf16x8(&fb)[3]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][hb * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i][s], acc[i][hb * 3 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    STAMP(0);
    issueA(0, 0), issueB(0, 0, 0), issueB(1, 0, 0);
    issueA(1, 1), issueB(0, 1, 1);
    WAIT_VM(9);
    BAR();
    if (wr == 1) BAR();
    STAMP(1);

    auto ktile = [&](const int kt, const int b) __attribute__((always_inline)) {
        const unsigned char* cur = smem + b * BUF_BYTES;
        // phase 1: k-step 0 x B lo
        readB(cur, 0, 0, fbx);
        readA(cur, 0);
        BAR();
        WAIT_LGKM(0);
        SB();
        mm(0, 0, fbx);
        BAR();
        // phase 2: k-step 1 x B lo.  DMA: B hi of the next K-tile (its buffer's B hi was last read two phases ago)
        readB(cur, 0, 1, fby);
        readA(cur, 1);
        issueB(1, kt + 1, b ^ 1);
        WAIT_VM(9);            // B hi of THIS K-tile has landed
        BAR();
        WAIT_LGKM(0);
        SB();
        mm(0, 1, fby);
        BAR();
        // phase 3: k-step 0 x B hi
        readB(cur, 1, 0, fbx);
        BAR();
        WAIT_LGKM(0);
        SB();
        mm(1, 0, fbx);
        BAR();
        // phase 4: k-step 1 x B hi.  DMA: A and B lo of K-tile + 2 into this buffer (last read in phase 2)
        readB(cur, 1, 1, fby);
        issueA(kt + 2, b), issueB(0, kt + 2, b);
        WAIT_VM(9);            // A and B lo of the next K-tile have landed
        BAR();
        WAIT_LGKM(0);
        SB();
        mm(1, 1, fby);
        BAR();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, 0);
        ktile(kt + 1, 1);
    }
    STAMP(2);
    if (wr == 0) BAR();
    WAIT_VM(0);

    // ---- store: lane (fr, g) owns rows wr*80 + i*16 + fr x columns wc*96 + jl*32 + g*8 .. +7
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int m = tile * 160 + wr * 80 + i * 16 + fr;
        if (m < M) {
#pragma unroll
            for (int jl = 0; jl < 3; ++jl) {
                const f32x4 lo = acc[i][jl], hi = acc[i][3 + jl];
                uint4 o;
                o.x = pack2(lo[0], lo[1]), o.y = pack2(lo[2], lo[3]), o.z = pack2(hi[0], hi[1]), o.w = pack2(hi[2], hi[3]);
                *(uint4*)(C + (size_t)m * 384 + wc * 96 + jl * 32 + g * 8) = o;
            }
        }
    }
    STAMP(3);
}

// ---- reference: one thread per output, fp32 accumulate ------------------------------------------------------------
__global__ void ref_kernel(const u16* A, const u16* B, float* C, int M, int N, int K, int m0, int rows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = m0 + blockIdx.y;
    if (n >= N || blockIdx.y >= rows) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        s += __uint_as_float((unsigned)A[(size_t)m * K + k] << 16) * __uint_as_float((unsigned)B[(size_t)n * K + k] << 16);
    C[(size_t)blockIdx.y * N + n] = s;
}

static u16 h_f2bf(float f) {
    unsigned int u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
static float h_bf2f(u16 h) {
    unsigned int u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static int run_case(int M, int K, int iters, bool check) {
    const int N = 384;
    const size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
    std::vector<u16> ha(na), hb(nb);
    unsigned long long s = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(M * 31 + N * 17 + K);
    auto rnd = [&]() {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        return (float)((s >> 40) & 0xffffff) / 8388608.0f - 1.0f;   // uniform [-1, 1)
    };
    for (auto& x : ha) x = h_f2bf(rnd());
    for (auto& x : hb) x = h_f2bf(rnd());
    u16 *dA, *dB, *dC;
    (void)hipMalloc(&dA, na * 2), (void)hipMalloc(&dB, nb * 2), (void)hipMalloc(&dC, nc * 2);
    (void)hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dC, 0xff, nc * 2);
    const int ntiles = (M + 159) / 160;
    const int lds = 2 * BUF_BYTES;
    (void)hipFuncSetAttribute((const void*)gemm_pn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    auto launch = [&]() { hipLaunchKernelGGL(gemm_pn_kernel, dim3(ntiles), dim3(512), lds, 0, dA, dB, dC, M, K); };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    int bad = 0;
    if (check) {
        const int rows = M < 64 ? M : 64;
        float* dR;
        (void)hipMalloc(&dR, (size_t)rows * N * 4);
        std::vector<float> hr((size_t)rows * N);
        std::vector<u16> hc(nc);
        (void)hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int blk = 0; blk < 6; ++blk) {
            const int m0 = (int)(((long long)blk * (M - rows)) / 5);
            hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, rows), dim3(256), 0, 0, dA, dB, dR, M, N, K, m0, rows);
            (void)hipMemcpy(hr.data(), dR, (size_t)rows * N * 4, hipMemcpyDeviceToHost);
            for (int r = 0; r < rows; ++r)
                for (int n = 0; n < N; ++n) {
                    const float ref = hr[(size_t)r * N + n], got = h_bf2f(hc[(size_t)(m0 + r) * N + n]);
                    const double err = fabs((double)ref - got), tol = 0.02 * sqrt((double)K) * 0.35 + 0.01 * fabs(ref);
                    if (err > maxerr) maxerr = err;
                    if (!(err <= tol)) {
                        if (bad < 5) printf("  MISMATCH m=%d n=%d ref=%f got=%f\n", m0 + r, n, ref, got);
                        ++bad;
                    }
                }
        }
        printf("  check %dx%dx%d: %s (max abs err %.4f, %d bad)\n", M, N, K, bad ? "FAIL" : "ok", maxerr, bad);
        (void)hipFree(dR);
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int warm = (int)(0.3e6 / (2.0 * M * N * K / 1e9 + 20.0)) + 3;     // >= 0.3 s of launches: the clocks have ramped up
    for (int i = 0; i < warm; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
#ifdef TIMING
    {
        std::vector<unsigned long long> st(256 * 2 * 16);
        (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        for (int blk : {0, 1, 100, 223}) {
            if (blk >= ntiles) continue;
            for (int g = 0; g < 2; ++g) {
                const unsigned long long* q = &st[(blk * 2 + g) * 16];
                printf("   wg %3d group %d (10 ns ticks):", blk, g);
                for (int k = 1; k < 4; ++k) printf(" %lld", (long long)(q[k] - q[0]));
                printf("\n");
            }
        }
    }
#endif
    printf("gemm_pn %6d x %5d x %5d : %8.1f us  %7.1f TFLOP/s  (%d panels)\n", M, N, K, us, tf, ntiles);
    (void)hipFree(dA), (void)hipFree(dB), (void)hipFree(dC);
    return bad != 0;
}

int main(int argc, char** argv) {
    int rc = 0;
    if (argc >= 3) return run_case(atoi(argv[1]), atoi(argv[2]), 50, true);
    rc |= run_case(160, 128, 5, true);
    rc |= run_case(1000, 384, 5, true);
    rc |= run_case(35840, 4608, 50, true);
    rc |= run_case(35840, 1152, 100, true);
    rc |= run_case(35840, 768, 100, false);
    rc |= run_case(35840, 384, 100, false);
    rc |= run_case(40960, 4608, 50, false);
    return rc;
}
