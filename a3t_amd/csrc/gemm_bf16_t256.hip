// gemm_bf16_t256.hip -- 256x256x64 bf16 MFMA GEMM for gfx950: 8 waves in two ping-pong groups,
// operands streamed HBM/L2 -> LDS by DMA (global_load_lds_dwordx4) three half-tiles ahead.
//
// Why a second kernel: the 128x128 kernel (gemm_bf16.hip) needs 64 B of operand DMA per MFMA cycle
// and per CU, more than the L2->LDS path delivers (measured ~11-13 TB/s chip-wide); it tops out at
// ~600-900 TFLOP/s.  A 256x256 tile halves the bytes per flop.  With one 128-KiB workgroup per CU
// nothing else hides latency, so the schedule does it explicitly:
//   * LDS = 2 K-tile buffers x {AH0, AH1, BH0, BH1}, each half-tile image 128 rows x 64 k (16 KiB),
//     the same lane-linear, source-swizzled images as the 128x128 kernel (ds_read_b128 for
//     k-contiguous operands, ds_read_b64_tr_b16 for reduction-strided ones);
//   * wave (wr, wc) of the 2x4 grid owns rows wr*64..+63 of BOTH A halves and columns wc*32..+31 of
//     BOTH B halves (a 128x64 output), so a K-tile is consumed in four quadrant phases
//     (a0 x b0, a0 x b1, a1 x b1, a1 x b0 : 8 v_mfma_f32_32x32x16_bf16 each) that need AH0+BH0, then
//     BH1, then AH1 -- which is also the order the half-tiles are (re)staged, one per phase;
//   * every phase is  [ds_reads + 2 DMA issues + counted vmcnt]  s_barrier  [8 MFMAs]  s_barrier ;
//     waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment
//     while its partner is in its load segment;
//   * a half-tile is issued 4-5 phases before it is read and waited for (s_waitcnt vmcnt(6): the three
//     youngest half-tiles stay in flight) one full phase -- two barriers -- before the first read, which
//     is what orders another wave's DMA before a ds_read; it is restaged >= 3 phases after its last read.
// Issues past the end of the K range copy the zero page (branch-free steady state, vmcnt stays counted).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/a3t_hip.h"
#include "gemm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

static __device__ __attribute__((aligned(16))) unsigned int t256_zero_page[16];
#ifdef T256_TIMING
__device__ unsigned long long t256_dbg[8192 * 8];
#define TSTAMP(i) do { if (tid == 0) t256_dbg[blockIdx.x % 8192 * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int a3t_debug_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(t256_dbg), bytes); }
#else
#define TSTAMP(i)
#endif

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))
// workgroup barrier that neither the IR optimizer nor the machine scheduler may move anything across
#define BAR()                                        \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        asm volatile("s_barrier" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

enum { L_NT = 0, L_NN = 1, L_TN = 2 };
enum { AH0 = 0, AH1 = 1, BH0 = 2, BH1 = 3 };

// CONV = im2col / token-shift addressing compiled in (taps > 1 or kshift_mode); plain GEMMs use the lean variant
template <int LAYOUT, bool CONV>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256_kernel(GP p) {
    constexpr int BK = 64, HALF_BYTES = 128 * BK * 2, TILE_BYTES = 4 * HALF_BYTES;
    constexpr bool A_KC = (LAYOUT != L_TN), B_KC = (LAYOUT == L_NT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][AH0 | AH1 | BH0 | BH1]

    const int tid = threadIdx.x, lane = tid & 63;
    TSTAMP(0);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;          // waves w and w+4 share a SIMD: they form the two ping-pong groups
    // 1-D grid over (batch / K-split, tile) work items.  Workgroup b runs on XCD b % 8, so every XCD is given a
    // CONTIGUOUS run of work items (bijective remap), ordered slice-major / tile-minor: the tiles of one batch element
    // or of one K-split -- which read the same operand slabs -- stay inside one XCD's private L2 instead of being
    // fetched by all eight (the 2-D grid did that: 5x the algorithmic HBM traffic on the split-K weight gradients).
    int wi = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wi & 7;
        wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wi >> 3);
    }
    const int bid = wi % p.ntiles, zy = wi / p.ntiles;
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int ks = zy % p.splitk, bz = zy / p.splitk;
    const int z0 = bz / p.batch_inner, z1 = bz % p.batch_inner;
    const u16* A = (const u16*)p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const u16* B = (const u16*)p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int64_t zoff = z0 * p.c_bs0 + z1 * p.c_bs1;
    const u16* ZP = (const u16*)t256_zero_page;

    const int ktiles = p.K / BK;                 // (host contract: K % 64 == 0)
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;
    const int nk = kt1 - kt0;

    // ---- per-lane DMA sources: half-tile h, instruction q of this wave (g = 2w+q of the 16 per half-tile) ------
    // k-contiguous image : row r = 8g + lane/8 of the half, 16-B chunk position lane%8 holds source chunk
    //                      (lane%8) ^ ((r>>1)&7)
    // row-contiguous image: k-row kr = 4g + lane/16, chunk position lane%16 holds source chunk (lane%16) ^ ((kr&3)<<2)
    const u16* src[4][2];
    bool ok[4][2];
    int tpos[2][2];      // CONV, A k-contiguous: position of the row inside its utterance
    int kpos[2][2];      // CONV, TN: position of the k-row (token) inside its utterance, per B half
    int kshift_h[2] = {0, 0};
    const bool WG = CONV && (LAYOUT == L_TN) && (p.taps > 1);   // fused conv weight gradient: output columns are (tap, c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (CONV && LAYOUT == L_TN) {
            if (WG) {
                const int cin = p.N / p.taps;
                kshift_h[h] = ((tn * 256 + h * 128) / cin - p.pad) * p.dil;   // a 128-column half lies inside one tap
            } else {
                kshift_h[h] = p.kshift;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = w * 2 + q;
            tpos[h][q] = 0, kpos[h][q] = 0;
            if (A_KC) {
                const int r = g * 8 + (lane >> 3), m = tm * 256 + h * 128 + r;
                ok[AH0 + h][q] = m < p.M;
                src[AH0 + h][q] = A + (int64_t)m * p.a_rs + (((lane & 7) ^ ((r >> 1) & 7)) * 8) + (int64_t)kt0 * BK;
                if (CONV) {
                    tpos[h][q] = m % p.Tseq;
                    src[AH0 + h][q] = A + (int64_t)m * p.a_rs + (((lane & 7) ^ ((r >> 1) & 7)) * 8);
                }
            } else {
                const int kr = g * 4 + (lane >> 4), col = tm * 256 + h * 128 + (((lane & 15) ^ ((kr & 3) << 2)) * 8);
                ok[AH0 + h][q] = col < p.M;
                src[AH0 + h][q] = A + col + (int64_t)(kt0 * BK + kr) * p.a_cs;
            }
            if (B_KC) {
                const int r = g * 8 + (lane >> 3), n = tn * 256 + h * 128 + r;
                ok[BH0 + h][q] = n < p.N;
                src[BH0 + h][q] = B + (int64_t)n * p.b_rs + (((lane & 7) ^ ((r >> 1) & 7)) * 8) + (int64_t)kt0 * BK;
            } else {
                const int kr = g * 4 + (lane >> 4), col = tn * 256 + h * 128 + (((lane & 15) ^ ((kr & 3) << 2)) * 8);
                ok[BH0 + h][q] = col < p.N;
                if (CONV && LAYOUT == L_TN) {
                    int cbase = col;
                    if (WG) {
                        const int cin = p.N / p.taps;
                        cbase = col - ((tn * 256 + h * 128) / cin) * cin;
                    }
                    src[BH0 + h][q] = B + cbase + (int64_t)(kt0 * BK + kr + kshift_h[h]) * p.b_cs;
                    kpos[h][q] = (kt0 * BK + kr) % p.Tseq;
                } else if (CONV) {   // NN conv (data gradient): B = W viewed [(tap, c)][n]; (tap, c) is tracked uniformly
                    src[BH0 + h][q] = B + col + (int64_t)kr * p.b_cs;
                } else {
                    src[BH0 + h][q] = B + col + (int64_t)(kt0 * BK + kr) * p.b_cs;
                }
            }
        }
    }
    // uniform (tap, channel) of the next K-tile each half stream will stage (Kc % 64 == 0: one tap per K-tile)
    int u_tap[4], u_cc[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        u_tap[h] = 0, u_cc[h] = 0;
        if (CONV && p.taps > 1 && LAYOUT != L_TN) {
            u_tap[h] = (kt0 * BK) / p.Kc;
            u_cc[h] = kt0 * BK - u_tap[h] * p.Kc;
        }
    }
    const int64_t a_step = A_KC ? BK : (int64_t)BK * p.a_cs;
    const int64_t b_step = B_KC ? BK : (int64_t)BK * p.b_cs;

    // stage half-tile H of K-tile (kt0 + t) into buffer `buf`; called once per t for every H, in t order
    auto issue = [&](const int H, int t, int buf) __attribute__((always_inline)) {
        unsigned char* dst = smem + buf * TILE_BYTES + H * HALF_BYTES + w * 2048;
        const bool live = t < nk;                       // past the K range: copy zeros, keep the DMA count steady
        const bool isA = H < 2;
        const int h = H & 1;
        if (CONV && isA && A_KC && p.taps > 1) {
            const int off = (u_tap[H] - p.pad) * p.dil;
            const int64_t roff = (int64_t)off * p.a_rs + u_cc[H];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool v = live && ok[H][q] && ((unsigned)(tpos[h][q] + off) < (unsigned)p.Tseq);
                const u16* s = v ? src[H][q] + roff : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(dst + q * 1024), 16, 0, 0);
            }
            u_cc[H] += BK;
            if (u_cc[H] >= p.Kc) u_cc[H] -= p.Kc, ++u_tap[H];
        } else if (CONV && !isA && LAYOUT == L_NN && p.taps > 1) {
            const int64_t toff = (int64_t)u_tap[H] * p.b_ts + (int64_t)u_cc[H] * p.b_cs;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u16* s = (live && ok[H][q]) ? src[H][q] + toff : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(dst + q * 1024), 16, 0, 0);
            }
            u_cc[H] += BK;
            if (u_cc[H] >= p.Kc) u_cc[H] -= p.Kc, ++u_tap[H];
        } else if (CONV && !isA && LAYOUT == L_TN) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool v = live && ok[H][q] && ((unsigned)(kpos[h][q] + kshift_h[h]) < (unsigned)p.Tseq);
                const u16* s = v ? src[H][q] : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(dst + q * 1024), 16, 0, 0);
                src[H][q] += b_step;
                kpos[h][q] += BK;
                if (p.Tseq >= BK)
                    kpos[h][q] = (kpos[h][q] >= p.Tseq) ? kpos[h][q] - p.Tseq : kpos[h][q];
                else
                    kpos[h][q] %= p.Tseq;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u16* s = (live && ok[H][q]) ? src[H][q] : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(dst + q * 1024), 16, 0, 0);
                src[H][q] += isA ? a_step : b_step;
            }
        }
    };

    f32x16 acc[4][2];   // [row block: 0,1 -> AH0 rows wr*64 + {0,32}; 2,3 -> AH1][column block: BH0 / BH1 cols wc*32]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lk = lane >> 5;
    auto frag_kc = [&](const unsigned char* img, int row, int kk) -> bf16x8 {
        const int kc = kk * 2 + lk;
        return *(const bf16x8*)(img + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
    };
    auto frag_rc = [&](const unsigned char* img, int row0, int kk) -> bf16x8 {
        // 16-lane group g: rows row0 + (g&1)*16 .. +15, k = kk*16 + (g>>1)*8 .. +7 (two 4-k transposed reads)
        const int g = lane >> 4, pp = lane & 15;
        const int col = row0 + (g & 1) * 16 + (pp & 3) * 4;
        const int kb = kk * 16 + (g >> 1) * 8 + (pp >> 2);
        const int k1 = kb + 4;
        const unsigned char* a0 = img + kb * 256 + ((((col >> 3) ^ ((kb & 3) << 2))) << 4) + (col & 7) * 2;
        const unsigned char* a1 = img + k1 * 256 + ((((col >> 3) ^ ((k1 & 3) << 2))) << 4) + (col & 7) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a1));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    bf16x8 fa[2][4], fb0[4], fb1[4];
    auto read_a = [&](const unsigned char* img) __attribute__((always_inline)) {   // this wave's 64 rows x 64 k of one A half
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                fa[i][kk] = A_KC ? frag_kc(img, wr * 64 + i * 32 + lr, kk) : frag_rc(img, wr * 64 + i * 32, kk);
    };
    auto read_b = [&](const unsigned char* img, bf16x8* fb) __attribute__((always_inline)) {   // 32 columns x 64 k of one B half
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb[kk] = B_KC ? frag_kc(img, wc * 32 + lr, kk) : frag_rc(img, wc * 32, kk);
    };
    auto quad = [&](const int i0, const int j, const bf16x8* fb) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[i0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][kk], fb[kk], acc[i0][j], 0, 0, 0);
            acc[i0 + 1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][kk], fb[kk], acc[i0 + 1][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K-tile 0 -> buffer 0, AH0 of K-tile 1 -> buffer 1 ---------------------------------------
    TSTAMP(1);
    issue(AH0, 0, 0);
    issue(BH0, 0, 0);
    issue(BH1, 0, 0);
    issue(AH1, 0, 0);
    issue(AH0, 1, 1);
    WAIT_VM(6);                 // AH0(0), BH0(0) have landed (this wave's share)
    BAR();                      // ... everyone's share
    if (wr == 1) BAR();         // stagger the second group by one barrier
    TSTAMP(2);

    for (int t = 0; t < nk; ++t) {
        const int b = t & 1;
        unsigned char* cur = smem + b * TILE_BYTES;
        // phase 0: a0 x b0
        read_b(cur + BH0 * HALF_BYTES, fb0);
        read_a(cur + AH0 * HALF_BYTES);
        issue(BH0, t + 1, b ^ 1);
        WAIT_VM(6);             // BH1(t) landed (read in phase 1)
        BAR();
        quad(0, 0, fb0);
        BAR();
        // phase 1: a0 x b1
        read_b(cur + BH1 * HALF_BYTES, fb1);
        issue(BH1, t + 1, b ^ 1);
        WAIT_VM(6);             // AH1(t) landed (read in phase 2)
        BAR();
        quad(0, 1, fb1);
        BAR();
        // phase 2: a1 x b1
        read_a(cur + AH1 * HALF_BYTES);
        issue(AH1, t + 1, b ^ 1);
        BAR();
        quad(2, 1, fb1);
        BAR();
        // phase 3: a1 x b0; AH0 of this buffer was last read three phases ago -> restage it for K-tile t+2
        issue(AH0, t + 2, b);
        WAIT_VM(6);             // AH0(t+1), BH0(t+1) landed (read in the next phase 0)
        BAR();
        quad(2, 0, fb0);
        BAR();
    }
    TSTAMP(3);
    if (wr == 0) BAR();
    WAIT_VM(0);                 // the trailing zero-page copies must not outlive the workgroup's LDS allocation

    if (!p.epi_vec || p.accumulate == A3T_ACC_ATOMIC) {   // coalesced 128-B atomic rows straight from the accumulators
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = tm * 256 + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const int col = tn * 256 + j * 128 + wc * 32 + lr;
                    epilogue_store(p, zoff, row, col, acc[i][j][r], ks);
                }
        return;
    }
    // Vector epilogue: each wave stages 64 rows x (32 + 32) columns of fp32 through its own 16 KiB of the idle
    // LDS per pass (pass = A half), then every lane finishes 4 consecutive columns of 16 rows.
    __syncthreads();
    float* ct = (float*)(smem + w * 16384);
    const int c4 = (lane & 15) * 4, r4 = lane >> 4;
    const int col = tn * 256 + (c4 >> 5) * 128 + wc * 32 + (c4 & 31);
    const bool col_ok = col < p.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && ks == 0 && col_ok) bias4 = *(const float4*)(p.bias + col);
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr] = acc[ps * 2 + i][j][r];
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int srow = it * 4 + r4;
            const int row = tm * 256 + ps * 128 + wr * 64 + srow;
            if (row >= p.M || !col_ok) continue;
            float4 v = *(const float4*)(ct + srow * 64 + c4);
            const int64_t idx = zoff + (int64_t)row * p.c_rs + col;
            epilogue_vec4(p, v, idx, bias4, ks, cs);
        }
    }
    if (p.colsum) colsum_flush(p, cs, lane, col_ok, z1, col, tm + z0);
    TSTAMP(4);
#ifdef T256_TIMING
    if (tid == 0) {
        unsigned int xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        t256_dbg[blockIdx.x % 8192 * 8 + 5] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

template <int LY, bool CV>
static void launch_t256(const GP& pv, dim3 grid, hipStream_t stream) {
    constexpr int lds = 2 * 4 * 128 * 64 * 2;   // 128 KiB
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_t256_kernel<LY, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_bf16_t256_kernel<LY, CV>), grid, dim3(512), lds, stream, pv);
}

// Called by a3t_gemm_bf16_glds after the alignment contract has been checked (pv.epi_vec resolved).
// Returns -1 when this kernel is not applicable / not profitable for the shape.
int a3t_gemm_bf16_t256(const GP& p, int batch, int ly, hipStream_t stream) {
    static int mode = -1;   // A3T_GEMM_T256 = 0: never, 1: whenever legal, unset: heuristic
    if (mode < 0) {
        const char* e = getenv("A3T_GEMM_T256");
        mode = e ? (e[0] == '0' ? 0 : 1) : 2;
    }
    if (mode == 0) return -1;
    if (p.K % 64 != 0) return -1;
    const bool conv = (p.taps > 1) || p.kshift_mode;
    if (p.taps > 1) {
        if (p.Kc % 64 != 0) return -1;
        if (ly == 0 && p.b_ts != p.Kc) return -1;                   // NT conv: weights must be [n][tap][c] contiguous
        if (ly == 2 && ((p.N / p.taps) % 128 != 0)) return -1;      // fused weight gradient: a B half inside one tap
    }
    if (p.kshift_mode && ly != 2) return -1;
    const long tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
    const long blocks = tm * tn * batch * p.splitk;
    if (mode == 2) {
        // Measured on MI355X (tools/gemm_bench2.py, tools/tn_bench3.py, tools/t256_timing.py): the main loop runs
        // at ~65 % MFMA issue (vs ~35 % for the 128x128 kernel) but with one workgroup per CU nothing overlaps the
        // ~25k-cycle prologue and ~35k-cycle epilogue, so it only wins when a workgroup owns >= ~64 K-tiles and the
        // grid fills whole rounds of the 256 CUs: +35 % at 4096^3, +7 % at 35840x1536x4608; -25 % at K = 1152 and
        // -10 % on the split-K weight gradients, which therefore stay on the 128x128 kernel.
        const double fill = (double)p.M * p.N / ((double)tm * 256 * tn * 256);
        const double rounds = (double)blocks / 256.0;
        const double quant = rounds / (double)((blocks + 255) / 256);   // last-round occupancy of the 256 CUs
        if (ly == 2 || fill < 0.9 || blocks < 200 || quant < 0.8) return -1;
        if ((long)p.K / p.splitk < 4096) return -1;
    }
    GP pv = p;
    pv.tiles_n = (int)tn;
    pv.ntiles = (int)(tm * tn);
    dim3 grid((unsigned)(tm * tn * batch * p.splitk));
#define V(LY, CV)                                 \
    if (ly == LY && conv == CV) {                 \
        launch_t256<LY, CV>(pv, grid, stream);    \
        a3t_note_kernel("gemm_bf16_t256_kernel<%d, %s>", LY, CV ? "true" : "false"); \
        return (int)hipGetLastError();            \
    }
    V(0, false) V(0, true) V(1, false) V(1, true) V(2, false) V(2, true)
#undef V
    return -1;
}
