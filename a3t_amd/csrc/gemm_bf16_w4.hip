// gemm_bf16_w4.hip -- 256x128x32 bf16 MFMA GEMM for gfx950: 4 waves x (128x64 outputs = 128 accumulator registers),
// operands streamed L2 -> LDS by DMA through a 3-stage ring, TWO independent workgroups per CU.
//
// Why this shape (measured on MI355X):
//   * the 128x128 kernel (gemm_bf16.hip: one LDS buffer, 4 workgroups per CU) needs 64 B of operand DMA per MFMA cycle and
//     is bound by the DMA round trip per K-step (MFMA pipe 37 % busy);
//   * the 256x256 8-wave ping-pong kernel (gemm_bf16_t256.hip) halves the bytes per flop, but ONE 128-KiB workgroup owns
//     the CU: nothing overlaps its prologue and its epilogue (bias / relu / dropout-RNG on 128 outputs per thread is
//     ~10k cycles of pure VALU), which costs more than the main loop gains at K = 1152.
//   Here a workgroup has 72 KiB of LDS and 4 waves of 256 registers, so two of them share a CU (2 waves per SIMD): while
//   one is in its epilogue or prologue the other one's MFMAs keep the pipe busy, with 0.75x the DMA bytes per flop of the
//   128x128 tile.  Inside a workgroup the K loop is a 3-deep ring of 32-wide K-steps (24 KiB each): step t+2 is issued
//   right after the barrier that retires step t-1's readers, and only `s_waitcnt vmcnt(6)` (one step in flight) + a raw
//   s_barrier separate the steps -- the DMA queue is never drained inside the loop.
//   * MFMA operands are swapped (acc = B_frag x A_frag), so a lane ends up with ONE output row and groups of 4 consecutive
//     output columns: the fused epilogue (bias, relu, relu-mask, dropout, residual, column sums, bf16 / fp32 store, or
//     split-K atomics) runs straight from the accumulators with 8 / 16-byte accesses -- no LDS staging, no barrier, and the
//     LDS ring is free for the next workgroup's prologue as soon as the loop ends.
// LDS images are lane-linear (DMA), the bank swizzle is an XOR on the per-lane SOURCE chunk and on the fragment read:
//   k-contiguous operand ([rows][32 k], 64-B rows): 16-B chunk c of row r is stored at chunk c ^ ((r >> 2) & 3);
//   row-contiguous operand ([32 k][128 n], 256-B k-rows, read with ds_read_b64_tr_b16): chunk c of k-row kr at
//   c ^ ((kr & 3) << 2) (the layout of gemm_bf16.hip's 128-column images).
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

static __device__ __attribute__((aligned(16))) unsigned int w4_zero_page[16];

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))
#define W4_BAR()                                     \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        asm volatile("s_barrier" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
#define W4_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

enum { W4_NT = 0, W4_NN = 1 };

// LAYOUT W4_NT: A [m][k], B [n][k] (both k-contiguous; Linear / Conv1d forward with weights [n][tap][c]);
//        W4_NN: A [m][k], B [k][n] (data gradients: B = W viewed [(tap, c)][n]).
// CONV: implicit im2col on A (taps > 1, channel count a multiple of 32: one uniform tap per K-step).
template <int LAYOUT, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_bf16_w4_kernel(GP p) {
    constexpr int BK = 32, A_BYTES = 256 * BK * 2, B_BYTES = 128 * BK * 2, STAGE = A_BYTES + B_BYTES, NST = 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [3][A image 16 KiB | B image 8 KiB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    int wi = blockIdx.x;
    {   // workgroup b runs on XCD b % 8: contiguous runs of work items per XCD (tiles that share operand slabs share an L2)
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
    const u16* ZP = (const u16*)w4_zero_page;

    const int ksteps = p.K / BK;                 // (host contract: K % 32 == 0)
    const int per = (ksteps + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per, kt1 = min(ksteps, kt0 + per);
    if (kt0 >= kt1) return;
    const int nk = kt1 - kt0;

    // ---- per-lane DMA sources ------------------------------------------------------------------------------------------
    // A: piece g = 4 w + q of 16 covers tile rows 16 g + lane / 4, chunk position lane % 4
    const u16* a_src[4];
    bool a_ok[4];
    int a_tp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (w * 4 + q) * 16 + (lane >> 2), m = tm * 256 + r;
        a_ok[q] = m < p.M;
        a_tp[q] = CONV ? (m % p.Tseq) : 0;
        a_src[q] = A + (int64_t)m * p.a_rs + (((lane & 3) ^ ((r >> 2) & 3)) * 8);
    }
    // B: NT: piece g = 2 w + q of 8 covers tile rows (n) 16 g + lane / 4;  NN: piece g covers k-rows 4 g + lane / 16
    const u16* b_src[2];
    bool b_ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g = w * 2 + q;
        if (LAYOUT == W4_NT) {
            const int r = g * 16 + (lane >> 2), n = tn * 128 + r;
            b_ok[q] = n < p.N;
            b_src[q] = B + (int64_t)n * p.b_rs + (((lane & 3) ^ ((r >> 2) & 3)) * 8);
        } else {
            const int kr = g * 4 + (lane >> 4), col = tn * 128 + (((lane & 15) ^ ((kr & 3) << 2)) * 8);
            b_ok[q] = col < p.N;
            b_src[q] = B + col + (int64_t)kr * p.b_cs;
        }
    }
    // uniform (tap, channel) of the next K-step to be issued (CONV: Kc % 32 == 0, so a K-step lies inside one tap)
    int u_tap = 0, u_cc = 0, u_k = kt0 * BK;
    if (CONV) {
        u_tap = u_k / p.Kc;
        u_cc = u_k - u_tap * p.Kc;
    }
    // issue K-step `t` (relative to kt0) into ring slot t % 3; called for consecutive t; past the range: zero page
    auto issue = [&](int t) __attribute__((always_inline)) {
        unsigned char* sA = smem + (t % NST) * STAGE;
        unsigned char* sB = sA + A_BYTES;
        const bool live = t < nk;
        if (CONV) {
            const int off = (u_tap - p.pad) * p.dil;
            const int64_t aoff = (int64_t)off * p.a_rs + u_cc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool v = live && a_ok[q] && ((unsigned)(a_tp[q] + off) < (unsigned)p.Tseq);
                const u16* s = v ? a_src[q] + aoff : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(sA + (w * 4 + q) * 1024), 16, 0, 0);
            }
            const int64_t boff = (LAYOUT == W4_NT) ? (int64_t)u_k : (int64_t)u_tap * p.b_ts + (int64_t)u_cc * p.b_cs;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u16* s = (live && b_ok[q]) ? b_src[q] + boff : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(sB + (w * 2 + q) * 1024), 16, 0, 0);
            }
            u_cc += BK, u_k += BK;
            if (u_cc >= p.Kc) u_cc -= p.Kc, ++u_tap;
        } else {
            const int64_t boff = (LAYOUT == W4_NT) ? (int64_t)u_k : (int64_t)u_k * p.b_cs;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u16* s = (live && a_ok[q]) ? a_src[q] + u_k : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(sA + (w * 4 + q) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u16* s = (live && b_ok[q]) ? b_src[q] + boff : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(sB + (w * 2 + q) * 1024), 16, 0, 0);
            }
            u_k += BK;
        }
    };

    f32x16 acc[4][2];   // [row block of 32 inside the wave's 128 rows][column block of 32 inside its 64 columns]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lk = lane >> 5;
    auto frag_kc = [&](const unsigned char* img, int row, int kk) -> bf16x8 {    // 64-byte rows
        return *(const bf16x8*)(img + row * 64 + ((((kk * 2 + lk) ^ ((row >> 2) & 3))) << 4));
    };
    auto frag_rc = [&](const unsigned char* img, int row0, int kk) -> bf16x8 {   // 256-byte k-rows (128 columns)
        const int g = lane >> 4, pp = lane & 15;
        const int col = row0 + (g & 1) * 16 + (pp & 3) * 4;
        const int kb = kk * 16 + (g >> 1) * 8 + (pp >> 2), k1 = kb + 4;
        const unsigned char* a0 = img + kb * 256 + ((((col >> 3) ^ ((kb & 3) << 2))) << 4) + (col & 7) * 2;
        const unsigned char* a1 = img + k1 * 256 + ((((col >> 3) ^ ((k1 & 3) << 2))) << 4) + (col & 7) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a1));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    issue(0);
    issue(1);
    for (int t = 0; t < nk; ++t) {
        W4_WAIT_VM(6);            // K-step t has landed (this wave's pieces); step t+1 stays in flight
        W4_BAR();                 // ... everyone's pieces; and every wave is done reading the slot of step t-1
        issue(t + 2);             // -> that slot
        const unsigned char* sA = smem + (t % NST) * STAGE;
        const unsigned char* sB = sA + A_BYTES;
        bf16x8 fa[4][2], fb[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fb[j][kk] = (LAYOUT == W4_NT) ? frag_kc(sB, wc * 64 + 32 * j + lr, kk) : frag_rc(sB, wc * 64 + 32 * j, kk);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[i][kk] = frag_kc(sA, wr * 128 + 32 * i + lr, kk);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
    }
    W4_WAIT_VM(0);                // the trailing zero-page copies must not outlive the workgroup's LDS allocation

    // ---- epilogue straight from the accumulators: lane = output row, 4 consecutive columns per register group ---------
    // acc[i][j][r]: row m = tm*256 + wr*128 + 32 i + (lane & 31); column n = tn*128 + wc*64 + 32 j + 8 (r>>2) + 4 (lane>>5) + (r&3)
    const int row0 = tm * 256 + wr * 128 + lr;
    const int col0 = tn * 128 + wc * 64 + 4 * lk;
    // (host contract: vector epilogue only -- scalar / atomic epilogues stay on the 128x128 kernel)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = col0 + 32 * j + 8 * g;
            const bool col_ok = col < p.N;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && ks == 0 && col_ok) bias4 = *(const float4*)(p.bias + col);
            float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 32 * i;
                if (row < p.M && col_ok) {
                    const float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    epilogue_vec4(p, v, zoff + (int64_t)row * p.c_rs + col, bias4, ks, cs);
                }
            }
            if (p.colsum) {   // the 32 lanes of a half-wave hold 32 different rows of the same 4 columns
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    cs.x += __shfl_xor(cs.x, o, 64), cs.y += __shfl_xor(cs.y, o, 64);
                    cs.z += __shfl_xor(cs.z, o, 64), cs.w += __shfl_xor(cs.w, o, 64);
                }
                if (lr == 0 && col_ok) {
                    float* o = p.colsum + z1 * p.colsum_bs1 + col;
                    if (p.colsum_slots > 1) o += (int64_t)((tm + z0) % p.colsum_slots) * p.colsum_ss;
                    atomicAdd(o + 0, p.colsum_scale * cs.x), atomicAdd(o + 1, p.colsum_scale * cs.y);
                    atomicAdd(o + 2, p.colsum_scale * cs.z), atomicAdd(o + 3, p.colsum_scale * cs.w);
                }
            }
        }
}

template <int LY, bool CV>
static void launch_w4(const GP& pv, dim3 grid, hipStream_t stream) {
    constexpr int lds = 3 * (256 + 128) * 32 * 2;   // 72 KiB
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_w4_kernel<LY, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_bf16_w4_kernel<LY, CV>), grid, dim3(256), lds, stream, pv);
}

// Called by a3t_gemm_bf16_glds after the alignment contract has been checked (pv.epi_vec resolved).  ly: 0 = NT, 1 = NN.
// Returns -1 when this kernel is not applicable / not chosen for the shape.
int a3t_gemm_bf16_w4(const GP& p, int batch, int ly, hipStream_t stream) {
    // Measured on MI355X (tools/ffn_gemm_bench.py, tools/w4_probe.py): slower than the 128x128 kernel on every shape of the
    // model (35840x1536x1152: 257 vs 187 us; x384x4608: 194 vs 180 us) -- its register epilogue stores 16-byte row pieces
    // (32 cache lines per instruction) and two 24-KiB-per-step workgroups per CU keep fewer DMA bytes in flight than four
    // 128x128 ones.  Kept as an experiment: A3T_GEMM_W4 = 1: whenever legal, 2: heuristic; default: never.
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("A3T_GEMM_W4");
        mode = e ? (e[0] == '1' ? 1 : (e[0] == '2' ? 2 : 0)) : 0;
    }
    if (mode == 0 || ly > 1) return -1;
    if (p.K % 32 != 0 || p.kshift_mode) return -1;
    const bool conv = p.taps > 1;
    if (conv) {
        if (p.Kc % 32 != 0) return -1;
        if (ly == 0 && p.b_ts != p.Kc) return -1;            // NT conv: weights [n][tap][c] contiguous
    }
    if (!p.epi_vec || p.accumulate == A3T_ACC_ATOMIC || p.splitk != 1) return -1;   // vector epilogue only
    const long tm = (p.M + 255) / 256, tn = (p.N + 127) / 128;
    const long blocks = tm * tn * batch * p.splitk;
    if (mode == 2) {
        // two workgroups per CU: worth it when the grid fills the 512 slots at least once and tiles are mostly full
        const double fill = (double)p.M * p.N / ((double)tm * 256 * tn * 128);
        if (fill < 0.85 || blocks < 400 || p.K / p.splitk < 256) return -1;
    }
    GP pv = p;
    pv.tiles_n = (int)tn;
    pv.ntiles = (int)(tm * tn);
    dim3 grid((unsigned)blocks);
#define V(LY, CV)                                \
    if (ly == LY && conv == CV) {                \
        launch_w4<LY, CV>(pv, grid, stream);     \
        a3t_note_kernel("gemm_bf16_w4_kernel<%d, %s>", LY, CV ? "true" : "false"); \
        return (int)hipGetLastError();           \
    }
    V(0, false) V(0, true) V(1, false) V(1, true)
#undef V
    return -1;
}
