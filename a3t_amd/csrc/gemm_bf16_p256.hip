// gemm_bf16_p256.hip -- PERSISTENT 256x256x32 bf16 MFMA GEMM for gfx950 (NT / NN operand layouts, optional im2col).
// Experiment, opt-in (A3T_GEMM_P256 / a3t_gemm_p256_mode): correct on every shape it accepts, but it only ties the
// 128x128 kernel on the model's GEMMs (DESIGN.md 4.1 has the measurements and what they rule out).
//
// One 512-thread workgroup per CU walks a LIST of 256x256 output tiles (round k -> tile (k*8 + xcd)*ncu + cu, so the CUs of
// one XCD multiply neighbouring tiles at the same time); wave (wr, wc) of the 2x4 grid owns a 128x64 block = 128 fp32
// accumulators.  What is different from the one-tile-per-workgroup 256x256 kernel (gemm_bf16_t256.hip):
//   * operands stream L2 -> LDS by DMA through a 4-stage ring of 32-wide K-steps (32 KiB each), three stages in flight;
//     the DMA cursor runs ahead of the MFMA cursor ACROSS tile boundaries, so a tile has no prologue;
//   * ONE barrier per K-step; every wave interleaves its own LDS reads (two alternating fragment sets), its share of the
//     DMA issue and its MFMAs in program order (sched_group_barrier) instead of the ping-pong specialisation, because a
//     wave that only reads LDS next to a wave that only issues MFMAs on the same SIMD slows BOTH down
//     (tools/probes/mfma_lds_probe.hip: 44 vs 19 ns per MFMA, 69 vs 302 B/ns of LDS reads);
//   * the epilogue runs straight from the registers: the MFMA operands are swapped (D = B-fragment x A-fragment) so a lane
//     owns one output row and four consecutive columns per register group -- bias (from LDS, N <= 6144), relu, ReLU' mask,
//     dropout, residual, column sums, 8- / 16-byte stores, no LDS round trip and no extra barrier; the next tile's first
//     stages are already in flight while it runs;
//   * the stores share the in-order vector-memory counter with the DMAs, so for the three K-steps after a tile boundary
//     the counted wait is raised by the 32 stores per wave (only when that number is certain; over-waiting is safe);
//   * optional per-workgroup K rotation (k = (koff_c + s) mod nk) de-synchronises the L2 misses of the CUs that share an
//     operand slab (measured: no gain).
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

static __device__ __attribute__((aligned(16))) unsigned int p256_zero_page[16];

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))
#define BAR()                                        \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        asm volatile("s_barrier" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

enum { L_NT = 0, L_NN = 1 };

struct P256Plan {
    int rotate;  // 1: per-workgroup K rotation
    int rounds;  // tiles per workgroup (last round may be partial)
    int ntot;    // output tiles (all batch elements)
};

constexpr int P256_RING = 4 * 2 * 256 * 32 * 2;   // 4 stages x (A + B image of one 32-wide K-step) = 128 KiB
constexpr int P256_MAX_BIAS = 6144;

// Register epilogue for 4 consecutive columns of one row.  Lean by construction (it is inlined 32 times per tile and has
// to stay inside the instruction cache next to the K loop): only the activations the model's GEMMs use (none / relu);
// RICH adds the ReLU'(S) mask, the residual, fused column sums and fp32 read-modify-write.
template <bool RICH>
__device__ __forceinline__ void p256_epilogue4(const GP& p, float4 v, int64_t idx, const float4& b4, float4& cs) {
    v.x += b4.x, v.y += b4.y, v.z += b4.z, v.w += b4.w;
    if (p.act == A3T_ACT_RELU) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    if (RICH && p.S) {
        float4 sv;
        if (p.s_dtype == A3T_BF16) {
            uint2 t = *(const uint2*)((const unsigned short*)p.S + idx);
            sv = make_float4(bf2f(t.x & 0xffff), bf2f(t.x >> 16), bf2f(t.y & 0xffff), bf2f(t.y >> 16));
        } else {
            sv = *(const float4*)(p.S + idx);
        }
        v.x = sv.x > 0.f ? v.x : 0.f, v.y = sv.y > 0.f ? v.y : 0.f;
        v.z = sv.z > 0.f ? v.z : 0.f, v.w = sv.w > 0.f ? v.w : 0.f;
    }
    if (p.drop_inv > 0.f) {
        bool kp[4];
        rng_keep4(p.drop_key, (unsigned int)idx, p.drop_thr, kp);
        v.x = kp[0] ? v.x * p.drop_inv : 0.f, v.y = kp[1] ? v.y * p.drop_inv : 0.f;
        v.z = kp[2] ? v.z * p.drop_inv : 0.f, v.w = kp[3] ? v.w * p.drop_inv : 0.f;
    }
    v.x *= p.alpha, v.y *= p.alpha, v.z *= p.alpha, v.w *= p.alpha;
    if (RICH && p.R) {
        float4 rv = *(const float4*)(p.R + idx);
        v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
    }
    if (RICH) cs.x += v.x, cs.y += v.y, cs.z += v.z, cs.w += v.w;
    if (p.c_dtype == A3T_BF16) {
        uint2 o;
        o.x = io_pack2(v.x, v.y);
        o.y = io_pack2(v.z, v.w);
        *(uint2*)((unsigned short*)p.C + idx) = o;
    } else {
        float* C = (float*)p.C + idx;
        if (RICH && p.accumulate == A3T_ACC_ADD) {
            float4 o = *(const float4*)C;
            v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
        }
        *(float4*)C = v;
    }
}

template <int LAYOUT, bool CONV, bool RICH>
__global__ __launch_bounds__(512, 2) void gemm_bf16_p256_kernel(GP p, P256Plan pl) {
    constexpr int BK = 32, IMG = 256 * BK * 2, STAGE = 2 * IMG;   // 16 KiB A image + 16 KiB B image per K-step
    constexpr bool B_KC = (LAYOUT == L_NT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [4 stages][A | B] | bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;          // wave tile: rows wr*128 .. +127, columns wc*64 .. +63
    const int G = gridDim.x, ncu = G >> 3, xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
    const int nk = p.K / BK;
    const u16* ZP = (const u16*)p256_zero_page;

    // work list: round k -> tile (k * 8 + xcd) * ncu + cu: the CUs of one XCD multiply neighbouring tiles at the same time
    auto tile_of = [&](int k) -> int { return (k * 8 + xcd) * ncu + cu; };
    int nit = 0;
    while (nit < pl.rounds && tile_of(nit) < pl.ntot) ++nit;
    if (nit == 0) return;

    if (p.bias)
        for (int i = tid; i < p.N; i += 512) ((float*)(smem + P256_RING))[i] = p.bias[i];

    // ---- DMA cursor: (item, K-step) of the next stage to be issued ------------------------------------------------
    // K rotation: workgroup c walks the K-steps of every tile as k = (koff_c + s) mod nk.  Without it the 32 CUs of an XCD
    // touch every operand line of a K-step at the same time, so every DMA is an L2 miss or a hit-under-miss (the whole
    // XCD waits ~2.5 us per step); with distinct offsets one CU misses a line and the 5-6 others that share its operand
    // slab find it in the L2 later.  (The fp32 accumulation order changes with the offset: results are equal to rounding.)
    int koff = 0;
    if (pl.rotate) koff = (int)((((unsigned)cu * 2654435769u) >> 16) * (unsigned)nk >> 16);
    int i_item = 0, i_kt = 0, i_cnt = 0, i_tap = 0, i_cc = 0, i_flat = 0;
    bool i_live = true;
    const u16* pa[2];      // A piece 2w+q: tile row 16 (2w+q) + lane/4, swizzled 16-B chunk, K-step 0
    const u16* pb[2];      // NT: same for B rows (n);  NN: piece 2w+q = 128-column half (2w+q)/8, k-rows 4 ((2w+q)%8) + lane/16
    unsigned okbits = 0;   // bit q: A row valid; bit 2+q: B row / column chunk valid
    int tpos[2] = {0, 0};
    auto setup_item = [&](int k) __attribute__((always_inline)) {
        i_live = k < nit;
        i_kt = koff, i_cnt = 0, i_tap = 0, i_cc = 0;
        if (CONV && p.taps > 1) {
            i_tap = (koff * BK) / p.Kc;
            i_cc = koff * BK - i_tap * p.Kc;
        }
        if (!i_live) return;
        const int tile = tile_of(k);
        const int bid = tile % p.ntiles, bz = tile / p.ntiles;
        const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
        const int z0 = bz / p.batch_inner, z1 = bz % p.batch_inner;
        const u16* A = (const u16*)p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
        const u16* B = (const u16*)p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
        okbits = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (w * 2 + q) * 16 + (lane >> 2);
            const int chunk = ((lane & 3) ^ ((r >> 2) & 3)) * 8;
            const int m = tm * 256 + r;
            pa[q] = A + (int64_t)m * p.a_rs + chunk;
            if (m < p.M) okbits |= 1u << q;
            if (CONV) tpos[q] = m % p.Tseq;
            if (B_KC) {
                const int n = tn * 256 + r;
                pb[q] = B + (int64_t)n * p.b_rs + chunk;
                if (n < p.N) okbits |= 4u << q;
            } else {
                const int g = w * 2 + q, kr = (g & 7) * 4 + (lane >> 4);
                const int col = tn * 256 + (g >> 3) * 128 + (((lane & 15) ^ ((kr & 3) << 2)) * 8);
                pb[q] = B + col + (int64_t)kr * p.b_cs;
                if (col < p.N) okbits |= 4u << q;
            }
        }
    };
    auto issue_stage = [&]() __attribute__((always_inline)) {
        unsigned char* sA = smem + (i_flat & 3) * STAGE;
        unsigned char* sB = sA + IMG;
        int shift = 0;
        int64_t aoff, boff;
        if (CONV && p.taps > 1) {
            shift = (i_tap - p.pad) * p.dil;
            aoff = (int64_t)shift * p.a_rs + i_cc;
            boff = B_KC ? (int64_t)i_kt * BK : (int64_t)i_tap * p.b_ts + (int64_t)i_cc * p.b_cs;
        } else {
            aoff = (int64_t)i_kt * BK;
            boff = B_KC ? (int64_t)i_kt * BK : (int64_t)i_kt * BK * p.b_cs;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bool v = i_live && ((okbits >> q) & 1);
            if (CONV && p.taps > 1) v = v && ((unsigned)(tpos[q] + shift) < (unsigned)p.Tseq);
            const u16* s = v ? pa[q] + aoff : ZP;
            __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(sA + (w * 2 + q) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool v = i_live && ((okbits >> (2 + q)) & 1);
            const u16* s = v ? pb[q] + boff : ZP;
            __builtin_amdgcn_global_load_lds(GLB_AS(s), LDS_AS(sB + (w * 2 + q) * 1024), 16, 0, 0);
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {   // (branchy: kept out of the scheduled regions)
        ++i_flat;
        if (i_live) {
            ++i_kt;
            if (CONV && p.taps > 1) {
                i_cc += BK;
                if (i_cc >= p.Kc) i_cc -= p.Kc, ++i_tap;
            }
            if (i_kt == nk) i_kt = 0, i_tap = 0, i_cc = 0;
            if (++i_cnt == nk) setup_item(++i_item);
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
    // fragment set of one 16-wide sub-step: 4 A row blocks + 2 B column blocks (24 registers); two sets alternate so the
    // LDS reads of a sub-step are in flight while the MFMAs of the previous one issue -- each wave overlaps its OWN reads
    // and MFMAs (a wave that only reads next to a wave that only multiplies on the same SIMD slows both down:
    // tools/probes/mfma_lds_probe.hip).
    bf16x8 fa[2][4], fb[2][2];
    auto read_frags = [&](const int set, int stage, const int kk) __attribute__((always_inline)) {
        const unsigned char* sA = smem + stage * STAGE;
        const unsigned char* sB = sA + IMG;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            fb[set][j] = B_KC ? frag_kc(sB, wc * 64 + 32 * j + lr, kk)
                              : frag_rc(sB + (wc >> 1) * (IMG / 2), (wc & 1) * 64 + 32 * j, kk);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = frag_kc(sA, wr * 128 + 32 * i + lr, kk);
    };
    // swapped operands: D[n][m] -- lane = output row m, register r = column (r & 3) + 8 * (r >> 2) + 4 * lk
    auto mfma_block = [&](const int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };

    // ---- epilogue of the tile the MFMA cursor just finished: straight from the registers ---------------------------
    auto epilogue = [&](int k) __attribute__((always_inline)) -> bool {
        const int tile = tile_of(k);
        const int bid = tile % p.ntiles, bz = tile / p.ntiles;
        const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
        const int z0 = bz / p.batch_inner, z1 = bz % p.batch_inner;
        const int64_t zoff = z0 * p.c_bs0 + z1 * p.c_bs1;
        const int row0 = tm * 256 + wr * 128 + lr;
        const int col0 = tn * 256 + wc * 64 + 4 * lk;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // The bias reads are inline asm on purpose: a compiler-visible LDS read here makes the waitcnt pass drain every
            // DMA in flight first (it cannot tell the bias area from the operand ring the DMAs write).
            f32x4 b4[4];
            if (p.bias) {
                const unsigned addr = P256_RING + (unsigned)(col0 + 32 * j) * 4u;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                             "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(b4[0]), "=&v"(b4[1]), "=&v"(b4[2]), "=&v"(b4[3]) : "v"(addr) : "memory");
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) b4[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            float4 cs[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) cs[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 32 * i;
                const int64_t rbase = zoff + (int64_t)row * p.c_rs + (col0 + 32 * j);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = col0 + 32 * j + 8 * g;
                    float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    if (row < p.M && col < p.N)
                        p256_epilogue4<RICH>(p, v, rbase + 8 * g, make_float4(b4[g][0], b4[g][1], b4[g][2], b4[g][3]), cs[g]);
                    acc[i][j][4 * g] = 0.f, acc[i][j][4 * g + 1] = 0.f, acc[i][j][4 * g + 2] = 0.f, acc[i][j][4 * g + 3] = 0.f;
                }
            }
            if (RICH && p.colsum) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 c = cs[g];
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) {
                        c.x += __shfl_xor(c.x, o, 64), c.y += __shfl_xor(c.y, o, 64);
                        c.z += __shfl_xor(c.z, o, 64), c.w += __shfl_xor(c.w, o, 64);
                    }
                    const int col = col0 + 32 * j + 8 * g;
                    if (lr == 0 && col < p.N) {
                        float* o = p.colsum + z1 * p.colsum_bs1 + col;
                        if (p.colsum_slots > 1) o += (int64_t)((tm + z0) % p.colsum_slots) * p.colsum_ss;
                        atomicAdd(o + 0, p.colsum_scale * c.x), atomicAdd(o + 1, p.colsum_scale * c.y);
                        atomicAdd(o + 2, p.colsum_scale * c.z), atomicAdd(o + 3, p.colsum_scale * c.w);
                    }
                }
            }
        }
        // exactly 32 stores per wave were issued (one per 4-column group): full tile, plain store epilogue
        return !RICH && (tm * 256 + 256 <= p.M) && (tn * 256 + 256 <= p.N);
    };

    // ---- K loop over the flat (tile, K-step) sequence: 4-stage ring, three stages in flight, ONE barrier per K-step ----
    //   read set 1 <- (step s, sub-step 1) | 8 MFMAs on set 0 | wait DMA(s+1) + barrier | issue DMA(s+4) into the slot of
    //   step s (every wave has read it) | read set 0 <- (step s+1, sub-step 0) | 8 MFMAs on set 1
    // The epilogue's 32 stores per wave enter the same in-order memory counter as the DMAs: for the three K-steps whose
    // awaited stage is older than those stores the count is raised by 32 (only when the number of stores is certain:
    // full tile, plain-store epilogue; otherwise the ordinary count over-waits, which is safe).
    setup_item(0);
    issue_stage(), advance();
    issue_stage(), advance();
    issue_stage(), advance();
    issue_stage(), advance();
    WAIT_VM(12);                // stage 0 has landed (this wave's pieces); the bias stores to LDS are older
    BAR();
    read_frags(0, 0, 0);
    int c_item = 0, c_kt = 0, after_stores = 0;
    const int nflat = nit * nk;
    // MFMA = 0x008, DS read = 0x100, VMEM read = 0x020 (sched_group_barrier masks).  A wave is blocked while its MFMA waits for
    // the matrix pipe, so everything else it has to issue is slotted BETWEEN its MFMAs, where it is free.
    constexpr int NDS = B_KC ? 6 : 8;     // LDS reads of one fragment set (transposed B fragments take two reads each)
    for (int n = 0; n < nflat; ++n) {
        const int st = n & 3;
        read_frags(1, st, 1);
        mfma_block(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave has finished reading stage st
        if (after_stores > 0) {
            WAIT_VM(40);
            --after_stores;
        } else {
            WAIT_VM(8);         // DMA(n+1) landed; (n+2), (n+3) stay in flight
        }
        BAR();
        issue_stage();          // DMA(n+4) -> the slot of step n
        read_frags(0, (n + 1) & 3, 0);
        mfma_block(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (i < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        advance();
        if (++c_kt == nk) {
            c_kt = 0;
            after_stores = (epilogue(c_item++) && nk >= 4) ? 3 : 0;
        }
    }
    WAIT_VM(0);                 // the trailing zero-page copies must not outlive the workgroup's LDS allocation
}

template <int LY, bool CV, bool RICH>
static void launch_p256(const GP& pv, const P256Plan& pl, int G, hipStream_t stream) {
    constexpr int lds = P256_RING + P256_MAX_BIAS * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_p256_kernel<LY, CV, RICH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_bf16_p256_kernel<LY, CV, RICH>), dim3(G), dim3(512), lds, stream, pv, pl);
}

// Called by a3t_gemm_bf16_glds after the alignment contract has been checked (pv.epi_vec resolved).
// Returns -1 when this kernel is not applicable / not selected for the shape.
static int g_p256_mode = -1;   // A3T_GEMM_P256 = 0 / unset: never, 1: whenever legal, 2: heuristic
// Run-time override of A3T_GEMM_P256 (tests and A/B benches compare both kernels inside one process); returns the old mode.
extern "C" int a3t_gemm_p256_mode(int mode) {
    const int old = g_p256_mode;
    g_p256_mode = mode;
    return old;
}

int a3t_gemm_bf16_p256(const GP& p, int batch, int ly, hipStream_t stream) {
    int& mode = g_p256_mode;
    static int ncus = 0;
    if (mode < 0) {
        const char* e = getenv("A3T_GEMM_P256");
        mode = e ? atoi(e) : 0;
    }
    if (mode == 0) return -1;
    if (ncus == 0) {   // (one process drives one GPU: trainer.py, bench.py)
        ncus = 256;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncus = prop.multiProcessorCount;
        const char* g = getenv("A3T_GEMM_P256_WGS");
        if (g) ncus = atoi(g);
        ncus &= ~7;
    }
    if (ncus < 8) return -1;
    if (ly != L_NT && ly != L_NN) return -1;
    if (p.K % 32 != 0 || p.splitk != 1 || p.kshift_mode) return -1;
    if (!p.epi_vec || p.accumulate == A3T_ACC_ATOMIC) return -1;
    if (p.act != A3T_ACT_NONE && p.act != A3T_ACT_RELU) return -1;
    const bool rich = p.S || p.R || p.colsum || p.accumulate == A3T_ACC_ADD;
    if (p.bias && p.N > P256_MAX_BIAS) return -1;
    const bool conv = p.taps > 1;
    if (conv) {
        if (p.Kc % 32 != 0 || p.Tseq <= 0) return -1;
        if (ly == L_NT && p.b_ts != p.Kc) return -1;
    }
    const long tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
    const long tiles = tm * tn * batch;
    if (tiles > (1l << 30)) return -1;
    if (mode == 2) {
        const double fill = (double)p.M * p.N / ((double)tm * 256 * tn * 256);
        if (fill < 0.9 || tiles < 2 * ncus) return -1;
    }
    GP pv = p;
    pv.tiles_n = (int)tn;
    pv.ntiles = (int)(tm * tn);
    P256Plan pl;
    pl.ntot = (int)tiles;
    pl.rounds = (int)((tiles + ncus - 1) / ncus);
    static int rot = -1;
    if (rot < 0) {
        const char* e = getenv("A3T_GEMM_P256_ROTATE");
        rot = e ? atoi(e) : 1;
    }
    pl.rotate = rot;
#define V(LY, CV, RC)                                    \
    if (ly == LY && conv == CV && rich == RC) {           \
        launch_p256<LY, CV, RC>(pv, pl, ncus, stream);    \
        a3t_note_kernel("gemm_bf16_p256_kernel<%d, %s, %s>", LY, CV ? "true" : "false", RC ? "true" : "false"); \
        return (int)hipGetLastError();                    \
    }
    V(L_NT, false, false) V(L_NT, true, false) V(L_NN, false, false) V(L_NN, true, false)
    V(L_NT, false, true) V(L_NT, true, true) V(L_NN, false, true) V(L_NN, true, true)
#undef V
    return -1;
}
