// gemm_bf16.hip -- bf16-operand MFMA GEMM for gfx950 with direct global->LDS staging.
//
// The production contraction kernel of the bf16 compute path (activations feeding GEMMs are
// stored in bf16, weights are cast once per step).  Per (64*WM) x 128 x 64 tile:
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip);
//     LDS images are lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
//     address and undone on the fragment read (same XOR involution on both sides);
//   * k-contiguous operands ([rows][k], NT GEMMs and the im2col conv loader) are read back with
//     conflict-free ds_read_b128; reduction-strided operands ([k][rows]: W for data gradients,
//     activations for weight gradients) are read with ds_read_b64_tr_b16, the CDNA4 LDS
//     transpose read, so no register transposes are needed for the NN / TN layouts;
//   * 2*WM waves, each owning a 64x64 output sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.
//     WM=2: 128x128 tile, 4 waves.  WM=4: 256x128 tile, 8 waves -- 25 % fewer DMA bytes per flop
//     (the kernel is bound by the L2->LDS DMA rate, ~11-13 TB/s aggregate, not by the MFMA pipe);
//   * STAGES=1: one LDS buffer, latency hidden by 3-5 co-resident workgroups per CU;
//     STAGES=2: double buffered inside the workgroup (used when the grid is small).
// Out-of-range rows / im2col padding / K tails read a 16-byte device zero page instead of
// branching around the DMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/a3t_hip.h"
#include "gemm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __attribute__((aligned(16))) unsigned int a3t_zero_page[16];

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))

enum { L_NT = 0, L_NN = 1, L_TN = 2 };

// -DGLDS_TIMING (A3T_EXTRA_FLAGS): per-workgroup phase times of the single-buffer K loop (round-2 tool, see profiles/NOTEBOOK_r01_r03.md)
#ifdef GLDS_TIMING
__device__ unsigned long long glds_dbg[16384 * 8];
extern "C" int a3t_debug_read_glds(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(glds_dbg), bytes); }
#define GT_NOW() __builtin_readcyclecounter()
#endif

// CONV = 0: plain GEMM (taps == 1, no token shift): the im2col / shift bookkeeping is compiled out, which brings the
// single-buffer variant under 128 VGPRs -> 4 workgroups per CU (1024 slots: the 840-tile N = 384 GEMMs run in one round)
// CONV = 1: conv / data-gradient GEMM whose channel count is a multiple of BK (one tap per K-tile, uniform tracking only);
// CONV = 2: generic per-lane (tap, channel) tracking, fused conv weight gradient, token-shifted weight gradient.
// WN = 2: 128 output columns per tile; WN = 3: 192 (each wave 64x96 = 2x3 accumulators, 3 workgroups per CU) for
// N = 192 operands (attention PV / dV / dK / d(q+v) with d_k = 192): no half-empty second column tile, the A operand
// (the T x T probabilities) is read once instead of twice.
template <int LAYOUT, int STAGES, int WM, int CONV, int WN = 2>
__global__ __launch_bounds__(128 * WM, (WN == 3 ? (STAGES == 2 ? 2 : 3) : (STAGES == 2 ? 2 : (CONV == 2 ? 3 : 4)))) void gemm_bf16_glds_kernel(GP p) {
    const int TAPS = CONV ? p.taps : 1;
    const int KSM = CONV ? p.kshift_mode : 0;
    constexpr int NW = 2 * WM;                 // waves
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 64;
    constexpr bool A_KC = (LAYOUT != L_TN), B_KC = (LAYOUT == L_NT);
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NA = 4;                      // A DMA instructions per wave per tile (BM*128 B / 1 KiB / NW)
    constexpr int NB = (BN / 8) / NW;          // B DMA instructions per wave per tile
    // row-contiguous images: one k-row = (rows*2) bytes; a 1-KiB DMA instruction covers KPI k-rows
    constexpr int A_LPR = BM / 8, A_KPI = 64 / A_LPR;   // lanes per k-row, k-rows per instruction
    constexpr int B_LPR = BN / 8;                       // (24 for the 192-column image: 2.67 k-rows per instruction)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [STAGES][A image | B image]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;      // (an L2-blocked tile order -- column groups -- was ruled out in round 3)
    const int ks = zy % p.splitk, bz = zy / p.splitk;
    const int z0 = bz / p.batch_inner, z1 = bz % p.batch_inner;
    const u16* A = (const u16*)p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const u16* B = (const u16*)p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int64_t zoff = z0 * p.c_bs0 + z1 * p.c_bs1;
    const u16* ZP = (const u16*)a3t_zero_page;

    const int ktiles = (p.K + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;

    const bool WG = (CONV == 2) && (LAYOUT == L_TN) && (TAPS > 1);
    const int wg_cin = WG ? p.N / TAPS : 1;
    const int wg_tap = WG ? (tn * BN) / wg_cin : 0;
    const int kshift = WG ? (wg_tap - p.pad) * p.dil : p.kshift;

    // ---- per-lane source bookkeeping --------------------------------------------------------------
    // k-contiguous image: DMA instruction g covers tile rows g*8 + (lane>>3); 16-B chunk position
    //   c = lane&7 holds source chunk c ^ ((row>>1)&7): ds_read_b128 is serviced in the lane groups
    //   {0-3,12-15,20-27}/{4-11,16-19,28-31}, two 128-B rows share a 256-B bank line, so rows r and r+8
    //   of one group must land on different 16-B slots (measured: (row&7) left a 2-way conflict).
    // row-contiguous image: instruction g covers k-rows g*KPI + lane/LPR, chunk position c = lane%LPR
    //   holds source chunk c ^ ((krow&3)<<2) (the 4 k-rows of a ds_read_b64_tr_b16 hit distinct banks).
    const u16* a_row[NA];
    int a_tp[NA];          // A_KC conv: position in utterance
    bool a_ok[NA];
    int a_sw[NA];          // k-contiguous: swizzled source chunk offset (elements); row-contiguous: k-row
    int a_tap[NA], a_cc[NA];
    const u16* b_row[NB];
    bool b_ok[NB];
    int b_sw[NB];
    int b_tap[NB], b_cc[NB];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        a_tap[q] = 0, a_cc[q] = 0, a_tp[q] = 0;
        const int g = w * NA + q;
        if (A_KC) {
            int r = g * 8 + (lane >> 3);
            int m = tm * BM + r;
            a_ok[q] = m < p.M;
            a_tp[q] = (TAPS > 1) ? (m % p.Tseq) : 0;
            a_row[q] = A + (int64_t)m * p.a_rs;
            a_sw[q] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
            if (TAPS > 1) {
                int kg = kt0 * BK + a_sw[q];
                a_tap[q] = kg / p.Kc;
                a_cc[q] = kg - a_tap[q] * p.Kc;
            }
        } else {
            int kr = g * A_KPI + lane / A_LPR;
            int col = tm * BM + (((lane % A_LPR) ^ ((kr & 3) << 2)) * 8);
            a_ok[q] = col < p.M;
            a_row[q] = A + col;
            a_sw[q] = kr;
        }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        b_tap[q] = 0, b_cc[q] = 0;
        const int g = w * NB + q;
        if (B_KC) {
            int r = g * 8 + (lane >> 3);
            int n = tn * BN + r;
            b_ok[q] = n < p.N;
            b_row[q] = B + (int64_t)n * p.b_rs;
            b_sw[q] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        } else {
            int kr, col;
            if (B_LPR == 16) {   // 256-B k-rows, 4 per DMA instruction: XOR swizzle
                kr = g * 4 + lane / 16;
                col = tn * BN + (((lane % 16) ^ ((kr & 3) << 2)) * 8);
            } else {             // 384-B k-rows (natural 128-B stagger): rotate k-rows 2,3 (mod 4) by 64 B
                const int o16 = g * 64 + lane;          // 16-byte slot of this lane in the lane-linear image
                kr = o16 / B_LPR;
                int csrc = o16 % B_LPR - 4 * ((kr >> 1) & 1);
                csrc = csrc < 0 ? csrc + B_LPR : csrc;
                col = tn * BN + csrc * 8;
            }
            b_ok[q] = col < p.N;
            b_row[q] = B + col;
            b_sw[q] = kr;
            // running decomposition of this lane's reduction index: (tap, channel) for W^T of a conv,
            // position inside the utterance for the token-shifted weight-gradient operand
            int kg = kt0 * BK + kr;
            if (WG) {
                // conv weight gradient in ONE launch: output columns are (tap, c); a 128-column tile
                // lies inside one tap (Cin % 128 == 0), which fixes this block's token shift
                b_row[q] = B + (col - wg_tap * wg_cin);
                b_cc[q] = kg % p.Tseq;
            } else if (TAPS > 1) {
                b_tap[q] = kg / p.Kc;
                b_cc[q] = kg - b_tap[q] * p.Kc;
            } else if (KSM) {
                b_cc[q] = kg % p.Tseq;
            }
        }
    }

    // Uniform (scalar) decomposition of the tile's first k into (tap, channel): when the channel count is
    // a multiple of BK every lane of a tile works on the same tap, so the im2col row shift, its 64-bit
    // row offset and the validity test reduce to a few VALU ops and one v_cndmask per DMA (no branches).
    if (CONV == 1) {   // fold the per-lane constants into the row pointers (frees 8 VGPRs: the NN variant then fits 128)
#pragma unroll
        for (int q = 0; q < NA; ++q) a_row[q] += a_sw[q];
        if (!B_KC) {
#pragma unroll
            for (int q = 0; q < NB; ++q) b_row[q] += (int64_t)b_sw[q] * p.b_cs;
        }
    }
    const bool a_fast = (CONV == 1) ? A_KC : (A_KC && TAPS > 1 && (p.Kc % BK == 0));
    const bool b_fast = (CONV == 1) ? !B_KC : (!B_KC && !WG && TAPS > 1 && (p.Kc % BK == 0));
    const bool ks_fast = p.Tseq >= BK;
    // CONV == 1 walks the K-tiles TAPS INNERMOST: K-tile kt = (channel block kt / TAPS, tap kt % TAPS).  The taps of one channel
    // block read the same rows shifted by a token -- L2 / L1 hits; tap-major order fetched the whole operand from HBM once per
    // tap (profiles/r03_hbm_traffic_per_kernel.json before the change: 318 MB per launch of the first FFN conv against 140 MB
    // algorithmic).  The weights ([n][tap][c]) are addressed to match; the sum is the same set of products in another order.
    // (CONV == 2 takes the same order whenever its operands run on the uniform (tap, channel) tracking: every k-contiguous conv
    //  with whole channel blocks sums in ONE order on all three kernel families -- bit-identical results, tests/test_gpu_*.py)
    const bool TAP_INNER = (CONV == 1) || (CONV == 2 && !WG && TAPS > 1 && (a_fast || b_fast));
    int u_tap = 0, u_cc = 0;
    if (a_fast || b_fast) {
        if (TAP_INNER) {
            u_tap = kt0 % TAPS;
            u_cc = (kt0 / TAPS) * BK;
        } else {
            u_tap = (kt0 * BK) / p.Kc;
            u_cc = kt0 * BK - u_tap * p.Kc;
        }
    }
    auto issue = [&](int k0, int stage) {   // NOTE: called for consecutive tiles only (running state above)
        unsigned char* sA = smem + stage * STAGE_BYTES;
        unsigned char* sB = sA + A_BYTES;
        if (A_KC && a_fast) {
            const int off = (u_tap - p.pad) * p.dil;
            const int64_t roff = (int64_t)off * p.a_rs + u_cc;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const bool ok = a_ok[q] && ((unsigned)(a_tp[q] + off) < (unsigned)p.Tseq);
                const u16* src = ok ? a_row[q] + roff + (CONV == 1 ? 0 : a_sw[q]) : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(src), LDS_AS(sA + (w * NA + q) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const u16* src = ZP;
                const int kg = k0 + a_sw[q];
                if (A_KC) {
                    if (TAPS > 1) {
                        if (a_ok[q] && kg < p.K) {
                            int off = (a_tap[q] - p.pad) * p.dil, tt = a_tp[q] + off;
                            if (tt >= 0 && tt < p.Tseq) src = a_row[q] + (int64_t)off * p.a_rs + a_cc[q];
                        }
                        a_cc[q] += BK;
                        while (a_cc[q] >= p.Kc) a_cc[q] -= p.Kc, ++a_tap[q];
                    } else {
                        src = (a_ok[q] && kg < p.K) ? a_row[q] + kg : ZP;
                    }
                } else {
                    src = (a_ok[q] && kg < p.K) ? a_row[q] + (int64_t)kg * p.a_cs : ZP;
                }
                __builtin_amdgcn_global_load_lds(GLB_AS(src), LDS_AS(sA + (w * NA + q) * 1024), 16, 0, 0);
            }
        }
        if (B_KC) {
            const bool split = (CONV == 2) && (TAPS > 1 && p.b_ts != p.Kc);   // (weights are [n][tap][c]: b_ts == Kc, no split)
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int kg = (TAP_INNER && TAPS > 1 ? u_tap * p.Kc + u_cc : k0) + b_sw[q];
                int64_t koff = kg;
                if (split) {
                    int tap = kg / p.Kc, cc = kg - tap * p.Kc;
                    koff = (int64_t)tap * p.b_ts + cc;
                }
                const u16* src = (b_ok[q] && kg < p.K) ? b_row[q] + koff : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(src), LDS_AS(sB + (w * NB + q) * 1024), 16, 0, 0);
            }
        } else if (b_fast) {
            const int64_t toff = (int64_t)u_tap * p.b_ts + (int64_t)u_cc * p.b_cs;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const u16* src = b_ok[q] ? b_row[q] + toff + (CONV == 1 ? (int64_t)0 : (int64_t)b_sw[q] * p.b_cs) : ZP;
                __builtin_amdgcn_global_load_lds(GLB_AS(src), LDS_AS(sB + (w * NB + q) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const u16* src = ZP;
                const int kg = k0 + b_sw[q];
                if (!WG && TAPS > 1) {
                    if (b_ok[q] && kg < p.K) src = b_row[q] + (int64_t)b_tap[q] * p.b_ts + (int64_t)b_cc[q] * p.b_cs;
                    b_cc[q] += BK;
                    while (b_cc[q] >= p.Kc) b_cc[q] -= p.Kc, ++b_tap[q];
                } else if (WG || KSM) {
                    const bool ok = b_ok[q] && kg < p.K && ((unsigned)(b_cc[q] + kshift) < (unsigned)p.Tseq);
                    src = ok ? b_row[q] + (int64_t)(kg + kshift) * p.b_cs : ZP;
                    b_cc[q] += BK;
                    if (ks_fast)
                        b_cc[q] = (b_cc[q] >= p.Tseq) ? b_cc[q] - p.Tseq : b_cc[q];
                    else
                        b_cc[q] %= p.Tseq;
                } else {
                    src = (b_ok[q] && kg < p.K) ? b_row[q] + (int64_t)kg * p.b_cs : ZP;
                }
                __builtin_amdgcn_global_load_lds(GLB_AS(src), LDS_AS(sB + (w * NB + q) * 1024), 16, 0, 0);
            }
        }
        if (a_fast || b_fast) {
            if (TAP_INNER) {
                if (++u_tap == TAPS) u_tap = 0, u_cc += BK;
            } else {
                u_cc += BK;
                if (u_cc >= p.Kc) u_cc -= p.Kc, ++u_tap;
            }
        }
    };

    f32x16 acc[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (w >> 1) * 64, wn = (w & 1) * (32 * WN), lr = lane & 31, lk = lane >> 5;
    auto frag_kc = [&](const unsigned char* img, int row, int kk) -> bf16x8 {
        int kc = kk * 2 + lk;
        return *(const bf16x8*)(img + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
    };
    auto frag_rc = [&](const unsigned char* img, auto rbc, int row0, int kk) -> bf16x8 {
        // 16-lane group g: rows row0 + (g&1)*16 .. +15, k = kk*16 + (g>>1)*8 .. +7 (two 4-k transposed reads)
        constexpr int rowbytes = decltype(rbc)::value;
        const int g = lane >> 4, pp = lane & 15;
        const int col = row0 + (g & 1) * 16 + (pp & 3) * 4;
        const int kb = kk * 16 + (g >> 1) * 8 + (pp >> 2);
        const int k1 = kb + 4;
        const unsigned char *a0, *a1;
        if (rowbytes == 256) {
            a0 = img + kb * rowbytes + ((((col >> 3) ^ ((kb & 3) << 2))) << 4) + (col & 7) * 2;
            a1 = img + k1 * rowbytes + ((((col >> 3) ^ ((k1 & 3) << 2))) << 4) + (col & 7) * 2;
        } else {   // 384-byte k-rows: k-rows 2,3 (mod 4) are rotated by 64 B (kb and kb + 4 get the same rotation)
            int p0 = (col >> 3) + 4 * ((kb >> 1) & 1);
            p0 = p0 >= 24 ? p0 - 24 : p0;
            a0 = img + kb * rowbytes + (p0 << 4) + (col & 7) * 2;
            a1 = a0 + 4 * rowbytes;
        }
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a1));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    const short afloor = p.a_signmask ? (short)0 : (short)-32768;      // (a3t_gemm_desc::a_signmask; plain TN launches only)
    constexpr std::integral_constant<int, BM * 2> RB_A{};
    constexpr std::integral_constant<int, BN * 2> RB_B{};

    if (STAGES == 2) issue(kt0 * BK, 0);
    int stage = 0;
#ifdef GLDS_TIMING
    unsigned long long gt_war = 0, gt_iss = 0, gt_dma = 0, gt_bar = 0, gt_mma = 0, gt_start = GT_NOW(), gt_t = gt_start;
#define GT_LAP(acc) do { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = GT_NOW(); acc += n_ - gt_t; gt_t = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
    for (int kt = kt0; kt < kt1; ++kt) {
#ifdef GLDS_TIMING
        if (STAGES == 1) {
            GT_LAP(gt_mma);                   // ds_reads + MFMA issue of the previous tile
            if (kt > kt0) asm volatile("s_barrier" ::: "memory");
            GT_LAP(gt_war);
            issue(kt * BK, 0);
            GT_LAP(gt_iss);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            GT_LAP(gt_dma);
            asm volatile("s_barrier" ::: "memory");
            GT_LAP(gt_bar);
        } else {
            __syncthreads();
        }
#else
        if (STAGES == 1) {
            if (kt > kt0) __syncthreads();   // WAR: every wave is done reading the previous tile
            issue(kt * BK, 0);
        }
        __syncthreads();  // drains this tile's DMA (vmcnt(0) precedes the barrier) + WAR on the other stage
#endif
        if (STAGES == 2 && kt + 1 < kt1) issue((kt + 1) * BK, stage ^ 1);
        const unsigned char* sA = smem + stage * STAGE_BYTES;
        const unsigned char* sB = sA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 a0, a1, bq[WN];
            if (A_KC) {
                a0 = frag_kc(sA, wm + lr, kk);
                a1 = frag_kc(sA, wm + 32 + lr, kk);
            } else {
                a0 = frag_rc(sA, RB_A, wm, kk);
                a1 = frag_rc(sA, RB_A, wm + 32, kk);
                if (LAYOUT == L_TN && CONV == 0) a0 = a3t_sign_floor(a0, afloor), a1 = a3t_sign_floor(a1, afloor);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) bq[j] = B_KC ? frag_kc(sB, wn + 32 * j + lr, kk) : frag_rc(sB, RB_B, wn + 32 * j, kk);
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[j], acc[0][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[j], acc[1][j], 0, 0, 0);
        }
        if (STAGES == 2) stage ^= 1;
    }
#ifdef GLDS_TIMING
    if (STAGES == 1) {
        GT_LAP(gt_mma);
        if (tid == 0) {
            unsigned long long* d = glds_dbg + (blockIdx.x % 16384) * 8;
            d[0] = gt_war, d[1] = gt_iss, d[2] = gt_dma, d[3] = gt_bar, d[4] = gt_mma, d[5] = gt_start, d[6] = gt_t, d[7] = kt1 - kt0;
        }
    }
#endif
    // (WN = 3 is dispatched with the vector epilogue only: host contract)
    if (WN == 2 && (!p.epi_vec || p.accumulate == A3T_ACC_ATOMIC)) {   // coalesced 128-B atomic rows straight from the accumulators
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = tm * BM + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    int col = tn * BN + wn + j * 32 + lr;
                    epilogue_store(p, zoff, row, col, acc[i][j][r], ks);
                }
        return;
    }
    // Stage each wave's fp32 sub-tile through its own slice of the (now idle) LDS in panels of 64 (or 32) columns,
    // RP rows per pass, so that every lane finishes 4 consecutive columns: bias / residual / mask reads and the
    // output stores become 16-byte (8-byte for bf16) accesses, 256 B (128 B) contiguous per output row segment.
    constexpr int LDS_PER_WAVE = STAGES * STAGE_BYTES / NW;            // 8/16 KiB (BN = 128), 10/20 KiB (BN = 192)
    __syncthreads();
    float* ct = (float*)(smem + w * LDS_PER_WAVE);
    // (only this wave reads its region back; a wave is lock-step, LDS ops are issued in order)
    auto panel = [&](auto njc, auto j0c) {
        constexpr int NJ = decltype(njc)::value, J0 = decltype(j0c)::value;
        constexpr int PC = 32 * NJ, LR = PC / 4, RPI = 64 / LR;        // panel columns, lanes per row, rows per sweep
        constexpr int RP = LDS_PER_WAVE / (PC * 4) >= 64 ? 64 : (LDS_PER_WAVE / (PC * 4) >= 32 ? 32 : 16);   // rows per pass
        constexpr int NPASS = 64 / RP;
        const int c4 = (lane % LR) * 4, r4 = lane / LR;
        const int col = tn * BN + wn + J0 * 32 + c4;
        const bool col_ok = col < p.N;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && ks == 0 && col_ok) bias4 = *(const float4*)(p.bias + col);
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            // rows [ps*RP, ps*RP+RP) of the wave tile: MFMA block i = row/32, registers with (r>>2) in the pass
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lrow = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;   // row inside the wave tile
                        // (the row range test only depends on i and r>>2: resolved at compile time per register)
                        if ((i * 32 + 8 * (r >> 2)) / RP == ps) ct[(lrow - ps * RP) * PC + j * 32 + lr] = acc[i][J0 + j][r];
                    }
#pragma unroll 4
            for (int it = 0; it < RP / RPI; ++it) {
                const int srow = it * RPI + r4;
                const int row = tm * BM + wm + ps * RP + srow;
                if (row >= p.M || !col_ok) continue;
                float4 v = *(const float4*)(ct + srow * PC + c4);
                const int64_t idx = zoff + (int64_t)row * p.c_rs + col;
                epilogue_vec4(p, v, idx, bias4, ks, cs);
            }
        }
        if (p.colsum) colsum_flush<LR>(p, cs, lane, col_ok, z1, col, tm + z0);
    };
    panel(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    if (WN == 3) panel(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
}

static inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }
static inline bool m8(int64_t v) { return (v % 8) == 0; }

int a3t_gemm_bf16_8p(const GP& p, int batch, int ly, hipStream_t stream);   // gemm_bf16_8p.hip
int a3t_gemm_bf16_pn(const GP& p, int batch, int ly, hipStream_t stream);   // gemm_bf16_pn.hip
int a3t_gemm_bf16_tt(const GP& p, int batch, int ly, hipStream_t stream);   // gemm_bf16_tt.hip

template <int LY, int ST, int WM, int CV, int WN = 2>
static void launch_variant(const GP& pv, dim3 grid, hipStream_t stream) {
    constexpr int lds = ST * (64 * WM + 64 * WN) * 64 * 2;
    // per device and cheap: a process that drives a second GPU must not launch there without the raised limit
    (void)hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<LY, ST, WM, CV, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((gemm_bf16_glds_kernel<LY, ST, WM, CV, WN>), grid, dim3(128 * WM), lds, stream, pv);
}

// returns -1 when the descriptor does not meet the alignment contract of this kernel
int a3t_gemm_bf16_glds(const GP& p, int batch, bool AK, bool BKC, hipStream_t stream) {
    const bool aview = (p.a_unaligned & 1) != 0;     // A: a 2-byte aligned strided view (the loaders only do pointer arithmetic)
    bool ok = (aview || al16(p.A)) && al16(p.B) && m8(p.a_bs0) && m8(p.a_bs1) && m8(p.b_bs0) && m8(p.b_bs1);
    if (AK || BKC) ok = ok && m8(p.Kc);      // (16-byte granules run along k only for k-contiguous operands)
    if (AK)
        ok = ok && (aview || m8(p.a_rs)) && m8(p.K);
    else
        ok = ok && (aview || m8(p.a_cs)) && m8(p.M);
    if (BKC)
        ok = ok && m8(p.b_rs) && m8(p.K) && m8(p.b_ts);
    else
        ok = ok && m8(p.b_cs) && m8(p.b_ts) && m8(p.N);
    if (!AK && p.taps > 1) ok = ok && ((p.N / p.taps) % 128 == 0) && (p.N % p.taps == 0) && p.Tseq > 0;
    if (!ok) return -1;
    GP pv = p;
    if (p.colsum && (p.accumulate == A3T_ACC_ATOMIC)) return -1;
    // vector epilogue contract: 4-column groups never straddle N and every C/R/S/bias access is aligned
    pv.epi_vec = (p.N % 4 == 0) && (p.c_rs % 4 == 0) && (p.c_bs0 % 4 == 0) && (p.c_bs1 % 4 == 0) &&
                 al16(p.C) && (!p.R || al16(p.R)) && (!p.S || ((uintptr_t)p.S & 7) == 0) && (!p.bias || al16(p.bias));
    if (p.colsum && !pv.epi_vec) return -1;

    const bool keep_rm = p.keep_layout == 1 && (p.keep_in || p.keep_out);      // row-major nibble image: vector epilogue / panel kernel
    if (keep_rm && (!pv.epi_vec || p.c_rs != p.N || batch != 1 || p.splitk != 1 || p.accumulate != A3T_ACC_STORE || p.N % 8 ||
                    (p.keep_out && p.R) || (p.keep_in && p.S)))
        return A3T_EINVAL;
    if (p.a_signmask && (AK || BKC || p.taps > 1 || p.kshift_mode)) return -1;   // (fragment-register pass of the m-contiguous A only)
    if (!(pv.keep_out || (pv.keep_in && !keep_rm) || p.a_signmask || p.A2 || aview)) {   // N = 384 outputs: one 160-row panel x all columns per workgroup (-1: does not qualify)
        const int rc = a3t_gemm_bf16_pn(pv, batch, (AK && BKC) ? L_NT : (AK ? L_NN : L_TN), stream);
        if (rc != -1) return rc;
    }
    if (!p.a_signmask && !keep_rm && !p.A2 && !aview) {   // many-tile k-contiguous GEMMs: persistent 256x256 8-phase kernel (returns -1 when the problem does not qualify)
        const int rc = a3t_gemm_bf16_8p(pv, batch, (AK && BKC) ? L_NT : (AK ? L_NN : L_TN), stream);
        if (rc != -1) return rc;
    }
    if (!BKC && !(pv.keep_in || pv.keep_out) && !aview) {   // score-sized A operand x [k][n] slices (attention backward): one streaming workgroup per CU
        const int rc = a3t_gemm_bf16_tt(pv, batch, AK ? L_NN : L_TN, stream);
        if (rc != -1) return rc;
    }
    if (p.A2) return A3T_EINVAL;        // two products in one launch: the streaming kernel only (ask a3t_gemm_tt_supported first)
    static int forced_st = -1;
    if (forced_st < 0) {
        const char* e = getenv("A3T_GEMM_STAGES");
        forced_st = e ? (e[0] == '1' ? 1 : 2) : 0;
    }
    // Variant choice (measured on MI355X, tools/gemm_bench*.py): single LDS buffer + 3-4 co-resident workgroups per CU
    // when the grid is large, double buffering inside the workgroup when it is small.  (A 256-row tile, WM = 4, moves
    // 25 % fewer DMA bytes per flop but halves the co-resident workgroups; it lost 5-40 % on every shape but one and
    // that one is now faster on the 4-per-CU 128-row variant, so only WM = 2 is instantiated.)
    constexpr int wm = 2;
    const int ly = (AK && BKC) ? L_NT : (AK ? L_NN : L_TN);
    int conv = 0;
    if (p.taps > 1 || p.kshift_mode)
        conv = (p.taps > 1 && ly != L_TN && p.Kc % 64 == 0 && p.K % 64 == 0 && (ly != L_NT || p.b_ts == p.Kc)) ? 1 : 2;
    // 192-column tiles when they waste clearly fewer padded columns than 128-column tiles (N = 192: 0 vs 33 %)
    static int wn_mode = -1;
    if (wn_mode < 0) {
        const char* e = getenv("A3T_GEMM_WN3");
        wn_mode = e ? (e[0] == '0' ? 0 : 2) : 1;
    }
    const long pad128 = (long)((p.N + 127) / 128) * 128, pad192 = (long)((p.N + 191) / 192) * 192;
    const bool wn3 = conv == 0 && pv.epi_vec && p.accumulate != A3T_ACC_ATOMIC && wn_mode != 0 && (wn_mode == 2 || pad192 * 10 <= pad128 * 8);
    pv.tiles_n = wn3 ? (p.N + 191) / 192 : (p.N + 127) / 128;
    const int tiles_m = (p.M + 64 * wm - 1) / (64 * wm);
    const long tiles = (long)pv.tiles_n * tiles_m * batch * p.splitk;
    const int stages = forced_st ? forced_st : (tiles >= (wn3 ? 512 : 768) ? 1 : 2);
    pv.ntiles = pv.tiles_n * tiles_m;
    dim3 grid((unsigned)((long)pv.ntiles * batch * p.splitk));
    if (wn3) {
#define V3(LY, ST)                                             \
    if (ly == LY && stages == ST) {                            \
        launch_variant<LY, ST, 2, 0, 3>(pv, grid, stream);     \
        a3t_note_kernel("gemm_bf16_glds_kernel<%d, %d, 2, 0, 3>", LY, ST); \
        return (int)hipGetLastError();                         \
    }
        V3(L_NT, 1) V3(L_NT, 2) V3(L_NN, 1) V3(L_NN, 2) V3(L_TN, 1) V3(L_TN, 2)
#undef V3
    }
#define V(LY, ST, WM_, CV)                                   \
    if (ly == LY && stages == ST && wm == WM_ && conv == CV) { \
        launch_variant<LY, ST, WM_, CV>(pv, grid, stream);     \
        a3t_note_kernel("gemm_bf16_glds_kernel<%d, %d, %d, %d, 2>", LY, ST, WM_, CV); \
        return (int)hipGetLastError();                         \
    }
    V(L_NT, 1, 2, 0) V(L_NT, 2, 2, 0) V(L_NT, 1, 2, 1) V(L_NT, 2, 2, 1) V(L_NT, 1, 2, 2) V(L_NT, 2, 2, 2)
    V(L_NN, 1, 2, 0) V(L_NN, 2, 2, 0) V(L_NN, 1, 2, 1) V(L_NN, 2, 2, 1) V(L_NN, 1, 2, 2) V(L_NN, 2, 2, 2)
    V(L_TN, 1, 2, 0) V(L_TN, 2, 2, 0) V(L_TN, 1, 2, 2) V(L_TN, 2, 2, 2)
#undef V
    return A3T_EINVAL;
}
