// gemm_bf16_pn.hip -- "row panel" bf16 MFMA GEMM for gfx950: the GEMMs of the model whose output width is a multiple of
// d_model = 384 (384: second FFN conv, linear_out, pointwise_conv2 and the data gradients into the residual stream; 768 /
// 1152 / 1536: pointwise_conv1, the q/k/v projection, the first FFN conv and the data gradient of the second).
//
// C[M][N] = epilogue( A[M][K] . B[N][K]^T ), both operands k-contiguous bf16 (Linear, Conv1d over time as implicit im2col,
// and -- through transposed weight shadows -- their data gradients), fp32 accumulate; N in chunks of 384 columns.
//
// Why a kernel of its own: a 256x256 tiling covers N = 384 with 1.5 column tiles and 128x128 tiles leave the K loop at
// ~730 TFLOP/s.  One workgroup = a 160-row panel x ALL 384 columns: B (the weights, <= 3.5 MB) is the same for every
// workgroup and lives in L2, A is read exactly once, and the benchmark's M = 35840 tokens are 224 panels -- ONE round of a
// 256-CU chip with no tail (calibrated stand-alone in tools/probes/gemm_pn.hip: 35840x384x4608 in 115 us = 1.10 PFLOP/s,
// K = 1152 in 30 us; the 128x128 kernel: 165 / 60 us).
//
// Structure (the discipline of the 8-phase kernel, gemm_bf16_8p.hip, on a different tile):
//   * 8 waves = 2 (wr) x 4 (wc); wave tile 80 x 96 = 5 x 6 blocks of v_mfma_f32_16x16x32_bf16 (120 accumulator registers);
//   * LDS = 2 K-tile buffers x {A: 160 rows, B lo: 192 rows, B hi: 192 rows} x 64 k (68 KiB per buffer);
//   * a K-tile is two phases of 30 MFMA (round 6; four of 15 before): (k-steps 0 and 1, B lo) (k-steps 0 and 1, B hi); the A
//     fragments of both k-steps stay in registers for the whole K-tile;
//   * phase = [ds_read_b128 fragments | DMA issue | counted vmcnt | lgkmcnt(0)] s_barrier [setprio 1 | 30 MFMA | setprio 0]
//     s_barrier; waves 4-7 run one barrier behind waves 0-3 (each SIMD holds one wave of each group: their MFMA segments
//     alternate).  The fragment reads are waited for IN FRONT of the barrier, so every read issued before a barrier has retired
//     when it releases and a DMA issued behind it may overwrite what was read (see the K loop);
//   * DMA (buffer_load ... lds, 1 KiB per wave instruction): B hi of K-tile t+1 in phase A, A and B lo of K-tile t+2 in phase B
//     -- one phase behind the last read of the region they overwrite, two to three phases ahead of their first read; 9
//     instructions per wave and K-tile (the A image is 2.5 per wave: waves 4-7 repeat the third instruction of waves 0-3 -- same
//     bytes to the same address -- so that every wave counts alike), two counted waits (vmcnt(9)) per K-tile;
//   * conv padding, utterance boundaries, the M tail and the K-tiles past the end read zeros through the buffer descriptor's
//     range check (voffset = 0x80000000): the steady state is branch-free;
//   * B rows are permuted in the LDS image so that lane group g of wave wc owns the 8 consecutive columns
//     wc*96 + jl*32 + g*8 .. +7 of pair jl: 16-byte bf16 stores, 64 contiguous bytes per row and instruction;
//   * a workgroup owns a panel and walks its 384-column chunks (N = 1536: four; the panel's rows stay in L2); the epilogue is a
//     plain tail (bias -> relu -> ReLU' mask S -> dropout -> alpha -> fp32 residual -> store bf16 | fp32, column sums through
//     LDS) that runs while the next tile's first K-tiles are already in flight.
// Tried on top of this kernel and taken out again (round 3, profiles/r03_pn_check.txt):
//   * batched operands for the attention scores (64 x 1120 x 1120 x 192 = 21 tiles of 3 K-tiles per batch element): 80 us against
//     73 us on the 128x128 kernel -- ~9 us of fill and epilogue per tile against 4.7 us of K loop;
//   * a 320 x 192 geometry (4 x 2 waves, same wave tile) for probabilities x V / dS x K / dBD x P with a transposed second
//     operand: 54 us against 55 us -- those products read a T x T matrix (160 MB per launch) and sit on the HBM roof already.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/a3t_hip.h"
#include "gemm_common.h"

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

namespace {
constexpr int PN_ROWS = 160, PN_COLS = 384;
constexpr int A_BYTES = PN_ROWS * 128, BH_BYTES = 192 * 128, BUF_BYTES = A_BYTES + 2 * BH_BYTES;   // 20 + 24 + 24 KiB
constexpr int LDS_CSUM = 2 * BUF_BYTES;        // 384 fp32 column sums of the tile in flight
constexpr int LDS_TOTAL = LDS_CSUM + PN_COLS * 4;                                                   // 140 800 B
constexpr unsigned OOB = 0x80000000u;         // voffset beyond every descriptor (host contract: operands < 2 GiB)

__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row, result in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}
}   // namespace

template <bool CONV>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pn_kernel(GP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, lane_ = lane;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), w_ = w;
    const int wr = w >> 2, wc = w & 3;
    const int nk = p.K >> 6;             // (an odd count runs one K-tile of zeros: the loop works on pairs)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
    typedef const __attribute__((address_space(4))) GP* kargp;
#define EPI_ARGS(q) kargp q = (kargp)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q))

    // ---- DMA lane geometry: a wave instruction fills 8 LDS rows of 128 B (lane -> row + (lane>>3), chunk position lane&7,
    // which holds SOURCE chunk (lane&7) ^ (row&7)).  A: instruction q covers panel rows (q*8 + w)*8 .. +7 (q = 2: waves 4-7
    // repeat waves 0-3).  B half hb: instruction q covers image rows R = q*64 + w*8 .. +7 = (wave column R/48, block jl,
    // row rho): output column  n = chunk*384 + (R/48)*96 + jl*32 + (rho>>2)*8 + hb*4 + (rho&3).
    const int srow = lane >> 3;
    const unsigned schunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
    const unsigned a_rsb = (unsigned)p.a_rs * 2u, b_rsb = (unsigned)p.b_rs * 2u;
    const int Tq = CONV ? p.Tseq : 1;
    const unsigned tap_magic = CONV ? (1u << 20) / (unsigned)p.taps + 1u : 0u;      // kt / taps = (kt * magic) >> 20 for kt < 2^10

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

    // ---- epilogue of tile (panel `tile`, column chunk `chunk`): lane (fr, g) owns rows wr*80 + i*16 + fr (i = 0..4) x columns
    // chunk*384 + wc*96 + jl*32 + g*8 .. +7 (jl = 0..2); leaves the accumulators cleared
    auto epi = [&](const int tile, const int chunk) __attribute__((always_inline)) {
        EPI_ARGS(q);
        int lane = lane_, w = w_;
        asm volatile("" : "+v"(lane), "+s"(w));
        const int fr = lane & 15, g = lane >> 4, wr = w >> 2, wc = w & 3;
        float* cs_l = (float*)(smem + LDS_CSUM);
        if (q->colsum) {
            if (tid < PN_COLS) cs_l[tid] = 0.f;
            __syncthreads();
        }
#pragma unroll
        for (int jl = 0; jl < 3; ++jl) {
            const int nloc = wc * 96 + jl * 32 + g * 8, ncol = chunk * PN_COLS + nloc;
            const bool nok = ncol < q->N;
            float b8[8], cs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] = 0.f, cs[e] = 0.f;
            if (q->bias && nok) {
                const float4 b0 = *(const float4*)(q->bias + ncol), b1 = *(const float4*)(q->bias + ncol + 4);
                b8[0] = b0.x, b8[1] = b0.y, b8[2] = b0.z, b8[3] = b0.w, b8[4] = b1.x, b8[5] = b1.y, b8[6] = b1.z, b8[7] = b1.w;
            }
            // keep image: the five mask words of this column block are requested together, ahead of the stores below (which the
            // compiler has to assume they alias: one exposed load latency per block instead of one per row group)
            unsigned kbv[5] = {0u, 0u, 0u, 0u, 0u};
            if (q->keep_in) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int m = tile * PN_ROWS + wr * 80 + i * 16 + fr;
                    if (nok && m < q->M) kbv[i] = *(const unsigned short*)(q->keep_in + (((int64_t)m * q->c_rs + ncol) >> 2));
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int m = tile * PN_ROWS + wr * 80 + i * 16 + fr;
                const bool ok = nok && (m < q->M);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[i][(e >> 2) * 3 + jl][e & 3] + b8[e];
                acc[i][jl] = f32x4{0.f, 0.f, 0.f, 0.f}, acc[i][3 + jl] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (q->act == A3T_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const int64_t idx = (int64_t)m * q->c_rs + ncol;
                if (q->S) {        // ReLU' mask: keep where the saved activation is positive
                    float sv[8];
                    if (!ok) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sv[e] = 0.f;
                    } else if (q->s_dtype == A3T_BF16) {
                        const uint4 t = *(const uint4*)((const u16*)q->S + idx);
                        sv[0] = bf2f(t.x & 0xffff), sv[1] = bf2f(t.x >> 16), sv[2] = bf2f(t.y & 0xffff), sv[3] = bf2f(t.y >> 16);
                        sv[4] = bf2f(t.z & 0xffff), sv[5] = bf2f(t.z >> 16), sv[6] = bf2f(t.w & 0xffff), sv[7] = bf2f(t.w >> 16);
                    } else {
                        const float4 s0 = *(const float4*)(q->S + idx), s1 = *(const float4*)(q->S + idx + 4);
                        sv[0] = s0.x, sv[1] = s0.y, sv[2] = s0.z, sv[3] = s0.w, sv[4] = s1.x, sv[5] = s1.y, sv[6] = s1.z, sv[7] = s1.w;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = sv[e] > 0.f ? v[e] : 0.f;
                }
                if (q->keep_in) {  // the same mask as two nibbles of the row-major keep image (a3t_gemm_desc::keep_layout = 1)
                    const unsigned kb = kbv[i];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((kb >> ((e >> 2) * 8 + (e & 3))) & 1u) ? v[e] : 0.f;
                }
                if (q->drop_inv > 0.f) {
                    const unsigned t = q->drop_thr >> 16;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const unsigned hsh = rng_pair(q->drop_key, ((unsigned)idx + (unsigned)e) >> 1);
                        v[e] = ((hsh & 0xffffu) >= t) ? v[e] * q->drop_inv : 0.f;
                        v[e + 1] = ((hsh >> 16) >= t) ? v[e + 1] * q->drop_inv : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= q->alpha;
                if (ok) {
                    if (q->R) {
                        const float4 r0 = *(const float4*)(q->R + idx), r1 = *(const float4*)(q->R + idx + 4);
                        v[0] += r0.x, v[1] += r0.y, v[2] += r0.z, v[3] += r0.w, v[4] += r1.x, v[5] += r1.y, v[6] += r1.z, v[7] += r1.w;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[e] += v[e];
                    if (q->c_dtype == A3T_BF16) {
                        uint4 o;
                        o.x = io_pack2(v[0], v[1]), o.y = io_pack2(v[2], v[3]), o.z = io_pack2(v[4], v[5]), o.w = io_pack2(v[6], v[7]);
                        *(uint4*)((u16*)q->C + idx) = o;
                    } else {
                        float* c = (float*)q->C + idx;
                        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                        *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
            }
            if (q->colsum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] = row16_sum(cs[e]);
                if (fr == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) __hip_atomic_fetch_add(cs_l + nloc + e, cs[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (q->colsum) {
            __syncthreads();
            if (tid < PN_COLS && chunk * PN_COLS + tid < q->N) atomicAdd(q->colsum + chunk * PN_COLS + tid, q->colsum_scale * cs_l[tid]);
        }
    };

    // A workgroup owns a panel and walks its column chunks (the panel's rows stay in L2), then the next panel.  The epilogue
    // of a tile runs AFTER the first K-tiles of the next tile have been requested: it hides their flight.
    bool pend = false;
    int e_tile = 0, e_chunk = 0;
    for (int tile = blockIdx.x; tile * p.tiles_n < p.ntiles; tile += gridDim.x)
        for (int chunk = 0; chunk < p.tiles_n; ++chunk) {
            unsigned voffB[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int R = (q % 3) * 64 + w * 8 + srow, hb = q / 3;
                const int wcr = R / 48, r2 = R % 48, jl = r2 >> 4, rho = r2 & 15;
                const int n = chunk * PN_COLS + wcr * 96 + jl * 32 + (rho >> 2) * 8 + hb * 4 + (rho & 3);
                voffB[q] = n < p.N ? (unsigned)n * b_rsb + schunk16 : OOB;
            }
            // per-lane row state of this panel: byte offset of the lane's three A rows and their positions inside the utterance
            unsigned voffA[3];
            int tposA[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int m = tile * PN_ROWS + (q * 8 + (q == 2 ? (w & 3) : w)) * 8 + srow;
                voffA[q] = (unsigned)m * a_rsb + schunk16;
                tposA[q] = (m < p.M) ? (CONV ? m % p.Tseq : 0) : -(1 << 24);
            }
            int a_kt = 0, a_tap = 0, a_c0 = 0;       // cursor of the A loader (uniform)
            auto issueA = [&](const int buf) __attribute__((always_inline)) {
                const bool live = a_kt < nk;
                const int shift = CONV ? (a_tap - p.pad) * p.dil : 0;
                // (the range check looks at voffset alone: the row shift -- negative for the first taps -- lives there)
                const unsigned so = CONV ? (unsigned)a_c0 * 2u : (unsigned)a_kt * 128u;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    unsigned char* dst = smem + buf * BUF_BYTES + (q * 8 + (q == 2 ? (w & 3) : w)) * 1024;
                    const bool ok = live && ((unsigned)(tposA[q] + shift) < (unsigned)Tq);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_AS(dst), 16, ok ? voffA[q] + (unsigned)shift * a_rsb : OOB, so, 0, 0);
                }
                ++a_kt;
                if (CONV) {       // taps innermost: the K-tiles of one 64-channel block read the same panel rows shifted by a token --
                    ++a_tap;      // L2 (mostly L1) hits; tap-major order fetched the panel from HBM once per tap (384 vs 160 MB per launch)
                    if (a_tap == p.taps) a_tap = 0, a_c0 += 64;
                }
            };
            auto issueB = [&](const int hb, const int kt, const int buf) __attribute__((always_inline)) {
                const bool live = kt < nk;
                // k offset of K-tile kt in the weight rows [tap][channel]: (kt % taps) * Kc + (kt / taps) * 64
                const unsigned cb = CONV ? ((unsigned)kt * tap_magic) >> 20 : 0u;
                const unsigned kof = CONV ? ((unsigned)kt - cb * (unsigned)p.taps) * (unsigned)p.Kc + cb * 64u : (unsigned)kt * 64u;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    unsigned char* dst = smem + buf * BUF_BYTES + A_BYTES + hb * BH_BYTES + (q * 8 + w) * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LDS_AS(dst), 16, live ? voffB[hb * 3 + q] : OOB, kof * 2u, 0, 0);
                }
            };
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
                    for (int j = 0; j < 3; ++j)
                        acc[i][hb * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i][s], acc[i][hb * 3 + j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            };

            // ---- prologue: K-tile 0 complete, A and B lo of K-tile 1 (its B hi goes out in phase 2); the previous tile's
            // epilogue runs under their flight (its loads and stores are younger than the DMA: everything is waited for)
            issueA(0), issueB(0, 0, 0), issueB(1, 0, 0);
            issueA(1), issueB(0, 1, 1);
            if (pend) {
                epi(e_tile, e_chunk);
                WAIT_VM(0);
            } else {
                WAIT_VM(9);
            }
            BAR();
            if (wr == 1) BAR();    // second group runs one barrier behind

            // A K-tile = TWO phases of 30 MFMAs (round 6; four of 15 before): (both k-steps x B lo) (both k-steps x B hi).  The MFMA
            // segments of the two wave groups alternate on a SIMD with a hand-over of ~90 cycles at every barrier
            // (profiles/r06_pn_segments.txt): half the barriers, half the hand-overs.  What makes the longer phases safe: a wave
            // waits for its fragment reads BEFORE the phase's first barrier (lgkmcnt(0), free -- it would wait there for the other
            // group's MFMA segment anyway), so when a barrier releases, every fragment read that any wave issued in front of it
            // has retired, and a DMA issued behind it may overwrite what those reads read:
            //   phase A of K-tile kt: B hi of K-tile kt+1 -> the other buffer (last read in phase B of kt-1, by the trailing
            //   group in front of the barrier this group has just passed);  phase B: A and B lo of K-tile kt+2 -> this buffer
            //   (last read in phase A of kt).  Waits as before: vmcnt(9) in phase A retires B hi of THIS K-tile, in phase B the A and
            //   B lo of the next one; the barrier behind the wait makes it everyone's.
            auto ktile = [&](const int kt, const int b) __attribute__((always_inline)) {
                const unsigned char* cur = smem + b * BUF_BYTES;
                // phase A: k-steps 0, 1 x B lo
                readB(cur, 0, 0, fbx);
                readB(cur, 0, 1, fby);
                readA(cur, 0);
                readA(cur, 1);
                issueB(1, kt + 1, b ^ 1);
                WAIT_VM(9);
                WAIT_LGKM(0);
                BAR();
                mm(0, 0, fbx);
                mm(0, 1, fby);
                BAR();
                // phase B: k-steps 0, 1 x B hi
                readB(cur, 1, 0, fbx);
                readB(cur, 1, 1, fby);
                issueA(b), issueB(0, kt + 2, b);
                WAIT_VM(9);
                WAIT_LGKM(0);
                BAR();
                mm(1, 0, fbx);
                mm(1, 1, fby);
                BAR();
            };
            for (int kt = 0; kt < nk; kt += 2) {
                ktile(kt, 0);
                ktile(kt + 1, 1);
            }
            if (wr == 0) BAR();    // both groups level again: every fragment read of this tile has retired
            WAIT_VM(0);            // (the trailing zero-fill DMA must not outlive the tile)
            e_tile = tile, e_chunk = chunk, pend = true;
        }
    if (pend) epi(e_tile, e_chunk);
}

static int pn_cus() {          // per device (a process may drive several GPUs)
    static int n[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n[dev]) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
        n[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    return n[dev];
}

// mode: 0 never, 1 whenever legal, 2 cost model (default), 3 cost model for N = 384 only; A3T_GEMM_PN or a3t_gemm_pn_mode()
static int g_pn_mode = -1;
static int pn_mode() {
    if (g_pn_mode < 0) {
        const char* e = getenv("A3T_GEMM_PN");
        g_pn_mode = e ? atoi(e) : 2;
    }
    return g_pn_mode;
}
extern "C" int a3t_gemm_pn_mode(int mode) {
    const int old = pn_mode();
    g_pn_mode = mode;
    return old;
}

static bool pn_applicable(const GP& p, int batch, int ly) {
    const int mode = pn_mode();
    if (mode == 0 || ly != 0 || batch != 1 || p.splitk != 1 || p.accumulate != A3T_ACC_STORE) return false;
    if (p.N % 8 != 0 || p.K % 64 != 0 || p.c_rs % 8 != 0 || p.a_cs != 1 || p.b_cs != 1) return false;
    if (p.kshift_mode || p.keep_out || (p.keep_in && p.keep_layout != 1)) return false;      // (keep_in: the row-major nibble image only)
    if (p.S && ((uintptr_t)p.S & 15)) return false;
    if (p.R && ((uintptr_t)p.R & 15)) return false;
    if (p.colsum && p.colsum_slots > 1) return false;
    if (p.act != A3T_ACT_NONE && p.act != A3T_ACT_RELU) return false;
    if (p.taps > 1 && (p.Kc % 64 != 0 || p.b_ts != p.Kc || p.Tseq <= 0)) return false;
    if (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) return false;
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    const int64_t a_bytes = ((int64_t)p.M * p.a_rs) * 2, b_bytes = ((int64_t)p.N * p.b_rs) * 2;
    if (a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31) || (int64_t)p.M * p.c_rs >= (1ll << 32)) return false;
    if (mode == 3 && p.N != PN_COLS) return false;
    if (mode >= 2) {
        // Cost model fitted on MI355X (tools/probes/gemm_pn.hip, profiles/r03_pn_check.txt): a tile (panel x 384-column chunk) costs ~1.55 us per
        // 64-wide K-tile plus ~8 us of pipeline fill and epilogue (dropout hashes and an fp32 residual add ~3 more; the later
        // chunks of a panel overlap their fill with the previous epilogue: ~5), all panels of a round run together; the 128x128
        // kernel does these problems at ~700-780 TFLOP/s for long K and ~450 for K = 384.
        // Wider outputs (N = 768 / 1152 / 1536: pointwise_conv1, q/k/v, the first FFN conv) measure 5-12 % faster in a loop of
        // their own (profiles/r03_pn_check.txt) and NOT faster inside the training step (configs[1], same box: 48.7-48.9 ms per step with
        // them, 48.2-48.6 without, gpurun_out/pn_step_ab4.log) -- a loop of one 224-workgroup kernel fills the 32 idle CUs with the
        // next launch, a step does not.  The exception is the data gradient of the second FFN conv, whose ReLU' mask read makes the
        // 128x128 kernel's epilogue slow: -0.5 ms per step.  Hence: several chunks only for problems that carry a mask tensor.
        if (p.N % PN_COLS != 0 || (p.N > PN_COLS && !p.S && !p.keep_in)) return false;
        const long panels = (p.M + PN_ROWS - 1) / PN_ROWS, chunks = p.N / PN_COLS;
        const int cus = pn_cus();
        const double rounds = (double)((panels + cus - 1) / cus);
        double fixed = 8.0;
        if (p.drop_inv > 0.f) fixed += 1.5;
        if (p.R || p.c_dtype == A3T_F32 || p.S || p.keep_in) fixed += 1.5;
        const int nk2 = ((p.K / 64 + 1) / 2) * 2;
        const double tpn = rounds * (chunks * (nk2 * 1.55 + fixed) - (chunks - 1) * 3.0);
        const double t128 = 2.0 * p.M * p.N * (double)p.K / (p.K >= 1024 ? (p.N >= 1024 ? 780e6 : 700e6) : 450e6) + 6.0;     // us
        if (panels < cus / 2 || tpn > 0.9 * t128) return false;
    }
    return true;
}

// flags as for a3t_gemm_8p_supported: 1 bias / activation, 2 dropout, 16 fp32 output / residual, 32 column sums, 64 ReLU' mask
// tensor S, 8 keep_in as the row-major nibble image of a3t_gemm_desc::keep_layout = 1 (4: never)
extern "C" int a3t_gemm_pn_supported(int M, int N, int K, int taps, int flags) {
    static float dummy[4] __attribute__((aligned(16)));
    GP p = {};
    p.M = M, p.N = N, p.K = K, p.taps = taps < 1 ? 1 : taps, p.Kc = K / p.taps, p.b_ts = p.Kc;
    p.a_rs = p.Kc, p.a_cs = 1, p.b_rs = K, p.b_cs = 1, p.c_rs = N, p.splitk = 1, p.accumulate = A3T_ACC_STORE;
    p.Tseq = 1, p.colsum_slots = 1, p.c_dtype = (flags & 16) ? A3T_F32 : A3T_BF16;
    if (flags & 1) p.bias = dummy, p.act = A3T_ACT_RELU;
    if (flags & 2) p.drop_inv = 1.25f;
    if (flags & 4) p.keep_out = (unsigned char*)dummy;
    if (flags & 8) p.keep_in = (const unsigned char*)dummy, p.keep_layout = 1;      // (the row-major nibble image)
    if (flags & 32) p.colsum = dummy;
    if (flags & 64) p.S = dummy, p.s_dtype = A3T_BF16;
    return pn_applicable(p, 1, 0) ? 1 : 0;
}

template <bool CV>
static void launch_pn(const GP& pv, int grid, hipStream_t stream) {
    constexpr int lds = LDS_TOTAL;
    (void)hipFuncSetAttribute((const void*)gemm_bf16_pn_kernel<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((gemm_bf16_pn_kernel<CV>), dim3(grid), dim3(512), lds, stream, pv);
}

// Called by a3t_gemm_bf16_glds after the alignment contract has been checked.  Returns -1 when not applicable.
int a3t_gemm_bf16_pn(const GP& p, int batch, int ly, hipStream_t stream) {
    if (!pn_applicable(p, batch, ly)) return -1;
    GP pv = p;
    const int panels = (int)((p.M + PN_ROWS - 1) / PN_ROWS);
    pv.tiles_n = (p.N + PN_COLS - 1) / PN_COLS;
    pv.ntiles = panels * pv.tiles_n;
    pv.a_bytes = (unsigned)(((int64_t)p.M * p.a_rs) * 2);
    pv.b_bytes = (unsigned)(((int64_t)p.N * p.b_rs) * 2);
    const int grid = panels < pn_cus() ? panels : pn_cus();
    const bool conv = p.taps > 1;
    if (conv)
        launch_pn<true>(pv, grid, stream);
    else
        launch_pn<false>(pv, grid, stream);
    a3t_note_kernel("gemm_bf16_pn_kernel<%s>", conv ? "true" : "false");
    return (int)hipGetLastError();
}
