// gemm_bf16_8p.hip -- 256x256x64 bf16 MFMA GEMM for gfx950 on the 8-phase schedule of the CDNA4 guide, as a persistent
// kernel with the conv (im2col) loader and the epilogue folded into the load segments of the K loop.
//
// C[M][N] = epilogue( A[M][K] . B[N][K]^T ), both operands k-contiguous bf16 (Linear, Conv1d over time as implicit
// im2col, and -- through a transposed weight shadow -- their data gradients), fp32 accumulate.
//
// Structure (calibrated stand-alone in tools/probes/gemm8p.hip: 1.22-1.32 PFLOP/s at 4096^3 on uniform random data):
//   * 8 waves = 2 (wr) x 4 (wc).  LDS = 2 K-tile buffers x {A0, A1, B0, B1} half-tile images of 128 rows x 64 k (16 KiB
//     each); wave (wr, wc) owns rows wr*64..+63 of both A halves and 32 rows of both B halves: a 128 x 64 output as four
//     64 x 32 quadrants.  A K-tile is four phases (A0xB0, A0xB1, A1xB1, A1xB0) of 16 v_mfma_f32_16x16x32_bf16 each.
//   * phase = [ds_read_b128 fragments | one half-tile of DMA (2 buffer_load ... lds per lane)] s_barrier
//     [lgkmcnt(0) | setprio 1 | 16 MFMA | setprio 0] s_barrier.  Waves 4-7 run one barrier behind waves 0-3: every SIMD holds
//     one wave of each group, one multiplies while the other loads.
//   * DMA order B0, A0, B1, A1, issued 5-7 phases ahead of the first read; ONE counted wait per K-tile (vmcnt(6) in phase 4:
//     the three youngest half-tiles stay in flight); a buffer is read one phase after the wait that retires it and
//     restaged >= 2 phases after its last read (B0: 1 phase, behind an lgkmcnt that retires its reads before the barrier).
//   * LDS images are lane-linear (DMA) with the 16-byte chunk index XOR (row & 7) applied to the SOURCE address and to the
//     fragment read: conflict-free ds_read_b128.  Out-of-range rows / conv padding / K-tiles past the end read through the
//     buffer descriptor's bounds check (voffset = 0x80000000 -> zeros), so the steady state is branch-free.
//   * Persistent: one workgroup per CU walks tiles pos, pos + G, ...; the DMA cursor runs across tile boundaries (no
//     per-tile prologue).  B rows are permuted in the LDS image so that a lane's accumulators of one quadrant row are 8
//     CONSECUTIVE output columns (16-byte bf16 stores, 128-byte rows per wave) without any cross-lane exchange.
//   * Epilogue per quadrant, in the load segments that follow its last MFMA phase (phase 2, phase 4 x 2, next phase 1),
//     always issued BEFORE that segment's DMA or after the counted wait, so the vmcnt bookkeeping only ever sees DMA among
//     the youngest six operations.  bias (fp32, via a 1-KiB DMA into LDS) -> relu -> keep-bit mask -> dropout -> alpha ->
//     store bf16 | fp32, column sums (bias gradient) by DPP row reduction + atomics.
//   * keep bits: the forward conv can emit one bit per output (value > 0 after relu/dropout) in a tile-major image
//     (16 bytes per lane and tile); the matching data-gradient GEMM (same M x N output, same tiling) applies it as its
//     ReLU'/dropout mask -- 1/16 of the bytes of re-reading the bf16 activations.
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

// LDS-DMA piece (64 lanes x 16 B -> 1 KiB at the wave-uniform LDS byte address `lds_addr`) as INLINE ASM, for the token-reduction
// kernels: their fragments are read with __builtin_amdgcn_ds_read_tr16_b64, and in front of that builtin hipcc waits vmcnt(0) for
// every LDS-DMA it has seen issued through __builtin_amdgcn_raw_ptr_buffer_load_lds (it cannot tell that the transposed read does
// not alias the tiles in flight): one full drain of the DMA queue at the top of every phase -- the counted waits of the schedule
// never got to wait for anything (found in round 5 in the .s of both TN kernels: 7 "s_waitcnt vmcnt(0)" in the K loop; the
// k-contiguous kernels, whose fragments are plain ds_read_b128, have none).  An asm DMA is invisible to that bookkeeping; the
// kernels wait for it themselves (counted vmcnt + s_barrier, as written).  M0 is set in the same statement that uses it; it cannot
// be listed as a clobber (hipcc rejects reserved registers there), so the token-reduction kernels use no other M0 consumer.
__device__ __forceinline__ unsigned g8_lds_base(const void* smem0) { return (unsigned)(uintptr_t)LDS_AS(smem0); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "buffer_load_dwordx4 ... lds (16-byte LDS-DMA) exists on gfx950 only: build with --offload-arch=gfx950"
#endif
__device__ __forceinline__ void g8_dma16(const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r), "s"(soff) : "memory");
}

#ifdef G8_TIMING
__device__ unsigned long long g8_stamps[256 * 2 * 16];
#define STAMP(k) do { if (lane == 0 && (w & 3) == 0 && (k) < 16) g8_stamps[(blockIdx.x * 2 + wr) * 16 + (k)] = wall_clock64(); } while (0)
extern "C" int a3t_debug_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g8_stamps), bytes); }
#else
#define STAMP(k)
#endif
namespace {
enum { HA0 = 0, HA1 = 1, HB0 = 2, HB1 = 3 };
constexpr int HALF_BYTES = 128 * 64 * 2, TILE_BYTES = 4 * HALF_BYTES;
constexpr int LDS_BIAS = 2 * TILE_BYTES;      // 2 x 1 KiB: fp32 bias of the tile's 256 columns, by tile parity
constexpr int LDS_BITS = LDS_BIAS + 2048;     // 2 x 8 x 1 KiB: keep bits of the tile (16 B per lane), per wave, by tile parity
constexpr int LDS_CSUM = LDS_BITS + 16384;    // 2 x 1 KiB: column sums of the tile's 256 columns (fp32), by tile parity
constexpr int LDS_TOTAL = LDS_CSUM + 2048;    // 151 552 B
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
__global__ __launch_bounds__(512, 2) void gemm_bf16_8p_kernel(GP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, lane_ = lane;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), w_ = w;
    const int wr = w >> 2, wc = w & 3;
    const int G = gridDim.x;
    int pos = blockIdx.x;
    {   // bijective XCD remap: workgroup b runs on XCD b % 8; every XCD gets a contiguous run of each round's tiles
        const int q = G >> 3, r = G & 7, xcd = pos & 7;
        pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pos >> 3);
    }
    const int tiles_n = p.tiles_n, ntiles = p.ntiles;
    if (pos >= ntiles) return;
    const int n_my = (ntiles - pos + G - 1) / G;
    const int nk = p.K >> 6;             // host contract: K % 128 == 0
    const int total = n_my * nk;
    const int dm = G / tiles_n, dn = G % tiles_n;   // row-major tile order; stepping by G tiles without a division
    auto step_tile = [&](int& tm, int& tn) __attribute__((always_inline)) {
        tm += dm, tn += dn;
        if (tn >= tiles_n) tn -= tiles_n, ++tm;
    };

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
    // Epilogue-only parameters are NOT kept in SGPRs across the K loop (the kernel would spill ~80 of them): the epilogue
    // re-reads them from the kernarg segment through a pointer the optimiser cannot see through.
    typedef const __attribute__((address_space(4))) GP* kargp;
#define EPI_ARGS(q) kargp q = (kargp)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q))

    // ---- DMA lane geometry.  Wave instruction (half, q) fills LDS rows rho = (q*8 + w)*8 .. +7 of the half (1 KiB,
    // lane-linear): lane -> row rho + (lane>>3), chunk position lane&7, which holds SOURCE chunk (lane&7) ^ (rho & 7).
    // A half h holds tile rows h*128 + rho.  B half hb holds tile COLUMNS  n = (rho>>5)*64 + ((rho>>2)&3)*16 + hb*8 +
    // ((rho>>4)&1)*4 + (rho&3): fragment row (16-block j, lane group g, register r) of wave wc is column
    // wc*64 + g*16 + hb*8 + j*4 + r, i.e. a lane's 8 accumulators of one quadrant row are 8 consecutive columns.
    const int srow = lane >> 3;
    const unsigned schunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
    const int rho0 = w * 8 + srow;                                            // q = 0; q = 1: rho0 + 64
    const int nloc0 = (rho0 >> 5) * 64 + ((rho0 >> 2) & 3) * 16 + ((rho0 >> 4) & 1) * 4 + (rho0 & 3);   // q = 1: + 128
    const unsigned a_rsb_ = (unsigned)p.a_rs * 2u, b_rsb_ = (unsigned)p.b_rs * 2u;
    const int Tq = CONV ? p.Tseq : 1;

    // issue cursor: uniform (unit, K-tile, tile, tap, channel) + per-lane row state of its tile
    int c_unit = 0, c_kt = 0, c_tm = pos / tiles_n, c_tn = pos % tiles_n, c_tap = 0, c_c0 = 0;
    unsigned voffA, voffB;
    int tposA[2], nB0;     // tposA[h] = position inside the utterance of the lane's rows (q = 0 | q = 1 << 16), -16384: no such row
    auto set_tile_lanes = [&]() __attribute__((always_inline)) {
        const int m0 = c_tm * 256 + rho0;
        voffA = (unsigned)m0 * a_rsb_ + schunk16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int t[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + h * 128 + q * 64;
                t[q] = (m < p.M) ? (CONV ? m % p.Tseq : 0) : -16384;
            }
            tposA[h] = (t[0] & 0xffff) | (t[1] << 16);
        }
        nB0 = c_tn * 256 + nloc0;
        voffB = (unsigned)nB0 * b_rsb_ + schunk16;
    };
    set_tile_lanes();
    auto advance = [&]() __attribute__((always_inline)) {
        ++c_unit, ++c_kt;
        if (CONV) {       // taps innermost (the K-tiles of one channel block read the same rows shifted by a token: cache hits
            ++c_tap;      //  instead of one HBM pass per tap; same order as the 128x128 and the panel kernels: identical sums)
            if (c_tap == p.taps) c_tap = 0, c_c0 += 64;
        }
        if (c_kt == nk) {
            c_kt = 0, c_tap = 0, c_c0 = 0;
            step_tile(c_tm, c_tn);
            if (c_unit < total) set_tile_lanes();
        }
    };
    auto issue = [&](const int H, const int buf) __attribute__((always_inline)) {
        // (uniform address parts are recomputed here on purpose: hoisted out of the K loop they cost ~30 SGPRs and spill)
        int wv = w;
        unsigned a_rsb = a_rsb_, b_rsb = b_rsb_;
        asm volatile("" : "+s"(wv), "+s"(a_rsb), "+s"(b_rsb));
        unsigned char* dst = smem + buf * TILE_BYTES + H * HALF_BYTES + wv * 1024;
        const bool live = c_unit < total;
        const int h = H & 1;
        if (H < 2) {
            const int shift = CONV ? (c_tap - p.pad) * p.dil : 0;
            // (the descriptor's range check looks at voffset alone: the row offset -- which may be negative for the
            //  first tap -- must live there, only the non-negative channel offset goes into soffset)
            const unsigned so = CONV ? (unsigned)c_c0 * 2u : (unsigned)c_kt * 128u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int tp = q ? (tposA[h] >> 16) : (int)(short)(tposA[h] & 0xffff);
                const bool ok = live && ((unsigned)(tp + shift) < (unsigned)Tq);
                const unsigned vb = voffA + (unsigned)(h * 128 + q * 64 + shift) * a_rsb;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_AS(dst + q * 8192), 16, ok ? vb : OOB, so, 0, 0);
            }
        } else {
            const unsigned so = (unsigned)(h * 8) * b_rsb + (CONV ? (unsigned)(c_tap * p.Kc + c_c0) * 2u : (unsigned)c_kt * 128u);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = live && (nB0 + h * 8 + q * 128 < p.N);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LDS_AS(dst + q * 8192), 16, ok ? voffB : OOB, so + (unsigned)(q * 128) * b_rsb, 0, 0);
            }
        }
    };
    // once per tile, in the first load segment of its first K-tile: bias of its 256 columns (wave 0) and its keep bits
    auto tile_extras = [&](const int tm, const int tn, const int par) __attribute__((always_inline)) {
        EPI_ARGS(q);
        int lane = lane_, w = w_;
        asm volatile("" : "+v"(lane), "+s"(w));
        if (q->bias && w == 0) {
            const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)q->bias, 0, q->N * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rBias, LDS_AS(smem + LDS_BIAS + par * 1024), 16, (unsigned)lane * 16u, (unsigned)tn * 1024u, 0, 0);
        }
        if (q->keep_in) {
            const __amdgpu_buffer_rsrc_t rKeep = __builtin_amdgcn_make_buffer_rsrc((void*)q->keep_in, 0, q->ntiles * 8192, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rKeep, LDS_AS(smem + LDS_BITS + par * 8192 + w * 1024), 16, (unsigned)lane * 16u,
                                                     (unsigned)((tm * tiles_n + tn) * 8 + w) * 1024u, 0, 0);
        }
    };

    // ---- fragment read addresses: row (l&15) of a 16-row block, k chunk (s*4 + (l>>4)) ^ (row & 7)
    const int fr = lane & 15, g = lane >> 4;
    const unsigned fch = (unsigned)(((lane >> 4) ^ (lane & 7)) << 4);
    const unsigned aoff = (unsigned)((wr * 64 + fr) * 128) + fch;     // + i*2048; ^64 for the second k step
    const unsigned boff = (unsigned)((wc * 32 + fr) * 128) + fch;     // + j*2048

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    auto readA = [&](const unsigned char* img) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *(const bf16x8*)(img + i * 2048 + aoff);
            fa[i][1] = *(const bf16x8*)(img + i * 2048 + (aoff ^ 64u));
        }
    };
    auto readB = [&](const unsigned char* img, bf16x8(&fb)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = *(const bf16x8*)(img + j * 2048 + boff);
            fb[j][1] = *(const bf16x8*)(img + j * 2048 + (boff ^ 64u));
        }
    };
    auto quad = [&](const int ha, const int hb, const bf16x8(&fb)[2][2]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ha][hb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][s], fa[i][s], acc[ha][hb][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- epilogue of quadrant (a, hb) of tile (tm, tn): lane (fr, g) owns rows a*128 + wr*64 + i*16 + fr (i = 0..3) x
    // the 8 columns wc*64 + g*16 + hb*8 .. +7
    auto epi = [&](const int a, const int hb, const int tm, const int tn, const int par) __attribute__((always_inline)) {
        EPI_ARGS(q);
        // (everything derived from the lane id is recomputed here: hoisted out of the K loop these addresses would be spilled,
        //  and a scratch reload costs a vmcnt(0), i.e. the whole DMA pipeline)
        int lane = lane_, w = w_;
        asm volatile("" : "+v"(lane), "+s"(w));
        const int fr = lane & 15, g = lane >> 4, wr = w >> 2, wc = w & 3;
        const int nloc = wc * 64 + g * 16 + hb * 8;
        const int ncol = tn * 256 + nloc;
        const bool nok = ncol < q->N;
        float b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = 0.f;
        if (q->bias) {
            const float4 b0 = *(const float4*)(smem + LDS_BIAS + par * 1024 + nloc * 4);
            const float4 b1 = *(const float4*)(smem + LDS_BIAS + par * 1024 + nloc * 4 + 16);
            b8[0] = b0.x, b8[1] = b0.y, b8[2] = b0.z, b8[3] = b0.w, b8[4] = b1.x, b8[5] = b1.y, b8[6] = b1.z, b8[7] = b1.w;
        }
        unsigned kin = 0xffffffffu, kout = 0u;
        if (q->keep_in) kin = *(const unsigned*)(smem + LDS_BITS + par * 8192 + w * 1024 + lane * 16 + (a * 2 + hb) * 4);
        float cs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = tm * 256 + a * 128 + wr * 64 + i * 16 + fr;
            const bool ok = nok && (m < q->M);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[a][hb][i][e >> 2][e & 3] + b8[e];
            if (q->act == A3T_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (q->keep_in) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ((kin >> (i * 8 + e)) & 1u) ? v[e] : 0.f;
            }
            const int64_t idx = (int64_t)m * q->c_rs + ncol;
            if (q->drop_inv > 0.f) {
                const unsigned t = q->drop_thr >> 16;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const unsigned hsh = rng_pair(q->drop_key, ((unsigned)idx + (unsigned)e) >> 1);
                    v[e] = ((hsh & 0xffffu) >= t) ? v[e] * q->drop_inv : 0.f;
                    v[e + 1] = ((hsh >> 16) >= t) ? v[e + 1] * q->drop_inv : 0.f;
                }
            }
            unsigned kb = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= q->alpha;
            if (q->R && ok) {      // fp32 residual (plain loads: the compiler drains the DMA queue for them, see the host heuristic)
                const float4 r0 = *(const float4*)(q->R + idx), r1 = *(const float4*)(q->R + idx + 4);
                v[0] += r0.x, v[1] += r0.y, v[2] += r0.z, v[3] += r0.w, v[4] += r1.x, v[5] += r1.y, v[6] += r1.z, v[7] += r1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) kb |= (v[e] > 0.f ? 1u : 0u) << e;
            kout |= kb << (i * 8);
            if (ok) {
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
            acc[a][hb][i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[a][hb][i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (q->keep_out)
            *(unsigned*)(q->keep_out + ((size_t)((tm * tiles_n + tn) * 8 + w) * 64 + lane) * 16 + (a * 2 + hb) * 4) = kout;
        if (q->colsum) {   // LDS accumulators of this tile; csum_flush() sends them on once every wave is done with the tile
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] = row16_sum(cs[e]);
            if (fr == 0) {
                float* acc_l = (float*)(smem + LDS_CSUM + par * 1024) + nloc;
#pragma unroll
                for (int e = 0; e < 8; ++e) __hip_atomic_fetch_add(acc_l + e, cs[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    // column sums of a finished tile (LDS, parity par): each wave forwards 32 of the 256 columns with ONE global atomic
    // instruction and clears them.  Called >= 2 barriers after the tile's last quadrant epilogue.
    auto csum_flush = [&](const int tn, const int par) __attribute__((always_inline)) {
        EPI_ARGS(q);
        int lane = lane_, w = w_;
        asm volatile("" : "+v"(lane), "+s"(w));
        if (q->colsum && lane < 32) {
            float* a_l = (float*)(smem + LDS_CSUM + par * 1024) + w * 32 + lane;
            const float v = *a_l;
            *a_l = 0.f;
            const int col = tn * 256 + w * 32 + lane;
            if (col < q->N) atomicAdd(q->colsum + col, q->colsum_scale * v);
        }
    };

    // ---- prologue: first tile's bias / keep bits, unit 0 complete, B0 A0 B1 of unit 1 (its A1 goes out in phase 1)
    int u_tm = c_tm, u_tn = c_tn, u_kt = 0, u_par = 0;     // compute cursor
    int e_tm = 0, e_tn = 0, e_par = 0;                     // tile whose last quadrant (A1 x B0) is still to be written
    bool pend = false;
    int f_tn = 0;                                          // column tile of the tile whose column sums await their flush
    if (tid < 512) ((float*)(smem + LDS_CSUM))[tid] = 0.f;     // (made visible by the prologue barrier)
    int st_k = 2;
    (void)st_k;
    STAMP(0);
    tile_extras(u_tm, u_tn, 0);
    issue(HB0, 0), issue(HA0, 0), issue(HB1, 0), issue(HA1, 0);
    advance();
    issue(HB0, 1), issue(HA0, 1), issue(HB1, 1);
    WAIT_VM(6);            // unit 0 and the extras have landed (this wave's share)
    WAIT_LGKM(0);          // (the cleared column-sum accumulators)
    BAR();                 // ... everyone's share
    if (wr == 1) BAR();    // second group runs one barrier behind
    STAMP(1);

    for (int u = 0; u < total; u += 2) {
        // =================== even K-tile (buffer 0); the first K-tile of a tile is always even =====================
        {
            const unsigned char* cur = smem;
            const bool first = (u_kt == 0);
            // phase 1: A0 x B0.  DMA: A1 of the unit under the cursor (this unit + 1); then the cursor moves on
            readB(cur + HB0 * HALF_BYTES, fb0);
            SB();
            readA(cur + HA0 * HALF_BYTES);
            if (first && u > 0) tile_extras(u_tm, u_tn, u_par);
            issue(HA1, 1);
            advance();
            WAIT_LGKM(8);      // the B0 reads have returned: B0 may be restaged in the next phase
            if (pend) {
                epi(1, 0, e_tm, e_tn, e_par);
                pend = false;
            }
            BAR();
            WAIT_LGKM(0);
            SB();
            quad(0, 0, fb0);
            BAR();
            // phase 2: A0 x B1.  DMA: B0 of unit + 2
            readB(cur + HB1 * HALF_BYTES, fb1);
            issue(HB0, 0);
            BAR();
            WAIT_LGKM(0);
            SB();
            quad(0, 1, fb1);
            BAR();
            // phase 3: A1 x B1.  DMA: A0 of unit + 2
            readA(cur + HA1 * HALF_BYTES);
            issue(HA0, 0);
            BAR();
            WAIT_LGKM(0);
            SB();
            quad(1, 1, fb1);
            BAR();
            // phase 4: A1 x B0.  DMA: B1 of unit + 2; all but the three youngest half-tiles have landed = unit + 1
            issue(HB1, 0);
            WAIT_VM(6);
            if (first && u > 0) csum_flush(f_tn, u_par ^ 1);     // previous tile: every wave wrote its last quadrant 3 phases ago
            BAR();
            quad(1, 0, fb0);
            BAR();
        }
        // =================== odd K-tile (buffer 1); the last K-tile of a tile is always odd ========================
        {
            const unsigned char* cur = smem + TILE_BYTES;
            const bool last = (u_kt + 2 == nk);
            readB(cur + HB0 * HALF_BYTES, fb0);
            SB();
            readA(cur + HA0 * HALF_BYTES);
            issue(HA1, 0);
            advance();
            WAIT_LGKM(8);
            BAR();
            WAIT_LGKM(0);
            SB();
            quad(0, 0, fb0);
            BAR();
            // phase 2: the quadrant A0 x B0 is complete -> its epilogue goes out BEFORE this segment's DMA
            readB(cur + HB1 * HALF_BYTES, fb1);
            if (last) epi(0, 0, u_tm, u_tn, u_par);
            issue(HB0, 1);
            BAR();
            WAIT_LGKM(0);
            SB();
            quad(0, 1, fb1);
            BAR();
            readA(cur + HA1 * HALF_BYTES);
            issue(HA0, 1);
            BAR();
            WAIT_LGKM(0);
            SB();
            quad(1, 1, fb1);
            BAR();
            // phase 4: after the counted wait (its six youngest operations are DMA only): A0 x B1 and A1 x B1
            issue(HB1, 1);
            WAIT_VM(6);
            if (last) {
                epi(0, 1, u_tm, u_tn, u_par);
                epi(1, 1, u_tm, u_tn, u_par);
            }
            BAR();
            quad(1, 0, fb0);
            BAR();
            u_kt += 2;
            if (last) {
                STAMP(st_k);
                ++st_k;
                // A1 x B0 was just issued: it is written in the next load segment (or after the loop)
                e_tm = u_tm, e_tn = u_tn, e_par = u_par, pend = true;
                f_tn = u_tn;
                u_kt = 0, u_par ^= 1;
                step_tile(u_tm, u_tn);
            }
        }
    }
    if (pend) epi(1, 0, e_tm, e_tn, e_par);
    STAMP(st_k);
    WAIT_LGKM(0);
    if (wr == 0) BAR();
    BAR();                 // every wave has added its last column sums
    csum_flush(f_tn, u_par ^ 1);
    WAIT_VM(0);            // the trailing (zero) DMA must not outlive the workgroup's LDS allocation
}


// =====================================================================================================================
// TN variant: C[M][N] (+)= alpha * sum_k A[k][m] * B[k][n], both operands REDUCTION-strided (token-major activations):
// the weight gradients of Linear / Conv1d (multi_layer_conv.py:36-63 backward), reduction over the B*T tokens split over
// workgroups (fp32 atomics).  Same 8-phase schedule; what changes:
//   * half-tile image = 64 k-rows x 128 m (256 B per k-row); a wave DMA instruction = 4 k-rows; chunk position p of k-row
//     kr holds source chunk p ^ (((kr & 3) << 2) | (((kr >> 3) & 1) << 1)): the 8 k-rows x 32 B that the 32 lanes of a
//     ds_read_b64_tr_b16 group touch fall on disjoint banks;
//   * fragments by two transposed LDS reads (4 k each) per 16x16x32 operand;
//   * fused conv weight gradient (WG): output columns are (tap, c); a 128-column B half lies inside one tap and reads
//     x[k + (tap - pad) * dil] with zeros across utterance boundaries (buffer range check, as in the forward loader);
//   * one tile x one K split per workgroup; epilogue = atomics straight from the accumulators (64-byte row segments).
template <bool WG>
__global__ __launch_bounds__(512, 2) void gemm_bf16_8p_tn_kernel(GP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    int wi = blockIdx.x;
    {   // slice-major, XCD-contiguous: the tiles of one K split (same operand slabs) stay inside one XCD's L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wi & 7;
        wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wi >> 3);
    }
    const int bid = wi % p.ntiles, ks = wi / p.ntiles;
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int nkt = (p.K + 63) >> 6;              // (tokens past K read zeros through the range check)
    int per = (nkt + p.splitk - 1) / p.splitk;
    per += per & 1;                               // whole pairs of K-tiles; tiles past the end read zeros
    const int kt0 = ks * per;
    if (kt0 >= nkt) return;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);

    // ---- DMA lane geometry: instruction (half, q) fills k-rows (q*8 + w)*4 + (lane>>4), chunk position lane&15
    const int krl = w * 4 + (lane >> 4);                                        // q = 0; q = 1: + 32
    const int sc = (lane & 15) ^ (((lane >> 4) << 2) | (((w >> 1) & 1) << 1));   // source chunk of this lane (same for q = 0, 1)
    const unsigned a_csb = (unsigned)p.a_cs * 2u, b_csb = (unsigned)p.b_cs * 2u;
    const int mA = tm * 256 + sc * 8;                                            // + h*128
    const unsigned voffA = (unsigned)krl * a_csb + (unsigned)mA * 2u;
    const int cin = WG ? p.N / p.taps : p.N;
    int shiftB[2], c0B[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n0 = tn * 256 + h * 128;
        const int tap = WG ? n0 / cin : 0;
        shiftB[h] = WG ? (tap - p.pad) * p.dil : 0;
        c0B[h] = n0 - tap * cin;
    }
    const unsigned voffB = (unsigned)krl * b_csb + (unsigned)(sc * 16);
    int tpos = 0;                                                                // (q = 0 | q = 1 << 16): token position inside its utterance
    if (WG) {
        const int t0 = (kt0 * 64 + krl) % p.Tseq, t1 = (kt0 * 64 + krl + 32) % p.Tseq;
        tpos = t0 | (t1 << 16);
    }
    int c_kt = kt0;
    auto advance = [&]() __attribute__((always_inline)) {
        ++c_kt;
        if (WG) {
            int t0 = (tpos & 0xffff) + 64, t1 = (tpos >> 16) + 64;
            if (p.Tseq >= 64) {
                t0 = t0 >= p.Tseq ? t0 - p.Tseq : t0, t1 = t1 >= p.Tseq ? t1 - p.Tseq : t1;
            } else {
                t0 %= p.Tseq, t1 %= p.Tseq;
            }
            tpos = t0 | (t1 << 16);
        }
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(g8_lds_base(smem));
    auto issue = [&](const int H, const int buf) __attribute__((always_inline)) {
        int wv = w;
        unsigned acs = a_csb, bcs = b_csb;
        asm volatile("" : "+s"(wv), "+s"(acs), "+s"(bcs));
        const unsigned dst = lds0 + (unsigned)(buf * TILE_BYTES + H * HALF_BYTES) + (unsigned)wv * 1024u;
        const int krem = p.K - c_kt * 64 - krl;       // > q*32: the lane's token row exists
        const int h = H & 1;
        if (H < 2) {
            const bool colok = mA + h * 128 < p.M;
            const unsigned so = (unsigned)c_kt * 64u * acs;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned vb = voffA + (unsigned)(h * 256) + (unsigned)(q * 32) * acs;
                g8_dma16(rA, dst + q * 8192, (colok && krem > q * 32) ? vb : OOB, so);
            }
        } else {
            const bool colok = tn * 256 + h * 128 + sc * 8 < p.N;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int tp = q ? (tpos >> 16) : (tpos & 0xffff);
                const bool ok = colok && (krem > q * 32) && (!WG || ((unsigned)(tp + shiftB[h]) < (unsigned)p.Tseq));
                // (the whole token offset lives in voffset: the range check ignores soffset, and the tap shift may be negative)
                const unsigned vb = voffB + (unsigned)(c_kt * 64 + q * 32 + shiftB[h]) * bcs + (unsigned)(c0B[h] * 2);
                g8_dma16(rB, dst + q * 8192, ok ? vb : OOB, 0u);
            }
        }
    };

    // ---- transposed fragment reads: lane (g, pp) supplies the address of 4 consecutive m of k-row g*8 + (pp>>2) (+4) and
    // receives column pp of the 16-column block: 8 k-values of row/column pp -- the MFMA operand layout
    const int g = lane >> 4, pp = lane & 15;
    const unsigned swz = (unsigned)(((pp >> 2) << 2) | ((g & 1) << 1));
    const unsigned kbyte = (unsigned)(g * 8 + (pp >> 2)) * 256u + (unsigned)(pp & 1) * 8u;
    unsigned offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = kbyte + ((((unsigned)(wr * 8 + i * 2) + (unsigned)((pp & 3) >> 1)) ^ swz) << 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = kbyte + ((((unsigned)(wc * 4 + j * 2) + (unsigned)((pp & 3) >> 1)) ^ swz) << 4);
    auto frag = [&](const unsigned char* img, unsigned off, int s) __attribute__((always_inline)) -> bf16x8 {
        const unsigned char* a0 = img + off + s * 8192;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0 + 1024));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    auto readA = [&](const unsigned char* img, const int i0, const int i1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = i0; i < i1; ++i) fa[i][0] = frag(img, offA[i], 0), fa[i][1] = frag(img, offA[i], 1);
    };
    auto readB = [&](const unsigned char* img, bf16x8(&fb)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j][0] = frag(img, offB[j], 0), fb[j][1] = frag(img, offB[j], 1);
    };
    auto quad = [&](const int ha, const int hb, const bf16x8(&fb)[2][2]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ha][hb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], fb[j][s], acc[ha][hb][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    STAMP(0);
    issue(HB0, 0), issue(HA0, 0), issue(HB1, 0), issue(HA1, 0);
    advance();
    issue(HB0, 1), issue(HA0, 1), issue(HB1, 1);
    WAIT_VM(6);
    BAR();
    if (wr == 1) BAR();
    STAMP(1);

    auto ktile = [&](const int buf) __attribute__((always_inline)) {
        const unsigned char* cur = smem + buf * TILE_BYTES;
        // phase 1: A0 x B0 (8 + 16 transposed reads; the lgkmcnt field counts to 15: the wait that retires the B0 reads sits
        // after the first half of the A reads)
        readB(cur + HB0 * HALF_BYTES, fb0);
        SB();
        readA(cur + HA0 * HALF_BYTES, 0, 2);
        WAIT_LGKM(8);
        SB();
        readA(cur + HA0 * HALF_BYTES, 2, 4);
        issue(HA1, buf ^ 1);
        advance();
        BAR();
        WAIT_LGKM(0);
        SB();
        quad(0, 0, fb0);
        BAR();
        readB(cur + HB1 * HALF_BYTES, fb1);
        issue(HB0, buf);
        BAR();
        WAIT_LGKM(0);
        SB();
        quad(0, 1, fb1);
        BAR();
        readA(cur + HA1 * HALF_BYTES, 0, 4);
        issue(HA0, buf);
        BAR();
        WAIT_LGKM(0);
        SB();
        quad(1, 1, fb1);
        BAR();
        issue(HB1, buf);
        WAIT_VM(6);
        BAR();
        quad(1, 0, fb0);
        BAR();
    };
    for (int u = 0; u < per; u += 2) {
        ktile(0);
        ktile(1);
    }
    STAMP(2);
    if (wr == 0) BAR();
    WAIT_VM(0);

    // ---- epilogue: lane (g, pp) holds rows a*128 + wr*64 + i*16 + g*4 + r, column hb*128 + wc*32 + j*16 + pp
    if (p.slab) {
        // split-K partial of this (tile, K split): the accumulators go out as they lie in the registers -- 16 bytes per lane and
        // fragment, 1 KiB contiguous per wave instruction (tn_slab_index) -- and gemm_8p_tn_fold_kernel sums the splits of a tile
        // in a fixed order.  (As fp32 atomics straight into C the same 64 values per lane are 64 instructions of 256 scattered
        // bytes each and resolve at the memory side: 48 us for 240 workgroups against 5 us of stores, tools/probes/atomic_epilogue.hip.)
        float* slab = p.slab + ((int64_t)ks * p.ntiles + bid) * 65536 + (w * 64 + lane) * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (tm * 256 + a * 128 >= p.M) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (tn * 256 + b * 128 >= p.N) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        *(f32x4*)(slab + ((((a * 2 + b) * 4 + i) * 2 + j) * 2048)) = acc[a][b][i][j];
            }
        }
        return;
    }
    float* C = (float*)p.C;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = tm * 256 + a * 128 + wr * 64 + i * 16 + g * 4 + r;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = tn * 256 + b * 128 + wc * 32 + j * 16 + pp;
                        if (m < p.M && n < p.N) {
                            const float v = p.alpha * acc[a][b][i][j][r];
                            float* c = C + (int64_t)m * p.c_rs + n;
                            if (p.accumulate == A3T_ACC_ATOMIC)
                                atomicAdd(c, v);
                            else if (p.accumulate == A3T_ACC_ADD)
                                *c += v;
                            else
                                *c = v;
                        }
                    }
            }
#ifdef G8_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(3);
#endif
}


// Several token reductions over the SAME tokens in one launch (a3t_gemm_tn3_group): problem i owns tiles [tile0, next tile0) of
// every K split.  Passed by value beside GP; n == 0: the single problem described by GP.
struct TN3Prob {
    const void* A;
    const void* B;
    float* C;
    int64_t c_rs;
    int M, N;
    unsigned a_csb, b_csb;      // operand row strides in bytes
    unsigned a_bytes, b_bytes;
    float alpha;
    int accumulate, tile0, tiles_n;
};
struct TN3Group {
    int n;
    TN3Prob q[8];
};

// =====================================================================================================================
// TN3 variant (round 5): the same token reduction on a 128 x 384 tile = ONE A half x THREE B halves, three phases per K-tile.
// Every weight gradient of the model has a 384-multiple of input channels per tap (d_model = 384, ff = 1536 = 4 x 384), so
// 128 x 384 tiles cover dW exactly: 1536 x (3 x 384) and 384 x (3 x 1536) are 36 full tiles each, where 256 x 256 tiles fill
// 0.90 / 0.75 of their 30 / 36 tiles; the Linear weight gradients (N = 384) are one tile wide.  The A fragments (64 rows per wave,
// the larger operand) are read once per K-tile and stay in registers for the three B halves: 20 KiB of LDS reads per wave and
// K-tile for 64 x 96 outputs (the 2 x 2 tile: 24 KiB for 128 x 64).
//   phase 1 of K-tile t: read B0 + A0 | DMA B1(t+1),          vmcnt(8)  -> B1(t) landed        | A0 x B0
//   phase 2:             read B1      | DMA B2(t+1), B0(t+2), vmcnt(10) -> B2(t) landed        | A0 x B1
//   phase 3:             read B2      | DMA A0(t+2),          vmcnt(8)  -> B0, A0(t+1) landed  | A0 x B2
//   One counted wait per phase, each retiring exactly the half that is read in the NEXT phase and was requested three (B0: four)
//   phases earlier; four to five half-tiles stay in flight across every barrier.  (A first version waited once per K-tile with
//   vmcnt(4): that wait also retired B1 / B2 of tile t+1, requested one and two phases earlier -- their L2 latency was exposed in
//   every K-tile: 138 us per FFN weight gradient against the figure in DESIGN.md.)
// Restaging distances: B0 one phase after its reads (retired by the lgkmcnt in front of phase 1's first barrier), A0 / B1 / B2
// two phases after theirs -- the rules of the 2 x 2 kernel above.  Epilogue: split-K partial tile by plain stores (slab).
template <bool WG>
__global__ __launch_bounds__(512, 2) void gemm_bf16_8p_tn3_kernel(GP p, TN3Group grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    enum { H3A = 0, H3B0 = 1, H3B1 = 2, H3B2 = 3 };
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    int wi = blockIdx.x;
    {   // slice-major, XCD-contiguous: the tiles of one K split (same operand slabs) stay inside one XCD's L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wi & 7;
        wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wi >> 3);
    }
    const int bid = wi % p.ntiles, ks = wi / p.ntiles;
    // the problem this tile belongs to (uniform): a group member or GP itself
    const void* Ap = p.A;
    const void* Bp = p.B;
    int Mp = p.M, Np = p.N, tiles_n = p.tiles_n, lbid = bid;
    unsigned a_csb = (unsigned)p.a_cs * 2u, b_csb = (unsigned)p.b_cs * 2u, a_bytes = p.a_bytes, b_bytes = p.b_bytes;
    if (grp.n > 0) {
        int k = 0;
        for (int i = 1; i < grp.n; ++i)
            if (bid >= grp.q[i].tile0) k = i;
        Ap = grp.q[k].A, Bp = grp.q[k].B, Mp = grp.q[k].M, Np = grp.q[k].N, tiles_n = grp.q[k].tiles_n;
        a_csb = grp.q[k].a_csb, b_csb = grp.q[k].b_csb, a_bytes = grp.q[k].a_bytes, b_bytes = grp.q[k].b_bytes;
        lbid = bid - grp.q[k].tile0;
    }
    const int tn = lbid % tiles_n, tm = lbid / tiles_n;
    const int nkt = (p.K + 63) >> 6;
    int per = (nkt + p.splitk - 1) / p.splitk;
    per += per & 1;
    const int kt0 = ks * per;
    if (kt0 >= nkt) return;                       // (the host folds only the splits that have K-tiles)

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)b_bytes, 0x00020000);

    // ---- DMA lane geometry (as the 2 x 2 kernel): instruction (half, q) fills k-rows (q*8 + w)*4 + (lane>>4), chunk position lane&15
    const int krl = w * 4 + (lane >> 4);
    const int sc = (lane & 15) ^ (((lane >> 4) << 2) | (((w >> 1) & 1) << 1));
    const int mA = tm * 128 + sc * 8;
    const unsigned voffA = (unsigned)krl * a_csb + (unsigned)mA * 2u;
    const int cin = WG ? Np / p.taps : Np;
    int shiftB[3], c0B[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const int n0 = tn * 384 + h * 128;
        const int tap = WG ? n0 / cin : 0;
        shiftB[h] = WG ? (tap - p.pad) * p.dil : 0;
        c0B[h] = n0 - tap * cin;
    }
    const unsigned voffB = (unsigned)krl * b_csb + (unsigned)(sc * 16);
    int tpos = 0;
    if (WG) {
        const int t0 = (kt0 * 64 + krl) % p.Tseq, t1 = (kt0 * 64 + krl + 32) % p.Tseq;
        tpos = t0 | (t1 << 16);
    }
    int c_kt = kt0;
    auto advance = [&]() __attribute__((always_inline)) {
        ++c_kt;
        if (WG) {
            int t0 = (tpos & 0xffff) + 64, t1 = (tpos >> 16) + 64;
            if (p.Tseq >= 64) {
                t0 = t0 >= p.Tseq ? t0 - p.Tseq : t0, t1 = t1 >= p.Tseq ? t1 - p.Tseq : t1;
            } else {
                t0 %= p.Tseq, t1 %= p.Tseq;
            }
            tpos = t0 | (t1 << 16);
        }
    };
    // kt / tp: K-tile and token positions the half belongs to (the B1 / B2 halves of K-tile t+1 are requested after the cursor
    // has moved on to t+2 for B0: the caller hands in the values it saved)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(g8_lds_base(smem));
    auto issue = [&](const int H, const int buf, const int kt, const int tp) __attribute__((always_inline)) {
        int wv = w;
        unsigned acs = a_csb, bcs = b_csb;
        asm volatile("" : "+s"(wv), "+s"(acs), "+s"(bcs));
        const unsigned dst = lds0 + (unsigned)(buf * TILE_BYTES + H * HALF_BYTES) + (unsigned)wv * 1024u;
        const int krem = p.K - kt * 64 - krl;
        if (H == H3A) {
            const bool colok = mA < Mp;
            const unsigned so = (unsigned)kt * 64u * acs;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned vb = voffA + (unsigned)(q * 32) * acs;
                g8_dma16(rA, dst + q * 8192, (colok && krem > q * 32) ? vb : OOB, so);
            }
        } else {
            const int h = H - 1;
            const bool colok = tn * 384 + h * 128 + sc * 8 < Np;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int t = q ? (tp >> 16) : (tp & 0xffff);
                const bool ok = colok && (krem > q * 32) && (!WG || ((unsigned)(t + shiftB[h]) < (unsigned)p.Tseq));
                const unsigned vb = voffB + (unsigned)(kt * 64 + q * 32 + shiftB[h]) * bcs + (unsigned)(c0B[h] * 2);
                g8_dma16(rB, dst + q * 8192, ok ? vb : OOB, 0u);
            }
        }
    };

    const int g = lane >> 4, pp = lane & 15;
    const unsigned swz = (unsigned)(((pp >> 2) << 2) | ((g & 1) << 1));
    const unsigned kbyte = (unsigned)(g * 8 + (pp >> 2)) * 256u + (unsigned)(pp & 1) * 8u;
    unsigned offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = kbyte + ((((unsigned)(wr * 8 + i * 2) + (unsigned)((pp & 3) >> 1)) ^ swz) << 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = kbyte + ((((unsigned)(wc * 4 + j * 2) + (unsigned)((pp & 3) >> 1)) ^ swz) << 4);
    auto frag = [&](const unsigned char* img, unsigned off, int s) __attribute__((always_inline)) -> bf16x8 {
        const unsigned char* a0 = img + off + s * 8192;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0 + 1024));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x4 acc[3][4][2];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[4][2], fb[2][2][2];        // fb[parity]: the B fragments of a phase are read while the previous phase's are in use
    auto readA = [&](const unsigned char* img, const int i0, const int i1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = i0; i < i1; ++i) fa[i][0] = frag(img, offA[i], 0), fa[i][1] = frag(img, offA[i], 1);
    };
    auto readB = [&](const unsigned char* img, bf16x8(&f)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) f[j][0] = frag(img, offB[j], 0), f[j][1] = frag(img, offB[j], 1);
    };
    auto quad = [&](const int hb, const bf16x8(&f)[2][2]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[hb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], f[j][s], acc[hb][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // K-tile kt0 -> buffer 0 (all four halves), B0 / A0 of K-tile kt0 + 1 -> buffer 1
    issue(H3B0, 0, c_kt, tpos), issue(H3A, 0, c_kt, tpos), issue(H3B1, 0, c_kt, tpos), issue(H3B2, 0, c_kt, tpos);
    advance();
    issue(H3B0, 1, c_kt, tpos), issue(H3A, 1, c_kt, tpos);
    WAIT_VM(4);
    BAR();
    if (wr == 1) BAR();

    auto ktile = [&](const int buf) __attribute__((always_inline)) {
        const unsigned char* cur = smem + buf * TILE_BYTES;
        // at entry the cursor (c_kt, tpos) is K-tile t+1, whose B0 / A0 are in flight or landed in buf ^ 1
        const int kt1 = c_kt, tp1 = tpos;
        // phase 1: A0 x B0
        readB(cur + H3B0 * HALF_BYTES, fb[0]);
        SB();
        readA(cur + H3A * HALF_BYTES, 0, 2);
        WAIT_LGKM(8);                    // the B0 reads (issued first) are retired before the barrier: B0 is restaged in phase 2
        SB();
        readA(cur + H3A * HALF_BYTES, 2, 4);
        issue(H3B1, buf ^ 1, kt1, tp1);
        WAIT_VM(8);                      // retires B1 of the current tile (requested 3 phases ago, read in phase 2)
        BAR();
        WAIT_LGKM(0);
        SB();
        quad(0, fb[0]);
        BAR();
        // phase 2: A0 x B1
        readB(cur + H3B1 * HALF_BYTES, fb[1]);
        issue(H3B2, buf ^ 1, kt1, tp1);
        advance();
        issue(H3B0, buf, c_kt, tpos);
        WAIT_VM(10);                     // retires B2 of the current tile (requested 3 phases ago, read in phase 3)
        BAR();
        WAIT_LGKM(0);
        SB();
        quad(1, fb[1]);
        BAR();
        // phase 3: A0 x B2
        readB(cur + H3B2 * HALF_BYTES, fb[0]);
        issue(H3A, buf, c_kt, tpos);
        WAIT_VM(8);                      // retires B0 / A0 of the next tile (requested 4 / 3 phases ago)
        BAR();
        WAIT_LGKM(0);
        SB();
        quad(2, fb[0]);
        BAR();
    };
    for (int u = 0; u < per; u += 2) {
        ktile(0);
        ktile(1);
    }
    if (wr == 0) BAR();
    WAIT_VM(0);

    // ---- epilogue: lane (g, pp) holds rows wr*64 + i*16 + g*4 + r, column b*128 + wc*32 + j*16 + pp of the tile
    float* slab = p.slab + ((int64_t)ks * p.ntiles + bid) * 49152 + (w * 64 + lane) * 4;
    if (tm * 128 >= Mp) return;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (tn * 384 + b * 128 >= Np) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) *(f32x4*)(slab + (((b * 4 + i) * 2 + j) * 2048)) = acc[b][i][j];
    }
}

// Fold of the TN3 kernel's partial tiles (128 x 384, 12288 16-byte pieces per tile).
__global__ __launch_bounds__(256) void gemm_8p_tn3_fold_kernel(GP p, TN3Group grp) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int bid = blockIdx.y;
    float* Cp = (float*)p.C;
    int64_t c_rs = p.c_rs;
    int Mp = p.M, Np = p.N, tiles_n = p.tiles_n, lbid = bid, accumulate = (p.accumulate == A3T_ACC_ATOMIC && p.sole_writer) ? A3T_ACC_ADD : p.accumulate;
    float alpha = p.alpha;
    if (grp.n > 0) {
        int k = 0;
        for (int i = 1; i < grp.n; ++i)
            if (bid >= grp.q[i].tile0) k = i;
        Cp = grp.q[k].C, c_rs = grp.q[k].c_rs, Mp = grp.q[k].M, Np = grp.q[k].N, tiles_n = grp.q[k].tiles_n;
        accumulate = grp.q[k].accumulate, alpha = grp.q[k].alpha, lbid = bid - grp.q[k].tile0;       // (SOLE arrives as ADD)
    }
    const int tn = lbid % tiles_n, tm = lbid / tiles_n;
    const int lane = t & 63, w = (t >> 6) & 7, q = t >> 9;
    const int j = q & 1, i = (q >> 1) & 3, b = q >> 3;
    const int wr = w >> 2, wc = w & 3, g = lane >> 4, pp = lane & 15;
    const int m = tm * 128 + wr * 64 + i * 16 + g * 4;
    const int n = tn * 384 + b * 128 + wc * 32 + j * 16 + pp;
    if (tn * 384 + b * 128 >= Np || n >= Np || m >= Mp) return;
    const float* s = p.slab + (int64_t)bid * 49152 + (int64_t)t * 4;
    const int64_t sstride = (int64_t)p.ntiles * 49152;
    // splits in ascending order, four loads in flight at a time (a fixed order: the result does not depend on the launch)
    f32x4 v = *(const f32x4*)s;
    int k = 1;
    for (; k + 4 <= p.splitk; k += 4) {
        const f32x4 u0 = *(const f32x4*)(s + (k + 0) * sstride), u1 = *(const f32x4*)(s + (k + 1) * sstride);
        const f32x4 u2 = *(const f32x4*)(s + (k + 2) * sstride), u3 = *(const f32x4*)(s + (k + 3) * sstride);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (((v[r] + u0[r]) + u1[r]) + u2[r]) + u3[r];
    }
    for (; k < p.splitk; ++k) {
        const f32x4 u = *(const f32x4*)(s + k * sstride);
        v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
    }
    float* C = Cp + (int64_t)m * c_rs + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (m + r >= Mp) break;
        const float x = alpha * v[r];
        if (accumulate == A3T_ACC_ATOMIC)
            atomicAdd(C + (int64_t)r * c_rs, x);
        else if (accumulate == A3T_ACC_ADD)
            C[(int64_t)r * c_rs] += x;
        else
            C[(int64_t)r * c_rs] = x;
    }
}

// Fold of the split-K partial tiles written by gemm_bf16_8p_tn_kernel: C (+)= alpha * sum_s slab[s][tile], splits summed in
// ascending order (deterministic).  One thread per 16-byte fragment piece: rows m..m+3 of one column.
__global__ __launch_bounds__(256) void gemm_8p_tn_fold_kernel(GP p) {
    const int t = blockIdx.x * 256 + threadIdx.x;            // 16384 pieces per tile
    const int bid = blockIdx.y;
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int lane = t & 63, w = (t >> 6) & 7, q = t >> 9;
    const int j = q & 1, i = (q >> 1) & 3, b = (q >> 3) & 1, a = q >> 4;
    const int wr = w >> 2, wc = w & 3, g = lane >> 4, pp = lane & 15;
    const int m = tm * 256 + a * 128 + wr * 64 + i * 16 + g * 4;
    const int n = tn * 256 + b * 128 + wc * 32 + j * 16 + pp;
    if (tm * 256 + a * 128 >= p.M || tn * 256 + b * 128 >= p.N) return;        // (never written)
    const float* s = p.slab + (int64_t)bid * 65536 + (int64_t)t * 4;
    const int64_t sstride = (int64_t)p.ntiles * 65536;
    f32x4 v = *(const f32x4*)s;
    for (int k = 1; k < p.splitk; ++k) {
        const f32x4 u = *(const f32x4*)(s + k * sstride);
        v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
    }
    if (n >= p.N) return;
    float* C = (float*)p.C + (int64_t)m * p.c_rs + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (m + r >= p.M) break;
        const float x = p.alpha * v[r];
        if (p.accumulate == A3T_ACC_ATOMIC && !p.sole_writer)
            atomicAdd(C + (int64_t)r * p.c_rs, x);
        else if (p.accumulate != A3T_ACC_STORE)
            C[(int64_t)r * p.c_rs] += x;
        else
            C[(int64_t)r * p.c_rs] = x;
    }
}

// Split-K partial workspace: one per (device, stream) -- a launch and its fold are ordered on their stream, launches on
// different streams must not share slabs.  Grow-only per stream (hipMalloc on first use / growth; a few launches during warm-up);
// the device is the STREAM's (a launch on another device's stream gets its slab there), at most G8_MAX_SLABS live at a time (the
// least recently used one is drained and freed), and a3t_release_workspaces() frees them all.
#include <map>
#include <mutex>
#define G8_MAX_SLABS 16
struct G8Slab {
    float* p;
    size_t bytes;
    unsigned long long used;
};
static std::mutex g8_slab_mu;
static std::map<std::pair<int, hipStream_t>, G8Slab> g8_slabs;
static unsigned long long g8_slab_clock = 0;
static float* g8_slab(hipStream_t stream, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (stream) {
        hipDevice_t sd = 0;
        if (hipStreamGetDevice(stream, &sd) == hipSuccess) dev = (int)sd;
    }
    std::lock_guard<std::mutex> lk(g8_slab_mu);
    const std::pair<int, hipStream_t> key(dev, stream);
    if (!g8_slabs.count(key) && g8_slabs.size() >= G8_MAX_SLABS) {
        auto lru = g8_slabs.begin();
        for (auto it = g8_slabs.begin(); it != g8_slabs.end(); ++it)
            if (it->second.used < lru->second.used) lru = it;
        if (lru->second.p) {
            (void)hipStreamSynchronize(lru->first.second);      // (a destroyed stream: the error is ignored, the memory is idle)
            (void)hipFree(lru->second.p);
        }
        g8_slabs.erase(lru);
    }
    G8Slab& e = g8_slabs[key];
    e.used = ++g8_slab_clock;
    if (e.bytes < bytes) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        if (e.p) {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(e.p);
        }
        e.p = nullptr, e.bytes = 0;
        const hipError_t r = hipMalloc((void**)&e.p, bytes);
        if (cur != dev) (void)hipSetDevice(cur);
        if (r != hipSuccess) {
            e.p = nullptr;
            return nullptr;
        }
        e.bytes = bytes;
    }
    return e.p;
}
// frees every split-K slab (after draining the stream it belongs to) and the attention key-split workspace; the next launch that
// needs one allocates again
void attn_release_split_ws();      // attn_fused.hip
extern "C" int a3t_release_workspaces(void) {
    attn_release_split_ws();
    std::lock_guard<std::mutex> lk(g8_slab_mu);
    for (auto& kv : g8_slabs)
        if (kv.second.p) {
            (void)hipStreamSynchronize(kv.first.second);
            (void)hipFree(kv.second.p);
        }
    g8_slabs.clear();
    (void)hipGetLastError();
    return 0;
}

static int g8_cus() {          // per device (a process may drive several GPUs)
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

template <bool CV>
static void launch_8p(const GP& pv, int grid, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_8p_kernel<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipLaunchKernelGGL((gemm_bf16_8p_kernel<CV>), dim3(grid), dim3(512), LDS_TOTAL, stream, pv);
}

// mode: 0 never, 1 whenever legal, 2 heuristic (default); A3T_GEMM_8P or a3t_gemm_8p_mode()
static int g_8p_mode = -1;
static int g8_mode() {
    if (g_8p_mode < 0) {
        const char* e = getenv("A3T_GEMM_8P");
        g_8p_mode = e ? atoi(e) : 2;
    }
    return g_8p_mode;
}
extern "C" int a3t_gemm_8p_mode(int mode) {
    const int old = g8_mode();
    g_8p_mode = mode;
    return old;
}
extern "C" int64_t a3t_gemm_keep_bytes(int M, int N) { return (int64_t)((M + 255) / 256) * ((N + 255) / 256) * 8192; }

// Is the 8-phase kernel the one a3t_gemm picks for this problem?  (The engine asks before it chooses the keep-bit
// protocol: both GEMMs of a pair must agree.)
static bool g8_applicable(const GP& p, int batch, int ly, bool keep) {
    const int mode = g8_mode();
    if (mode == 0 || ly != 0 || batch != 1 || p.splitk != 1 || p.accumulate != A3T_ACC_STORE) return false;
    if (p.K % 128 != 0 || p.N % 8 != 0 || p.c_rs % 8 != 0 || p.a_cs != 1 || p.b_cs != 1) return false;
    if (p.S || p.kshift_mode) return false;
    if (p.R && (((uintptr_t)p.R & 15) || p.keep_out)) return false;
    if (p.colsum && p.colsum_slots > 1) return false;
    if (p.act != A3T_ACT_NONE && p.act != A3T_ACT_RELU) return false;
    if (p.taps > 1 && (p.Kc % 64 != 0 || p.b_ts != p.Kc || p.Tseq <= 0)) return false;
    if (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) return false;
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    const int64_t a_bytes = ((int64_t)p.M * p.a_rs) * 2, b_bytes = ((int64_t)p.N * p.b_rs) * 2;
    if (a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31) || (int64_t)p.M * p.c_rs >= (1ll << 32)) return false;
    if (keep && (p.N % 256 != 0)) return false;
    if (mode == 2) {
        // Cost model fitted on MI355X (round 3: profiles/r03_g8_check.txt, r03_g8_c4_shapes.txt): one 128-KiB workgroup per CU runs its K loop
        // at ~1.65 us per 64-wide K-tile (1.3 PFLOP/s) but nothing overlaps a tile's fixed costs -- pipeline refill and
        // quadrant epilogues ~7 us, bias/activation 1.5, dropout hashes 5, keep bits / fp32 + residual 2 each -- nor the partially
        // filled last round of the grid; the 128x128 kernel (4 workgroups per CU, epilogues hidden behind its neighbours)
        // sustains ~780 TFLOP/s on the same problems.  It wins for long K and grids that fill their rounds: configs[3]'s
        // d=512 / ff=2048 FFN (+8 % and +28 %), not configs[1]'s N=1536, K=1152 convs (4 rounds for 3.28, 18 K-tiles: -8 %).
        const long tm = (p.M + 255) / 256, tn = (p.N + 255) / 256, tiles = tm * tn;
        const int cus = g8_cus();
        const double rounds = (double)((tiles + cus - 1) / cus);
        double fixed = 7.0;
        if (p.bias || p.act != A3T_ACT_NONE) fixed += 1.5;
        if (p.drop_inv > 0.f) fixed += 5.0;
        if (p.keep_out || p.keep_in) fixed += 2.0;
        if (p.R || p.c_dtype == A3T_F32) fixed += 2.0;
        if (p.colsum) fixed += 2.0;
        const double t8 = rounds * ((p.K / 64) * 1.65 + fixed);
        const double t128 = 2.0 * p.M * p.N * (double)p.K / 780e6;     // us
        if (tiles < cus / 2 || t8 > 0.95 * t128) return false;
    }
    return true;
}

// flags: 1 bias / activation, 2 dropout, 4 keep bits out, 8 keep bits in, 16 fp32 output / residual, 32 column sums
extern "C" int a3t_gemm_8p_supported(int M, int N, int K, int taps, int flags) {
    static float dummy[4] __attribute__((aligned(16)));
    GP p = {};
    p.M = M, p.N = N, p.K = K, p.taps = taps < 1 ? 1 : taps, p.Kc = K / p.taps, p.b_ts = p.Kc;
    p.a_rs = p.Kc, p.a_cs = 1, p.b_rs = K, p.b_cs = 1, p.c_rs = N, p.splitk = 1, p.accumulate = A3T_ACC_STORE;
    p.Tseq = 1, p.colsum_slots = 1, p.c_dtype = (flags & 16) ? A3T_F32 : A3T_BF16;
    if (flags & 1) p.bias = dummy, p.act = A3T_ACT_RELU;
    if (flags & 2) p.drop_inv = 1.25f;
    if (flags & 4) p.keep_out = (unsigned char*)dummy;
    if (flags & 8) p.keep_in = (const unsigned char*)dummy;
    if (flags & 32) p.colsum = dummy;
    return g8_applicable(p, 1, 0, (flags & 12) != 0) ? 1 : 0;
}

template <bool WGF>
static void launch_8p_tn(const GP& pv, int grid, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_8p_tn_kernel<WGF>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES);
    hipLaunchKernelGGL((gemm_bf16_8p_tn_kernel<WGF>), dim3(grid), dim3(512), 2 * TILE_BYTES, stream, pv);
}

template <bool WGF>
static void launch_8p_tn3(const GP& pv, const TN3Group& grp, int grid, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_8p_tn3_kernel<WGF>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES);
    hipLaunchKernelGGL((gemm_bf16_8p_tn3_kernel<WGF>), dim3(grid), dim3(512), 2 * TILE_BYTES, stream, pv, grp);
}

// 128 x 384 tiles (gemm_bf16_8p_tn3_kernel): -1 = not applicable / not chosen.  A3T_GEMM_8P_TN3 = 0 never, 1 whenever legal,
// 2 (default) when the tiles fit the output exactly enough and there is enough K per workgroup.
static int g_tn3_mode = -1;
static int tn3_mode() {
    if (g_tn3_mode < 0) {
        const char* e = getenv("A3T_GEMM_8P_TN3");
        g_tn3_mode = e ? atoi(e) : 2;
    }
    return g_tn3_mode;
}
extern "C" int a3t_gemm_tn3_mode(int mode) {
    const int old = tn3_mode();
    g_tn3_mode = mode;
    return old;
}
static int gemm_8p_tn3(const GP& p, hipStream_t stream, int64_t a_bytes, int64_t b_bytes) {
    const int on = tn3_mode();
    if (on == 0) return -1;
    const long tm = (p.M + 127) / 128, tn = (p.N + 383) / 384, tiles = tm * tn;
    const int cus = g8_cus(), nkt = (p.K + 63) / 64;
    int splits = (int)(cus / tiles);
    if (splits < 1) splits = 1;
    if (splits > nkt / 16) splits = nkt / 16 > 0 ? nkt / 16 : 1;       // >= 16 K-tiles per workgroup
    if (on == 2) {
        const double fill = (double)p.M * p.N / ((double)tm * 128 * tn * 384);
        if (fill < 0.85 || tiles * splits < 96) return -1;
    }
    int per = (nkt + splits - 1) / splits;
    per += per & 1;
    GP pv = p;
    pv.tiles_n = (int)tn, pv.ntiles = (int)tiles, pv.splitk = splits;
    pv.a_bytes = (unsigned)a_bytes, pv.b_bytes = (unsigned)b_bytes;
    pv.slab = g8_slab(stream, (size_t)tiles * splits * 49152 * sizeof(float));
    if (!pv.slab) return (int)hipErrorOutOfMemory;
    const bool wg = p.taps > 1;
    const int grid = (int)(tiles * splits);
    TN3Group none = {};
    if (wg)
        launch_8p_tn3<true>(pv, none, grid, stream);
    else
        launch_8p_tn3<false>(pv, none, grid, stream);
    GP pf = pv;
    pf.splitk = (nkt + per - 1) / per;
    hipLaunchKernelGGL(gemm_8p_tn3_fold_kernel, dim3(48, (unsigned)tiles), dim3(256), 0, stream, pf, none);
    a3t_note_kernel("gemm_bf16_8p_tn3_kernel<%s>", wg ? "true" : "false");
    return (int)hipGetLastError();
}

// Several Linear weight gradients over the same tokens (dW_i[M_i][N_i] (+)= alpha_i * dy_i^T x_i, K tokens each) in ONE launch of
// the 128 x 384-tile kernel: their tiles share the K splits, so the four small gradients of a Conformer block (linear_out,
// linear_q/k/v, pointwise_conv1/2: 3 + 9 + 3 + 6 tiles) fill the chip with 47 K-tiles per workgroup instead of 16-20 each, and pay
// one prologue / slab / fold instead of four.  Returns -1 when a member does not fit (the caller launches them one by one).
extern "C" int a3t_gemm_tn3_group(const a3t_gemm_desc* d, int n, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || n < 1 || n > 8) return A3T_EINVAL;
    if (g8_mode() == 0 || tn3_mode() == 0) return -1;
    TN3Group grp = {};
    grp.n = n;
    long tiles = 0;
    const int K = d[0].K;
    for (int i = 0; i < n; ++i) {
        const a3t_gemm_desc& e = d[i];
        if (!e.A || !e.B || !e.C || e.M <= 0 || e.N <= 0 || e.K != K) return A3T_EINVAL;
        if (e.compute != A3T_BF16 || e.a_dtype != A3T_BF16 || e.b_dtype != A3T_BF16 || e.c_dtype != A3T_F32) return -1;
        if (e.a_rs != 1 || e.b_rs != 1 || e.taps > 1 || e.batch > 1 || e.Tseq > 0 || e.kshift) return -1;
        if (e.bias || e.R || e.S || e.colsum || e.act != A3T_ACT_NONE || e.drop_p > 0.f || e.keep_in || e.keep_out) return -1;
        if (e.M % 8 || e.N % 8 || e.a_cs % 8 || e.b_cs % 8 || (((uintptr_t)e.A | (uintptr_t)e.B | (uintptr_t)e.C) & 15)) return -1;
        const int64_t ab = (int64_t)K * e.a_cs * 2, bb = (int64_t)K * e.b_cs * 2;
        if (ab >= (1ll << 31) || bb >= (1ll << 31)) return -1;
        TN3Prob& q = grp.q[i];
        q.A = e.A, q.B = e.B, q.C = (float*)e.C, q.c_rs = e.c_rs, q.M = e.M, q.N = e.N;
        q.a_csb = (unsigned)(e.a_cs * 2), q.b_csb = (unsigned)(e.b_cs * 2), q.a_bytes = (unsigned)ab, q.b_bytes = (unsigned)bb;
        q.alpha = e.alpha, q.accumulate = e.accumulate == A3T_ACC_SOLE ? A3T_ACC_ADD : e.accumulate;
        q.tile0 = (int)tiles, q.tiles_n = (e.N + 383) / 384;
        tiles += (long)((e.M + 127) / 128) * q.tiles_n;
    }
    // A3T_ACC_SOLE members are folded with plain read-modify-writes: two members that write overlapping ranges of one gradient (a
    // tied weight) would race inside the one fold launch -> the caller launches them one by one (ordered on the stream)
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const char* ci = (const char*)d[i].C, *cj = (const char*)d[j].C;
            const char* ei = ci + ((int64_t)(d[i].M - 1) * d[i].c_rs + d[i].N) * 4, *ej = cj + ((int64_t)(d[j].M - 1) * d[j].c_rs + d[j].N) * 4;
            if (ci < ej && cj < ei) return -1;
        }
    if (tn3_mode() == 2) {       // partly empty 384-column tiles (input widths that are no multiple of 384) lose to the single launches
        double out = 0.0;
        for (int i = 0; i < n; ++i) out += (double)d[i].M * d[i].N;
        if (out / ((double)tiles * 128 * 384) < 0.85) return -1;
    }
    const int cus = g8_cus(), nkt = (K + 63) / 64;
    int splits = (int)(cus / tiles);
    if (splits < 1) splits = 1;
    if (splits > nkt / 16) splits = nkt / 16 > 0 ? nkt / 16 : 1;
    int per = (nkt + splits - 1) / splits;
    per += per & 1;
    GP pv = {};
    pv.K = K, pv.taps = 1, pv.Tseq = 1, pv.ntiles = (int)tiles, pv.splitk = splits, pv.tiles_n = 1;
    pv.slab = g8_slab(stream, (size_t)tiles * splits * 49152 * sizeof(float));
    if (!pv.slab) return (int)hipErrorOutOfMemory;
    launch_8p_tn3<false>(pv, grp, (int)(tiles * splits), stream);
    GP pf = pv;
    pf.splitk = (nkt + per - 1) / per;
    hipLaunchKernelGGL(gemm_8p_tn3_fold_kernel, dim3(48, (unsigned)tiles), dim3(256), 0, stream, pf, grp);
    a3t_note_kernel("gemm_bf16_8p_tn3_kernel<false>");
    return (int)hipGetLastError();
}

// weight gradients: reduction-strided operands, K (tokens) split over workgroups
static int gemm_8p_tn(const GP& p, int batch, hipStream_t stream) {
    // Its K loop runs 1.55 us per K-tile (1.39 PFLOP/s) but the ~240 workgroups of a split-K grid finish together and their
    // 15.7 M fp32 atomics cost 20-35 us with nothing to hide them behind, and a 128-KiB / 496-register workgroup shares its CU
    // with nobody (the 128x128 weight-gradient kernel runs beside the main stream's kernels).  configs[1]'s FFN weight
    // gradients: 153 / 165 us against 170 / 171 us alone, +1 ms per step inside the step; configs[3]'s (K = 28800, 90 K-tiles per
    // workgroup): -1 ms per step.  Hence the margin below.  A3T_GEMM_8P_TN=0 / 1: never / whenever legal.
    static int tn_on = -1;
    if (tn_on < 0) {
        const char* e = getenv("A3T_GEMM_8P_TN");
        tn_on = e ? atoi(e) : 2;
    }
    const int mode = g8_mode();
    if (mode == 0 || tn_on == 0) return -1;
    if (batch != 1 || p.c_dtype != A3T_F32 || p.M % 8 != 0 || p.N % 8 != 0) return -1;
    if (p.a_rs != 1 || p.b_rs != 1 || p.bias || p.R || p.S || p.colsum || p.act != A3T_ACT_NONE || p.drop_inv > 0.f) return -1;
    const bool wg = p.taps > 1;
    if (wg && ((p.N % p.taps) || ((p.N / p.taps) % 128) || p.Tseq <= 0 || p.Tseq >= 32768)) return -1;
    if (!wg && p.kshift_mode) return -1;
    if (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) return -1;
    const int64_t a_bytes = (int64_t)p.K * p.a_cs * 2, b_bytes = (int64_t)p.K * p.b_cs * 2;
    if (a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31) || p.a_cs % 8 || p.b_cs % 8) return -1;
    const long tm = (p.M + 255) / 256, tn = (p.N + 255) / 256, tiles = tm * tn;
    const int cus = g8_cus(), nkt = (p.K + 63) / 64;
    int splits = (int)(cus / tiles);
    if (splits < 1) splits = 1;
    if (splits > nkt / 16) splits = nkt / 16 > 0 ? nkt / 16 : 1;       // >= 16 K-tiles per workgroup
    // Which tile: 256 x 256 (four quadrants per K-tile, 1.55 us: the better K loop) when it covers the output exactly --
    // configs[3]'s 2048 x 1536 / 512 x 6144: 67.97 ms per step against 68.70 on the 128 x 384 tile -- and 128 x 384 (three
    // quadrants, 1.41 us, 176 registers: leaves the CU's other wave slots to the main stream's kernels) where 256 x 256 tiles
    // would be partly empty -- configs[1]'s 1536 x 1152 / 384 x 4608 (fill 0.90 / 0.75): 43.2 ms per step against 44.6.
    const double fill = (double)p.M * p.N / ((double)tm * 256 * tn * 256);
    bool ok22 = true;
    if (mode == 2 && tn_on == 2) {
        const double t8 = (double)((nkt + splits - 1) / splits) * 1.55 + 25.0;      // us: K loop + prologue, partial stores, fold
        const double t128 = 2.0 * p.M * p.N * (double)p.K / 680e6;                 // us at the 128x128 kernel's ~680 TFLOP/s
        ok22 = !(fill < 0.7 || tiles * splits < 160 || nkt / splits < 32 || t8 > 0.7 * t128);
    }
    const int t3 = tn3_mode();
    if (t3 == 1 || (t3 == 2 && !(ok22 && fill >= 0.95))) {
        const int r3 = gemm_8p_tn3(p, stream, a_bytes, b_bytes);
        if (r3 != -1) return r3;
    }
    if (!ok22) return -1;
    GP pv = p;
    pv.tiles_n = (int)tn, pv.ntiles = (int)tiles, pv.splitk = splits;
    pv.a_bytes = (unsigned)a_bytes, pv.b_bytes = (unsigned)b_bytes;
    pv.slab = nullptr;
    // K splits always leave through the slab + fold (the fp32-atomic epilogue of rounds 3-4 lost by 21 us per launch and left in
    // round 6; a single split writes C directly in the mode the descriptor asks for)
    if (splits > 1) {
        pv.slab = g8_slab(stream, (size_t)tiles * splits * 65536 * sizeof(float));
        if (!pv.slab) return (int)hipErrorOutOfMemory;
    }
    const int grid = (int)(tiles * splits);
    if (wg)
        launch_8p_tn<true>(pv, grid, stream);
    else
        launch_8p_tn<false>(pv, grid, stream);
    if (pv.slab) {
        int per = (nkt + splits - 1) / splits;       // as the kernel computes it: splits past the last K-tile write nothing
        per += per & 1;
        GP pf = pv;
        pf.splitk = (nkt + per - 1) / per;
        hipLaunchKernelGGL(gemm_8p_tn_fold_kernel, dim3(64, (unsigned)tiles), dim3(256), 0, stream, pf);
    }
    a3t_note_kernel("gemm_bf16_8p_tn_kernel<%s>", wg ? "true" : "false");
    return (int)hipGetLastError();
}

// Called by a3t_gemm_bf16_glds after the alignment contract has been checked.  Returns -1 when not applicable.
int a3t_gemm_bf16_8p(const GP& p, int batch, int ly, hipStream_t stream) {
    if (ly == 2) return (p.keep_in || p.keep_out) ? A3T_EINVAL : gemm_8p_tn(p, batch, stream);
    const bool keep = p.keep_in || p.keep_out;
    if (!g8_applicable(p, batch, ly, keep)) return keep ? A3T_EINVAL : -1;
    GP pv = p;
    const long tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
    pv.tiles_n = (int)tn;
    pv.ntiles = (int)(tm * tn);
    pv.a_bytes = (unsigned)(((int64_t)p.M * p.a_rs) * 2);
    pv.b_bytes = (unsigned)(((int64_t)p.N * p.b_rs) * 2);
    const int grid = (int)(pv.ntiles < g8_cus() ? pv.ntiles : g8_cus());
    const bool conv = p.taps > 1;
    if (conv)
        launch_8p<true>(pv, grid, stream);
    else
        launch_8p<false>(pv, grid, stream);
    a3t_note_kernel("gemm_bf16_8p_kernel<%s>", conv ? "true" : "false");
    return (int)hipGetLastError();
}
