// gemm_bf16_tt.hip -- batched bf16 MFMA GEMM for gfx950 whose A operand is a score-sized (T x T) matrix per batch element and
// whose B operand / output are T x d_k slices: the products of the attention backward that the fused kernels leave as GEMMs,
//   dQu = dS K, dQv = dBD P             (A = [m][k], k-contiguous:  "NN")
//   dV = P^T dctx, dK = dS^T (q + u)    (A = [k][m], m-contiguous:  "TN")        (espnet attention.py:64-96,145-209)
// with B = [k][n], n-contiguous, N = d_k <= 192.
//
// These launches do 2 T^2 d_k flops per T^2 bf16 elements read exactly once (96 flop / byte at d_k = 192): whatever the tile, they
// stream 160 MB of score-sized operand per launch (configs[1]) and are bound by how many of those bytes a CU keeps in flight, not
// by the MFMA pipe.  The 128 x 192 tiles of gemm_bf16.hip (one LDS buffer, three workgroups per CU) expose the whole DMA latency
// once per K-tile and per workgroup: 2.1-2.3 TB/s.  This kernel is built around the stream instead:
//   * one workgroup per CU owns TR <= 320 rows x ALL N columns of one batch element (T = 1120: four tiles of 288 rows per
//     (b, h), 64 x 4 = 256 workgroups = one round of the chip); A is read once, B (430 KB per batch element) comes from L2;
//   * K-tiles of 32, FOUR LDS stages of {A: 20 KiB, B: 12 KiB}: three K-tiles (~54 KB of score bytes per CU, ~14 MB on the chip)
//     are in flight while the fourth is multiplied;
//   * LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction): every wave issues exactly four per K-tile (waves 0-4 the
//     20 A pieces, waves 5-7 the 12 B pieces; pieces past the tile / N / K read zeros through voffset = 0x80000000), so one counted
//     wait -- vmcnt(8) -- is right for every wave; one s_barrier per K-tile: [vmcnt(8) | s_barrier | issue K-tile t+3 | fragments
//     + 30 MFMA of K-tile t].  The stage K-tile t+3 overwrites was read in iteration t-1, whose MFMAs (hence fragment reads) every
//     wave has issued before it reaches the barrier;
//   * 8 waves = 2 (wr) x 4 (wc); wave tile (TR / 2) x (16 NJ) = up to 10 x 3 blocks of v_mfma_f32_16x16x32_bf16;
//   * LDS images: [m][k] rows of 64 B read with ds_read_b128, chunk position g ^ h(row >> 2); [k][m] / [k][n] operands in panels
//     of 64 columns (k-rows of 128 B) read with ds_read_b64_tr_b16, chunk position c ^ 2 ((r >> 1 & 1) | (r >> 3 & 1) << 1): the
//     eight 32-byte pieces a half-wave reads lie on distinct banks.  The swizzle is applied to the per-lane SOURCE address;
//   * the MFMAs form C^T blocks, so a lane ends up with four consecutive columns of one row: the epilogue (alpha -> store bf16 or
//     add to the stored bf16 in fp32, fused column sums) goes straight from the accumulators to 8-byte stores -- no LDS pass
//     (the 128 x 128 kernel's staged epilogue took 11 us of this kernel's 46).  Nothing else is supported (host contract).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/a3t_hip.h"
#include "gemm_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
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

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "buffer_load_dwordx4 ... lds (16-byte LDS-DMA) exists on gfx950 only: build with --offload-arch=gfx950"
#endif

namespace {
constexpr int TT_MAX_ROWS = 320, TT_BK = 32, TT_STAGES = 4;
constexpr int TT_A_BYTES = TT_MAX_ROWS * TT_BK * 2, TT_B_BYTES = 192 * TT_BK * 2, TT_STAGE = TT_A_BYTES + TT_B_BYTES;   // 20 + 12 KiB
constexpr int TT_LDS = TT_STAGES * TT_STAGE;                                                                            // 128 KiB
constexpr unsigned OOB = 0x80000000u;         // voffset beyond every descriptor (operands of one batch element < 2 GiB)

// (asm: in front of __builtin_amdgcn_ds_read_tr16_b64 hipcc drains every LDS-DMA it has seen issued through the builtin;
//  see gemm_bf16_8p.hip)
__device__ __forceinline__ void tt_dma16(const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ float tt_row16_sum(float v) {   // sum over the 16 lanes of a DPP row, result in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}
}   // namespace

// -DTT_TIMING (probe build): wall-clock stamps (100 MHz) per workgroup: start, first barrier passed, K loop done, epilogue done
#ifdef TT_TIMING
__device__ unsigned long long tt_stamps[1024 * 4];
extern "C" int a3t_debug_read_tt(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tt_stamps), bytes); }
#define TT_STAMP(k) do { if (tid == 0 && blockIdx.x < 1024) tt_stamps[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#else
#define TT_STAMP(k)
#endif

// DUAL ([m][k] operand only): C = alpha (A B + A2 B2) in ONE launch -- dq = dS K + dBD P of the attention backward (attention.py:190-203
// on its way back): the K loop walks the K-tiles of the first product, then those of the second (A2 has A's strides, B2 its own);
// the column sums of the first product are taken off the accumulators at the hand-over, those of the second are the rest.
template <bool ATN, int NJ, bool ASGN = false, bool DUAL = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tt_kernel(GP p, int TR) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    TT_STAMP(0);
    int wi = blockIdx.x;
    {   // workgroup b runs on XCD b % 8: every XCD gets a contiguous run of (batch element, row tile) -- the row tiles of a
        // batch element read the same B from one L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wi & 7;
        wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wi >> 3);
    }
    const int tm = wi % p.ntiles, bz = wi / p.ntiles;
    const int z0 = bz / p.batch_inner, z1 = bz % p.batch_inner;
    const u16* A = (const u16*)p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const u16* B = (const u16*)p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int64_t zoff = z0 * p.c_bs0 + z1 * p.c_bs1;
    const int m0 = tm * TR;
    const int HR = TR >> 1, nbi = HR >> 4;        // rows / 16-row blocks of a wave row
    const int nkt = (p.K + TT_BK - 1) / TT_BK;

    const unsigned a_ld = (unsigned)(ATN ? p.a_cs : p.a_rs) * 2u, b_ld = (unsigned)p.b_cs * 2u;     // bytes between rows of the operand as stored
    const unsigned a_ext = ATN ? (unsigned)(p.K - 1) * a_ld + (unsigned)p.M * 2u : (unsigned)(p.M - 1) * a_ld + (unsigned)p.K * 2u;
    const unsigned b_ext = (unsigned)(p.K - 1) * b_ld + (unsigned)p.N * 2u;
    const bool is_a = w < 5;        // waves 0-4 request the A pieces, waves 5-7 the B pieces
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(is_a ? A : B), 0, (int)(is_a ? a_ext : b_ext), 0x00020000);
    const unsigned b2_ld = DUAL ? (unsigned)p.b2_cs * 2u : 0u;
    const void* X2 = DUAL ? (is_a ? (const void*)((const u16*)p.A2 + z0 * p.a_bs0 + z1 * p.a_bs1) : (const void*)((const u16*)p.B2 + z0 * p.b2_bs0 + z1 * p.b2_bs1)) : (const void*)A;
    const unsigned a2_ld = (DUAL && p.a2_rs) ? (unsigned)p.a2_rs * 2u : a_ld;      // (A2 may be a view with a row stride of its own)
    const __amdgpu_buffer_rsrc_t rX2 = __builtin_amdgcn_make_buffer_rsrc((void*)X2, 0, (int)(is_a ? (unsigned)(p.M - 1) * a2_ld + (unsigned)p.K * 2u : (unsigned)(p.K - 1) * b2_ld + (unsigned)p.N * 2u), 0x00020000);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_AS(smem));

    // ---- DMA lane geometry.  Slot s = 4 w + q: s < 20 -> A piece s, else B piece s - 20.
    //  [m][k] image: piece s = tile rows 16 s .. +15 (64 B each); lane -> row 16 s + (lane >> 2), position lane & 3 holds source
    //      chunk (lane & 3) ^ h((lane >> 4) & 3), h(q) = (4 - q) & 3
    //  [k][x] image: piece s = panel s >> 2 (64 columns), k-rows 8 (s & 3) .. +7 (128 B each); lane -> k-row 8 (s & 3) + (lane >> 3),
    //      position lane & 7 holds source chunk (lane & 7) ^ 2 sw(k-row), sw(r) = (r >> 1 & 1) | (r >> 3 & 1) << 1
    unsigned voff[4];       // byte offset inside the batch element's operand at K-tile 0 (OOB: never valid)
    unsigned voff2[4];      // DUAL: the same for the second product (differs for the B waves: B2's row stride)
    int kq[4];              // first k this lane's 16 bytes hold (NN: the chunk's k; TN / B: the k-row), for the K tail
    const unsigned ld = is_a ? a_ld : b_ld;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (is_a && !ATN) {
            const int R = (w * 4 + q) * 16 + (lane >> 2);
            const int g = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
            const bool ok = R < TR && m0 + R < p.M;
            voff[q] = ok ? (unsigned)(m0 + R) * a_ld + (unsigned)g * 16u : OOB;
            voff2[q] = ok ? (unsigned)(m0 + R) * a2_ld + (unsigned)g * 16u : OOB;
            kq[q] = g * 8;
        } else {
            const int s = is_a ? w * 4 + q : (w - 5) * 4 + q;
            const int r = (s & 3) * 8 + (lane >> 3);
            const int sw = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
            const int col = (s >> 2) * 64 + (((lane & 7) ^ (sw << 1)) << 3);
            const bool ok = is_a ? (col < TR && m0 + col < p.M) : (col < p.N && col < 64 * NJ);
            voff[q] = ok ? (unsigned)r * ld + (unsigned)((is_a ? m0 : 0) + col) * 2u : OOB;
            voff2[q] = ok ? (unsigned)r * (is_a ? ld : b2_ld) + (unsigned)((is_a ? m0 : 0) + col) * 2u : OOB;
            kq[q] = r;
        }
    }
    const unsigned kstep = (is_a && !ATN) ? 64u : 32u * ld;       // bytes one K-tile advances the lane's source
    const unsigned kstep2 = (is_a && !ATN) ? 64u : 32u * (is_a ? ld : b2_ld);
    const unsigned dst0 = lds0 + (is_a ? (unsigned)(w * 4) * 1024u : (unsigned)TT_A_BYTES + (unsigned)((w - 5) * 4) * 1024u);
    auto issue = [&](const int kt) __attribute__((always_inline)) {
        const unsigned dst = dst0 + (unsigned)(kt & 3) * (unsigned)TT_STAGE;
        if (DUAL && kt >= nkt) {                // the second product's K-tiles
            const int k2 = kt - nkt;
            const unsigned so = (unsigned)k2 * kstep2;
            const int krem = p.K - k2 * TT_BK;
#pragma unroll
            for (int q = 0; q < 4; ++q) tt_dma16(rX2, dst + q * 1024, kq[q] < krem ? voff2[q] : OOB, so);
            return;
        }
        const unsigned so = (unsigned)kt * kstep;
        const int krem = p.K - kt * TT_BK;      // <= 0: a K-tile past the end (zeros)
#pragma unroll
        for (int q = 0; q < 4; ++q) tt_dma16(rX, dst + q * 1024, kq[q] < krem ? voff[q] : OOB, so);
    };

    // ---- fragment geometry: lane (g, pp)
    const int g = lane >> 4, pp = lane & 15;
    //  [m][k]: row pp of a block, chunk position g ^ h(pp >> 2)
    const unsigned offkc = (unsigned)pp * 64u + (unsigned)((g ^ ((4 - (pp >> 2)) & 3)) << 4);
    //  [k][x]: k-rows 8 g + (pp >> 2) (+ 4), chunk cb + (pp >> 1 & 1), 8 bytes (pp & 1)
    const unsigned swr = (unsigned)((((pp >> 3) & 1) | ((g & 1) << 1)) << 1);
    const unsigned kbyte = (unsigned)(g * 8 + (pp >> 2)) * 128u + (unsigned)(pp & 1) * 8u;
    auto frag_rc = [&](const unsigned char* img, const int c0) __attribute__((always_inline)) -> bf16x8 {     // 16 columns from c0
        const unsigned cb = (unsigned)((c0 & 63) >> 3) + (unsigned)((pp >> 1) & 1);
        const unsigned char* a0 = img + (c0 >> 6) * 4096 + kbyte + ((cb ^ swr) << 4);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0 + 512));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x4 acc[10][NJ];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0), issue(1), issue(2);
    const int rb0 = wr * nbi;       // first 16-row block of this wave row
    const int nkt_all = DUAL ? 2 * nkt : nkt;
    float cs1[NJ][4];               // DUAL: this lane's column partial sums of the FIRST product
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs1[j][r] = 0.f;
    for (int kt = 0; kt < nkt_all; ++kt) {
        WAIT_VM(8);
        BAR();
#ifdef TT_TIMING
        if (kt == 0) TT_STAMP(1);
#endif
        issue(kt + 3);
        const unsigned char* sA = smem + (kt & 3) * TT_STAGE;
        const unsigned char* sB = sA + TT_A_BYTES;
        bf16x8 fb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = frag_rc(sB, wc * (16 * NJ) + j * 16);
        // (all ten blocks, unconditionally: blocks past this wave row's nbi hold the other wave row's rows or the zeros of the
        //  pieces past the tile -- their accumulators are never stored -- and the loop stays free of branches)
        bf16x8 fa[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if (ATN)
                fa[i] = ASGN ? a3t_sign_floor(frag_rc(sA, (rb0 + i) * 16), 0) : frag_rc(sA, (rb0 + i) * 16);     // (a3t_gemm_desc::a_signmask)
            else
                fa[i] = *(const bf16x8*)(sA + (rb0 + i) * 1024 + offkc);
        }
        SB();       // (all reads issued before the first MFMA: the fragments keep registers of their own and return behind the MFMAs)
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if (DUAL && kt == nkt - 1 && p.colsum) {      // hand-over: the accumulators hold the first product alone
#pragma unroll
            for (int i = 0; i < 10; ++i)
                if (i < nbi) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) cs1[j][r] += acc[i][j][r];
                }
        }
    }
    WAIT_VM(0);
    __syncthreads();
    TT_STAMP(2);

    // ---- epilogue.  The products were formed as C^T blocks (B fragment as the MFMA's first operand): lane (g, pp) holds, for row
    // rb0*16 + i*16 + pp of the tile, the FOUR consecutive columns wc*16*NJ + j*16 + g*4 .. +3 -- one 8-byte bf16 store per block
    // straight from the accumulators (a wave instruction = 16 rows x 32 B; the three blocks of a wave column complete 96
    // contiguous bytes per row in L2), A3T_ACC_ADD reads the 8 bytes first (fp32 sum, one rounding).  No LDS, no waits between
    // the blocks: all loads of a wave row are in flight together.
    const int crow = m0 + rb0 * 16 + pp;
    const int ccol = wc * (16 * NJ) + g * 4;
    u16* Cb = (u16*)p.C + zoff + ccol;
    float cs[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
    const bool add = p.accumulate != A3T_ACC_STORE;
    const float alpha = p.alpha;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int row = crow + i * 16;
        const bool rok = i < nbi && row < p.M;      // (no break: the accumulators must stay statically indexed)
        uint2 old[NJ];
        if (add) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                old[j] = (rok && ccol + j * 16 < p.N) ? *(const uint2*)(Cb + (int64_t)row * p.c_rs + j * 16) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!rok || ccol + j * 16 >= p.N) continue;
            float v0 = acc[i][j][0] * alpha, v1 = acc[i][j][1] * alpha, v2 = acc[i][j][2] * alpha, v3 = acc[i][j][3] * alpha;
            cs[j][0] += v0, cs[j][1] += v1, cs[j][2] += v2, cs[j][3] += v3;
            if (add) v0 += bf2f(old[j].x & 0xffff), v1 += bf2f(old[j].x >> 16), v2 += bf2f(old[j].y & 0xffff), v3 += bf2f(old[j].y >> 16);
            uint2 o;
            o.x = io_pack2(v0, v1), o.y = io_pack2(v2, v3);
            *(uint2*)(Cb + (int64_t)row * p.c_rs + j * 16) = o;
        }
    }
    if (p.colsum) {     // column sums of what was stored (the increments): over the 16 rows of a DPP row, then one atomic per column
        float* o = p.colsum + z1 * p.colsum_bs1 + ccol;
        if (p.colsum_slots > 1) o += (int64_t)((tm + z0) % p.colsum_slots) * p.colsum_ss;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = tt_row16_sum(cs[j][r]);
                if (DUAL) {     // first product's sums from the hand-over, second = the rest
                    const float t1 = tt_row16_sum(cs1[j][r] * alpha);
                    if (pp == 0 && ccol + j * 16 < p.N) {
                        atomicAdd(o + j * 16 + r, p.colsum_scale * t1);
                        atomicAdd(o + (p.colsum2 - p.colsum) + j * 16 + r, p.colsum_scale * (t - t1));
                    }
                } else if (pp == 0 && ccol + j * 16 < p.N) {
                    atomicAdd(o + j * 16 + r, p.colsum_scale * t);
                }
            }
    }
#ifdef TT_TIMING
    __syncthreads();
    TT_STAMP(3);
#endif
}

// mode: 0 never, 1 whenever legal, 2 (default) when the grid fills the chip; A3T_GEMM_TT or a3t_gemm_tt_mode()
static int g_tt_mode = -1;
static int tt_mode() {
    if (g_tt_mode < 0) {
        const char* e = getenv("A3T_GEMM_TT");
        g_tt_mode = e ? atoi(e) : 2;
    }
    return g_tt_mode;
}
extern "C" int a3t_gemm_tt_mode(int mode) {
    const int old = tt_mode();
    g_tt_mode = mode;
    return old;
}

static int tt_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return n;
}

template <bool ATN, int NJ, bool ASGN = false, bool DUAL = false>
static void launch_tt(const GP& pv, int TR, int grid, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tt_kernel<ATN, NJ, ASGN, DUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, TT_LDS);
    hipLaunchKernelGGL((gemm_bf16_tt_kernel<ATN, NJ, ASGN, DUAL>), dim3(grid), dim3(512), TT_LDS, stream, pv, TR);
}

// Row tiling and (mode 2) the cost model of the streaming kernel for M x N outputs over a reduction of K per batch element.
// Row tiling: the fewest workgroup-rounds x rows per tile.  A workgroup streams its rows' share of A whatever happens beside
// it, so the launch takes rounds(tiles x batch) x (TR + ~64 rows' worth of fill and epilogue): configs[1] (M = 1120, 64 batch
// elements) -> 4 tiles of 288 rows = 256 workgroups = one round; configs[3] (M = 1800) -> 8 tiles of 256 rows = two full rounds
// (6 tiles of 320 would be 1.5 rounds for the price of two).  A last tile with a few rows is cheap (its A pieces are zeros).
static bool tt_tiling(int M, int N, int K, int batch, int mode, int& tiles, int& TR) {
    const int cus = tt_cus();
    tiles = 0, TR = 0;
    long best = 0;
    for (int t = (M + TT_MAX_ROWS - 1) / TT_MAX_ROWS, n = 0; n < 6; ++t, ++n) {
        const int tr = (((M + t - 1) / t) + 31) / 32 * 32;
        if (tr > TT_MAX_ROWS || (long)tr * (t - 1) >= M) continue;       // (an empty last tile: t is not a tiling of its own)
        const long cost = (((long)t * batch + cus - 1) / cus) * (tr + 64);
        if (!tiles || cost < best) tiles = t, TR = tr, best = cost;
    }
    if (!tiles) return false;
    const long units = (long)tiles * batch;
    if (mode == 2) {
        // long reductions over a score-sized operand on ONE well-filled round of the chip (configs[1]).  Two rounds pay the first
        // tile's latency and the epilogue twice with nothing beside them on the CU: configs[3] (M = 1800, d_k = 128: 512
        // workgroups) runs these products in 135 / 155 / 110 us against 136 / 141 / 117 on the 128-row kernel and its step
        // 0.3-0.4 ms slower (profiles/r06_tt_gemm.txt)
        if (K < 512 || M < 256 || N < 96 || units > cus || units * TR * 4 > (long)M * batch * 5 || units * 10 < cus * 7) return false;
    }
    return true;
}

// 1 when a3t_gemm runs the batched bf16 product (M x N per batch element, reduction K, A score-sized [m][k], B [k][n]) on the
// streaming kernel under the current mode -- the engine asks before it hands it TWO products for one launch (a3t_gemm_desc::A2)
extern "C" int a3t_gemm_tt_supported(int M, int N, int K, int batch) {
    int tiles, TR;
    const int mode = tt_mode();
    if (mode == 0 || N > 192 || N % 8 || M % 8 || K % 8) return 0;
    return tt_tiling(M, N, K, batch, mode, tiles, TR) ? 1 : 0;
}

// Called by a3t_gemm_bf16_glds after the alignment contract has been checked (ly: 1 = NN, 2 = TN; B is [k][n], n-contiguous).
// Returns -1 when not applicable.
int a3t_gemm_bf16_tt(const GP& p, int batch, int ly, hipStream_t stream) {
    const int mode = tt_mode();
    if (mode == 0 || (ly != 1 && ly != 2) || p.splitk != 1 || p.accumulate == A3T_ACC_ATOMIC) return -1;
    if (p.a_signmask && ly != 2) return -1;
    if (p.A2) {
        const bool v2 = (p.a_unaligned & 2) != 0;       // A2: a 2-byte aligned view with any row stride
        const int64_t a2 = p.a2_rs ? p.a2_rs : p.a_rs;
        if (ly != 1 || p.accumulate != A3T_ACC_STORE || !p.B2 || p.b2_cs % 8 || (p.b2_bs0 | p.b2_bs1) % 8 || ((uintptr_t)p.B2 & 15) ||
            ((uintptr_t)p.A2 & (v2 ? 1 : 15)) || (!v2 && a2 % 8) || a2 < p.K || a2 * 2 * (int64_t)p.M >= (1ll << 31) ||
            p.b2_cs * 2 * (int64_t)p.K >= (1ll << 31) || (p.colsum != nullptr) != (p.colsum2 != nullptr))
            return -1;
    }
    if (p.taps > 1 || p.kshift_mode || p.keep_in || p.keep_out || !p.epi_vec) return -1;
    if (p.b_rs != 1 || p.N > 192 || p.N % 8 || p.M % 8 || p.K % 8) return -1;
    if (ly == 1 ? (p.a_cs != 1 || p.a_rs % 8) : (p.a_rs != 1 || p.a_cs % 8)) return -1;
    if (p.b_cs % 8 || (p.a_bs0 | p.a_bs1 | p.b_bs0 | p.b_bs1) % 8) return -1;
    if (p.bias || p.R || p.S || p.act != A3T_ACT_NONE || p.drop_inv > 0.f || p.c_dtype != A3T_BF16) return -1;
    // one batch element's operands through 32-bit buffer offsets
    const int64_t a_ld = ly == 1 ? p.a_rs : p.a_cs;
    if (a_ld * 2 * (int64_t)(ly == 1 ? p.M : p.K) >= (1ll << 31) || p.b_cs * 2 * (int64_t)p.K >= (1ll << 31)) return -1;
    int tiles = 0, TR = 0;
    if (!tt_tiling(p.M, p.N, p.K, batch, mode, tiles, TR)) return -1;
    const long units = (long)tiles * batch;
    GP pv = p;
    pv.ntiles = tiles;
    const int nj = p.N <= 128 ? 2 : 3;
    if (ly == 2 && p.a_signmask) {
        if (nj == 3)
            launch_tt<true, 3, true>(pv, TR, (int)units, stream);
        else
            launch_tt<true, 2, true>(pv, TR, (int)units, stream);
    } else if (ly == 2) {
        if (nj == 3)
            launch_tt<true, 3>(pv, TR, (int)units, stream);
        else
            launch_tt<true, 2>(pv, TR, (int)units, stream);
    } else if (p.A2) {
        if (nj == 3)
            launch_tt<false, 3, false, true>(pv, TR, (int)units, stream);
        else
            launch_tt<false, 2, false, true>(pv, TR, (int)units, stream);
    } else {
        if (nj == 3)
            launch_tt<false, 3>(pv, TR, (int)units, stream);
        else
            launch_tt<false, 2>(pv, TR, (int)units, stream);
    }
    a3t_note_kernel("gemm_bf16_tt_kernel<%s, %d, %s, %s>", ly == 2 ? "true" : "false", nj, (ly == 2 && p.a_signmask) ? "true" : "false",
                    (ly == 1 && p.A2) ? "true" : "false");
    return (int)hipGetLastError();
}
