// attn_fused.hip -- fused legacy relative-position multi-head attention for gfx950 (bf16 MFMA, fp32 softmax).
//
// Reference semantics: LegacyRelPositionMultiHeadedAttention (espnet/nets/pytorch_backend/transformer/attention.py):
//   scores = ((q+u) k^T + rel_shift((q+v) p^T)) / sqrt(d_k)          :190-206
//   rel_shift (legacy, :145-165) on BD = (q+v) p^T (T x T):  j <= i -> BD[i][T-1-i+j];  j == i+1 -> 0;
//                                                            j >  i+1 -> BD[i+1][j-i-2]
//   attn = softmax(masked_fill(scores, ~keymask, min)) . masked_fill(~keymask, 0);  ctx = dropout(attn) v     :78-96
// No (T, T) tensor is written to HBM by the forward pass: scores, the shifted position term, probabilities and their
// dropout live in registers / LDS per (32-query x 32-key) tile; the forward keeps only ctx and one log-sum-exp per row.
//
// The band.  With x = j - i + T - 1 (0 <= x <= 2T-2) the shifted position term is ONE banded product
//     bd[i][j] = Qx[i] . Pext[x],   Pext[x] = P[x] (x < T) | 0 (x == T) | P[x-T-1] (x > T),
//                                   Qx[i]   = (q+v)[i] for x < T, (q+v)[i+1] for x > T
// so a (query block, key block) tile needs 63 consecutive band rows = two 32-row band blocks, and advancing the key
// block by 32 advances the band by exactly one block: every band block is computed ONCE per query block (12 MFMAs for
// d_k = 192, the same work as the un-shifted GEMM), written to a per-wave fp32 LDS scratch as [query][band column] and
// read back skewed ([query ii][31 - ii + key kk]) -- the closed-form rel_shift as an LDS address.  Query blocks start at
// multiples of 32 and T - xb is a multiple of 32 for every block base xb, so the zero row x == T is always row 0 of a
// block and a block uses either (q+v)[i] or (q+v)[i+1] as a whole.
//
// Everything is computed transposed (S^T = K Q^T, O^T = V^T P^T) so that a lane owns ONE query: softmax statistics are
// per-lane scalars, P^T leaves the MFMA accumulator layout already shaped like the B operand of the PV product (with a
// fixed permutation of the keys inside a 16-key step that the V^T fragment reads -- ds_read_b64_tr_b16 -- simply mirror).
//
// Workgroup = 4 waves x 32 queries; K / V tiles and a 5-slot ring of Pext blocks are shared through LDS (register
// staged: the next tiles' global loads are issued before the tile's compute and written to LDS after it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "../../include/a3t_hip.h"
#include "dtype_io.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))

struct AttnArgs {
    const u16* qu;        // (q + pos_bias_u)  [B*T][ldq], head h at column h*dk
    const u16* qv;        // (q + pos_bias_v)
    const float* bu;      // optional pos_bias_u / pos_bias_v [H*dk] (fp32): qu / qv then hold q itself and the kernel forms
    const float* bv;      // bf16(q + bias) as it loads its query fragments (= what a3t_add_pos_bias would have stored)
    const u16* k;         // keys   [B*T][ldkv]
    const u16* v;         // values [B*T][ldkv]
    const u16* pos;       // linear_pos(pos_emb) [T][ldp]
    const uint8_t* keymask;   // [B][T], 1 = valid key
    u16* ctx;             // out [B*T][ldo]
    float* lse;           // out [B][H][T]: log sum exp of the scaled scores (+inf for a fully masked row)
    u16* dbd;             // (A3T_ATTN_TIMING builds: where the per-stage cycle stamps go)
    int B, H, T;
    int64_t ldq, ldkv, ldp, ldo;
    float scale;
    unsigned int drop_thr, drop_key;
    float drop_inv;
    int use_redo;         // forward, 16-query kernel: only recompute blocks whose attn_redo flag is set
    // forward, training: un-normalised probabilities exp(s - m_ref) for the materialised backward (a3t_attn_fwd_train)
    u16* probs;           // [B][H][T][T]
    u16* pdrop;           // the same after attention dropout (NULL without dropout)
    float* rowscale;      // [B][H][T]: 1 / row sum, the factor that normalises a row of probs / pdrop
    // forward, key-split launch of the tail blocks (launch_fwd16): blocks item0 .. of the (b, h, query block) order, nparts
    // key ranges each; partial sums go to part_ws and attn_split_finish_kernel folds them
    int item0, nparts;
    float* part_ws;       // [items][nparts][128][dk] O, then [items][nparts][128] l, then [items][128] m2
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8 zero_frag() {
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, z);
}
// Out-of-range rows read this page instead of branching around the load: a conditional load costs a branch AND a
// vmcnt(0) at the join, which serialises every prefetch behind it.
static __device__ __attribute__((aligned(16))) unsigned int attn_zero_page[8];
// per 128-query block of the last a3t_attn_fwd launch: 1 = the fixed-reference kernel overflowed, recompute with the rescaling one
#ifndef A3T_SAVE_AUX
#define A3T_SAVE_AUX 0   // (nt = 2 measured: 348 us instead of 290 -- the L2 merges the 32-byte pieces of a line)
#endif
#if defined(A3T_SAVE_EXP) && A3T_SAVE_EXP == 1      // experiment: no store instruction at all
#define A3T_SAVE_STORE(d, r, vo) asm volatile("" ::"v"(d), "v"(vo))
#elif defined(A3T_SAVE_EXP) && A3T_SAVE_EXP == 2    // experiment: every store hits the same 1 KiB
#define A3T_SAVE_STORE(d, r, vo) __builtin_amdgcn_raw_buffer_store_b128(d, r, (vo) == 0x80000000u ? (vo) : (unsigned)(lane * 16), 0, 0)
#else
#define A3T_SAVE_STORE(d, r, vo) __builtin_amdgcn_raw_buffer_store_b128(d, r, vo, 0, A3T_SAVE_AUX)
#endif
static __device__ int attn_redo[1 << 16];
static void* g_attn_timing_buf = nullptr;      // a3t_attn_timing_buf(): where timing builds (-DA3T_ATTN_TIMING / -DA3T_DS_TIMING) file their stamps

// LDS-DMA piece (64 lanes x 16 B -> 1 KiB at lds_addr) as inline asm ON PURPOSE: for the builtin the compiler tracks "an LDS
// write is in flight" and, because every tile pointer is an offset into the one dynamic LDS array, puts a vmcnt wait for ALL
// outstanding DMA in front of the next LDS read that might alias -- i.e. in the middle of the iteration that issued it
// (measured: +80 us per launch).  The kernels below wait for their DMA themselves (counted vmcnt + s_barrier).
// (lds_addr: 32-bit LDS byte address, wave-uniform.  Callers form it as lds_base(smem) + offset: casting every tile pointer
//  from the generic address space makes the compiler emit a null check against src_shared_base that some instantiations
//  fail to select -- "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base".)
__device__ __forceinline__ unsigned lds_base(const void* smem0) { return (unsigned)(uintptr_t)LDS_AS(smem0); }
// (M0 is written and consumed inside the one statement.  It cannot be declared as a clobber -- hipcc: "inline asm clobber list
//  contains reserved registers: m0 ... may lead to undefined behaviour" -- so the kernels that call this use no other M0 consumer:
//  no LDS-DMA builtin, s_movrel, sendmsg or GWS; the gfx950 compiler does not keep values in M0 across statements on its own.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "buffer_load_dwordx4 ... lds (16-byte LDS-DMA) exists on gfx950 only: build with --offload-arch=gfx950"
#endif
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff) {
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff), "s"(r) : "memory");
}
__device__ __forceinline__ bf16x8 ld_frag_g(const u16* p, bool ok) {
    const u16* src = ok ? p : (const u16*)attn_zero_page;
    return *(const bf16x8*)src;
}
// query fragment with the positional bias added on the fly (attention.py:190-194: q + pos_bias_u / pos_bias_v): the bf16 q
// values plus 8 consecutive fp32 biases, rounded to bf16 exactly as a3t_add_pos_bias rounds; rows outside the utterance stay zero
// (bias8: this lane's 8 biases in LDS -- the workgroup stages its head's 2 x d_k biases there once: read from global memory
//  per fragment, the 36 dependent load pairs of a wave's prologue cost the 14 us per launch that a3t_add_pos_bias took.)
typedef const __attribute__((address_space(3))) float* lds_cfp;
__device__ __forceinline__ bf16x8 ld_frag_qb(const u16* p, bool ok, lds_cfp bias8, bool biased) {
    const bf16x8 f = ld_frag_g(p, ok);
    if (!biased) return f;                              // (wave-uniform)
    const uint4 u = __builtin_bit_cast(uint4, f);
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ b0 = *(const __attribute__((address_space(3))) v4f_*)bias8;
    const v4f_ b1 = *(const __attribute__((address_space(3))) v4f_*)(bias8 + 4);
    uint4 o;
    o.x = io_pack2(io_bf2f(u.x & 0xffff) + b0.x, io_bf2f(u.x >> 16) + b0.y);
    o.y = io_pack2(io_bf2f(u.y & 0xffff) + b0.z, io_bf2f(u.y >> 16) + b0.w);
    o.z = io_pack2(io_bf2f(u.z & 0xffff) + b1.x, io_bf2f(u.z >> 16) + b1.y);
    o.w = io_pack2(io_bf2f(u.w & 0xffff) + b1.z, io_bf2f(u.w >> 16) + b1.w);
    if (!ok) o = make_uint4(0, 0, 0, 0);
    return __builtin_bit_cast(bf16x8, o);
}
// row-major [row][k] tile image with padded rows: this lane's 8 consecutive k of row `row`
template <int RSB>
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* tile, int row, int chunk) {
    return *(const bf16x8*)(tile + row * RSB + chunk * 16);
}
// transposed fragment: A[m = col0 + (lane&31)][k = 8 rows krow0 + {0..3}, krow0 + 8 + {0..3}] of a [krow][col] image
template <int RSB>
__device__ __forceinline__ bf16x8 frag_cols(const unsigned char* tile, int krow0, int col0, int lane) {
    const int gq = lane >> 4, pp = lane & 15;
    const unsigned char* a0 = tile + (krow0 + 4 * (gq >> 1) + (pp >> 2)) * RSB + (col0 + 16 * (gq & 1) + 4 * (pp & 3)) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0 + 8 * RSB));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_frag(const float* f) {
    uint4 u;
    u.x = io_pack2(f[0], f[1]), u.y = io_pack2(f[2], f[3]), u.z = io_pack2(f[4], f[5]), u.w = io_pack2(f[6], f[7]);
    return __builtin_bit_cast(bf16x8, u);
}

// 32-row x DK tile, global -> registers (16-byte chunks, 256 threads) -> LDS image with RSB-byte rows
template <int NDB>
struct Tile {
    static constexpr int DK = 32 * NDB, CPR = DK / 8, RSB = DK * 2 + 16, BYTES = 32 * RSB;
    static constexpr int NCH = (32 * CPR + 255) / 256;
    uint4 r[NCH];
    // rowmap(r) -> global row index or -1 (zero fill)
    template <typename F>
    __device__ __forceinline__ void load(const u16* base, int64_t ld, int tid, F rowmap) {
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int c = tid + 256 * n;
            const int row = c / CPR, ch = c - row * CPR;
            const int64_t g = rowmap(row);
            const u16* src = (c < 32 * CPR && g >= 0) ? base + g * ld + ch * 8 : (const u16*)attn_zero_page;
            r[n] = *(const uint4*)src;
        }
    }
    __device__ __forceinline__ void commit(unsigned char* tile, int tid) const {
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int c = tid + 256 * n;
            const int row = c / CPR, ch = c - row * CPR;
            if (c < 32 * CPR) *(uint4*)(tile + row * RSB + ch * 16) = r[n];
        }
    }
};

#define SC_LD 68   // scratch row stride in floats: conflict-free 16-byte row writes and skewed 4-byte reads
#define GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))

#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)

// after the LAST asm MFMA of a phase: the compiler may copy / spill an accumulator right behind the statement (it does, at
// the loop back edge) and knows nothing of the MFMA still in flight -> pad the XDL-write -> VALU-read hazard here.
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 7"); }

// =====================================================================================================================
// Forward: 16 queries per wave (v_mfma_f32_16x16x32_bf16), 8 waves per workgroup = 2 per SIMD.
// A 32-query-per-wave version of this kernel (32x32x16 MFMA, the layout the backward below still uses) needs ~450
// registers per wave at d_k = 192, i.e. ONE wave per SIMD, and a lone wave issues one instruction every ~4 cycles: ~1000
// instructions per key block (softmax, dropout RNG, addressing) against 36 MFMAs made it issue bound at 11 % MFMA
// utilisation (profiles/r02_attn_pmc_fwd32.json, 363 us at the benchmark shape).  Halving the queries per wave halves O
// and both query-fragment sets (96 registers): two waves share every SIMD and their VALU / LDS / MFMA streams overlap.
// K / V / Pext tiles arrive by direct-to-LDS DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass); the
// DMA image is lane-linear, so rows are unpadded and the bank swizzle is an XOR of the 16-byte chunk index applied to
// the per-lane SOURCE address and again on every fragment read.
// Lane l: query = l & 15, lg = l >> 4.  MFMA k-steps are 32 deep; lane group lg supplies k-chunk PI(lg) of the step (any
// permutation is legal as long as A and B agree; {0,2,1,3} keeps the four 16-lane groups of a ds_read_b128 on distinct
// bank slots).  C layout: column = query, rows 4*lg + r.
// =====================================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SC16_LD 68

__device__ __forceinline__ void mfma16_cacc(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void chain16_begin(f32x4& acc) { asm volatile("s_nop 4" : "+a"(acc)); }
__device__ __forceinline__ void chain16_end(f32x4& acc) { asm volatile("s_nop 15" : "+a"(acc)); }
__device__ __forceinline__ f32x4 zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}

template <int NDB>
struct DT16 {   // 32-row x DK tile, DMA-filled by 8 waves, swizzled for 16-row fragment reads
    static constexpr int DK = 32 * NDB, CPR = DK / 8, RB = DK * 2;
    static constexpr int BYTES = ((32 * RB + 1023) / 1024) * 1024, NP = BYTES / 1024;
    __device__ static __forceinline__ int sw(int row) {
        if (CPR % 8 == 0) return (CPR == 16 ? ((row & 1) << 3) : 0) | (((row >> 1) & 3) << 1) | ((row >> 3) & 1);
        return (row >> 2) & 3;
    }
    template <typename F>
    __device__ static __forceinline__ void issue(unsigned char* tile, const u16* base, int64_t ld, int w, int lane, F rowmap) {
#pragma unroll
        for (int q = 0; q * 8 < NP; ++q) {
            const int pi = q * 8 + w;
            if (pi < NP) {
                const int ci = pi * 64 + lane;
                const int row = ci / CPR, pos = ci - row * CPR;
                const int64_t g = row < 32 ? rowmap(row) : -1;
                const u16* src = g >= 0 ? base + g * ld + ((pos ^ sw(row)) * 8) : (const u16*)attn_zero_page;
                __builtin_amdgcn_global_load_lds(GLB_AS(src), LDS_AS(tile + pi * 1024), 16, 0, 0);
            }
        }
    }
    __device__ static __forceinline__ bf16x8 rows(const unsigned char* tile, int row, int c) {
        return *(const bf16x8*)(tile + row * RB + ((c ^ sw(row)) << 4));
    }
    // transposed fragment of a 16-column tile: A[m = col0 + (lane&15)][k slots e<4: row kA + 4*lg + e, e>=4: row kB + 4*lg + e-4]
    __device__ static __forceinline__ bf16x8 cols(const unsigned char* tile, int kA, int kB, int col0, int lane) {
        const int lg = lane >> 4, pp = lane & 15;
        const int r0 = kA + 4 * lg + (pp >> 2), r1 = kB + 4 * lg + (pp >> 2);
        const int colb = col0 + 4 * (pp & 3), c = colb >> 3, in = (colb & 7) * 2;
        const unsigned char* a0 = tile + r0 * RB + ((c ^ sw(r0)) << 4) + in;
        const unsigned char* a1 = tile + r1 * RB + ((c ^ sw(r1)) << 4) + in;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a1));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
};

template <int NDB>
__global__ __launch_bounds__(512, 2) void attn_fwd16_kernel(AttnArgs p) {
    using D = DT16<NDB>;
    constexpr int DK = D::DK, KS = NDB, NDT = 2 * NDB, TB = D::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kb = smem;                            // [2]
    unsigned char* Vb = smem + 2 * TB;                   // [2]
    unsigned char* Pr = smem + 4 * TB;                   // 6 ring slots of 32-row Pext tiles
    float* sc = (float*)(smem + 10 * TB);                // [8 waves][16][SC16_LD]: ring of four 16-column band blocks
    unsigned int* kmw = (unsigned int*)(sc + 8 * 16 * SC16_LD);
    float* pbl = (float*)(kmw + 128);                    // [2][DK]: this head's pos_bias_u | pos_bias_v (when the kernel adds them)

    const int tid = threadIdx.x, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int PI = ((lg & 1) << 1) | (lg >> 1);          // k-chunk of this lane group inside a 32-deep MFMA step
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, NQB = (T + 127) / 128, NS = (T + 31) / 32;
    int wi = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wi & 7;
        wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wi >> 3);
    }
    if (p.use_redo && attn_redo[wi & 0xffff] == 0) return;
    const int bh = wi / NQB, qb = wi - bh * NQB;
    const int b = bh / p.H, h = bh - b * p.H;
    const int Q0 = qb * 128, q0 = Q0 + 16 * w, i = q0 + lq;
    const int XB = T - Q0 - 128;                         // band block n (16 rows) covers x in [XB + 16 n, +16); ring tile u = blocks 2u, 2u+1
    const u16* quB = p.qu + (int64_t)b * T * p.ldq + h * DK;
    const u16* qvB = p.qv + (int64_t)b * T * p.ldq + h * DK;
    const u16* kB = p.k + (int64_t)b * T * p.ldkv + h * DK;
    const u16* vB = p.v + (int64_t)b * T * p.ldkv + h * DK;
    const u16* pB = p.pos + h * DK;
    const uint8_t* mkB = p.keymask + (int64_t)b * T;
    float* scw = sc + w * 16 * SC16_LD;

    auto prow = [&](int u, int r) -> int64_t {           // Pext row of ring tile u, row r
        const int x = XB + 32 * u + r;
        return (x >= 0 && x < T) ? x : ((x > T && x - T - 1 < T) ? x - T - 1 : -1);
    };
    auto krow = [&](int s, int r) -> int64_t { return (32 * s + r < T) ? 32 * s + r : -1; };

    D::issue(Kb, kB, p.ldkv, w, lane, [&](int r) { return krow(0, r); });
    D::issue(Vb, vB, p.ldkv, w, lane, [&](int r) { return krow(0, r); });
#pragma unroll
    for (int u = 0; u < 5; ++u) D::issue(Pr + u * TB, pB, p.ldp, w, lane, [&](int r) { return prow(u, r); });
    const bool biased = p.bu != nullptr;
    if (biased) {
        if (tid < DK) pbl[tid] = p.bu[h * DK + tid], pbl[DK + tid] = p.bv[h * DK + tid];
        __syncthreads();
    }
    const lds_cfp pbu = (lds_cfp)LDS_AS(pbl), pbv = (lds_cfp)LDS_AS(pbl + DK);
    bf16x8 fqu[KS], fqv[KS];
    bool upper = false;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int off = 32 * kk + 8 * PI;
        fqu[kk] = ld_frag_qb(quB + (int64_t)i * p.ldq + off, i < T, pbu + off, biased);
        fqv[kk] = ld_frag_qb(qvB + (int64_t)i * p.ldq + off, i < T, pbv + off, biased);
    }
    for (int sb = w; sb < NS; sb += 8) {
        const int jl = 32 * sb + (lane & 31);
        const bool kvalid = (jl < T) && (mkB[jl < T ? jl : 0] != 0);
        const unsigned int vmw = (unsigned int)__ballot(kvalid);
        if (lane == 0) kmw[sb] = vmw;
    }
    __syncthreads();

    // block n of the band (16 x-rows): base x = XB + 16 n; it lies in the (q+v)[i+1] half iff x >= T <=> 16 n >= Q0 + 128
    auto band_switch = [&](int n) {
        if (16 * n >= Q0 + 128 && !upper) {
            upper = true;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                fqv[kk] = ld_frag_qb(qvB + (int64_t)(i + 1) * p.ldq + 32 * kk + 8 * PI, i + 1 < T, pbv + 32 * kk + 8 * PI, biased);
        }
    };
    auto band_block = [&](int n) -> f32x4 {
        const unsigned char* slot = Pr + ((n >> 1) % 6) * TB;
        const int row = 16 * (n & 1) + lq;
        f32x4 acc = zero4();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(D::rows(slot, row, 4 * kk + PI), fqv[kk], acc, 0, 0, 0);
        return acc;
    };
    auto band_store = [&](int n, const f32x4& acc) {
        *(float4*)(scw + lq * SC16_LD + 16 * (n & 3) + 4 * lg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    };
    {
        const int n = 7 - w;
        band_switch(n);
        band_store(n, band_block(n));
    }

    f32x4 O[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d) O[d] = zero4();
    float m_run = -1e30f, l_run = 0.f;
    const float NEG_INF = -__builtin_inff();
    const unsigned int ibase = (unsigned int)(((int64_t)bh * T + i) * T);

    for (int s = 0; s < NS; ++s) {
        const unsigned char* Kt = Kb + (s & 1) * TB;
        const unsigned char* Vt = Vb + (s & 1) * TB;
        const int n0 = 7 - w + 2 * s;                    // tile uses band blocks n0, n0+1, n0+2 (n0 from the previous step)
        band_switch(n0 + 1);                             // (a 16-block never straddles x = T; both new blocks share a side
        PHASE_FENCE();                                   //  unless n0+2 is the first upper one, handled below)
        // ---- S^T (two 16-key halves) and the two new band blocks ---------------------------------------------------------
        f32x4 sa0 = zero4(), sa1 = zero4();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(D::rows(Kt, lq, 4 * kk + PI), fqu[kk], sa0, 0, 0, 0);
            sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(D::rows(Kt, 16 + lq, 4 * kk + PI), fqu[kk], sa1, 0, 0, 0);
        }
        const f32x4 b1 = band_block(n0 + 1);
        band_store(n0 + 1, b1);
        band_switch(n0 + 2);
        const f32x4 b2 = band_block(n0 + 2);
        band_store(n0 + 2, b2);
        PHASE_FENCE();
        // ---- skewed band read: (query ii, key kk) <- band column 15 - ii + kk of blocks n0 .. n0+2 ------------------------
        const unsigned int vm = kmw[s];
        const int c0 = 16 * (n0 & 3) + 15 - lq + 4 * lg;
        const float* srow = scw + lq * SC16_LD;
        float sv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) sv[r] = srow[(c0 + (r & 3) + 16 * (r >> 2)) & 63];
        asm volatile("" ::: "memory");
        if (s + 1 < NS) {   // next tiles by DMA (after this step's last LDS write, see the 32-query kernel)
            D::issue(Kb + ((s + 1) & 1) * TB, kB, p.ldkv, w, lane, [&](int r) { return krow(s + 1, r); });
            D::issue(Vb + ((s + 1) & 1) * TB, vB, p.ldkv, w, lane, [&](int r) { return krow(s + 1, r); });
            D::issue(Pr + ((s + 5) % 6) * TB, pB, p.ldp, w, lane, [&](int r) { return prow(s + 5, r); });
        }
        float mloc = NEG_INF;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int kk = (r & 3) + 16 * (r >> 2) + 4 * lg;
            const float x = ((r < 4 ? sa0[r & 3] : sa1[r & 3]) + sv[r]) * p.scale;
            sv[r] = ((vm >> kk) & 1u) ? x : NEG_INF;
            mloc = fmaxf(mloc, sv[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (__builtin_expect(__any(mloc > m_run + 8.f), 0)) {
            const float mnew = fmaxf(m_run, mloc);
            const float alpha = __expf(m_run - mnew);
            l_run *= alpha;
            m_run = mnew;
#pragma unroll
            for (int d = 0; d < NDT; ++d) {
                chain16_end(O[d]);
#pragma unroll
                for (int r = 0; r < 4; ++r) O[d][r] *= alpha;
                chain16_begin(O[d]);
            }
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            sv[r] = __expf(sv[r] - m_run);
            psum += sv[r];
        }
        l_run += psum;
        if (p.drop_thr) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                bool kp[4];
                rng_keep4(p.drop_key, ibase + (unsigned int)(32 * s + 16 * g + 4 * lg), p.drop_thr, kp);
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[4 * g + e] = kp[e] ? sv[4 * g + e] * p.drop_inv : 0.f;
            }
        }
        const bf16x8 pf = pack_frag(sv);                 // B operand slots e<4: keys 4 lg + e, e>=4: keys 16 + 4 lg + e - 4
        PHASE_FENCE();
        // ---- O^T += V^T P^T: one 32-key MFMA step per 16-wide dv tile -----------------------------------------------------
#pragma unroll
        for (int d = 0; d < NDT; ++d) mfma16_cacc(O[d], D::cols(Vt, 0, 16, 16 * d, lane), pf);
        mfma_drain();
        __syncthreads();
    }

    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float invl = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int d = 0; d < NDT; ++d) chain16_end(O[d]);
    if (i < T) {
        u16* o = p.ctx + ((int64_t)b * T + i) * p.ldo + h * DK + 4 * lg;
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
            uint2 v2;
            v2.x = io_pack2(O[d][0] * invl, O[d][1] * invl);
            v2.y = io_pack2(O[d][2] * invl, O[d][3] * invl);
            *(uint2*)(o + 16 * d) = v2;
        }
        if (lg == 0) p.lse[(int64_t)bh * T + i] = l > 0.f ? m_run + __logf(l) : __builtin_inff();
    }
}

__device__ __forceinline__ void st4_bf16(u16* dst, float a, float b, float c, float d);

// =====================================================================================================================
// Forward, round 4: 4 waves x 32 queries, ONE wave per SIMD with the whole 512-entry register file, 32x32x16 MFMAs.
// Why not more waves: at d_k = 192 a wave's resident set is O (16 q x 192 -> 48 registers per 16 queries), (q+u) and
// (q+v) fragments (24 + 24 per 16 queries) plus staging; with 16 queries per wave and two waves per SIMD that is > 256
// registers (measured: 35-80 spills, and a spill reload is a vector-memory load whose vmcnt wait also waits for every
// tile DMA in flight -- 516 us instead of 290).  With 32 queries per wave every LDS fragment feeds twice the MFMA work
// (LDS fragment traffic per flop halves) and nothing spills; what one wave per SIMD loses -- another wave to fill its
// stalls -- is bought back by software pipelining INSIDE the wave: iteration s issues the products of step s+1
// (S = K (q+u)^T and the new band block) while the VALU exponentiates step s, then PV(s).
// No running maximum: the reference maximum of a row is fixed at the first key tile that holds a valid key (tiles in
// front of it are skipped altogether; the key mask does not depend on the query, so that tile is the same for the whole
// workgroup) and P = exp(s - m_ref) is used as is -- bf16 and fp32 carry 8 exponent bits, a probability of e^60 is as
// precise as one of 1.  A row whose later scores exceed m_ref by more than ~88 overflows l to +inf; the workgroup then
// raises its redo flag and the launcher's second kernel (the rescaling 16-query kernel above, which exits at once for
// every block whose flag is clear) recomputes that block.  This takes the max tracking, its cross-lane reductions and
// the O rescale (a second definition of the accumulators = 96 register moves per step) out of the loop.
// =====================================================================================================================
template <int NDB>
struct DT32 {   // 32-row x DK tile, DMA-filled (1-KiB pieces = 64 lanes x 16 B, lane-linear image), XOR-swizzled 16-byte chunks
    static constexpr int DK = 32 * NDB, CPR = DK / 8, RB = DK * 2, BYTES = 32 * RB, NP = BYTES / 1024;
    static constexpr int SWB = (CPR % 8 == 0) ? 3 : 2;
    // rows r and r+1 sit in different bank halves for free (RB = 64 NDB bytes); the swizzle spreads the rows of one parity:
    // bit-reversed so that rows r, r+2 (transposed reads) differ in the HIGH chunk bit and 8 consecutive same-parity rows
    // (ds_read_b128 lane groups) in all of them
    __device__ static __forceinline__ int sw(int row) {
        const int v = row >> 1;
        if (SWB == 3) return ((v & 1) << 2) | (v & 2) | ((v >> 2) & 1);
        return ((v & 1) << 1) | ((v >> 1) & 1);
    }
    // A / B fragment of a 32x32x16 MFMA, k-contiguous operand: row lr, k = 16 kk + 8 lh .. +7
    __device__ static __forceinline__ bf16x8 rows(const unsigned char* tile, int lr, int c) {
        return *(const bf16x8*)(tile + lr * RB + ((c ^ sw(lr)) << 4));
    }
};


__device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int NDB, bool DROP, bool SAVE, bool TWOPASS = false, bool SPLIT = false, bool SGN = false>
__global__ __launch_bounds__(256, 1) void attn_fwd32_kernel(AttnArgs p) {
    // SGN (training with dropout, probs_drop == NULL): ONE saved tensor -- probs = exp(s - m_ref) with the sign bit set where the
    // dropout mask dropped the element.  Its readers take |x| as the probability and the sign as the mask (a3t_attn_bwd_ds,
    // a3t_gemm_desc::a_signmask for dV): the dropped copy (T x T per head, written here and read once) does not exist.
    constexpr bool SG = SGN && DROP && SAVE;
    using D = DT32<NDB>;
    constexpr int DK = D::DK, KS = DK / 16, TB = D::BYTES, CPR = D::CPR, RB = D::RB, NPW = D::NP / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kb = smem;                            // [2]
    unsigned char* Vb = smem + 2 * TB;                   // [2]
    unsigned char* Pr = smem + 4 * TB;                   // 5 ring slots of 32-row Pext tiles
    float* sc = (float*)(smem + 9 * TB);                 // [4 waves][32][SC_LD] band scratch: two 32-column blocks
    unsigned int* kmw = (unsigned int*)(sc + 4 * 32 * SC_LD);   // [136] key-mask words
    unsigned char* stg = (unsigned char*)(kmw + 136);    // SAVE: [4 waves][2 tensors][32 rows][64 B] probability tiles on their way out

    const unsigned smem0 = lds_base(smem);
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, NQB = (T + 127) / 128, NS = (T + 31) / 32;
    // SPLIT launch: workgroups item0 .. are the tail blocks, one workgroup per (block, key range); they are dispatched last
    int wi = blockIdx.x, part = 0, titem = 0;
    const bool split = SPLIT && wi >= p.item0;
    if (split) {
        const int t = wi - p.item0;
        titem = t / p.nparts, part = t - titem * p.nparts;
        wi = p.item0 + titem;
    } else {
        const int nwg = SPLIT ? p.item0 : (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wi & 7;
        wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wi >> 3);
    }
    if (TWOPASS && attn_redo[wi & 0xffff] == 0) return;     // fixup launch: only the blocks whose row sums overflowed
    const int bh = wi / NQB, qb = wi - bh * NQB;
    const int sa_ = split ? (part * NS) / p.nparts : 0, sb = split ? ((part + 1) * NS) / p.nparts : NS;
    float* wsO = nullptr;
    float* wsL = nullptr;
    float* wsM = nullptr;
    if (split) {
        const int64_t pairs = (int64_t)gridDim.x - p.item0, nrow = pairs * 128, pr = (int64_t)(blockIdx.x - p.item0) * 128 + 32 * w + lr;
        wsO = p.part_ws + pr * DK;                        // [pair][128][DK] | [pair][128] | [block][128]
        wsL = p.part_ws + nrow * DK + pr;
        wsM = p.part_ws + nrow * (DK + 1) + (int64_t)titem * 128 + 32 * w + lr;
    }
    // a part's partial sums: un-normalised O and l relative to the block's reference maximum m2 (part 0 files m2)
    auto part_out = [&](const f32x16* O_, const float l_, const float m2_) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(wsO + 32 * d + 8 * g + 4 * lh) = make_float4(O_[d][4 * g], O_[d][4 * g + 1], O_[d][4 * g + 2], O_[d][4 * g + 3]);
        if (lh == 0) {
            *wsL = l_;
            if (part == 0) *wsM = m2_;
        }
        if (tid == 0 && part == 0) attn_redo[wi & 0xffff] = 0;      // attn_split_finish_kernel raises it
    };
    const int b = bh / p.H, h = bh - b * p.H;
    const int Q0 = qb * 128, q0 = Q0 + 32 * w, i = q0 + lr;
    const int X0 = T - 32 - Q0;                          // band block u covers x in [X0 + 32 (u - 3), +32)
    const u16* quB = p.qu + (int64_t)b * T * p.ldq + h * DK;
    const u16* qvB = p.qv + (int64_t)b * T * p.ldq + h * DK;
    const u16* kB = p.k + (int64_t)b * T * p.ldkv + h * DK;
    const u16* vB = p.v + (int64_t)b * T * p.ldkv + h * DK;
    const u16* pB = p.pos + h * DK;
    const uint8_t* mkB = p.keymask + (int64_t)b * T;
    float* scw = sc + w * 32 * SC_LD;

    // this head's pos_bias_u | pos_bias_v (when the kernel adds them) in the staging area of the probability stores, which is
    // idle until the key loop; the barrier below publishes them
    const bool biased = p.bu != nullptr;
    float* pbl = (float*)stg;
    if (biased && tid < DK) pbl[tid] = p.bu[h * DK + tid], pbl[DK + tid] = p.bv[h * DK + tid];
    // ---- key-mask words; the first tile with a valid key (the same for every query: the mask is per key) ----------------
    for (int sb = w; sb <= NS; sb += 4) {
        const int jl = 32 * sb + lr;
        const bool kvalid = (jl < T) && (mkB[jl < T ? jl : 0] != 0);
        const unsigned int vmw = (unsigned int)__ballot(kvalid);
        if (lane == 0) kmw[sb] = vmw;
    }
    __syncthreads();
    int s0 = 0;
    while (s0 < NS && kmw[s0] == 0) ++s0;
    s0 = __builtin_amdgcn_readfirstlane(s0);
    if (split && s0 >= NS) {                              // (no valid key: part 0 files m2 = +inf, the fold writes ctx = 0, lse = +inf)
        if (SAVE && i < T)
            for (int c = 32 * sa_ + lh * 4; c < 32 * sb && c < T; c += 8) {
                *(uint2*)(p.probs + ((int64_t)bh * T + i) * T + c) = make_uint2(0, 0);
                if (DROP && !SG) *(uint2*)(p.pdrop + ((int64_t)bh * T + i) * T + c) = make_uint2(0, 0);
            }
        f32x16 Oz[NDB];
#pragma unroll
        for (int d = 0; d < NDB; ++d) Oz[d] = zero16();
        part_out(Oz, 0.f, __builtin_inff());
        return;
    }
    if (s0 >= NS) {                                       // no valid key at all: zero context, lse = +inf
        if (i < T) {
            u16* o = p.ctx + ((int64_t)b * T + i) * p.ldo + h * DK;
            for (int c = lh * 4; c < DK; c += 8) *(uint2*)(o + c) = make_uint2(0, 0);
            if (lh == 0) p.lse[(int64_t)bh * T + i] = __builtin_inff();
            if (SAVE) {
                if (lh == 0) p.rowscale[(int64_t)bh * T + i] = 0.f;
                for (int c = lh * 4; c < T; c += 8) {
                    *(uint2*)(p.probs + ((int64_t)bh * T + i) * T + c) = make_uint2(0, 0);
                    if (DROP && !SG) *(uint2*)(p.pdrop + ((int64_t)bh * T + i) * T + c) = make_uint2(0, 0);
                }
            }
        }
        if (tid == 0) attn_redo[wi & 0xffff] = 0;
        return;
    }
    const int sa = split ? (sa_ > s0 ? sa_ : s0) : s0;    // this workgroup's key tiles: [sa, sb)
    if (SAVE && i < T)                                    // key tiles in front of the first valid one are skipped: their probabilities are 0
        for (int c = 32 * sa_ + lh * 4; c < 32 * s0 && c < 32 * sb; c += 8) {
            *(uint2*)(p.probs + ((int64_t)bh * T + i) * T + c) = make_uint2(0, 0);
            if (DROP && !SG) *(uint2*)(p.pdrop + ((int64_t)bh * T + i) * T + c) = make_uint2(0, 0);
        }

    // ---- DMA geometry: buffer descriptors whose range check supplies every zero (rows past the utterance, band rows
    // x < 0, x == T (xr = -1), x - T - 1 >= T all land beyond num_records; the row offset sits in voffset) ---------------
    const unsigned ldkv2 = (unsigned)p.ldkv * 2u, ldp2 = (unsigned)p.ldp * 2u;
    const int kvbytes = (int)((unsigned)(T - 1) * ldkv2 + DK * 2u), pbytes = (int)((unsigned)(T - 1) * ldp2 + DK * 2u);
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)kB, 0, kvbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)vB, 0, kvbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)pB, 0, pbytes, 0x00020000);
    unsigned voffK[3], dcol[3];
    int drow[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ci = (q * 4 + w) * 64 + lane;
        const int row = ci / CPR, pos = ci - row * CPR;
        drow[q] = row;
        dcol[q] = (unsigned)((pos ^ D::sw(row)) * 16);
        voffK[q] = (unsigned)row * ldkv2 + dcol[q];
    }
    auto issue_kv1 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned char* tile, int s, const int q) __attribute__((always_inline)) {
        if (q < NPW || (q * 4 + w) * 1024 < TB)
            dma16(r, smem0 + (unsigned)(tile - smem) + (q * 4 + w) * 1024, voffK[q] + (unsigned)(32 * s) * ldkv2);
    };
    auto issue_p1 = [&](int u, const int q) __attribute__((always_inline)) {
        if (q < NPW || (q * 4 + w) * 1024 < TB) {
            const int x = X0 + 32 * (u - 3) + drow[q];
            const int xr = x < T ? x : x - T - 1;
            dma16(rP, smem0 + (unsigned)(4 * TB + (u % 5) * TB + (q * 4 + w) * 1024), (unsigned)xr * ldp2 + dcol[q]);
        }
    };
    auto issue_kv = [&](const __amdgpu_buffer_rsrc_t& r, unsigned char* tile, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) issue_kv1(r, tile, s, q);
    };
    auto issue_p = [&](int u) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) issue_p1(u, q);
    };
    // K(s0), K(s0+1), V(s0), ring tiles s0 .. s0+4 (tile s0+5 takes tile s0's slot once the prologue is through with it)
    issue_kv(rK, Kb + (s0 & 1) * TB, s0);
    issue_kv(rK, Kb + ((s0 + 1) & 1) * TB, s0 + 1);
    issue_kv(rV, Vb + (s0 & 1) * TB, s0);
#pragma unroll
    for (int u = 0; u < 5; ++u) issue_p(s0 + u);

    bf16x8 fqu[KS], fqvL[KS], fqvU[KS];                  // (q+u)[i]; (q+v)[i] for the x < T half of the band, (q+v)[i+1] for x > T
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int off = 16 * kk + 8 * lh;
        const lds_cfp bu8 = (lds_cfp)LDS_AS(pbl + off), bv8 = (lds_cfp)LDS_AS(pbl + DK + off);
        fqu[kk] = ld_frag_qb(quB + (int64_t)i * p.ldq + off, i < T, bu8, biased);
        fqvL[kk] = ld_frag_qb(qvB + (int64_t)i * p.ldq + off, i < T, bv8, biased);
        fqvU[kk] = ld_frag_qb(qvB + (int64_t)(i + 1) * p.ldq + off, i + 1 < T, bv8, biased);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA (invisible to the compiler) and the fragment loads
    __syncthreads();

    // ---- fragment addressing -------------------------------------------------------------------------------------------
    const int swr = D::sw(lr);
    auto frag_rows = [&](const unsigned char* tile, const int kk) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(tile + lr * RB + (((2 * kk + lh) ^ swr) << 4));
    };
    // transposed fragment A[m = col0 + (lane&31)][k slots e<4: row krow0 + 4 lh + e, e>=4: row krow0 + 8 + 4 lh + e-4], col0 = 32 d.
    // Chunk of this lane's column = 4 d + c0 (c0 < 4); the swizzle XORs c0 with its low two bits and, when it has three
    // bits, d's parity with its third: 64 (d ^ b) = 64 d + 64 b (1 - 2 (d & 1)) -> one per-lane base per (row set, parity of d),
    // everything else is an immediate offset
    const int gq = lane >> 4, pp = lane & 15;
    const int tr_r0 = 4 * (gq >> 1) + (pp >> 2), tr_c = 16 * (gq & 1) + 4 * (pp & 3);
    int trb[2][2][2];                                     // [krow0 / 16][row r0 | r0 + 8][parity of d]
#pragma unroll
    for (int kr = 0; kr < 2; ++kr)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int r = 16 * kr + 8 * hf + tr_r0, sw = D::sw(r);
            const int base = r * RB + ((((tr_c >> 3) ^ sw) & 3) << 4) + (tr_c & 7) * 2;
            const int bflip = (D::SWB == 3) ? ((sw >> 2) & 1) * 64 : 0;
            trb[kr][hf][0] = base + bflip, trb[kr][hf][1] = base - bflip;
        }
    auto frag_cols = [&](const int vt, const int kr, const int d) __attribute__((always_inline)) -> bf16x8 {   // vt: tile offset in smem
        const unsigned char* a0 = smem + (vt + trb[kr][0][d & 1]) + 64 * d;
        const unsigned char* a1 = smem + (vt + trb[kr][1][d & 1]) + 64 * d;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_AS(a1));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto use_upper = [&](int u) -> bool { return (u - 3) * 32 >= 32 + Q0; };
    // band block u (32 band rows x this wave's 32 queries) -> fp32 scratch half (u & 1), stored [query][band column]
    auto band_mm = [&](int u) __attribute__((always_inline)) -> f32x16 {
        const unsigned char* slot = Pr + (u % 5) * TB;
        f32x16 acc = zero16();
        if (use_upper(u)) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) acc = mfma32(frag_rows(slot, kk), fqvU[kk], acc);
        } else {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) acc = mfma32(frag_rows(slot, kk), fqvL[kk], acc);
        }
        return acc;
    };
    auto band_store = [&](int u, const f32x16& acc) __attribute__((always_inline)) {
        float* row = scw + lr * SC_LD + 32 * (u & 1) + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float4*)(row + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    };
    // (query lr, key kk) <- band column 31 - lr + kk of blocks u (columns 0..31) and u + 1 (32..63), u = s - w + 3
    auto band_read = [&](int u, float* bd) __attribute__((always_inline)) {
        const int c0 = 32 * (u & 1) + 31 - lr + 4 * lh;
        const float* srow = scw + lr * SC_LD;
#pragma unroll
        for (int r = 0; r < 16; ++r) bd[r] = srow[(c0 + (r & 3) + 8 * (r >> 2)) & 63];
    };
    auto s_mm = [&](const unsigned char* Kt) __attribute__((always_inline)) -> f32x16 {
        f32x16 acc = zero16();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = mfma32(frag_rows(Kt, kk), fqu[kk], acc);
        return acc;
    };

    // ---- prologue: S(s0), both band blocks of step s0, the reference maximum ---------------------------------------------
    const float sl2 = p.scale * 1.4426950408889634f;     // exponentials in base 2: p = exp2(s sl2 - m2)
    f32x16 Sc = s_mm(Kb + (s0 & 1) * TB);
    float bd[16];
    {
        const int u = s0 - w + 3;
        band_store(u, band_mm(u));
        band_store(u + 1, band_mm(u + 1));
        band_read(u, bd);
    }
    float m2;
    auto tile_max = [&](const int s, const f32x16& S_, const float* bd_) __attribute__((always_inline)) -> float {
        const unsigned int vm = kmw[s];
        float mx = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if ((vm >> kk) & 1u) mx = fmaxf(mx, (S_[r] + bd_[r]) * sl2);
        }
        return fmaxf(mx, __shfl_xor(mx, 32, 64));
    };
    m2 = tile_max(s0, Sc, bd);                            // finite: tile s0 holds a valid key
    if (TWOPASS) {
        // Fixup of a block whose scores outgrew the first tile's maximum by more than ~88 (l overflowed): one plain sweep over
        // all key tiles for the TRUE row maximum (no pipelining -- this path is rare), then the normal pipeline with it.
        for (int s = s0 + 1; s < NS; ++s) {
            if (__builtin_amdgcn_readfirstlane(kmw[s]) == 0u) continue;
            __syncthreads();
            issue_kv(rK, Kb + (s & 1) * TB, s);
#pragma unroll
            for (int u = 0; u < 5; ++u) issue_p(s + u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const f32x16 S2 = s_mm(Kb + (s & 1) * TB);
            float bd2[16];
            const int u = s - w + 3;
            band_store(u, band_mm(u));
            band_store(u + 1, band_mm(u + 1));
            band_read(u, bd2);
            m2 = fmaxf(m2, tile_max(s, S2, bd2));
        }
    }
    if (split && sa >= sb) {                              // a key range in front of the first valid key: nothing to add
        f32x16 Oz[NDB];
#pragma unroll
        for (int d = 0; d < NDB; ++d) Oz[d] = zero16();
        part_out(Oz, 0.f, m2);
        return;
    }
    if (TWOPASS || (split && sa != s0)) {                 // (re)start the pipeline at key tile sa
        __syncthreads();
        issue_kv(rK, Kb + (sa & 1) * TB, sa);
        issue_kv(rK, Kb + ((sa + 1) & 1) * TB, sa + 1);
        issue_kv(rV, Vb + (sa & 1) * TB, sa);
#pragma unroll
        for (int u = 0; u < 5; ++u) issue_p(sa + u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        Sc = s_mm(Kb + (sa & 1) * TB);
        const int u = sa - w + 3;
        band_store(u, band_mm(u));
        band_store(u + 1, band_mm(u + 1));
        band_read(u, bd);
    }
    __syncthreads();                                      // every wave is through with ring tile sa
    issue_p(sa + 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef A3T_ATTN_TIMING
    unsigned long long tacc[32] = {}, tprev = __builtin_readcyclecounter();
#define TSTAMP(k) do { PHASE_FENCE(); const unsigned long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tprev; tprev = tn; PHASE_FENCE(); } while (0)
#else
#define TSTAMP(k) ((void)0)
#endif
    const int ttb = (int)((unsigned)T * (unsigned)T * 2u);
    const __amdgpu_buffer_rsrc_t rSP = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)(p.probs + (int64_t)bh * T * T) : (void*)p.ctx, 0, ttb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rSD = __builtin_amdgcn_make_buffer_rsrc((SAVE && DROP) ? (void*)((SG ? p.probs : p.pdrop) + (int64_t)bh * T * T) : (void*)p.ctx, 0, ttb, 0x00020000);
    f32x16 O[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d) O[d] = zero16();
    float l_run = 0.f;
    const unsigned int ibase = (unsigned int)(((int64_t)bh * T + i) * T);

    // One iteration = KS + NDB stages of two MFMAs each: stages 0 .. KS-1 multiply step s+1 (S and the new band block, two
    // independent accumulator chains), stages KS .. KS+NDB-1 are PV(s).  Fragments are fetched from LDS two stages ahead
    // (three staging pairs); the softmax of step s is cut into slices that ride on the first KS stages, the hand-over of step
    // s+1 (band block to the scratch, skewed read back) on the PV stages.  sched_barriers pin the slices to their stages --
    // left alone the scheduler emits [12 dependent MFMAs][12 dependent MFMAs][softmax][PV] with a one-deep LDS lookahead.
    constexpr int NSTG = KS + NDB;
    bf16x8 st[3][2];
    bf16x8 pf0, pf1;
    f32x16 Sn, Bn;
    auto rd = [&](const int t, const unsigned char* Kt, const unsigned char* slot, const int Vt) __attribute__((always_inline)) {
        if (t < KS) {
            st[t % 3][0] = frag_rows(Kt, t);
            st[t % 3][1] = frag_rows(slot, t);
        } else if (t < NSTG) {
            st[t % 3][0] = frag_cols(Vt, 0, t - KS);
            st[t % 3][1] = frag_cols(Vt, 1, t - KS);
        }
    };
    // product stages of step s+1 + the softmax of step s; two copies (the new band block takes (q+v)[i] or (q+v)[i+1]).  The PV
    // stages below exist ONCE: with two copies of them the accumulators O get two register homes and 96 moves per iteration.
    auto products = [&](auto UP, auto MASKED, const int s) __attribute__((always_inline)) {
        constexpr bool up = decltype(UP)::value, msk = decltype(MASKED)::value;   // msk: the key tile holds masked keys (utterance ends only)
        const unsigned char* Kt = Kb + ((s + 1) & 1) * TB;
        const int Vt = 2 * TB + (s & 1) * TB;
        const int un = s + 1 - w + 4;                     // the one new band block of step s+1
        const unsigned char* slot = Pr + (un % 5) * TB;
        const unsigned int vm = msk ? kmw[s] >> (4 * lh) : 0u;   // bit (r & 3) + 8 (r >> 2) = this lane's key of register r
        Sn = zero16(), Bn = zero16();
        float pv[16];
        unsigned int sw[8];       // SG: the tile's sixteen tagged probabilities as packed bf16 pairs (quad g -> sw[2 g], sw[2 g + 1])
        float psum = 0.f;
        auto expo = [&](const int r) __attribute__((always_inline)) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(Sc[r] + bd[r], sl2, -m2));
            pv[r] = (!msk || ((vm >> ((r & 3) + 8 * (r >> 2))) & 1u)) ? e : 0.f;
            psum += pv[r];
        };
        // Saved probabilities.  A lane holds 4 keys of 4 quads of ONE row: stored from there, an instruction touches 32 rows x
        // 32 B and the store path takes ~150 cycles per instruction (+65 us per launch, whatever the width of the pieces).  The
        // tile therefore goes through a per-wave LDS image ([32 rows][64 B], chunk-swizzled) and leaves as 16 rows x 64 B per
        // instruction.  Always exactly two store instructions per tensor and step (rows / columns outside the tensor are
        // dropped by the range check), so the end-of-iteration wait can be COUNTED: vmcnt(NSV) waits for the tile DMAs issued
        // before the stores and not for the write acknowledgements.
        auto stage4 = [&](const int ten, const int g) __attribute__((always_inline)) {     // this lane's quad g -> LDS image
            unsigned char* im = stg + (w * 2 + ten) * 2048;
            uint2 v2;
            v2.x = io_pack2(pv[4 * g], pv[4 * g + 1]), v2.y = io_pack2(pv[4 * g + 2], pv[4 * g + 3]);
            *(uint2*)(im + lr * 64 + ((g ^ ((lr >> 2) & 3)) << 4) + 8 * lh) = v2;
        };
        auto flush = [&](const __amdgpu_buffer_rsrc_t& r, const int ten) __attribute__((always_inline)) {
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            const unsigned char* im = stg + (w * 2 + ten) * 2048;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 16 * q + (lane >> 2), gl = (lane & 3) ^ ((row >> 2) & 3);      // image chunk lane & 3 holds keys 8 gl .. + 7
                const v4i_ dta = *(const v4i_*)(im + row * 64 + ((lane & 3) << 4));
                const int gi = q0 + row, col = 32 * s + 8 * gl;
                const unsigned vo = (gi < T && col < T) ? ((unsigned)gi * (unsigned)T + (unsigned)col) * 2u : 0x80000000u;
                A3T_SAVE_STORE(dta, r, vo);
            }
        };
        auto drop4 = [&](const int g) __attribute__((always_inline)) {
            if (DROP) {
                bool kp[4];
                rng_keep4(p.drop_key, ibase + (unsigned int)(32 * s + 8 * g + 4 * lh), p.drop_thr, kp);
                if (SG) {       // the saved probability itself, tagged: image 1 (the dropped copy's) carries the one tensor.  The
                    // P operand of the PV product comes off the SAME packed words (sign-tagged -> +0 by one packed integer max per
                    // pair, below); 1 / (1 - p) multiplies the accumulators once, after the key loop
                    unsigned char* im = stg + (w * 2 + 1) * 2048;
                    uint2 v2;
                    v2.x = io_pack2(kp[0] ? pv[4 * g] : -pv[4 * g], kp[1] ? pv[4 * g + 1] : -pv[4 * g + 1]);
                    v2.y = io_pack2(kp[2] ? pv[4 * g + 2] : -pv[4 * g + 2], kp[3] ? pv[4 * g + 3] : -pv[4 * g + 3]);
                    *(uint2*)(im + lr * 64 + ((g ^ ((lr >> 2) & 3)) << 4) + 8 * lh) = v2;
                    sw[2 * g] = v2.x, sw[2 * g + 1] = v2.y;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[4 * g + e] = kp[e] ? pv[4 * g + e] * p.drop_inv : 0.f;
                }
            }
        };
        auto dsave = [&](const int g) __attribute__((always_inline)) {
            if (SAVE && DROP && !SG) stage4(1, g);
        };
        // softmax slices over the KS product stages: exponentials first, then the dropout quads, then the packing
        // (measured: 4 / 6 / 8 exponential stages and a fence every stage or every second one all land within 2 %)
        constexpr int EXS = KS >= 8 ? KS - 4 : (KS > 1 ? KS - 1 : 1);        // stages that carry exponentials (the dropout quads follow)
        auto slice = [&](const int t) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r * EXS / 16 == t) expo(r);
            if (KS >= 8) {
                if (t == EXS - 1 && SAVE && !SG) stage4(0, 0), stage4(0, 1), stage4(0, 2), stage4(0, 3);   // (all sixteen exponentials are done after stage EXS - 1)
                if (t == EXS + 1 && SAVE && !SG) flush(rSP, 0);
                if (t >= EXS && t < EXS + 4) drop4(t - EXS), dsave(t - EXS);
            } else if (t == KS - 1) {
                if (SAVE && !SG) stage4(0, 0), stage4(0, 1), stage4(0, 2), stage4(0, 3), flush(rSP, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) drop4(g), dsave(g);
            }
            if (t == KS - 1) {
                if (SG) {
                    typedef short s16x8_ __attribute__((ext_vector_type(8)));
                    const s16x8_ z = {0, 0, 0, 0, 0, 0, 0, 0};
                    const uint4 u0 = make_uint4(sw[0], sw[1], sw[2], sw[3]), u1 = make_uint4(sw[4], sw[5], sw[6], sw[7]);
                    pf0 = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8_, u0), z));
                    pf1 = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8_, u1), z));
                } else {
                    pf0 = pack_frag(pv), pf1 = pack_frag(pv + 8);
                }
            }
        };
        rd(0, Kt, slot, Vt);
        rd(1, Kt, slot, Vt);
        PHASE_FENCE();
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            TSTAMP(5 + t);
            rd(t + 2, Kt, slot, Vt);
            Sn = mfma32(st[t % 3][0], fqu[t], Sn);
            Bn = mfma32(st[t % 3][1], up ? fqvU[t] : fqvL[t], Bn);
            // this iteration's tile DMA, one 1-KiB piece per stage (an LDS-DMA instruction holds the issue port for 50-150
            // cycles: under the MFMAs here, not in front of them): K(s+2), V(s+1), ring tile s+6 -- all three targets were
            // last read in iteration s-1
            if (KS >= 9) {
                if (t < 3) issue_kv1(rK, Kb + (s & 1) * TB, s + 2, t);
                else if (t < 6) issue_kv1(rV, Vb + ((s + 1) & 1) * TB, s + 1, t - 3);
                else if (t < 9) issue_p1(s + 6, t - 6);
            } else if (t == 0) {
                issue_kv(rK, Kb + (s & 1) * TB, s + 2);
                issue_kv(rV, Vb + ((s + 1) & 1) * TB, s + 1);
                issue_p(s + 6);
            }
            slice(t);
            PHASE_FENCE();
        }
        l_run += psum;
    };
    auto flush_d = [&](const int s) __attribute__((always_inline)) {    // the dropped tile of step s (staged by the product stages)
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        const unsigned char* im = stg + (w * 2 + 1) * 2048;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = 16 * q + (lane >> 2), gl = (lane & 3) ^ ((row >> 2) & 3);
            const v4i_ dta = *(const v4i_*)(im + row * 64 + ((lane & 3) << 4));
            const int gi = q0 + row, col = 32 * s + 8 * gl;
            const unsigned vo = (gi < T && col < T) ? ((unsigned)gi * (unsigned)T + (unsigned)col) * 2u : 0x80000000u;
            A3T_SAVE_STORE(dta, rSD, vo);
        }
    };
    auto pv_stages = [&](const int s) __attribute__((always_inline)) {
        const int Vt = 2 * TB + (s & 1) * TB;
        const int un = s + 1 - w + 4;
#pragma unroll
        for (int t = KS; t < NSTG; ++t) {
            TSTAMP(5 + t);
            rd(t + 2, nullptr, nullptr, Vt);
            const int d = t - KS;
            O[d] = mfma32(st[t % 3][0], pf0, O[d]);
            O[d] = mfma32(st[t % 3][1], pf1, O[d]);
            if (d == 0) band_store(un, Bn);
            if (SAVE && DROP && d == (NDB > 2 ? 2 : NDB - 1)) flush_d(s);
            if (d == 1 || NDB == 1) band_read(un - 1, bd);
            PHASE_FENCE();
        }
        TSTAMP(3);
        Sc = Sn;
    };
    typedef std::integral_constant<bool, false> FalseT;
    typedef std::integral_constant<bool, true> TrueT;
    constexpr int NSV = SAVE ? ((DROP && !SG) ? 4 : 2) : 0;      // probability stores per iteration

    for (int s = sa; s < sb; ++s) {
        // resident: K(s+1), V(s), ring tiles s+1 .. s+5; registers: Sc = S(s), bd = band values of step s
        TSTAMP(0);
        TSTAMP(1);
        const bool full = __builtin_amdgcn_readfirstlane(kmw[s]) == 0xffffffffu;
        if (use_upper(s + 1 - w + 4)) {
            if (full) products(TrueT(), FalseT(), s);
            else products(TrueT(), TrueT(), s);
        } else {
            if (full) products(FalseT(), FalseT(), s);
            else products(FalseT(), TrueT(), s);
        }
        pv_stages(s);
        // the DMA of this iteration has landed (it is older than the NSV probability stores), everyone is done reading
        PHASE_FENCE();
        if (NSV == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (NSV == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        PHASE_FENCE();
        TSTAMP(4);
    }
#ifdef A3T_ATTN_TIMING
    if (lane == 0 && p.dbd) {
        unsigned long long* o = (unsigned long long*)p.dbd + ((int64_t)blockIdx.x * 4 + w) * 32;
        for (int e = 0; e < 31; ++e) o[e] = tacc[e];
        o[31] = (unsigned long long)(NS - s0);
    }
#endif

    if (SG) {     // (the P operands were bf16(p) * keep: attention dropout's 1 / (1 - p) once per output element)
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[d][r] *= p.drop_inv;
    }
    float l = l_run + __shfl_xor(l_run, 32, 64);
    if (split) {
        part_out(O, l, m2);
        return;
    }
    const bool bad = !(l < 3.0e38f);                      // +inf / NaN: some score exceeded the reference maximum by > ~88
    const float invl = l > 0.f ? 1.f / l : 0.f;
    if (i < T) {
        u16* o = p.ctx + ((int64_t)b * T + i) * p.ldo + h * DK + 4 * lh;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                st4_bf16(o + 32 * d + 8 * g, O[d][4 * g] * invl, O[d][4 * g + 1] * invl, O[d][4 * g + 2] * invl, O[d][4 * g + 3] * invl);
        if (lh == 0) p.lse[(int64_t)bh * T + i] = (m2 + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
        if (SAVE && lh == 0) p.rowscale[(int64_t)bh * T + i] = invl;
    }
    const int anybad = __syncthreads_or(bad && i < T);
    if (tid == 0 && !TWOPASS) attn_redo[wi & 0xffff] = anybad ? 1 : 0;
}

// =====================================================================================================================
// Score gradients of the training path that keeps the forward's probabilities (a3t_attn_fwd_train):
//   dP = dctx V^T on the matrix cores, dS = p (keep/(1-p_drop) dP - delta) scale with p = probs * rowscale and
//   delta_i = dctx_i . ctx_i (formed in the kernel when a query block's dctx / ctx fragments arrive), written twice: row-major (operand of dK = dS^T (q+u) and d(q+u) = dS K) and
//   through the inverse legacy skew (attention.py:145-165) into the compact dBD matrix (operand of d(q+v) and d linear_pos).
// It replaces the dprobs GEMM (a T x T write) and a3t_relpos_softmax_bwd (a T x T read): 480 MB per launch at configs[1] instead
// of 800.  No state is carried along the keys (delta is known up front), so the unit of work is a strip of 128 queries x
// (KT x 32) keys; the kernel below says how a workgroup moves one.
struct DsArgs {
    const u16* dctx;      // [B*T][ldo]
    const u16* v;         // [B*T][ldkv], head h at column h*dk
    const u16* probs;     // [B][H][T][T] exp(s - m_ref)
    const float* rowscale;   // [B][H][T]
    const u16* ctx;       // [B*T][ldo]: the forward's output; delta_i = dctx_i . ctx_i is formed in the kernel
    u16* ds;              // [B][H][T][T]
    u16* dbd;             // compact dBD, block (b, h) at b*dbd_bsb + h*dbd_bsh
    int B, H, T;
    int64_t ldo, ldkv, dbd_bsb, dbd_bsh;
    float scale, drop_inv;
    unsigned int drop_thr, drop_key;
    unsigned long long* timing;   // A3T_DS_TIMING builds: per-wave cycle totals of the eight phases
    int signed_probs;     // the dropout mask is the sign bit of probs (a3t_attn_fwd_train without probs_drop)
    int64_t ds_bs;        // elements between the (b, h) blocks of ds (T * T, or more: blocks with room in front of them)
};

// chunk ci (row-contiguous 16-byte pieces, chunk index fastest) of the wave's probability strip
// (x / d for x * d < 2^20 as a multiply: the strip geometry is a runtime value and an integer division is ~40 instructions --
//  forty of them per task were a third of the kernel's VALU work)
__device__ __forceinline__ unsigned ds_magic(int d) { return (unsigned)(1048576.0f / (float)d) + 1u; }
__device__ __forceinline__ int ds_div(int x, unsigned magic) { return (int)(((unsigned)x * magic) >> 20); }
__device__ __forceinline__ uint4 ds_ld_chunk(const u16* prB, int T, int q0, int J0, int cpr, int npair, int ci) {
    const int row = ds_div(ci, ds_magic(cpr)), ch = ci - row * cpr;
    const int gi = q0 + row, col = J0 + 8 * ch;
    const bool ok = ci < npair && gi < T && col < T;
    const u16* src = ok ? prB + (int64_t)gi * T + col : (const u16*)attn_zero_page;
    return *(const uint4*)src;
}

// One workgroup = 128 queries (4 waves x 32) x a strip of up to KT key tiles.  Everything global is touched in row-contiguous
// 16-byte pieces (a vector-memory instruction costs with the number of cache lines it touches: the first version of this
// kernel, tile by tile with 16 rows x 64 B per instruction and 2-byte dBD stores, ran 297 us against 184 for the two kernels it
// replaces):  the wave's probability strip [32][nt*32] comes into a per-wave LDS image at the start, dS overwrites it in
// place tile by tile, and at the end the image leaves twice -- as rows of dS, and as the dBD rows, which are the SAME flat
// sequence shifted (dbd[r][c] = ds_flat[(r-1) T + c + r + 1]: two runs per query row, keys <= i into row i from column
// T-1-i, keys >= i+2 into row i+1 from column 0) and are written in DESTINATION-aligned 16-byte chunks: five dwords of the
// image funnel-shifted by the source's parity; the partial chunks at the ends of a run go out element by element.
// V tiles are shared by the four waves: DMA into a double buffer, one barrier per tile.
// DROP: 0 no dropout, 1 the mask comes back from the counter RNG, 2 it is read off the sign bits of probs (the one-tensor save of
// a3t_attn_fwd_train: p = |x|, dropped where x carries the sign bit)
// WDBD = false: dbd is not written at all -- the compact dBD matrix is the SAME flat sequence as dS shifted by T - 1 elements
// (dbd[r][c] = ds_flat[r (T + 1) + c - (T - 1)]), so a consumer can read it as a strided VIEW of dS (row stride T + 1, rows that start
// at odd element offsets: the 16-byte LDS-DMA takes 2-byte aligned sources, tools/probes/unaligned_dma_probe.hip) when T zeros sit in
// front of every (b, h) block of dS (ds_bs >= T * T + T; row 0 of dBD, whose first T - 1 entries never reach the scores, reads them).
template <int NDB, int DROP, int KT, bool WDBD = true>
__global__ __launch_bounds__(256, 2) void attn_bwd_ds_kernel(DsArgs p) {
    using D = DT32<NDB>;
    constexpr int DK = D::DK, KS = DK / 16, TB = D::BYTES, CPR = D::CPR, RS = KT * 64 + 16, IMG = 32 * RS, NPMIN = D::NP / 4;
    constexpr int NRING = 3;                             // V tiles in flight: DMA two tiles ahead, across task boundaries
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned smem0 = lds_base(smem);
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* img = smem + NRING * TB + w * IMG;
    const int T = p.T, NS = (T + 31) / 32, NKC = (NS + KT - 1) / KT, NQB = (T + 127) / 128;
    const int64_t ntasks = (int64_t)p.B * p.H * NQB * NKC;
    const unsigned ldkv2 = (unsigned)p.ldkv * 2u;
    const int vbytes = (int)((unsigned)(T - 1) * ldkv2 + DK * 2u);
    unsigned voffV[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ci = (q * 4 + w) * 64 + lane;
        const int row = ci / CPR, pos = ci - row * CPR;
        voffV[q] = (unsigned)row * ldkv2 + (unsigned)((pos ^ D::sw(row)) * 16);
    }
    const int swr = D::sw(lr);
    // a task = (b, h, 128-query block, strip of key tiles) as plain scalars (a struct handed around by reference lands in scratch)
#define A3T_DS_DECODE(P, task_)                                                        \
    do {                                                                               \
        const int kc_ = (int)((task_) % NKC);                                          \
        const int64_t t2_ = (task_) / NKC;                                             \
        const int qb_ = (int)(t2_ % NQB);                                              \
        P##bh = (int)(t2_ / NQB), P##b = P##bh / p.H, P##h = P##bh - P##b * p.H;       \
        P##q0 = 128 * qb_ + 32 * w;                                                    \
        P##sa = kc_ * KT;                                                              \
        const int sb_ = (P##sa + KT < NS) ? P##sa + KT : NS;                           \
        P##nt = sb_ - P##sa, P##J0 = 32 * P##sa, P##J1 = (32 * sb_ < T) ? 32 * sb_ : T; \
    } while (0)
#define A3T_DS_ISSUE_V(P, k_, slot_)                                                                                     \
    do {                                                                                                                 \
        const u16* vB_ = p.v + ((int64_t)P##b * T) * p.ldkv + P##h * DK;                                                 \
        const __amdgpu_buffer_rsrc_t rV_ = __builtin_amdgcn_make_buffer_rsrc((void*)vB_, 0, vbytes, 0x00020000);        \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                    \
            if ((q * 4 + w) * 1024 < TB)                                                                                 \
                dma16(rV_, smem0 + (unsigned)((slot_) * TB + (q * 4 + w) * 1024), voffV[q] + (unsigned)(32 * (P##sa + (k_))) * ldkv2); \
    } while (0)
    // the wave's inputs of a task are held in registers until its image is free: probability strip (row-contiguous 16-byte
    // chunks), dctx fragments, row factors.  (A macro, not a lambda: arrays handed to a lambda by reference end up in scratch.)
    static_assert(KT == 5, "ten named strip registers below");
    uint4 pr0, pr1, pr2, pr3, pr4, pr5, pr6, pr7, pr8, pr9;      // (named: an array that lives across the task loop stays in scratch)
    bf16x8 in_fd[KS], in_fc[KS];
    float in_rsc, in_dl = 0.f;
#define A3T_DS_FETCH(P)                                                                                             \
    do {                                                                                                            \
        const int i_ = P##q0 + lr, cpr_ = 4 * P##nt, np_ = 32 * cpr_;                                               \
        const u16* prB_ = p.probs + (int64_t)P##bh * T * T;                                                         \
        const u16* dcB_ = p.dctx + ((int64_t)P##b * T) * p.ldo + P##h * DK;                                         \
        const u16* cxB_ = p.ctx + ((int64_t)P##b * T) * p.ldo + P##h * DK;                                          \
        pr0 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, lane), pr1 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 64 + lane);         \
        pr2 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 128 + lane), pr3 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 192 + lane);  \
        pr4 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 256 + lane), pr5 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 320 + lane);  \
        pr6 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 384 + lane), pr7 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 448 + lane);  \
        pr8 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 512 + lane), pr9 = ds_ld_chunk(prB_, T, P##q0, P##J0, cpr_, np_, 576 + lane);  \
        if (fetch_rows) {       /* (wave-uniform: a new query block) */                                            \
            _Pragma("unroll") for (int kk = 0; kk < KS; ++kk) {                                                     \
                in_fd[kk] = ld_frag_g(dcB_ + (int64_t)i_ * p.ldo + 16 * kk + 8 * lh, i_ < T);                       \
                in_fc[kk] = ld_frag_g(cxB_ + (int64_t)i_ * p.ldo + 16 * kk + 8 * lh, i_ < T);                       \
            }                                                                                                       \
            in_rsc = i_ < T ? p.rowscale[(int64_t)P##bh * T + i_] * p.scale : 0.f;                                  \
        }                                                                                                           \
    } while (0)
#define A3T_DS_PUT(u_, v_)                                                          \
    do {                                                                            \
        const int ci = 64 * (u_) + lane, row = ds_div(ci, mg_cpr), ch = ci - row * cpr; \
        if (ci < npair) *(uint4*)(img + row * RS + ch * 16) = (v_);                 \
    } while (0)

    // a workgroup walks a CONTIGUOUS range of tasks = the key strips of one query block after the other: the dctx fragments and
    // row factors are requested once per query block, not once per strip (193 MB less through the CU's address path per launch)
    // XCD-contiguous: workgroup ids are dealt round-robin over the 8 XCDs, each with an L2 of its own; the workgroups of ONE XCD
    // take neighbouring task ranges = the query blocks of the same few (b, h) and share their V tiles there.  (Round 6 measured
    // what that is worth: FETCH_SIZE 514 MB against 538 MB per launch with workgroup id = range index, 182 against 184 us --
    // profiles/r06_ds_map_ab.txt.  The V tiles were never the kernel's problem; the mapping stays because it costs nothing.)
    int vb = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = vb & 7;
        vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    }
    const int64_t tper = (ntasks + gridDim.x - 1) / gridDim.x;
    int64_t task = (int64_t)vb * tper;
    const int64_t tend = task + tper < ntasks ? task + tper : ntasks;
    if (task >= tend) return;
#ifdef A3T_DS_TIMING
    unsigned long long tacc[8] = {}, tprev = __builtin_readcyclecounter();
#define DSTAMP(k) do { PHASE_FENCE(); const unsigned long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tprev; tprev = tn; PHASE_FENCE(); } while (0)
#else
#define DSTAMP(k) ((void)0)
#endif
    int c_bh, c_b, c_h, c_q0, c_sa, c_nt, c_J0, c_J1, n_bh, n_b, n_h, n_q0, n_sa, n_nt, n_J0, n_J1;
    A3T_DS_DECODE(c_, task);
    int gt = 0;                                          // running tile count of this workgroup: ring slot = gt % NRING
    A3T_DS_ISSUE_V(c_, 0, 0);
    if (c_nt > 1) A3T_DS_ISSUE_V(c_, 1, 1);
    bool fetch_rows = true;
    A3T_DS_FETCH(c_);
    while (true) {
        const int64_t ntask = task + 1;
        const bool more = ntask < tend;
        A3T_DS_DECODE(n_, (more ? ntask : task));
        const int cpr = 4 * c_nt, npair = 32 * cpr;
        const unsigned mg_cpr = ds_magic(cpr), mg_slot = ds_magic(cpr + 1);
        const int i = c_q0 + lr;
        DSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this task's inputs (and its first V tiles, and the last task's stores)
        DSTAMP(1);
        A3T_DS_PUT(0, pr0); A3T_DS_PUT(1, pr1); A3T_DS_PUT(2, pr2); A3T_DS_PUT(3, pr3); A3T_DS_PUT(4, pr4);
        A3T_DS_PUT(5, pr5); A3T_DS_PUT(6, pr6); A3T_DS_PUT(7, pr7); A3T_DS_PUT(8, pr8); A3T_DS_PUT(9, pr9);
        bf16x8 fd[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) fd[kk] = in_fd[kk];
        if (fetch_rows) {       // a new query block: delta_i = dctx_i . ctx_i, the row term of the softmax backward (a lane holds
            float sdl = 0.f;    // half of the row's d_k columns, its partner lane ^ 32 the other half)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const uint4 a4 = __builtin_bit_cast(uint4, in_fd[kk]), c4 = __builtin_bit_cast(uint4, in_fc[kk]);
                sdl = __builtin_fmaf(io_bf2f(a4.x & 0xffff), io_bf2f(c4.x & 0xffff), sdl), sdl = __builtin_fmaf(io_bf2f(a4.x >> 16), io_bf2f(c4.x >> 16), sdl);
                sdl = __builtin_fmaf(io_bf2f(a4.y & 0xffff), io_bf2f(c4.y & 0xffff), sdl), sdl = __builtin_fmaf(io_bf2f(a4.y >> 16), io_bf2f(c4.y >> 16), sdl);
                sdl = __builtin_fmaf(io_bf2f(a4.z & 0xffff), io_bf2f(c4.z & 0xffff), sdl), sdl = __builtin_fmaf(io_bf2f(a4.z >> 16), io_bf2f(c4.z >> 16), sdl);
                sdl = __builtin_fmaf(io_bf2f(a4.w & 0xffff), io_bf2f(c4.w & 0xffff), sdl), sdl = __builtin_fmaf(io_bf2f(a4.w >> 16), io_bf2f(c4.w >> 16), sdl);
            }
            in_dl = sdl + __shfl_xor(sdl, 32, 64);
        }
        const float rsc = in_rsc, dl = in_dl;
        const unsigned int ibase = (unsigned int)(((int64_t)c_bh * T + i) * T);
        u16* dsB = p.ds + (int64_t)c_bh * p.ds_bs;
        u16* dbB = WDBD ? p.dbd + (int64_t)c_b * p.dbd_bsb + (int64_t)c_h * p.dbd_bsh : nullptr;
        if (WDBD && c_q0 == 0 && c_sa == 0)          // BD[0][0 .. T-2] never reaches the scores (attention.py:157-165)
            for (int c = lane; c < T - 1; c += 64) dbB[c] = 0;
        DSTAMP(2);
        // ---- tiles
        bool issued_prev = false;
        for (int k = 0; k < c_nt; ++k, ++gt) {
            // V(k) has landed: of the DMA issued after it only the tile requested during tile k-1 may still be in flight (if one
            // was requested then: the last tiles of a workgroup's last task request nothing, and vmcnt(N) with N pieces of V(k)
            // itself outstanding would let them through -- a race the full test suite found under A3T_ATTN_BWD_DS=1)
            if (k == 0 || !issued_prev) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (NPMIN >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (NPMIN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (NPMIN == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            DSTAMP(3);
            // two tiles ahead, into the slot every wave left before this barrier; past the end of the task: the next task's
            issued_prev = true;
            if (k + 2 < c_nt) A3T_DS_ISSUE_V(c_, k + 2, (gt + 2) % NRING);
            else if (more && k + 2 - c_nt < n_nt) A3T_DS_ISSUE_V(n_, k + 2 - c_nt, (gt + 2) % NRING);
            else issued_prev = false;
            const unsigned char* Vt = smem + (gt % NRING) * TB;
            f32x16 dP = zero16();
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                dP = mfma32(*(const bf16x8*)(Vt + lr * D::RB + (((2 * kk + lh) ^ swr) << 4)), fd[kk], dP);
            unsigned char* cell = img + lr * RS + k * 64 + 8 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 pk = *(const uint2*)(cell + 16 * g);
                const float p0 = __uint_as_float((pk.x << 16) & 0x7fffffffu), p1 = __uint_as_float(pk.x & 0x7fff0000u);
                const float p2 = __uint_as_float((pk.y << 16) & 0x7fffffffu), p3 = __uint_as_float(pk.y & 0x7fff0000u);
                float d0_ = dP[4 * g], d1_ = dP[4 * g + 1], d2_ = dP[4 * g + 2], d3_ = dP[4 * g + 3];
                if (DROP == 2) {
                    d0_ = (pk.x & 0x8000u) ? 0.f : d0_ * p.drop_inv, d1_ = (pk.x & 0x80000000u) ? 0.f : d1_ * p.drop_inv;
                    d2_ = (pk.y & 0x8000u) ? 0.f : d2_ * p.drop_inv, d3_ = (pk.y & 0x80000000u) ? 0.f : d3_ * p.drop_inv;
                } else if (DROP) {
                    const unsigned int i0 = ibase + (unsigned int)(32 * (c_sa + k) + 8 * g + 4 * lh);
                    const unsigned int h0 = rng_pair(p.drop_key, i0 >> 1), h1 = rng_pair(p.drop_key, (i0 >> 1) + 1), t = p.drop_thr >> 16;
                    d0_ = (h0 & 0xffffu) >= t ? d0_ * p.drop_inv : 0.f, d1_ = (h0 >> 16) >= t ? d1_ * p.drop_inv : 0.f;
                    d2_ = (h1 & 0xffffu) >= t ? d2_ * p.drop_inv : 0.f, d3_ = (h1 >> 16) >= t ? d3_ * p.drop_inv : 0.f;
                }
                uint2 o;
                o.x = io_pack2(p0 * rsc * (d0_ - dl), p1 * rsc * (d1_ - dl)), o.y = io_pack2(p2 * rsc * (d2_ - dl), p3 * rsc * (d3_ - dl));
                *(uint2*)(cell + 16 * g) = o;
            }
            DSTAMP(4);
        }
        // (a one-tile task leaves the next task's second V tile to be issued here; a barrier first: slot gt+1 may still be read)
        if (more && c_nt == 1 && n_nt > 1) {
            __syncthreads();
            A3T_DS_ISSUE_V(n_, 1, (gt + 1) % NRING);
        }
        // ---- the next task's inputs travel while this one's image leaves
        fetch_rows = n_bh != c_bh || n_q0 != c_q0;
        if (more) A3T_DS_FETCH(n_);
        DSTAMP(5);
        // ---- dS rows
        for (int c0 = 0; c0 < npair; c0 += 64) {
            const int ci = c0 + lane, row = ds_div(ci, mg_cpr), ch = ci - row * cpr;
            const int gi = c_q0 + row, col = c_J0 + 8 * ch;
            if (ci < npair && gi < T && col < T) *(uint4*)(dsB + (int64_t)gi * T + col) = *(const uint4*)(img + row * RS + ch * 16);
        }
        DSTAMP(6);
        // ---- dBD: per query row a lower run (keys <= i -> row i) and an upper run (keys >= i+2 -> row i+1)
        const int nslot = cpr + 1, nps = 32 * nslot;
#pragma unroll 1
        for (int run = 0; run < (WDBD ? 2 : 0); ++run) {
            if (run == 0 ? (c_J0 > c_q0 + 31) : (c_J1 - 1 < c_q0 + 2)) continue;        // no row of this wave has that run
            // whole 16-byte chunks of the destination
            for (int c0 = 0; c0 < nps; c0 += 64) {
                const int id = c0 + lane, row = ds_div(id, mg_slot), slot = id - row * nslot, gi = c_q0 + row;
                if (id >= nps || gi >= T) continue;
                int jlo, jhi, drow, dc0;
                if (run == 0) jlo = c_J0, jhi = (c_J1 - 1 < gi) ? c_J1 - 1 : gi, drow = gi, dc0 = T - 1 - gi + jlo;
                else jlo = (c_J0 > gi + 2) ? c_J0 : gi + 2, jhi = c_J1 - 1, drow = gi + 1, dc0 = jlo - gi - 2;
                const int n = jhi - jlo + 1;
                const int cc = (dc0 & ~7) + 8 * slot;
                if (n <= 0 || cc < dc0 || cc + 8 > dc0 + n) continue;
                const int so = jlo - c_J0 - dc0 + cc;              // source element (strip-local) of destination column cc
                const unsigned int* dw = (const unsigned int*)(img + row * RS + (so >> 1) * 4);
                const unsigned int a0 = dw[0], a1 = dw[1], a2 = dw[2], a3 = dw[3], a4 = dw[4];
                const unsigned int sh = (unsigned int)(so & 1) * 16u;
                uint4 o;
                o.x = __builtin_amdgcn_alignbit(a1, a0, sh), o.y = __builtin_amdgcn_alignbit(a2, a1, sh);
                o.z = __builtin_amdgcn_alignbit(a3, a2, sh), o.w = __builtin_amdgcn_alignbit(a4, a3, sh);
                *(uint4*)(dbB + (int64_t)drow * T + cc) = o;
            }
            // the partial chunks at the two ends of each row's run: all 64 of them in ONE pass of element stores (a vector-memory
            // instruction costs ~50 cycles of the CU's address path whatever its mask: spread over the loop above they were 88
            // instructions per wave and task)
            {
                const int row = lane >> 1, tail = lane & 1, gi = c_q0 + row;
                int jlo, jhi, drow, dc0;
                if (run == 0) jlo = c_J0, jhi = (c_J1 - 1 < gi) ? c_J1 - 1 : gi, drow = gi, dc0 = T - 1 - gi + jlo;
                else jlo = (c_J0 > gi + 2) ? c_J0 : gi + 2, jhi = c_J1 - 1, drow = gi + 1, dc0 = jlo - gi - 2;
                const int n = jhi - jlo + 1;
                const int chd = dc0 & ~7, ctl = (dc0 + n - 1) & ~7, cc = tail ? ctl : chd;
                const bool mine = gi < T && n > 0 && !(cc >= dc0 && cc + 8 <= dc0 + n) && (!tail || ctl != chd);
                u16* dst = dbB + (int64_t)drow * T;
                const unsigned char* srow = img + row * RS;
                const int so0 = jlo - c_J0 - dc0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int x = cc + e;
                    if (mine && x >= dc0 && x < dc0 + n) dst[x] = *(const u16*)(srow + (so0 + x) * 2);
                }
            }
        }
        DSTAMP(7);
        if (!more) break;
        task = ntask;
        c_bh = n_bh, c_b = n_b, c_h = n_h, c_q0 = n_q0, c_sa = n_sa, c_nt = n_nt, c_J0 = n_J0, c_J1 = n_J1;
    }
#ifdef A3T_DS_TIMING
    if (lane == 0 && p.timing) {
        unsigned long long* o = p.timing + ((int64_t)vb * 4 + w) * 8;
        for (int e = 0; e < 8; ++e) o[e] = tacc[e];
    }
#endif
#undef DSTAMP
#undef A3T_DS_FETCH
#undef A3T_DS_PUT
#undef A3T_DS_DECODE
#undef A3T_DS_ISSUE_V
}

// =====================================================================================================================
// Helpers of the backward.  (The two-pass flash backward of round 2 -- attn_bwd_q_kernel / attn_bwd_kv_kernel, recomputing the
// probabilities from the saved log-sum-exp: 1.2 + 1.5 ms per layer against ~0.6 ms for the materialised path -- left the library
// in round 5: tools/experiments/attn_flash_backward.patch.)
// =====================================================================================================================
#define GS_LD 72   // gradient scratch row stride in bf16 elements

__device__ __forceinline__ void st4_bf16(u16* dst, float a, float b, float c, float d) {
    uint2 v2;
    v2.x = io_pack2(a, b), v2.y = io_pack2(c, d);
    *(uint2*)dst = v2;
}

// Fold of the key-split tail blocks (launch_fwd16): O = sum of the parts' un-normalised sums, l likewise, both relative to
// the block's one reference maximum -> ctx, lse, rowscale and the overflow flag exactly as the unsplit epilogue writes them.
// grid = items x 32, one row per wave, lane = 4 columns; the (<= 4) parts of a row are requested together.
__global__ __launch_bounds__(256) void attn_split_finish_kernel(AttnArgs p, int nitems, int DK) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int titem = blockIdx.x >> 5, P = p.nparts, T = p.T, NQB = (T + 127) / 128;
    const int wi = p.item0 + titem, bh = wi / NQB, qb = wi - bh * NQB, b = bh / p.H, h = bh - b * p.H;
    const int64_t nrow = (int64_t)nitems * P * 128;
    const int row = (blockIdx.x & 31) * 4 + w, i = qb * 128 + row;
    bool bad = false;
    if (i < T) {
        float lp[4] = {0.f, 0.f, 0.f, 0.f};
        float4 op[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            op[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < P) {
                const int64_t pr = ((int64_t)titem * P + q) * 128 + row;
                lp[q] = p.part_ws[nrow * DK + pr];
                if (lane * 4 < DK) op[q] = *(const float4*)(p.part_ws + pr * DK + lane * 4);
            }
        }
        const float m2 = p.part_ws[nrow * (DK + 1) + (int64_t)titem * 128 + row];
        const float l = (lp[0] + lp[1]) + (lp[2] + lp[3]);
        const float4 o = make_float4((op[0].x + op[1].x) + (op[2].x + op[3].x), (op[0].y + op[1].y) + (op[2].y + op[3].y),
                                     (op[0].z + op[1].z) + (op[2].z + op[3].z), (op[0].w + op[1].w) + (op[2].w + op[3].w));
        bad = !(l < 3.0e38f);
        const float invl = l > 0.f ? 1.f / l : 0.f;
        if (lane * 4 < DK) {
            uint2 pk;
            pk.x = io_pack2(o.x * invl, o.y * invl), pk.y = io_pack2(o.z * invl, o.w * invl);
            *(uint2*)(p.ctx + ((int64_t)b * T + i) * p.ldo + h * DK + lane * 4) = pk;
        }
        if (lane == 0) {
            p.lse[(int64_t)bh * T + i] = m2 == __builtin_inff() ? m2 : (m2 + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
            if (p.rowscale) p.rowscale[(int64_t)bh * T + i] = invl;
        }
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(&attn_redo[wi & 0xffff], 1);
}

static inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

// workspace of the key-split launch, one per device, allocated at the first use (<= 256 (block, part) pairs of 128 rows):
// like the overflow flags it serves ONE attention launch at a time per device (the engine issues them on one stream)
static float* attn_ws_ptr[64];
static size_t attn_ws_cap[64];
void attn_release_split_ws() {      // a3t_release_workspaces (gemm_bf16_8p.hip); the caller has drained the attention stream
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int dev = 0; dev < 64; ++dev)
        if (attn_ws_ptr[dev]) {
            (void)hipSetDevice(dev);
            (void)hipDeviceSynchronize();
            (void)hipFree(attn_ws_ptr[dev]);
            attn_ws_ptr[dev] = nullptr, attn_ws_cap[dev] = 0;
        }
    (void)hipSetDevice(cur);
}
static float* attn_split_ws(size_t floats) {
    float** ws = attn_ws_ptr;
    size_t* cap = attn_ws_cap;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (cap[dev] < floats) {
        if (ws[dev]) (void)hipFree(ws[dev]);
        ws[dev] = nullptr, cap[dev] = 0;
        if (hipMalloc((void**)&ws[dev], floats * sizeof(float)) != hipSuccess) return nullptr;
        cap[dev] = floats;
    }
    return ws[dev];
}
static int attn_cus() {
    static int n[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n[dev]) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
        n[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    return n[dev];
}
// A3T_ATTN_SPLIT=0 / a3t_attn_split_mode(0): no key-split of the tail blocks
static int g_attn_split = -1;
static int attn_split_on() {
    if (g_attn_split < 0) {
        const char* e = getenv("A3T_ATTN_SPLIT");
        g_attn_split = (e && !strcmp(e, "0")) ? 0 : 1;
    }
    return g_attn_split;
}
extern "C" int a3t_attn_split_mode(int mode) {
    const int old = attn_split_on();
    g_attn_split = mode < 0 ? -1 : (mode ? 1 : 0);
    return old;
}

template <int NDB>
static int launch_fwd16(const AttnArgs& a, hipStream_t s) {
    constexpr int TB = DT16<NDB>::BYTES;
    constexpr int lds = 10 * TB + 8 * 16 * SC16_LD * 4 + 4 * 128 + 2 * 32 * NDB * 4;
    // (the attribute is per device: set it on every launch -- a process that drives several GPUs would otherwise launch on
    //  its second device without the raised LDS limit)
    const int nqb = (a.T + 127) / 128;
    const unsigned grid = (unsigned)(a.B * a.H * nqb);
    AttnArgs a16 = a;
    // The 32-query one-wave-per-SIMD kernel with the rescaling 16-query kernel as its overflow fixup; beyond 2^16
    // blocks (the overflow flags) the 16-query kernel alone.  Only the fixed-reference kernel can save probabilities.
    if (a.probs && grid > (1u << 16)) return A3T_EINVAL;
    if (grid <= (1u << 16)) {
        constexpr int TB32 = DT32<NDB>::BYTES;
        constexpr int lds32 = 9 * TB32 + 4 * 32 * SC_LD * 4 + 136 * 4 + 4 * 2 * 2048;
        // One workgroup per CU and equal workgroups: the launch runs in ceil(grid / CUs) rounds and the last one may be mostly
        // empty (configs[1]: 576 blocks = 2.25 rounds on 256 CUs -> the time of 3).  The blocks of a last round that fills at most
        // half the chip go to the END of the grid, each split into 2..4 key ranges (one short round instead of a full one); the parts
        // share the block's reference maximum (every part evaluates the first valid key tile), so their un-normalised sums add.
        const int cus = attn_cus();
        const unsigned nfull = (grid / (unsigned)cus) * (unsigned)cus, ntail = grid - nfull;
        int nparts = ntail ? (int)((unsigned)cus / ntail) : 1;
        if (nparts > 4) nparts = 4;
        const int ns = (a.T + 31) / 32;
        AttnArgs at = a;
        if (attn_split_on() && nfull && ntail && nparts >= 2 && ns >= 4 * nparts) {
            const size_t rows = (size_t)ntail * nparts * 128;
            at.part_ws = attn_split_ws(rows * (DT32<NDB>::DK + 1) + (size_t)ntail * 128);
            at.item0 = (int)nfull, at.nparts = nparts;
        }
        const bool split = at.part_ws != nullptr;
#define A3T_L32(DR, SV, SG)                                                                                                             \
    do {                                                                                                                             \
        (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel<NDB, DR, SV, false, false, SG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds32);   \
        if (split) {                                                                                                                 \
            (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel<NDB, DR, SV, false, true, SG>,                                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds32);                                            \
            hipLaunchKernelGGL((attn_fwd32_kernel<NDB, DR, SV, false, true, SG>), dim3(nfull + ntail * nparts), dim3(256), lds32, s, at); \
            hipLaunchKernelGGL(attn_split_finish_kernel, dim3(ntail * 32), dim3(256), 0, s, at, (int)ntail, DT32<NDB>::DK);            \
        } else {                                                                                                                     \
            hipLaunchKernelGGL((attn_fwd32_kernel<NDB, DR, SV, false, false, SG>), dim3(grid), dim3(256), lds32, s, a);                                 \
        }                                                                                                                            \
    } while (0)
        if (a.probs) {
            // training: the fixup of an overflowed block has to re-write its saved probabilities too -> the same kernel with
            // a first sweep for the true row maximum (every block but the flagged ones exits at once)
            if (a.drop_thr && !a.pdrop) {       // one saved tensor: the dropout mask rides on the sign bits of probs
                A3T_L32(true, true, true);
                (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel<NDB, true, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds32);
                hipLaunchKernelGGL((attn_fwd32_kernel<NDB, true, true, true, false, true>), dim3(grid), dim3(256), lds32, s, a);
            } else if (a.drop_thr) {
                A3T_L32(true, true, false);
                (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel<NDB, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds32);
                hipLaunchKernelGGL((attn_fwd32_kernel<NDB, true, true, true>), dim3(grid), dim3(256), lds32, s, a);
            } else {
                A3T_L32(false, true, false);
                (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel<NDB, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds32);
                hipLaunchKernelGGL((attn_fwd32_kernel<NDB, false, true, true>), dim3(grid), dim3(256), lds32, s, a);
            }
            return (int)hipGetLastError();
        } else {
            if (a.drop_thr) A3T_L32(true, false, false);
            else A3T_L32(false, false, false);
        }
#undef A3T_L32
        a16.use_redo = 1;        // the same grid again: a block exits at once unless its row sums overflowed
    }
    (void)hipFuncSetAttribute((const void*)attn_fwd16_kernel<NDB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(attn_fwd16_kernel<NDB>, dim3(grid), dim3(512), lds, s, a16);
    return (int)hipGetLastError();
}



static bool attn_shape_ok(int dk, int T) { return dk % 32 == 0 && dk <= 192 && dk != 160 && T % 8 == 0 && T >= 8 && T <= 4096; }

extern "C" int a3t_attn_bwd_ds(const void* dctx, const void* ctx, const void* v, const void* probs, const float* rowscale,
                               void* ds, void* dbd, int B, int H, int T, int dk, int64_t ldo, int64_t ldkv, int64_t dbd_bsb,
                               int64_t dbd_bsh, float scale, float drop_p, uint32_t drop_key, int signed_probs, int64_t ds_bs, void* stream) {
    if (!attn_shape_ok(dk, T) || drop_p < 0.f || drop_p >= 1.f || !dctx || !ctx || !v || !probs || !rowscale || !ds)
        return A3T_EINVAL;
    if (ds_bs == 0) ds_bs = (int64_t)T * T;
    if (!al16(dctx) || !al16(ctx) || !al16(v) || !al16(probs) || !al16(ds) || !al16(dbd) || ldo % 8 || ldkv % 8 || dbd_bsb % 8 ||
        dbd_bsh % 8 || ds_bs % 8 || ds_bs < (int64_t)T * T)
        return A3T_EINVAL;
    if ((int64_t)B * H * T * T >= (1ll << 32)) return A3T_EINVAL;       // (the dropout counter is 32 bits, as in the forward)
    if (dbd_bsb == 0 && dbd_bsh == 0) dbd_bsb = (int64_t)H * T * T, dbd_bsh = (int64_t)T * T;
    DsArgs a = {};
    a.dctx = (const u16*)dctx, a.ctx = (const u16*)ctx, a.v = (const u16*)v, a.probs = (const u16*)probs, a.rowscale = rowscale;
    a.ds = (u16*)ds, a.dbd = (u16*)dbd, a.B = B, a.H = H, a.T = T, a.ldo = ldo, a.ldkv = ldkv, a.dbd_bsb = dbd_bsb, a.dbd_bsh = dbd_bsh;
    a.scale = scale;
    a.drop_thr = drop_p > 0.f ? (unsigned int)((double)drop_p * 4294967296.0) : 0u, a.drop_key = drop_key, a.drop_inv = 1.f / (1.f - drop_p);
    a.timing = (unsigned long long*)g_attn_timing_buf;
    a.signed_probs = (signed_probs && a.drop_thr) ? 1 : 0;
    a.ds_bs = ds_bs;
    // tasks = (128 queries) x (5 key tiles): several times more tasks than workgroup slots (2 per CU), so the last round is short
    constexpr int KT = 5;
    const int NS = (T + 31) / 32;
    const int64_t ntasks = (int64_t)B * H * ((T + 127) / 128) * ((NS + KT - 1) / KT);
    int64_t grid = (int64_t)attn_cus() * 2;
    if (grid > ntasks) grid = ntasks;
    hipStream_t s = (hipStream_t)stream;
#define A3T_DS2(NDB, DR, WD)                                                                                                         \
    do {                                                                                                                             \
        constexpr int lds = 3 * DT32<NDB>::BYTES + 4 * 32 * (KT * 64 + 16);                                                          \
        (void)hipFuncSetAttribute((const void*)attn_bwd_ds_kernel<NDB, DR, KT, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((attn_bwd_ds_kernel<NDB, DR, KT, WD>), dim3((unsigned)grid), dim3(256), lds, s, a);                        \
    } while (0)
#define A3T_DS1(NDB, DR)              \
    do {                              \
        if (a.dbd) A3T_DS2(NDB, DR, true); \
        else A3T_DS2(NDB, DR, false); \
    } while (0)
#define A3T_DS(NDB)                      \
    do {                                 \
        if (a.signed_probs) A3T_DS1(NDB, 2); \
        else if (a.drop_thr) A3T_DS1(NDB, 1); \
        else A3T_DS1(NDB, 0);            \
    } while (0)
    switch (dk / 32) {
        case 1: A3T_DS(1); break;
        case 2: A3T_DS(2); break;
        case 3: A3T_DS(3); break;
        case 4: A3T_DS(4); break;
        case 6: A3T_DS(6); break;
        default: return A3T_EINVAL;
    }
#undef A3T_DS1
#undef A3T_DS2
#undef A3T_DS
    return (int)hipGetLastError();
}

#if defined(A3T_ATTN_TIMING) || defined(A3T_DS_TIMING)
extern "C" void a3t_attn_timing_buf(void* ptr) { g_attn_timing_buf = ptr; }
#endif
__global__ __launch_bounds__(256) void attn_scale_rows_kernel(const u16* __restrict__ x, const float* __restrict__ rs,
                                                              u16* __restrict__ y, int H, int T, int dk, int64_t n4) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // group of 4 consecutive columns
    if (e >= n4) return;
    const int d4 = H * dk / 4;
    const int64_t r = e / d4;
    const int c = (int)(e - r * d4) * 4, h = c / dk;
    const int64_t b = r / T;
    const float f = rs[(b * H + h) * T + (r - b * T)];
    const uint2 v = *(const uint2*)(x + r * (H * dk) + c);
    st4_bf16(y + r * (H * dk) + c, io_bf2f(v.x & 0xffff) * f, io_bf2f(v.x >> 16) * f, io_bf2f(v.y & 0xffff) * f, io_bf2f(v.y >> 16) * f);
}

extern "C" int a3t_attn_scale_rows(const void* x, const float* rowscale, void* y, int B, int H, int T, int dk, void* stream) {
    if (dk % 4) return A3T_EINVAL;
    const int64_t n4 = (int64_t)B * T * H * dk / 4;
    hipLaunchKernelGGL(attn_scale_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)x,
                       rowscale, (u16*)y, H, T, dk, n4);
    return (int)hipGetLastError();
}

static int attn_fwd_impl(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                         const uint8_t* keymask, void* ctx, float* lse, void* probs, void* probs_drop, float* rowscale, int B,
                         int H, int T, int dk, int64_t ldq, int64_t ldkv, int64_t ldp, int64_t ldo, float scale, float drop_p,
                         uint32_t drop_key, const float* bias_u, const float* bias_v, void* stream) {
    if (!attn_shape_ok(dk, T)) return A3T_EINVAL;
    if ((bias_u == nullptr) != (bias_v == nullptr) || !al16(bias_u) || !al16(bias_v)) return A3T_EINVAL;
    if (!(al16(qu) && al16(qv) && al16(k) && al16(v) && al16(pos) && al16(ctx))) return A3T_EINVAL;
    if ((ldq % 8) || (ldkv % 8) || (ldp % 8) || (ldo % 4)) return A3T_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return A3T_EINVAL;
    AttnArgs a = {};
    a.qu = (const u16*)qu, a.qv = (const u16*)qv, a.k = (const u16*)k, a.v = (const u16*)v, a.pos = (const u16*)pos;
    a.keymask = keymask, a.ctx = (u16*)ctx, a.lse = lse, a.bu = bias_u, a.bv = bias_v;
    a.probs = (u16*)probs, a.pdrop = (u16*)probs_drop, a.rowscale = rowscale;
#ifdef A3T_ATTN_TIMING
    a.dbd = (u16*)g_attn_timing_buf;
#endif
    a.B = B, a.H = H, a.T = T, a.ldq = ldq, a.ldkv = ldkv, a.ldp = ldp, a.ldo = ldo, a.scale = scale;
    a.drop_thr = drop_p > 0.f ? (unsigned int)((double)drop_p * 4294967296.0) : 0u;
    a.drop_key = drop_key, a.drop_inv = 1.f / (1.f - drop_p);
    hipStream_t s = (hipStream_t)stream;
    switch (dk / 32) {
        case 1: return launch_fwd16<1>(a, s);
        case 2: return launch_fwd16<2>(a, s);
        case 3: return launch_fwd16<3>(a, s);
        case 4: return launch_fwd16<4>(a, s);
        case 6: return launch_fwd16<6>(a, s);
    }
    return A3T_EINVAL;
}

extern "C" int a3t_attn_fwd(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                            const uint8_t* keymask, void* ctx, float* lse, int B, int H, int T, int dk, int64_t ldq,
                            int64_t ldkv, int64_t ldp, int64_t ldo, float scale, float drop_p, uint32_t drop_key,
                            const float* bias_u, const float* bias_v, void* stream) {
    return attn_fwd_impl(qu, qv, k, v, pos, keymask, ctx, lse, nullptr, nullptr, nullptr, B, H, T, dk, ldq, ldkv, ldp, ldo, scale,
                         drop_p, drop_key, bias_u, bias_v, stream);
}

extern "C" int a3t_attn_fwd_train(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                                  const uint8_t* keymask, void* ctx, float* lse, void* probs, void* probs_drop,
                                  float* rowscale, int B, int H, int T, int dk, int64_t ldq, int64_t ldkv, int64_t ldp,
                                  int64_t ldo, float scale, float drop_p, uint32_t drop_key, const float* bias_u,
                                  const float* bias_v, void* stream) {
    if (!probs || !rowscale || (T % 8)) return A3T_EINVAL;      // (drop_p > 0 without probs_drop: the sign-tagged single tensor)
    if (((uintptr_t)probs & 7) || ((uintptr_t)probs_drop & 7)) return A3T_EINVAL;
    return attn_fwd_impl(qu, qv, k, v, pos, keymask, ctx, lse, probs, drop_p > 0.f ? probs_drop : nullptr, rowscale, B, H, T, dk, ldq,
                         ldkv, ldp, ldo, scale, drop_p, drop_key, bias_u, bias_v, stream);
}
