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
    const u16* k;         // keys   [B*T][ldkv]
    const u16* v;         // values [B*T][ldkv]
    const u16* pos;       // linear_pos(pos_emb) [T][ldp]
    const uint8_t* keymask;   // [B][T], 1 = valid key
    u16* ctx;             // out [B*T][ldo]
    float* lse;           // out [B][H][T]: log sum exp of the scaled scores (+inf for a fully masked row)
    // backward only
    const u16* dctx;      // [B*T][ldo]
    const float* delta;   // [B][H][T]: sum_d dctx * ctx
    u16* dqu;             // [B*T][ldq]
    u16* dqvl;            // gradient of (q+v)[i] through the x < T half of the band
    u16* dqvu;            // gradient of (q+v)[i+1] through the x > T half, written AT row i+1
    u16* dbd;             // compact dBD [B][H][T][T] (input of the d linear_pos GEMM)
    u16* dk;              // [B*T][lddkv]
    u16* dv;
    int B, H, T;
    int64_t ldq, ldkv, ldp, ldo, lddkv;
    float scale;
    unsigned int drop_thr, drop_key;
    float drop_inv;
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
__device__ __forceinline__ bf16x8 ld_frag_g(const u16* p, bool ok) {
    if (!ok) return zero_frag();
    return *(const bf16x8*)p;
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
            r[n] = make_uint4(0, 0, 0, 0);
            if (c < 32 * CPR) {
                const int64_t g = rowmap(row);
                if (g >= 0) r[n] = *(const uint4*)(base + g * ld + ch * 8);
            }
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

template <int NDB>
__global__ __launch_bounds__(256, 1) void attn_fwd_kernel(AttnArgs p) {
    using TL = Tile<NDB>;
    constexpr int DK = TL::DK, KS = DK / 16, RSB = TL::RSB, TB = TL::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kt = smem;
    unsigned char* Vt = smem + TB;
    unsigned char* Pr = smem + 2 * TB;                   // 5 ring slots
    float* sc = (float*)(smem + 7 * TB);                 // [4 waves][32][SC_LD]

    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, NQB = (T + 127) / 128, NS = (T + 31) / 32;
    int wi = blockIdx.x;
    {   // workgroup b runs on XCD b % 8: give every XCD a contiguous run of (batch, head) pairs (K / V / P stay in its L2)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wi & 7;
        wi = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wi >> 3);
    }
    const int bh = wi / NQB, qb = wi - bh * NQB;
    const int b = bh / p.H, h = bh - b * p.H;
    const int Q0 = qb * 128, q0 = Q0 + 32 * w;
    const int X0 = T - 32 - Q0;                          // band base of (wave 0, key block 0)
    const u16* quB = p.qu + (int64_t)b * T * p.ldq + h * DK;
    const u16* qvB = p.qv + (int64_t)b * T * p.ldq + h * DK;
    const u16* kB = p.k + (int64_t)b * T * p.ldkv + h * DK;
    const u16* vB = p.v + (int64_t)b * T * p.ldkv + h * DK;
    const u16* pB = p.pos + h * DK;
    const uint8_t* mkB = p.keymask + (int64_t)b * T;
    float* scw = sc + w * 32 * SC_LD;

    auto prow = [&](int u, int r) -> int64_t {           // Pext row of band block u, row r
        const int x = X0 + 32 * (u - 3) + r;
        if (x >= 0 && x < T) return x;
        if (x > T && x - T - 1 < T) return x - T - 1;
        return -1;
    };
    auto krow = [&](int s, int r) -> int64_t { return (32 * s + r < T) ? 32 * s + r : -1; };

    // ---- prologue: query fragments (B operands, registers for the whole key loop), first tiles ------------------
    // (q+v): band blocks arrive in increasing x, so a wave works with (q+v)[i] until its first block at x >= T and
    // with (q+v)[i+1] from then on: ONE fragment set, reloaded once (wave-uniform) at the switch.
    bf16x8 fqu[KS], fqv[KS];
    bool upper = false;
    {
        const int i = q0 + lr;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int off = 16 * kk + 8 * lh;
            fqu[kk] = ld_frag_g(quB + (int64_t)i * p.ldq + off, i < T);
            fqv[kk] = ld_frag_g(qvB + (int64_t)i * p.ldq + off, i < T);
        }
    }
    {
        TL tk, tv, tp[5];
        tk.load(kB, p.ldkv, tid, [&](int r) { return krow(0, r); });
        tv.load(vB, p.ldkv, tid, [&](int r) { return krow(0, r); });
#pragma unroll
        for (int u = 0; u < 5; ++u) tp[u].load(pB, p.ldp, tid, [&](int r) { return prow(u, r); });
        tk.commit(Kt, tid);
        tv.commit(Vt, tid);
#pragma unroll
        for (int u = 0; u < 5; ++u) tp[u].commit(Pr + u * TB, tid);
    }
    __syncthreads();

    auto band = [&](int u) {      // band block u of this wave's queries -> scratch half (u & 1)
        const unsigned char* slot = Pr + (u % 5) * TB;
        const bool useU = (u - 3) * 32 >= 32 + Q0;       // block base xb >= T: the (q+v)[i+1] half of the band
        if (useU && !upper) {
            upper = true;
            const int i1 = q0 + lr + 1;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) fqv[kk] = ld_frag_g(qvB + (int64_t)i1 * p.ldq + 16 * kk + 8 * lh, i1 < T);
        }
        f32x16 acc = zero16();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<RSB>(slot, lr, 2 * kk + lh), fqv[kk], acc, 0, 0, 0);
        float* row = scw + lr * SC_LD + 32 * (u & 1) + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float4*)(row + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    };
    band(3 - w);

    f32x16 O[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d) O[d] = zero16();
    float m_run = -1e30f, l_run = 0.f;
    const float NEG_INF = -__builtin_inff();
    const unsigned int ibase = (unsigned int)(((int64_t)bh * T + (q0 + lr)) * T);   // RNG index of (i, key 0)

    for (int s = 0; s < NS; ++s) {
        TL tk, tv, tp;
        const bool more = s + 1 < NS;
        if (more) {   // next tiles: global loads in flight during this tile's compute
            tk.load(kB, p.ldkv, tid, [&](int r) { return krow(s + 1, r); });
            tv.load(vB, p.ldkv, tid, [&](int r) { return krow(s + 1, r); });
            tp.load(pB, p.ldp, tid, [&](int r) { return prow(s + 5, r); });
        }
        // ---- S^T = K (q+u)^T --------------------------------------------------------------------------------------
        f32x16 sa = zero16();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<RSB>(Kt, lr, 2 * kk + lh), fqu[kk], sa, 0, 0, 0);
        const int u = s - w + 3;
        band(u + 1);
        // ---- key mask of the block (wave-uniform word) -----------------------------------------------------------
        const int jl = 32 * s + lr;
        const bool kvalid = (jl < T) && (mkB[jl < T ? jl : 0] != 0);
        const unsigned int vm = (unsigned int)__ballot(kvalid);
        // ---- skewed read of the band + scale + mask ----------------------------------------------------------------
        const int c0 = 32 * (u & 1) + 31 - lr;
        const float* srow = scw + lr * SC_LD;
        float sv[16];
        float mloc = NEG_INF;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float bd = srow[(c0 + kk) & 63];
            sv[r] = ((vm >> kk) & 1u) ? (sa[r] + bd) * p.scale : NEG_INF;
            mloc = fmaxf(mloc, sv[r]);
        }
        // ---- online softmax (a lane owns one query; its other 16 keys sit in lane ^ 32) ---------------------------
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m_run, mloc);
        const float alpha = __expf(m_run - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sv[r] = __expf(sv[r] - mnew);
            psum += sv[r];
        }
        l_run = l_run * alpha + psum;
        m_run = mnew;
        if (!__all(alpha == 1.f)) {
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
        }
        // ---- attention dropout (counter RNG on the (b, h, i, j) index of the probability) ------------------------
        if (p.drop_thr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bool kp[4];
                rng_keep4(p.drop_key, ibase + (unsigned int)(32 * s + 8 * g + 4 * lh), p.drop_thr, kp);
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[4 * g + e] = kp[e] ? sv[4 * g + e] * p.drop_inv : 0.f;
            }
        }
        const bf16x8 pf0 = pack_frag(sv), pf1 = pack_frag(sv + 8);
        // ---- O^T += V^T P^T ------------------------------------------------------------------------------------------
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<RSB>(Vt, 0, 32 * d, lane), pf0, O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<RSB>(Vt, 16, 32 * d, lane), pf1, O[d], 0, 0, 0);
        }
        __syncthreads();          // every wave is done with K / V and with ring slot (s + 5) % 5 (block s, read in step s-1)
        if (more) {
            tk.commit(Kt, tid);
            tv.commit(Vt, tid);
            tp.commit(Pr + ((s + 5) % 5) * TB, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: normalise, store this lane's query row -------------------------------------------------------------
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float invl = l > 0.f ? 1.f / l : 0.f;
    const int i = q0 + lr;
    if (i < T) {
        u16* o = p.ctx + ((int64_t)b * T + i) * p.ldo + h * DK + 4 * lh;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 v2;
                v2.x = io_pack2(O[d][4 * g] * invl, O[d][4 * g + 1] * invl);
                v2.y = io_pack2(O[d][4 * g + 2] * invl, O[d][4 * g + 3] * invl);
                *(uint2*)(o + 32 * d + 8 * g) = v2;
            }
        if (lh == 0) p.lse[(int64_t)bh * T + i] = l > 0.f ? m_run + __logf(l) : __builtin_inff();
    }
}

static inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

template <int NDB>
static int launch_fwd(const AttnArgs& a, hipStream_t s) {
    constexpr int TB = Tile<NDB>::BYTES;
    constexpr int lds = 7 * TB + 4 * 32 * SC_LD * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<NDB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    const int nqb = (a.T + 127) / 128;
    hipLaunchKernelGGL(attn_fwd_kernel<NDB>, dim3((unsigned)(a.B * a.H * nqb)), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

extern "C" int a3t_attn_fwd(const void* qu, const void* qv, const void* k, const void* v, const void* pos,
                            const uint8_t* keymask, void* ctx, float* lse, int B, int H, int T, int dk, int64_t ldq,
                            int64_t ldkv, int64_t ldp, int64_t ldo, float scale, float drop_p, uint32_t drop_key,
                            void* stream) {
    if (dk % 32 != 0 || dk > 192 || dk == 160 || T % 8 != 0 || T < 8) return A3T_EINVAL;
    if (!(al16(qu) && al16(qv) && al16(k) && al16(v) && al16(pos) && al16(ctx))) return A3T_EINVAL;
    if ((ldq % 8) || (ldkv % 8) || (ldp % 8) || (ldo % 4)) return A3T_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return A3T_EINVAL;
    AttnArgs a = {};
    a.qu = (const u16*)qu, a.qv = (const u16*)qv, a.k = (const u16*)k, a.v = (const u16*)v, a.pos = (const u16*)pos;
    a.keymask = keymask, a.ctx = (u16*)ctx, a.lse = lse;
    a.B = B, a.H = H, a.T = T, a.ldq = ldq, a.ldkv = ldkv, a.ldp = ldp, a.ldo = ldo, a.scale = scale;
    a.drop_thr = drop_p > 0.f ? (unsigned int)((double)drop_p * 4294967296.0) : 0u;
    a.drop_key = drop_key, a.drop_inv = 1.f / (1.f - drop_p);
    hipStream_t s = (hipStream_t)stream;
    switch (dk / 32) {
        case 1: return launch_fwd<1>(a, s);
        case 2: return launch_fwd<2>(a, s);
        case 3: return launch_fwd<3>(a, s);
        case 4: return launch_fwd<4>(a, s);
        case 6: return launch_fwd<6>(a, s);
    }
    return A3T_EINVAL;
}
