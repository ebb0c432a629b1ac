// gemm.hip -- the one dense-contraction kernel family of the A3T path (gfx950 / MI355X).
//
// Covers torch.nn.Linear, Conv1d-as-implicit-im2col (FFN k=3, postnet k=5, PWG dilated k=3),
// the attention bmm's and every data / weight gradient of those (NT, NN, TN operand layouts).
// Two arithmetic modes:
//   * A3T_F32  : v_mfma_f32_32x32x2_f32  -- exact fp32 (bitwise an fmaf chain), the parity path
//   * A3T_BF16 : v_mfma_f32_32x32x16_bf16 -- bf16 operands staged through LDS, fp32 accumulate
// Tiling is wave64-native: 256 threads = 4 waves in a 2x2 arrangement, each wave owns a 64x64
// output sub-tile = 2x2 MFMA 32x32 accumulators (64 AGPR-class registers), block tile 128x128.
// Operands are register-staged (global -> VGPR -> LDS) so the loader can do the im2col row
// shift / utterance-boundary zeroing, the fp32->bf16 conversion and the transposing store for
// reduction-strided operands; LDS is double buffered, one barrier per K-tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#include "gemm_common.h"

int a3t_gemm_bf16_glds(const GP& p, int batch, bool AK, bool BKC, hipStream_t stream);

#include <stdarg.h>
#include <stdio.h>
static thread_local char g_last_kernel[128] = "";
void a3t_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}
extern "C" const char* a3t_gemm_last_kernel(void) { return g_last_kernel; }

// =============================================================================================
// fp32 MFMA kernel: LDS tiles are k-major ([BK][BM+4]) so a lane's single-float fragment read is
// conflict free (lanes 0-31 -> 32 consecutive rows, lanes 32-63 -> next k).
// =============================================================================================
template <bool AK, bool BKC, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GP p) {
    constexpr int BM = 128, BN = 128, BK = 16, LD = 132;
    __shared__ __attribute__((aligned(16))) float smem[4 * BK * LD];
    float(*As)[BK][LD] = (float(*)[BK][LD])smem;
    float(*Bs)[BK][LD] = (float(*)[BK][LD])(smem + 2 * BK * LD);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tn = blockIdx.x % p.tiles_n, tm = blockIdx.x / p.tiles_n;
    const int ks = blockIdx.y % p.splitk, bz = blockIdx.y / p.splitk;
    const int z0 = bz / p.batch_inner, z1 = bz % p.batch_inner;
    const float* A = (const float*)p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const float* B = (const float*)p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int64_t zoff = z0 * p.c_bs0 + z1 * p.c_bs1;

    const int ktiles = (p.K + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;

    float4 ra[2], rb[2];
    int a_t[2];
    if (AK) {
        for (int ps = 0; ps < 2; ++ps) {
            int m = tm * BM + (tid >> 2) + ps * 64;
            a_t[ps] = (p.taps > 1) ? (m % p.Tseq) : 0;
        }
    }

    auto load_tiles = [&](int k0) {
        if (AK) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                int m = tm * BM + r + ps * 64, kg = k0 + kq * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < p.M) {
                    if (VEC) {
                        if (kg < p.K) {
                            const float* q = a_ptr_k(p, A, m, a_t[ps], kg);
                            if (q) v = *(const float4*)q;
                        }
                    } else {
                        float t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            t[i] = 0.f;
                            if (kg + i < p.K) {
                                const float* q = a_ptr_k(p, A, m, a_t[ps], kg + i);
                                if (q) t[i] = *q;
                            }
                        }
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                ra[ps] = v;
            }
        } else {
            const int kr = tid >> 5, mq = tid & 31;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                int k = k0 + kr + ps * 8, m0 = tm * BM + mq * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) {
                    const float* q = A + (int64_t)k * p.a_cs + m0;
                    if (VEC && m0 + 3 < p.M) {
                        v = *(const float4*)q;
                    } else {
                        if (m0 + 0 < p.M) v.x = q[0];
                        if (m0 + 1 < p.M) v.y = q[1];
                        if (m0 + 2 < p.M) v.z = q[2];
                        if (m0 + 3 < p.M) v.w = q[3];
                    }
                }
                ra[ps] = v;
            }
        }
        if (BKC) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                int n = tn * BN + r + ps * 64, kg = k0 + kq * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N) {
                    if (VEC) {
                        if (kg < p.K) {
                            int64_t koff;
                            if (b_koff(p, kg, koff)) v = *(const float4*)(B + (int64_t)n * p.b_rs + koff);
                        }
                    } else {
                        float t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            t[i] = 0.f;
                            int64_t koff;
                            if (kg + i < p.K && b_koff(p, kg + i, koff)) t[i] = B[(int64_t)n * p.b_rs + koff];
                        }
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                rb[ps] = v;
            }
        } else {
            const int kr = tid >> 5, nq = tid & 31;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                int k = k0 + kr + ps * 8, n0 = tn * BN + nq * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                int64_t koff;
                if (k < p.K && b_koff(p, k, koff)) {
                    const float* q = B + koff + n0;
                    if (VEC && n0 + 3 < p.N) {
                        v = *(const float4*)q;
                    } else {
                        if (n0 + 0 < p.N) v.x = q[0];
                        if (n0 + 1 < p.N) v.y = q[1];
                        if (n0 + 2 < p.N) v.z = q[2];
                        if (n0 + 3 < p.N) v.w = q[3];
                    }
                }
                rb[ps] = v;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        if (AK) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                As[buf][kq * 4 + 0][r + ps * 64] = ra[ps].x;
                As[buf][kq * 4 + 1][r + ps * 64] = ra[ps].y;
                As[buf][kq * 4 + 2][r + ps * 64] = ra[ps].z;
                As[buf][kq * 4 + 3][r + ps * 64] = ra[ps].w;
            }
        } else {
            const int kr = tid >> 5, mq = tid & 31;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) *(float4*)&As[buf][kr + ps * 8][mq * 4] = ra[ps];
        }
        if (BKC) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                Bs[buf][kq * 4 + 0][r + ps * 64] = rb[ps].x;
                Bs[buf][kq * 4 + 1][r + ps * 64] = rb[ps].y;
                Bs[buf][kq * 4 + 2][r + ps * 64] = rb[ps].z;
                Bs[buf][kq * 4 + 3][r + ps * 64] = rb[ps].w;
            }
        } else {
            const int kr = tid >> 5, nq = tid & 31;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) *(float4*)&Bs[buf][kr + ps * 8][nq * 4] = rb[ps];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (w >> 1) * 64, wn = (w & 1) * 64, lr = lane & 31, lk = lane >> 5;
    load_tiles(kt0 * BK);
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1 < kt1);
        if (more) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a0 = As[buf][kk * 2 + lk][wm + lr], a1 = As[buf][kk * 2 + lk][wm + 32 + lr];
            float b0 = Bs[buf][kk * 2 + lk][wn + lr], b1 = Bs[buf][kk * 2 + lk][wn + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = tm * BM + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int col = tn * BN + wn + j * 32 + lr;
                epilogue_store(p, zoff, row, col, acc[i][j][r], ks);
            }
}

// =============================================================================================
// bf16 MFMA kernel: LDS tiles are row-major bf16 [128][BK+8] (k contiguous, 80-byte rows ->
// conflict-free ds_read_b128 fragment reads); sources may be fp32 (converted in the loader) or bf16.
// =============================================================================================
template <typename TA, typename TB, bool AK, bool BKC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GP p) {
    constexpr int BM = 128, BN = 128, BK = 32, LD = 40;
    __shared__ __attribute__((aligned(16))) unsigned short smem[4 * BM * LD];
    unsigned short(*As)[BM][LD] = (unsigned short(*)[BM][LD])smem;
    unsigned short(*Bs)[BN][LD] = (unsigned short(*)[BN][LD])(smem + 2 * BM * LD);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tn = blockIdx.x % p.tiles_n, tm = blockIdx.x / p.tiles_n;
    const int ks = blockIdx.y % p.splitk, bz = blockIdx.y / p.splitk;
    const int z0 = bz / p.batch_inner, z1 = bz % p.batch_inner;
    const TA* A = (const TA*)p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const TB* B = (const TB*)p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int64_t zoff = z0 * p.c_bs0 + z1 * p.c_bs1;

    const int ktiles = (p.K + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;

    // staged registers: k-contig: 2 chunks of 8 bf16 (uint4); row-contig: 4x4 micro tile -> 4 x uint2
    uint4 sa[2], sb[2];
    int a_t[2];
    if (AK) {
        for (int ps = 0; ps < 2; ++ps) {
            int m = tm * BM + (tid >> 2) + ps * 64;
            a_t[ps] = (p.taps > 1) ? (m % p.Tseq) : 0;
        }
    }
    auto pack8 = [](const float* f) {
        uint4 u;
        u.x = io_pack2(f[0], f[1]);
        u.y = io_pack2(f[2], f[3]);
        u.z = io_pack2(f[4], f[5]);
        u.w = io_pack2(f[6], f[7]);
        return u;
    };
    auto load8 = [&](const float* q) {  // 8 consecutive fp32 -> 8 bf16
        float4 v0 = *(const float4*)q, v1 = *(const float4*)(q + 4);
        float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        return pack8(f);
    };
    auto load8h = [&](const unsigned short* q) { return *(const uint4*)q; };

    auto load_k = [&](auto* base, bool isA, int row, int kg, int tpos, uint4& out) {
        out = make_uint4(0, 0, 0, 0);
        if (isA) {
            if (row < p.M && kg < p.K) {
                auto* q = a_ptr_k(p, base, row, tpos, kg);
                if (q) {
                    if constexpr (sizeof(*base) == 4)
                        out = load8((const float*)q);
                    else
                        out = load8h((const unsigned short*)q);
                }
            }
        } else {
            int64_t koff;
            if (row < p.N && kg < p.K && b_koff(p, kg, koff)) {
                auto* q = base + (int64_t)row * p.b_rs + koff;
                if constexpr (sizeof(*base) == 4)
                    out = load8((const float*)q);
                else
                    out = load8h((const unsigned short*)q);
            }
        }
    };
    // row-contiguous operand: 4 rows x 4 k micro tile, transposed in registers
    auto load_r = [&](auto* base, bool isA, int r0, int rmax, int k0, uint4& o01, uint4& o23) {
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int k = k0 + i;
            v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
            int64_t koff;
            bool ok = k < p.K;
            if (isA)
                koff = (int64_t)k * p.a_cs;
            else
                ok = ok && b_koff(p, k, koff);
            if (ok && r0 < rmax) {
                auto* q = base + koff + r0;
                if (r0 + 3 < rmax) {
                    if constexpr (sizeof(*base) == 4) {
                        float4 t = *(const float4*)q;
                        v[i][0] = t.x, v[i][1] = t.y, v[i][2] = t.z, v[i][3] = t.w;
                    } else {
                        uint2 t = *(const uint2*)q;
                        v[i][0] = bf2f(t.x & 0xffff), v[i][1] = bf2f(t.x >> 16);
                        v[i][2] = bf2f(t.y & 0xffff), v[i][3] = bf2f(t.y >> 16);
                    }
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (r0 + j < rmax) v[i][j] = ldf(q + j);
                }
            }
        }
        // row j gets (v[0][j], v[1][j], v[2][j], v[3][j]) = 4 consecutive k
        o01.x = io_pack2(v[0][0], v[1][0]);
        o01.y = io_pack2(v[2][0], v[3][0]);
        o01.z = io_pack2(v[0][1], v[1][1]);
        o01.w = io_pack2(v[2][1], v[3][1]);
        o23.x = io_pack2(v[0][2], v[1][2]);
        o23.y = io_pack2(v[2][2], v[3][2]);
        o23.z = io_pack2(v[0][3], v[1][3]);
        o23.w = io_pack2(v[2][3], v[3][3]);
    };

    auto load_tiles = [&](int k0) {
        if (AK) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) load_k(A, true, tm * BM + r + ps * 64, k0 + kq * 8, a_t[ps], sa[ps]);
        } else {
            load_r(A, true, tm * BM + (tid & 31) * 4, p.M, k0 + (tid >> 5) * 4, sa[0], sa[1]);
        }
        if (BKC) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) load_k(B, false, tn * BN + r + ps * 64, k0 + kq * 8, 0, sb[ps]);
        } else {
            load_r(B, false, tn * BN + (tid & 31) * 4, p.N, k0 + (tid >> 5) * 4, sb[0], sb[1]);
        }
    };
    auto store_tiles = [&](int buf) {
        if (AK) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) *(uint4*)&As[buf][r + ps * 64][kq * 8] = sa[ps];
        } else {
            const int m0 = (tid & 31) * 4, kq = tid >> 5;
            *(uint2*)&As[buf][m0 + 0][kq * 4] = make_uint2(sa[0].x, sa[0].y);
            *(uint2*)&As[buf][m0 + 1][kq * 4] = make_uint2(sa[0].z, sa[0].w);
            *(uint2*)&As[buf][m0 + 2][kq * 4] = make_uint2(sa[1].x, sa[1].y);
            *(uint2*)&As[buf][m0 + 3][kq * 4] = make_uint2(sa[1].z, sa[1].w);
        }
        if (BKC) {
            const int r = tid >> 2, kq = tid & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) *(uint4*)&Bs[buf][r + ps * 64][kq * 8] = sb[ps];
        } else {
            const int n0 = (tid & 31) * 4, kq = tid >> 5;
            *(uint2*)&Bs[buf][n0 + 0][kq * 4] = make_uint2(sb[0].x, sb[0].y);
            *(uint2*)&Bs[buf][n0 + 1][kq * 4] = make_uint2(sb[0].z, sb[0].w);
            *(uint2*)&Bs[buf][n0 + 2][kq * 4] = make_uint2(sb[1].x, sb[1].y);
            *(uint2*)&Bs[buf][n0 + 3][kq * 4] = make_uint2(sb[1].z, sb[1].w);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (w >> 1) * 64, wn = (w & 1) * 64, lr = lane & 31, lk = lane >> 5;
    load_tiles(kt0 * BK);
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1 < kt1);
        if (more) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 a0 = *(const bf16x8*)&As[buf][wm + lr][kk * 16 + lk * 8];
            bf16x8 a1 = *(const bf16x8*)&As[buf][wm + 32 + lr][kk * 16 + lk * 8];
            bf16x8 b0 = *(const bf16x8*)&Bs[buf][wn + lr][kk * 16 + lk * 8];
            bf16x8 b1 = *(const bf16x8*)&Bs[buf][wn + 32 + lr][kk * 16 + lk * 8];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = tm * BM + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int col = tn * BN + wn + j * 32 + lr;
                epilogue_store(p, zoff, row, col, acc[i][j][r], ks);
            }
}

// =============================================================================================
// host dispatch
// =============================================================================================
static inline bool al(const void* p, int b) { return ((uintptr_t)p % b) == 0; }
static inline bool m4(int64_t v, int m) { return (v % m) == 0; }

extern "C" int a3t_gemm(const a3t_gemm_desc* d, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !d->A || !d->B || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0) return A3T_EINVAL;
    GP p;
    p.A = d->A, p.B = d->B, p.C = d->C, p.bias = d->bias, p.R = d->R, p.S = d->S;
    p.M = d->M, p.N = d->N, p.K = d->K;
    p.taps = d->taps < 1 ? 1 : d->taps;
    const bool wgrad_fused = (d->taps > 1 && d->a_cs != 1);   // TN conv weight gradient, columns = (tap, c)
    if (!wgrad_fused && (p.K % p.taps)) return A3T_EINVAL;
    p.Kc = wgrad_fused ? p.K : p.K / p.taps;
    p.a_rs = d->a_rs, p.a_cs = d->a_cs, p.b_rs = d->b_rs, p.b_cs = d->b_cs, p.b_ts = d->b_ts, p.c_rs = d->c_rs;
    int batch = d->batch < 1 ? 1 : d->batch;
    p.batch_inner = d->batch_inner < 1 ? 1 : d->batch_inner;
    p.a_bs0 = d->a_bs0, p.a_bs1 = d->a_bs1, p.b_bs0 = d->b_bs0, p.b_bs1 = d->b_bs1;
    p.c_bs0 = d->c_bs0, p.c_bs1 = d->c_bs1;
    p.pad = d->pad, p.dil = d->dil < 1 ? 1 : d->dil, p.Tseq = d->Tseq, p.kshift = d->kshift;
    p.kshift_mode = (p.taps == 1 && d->Tseq > 0) ? 1 : 0;
    if (p.taps > 1 && p.Tseq <= 0) return A3T_EINVAL;
    p.alpha = d->alpha, p.act = d->act, p.accumulate = d->accumulate;
    p.sole_writer = 0;
    if (p.accumulate == A3T_ACC_SOLE) p.accumulate = A3T_ACC_ATOMIC, p.sole_writer = 1;
    p.splitk = d->splitk < 1 ? 1 : d->splitk;
    if (p.splitk > 1 && p.accumulate != A3T_ACC_ATOMIC) return A3T_EINVAL;
    p.c_dtype = d->c_dtype;
    p.s_dtype = d->s_dtype;
    p.epi_vec = 0;
    p.colsum = d->colsum, p.colsum_bs1 = d->colsum_bs1, p.colsum_scale = d->colsum_scale;
    p.colsum_slots = d->colsum_slots > 1 ? d->colsum_slots : 1, p.colsum_ss = d->colsum_ss;
    if (d->drop_p < 0.f || d->drop_p >= 1.f) return A3T_EINVAL;
    p.drop_key = d->drop_key;
    p.drop_thr = (unsigned int)((double)d->drop_p * 4294967296.0);
    p.drop_inv = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 0.f;
    p.keep_out = (unsigned char*)d->keep_out, p.keep_in = (const unsigned char*)d->keep_in;
    p.a_bytes = p.b_bytes = 0;
    p.slab = nullptr;
    p.a_signmask = d->a_signmask ? 1 : 0;
    p.A2 = d->A2, p.B2 = d->B2, p.b2_cs = d->b2_cs, p.b2_bs0 = d->b2_bs0, p.b2_bs1 = d->b2_bs1, p.colsum2 = d->colsum2;
    p.a2_rs = d->a2_rs, p.a_unaligned = d->a_unaligned;
    if ((p.A2 == nullptr) != (p.B2 == nullptr) || (!p.A2 && (p.colsum2 || p.a2_rs || (p.a_unaligned & 2))) || (p.a_unaligned & ~3)) return A3T_EINVAL;
    if ((p.a_unaligned & 1) && (d->compute != A3T_BF16 || d->a_dtype != A3T_BF16 || d->b_dtype != A3T_BF16 || p.taps > 1 || d->Tseq > 0 ||
                                ((uintptr_t)p.A & 1)))
        return A3T_EINVAL;
    p.keep_layout = (p.keep_out || p.keep_in) ? d->keep_layout : 0;
    if (p.keep_layout != 0 && p.keep_layout != 1) return A3T_EINVAL;
    const bool keep = p.keep_out || p.keep_in;
    if (keep && (d->compute != A3T_BF16 || d->a_dtype != A3T_BF16 || d->b_dtype != A3T_BF16)) return A3T_EINVAL;
    // bf16 C accumulates only by plain read-modify-write (A3T_ACC_ADD, one launch per element at a time): no bf16 atomics
    if (p.c_dtype == A3T_BF16 && p.accumulate != A3T_ACC_STORE && (p.accumulate != A3T_ACC_ADD || p.splitk > 1)) return A3T_EINVAL;
    const bool AK = (d->a_cs == 1), BKC = (d->b_cs == 1);
    if (!AK && d->a_rs != 1) return A3T_EINVAL;
    if (!BKC && d->b_rs != 1) return A3T_EINVAL;
    if (p.taps > 1 && !AK) {  // fused conv weight gradient: only the direct-to-LDS bf16 kernel implements it
        if (d->compute != A3T_BF16 || d->a_dtype != A3T_BF16 || d->b_dtype != A3T_BF16 || d->b_cs == 1 || p.a_signmask || p.A2) return A3T_EINVAL;
        p.tiles_n = (p.N + 127) / 128;
        int rc = a3t_gemm_bf16_glds(p, batch, false, false, stream);
        return rc >= 0 ? rc : A3T_EINVAL;
    }
    if (!AK && BKC) return A3T_EINVAL;         // (TT layout is never needed on this path)
    p.tiles_n = (p.N + 127) / 128;
    const int tiles_m = (p.M + 127) / 128;
    dim3 grid((unsigned)(p.tiles_n * tiles_m), (unsigned)(batch * p.splitk)), block(256);

    if (d->compute == A3T_F32) {
        if (d->a_dtype != A3T_F32 || d->b_dtype != A3T_F32 || d->colsum || p.a_signmask || p.A2) return A3T_EINVAL;
        bool vec = al(p.A, 16) && al(p.B, 16) && m4(p.a_bs0, 4) && m4(p.a_bs1, 4) && m4(p.b_bs0, 4) && m4(p.b_bs1, 4);
        if (AK)
            vec = vec && m4(p.a_rs, 4) && m4(p.K, 4) && m4(p.Kc, 4);
        else
            vec = vec && m4(p.a_cs, 4);
        if (BKC)
            vec = vec && m4(p.b_rs, 4) && m4(p.K, 4) && m4(p.Kc, 4) && m4(p.b_ts, 4);
        else
            vec = vec && m4(p.b_cs, 4) && m4(p.b_ts, 4);
#define LAUNCH_F32(ak, bk)                                                                    \
    do {                                                                                      \
        if (vec)                                                                              \
            hipLaunchKernelGGL((gemm_f32_kernel<ak, bk, true>), grid, block, 0, stream, p);   \
        else                                                                                  \
            hipLaunchKernelGGL((gemm_f32_kernel<ak, bk, false>), grid, block, 0, stream, p);  \
    } while (0)
        if (AK && BKC)
            LAUNCH_F32(true, true);
        else if (AK && !BKC)
            LAUNCH_F32(true, false);
        else
            LAUNCH_F32(false, false);
#undef LAUNCH_F32
        a3t_note_kernel("gemm_f32_kernel<%s, %s, %s>", AK ? "true" : "false", BKC ? "true" : "false", vec ? "true" : "false");
        return (int)hipGetLastError();
    }
    if (d->compute != A3T_BF16) return A3T_EINVAL;
    if (d->a_dtype == A3T_BF16 && d->b_dtype == A3T_BF16) {
        int rc = a3t_gemm_bf16_glds(p, batch, AK, BKC, stream);   // direct-to-LDS production kernel
        if (rc >= 0) return rc;                                   // -1: alignment contract not met
    }
    if (keep || p.a_signmask || p.A2 || p.a_unaligned) return A3T_EINVAL;        // keep-bit images only exist in the 8-phase kernel, sign masks in the direct-to-LDS TN kernels
    if (d->colsum) return A3T_EINVAL;   // fused column sums live in the direct-to-LDS kernel's epilogue
    {
        const int ea = d->a_dtype == A3T_BF16 ? 2 : 4, eb = d->b_dtype == A3T_BF16 ? 2 : 4;
        bool ok = al(p.A, 16) && al(p.B, 16);
        // k-contiguous operands are read in chunks of 8 elements, row-contiguous in chunks of 4
        if (AK)
            ok = ok && m4(p.a_rs, 8) && m4(p.K, 8) && m4(p.Kc, 8) && m4(p.a_bs0, 8) && m4(p.a_bs1, 8);
        else
            ok = ok && m4(p.a_cs, 4) && m4(p.a_bs0, 4) && m4(p.a_bs1, 4);
        if (BKC)
            ok = ok && m4(p.b_rs, 8) && m4(p.K, 8) && m4(p.Kc, 8) && m4(p.b_ts, 8) && m4(p.b_bs0, 8) && m4(p.b_bs1, 8);
        else
            ok = ok && m4(p.b_cs, 4) && m4(p.b_ts, 4) && m4(p.b_bs0, 4) && m4(p.b_bs1, 4);
        (void)ea, (void)eb;
        if (!ok) return A3T_EINVAL;
    }
#define LAUNCH_BF(TA, TB)                                                                              \
    do {                                                                                               \
        if (AK && BKC)                                                                                 \
            hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true, true>), grid, block, 0, stream, p);     \
        else if (AK && !BKC)                                                                           \
            hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true, false>), grid, block, 0, stream, p);    \
        else                                                                                           \
            hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, false, false>), grid, block, 0, stream, p);   \
    } while (0)
    if (d->a_dtype == A3T_F32 && d->b_dtype == A3T_F32)
        LAUNCH_BF(float, float);
    else if (d->a_dtype == A3T_F32 && d->b_dtype == A3T_BF16)
        LAUNCH_BF(float, unsigned short);
    else if (d->a_dtype == A3T_BF16 && d->b_dtype == A3T_F32)
        LAUNCH_BF(unsigned short, float);
    else
        LAUNCH_BF(unsigned short, unsigned short);
#undef LAUNCH_BF
    a3t_note_kernel("gemm_bf16_kernel<%s, %s, %s, %s>", d->a_dtype == A3T_F32 ? "float" : "unsigned short",
                    d->b_dtype == A3T_F32 ? "float" : "unsigned short", AK ? "true" : "false", BKC ? "true" : "false");
    return (int)hipGetLastError();
}
