// dwconv_vec.hip -- GLU + depthwise Conv1d (conformer/convolution.py:56-79), bf16-storage fast path.
//
// Same math and tiling as the generic kernels in convmod_attn.hip (block = 64 channels x 64 time steps of one
// utterance, window staged in LDS, thread = one channel in the compute phase), but every HBM access is a 16-byte
// lane access: the generic kernels move 2-4 bytes per lane, which leaves too few bytes in flight per CU
// (measured 1.9 TB/s forward, ~1.2 TB/s backward at M = 35840, C = 384).  Inputs are staged row-wise
// (8 bf16 / 4 fp32 channels per lane), results leave through an LDS tile and row-wise 16-byte stores.
// Contract (checked by the callers in convmod_attn.hip): C % 64 == 0, g / glu / dg stored in bf16, 16-byte aligned
// base pointers; K odd, K <= 31.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"
#include "dtype_io.h"

#define DW_TT 64
#define DW_KMAX 31
typedef unsigned short u16;

__device__ __forceinline__ float dv_sigm(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float dv_bf(unsigned int h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ void dv_unpack8(const uint4& u, float* f) {
    f[0] = dv_bf(u.x & 0xffff), f[1] = dv_bf(u.x >> 16), f[2] = dv_bf(u.y & 0xffff), f[3] = dv_bf(u.y >> 16);
    f[4] = dv_bf(u.z & 0xffff), f[5] = dv_bf(u.z >> 16), f[6] = dv_bf(u.w & 0xffff), f[7] = dv_bf(u.w >> 16);
}
__device__ __forceinline__ uint4 dv_pack8(const float* f) {
    uint4 u;
    u.x = io_pack2(f[0], f[1]), u.y = io_pack2(f[2], f[3]);
    u.z = io_pack2(f[4], f[5]), u.w = io_pack2(f[6], f[7]);
    return u;
}

// ---------------------------------------------------------------------------------------------------------------
// forward: glu = a * sigmoid(b) (bf16 out), z = depthwise_conv(glu) + bias (fp32 out)
// ---------------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void glu_dwconv_fwd_vec_kernel(const u16* __restrict__ g, const float* __restrict__ wdw,
                                                                 const float* __restrict__ bdw, u16* __restrict__ glu,
                                                                 float* __restrict__ z, int C, int K, int Tseq,
                                                                 int tiles_t) {
    __shared__ __attribute__((aligned(16))) float win[DW_TT + DW_KMAX - 1][64];
    __shared__ __attribute__((aligned(16))) float zt[DW_TT][64];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int cblk = blockIdx.x * 64, c = cblk + tx;
    const int b = blockIdx.y / tiles_t, t0 = (blockIdx.y % tiles_t) * DW_TT;
    const int pad = (K - 1) / 2;
    const int64_t mbase = (int64_t)b * Tseq;
    const int rows = DW_TT + K - 1;
    {   // stage the GLU'd window: 32 rows per pass, 8 channels per lane
        const int rr = tid >> 3, c8 = (tid & 7) * 8;
        uint4 av[3], bv[3];
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            const int r = ps * 32 + rr, t = t0 - pad + r;
            const bool ok = (r < rows) && (t >= 0) && (t < Tseq);
            const u16* src = g + (ok ? (mbase + t) * (int64_t)(2 * C) + cblk + c8 : 0);
            av[ps] = *(const uint4*)src;
            bv[ps] = *(const uint4*)(src + (ok ? C : 0));
        }
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            const int r = ps * 32 + rr, t = t0 - pad + r;
            const bool ok = (r < rows) && (t >= 0) && (t < Tseq);
            if (r < rows) {
                float a[8], bb[8], v[8];
                dv_unpack8(av[ps], a);
                dv_unpack8(bv[ps], bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ok ? a[e] * dv_sigm(bb[e]) : 0.f;
                *(float4*)&win[r][c8] = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)&win[r][c8 + 4] = make_float4(v[4], v[5], v[6], v[7]);
                if (ok && r >= pad && r < pad + DW_TT) *(uint4*)(glu + (mbase + t) * (int64_t)C + cblk + c8) = dv_pack8(v);
            }
        }
    }
    __syncthreads();
    float w[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) w[k] = (k < K) ? wdw[(int64_t)c * K + k] : 0.f;
    const float bias = bdw[c];
    if (K == KT) {   // register-blocked: 4 consecutive time steps share one (KT+3)-value window
        for (int r0 = ty * 4; r0 < DW_TT; r0 += 16) {
            float v[KT + 3];
#pragma unroll
            for (int j = 0; j < KT + 3; ++j) v[j] = win[r0 + j][tx];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                float acc = bias;
#pragma unroll
                for (int k = 0; k < KT; ++k) acc = __builtin_fmaf(w[k], v[tt + k], acc);
                zt[r0 + tt][tx] = acc;
            }
        }
    } else {
        for (int r = ty; r < DW_TT; r += 4) {
            float acc = bias;
#pragma unroll
            for (int k = 0; k < KT; ++k)
                if (k < K) acc += w[k] * win[r + k][tx];
            zt[r][tx] = acc;
        }
    }
    __syncthreads();
    {   // row-wise 16-byte stores: 16 rows per pass, 4 channels per lane
        const int rr = tid >> 4, c4 = (tid & 15) * 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = ps * 16 + rr, t = t0 + r;
            if (t < Tseq) *(float4*)(z + (mbase + t) * (int64_t)C + cblk + c4) = *(const float4*)&zt[r][c4];
        }
    }
}

int a3t_glu_dwconv_fwd_vec(const void* g, const float* wdw, const float* bdw, void* glu, float* z, int M, int C, int K,
                           int Tseq, hipStream_t stream) {
    const int B = M / Tseq, tiles_t = (Tseq + DW_TT - 1) / DW_TT;
    dim3 grid(C / 64, B * tiles_t);
#define DWF(KT)                                                                                                         \
    hipLaunchKernelGGL(glu_dwconv_fwd_vec_kernel<KT>, grid, dim3(256), 0, stream, (const u16*)g, wdw, bdw, (u16*)glu, z, \
                       C, K, Tseq, tiles_t)
    if (K <= 7)
        DWF(7);
    else if (K <= 15)
        DWF(15);
    else
        DWF(DW_KMAX);
#undef DWF
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// backward: dglu = conv^T(dz), (da, db) = GLU'(dglu) -> dg (bf16), dW, dbias, column sums of dg (atomics, once per
// block: the weight-gradient partials stay in registers across the block's time tiles)
// ---------------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256, 3) void glu_dwconv_bwd_vec_kernel(const float* __restrict__ dz, const u16* __restrict__ g,
                                                                    const u16* __restrict__ glu,
                                                                    const float* __restrict__ wdw, u16* __restrict__ dg,
                                                                    float* dwdw, float* dbdw, float* dgsum, int C, int K,
                                                                    int Tseq, int tiles_t, int tiles_per_block,
                                                                    int chunks) {
    __shared__ __attribute__((aligned(16))) float wdz[DW_TT + DW_KMAX - 1][64];   // dz window  (rows t0-pad .. t0+TT+pad)
    __shared__ __attribute__((aligned(16))) u16 wgl[DW_TT + DW_KMAX - 1][64];     // glu window (bf16 as stored)
    __shared__ __attribute__((aligned(16))) u16 gt[2][DW_TT][64];                 // in: GLU inputs a, b; out: da, db
    __shared__ float red[4][64];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int cblk = blockIdx.x * 64, c = cblk + tx;
    const int b = blockIdx.y / chunks, tile0 = (blockIdx.y % chunks) * tiles_per_block;
    const int pad = (K - 1) / 2;
    const int64_t mbase = (int64_t)b * Tseq;
    const int rows = DW_TT + K - 1;
    float w[KT], dw[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        w[k] = (k < K) ? wdw[(int64_t)c * K + k] : 0.f;
        dw[k] = 0.f;
    }
    float db = 0.f, sga = 0.f, sgb = 0.f;
    for (int tile = tile0; tile < min(tiles_t, tile0 + tiles_per_block); ++tile) {
        const int t0 = tile * DW_TT;
        __syncthreads();   // the previous tile's LDS is free
        {
            // all of the tile's loads are issued back to back: 6 x float4 (dz), 3 + 2 + 2 x uint4 (glu, a, b)
            const int r16 = tid >> 4, c4 = (tid & 15) * 4;
            const int r32 = tid >> 3, c8 = (tid & 7) * 8;
            float4 vz[6];
            uint4 vg[3], va[2], vb[2];
#pragma unroll
            for (int ps = 0; ps < 6; ++ps) {
                const int r = ps * 16 + r16, t = t0 - pad + r;
                const bool ok = (r < rows) && (t >= 0) && (t < Tseq);
                vz[ps] = *(const float4*)(dz + (ok ? (mbase + t) * (int64_t)C + cblk + c4 : 0));
                if (!ok) vz[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int ps = 0; ps < 3; ++ps) {
                const int r = ps * 32 + r32, t = t0 - pad + r;
                const bool ok = (r < rows) && (t >= 0) && (t < Tseq);
                vg[ps] = *(const uint4*)(glu + (ok ? (mbase + t) * (int64_t)C + cblk + c8 : 0));
                if (!ok) vg[ps] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int r = ps * 32 + r32, t = t0 + r;
                const bool ok = t < Tseq;
                const u16* src = g + (ok ? (mbase + t) * (int64_t)(2 * C) + cblk + c8 : 0);
                va[ps] = *(const uint4*)src;
                vb[ps] = *(const uint4*)(src + (ok ? C : 0));
            }
#pragma unroll
            for (int ps = 0; ps < 6; ++ps) {
                const int r = ps * 16 + r16;
                if (r < rows) *(float4*)&wdz[r][c4] = vz[ps];
            }
#pragma unroll
            for (int ps = 0; ps < 3; ++ps) {
                const int r = ps * 32 + r32;
                if (r < rows) *(uint4*)&wgl[r][c8] = vg[ps];
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int r = ps * 32 + r32;
                *(uint4*)&gt[0][r][c8] = va[ps];
                *(uint4*)&gt[1][r][c8] = vb[ps];
            }
        }
        __syncthreads();
        if (K == KT) {   // register-blocked: 4 consecutive time steps per window read
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int r0 = ty * 4 + 16 * i;
                float acc[4], dzt[4];
                {   // data gradient: dglu[t] = sum_k w[k] * dz[t + pad - k]
                    float u[KT + 3];
#pragma unroll
                    for (int j = 0; j < KT + 3; ++j) u[j] = wdz[r0 + j][tx];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        float a = 0.f;
#pragma unroll
                        for (int k = 0; k < KT; ++k) a = __builtin_fmaf(w[k], u[tt + KT - 1 - k], a);
                        acc[tt] = a;
                        dzt[tt] = u[tt + (KT - 1) / 2];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the two (KT+3)-value windows from being live at once
                {   // weight gradient: dw[k] += dz[t] * glu[t + k - pad]
                    float q[KT + 3];
#pragma unroll
                    for (int j = 0; j < KT + 3; ++j) q[j] = dv_bf(wgl[r0 + j][tx]);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                        for (int k = 0; k < KT; ++k) dw[k] = __builtin_fmaf(dzt[tt], q[tt + k], dw[k]);
                        db += dzt[tt];
                    }
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int r = r0 + tt;
                    const float ga = dv_bf(gt[0][r][tx]), sb = dv_sigm(dv_bf(gt[1][r][tx]));
                    const float da = acc[tt] * sb, dbb = acc[tt] * ga * sb * (1.f - sb);
                    gt[0][r][tx] = io_f2bf(da);
                    gt[1][r][tx] = io_f2bf(dbb);
                    if (t0 + r < Tseq) sga += da, sgb += dbb;
                }
            }
        } else {
            for (int r = ty; r < DW_TT; r += 4) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < K) acc += w[k] * wdz[r + 2 * pad - k][tx];
                const float ga = dv_bf(gt[0][r][tx]), sb = dv_sigm(dv_bf(gt[1][r][tx]));
                const float da = acc * sb, dbb = acc * ga * sb * (1.f - sb);
                gt[0][r][tx] = io_f2bf(da);
                gt[1][r][tx] = io_f2bf(dbb);
                if (t0 + r < Tseq) sga += da, sgb += dbb;
                const float dzt = wdz[r + pad][tx];
                db += dzt;
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < K) dw[k] += dzt * dv_bf(wgl[r + k][tx]);
            }
        }
        __syncthreads();
        {   // dg rows: 32 rows per pass, 8 channels per lane, both GLU halves
            const int r32 = tid >> 3, c8 = (tid & 7) * 8;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int r = ps * 32 + r32, t = t0 + r;
                if (t < Tseq) {
                    u16* dst = dg + (mbase + t) * (int64_t)(2 * C) + cblk + c8;
                    *(uint4*)dst = *(const uint4*)&gt[0][r][c8];
                    *(uint4*)(dst + C) = *(const uint4*)&gt[1][r][c8];
                }
            }
        }
    }
    // Weight-gradient partials: summed over the block's 4 row groups with LDS float atomics into the block's
    // contiguous [64 channels][K] slice (odd K -> lane stride K is bank-conflict free), then added to dW with
    // COALESCED global atomics (consecutive lanes -> consecutive addresses: 2 cache lines per wave instruction
    // instead of one line per lane).
    __syncthreads();
    float* part = &wdz[0][0];
    for (int i = tid; i < 64 * K; i += 256) part[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KT; ++k)
        if (k < K) atomicAdd(&part[tx * K + k], dw[k]);
    __syncthreads();
    for (int i = tid; i < 64 * K; i += 256) atomicAdd(&dwdw[(int64_t)cblk * K + i], part[i]);
    __syncthreads();
    red[ty][tx] = db;
    __syncthreads();
    if (ty == 0) atomicAdd(&dbdw[c], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
    if (dgsum) {
        __syncthreads();
        red[ty][tx] = sga;
        __syncthreads();
        if (ty == 0) atomicAdd(&dgsum[c], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
        __syncthreads();
        red[ty][tx] = sgb;
        __syncthreads();
        if (ty == 0) atomicAdd(&dgsum[C + c], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
    }
}

int a3t_glu_dwconv_bwd_vec(const float* dz, const void* g, const void* glu, const float* wdw, void* dg, float* dwdw,
                           float* dbdw, float* dg_colsum, int M, int C, int K, int Tseq, hipStream_t stream) {
    const int B = M / Tseq, tiles_t = (Tseq + DW_TT - 1) / DW_TT;
    // one resident wave of blocks (3 per CU); fewer blocks = fewer same-address atomics at the end
    const int cb = C / 64;
    int chunks = 1;
    while (cb * B * chunks < 640 && chunks < tiles_t) ++chunks;
    const int tpb = (tiles_t + chunks - 1) / chunks;
    chunks = (tiles_t + tpb - 1) / tpb;
    dim3 grid(cb, B * chunks);
#define DWB(KT)                                                                                                       \
    hipLaunchKernelGGL(glu_dwconv_bwd_vec_kernel<KT>, grid, dim3(256), 0, stream, dz, (const u16*)g, (const u16*)glu, \
                       wdw, (u16*)dg, dwdw, dbdw, dg_colsum, C, K, Tseq, tiles_t, tpb, chunks)
    if (K <= 7)
        DWB(7);
    else if (K <= 15)
        DWB(15);
    else
        DWB(DW_KMAX);
#undef DWB
    return (int)hipGetLastError();
}
