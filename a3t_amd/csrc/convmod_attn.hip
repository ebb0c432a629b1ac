// convmod_attn.hip -- Conformer convolution-module kernels (GLU + depthwise Conv1d with the time
// window staged in LDS) and the rel-pos softmax kernels of the materialised attention path.
// Tensors that feed / come from the MFMA GEMMs may be stored in bf16 (runtime dtype flags).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"
#include "dtype_io.h"

#define WAVE 64
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float bf2f_(unsigned int h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

// ------------------------------------------------------------------------------------------
// GLU + depthwise conv.  Block = 64 channels (lanes) x 4 row lanes; time tile TT rows of one
// utterance; the GLU'd window [t0-pad, t0+TT+pad) x 64 channels lives in LDS (<= 94x64 floats).
// ------------------------------------------------------------------------------------------
#define DW_TT 64
#define DW_KMAX 31

template <int KT>
__global__ __launch_bounds__(256) void glu_dwconv_fwd_kernel(const void* __restrict__ g, int g_dt,
                                                             const float* __restrict__ wdw,
                                                             const float* __restrict__ bdw, void* __restrict__ glu,
                                                             int glu_dt, float* __restrict__ z, int C, int K, int Tseq,
                                                             int tiles_t) {
    __shared__ float win[DW_TT + DW_KMAX - 1][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int b = blockIdx.y / tiles_t, t0 = (blockIdx.y % tiles_t) * DW_TT;
    const int pad = (K - 1) / 2;
    const int64_t mbase = (int64_t)b * Tseq;
    const int rows = DW_TT + K - 1;
    // (batches of 8 rows per thread, loads issued back to back -- see the backward kernel)
    for (int rb = ty; rb < rows; rb += 32) {
        float ga[8], gb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rb + 4 * u, t = t0 - pad + r;
            const bool ok = (c < C) && (r < rows) && (t >= 0) && (t < Tseq);
            const int64_t gi = ok ? (mbase + t) * (int64_t)(2 * C) + c : 0;
            ga[u] = ldx(g, g_dt, gi);
            gb[u] = ldx(g, g_dt, gi + (ok ? C : 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rb + 4 * u, t = t0 - pad + r;
            const bool ok = (c < C) && (r < rows) && (t >= 0) && (t < Tseq);
            const float v = ok ? ga[u] * sigm(gb[u]) : 0.f;
            if (ok && r >= pad && r < pad + DW_TT) stx(glu, glu_dt, (mbase + t) * (int64_t)C + c, v);
            if (r < rows) win[r][tx] = v;
        }
    }
    __syncthreads();
    if (c >= C) return;
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
                for (int k = 0; k < KT; ++k) acc += w[k] * v[tt + k];
                int t = t0 + r0 + tt;
                if (t < Tseq) z[(mbase + t) * (int64_t)C + c] = acc;
            }
        }
        return;
    }
    for (int r = ty; r < DW_TT; r += 4) {
        int t = t0 + r;
        if (t >= Tseq) break;
        float acc = bias;
#pragma unroll
        for (int k = 0; k < KT; ++k)
            if (k < K) acc += w[k] * win[r + k][tx];
        z[(mbase + t) * (int64_t)C + c] = acc;
    }
}

// 16-byte-per-lane fast paths for bf16 storage (dwconv_vec.hip)
int a3t_glu_dwconv_fwd_vec(const void* g, const float* wdw, const float* bdw, void* glu, float* z, int M, int C, int K,
                           int Tseq, hipStream_t stream);
int a3t_glu_dwconv_bwd_vec(const float* dz, const void* g, const void* glu, const float* wdw, void* dg, float* dwdw,
                           float* dbdw, float* dg_colsum, int M, int C, int K, int Tseq, hipStream_t stream);
static inline bool dw_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int a3t_glu_dwconv_fwd(const void* g, int g_dtype, const float* wdw, const float* bdw, void* glu,
                                  int glu_dtype, float* z, int M, int C, int K, int Tseq, void* stream) {
    if (K > DW_KMAX || (K & 1) == 0 || Tseq <= 0 || M % Tseq) return A3T_EINVAL;
    if (g_dtype == A3T_BF16 && glu_dtype == A3T_BF16 && C % 64 == 0 && dw_al16(g) && dw_al16(glu) && dw_al16(z))
        return a3t_glu_dwconv_fwd_vec(g, wdw, bdw, glu, z, M, C, K, Tseq, (hipStream_t)stream);
    int B = M / Tseq, tiles_t = (Tseq + DW_TT - 1) / DW_TT;
    dim3 grid((C + 63) / 64, B * tiles_t);
    if (K <= 7)
        hipLaunchKernelGGL(glu_dwconv_fwd_kernel<7>, grid, dim3(256), 0, (hipStream_t)stream, g, g_dtype, wdw, bdw, glu,
                           glu_dtype, z, C, K, Tseq, tiles_t);
    else if (K <= 15)
        hipLaunchKernelGGL(glu_dwconv_fwd_kernel<15>, grid, dim3(256), 0, (hipStream_t)stream, g, g_dtype, wdw, bdw, glu,
                           glu_dtype, z, C, K, Tseq, tiles_t);
    else
        hipLaunchKernelGGL(glu_dwconv_fwd_kernel<DW_KMAX>, grid, dim3(256), 0, (hipStream_t)stream, g, g_dtype, wdw, bdw,
                           glu, glu_dtype, z, C, K, Tseq, tiles_t);
    return (int)hipGetLastError();
}

// One block = 64 channels x `tiles_per_block` consecutive time tiles of one utterance: the weight
// gradient partials stay in registers across the tiles, so the LDS reduction + atomics happen once.
template <int KT>
__global__ __launch_bounds__(256, 3) void glu_dwconv_bwd_kernel(const float* __restrict__ dz, const void* __restrict__ g,
                                                             int g_dt, const void* __restrict__ glu, int glu_dt,
                                                             const float* __restrict__ wdw, void* __restrict__ dg,
                                                             int dg_dt, float* dwdw, float* dbdw, float* dgsum, int C,
                                                             int K, int Tseq, int tiles_t, int tiles_per_block,
                                                             int chunks) {
    __shared__ float wdz[DW_TT + DW_KMAX - 1][64];   // dz window  (rows t0-pad .. t0+TT+pad)
    __shared__ float wgl[DW_TT + DW_KMAX - 1][64];   // glu window
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int b = blockIdx.y / chunks, tile0 = (blockIdx.y % chunks) * tiles_per_block;
    const int pad = (K - 1) / 2;
    const int64_t mbase = (int64_t)b * Tseq;
    const int rows = DW_TT + K - 1;
    float w[KT], dw[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        w[k] = (k < K && c < C) ? wdw[(int64_t)c * K + k] : 0.f;
        dw[k] = 0.f;
    }
    float db = 0.f, sga = 0.f, sgb = 0.f;
    for (int tile = tile0; tile < min(tiles_t, tile0 + tiles_per_block); ++tile) {
        const int t0 = tile * DW_TT;
        // The GLU inputs of this thread's next 4 outputs are requested one step ahead (first group: before the window
        // staging), so the loads complete under LDS work / FMAs instead of one dependent HBM round trip per output.
        float gan[4], gbn[4];
        auto prefetch_g = [&](int i) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = t0 + ty * 4 + 16 * i + tt;
                gan[tt] = gbn[tt] = 0.f;
                if (c < C && t < Tseq) {
                    const int64_t gi = (mbase + t) * (int64_t)(2 * C);
                    gan[tt] = ldx(g, g_dt, gi + c);
                    gbn[tt] = ldx(g, g_dt, gi + C + c);
                }
            }
        };
        if (K == KT) prefetch_g(0);
        __syncthreads();
        // window staging in batches of 8 rows per thread: the 16 loads of a batch are issued back to back (invalid
        // rows read element 0 and are zeroed by a select -- no branch that would serialise one round trip per row)
        for (int rb = ty; rb < rows; rb += 32) {
            float a[8], q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + 4 * u, t = t0 - pad + r;
                const bool ok = (c < C) && (r < rows) && (t >= 0) && (t < Tseq);
                const int64_t idx = ok ? (mbase + t) * (int64_t)C + c : 0;
                a[u] = dz[idx];
                q[u] = ldx(glu, glu_dt, idx);
                a[u] = ok ? a[u] : 0.f;
                q[u] = ok ? q[u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + 4 * u;
                if (r < rows) wdz[r][tx] = a[u], wgl[r][tx] = q[u];
            }
        }
        __syncthreads();
        if (c < C && K == KT) {   // register-blocked: 4 consecutive time steps per window read
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int r0 = ty * 4 + 16 * i;
                float ga[4], gb[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) ga[tt] = gan[tt], gb[tt] = gbn[tt];
                if (i < 3) prefetch_g(i + 1);
                float acc[4], dzt[4];
                {   // data gradient: dglu[t] = sum_k w[k] * dz[t + pad - k]
                    float u[KT + 3];
#pragma unroll
                    for (int j = 0; j < KT + 3; ++j) u[j] = wdz[r0 + j][tx];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        float a = 0.f;
#pragma unroll
                        for (int k = 0; k < KT; ++k) a += w[k] * u[tt + KT - 1 - k];
                        acc[tt] = a;
                        dzt[tt] = u[tt + (KT - 1) / 2];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the two 34-value windows from being live at once
                {   // weight gradient: dw[k] += dz[t] * glu[t + k - pad]
                    float q[KT + 3];
#pragma unroll
                    for (int j = 0; j < KT + 3; ++j) q[j] = wgl[r0 + j][tx];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                        for (int k = 0; k < KT; ++k) dw[k] += dzt[tt] * q[tt + k];
                        db += dzt[tt];
                    }
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = t0 + r0 + tt;
                    if (t < Tseq) {
                        const int64_t gi = (mbase + t) * (int64_t)(2 * C);
                        const float sb = sigm(gb[tt]);
                        const float da = acc[tt] * sb, dbb = acc[tt] * ga[tt] * sb * (1.f - sb);
                        stx(dg, dg_dt, gi + c, da);
                        stx(dg, dg_dt, gi + C + c, dbb);
                        sga += da, sgb += dbb;
                    }
                }
            }
        } else if (c < C) {
            for (int r = ty; r < DW_TT; r += 4) {
                int t = t0 + r;
                if (t >= Tseq) break;
                // data gradient: dglu[t] = sum_k w[k] * dz[t + pad - k]  (window row r + 2*pad - k)
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < K) acc += w[k] * wdz[r + 2 * pad - k][tx];
                const int64_t gi = (mbase + t) * (int64_t)(2 * C);
                float ga = ldx(g, g_dt, gi + c), sb = sigm(ldx(g, g_dt, gi + C + c));
                const float da = acc * sb, dbb = acc * ga * sb * (1.f - sb);
                stx(dg, dg_dt, gi + c, da);
                stx(dg, dg_dt, gi + C + c, dbb);
                sga += da, sgb += dbb;
                // weight gradient: dw[k] += dz[t] * glu[t + k - pad]
                float dzt = wdz[r + pad][tx];
                db += dzt;
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < K) dw[k] += dzt * wgl[r + k][tx];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        if (k >= K) break;
        __syncthreads();
        red[ty][tx] = dw[k];
        __syncthreads();
        if (ty == 0 && c < C) atomicAdd(&dwdw[(int64_t)c * K + k], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
    }
    __syncthreads();
    red[ty][tx] = db;
    __syncthreads();
    if (ty == 0 && c < C) atomicAdd(&dbdw[c], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
    if (dgsum) {
        __syncthreads();
        red[ty][tx] = sga;
        __syncthreads();
        if (ty == 0 && c < C) atomicAdd(&dgsum[c], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
        __syncthreads();
        red[ty][tx] = sgb;
        __syncthreads();
        if (ty == 0 && c < C) atomicAdd(&dgsum[C + c], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
    }
}

extern "C" int a3t_glu_dwconv_bwd(const float* dz, const void* g, int g_dtype, const void* glu, int glu_dtype,
                                  const float* wdw, void* dg, int dg_dtype, float* dwdw, float* dbdw, float* dg_colsum,
                                  int M, int C, int K, int Tseq, void* stream) {
    if (K > DW_KMAX || (K & 1) == 0 || Tseq <= 0 || M % Tseq) return A3T_EINVAL;
    if (g_dtype == A3T_BF16 && glu_dtype == A3T_BF16 && dg_dtype == A3T_BF16 && C % 64 == 0 && dw_al16(g) &&
        dw_al16(glu) && dw_al16(dg) && dw_al16(dz))
        return a3t_glu_dwconv_bwd_vec(dz, g, glu, wdw, dg, dwdw, dbdw, dg_colsum, M, C, K, Tseq, (hipStream_t)stream);
    int B = M / Tseq, tiles_t = (Tseq + DW_TT - 1) / DW_TT;
    // enough blocks to fill 256 CUs a few times, as few weight-gradient reductions as possible
    int cb = (C + 63) / 64;
    int chunks = 1;
    while (cb * B * chunks < 1024 && chunks < tiles_t) ++chunks;
    int tpb = (tiles_t + chunks - 1) / chunks;
    chunks = (tiles_t + tpb - 1) / tpb;
    dim3 grid(cb, B * chunks);
#define DWB(KT)                                                                                                       \
    hipLaunchKernelGGL(glu_dwconv_bwd_kernel<KT>, grid, dim3(256), 0, (hipStream_t)stream, dz, g, g_dtype, glu, glu_dtype, \
                       wdw, dg, dg_dtype, dwdw, dbdw, dg_colsum, C, K, Tseq, tiles_t, tpb, chunks)
    if (K <= 7)
        DWB(7);
    else if (K <= 15)
        DWB(15);
    else
        DWB(DW_KMAX);
#undef DWB
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// attention helpers
// ------------------------------------------------------------------------------------------
__global__ void add_pos_bias_kernel(const void* __restrict__ qkv, const float* __restrict__ bu,
                                    const float* __restrict__ bv, void* __restrict__ qu, void* __restrict__ qv,
                                    int dt, int64_t n, int d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t m = i / d;
        int c = (int)(i - m * d);
        float q = ldx(qkv, dt, m * 3 * d + c);
        stx(qu, dt, i, q + bu[c]);
        stx(qv, dt, i, q + bv[c]);
    }
}
// bf16 storage, d % 8 == 0: 8 channels (16 bytes) per lane
__device__ __forceinline__ void apb_unpack8(const uint4& u, float* f) {
    f[0] = bf2f_(u.x & 0xffff), f[1] = bf2f_(u.x >> 16), f[2] = bf2f_(u.y & 0xffff), f[3] = bf2f_(u.y >> 16);
    f[4] = bf2f_(u.z & 0xffff), f[5] = bf2f_(u.z >> 16), f[6] = bf2f_(u.w & 0xffff), f[7] = bf2f_(u.w >> 16);
}
__device__ __forceinline__ uint4 apb_pack8(const float* f) {
    uint4 u;
    u.x = io_pack2(f[0], f[1]), u.y = io_pack2(f[2], f[3]);
    u.z = io_pack2(f[4], f[5]), u.w = io_pack2(f[6], f[7]);
    return u;
}
__global__ void add_pos_bias_bf16_kernel(const unsigned short* __restrict__ qkv, const float* __restrict__ bu,
                                         const float* __restrict__ bv, unsigned short* __restrict__ qu,
                                         unsigned short* __restrict__ qv, int64_t n8, int d8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / d8;
        const int c = (int)(i - m * d8) * 8;
        float q[8], o[8];
        apb_unpack8(*(const uint4*)(qkv + m * 3 * d8 * 8 + c), q);
        const float4 u0 = *(const float4*)(bu + c), u1 = *(const float4*)(bu + c + 4);
        const float4 v0 = *(const float4*)(bv + c), v1 = *(const float4*)(bv + c + 4);
        const float uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = q[e] + uu[e];
        *(uint4*)(qu + i * 8) = apb_pack8(o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = q[e] + vv[e];
        *(uint4*)(qv + i * 8) = apb_pack8(o);
    }
}
static inline bool apb_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
extern "C" int a3t_add_pos_bias(const void* qkv, const float* bias_u, const float* bias_v, void* qu, void* qv,
                                int dtype, int M, int d, void* stream) {
    int64_t n = (int64_t)M * d;
    if (dtype == A3T_BF16 && d % 8 == 0 && apb_al16(qkv) && apb_al16(qu) && apb_al16(qv) && apb_al16(bias_u) &&
        apb_al16(bias_v)) {
        int64_t n8 = n / 8;
        int blocks = (int)((n8 + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(add_pos_bias_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)qkv, bias_u, bias_v, (unsigned short*)qu, (unsigned short*)qv, n8, d / 8);
        return (int)hipGetLastError();
    }
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_pos_bias_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, qkv, bias_u, bias_v, qu,
                       qv, dtype, n, d);
    return (int)hipGetLastError();
}
__global__ void add_pos_bias_bwd_kernel(const void* __restrict__ dqu, const void* __restrict__ dqv,
                                        void* __restrict__ dqkv, int dt, int64_t n, int d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t m = i / d;
        int c = (int)(i - m * d);
        stx(dqkv, dt, m * 3 * d + c, ldx(dqu, dt, i) + ldx(dqv, dt, i));
    }
}
__global__ void add_pos_bias_bwd_bf16_kernel(const unsigned short* __restrict__ dqu, const unsigned short* __restrict__ dqv,
                                             unsigned short* __restrict__ dqkv, int64_t n8, int d8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / d8;
        const int c = (int)(i - m * d8) * 8;
        float a[8], b[8], o[8];
        apb_unpack8(*(const uint4*)(dqu + i * 8), a);
        apb_unpack8(*(const uint4*)(dqv + i * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a[e] + b[e];
        *(uint4*)(dqkv + m * 3 * d8 * 8 + c) = apb_pack8(o);
    }
}
extern "C" int a3t_add_pos_bias_bwd(const void* dqu, const void* dqv, void* dqkv, int dtype, int M, int d,
                                    void* stream) {
    int64_t n = (int64_t)M * d;
    if (dtype == A3T_BF16 && d % 8 == 0 && apb_al16(dqu) && apb_al16(dqv) && apb_al16(dqkv)) {
        int64_t n8 = n / 8;
        int blocks = (int)((n8 + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(add_pos_bias_bwd_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)dqu, (const unsigned short*)dqv, (unsigned short*)dqkv, n8, d / 8);
        return (int)hipGetLastError();
    }
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_pos_bias_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dqu, dqv, dqkv, dtype,
                       n, d);
    return (int)hipGetLastError();
}

// Legacy rel_shift (attention.py:145-165) in closed form on the COMPACT BD = (q+v) P^T matrix:
//   j <= i : BD[i][T-1-(i-j)]     j == i+1 : 0     j > i+1 : BD[i+1][j-i-2]
// i.e. each shifted row is two contiguous segments of BD -> coalesced reads, no padded copy.
// (Equivalently BD[r][c] <-> flat score index r*(T+1) + c - (T-1): consecutive BD rows are consecutive flat windows
//  of the score matrix separated by the one skipped element S[r][r+1].)
__device__ __forceinline__ float bd_shift(const void* __restrict__ BDz, int dt, int T, int i, int j) {
    if (j <= i) return ldx(BDz, dt, (int64_t)i * T + (T - 1 - i + j));
    if (j == i + 1) return 0.f;
    return ldx(BDz, dt, (int64_t)(i + 1) * T + (j - i - 2));
}

// one wave per (z, i) row.  NV > 0: the row is read ONCE and kept in NV registers per lane
// (T <= 64*NV); NV == 0: generic three-pass fallback for very long rows.
template <int NV>
__global__ __launch_bounds__(256) void relpos_softmax_fwd_kernel(const void* __restrict__ ac,
                                                                 const void* __restrict__ bd, int s_dt,
                                                                 const uint8_t* __restrict__ keymask,
                                                                 void* __restrict__ probs, int p_dt, int H, int T,
                                                                 int64_t ac_bs, int64_t bd_bs, int64_t p_bs,
                                                                 float scale, int64_t nrows, void* __restrict__ pdrop,
                                                                 unsigned int thr, float inv, unsigned int key) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t zz = row / T;
    const int i = (int)(row - zz * T);
    const int b = (int)(zz / H);
    const int64_t ao = zz * ac_bs + (int64_t)i * T;
    const void* bz = (s_dt == A3T_BF16) ? (const void*)((const unsigned short*)bd + zz * bd_bs)
                                        : (const void*)((const float*)bd + zz * bd_bs);
    const uint8_t* mk = keymask + (int64_t)b * T;
    const int64_t po = zz * p_bs + (int64_t)i * T;
    const float NEG = -3.4028235e38f;
    if (NV > 0) {
        float v[NV > 0 ? NV : 1];
        float mx = NEG;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            int j = lane + q * 64;
            v[q] = NEG;
            if (j < T && mk[j]) v[q] = (ldx(ac, s_dt, ao + j) + bd_shift(bz, s_dt, T, i, j)) * scale;
            mx = fmaxf(mx, v[q]);
        }
        mx = wmax(mx);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            v[q] = (v[q] > NEG) ? expf(v[q] - mx) : 0.f;   // masked keys and the all-masked row -> 0
            s += v[q];
        }
        s = wsum(s);
        const float inv_s = s > 0.f ? 1.f / s : 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            int j = lane + q * 64;
            if (j < T) {
                const float pj = v[q] * inv_s;
                stx(probs, p_dt, po + j, pj);
                if (pdrop) stx(pdrop, p_dt, po + j, rng_keep(key, (unsigned int)(po + j), thr) ? pj * inv : 0.f);
            }
        }
        return;
    }
    float mx = NEG;
    int any = 0;
    for (int j = lane; j < T; j += 64)
        if (mk[j]) {
            mx = fmaxf(mx, (ldx(ac, s_dt, ao + j) + bd_shift(bz, s_dt, T, i, j)) * scale);
            any = 1;
        }
    mx = wmax(mx);
    any = __any(any);
    if (!any) {  // every key padded: softmax over equal fills then masked_fill(0) -> zeros
        for (int j = lane; j < T; j += 64) {
            stx(probs, p_dt, po + j, 0.f);
            if (pdrop) stx(pdrop, p_dt, po + j, 0.f);
        }
        return;
    }
    float s = 0.f;
    for (int j = lane; j < T; j += 64)
        if (mk[j]) s += expf((ldx(ac, s_dt, ao + j) + bd_shift(bz, s_dt, T, i, j)) * scale - mx);
    s = wsum(s);
    const float inv_s = 1.f / s;
    for (int j = lane; j < T; j += 64) {
        const float pj = mk[j] ? expf((ldx(ac, s_dt, ao + j) + bd_shift(bz, s_dt, T, i, j)) * scale - mx) * inv_s : 0.f;
        stx(probs, p_dt, po + j, pj);
        if (pdrop) stx(pdrop, p_dt, po + j, rng_keep(key, (unsigned int)(po + j), thr) ? pj * inv : 0.f);
    }
}

// ---- all-bf16 fast path (bf16 compute mode): T % 8 == 0, every lane owns NC chunks of 8 consecutive keys, so the
// row-aligned tensors (ac, key mask, probs, dropped probs, dprobs, ds) move as 16-byte accesses; only the shifted
// bd / dbd elements (arbitrary 2-byte alignment) are touched one bf16 at a time.
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = bf2f_(u.x & 0xffff), f[1] = bf2f_(u.x >> 16), f[2] = bf2f_(u.y & 0xffff), f[3] = bf2f_(u.y >> 16);
    f[4] = bf2f_(u.z & 0xffff), f[5] = bf2f_(u.z >> 16), f[6] = bf2f_(u.w & 0xffff), f[7] = bf2f_(u.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 u;
    u.x = io_pack2(f[0], f[1]), u.y = io_pack2(f[2], f[3]);
    u.z = io_pack2(f[4], f[5]), u.w = io_pack2(f[6], f[7]);
    return u;
}

template <int NC>
__global__ __launch_bounds__(256) void relpos_softmax_fwd_bf16_kernel(
    const unsigned short* __restrict__ ac, const unsigned short* __restrict__ bd, const uint8_t* __restrict__ keymask,
    unsigned short* __restrict__ probs, int H, int T, int64_t ac_bs, int64_t bd_bs, int64_t p_bs, float scale,
    int64_t nrows, unsigned short* __restrict__ pdrop, unsigned int thr, float inv, unsigned int key) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t zz = row / T;
    const int i = (int)(row - zz * T);
    const int b = (int)(zz / H);
    const unsigned short* ar = ac + zz * ac_bs + (int64_t)i * T;
    const unsigned short* b0 = bd + zz * bd_bs + (int64_t)i * T + (T - 1 - i);   // + j        (j <= i)
    const unsigned short* b1 = bd + zz * bd_bs + (int64_t)(i + 1) * T - i - 2;   // + j        (j >= i + 2)
    const uint8_t* mk = keymask + (int64_t)b * T;
    const int64_t po = zz * p_bs + (int64_t)i * T;
    const float NEG = -3.4028235e38f;
    const int nch = T >> 3;
    float v[NC][8];
    float mx = NEG;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = lane + q * 64, j0 = c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q][e] = NEG;
        if (c < nch) {
            float a[8];
            unpack8(*(const uint4*)(ar + j0), a);
            const uint2 m2 = *(const uint2*)(mk + j0);
            float bv[8];
            if (j0 + 7 <= i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = bf2f_(b0[j0 + e]);
            } else if (j0 >= i + 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = bf2f_(b1[j0 + e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = j0 + e;
                    bv[e] = (j <= i) ? bf2f_(b0[j]) : (j == i + 1 ? 0.f : bf2f_(b1[j]));
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned int mb = ((e < 4 ? m2.x : m2.y) >> (8 * (e & 3))) & 0xff;
                if (mb) v[q][e] = (a[e] + bv[e]) * scale;
                mx = fmaxf(mx, v[q][e]);
            }
        }
    }
    mx = wmax(mx);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[q][e] = (v[q][e] > NEG) ? __expf(v[q][e] - mx) : 0.f;   // v_exp_f32: bf16 probabilities need no more
            s += v[q][e];
        }
    s = wsum(s);
    const float inv_s = s > 0.f ? 1.f / s : 0.f;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = lane + q * 64, j0 = c * 8;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[q][e] * inv_s;
            *(uint4*)(probs + po + j0) = pack8(o);
            if (pdrop) {
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    bool kp[4];
                    rng_keep4(key, (unsigned int)(po + j0 + e), thr, kp);
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[e + u] = kp[u] ? o[e + u] * inv : 0.f;
                }
                *(uint4*)(pdrop + po + j0) = pack8(o);
            }
        }
    }
}

#define SM_DISPATCH(T, CALL)               \
    do {                                   \
        if ((T) <= 128) { CALL(2); }       \
        else if ((T) <= 256) { CALL(4); }  \
        else if ((T) <= 512) { CALL(8); }  \
        else if ((T) <= 1152) { CALL(18); } \
        else if ((T) <= 2048) { CALL(32); } \
        else { CALL(0); }                  \
    } while (0)
#define SM_DISPATCH_VEC(T, CALL)            \
    do {                                    \
        if ((T) <= 512) { CALL(1); }        \
        else if ((T) <= 1024) { CALL(2); }  \
        else if ((T) <= 1536) { CALL(3); }  \
        else { CALL(4); }                   \
    } while (0)
static inline bool sm_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int a3t_relpos_softmax_fwd(const void* ac, const void* bd, int scores_dtype, const uint8_t* keymask,
                                      void* probs, int probs_dtype, int B, int H, int T, int64_t ac_bs, int64_t bd_bs,
                                      int64_t p_bs, float scale, void* probs_drop, float drop_p, uint32_t drop_key,
                                      void* stream) {
    int64_t nrows = (int64_t)B * H * T;
    if (drop_p < 0.f || drop_p >= 1.f) return A3T_EINVAL;
    if (drop_p == 0.f) probs_drop = nullptr;
    const unsigned int thr = (unsigned int)((double)drop_p * 4294967296.0);
    const float dinv = 1.f / (1.f - drop_p);
    if (scores_dtype == A3T_BF16 && probs_dtype == A3T_BF16 && T % 8 == 0 && T <= 2048 && ac_bs % 8 == 0 &&
        p_bs % 8 == 0 && sm_al16(ac) && sm_al16(probs) && sm_al16(keymask) && (!probs_drop || sm_al16(probs_drop))) {
#define CALLV(NC)                                                                                                     \
    hipLaunchKernelGGL(relpos_softmax_fwd_bf16_kernel<NC>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0,           \
                       (hipStream_t)stream, (const unsigned short*)ac, (const unsigned short*)bd, keymask,            \
                       (unsigned short*)probs, H, T, ac_bs, bd_bs, p_bs, scale, nrows, (unsigned short*)probs_drop, \
                       thr, dinv, drop_key)
        SM_DISPATCH_VEC(T, CALLV);
#undef CALLV
        return (int)hipGetLastError();
    }
#define CALL(NV)                                                                                                 \
    hipLaunchKernelGGL(relpos_softmax_fwd_kernel<NV>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0,           \
                       (hipStream_t)stream, ac, bd, scores_dtype, keymask, probs, probs_dtype, H, T, ac_bs, bd_bs, p_bs, \
                       scale, nrows, probs_drop, thr, dinv, drop_key)
    SM_DISPATCH(T, CALL);
#undef CALL
    return (int)hipGetLastError();
}

// ds = probs*(dprobs - sum_j dprobs*probs)*scale: written to ds (same dtype as dbd; may alias dprobs
// when fp32) and scattered un-shifted into the compact dBD (every dBD element written exactly once).
template <int NV>
__global__ __launch_bounds__(256) void relpos_softmax_bwd_kernel(const void* __restrict__ probs, int p_dt,
                                                                 const void* dprobs, int dp_dt, void* ds,
                                                                 void* __restrict__ dbd,
                                                                 int o_dt, int T, int64_t p_bs, int64_t dp_bs,
                                                                 int64_t o_bs, float scale, int64_t nrows,
                                                                 const void* __restrict__ pdrop, float dinv, int H,
                                                                 int64_t dbd_bsb, int64_t dbd_bsh) {
    // attention dropout: dprobs is the gradient of the DROPPED probabilities; the mask is read off the
    // saved dropped tensor (pdrop != 0), so no RNG replay is needed: dP = dPd * mask/(1-p)
    auto dpv = [&](int64_t pidx, float d) -> float {
        if (!pdrop) return d;
        return (ldx(pdrop, p_dt, pidx) != 0.f) ? d * dinv : 0.f;
    };
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t zz = row / T;
    const int i = (int)(row - zz * T);
    const int64_t po = zz * p_bs + (int64_t)i * T;
    const int64_t dro = zz * dp_bs + (int64_t)i * T;
    const int64_t oz = zz * o_bs;
    const int64_t dz = (zz / H) * dbd_bsb + (zz % H) * dbd_bsh;     // dBD block of (b, h): its own strides (head-major for the
                                                                    // batch-folded d linear_pos GEMM)
    float pv[NV > 0 ? NV : 1], dv[NV > 0 ? NV : 1];
    float s = 0.f;
    if (NV > 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            int j = lane + q * 64;
            pv[q] = (j < T) ? ldx(probs, p_dt, po + j) : 0.f;
            dv[q] = (j < T) ? dpv(po + j, ldx(dprobs, dp_dt, dro + j)) : 0.f;
            s += pv[q] * dv[q];
        }
    } else {
        for (int j = lane; j < T; j += 64) s += ldx(probs, p_dt, po + j) * dpv(po + j, ldx(dprobs, dp_dt, dro + j));
    }
    s = wsum(s);
    auto emit = [&](int j, float v) {
        stx(ds, o_dt, oz + (int64_t)i * T + j, v);
        if (j <= i)
            stx(dbd, o_dt, dz + (int64_t)i * T + (T - 1 - i + j), v);
        else if (j > i + 1)
            stx(dbd, o_dt, dz + (int64_t)(i + 1) * T + (j - i - 2), v);
    };
    if (NV > 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            int j = lane + q * 64;
            if (j < T) emit(j, pv[q] * (dv[q] - s) * scale);
        }
    } else {
        for (int j = lane; j < T; j += 64)
            emit(j, ldx(probs, p_dt, po + j) * (dpv(po + j, ldx(dprobs, dp_dt, dro + j)) - s) * scale);
    }
    if (i == 0)  // BD[0][0..T-2] never reaches the scores
        for (int j = lane; j < T - 1; j += 64) stx(dbd, o_dt, dz + j, 0.f);
}

template <int NC>
__global__ __launch_bounds__(256) void relpos_softmax_bwd_bf16_kernel(
    const unsigned short* __restrict__ probs, const unsigned short* __restrict__ dprobs, unsigned short* __restrict__ ds,
    unsigned short* __restrict__ dbd, int T, int64_t p_bs, int64_t dp_bs, int64_t o_bs, float scale, int64_t nrows,
    const unsigned short* __restrict__ pdrop, float dinv, int H, int64_t dbd_bsb, int64_t dbd_bsh, unsigned int rng_thr_,
    unsigned int rng_key, const float* __restrict__ rowscale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float rs = rowscale ? rowscale[row] : 1.f;       // probs hold exp(s - m_ref) un-normalised (a3t_attn_fwd_train)
    const int64_t zz = row / T;
    const int i = (int)(row - zz * T);
    const int64_t po = zz * p_bs + (int64_t)i * T;
    const unsigned short* dr = dprobs + zz * dp_bs + (int64_t)i * T;
    unsigned short* so = ds + zz * o_bs + (int64_t)i * T;
    const int64_t dz = (zz / H) * dbd_bsb + (zz % H) * dbd_bsh;            // dBD block of (b, h), own strides
    unsigned short* d0 = dbd + dz + (int64_t)i * T + (T - 1 - i);          // + j   (j <= i)
    unsigned short* d1 = dbd + dz + (int64_t)(i + 1) * T - i - 2;          // + j   (j >= i + 2)
    const int nch = T >> 3;
    float pv[NC][8], dv[NC][8];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = lane + q * 64, j0 = c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[q][e] = 0.f, dv[q][e] = 0.f;
        if (c < nch) {
            unpack8(*(const uint4*)(probs + po + j0), pv[q]);
            unpack8(*(const uint4*)(dr + j0), dv[q]);
            if (rowscale) {
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[q][e] *= rs;
            }
            if (pdrop) {
                const uint4 m = *(const uint4*)(pdrop + po + j0);
                const unsigned int mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned int h = (mw[e >> 1] >> (16 * (e & 1))) & 0x7fff;   // |x| != 0  (ignores -0)
                    dv[q][e] = h ? dv[q][e] * dinv : 0.f;
                }
            } else if (rng_thr_) {   // the forward's mask regenerated from the counter RNG (same index): 2 B / score less to read
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    bool kp[4];
                    rng_keep4(rng_key, (unsigned int)(po + j0 + e), rng_thr_, kp);
#pragma unroll
                    for (int f = 0; f < 4; ++f) dv[q][e + f] = kp[f] ? dv[q][e + f] * dinv : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += pv[q][e] * dv[q][e];
        }
    }
    s = wsum(s);
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = lane + q * 64, j0 = c * 8;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = pv[q][e] * (dv[q][e] - s) * scale;
            const uint4 u = pack8(o);
            *(uint4*)(so + j0) = u;
            const unsigned int uw[4] = {u.x, u.y, u.z, u.w};
            if (j0 + 7 <= i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) d0[j0 + e] = (unsigned short)(uw[e >> 1] >> (16 * (e & 1)));
            } else if (j0 >= i + 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) d1[j0 + e] = (unsigned short)(uw[e >> 1] >> (16 * (e & 1)));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = j0 + e;
                    const unsigned short hv = (unsigned short)(uw[e >> 1] >> (16 * (e & 1)));
                    if (j <= i)
                        d0[j] = hv;
                    else if (j > i + 1)
                        d1[j] = hv;
                }
            }
        }
    }
    if (i == 0)  // BD[0][0..T-2] never reaches the scores
        for (int j = lane; j < T - 1; j += 64) dbd[dz + j] = 0;
}

extern "C" int a3t_relpos_softmax_bwd(const void* probs, int probs_dtype, const void* dprobs, int dprobs_dtype, void* ds,
                                      void* dbd, int out_dtype, int B, int H, int T, int64_t p_bs, int64_t dp_bs,
                                      int64_t o_bs, float scale, const void* probs_drop, float drop_p, int64_t dbd_bsb,
                                      int64_t dbd_bsh, uint32_t drop_key, const float* rowscale, void* stream) {
    if (dbd_bsb == 0 && dbd_bsh == 0) dbd_bsb = (int64_t)H * o_bs, dbd_bsh = o_bs;     // default: [B][H][T][T] like ds
    if (drop_p < 0.f || drop_p >= 1.f) return A3T_EINVAL;
    const float dinv = 1.f / (1.f - drop_p);
    if (drop_p == 0.f) probs_drop = nullptr;
    // drop_p > 0 without probs_drop: the mask is regenerated from the counter RNG (key drop_key, element index
    // z*p_bs + i*T + j as in a3t_relpos_softmax_fwd) -- bf16 vector kernel with the standard p_bs = T*T only
    const unsigned int rthr = (drop_p > 0.f && !probs_drop) ? (unsigned int)((double)drop_p * 4294967296.0) : 0u;
    int64_t nrows = (int64_t)B * H * T;
    if (probs_dtype == A3T_BF16 && dprobs_dtype == A3T_BF16 && out_dtype == A3T_BF16 && T % 8 == 0 && T <= 2048 &&
        p_bs % 8 == 0 && dp_bs % 8 == 0 && o_bs % 8 == 0 && dbd_bsb % 8 == 0 && dbd_bsh % 8 == 0 && sm_al16(probs) && sm_al16(dprobs) && sm_al16(ds) &&
        (!probs_drop || sm_al16(probs_drop)) && ds != dprobs) {
#define CALLV(NC)                                                                                                  \
    hipLaunchKernelGGL(relpos_softmax_bwd_bf16_kernel<NC>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0,        \
                       (hipStream_t)stream, (const unsigned short*)probs, (const unsigned short*)dprobs,          \
                       (unsigned short*)ds, (unsigned short*)dbd, T, p_bs, dp_bs, o_bs, scale, nrows,             \
                       (const unsigned short*)probs_drop, dinv, H, dbd_bsb, dbd_bsh, rthr, drop_key, rowscale)
        SM_DISPATCH_VEC(T, CALLV);
#undef CALLV
        return (int)hipGetLastError();
    }
    if (rthr || rowscale) return A3T_EINVAL;    // (the generic kernel reads the mask off probs_drop and takes normalised probabilities)
#define CALL(NV)                                                                                                 \
    hipLaunchKernelGGL(relpos_softmax_bwd_kernel<NV>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0,           \
                       (hipStream_t)stream, probs, probs_dtype, dprobs, dprobs_dtype, ds, dbd, out_dtype, T, p_bs, dp_bs, \
                       o_bs, scale, nrows, probs_drop, dinv, H, dbd_bsb, dbd_bsh)
    SM_DISPATCH(T, CALL);
#undef CALL
    return (int)hipGetLastError();
}
