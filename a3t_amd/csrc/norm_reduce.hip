// norm_reduce.hip -- HBM-bound row/column kernels of the A3T path for gfx950:
// LayerNorm fwd/bwd (one wave64 per row, __shfl_xor butterflies), column reductions with
// double-precision atomics (bias grads, BatchNorm statistics), BatchNorm(+Swish/tanh) fwd/bwd.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"
#include "dtype_io.h"

#define WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave64 per row, the row lives in V registers per lane (V = ceil(D/64), chosen at
// launch so the d=384 rows of the Conformer use exactly 6), __shfl_xor butterflies for the stats.
// ------------------------------------------------------------------------------------------
#define LN_MAXV 24  // D <= 1536

template <int V>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ b, void* __restrict__ y, int y_dt,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                     int D, float eps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    if (row >= M) return;
    const float* xr = x + (int64_t)row * D;
    float v[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        int c = lane + i * 64;
        v[i] = (c < D) ? xr[c] : 0.f;
        s += v[i];
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        int c = lane + i * 64;
        float dlt = (c < D) ? (v[i] - mu) : 0.f;
        q += dlt * dlt;
    }
    const float var = wave_sum(q) / (float)D;
    const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        int c = lane + i * 64;
        if (c < D) stx(y, y_dt, (int64_t)row * D + c, (v[i] - mu) * rs * g[c] + b[c]);
    }
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// D % 128 == 0 fast path: half a wave (32 lanes) per row, NQ float4 per lane (D = 128*NQ): 16-byte loads, 8-byte
// (bf16) / 16-byte stores, two rows in flight per wave.
template <int NQ>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ b, void* __restrict__ y, int y_dt,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                         int D, float eps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
    const int row = (blockIdx.x * 4 + wv) * 2 + half;
    const bool ok = row < M;
    const int64_t ro = (int64_t)(ok ? row : 0) * D;
    float4 v[NQ];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        v[i] = *(const float4*)(x + ro + (hl + 32 * i) * 4);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
    const float mu = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const float a = v[i].x - mu, c = v[i].y - mu, d = v[i].z - mu, e = v[i].w - mu;
        q += a * a + c * c + d * d + e * e;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, WAVE);
    const float rs = 1.0f / sqrtf(q / (float)D + eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int c = (hl + 32 * i) * 4;
        const float4 gg = *(const float4*)(g + c), bb = *(const float4*)(b + c);
        const float o0 = (v[i].x - mu) * rs * gg.x + bb.x, o1 = (v[i].y - mu) * rs * gg.y + bb.y;
        const float o2 = (v[i].z - mu) * rs * gg.z + bb.z, o3 = (v[i].w - mu) * rs * gg.w + bb.w;
        if (y_dt == A3T_BF16) {
            uint2 h;
            h.x = io_pack2(o0, o1), h.y = io_pack2(o2, o3);
            *(uint2*)((unsigned short*)y + ro + c) = h;
        } else {
            *(float4*)((float*)y + ro + c) = make_float4(o0, o1, o2, o3);
        }
    }
    if (hl == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// dx = dres + LN'(dy); optional bf16 copy; column sums of dy*xhat, dy (and optionally of dx, the
// bias gradient of whatever produced the residual-stream gradient) reduced per block, then atomics.
template <int V>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, int dy_dt, const float* __restrict__ x,
                                                     const float* __restrict__ g, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* dres, float* dx,
                                                     unsigned short* __restrict__ dx16, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dxsum,
                                                     float dxsum_scale, int M, int D) {
    __shared__ float red[3][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float gam[V], ag[V], ab[V], ax[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        int c = lane + i * 64;
        gam[i] = (c < D) ? g[c] : 0.f;
        ag[i] = ab[i] = ax[i] = 0.f;
    }
    for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const int64_t ro = (int64_t)row * D;
        float xh[V], dg[V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            int c = lane + i * 64;
            float d = (c < D) ? ldx(dy, dy_dt, ro + c) : 0.f;
            xh[i] = (c < D) ? (x[ro + c] - mu) * rs : 0.f;
            dg[i] = d * gam[i];
            s1 += dg[i];
            s2 += dg[i] * xh[i];
            ag[i] += d * xh[i];
            ab[i] += d;
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            int c = lane + i * 64;
            if (c < D) {
                float o = rs * (dg[i] - s1 - xh[i] * s2);
                if (dres) o += dres[ro + c];
                dx[ro + c] = o;
                if (dx16) dx16[ro + c] = io_f2bf(o);
                ax[i] += o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if (i * 64 >= D) break;
        red[0][wv][lane] = ag[i];
        red[1][wv][lane] = ab[i];
        red[2][wv][lane] = ax[i];
        __syncthreads();
        if (wv == 0) {
            int c = lane + i * 64;
            if (c < D) {
                atomicAdd(&dgamma[c], red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
                atomicAdd(&dbeta[c], red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
                if (dxsum)
                    atomicAdd(&dxsum[c], dxsum_scale * (red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane]));
            }
        }
        __syncthreads();
    }
}


// D % 128 == 0 (round 5): a whole wave per row, NP float2 per lane (D = 128*NP).  Half the per-lane state of the half-wave-per-row
// layout it replaced (76 registers against 132 at D = 384; that kernel left the library in round 6): inside the training step this kernel runs beside the weight-gradient
// GEMM of the other stream, whose two waves per SIMD leave 160 of the 512 registers -- one wave of the 132-register kernel, two of
// this one (LayerNorm backward inside the step: 93 us with the half-wave layout, 54 us alone).  The column-sum partials of a
// block's eight waves go through a 12-KiB LDS buffer one array at a time.  amdgpu_num_vgpr: at D = 384 the kernel needs 82 registers
// without the hint and 80 with it -- the allocation granule is 8, and 88 would be one wave per SIMD beside the GEMM again
// (tests/test_host_logic.py holds it to 80).
template <int NP, bool DY16>
__global__ __attribute__((amdgpu_num_vgpr(80))) __launch_bounds__(512) void ln_bwd_row64_kernel(const void* __restrict__ dy, int dy_dt, const float* __restrict__ x,
                                                           const float* __restrict__ g, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* dres, float* dx,
                                                           unsigned short* __restrict__ dx16, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dxsum,
                                                           float dxsum_scale, int M, int D, unsigned int drop_thr,
                                                           float drop_inv, unsigned int drop_key) {
    __shared__ float red[8][128 * NP];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 gam[NP], ag[NP], ab[NP], ax[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        gam[i] = *(const float2*)(g + (lane + 64 * i) * 2);
        ag[i] = ab[i] = ax[i] = make_float2(0.f, 0.f);
    }
    const float invD = 1.f / (float)D;
    const unsigned int t16 = drop_thr >> 16;
    // Software pipeline over the wave's rows: dy and x of the NEXT row are requested between the row reduction and the second pass of
    // this one (into the registers the first pass has just emptied), the residual gradient of a row at its top -- a wave always has a
    // row's worth of loads in flight while it computes, which is what two waves per SIMD need to keep HBM busy.
    float2 xn[NP];
    unsigned int dn16[DY16 ? NP : 1];
    float2 dn32[DY16 ? 1 : NP];
    auto request = [&](const int row) __attribute__((always_inline)) {
        const int64_t ro = (int64_t)row * D;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = (lane + 64 * i) * 2;
            if (DY16) dn16[i] = *(const unsigned int*)((const unsigned short*)dy + ro + c);
            else dn32[i] = *(const float2*)((const float*)dy + ro + c);
            xn[i] = *(const float2*)(x + ro + c);
        }
    };
    int row = blockIdx.x * 8 + wv;
    const int stride = gridDim.x * 8;
    if (row < M) request(row);
    for (; row < M; row += stride) {
        const float mu = mean[row], rs = rstd[row];
        const int64_t ro = (int64_t)row * D;
        float2 xh[NP], dg[NP], rr[NP];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            rr[i] = make_float2(0.f, 0.f);
            if (dres) rr[i] = *(const float2*)(dres + ro + (lane + 64 * i) * 2);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float2 d;
            if (DY16) d = make_float2(io_bf2f(dn16[i] & 0xffff), io_bf2f(dn16[i] >> 16));
            else d = dn32[i];
            const float2 xv = xn[i];
            xh[i] = make_float2((xv.x - mu) * rs, (xv.y - mu) * rs);
            dg[i] = make_float2(d.x * gam[i].x, d.y * gam[i].y);
            s1 += dg[i].x + dg[i].y;
            s2 += dg[i].x * xh[i].x + dg[i].y * xh[i].y;
            ag[i].x += d.x * xh[i].x, ag[i].y += d.y * xh[i].y;
            ab[i].x += d.x, ab[i].y += d.y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, WAVE);
            s2 += __shfl_xor(s2, o, WAVE);
        }
        s1 *= invD, s2 *= invD;
        if (row + stride < M) request(row + stride);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = (lane + 64 * i) * 2;
            float2 o = make_float2(rs * (dg[i].x - s1 - xh[i].x * s2) + rr[i].x, rs * (dg[i].y - s1 - xh[i].y * s2) + rr[i].y);
            *(float2*)(dx + ro + c) = o;
            if (drop_inv > 0.f) {   // the consumer sub-layer's output dropout (same element pairs as rng_keep4)
                const unsigned int h = rng_pair(drop_key, (unsigned int)(ro + c) >> 1);
                o.x = (h & 0xffffu) >= t16 ? o.x * drop_inv : 0.f;
                o.y = (h >> 16) >= t16 ? o.y * drop_inv : 0.f;
            }
            if (dx16) *(unsigned int*)(dx16 + ro + c) = io_pack2(o.x, o.y);
            ax[i].x += o.x, ax[i].y += o.y;
        }
    }
    // eight waves per block hold partial column sums of the same columns: one array at a time through LDS
    for (int a = 0; a < 3; ++a) {
        if (a == 2 && !dxsum) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) *(float2*)&red[wv][(lane + 64 * i) * 2] = a == 0 ? ag[i] : (a == 1 ? ab[i] : ax[i]);
        __syncthreads();
        for (int col = threadIdx.x; col < 128 * NP; col += 512) {
            const float v = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
            if (a == 0) atomicAdd(&dgamma[col], v);
            else if (a == 1) atomicAdd(&dbeta[col], v);
            else atomicAdd(&dxsum[col], dxsum_scale * v);
        }
    }
}

#define LN_DISPATCH(D, CALL)              \
    do {                                  \
        if ((D) <= 64) { CALL(1); }       \
        else if ((D) <= 128) { CALL(2); } \
        else if ((D) <= 256) { CALL(4); } \
        else if ((D) <= 384) { CALL(6); } \
        else if ((D) <= 512) { CALL(8); } \
        else { CALL(LN_MAXV); }           \
    } while (0)

extern "C" int a3t_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                                 float* mean, float* rstd, int M, int D, float eps, void* stream) {
    if (D > 64 * LN_MAXV || M <= 0) return A3T_EINVAL;
    if (D % 128 == 0 && D <= 512 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
        ((uintptr_t)beta % 16 == 0)) {
#define VCALL(NQ)                                                                                                     \
    hipLaunchKernelGGL(ln_fwd_vec_kernel<NQ>, dim3((M + 7) / 8), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, \
                       y_dtype, mean, rstd, M, D, eps)
        if (D == 128) VCALL(1);
        else if (D == 256) VCALL(2);
        else if (D == 384) VCALL(3);
        else VCALL(4);
#undef VCALL
        return (int)hipGetLastError();
    }
#define CALL(V)                                                                                                  \
    hipLaunchKernelGGL(ln_fwd_kernel<V>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, \
                       y_dtype, mean, rstd, M, D, eps)
    LN_DISPATCH(D, CALL);
#undef CALL
    return (int)hipGetLastError();
}

extern "C" int a3t_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                                 const float* mean, const float* rstd, const float* dres, float* dx, void* dx_bf16,
                                 float* dgamma, float* dbeta, float* dx_colsum, float dx_colsum_scale, int M, int D,
                                 float drop_p, uint32_t drop_key, void* stream) {
    if (D > 64 * LN_MAXV || M <= 0) return A3T_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return A3T_EINVAL;
    const unsigned int drop_thr = (unsigned int)((double)drop_p * 4294967296.0);
    const float drop_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 0.f;
    if (D % 128 == 0 && D <= 512 && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dx % 16 == 0) &&
        (!dres || (uintptr_t)dres % 16 == 0) && ((uintptr_t)gamma % 16 == 0)) {
        // 2 workgroups per CU: every workgroup ends with 3*D same-address atomics (dgamma, dbeta, dx column sums), and with
        // 2048 workgroups that serialised tail cost more than the row streaming itself (66 -> 45 us at M = 35840, D = 384)
        const int vb_max = 512;
        // a whole wave per row, rows software-pipelined (round 5; the half-wave-per-row layout it replaced, +0.4 ms per step, left
        // the library in round 6)
        int rb = (M + 7) / 8;
        if (rb > vb_max) rb = vb_max;
#define RCALL(NP)                                                                                                        \
    do {                                                                                                                 \
        if (dy_dtype == A3T_BF16)                                                                                        \
            hipLaunchKernelGGL((ln_bwd_row64_kernel<NP, true>), dim3(rb), dim3(512), 0, (hipStream_t)stream, dy, dy_dtype, x, \
                               gamma, mean, rstd, dres, dx, (unsigned short*)dx_bf16, dgamma, dbeta, dx_colsum,         \
                               dx_colsum_scale, M, D, drop_thr, drop_inv, drop_key);                                     \
        else                                                                                                             \
            hipLaunchKernelGGL((ln_bwd_row64_kernel<NP, false>), dim3(rb), dim3(512), 0, (hipStream_t)stream, dy, dy_dtype, x, \
                               gamma, mean, rstd, dres, dx, (unsigned short*)dx_bf16, dgamma, dbeta, dx_colsum,         \
                               dx_colsum_scale, M, D, drop_thr, drop_inv, drop_key);                                     \
    } while (0)
        if (D == 128) RCALL(1);
        else if (D == 256) RCALL(2);
        else if (D == 384) RCALL(3);
        else RCALL(4);
#undef RCALL
        return (int)hipGetLastError();
    }
    if (drop_p > 0.f) return A3T_EINVAL;   // the fused consumer-dropout output lives in the D % 128 == 0 kernel only
    int blocks = (M + 3) / 4;
    if (blocks > 2048) blocks = 2048;   // 8 blocks/CU: the kernel is latency-bound on its row loads
#define CALL(V)                                                                                                   \
    hipLaunchKernelGGL(ln_bwd_kernel<V>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dy_dtype, x, gamma, mean, \
                       rstd, dres, dx, (unsigned short*)dx_bf16, dgamma, dbeta, dx_colsum, dx_colsum_scale, M, D)
    LN_DISPATCH(D, CALL);
#undef CALL
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// column reductions: block = 64 columns x 4 row lanes; grid (ceil(C/64), row blocks)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col_reduce_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ y,
                                                         const uint8_t* __restrict__ rowmask, double* out0,
                                                         double* out1, int M, int C, int64_t ld, int mode,
                                                         int rows_per_block) {
    __shared__ double red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        float f0 = 0.f, f1 = 0.f;
        int cnt = 0;
        for (int r = r0 + ty; r < r1; r += 4) {
            if (rowmask && !rowmask[r]) continue;
            float v = ldx(x, x_dt, (int64_t)r * ld + c);
            f0 += v;
            if (mode == 1)
                f1 += v * v;
            else if (mode == 2)
                f1 += v * y[(int64_t)r * ld + c];
            if (++cnt == 32) {
                a0 += f0, a1 += f1, f0 = f1 = 0.f, cnt = 0;
            }
        }
        a0 += f0, a1 += f1;
    }
    red[0][ty][tx] = a0;
    red[1][ty][tx] = a1;
    __syncthreads();
    if (ty == 0 && c < C) {
        atomicAdd(&out0[c], red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx]);
        if (mode) atomicAdd(&out1[c], red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx]);
    }
}

// fp32 x, no row mask, C % 64 == 0: block = 64 columns x 16 row lanes, 4 consecutive columns (16 bytes) per lane
__global__ __launch_bounds__(256) void col_reduce_vec_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             double* out0, double* out1, int M, int C, int64_t ld, int mode,
                                                             int rows_per_block) {
    __shared__ float red[2][16][64];
    const int cg = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cg * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
    for (int r = r0 + ry; r < r1; r += 16) {      // <= 16 rows per lane and block in fp32, fp64 across lanes / blocks
        const float4 v = *(const float4*)(x + (int64_t)r * ld + c);
        f0.x += v.x, f0.y += v.y, f0.z += v.z, f0.w += v.w;
        if (mode == 1) {
            f1.x += v.x * v.x, f1.y += v.y * v.y, f1.z += v.z * v.z, f1.w += v.w * v.w;
        } else if (mode == 2) {
            const float4 w = *(const float4*)(y + (int64_t)r * ld + c);
            f1.x += v.x * w.x, f1.y += v.y * w.y, f1.z += v.z * w.z, f1.w += v.w * w.w;
        }
    }
    *(float4*)&red[0][ry][cg * 4] = f0;
    *(float4*)&red[1][ry][cg * 4] = f1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
        if (which == 0 || mode) {
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) a += (double)red[which][q][cc];
            atomicAdd(&(which ? out1 : out0)[blockIdx.x * 64 + cc], a);
        }
    }
}

extern "C" int a3t_col_reduce(const void* x, int x_dtype, const float* y, const uint8_t* rowmask, double* out0,
                              double* out1, int M, int C, int64_t ld, int mode, void* stream) {
    if (M <= 0 || C <= 0) return A3T_EINVAL;
    int rpb = 256;
    dim3 grid((C + 63) / 64, (M + rpb - 1) / rpb);
    if (x_dtype == A3T_F32 && !rowmask && C % 64 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16 == 0) &&
        (mode != 2 || ((uintptr_t)y % 16 == 0))) {
        hipLaunchKernelGGL(col_reduce_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, y, out0, out1, M,
                           C, ld, mode, rpb);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(col_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, rowmask, out0, out1, M,
                       C, ld, mode, rpb);
    return (int)hipGetLastError();
}

__global__ void f64_to_f32_add_kernel(const double* src, float* dst, int n, float scale) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += (float)(src[i] * (double)scale);
}
extern "C" int a3t_f64_to_f32_add(const double* src, float* dst, int n, float scale, void* stream) {
    hipLaunchKernelGGL(f64_to_f32_add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, n,
                       scale);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// BatchNorm1d (+ activation), channels-last
// ------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const double* stats, float* running_mean, float* running_var, float* mean_out,
                                   float* rstd_out, int M, int C, float eps, float momentum, int training) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (training) {
        double mu = stats[c] / (double)M;
        double var = stats[C + c] / (double)M - mu * mu;
        if (var < 0.0) var = 0.0;
        mean_out[c] = (float)mu;
        rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (momentum > 0.f && running_mean) {
            double unb = (M > 1) ? var * (double)M / (double)(M - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    } else {
        mean_out[c] = running_mean[c];
        rstd_out[c] = 1.0f / sqrtf(running_var[c] + eps);
    }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ y,
                                                         int y_dt, int64_t n, int C, int act, int vec) {
    if (vec) {   // C % 4 == 0, 16-byte aligned: 4 consecutive columns per lane
        const int64_t n4 = n >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = q << 2;
            const int c = (int)(i % C);
            const float4 zv = *(const float4*)(z + i);
            const float4 mu = *(const float4*)(mean + c), rs = *(const float4*)(rstd + c);
            const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
            float o[4] = {(zv.x - mu.x) * rs.x * g.x + b.x, (zv.y - mu.y) * rs.y * g.y + b.y,
                          (zv.z - mu.z) * rs.z * g.z + b.z, (zv.w - mu.w) * rs.w * g.w + b.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (act == A3T_ACT_SWISH)
                    o[e] = o[e] * sigmoidf_(o[e]);
                else if (act == A3T_ACT_TANH)
                    o[e] = tanhf(o[e]);
            }
            if (y_dt == A3T_BF16) {
                uint2 h;
                h.x = io_pack2(o[0], o[1]);
                h.y = io_pack2(o[2], o[3]);
                *(uint2*)((unsigned short*)y + i) = h;
            } else {
                *(float4*)((float*)y + i) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        float bn = (z[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
        float o = bn;
        if (act == A3T_ACT_SWISH)
            o = bn * sigmoidf_(bn);
        else if (act == A3T_ACT_TANH)
            o = tanhf(bn);
        stx(y, y_dt, i, o);
    }
}

extern "C" int a3t_bn_act_fwd(const float* z, const double* stats, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* mean_out, float* rstd_out, void* y,
                              int y_dtype, int M, int C, float eps, float momentum, int training, int act,
                              void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, stats, running_mean, running_var,
                       mean_out, rstd_out, M, C, eps, momentum, training);
    int64_t n = (int64_t)M * C;
    const int vec = (C % 4 == 0 && ((uintptr_t)z % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)mean_out % 16 == 0) &&
                     ((uintptr_t)rstd_out % 16 == 0) && ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0)) ? 1 : 0;
    int blocks = (int)(((vec ? n / 4 : n) + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(blocks), dim3(256), 0, s, z, mean_out, rstd_out, gamma, beta, y, y_dtype, n,
                       C, act, vec);
    return (int)hipGetLastError();
}

// d(bn) = dy * act'(bn): recomputed by both backward steps from (dy, z) instead of being materialised
__device__ __forceinline__ float bn_dact(float d, float bn, int act) {
    if (act == A3T_ACT_SWISH) {
        const float sg = sigmoidf_(bn);
        return d * sg * (1.f + bn * (1.f - sg));
    }
    if (act == A3T_ACT_TANH) {
        const float t = tanhf(bn);
        return d * (1.f - t * t);
    }
    return d;
}

// step A (column sums).  Generic: block = 64 columns x 4 row lanes, 4 bytes per lane.
__global__ __launch_bounds__(256) void bn_act_bwd_a_kernel(const void* __restrict__ dy, int dy_dt, const float* __restrict__ z,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           double* sums, int M, int C, int act, int rows_per_block) {
    __shared__ double red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        const float mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c];
        float f0 = 0.f, f1 = 0.f;
        int cnt = 0;
        for (int r = r0 + ty; r < r1; r += 4) {
            int64_t i = (int64_t)r * C + c;
            float zh = (z[i] - mu) * rs;
            float d = bn_dact(ldx(dy, dy_dt, i), zh * g + b, act);
            f0 += d;
            f1 += d * zh;
            if (++cnt == 32) {
                a0 += f0, a1 += f1, f0 = f1 = 0.f, cnt = 0;
            }
        }
        a0 += f0, a1 += f1;
    }
    red[0][ty][tx] = a0;
    red[1][ty][tx] = a1;
    __syncthreads();
    if (ty == 0 && c < C) {
        atomicAdd(&sums[c], red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx]);
        atomicAdd(&sums[C + c], red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx]);
    }
}

// C % 64 == 0, fp32 dy: block = 64 columns x 16 row lanes, every lane owns 4 consecutive columns (16-byte loads)
__global__ __launch_bounds__(256) void bn_act_bwd_a_vec_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, double* sums, int M, int C,
                                                               int act, int rows_per_block) {
    __shared__ float red[2][16][64];
    const int cg = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cg * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const float4 mu = *(const float4*)(mean + c), rs = *(const float4*)(rstd + c);
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
    for (int r = r0 + ry; r < r1; r += 16) {     // <= 16 rows per lane and block: fp32 partials, fp64 across blocks
        const int64_t i = (int64_t)r * C + c;
        const float4 zv = *(const float4*)(z + i), dv = *(const float4*)(dy + i);
        float zh, d;
        zh = (zv.x - mu.x) * rs.x, d = bn_dact(dv.x, zh * g.x + b.x, act), f0.x += d, f1.x += d * zh;
        zh = (zv.y - mu.y) * rs.y, d = bn_dact(dv.y, zh * g.y + b.y, act), f0.y += d, f1.y += d * zh;
        zh = (zv.z - mu.z) * rs.z, d = bn_dact(dv.z, zh * g.z + b.z, act), f0.z += d, f1.z += d * zh;
        zh = (zv.w - mu.w) * rs.w, d = bn_dact(dv.w, zh * g.w + b.w, act), f0.w += d, f1.w += d * zh;
    }
    *(float4*)&red[0][ry][cg * 4] = f0;
    *(float4*)&red[1][ry][cg * 4] = f1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += (double)red[which][q][cc];
        atomicAdd(&sums[which * C + blockIdx.x * 64 + cc], a);
    }
}

extern "C" int a3t_bn_act_bwd_a(const void* dy, int dy_dtype, const float* z, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, double* sums, int M, int C, int act,
                                void* stream) {
    int rpb = 256;
    dim3 grid((C + 63) / 64, (M + rpb - 1) / rpb);
    if (dy_dtype == A3T_F32 && C % 64 == 0 && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)z % 16 == 0) &&
        ((uintptr_t)mean % 16 == 0) && ((uintptr_t)rstd % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
        ((uintptr_t)beta % 16 == 0)) {
        hipLaunchKernelGGL(bn_act_bwd_a_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, z, mean, rstd,
                           gamma, beta, sums, M, C, act, rpb);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(bn_act_bwd_a_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dy_dtype, z, mean, rstd, gamma, beta,
                       sums, M, C, act, rpb);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void bn_act_bwd_b_kernel(const void* __restrict__ dy, int dy_dt, const float* __restrict__ z,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const double* sums,
                                                           float* __restrict__ dz, float* dgamma, float* dbeta,
                                                           int M, int C, int training, int act, int vec) {
    const int64_t n = (int64_t)M * C;
    const double invM = 1.0 / (double)M;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            dgamma[c] += (float)sums[C + c];
            dbeta[c] += (float)sums[c];
        }
    }
    if (vec) {   // C % 4 == 0, fp32 dy, 16-byte aligned: 4 consecutive columns per lane
        const int64_t n4 = n >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = q << 2;
            const int c = (int)(i % C);
            const float4 zv = *(const float4*)(z + i), dv = *(const float4*)((const float*)dy + i);
            const float4 mu = *(const float4*)(mean + c), rs = *(const float4*)(rstd + c);
            const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
            const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
            const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
            const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float zh = (zz[e] - mm[e]) * rr[e];
                float d = bn_dact(dd[e], zh * gg[e] + bb[e], act);
                if (training) d = d - (float)(sums[c + e] * invM) - zh * (float)(sums[C + c + e] * invM);
                o[e] = gg[e] * rr[e] * d;
            }
            *(float4*)(dz + i) = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        float rs = rstd[c];
        float zh = (z[i] - mean[c]) * rs;
        float d = bn_dact(ldx(dy, dy_dt, i), zh * gamma[c] + beta[c], act);
        if (training) d = d - (float)(sums[c] * invM) - zh * (float)(sums[C + c] * invM);
        dz[i] = gamma[c] * rs * d;
    }
}

extern "C" int a3t_bn_act_bwd_b(const void* dy, int dy_dtype, const float* z, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, const double* sums, float* dz, float* dgamma,
                                float* dbeta, int M, int C, int training, int act, void* stream) {
    int64_t n = (int64_t)M * C;
    const int vec = (dy_dtype == A3T_F32 && C % 4 == 0 && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)z % 16 == 0) &&
                     ((uintptr_t)dz % 16 == 0) && ((uintptr_t)mean % 16 == 0) && ((uintptr_t)rstd % 16 == 0) &&
                     ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0)) ? 1 : 0;
    int blocks = (int)(((vec ? n / 4 : n) + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_act_bwd_b_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dy_dtype, z, mean, rstd,
                       gamma, beta, sums, dz, dgamma, dbeta, M, C, training, act, vec);
    return (int)hipGetLastError();
}
