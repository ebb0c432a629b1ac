// misc.hip -- encoder prologue, masked L1/L2 loss, flat-buffer clip+Adam, ParallelWaveGAN
// element-wise helpers, counter-based dropout.  All HBM-bound, coalesced row-major access.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"
#include "dtype_io.h"

static inline int nblocks(int64_t n, int cap = 4096) {
    int64_t b = (n + 255) / 256;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- encoder prologue
__global__ void mask_fill_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                 const float* __restrict__ mf, void* __restrict__ y, int y_dt, int64_t n, int C) {
    GRID_STRIDE(i, n) {
        int64_t r = i / C;
        int c = (int)(i - r * C);
        stx(y, y_dt, i, m[r] ? mf[c] : x[i]);
    }
}
extern "C" int a3t_mask_fill(const float* speech, const uint8_t* masked, const float* mask_feature, void* out,
                             int out_dtype, int M, int C, void* stream) {
    int64_t n = (int64_t)M * C;
    hipLaunchKernelGGL(mask_fill_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, speech, masked,
                       mask_feature, out, out_dtype, n, C);
    return (int)hipGetLastError();
}

__global__ void embed_finish_fwd_kernel(const float* __restrict__ e, const float* __restrict__ emb,
                                        const float* __restrict__ seg, const int64_t* __restrict__ text,
                                        const int64_t* __restrict__ spos, const int64_t* __restrict__ tpos,
                                        float* __restrict__ xs, int B, int Tm, int Tp, int D, float xscale,
                                        unsigned int thr, float inv, unsigned int key, const float* __restrict__ spk) {
    const int T = Tm + Tp;
    const int64_t n = (int64_t)B * T * D;
    GRID_STRIDE(i, n) {
        int64_t row = i / D;
        int c = (int)(i - row * D);
        int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
        float v, sg;
        if (t < Tm) {
            int64_t r = (int64_t)b * Tm + t;
            v = fmaxf(e[r * D + c], 0.f) * xscale;
            sg = seg[spos[r] * D + c];
        } else {
            int64_t r = (int64_t)b * Tp + (t - Tm);
            v = emb[text[r] * D + c] * xscale;
            sg = seg[tpos[r] * D + c];
        }
        if (inv > 0.f) v = rng_keep(key, (unsigned int)i, thr) ? v * inv : 0.f;   // positional dropout, before + seg
        if (spk) sg += spk[(int64_t)b * D + c];     // projected speaker embedding, the same vector for every token of b
        xs[i] = v + sg;
    }
}
extern "C" int a3t_embed_finish_fwd(const float* e, const float* emb, const float* seg, const int64_t* text,
                                    const int64_t* spos, const int64_t* tpos, float* xs, int B, int Tm, int Tp, int D,
                                    float xscale, float drop_p, uint32_t drop_key, const float* spk, void* stream) {
    int64_t n = (int64_t)B * (Tm + Tp) * D;
    hipLaunchKernelGGL(embed_finish_fwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, e, emb, seg, text,
                       spos, tpos, xs, B, Tm, Tp, D, xscale, (unsigned int)((double)drop_p * 4294967296.0),
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 0.f, drop_key, spk);
    return (int)hipGetLastError();
}

__global__ void embed_finish_bwd_kernel(const float* __restrict__ dxs, const float* __restrict__ e,
                                        const int64_t* __restrict__ text, const int64_t* __restrict__ spos,
                                        const int64_t* __restrict__ tpos, float* __restrict__ de, float* demb,
                                        float* dseg, int B, int Tm, int Tp, int D, int V, int nseg, float xscale,
                                        unsigned int thr, float inv, unsigned int key) {
    const int T = Tm + Tp;
    const int64_t n = (int64_t)B * T * D;
    GRID_STRIDE(i, n) {
        int64_t row = i / D;
        int c = (int)(i - row * D);
        int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
        float g = dxs[i];
        float gd = g;                                            // gradient through the positional dropout
        if (inv > 0.f) gd = rng_keep(key, (unsigned int)i, thr) ? g * inv : 0.f;
        if (t < Tm) {
            int64_t r = (int64_t)b * Tm + t;
            de[r * D + c] = (e[r * D + c] > 0.f) ? gd * xscale : 0.f;
            int64_t s = spos[r];
            if (s != nseg - 1) atomicAdd(&dseg[s * D + c], g);  // padding_idx=-1 -> last row gets no grad
        } else {
            int64_t r = (int64_t)b * Tp + (t - Tm);
            int64_t tok = text[r];
            if (tok != V - 1) atomicAdd(&demb[tok * D + c], gd * xscale);
            int64_t s = tpos[r];
            if (s != nseg - 1) atomicAdd(&dseg[s * D + c], g);
        }
    }
}
extern "C" int a3t_embed_finish_bwd(const float* dxs, const float* e, const int64_t* text, const int64_t* spos,
                                    const int64_t* tpos, float* de, float* demb, float* dseg, int B, int Tm, int Tp,
                                    int D, int V, int nseg, float xscale, float drop_p, uint32_t drop_key,
                                    void* stream) {
    int64_t n = (int64_t)B * (Tm + Tp) * D;
    hipLaunchKernelGGL(embed_finish_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, dxs, e, text, spos,
                       tpos, de, demb, dseg, B, Tm, Tp, D, V, nseg, xscale,
                       (unsigned int)((double)drop_p * 4294967296.0), drop_p > 0.f ? 1.f / (1.f - drop_p) : 0.f,
                       drop_key);
    return (int)hipGetLastError();
}

__global__ void scale_kernel(const float* x, float* y, int64_t n, float s) {
    GRID_STRIDE(i, n) y[i] = x[i] * s;
}
extern "C" int a3t_scale(const float* x, float* y, int64_t n, float s, void* stream) {
    hipLaunchKernelGGL(scale_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, s);
    return (int)hipGetLastError();
}
__global__ void scale_dev_kernel(const float* x, float* y, int64_t n, const float* s) {
    const float k = s[0];
    GRID_STRIDE(i, n) y[i] = x[i] * k;
}
extern "C" int a3t_scale_dev(const float* x, float* y, int64_t n, const float* s, void* stream) {
    hipLaunchKernelGGL(scale_dev_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, s);
    return (int)hipGetLastError();
}
__global__ void axpy_kernel(const float* x, float* y, int64_t n, float a) {
    GRID_STRIDE(i, n) y[i] += a * x[i];
}
extern "C" int a3t_axpy(const float* x, float* y, int64_t n, float a, void* stream) {
    hipLaunchKernelGGL(axpy_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, a);
    return (int)hipGetLastError();
}
// Fold of the slot-spread column sums of the attention data-gradient GEMMs (a3t_gemm_desc::colsum_slots):
// slots[s][0:d] = colsum(d(q+u)), [d:2d] = colsum(d(q+v)), [2d:3d] = colsum(dK), [3d:4d] = colsum(dV)
//   -> d pos_bias_u += su, d pos_bias_v += sv, d b_qkv += (su + sv | sk | sV)      (attention.py:190-196: q feeds both)
__global__ void attn_bias_fold_kernel(const float* __restrict__ slots, int S, int d, float* gu, float* gv, float* gbqkv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * d) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += slots[(int64_t)s * 4 * d + i];
    const int part = i / d, c = i - part * d;
    if (part == 0) {
        gu[c] += a;
        atomicAdd(&gbqkv[c], a);       // (the u and v lanes of one column both add into d b_q)
    } else if (part == 1) {
        gv[c] += a;
        atomicAdd(&gbqkv[c], a);
    } else {
        gbqkv[(part - 1) * d + c] += a;
    }
}
extern "C" int a3t_attn_bias_fold(const float* slots, int S, int d, float* gu, float* gv, float* gbqkv, void* stream) {
    if (S < 1 || d < 1) return A3T_EINVAL;
    hipLaunchKernelGGL(attn_bias_fold_kernel, dim3((4 * d + 255) / 256), dim3(256), 0, (hipStream_t)stream, slots, S, d, gu,
                       gv, gbqkv);
    return (int)hipGetLastError();
}
__global__ void cast_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n4) {
    GRID_STRIDE(i, n4) {
        float4 v = ((const float4*)x)[i];
        uint2 o;
        o.x = io_pack2(v.x, v.y);
        o.y = io_pack2(v.z, v.w);
        ((uint2*)y)[i] = o;
    }
}
extern "C" int a3t_cast_bf16(const float* x, void* y, int64_t n, void* stream) {
    if (n % 4 || ((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return A3T_EINVAL;
    hipLaunchKernelGGL(cast_kernel, dim3(nblocks(n / 4, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)y, n / 4);
    return (int)hipGetLastError();
}
// Transposed bf16 shadow of Conv1d weights for the data gradient as a k-contiguous GEMM (gemm_bf16_8p.hip):
// Wt[c][taps-1-t][n] = bf16(W[n][t][c]).  `count` weights of identical shape at element offsets src_off[i] / dst_off[i] of
// the flat fp32 parameter buffer / the shadow buffer, one launch.  32 x 32 tiles of the (n, c) plane through LDS.
__global__ __launch_bounds__(256) void cast_conv_t_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                         const int64_t* __restrict__ src_off, const int64_t* __restrict__ dst_off,
                                                         int N, int taps, int C) {
    __shared__ float tile[32][33];
    const float* W = src + src_off[blockIdx.z];
    unsigned short* Wt = dst + dst_off[blockIdx.z];
    const int tc = (C + 31) / 32;
    const int t = blockIdx.y, n0 = (blockIdx.x / tc) * 32, c0 = (blockIdx.x % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        tile[r][tx] = (n < N && c < C) ? W[((int64_t)n * taps + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        if (c < C && n < N) Wt[((int64_t)c * taps + (taps - 1 - t)) * N + n] = io_f2bf(tile[tx][r]);
    }
}
extern "C" int a3t_cast_bf16_conv_t(const float* src, void* dst, const int64_t* src_off, const int64_t* dst_off, int count, int N,
                                    int taps, int C, void* stream) {
    if (count <= 0 || N <= 0 || taps <= 0 || C <= 0) return A3T_EINVAL;
    dim3 grid((unsigned)(((N + 31) / 32) * ((C + 31) / 32)), (unsigned)taps, (unsigned)count);
    hipLaunchKernelGGL(cast_conv_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, src_off, dst_off, N, taps, C);
    return (int)hipGetLastError();
}
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): two bf16 GEMMs on (hi, lo) reproduce the fp32 operand to ~2^-17
__global__ void split_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                  int64_t n4) {
    GRID_STRIDE(i, n4) {
        const float4 v = ((const float4*)x)[i];
        uint2 h, l;
        h.x = io_pack2(v.x, v.y);
        h.y = io_pack2(v.z, v.w);
        l.x = io_pack2(v.x - io_bf2f(h.x & 0xffff), v.y - io_bf2f(h.x >> 16));
        l.y = io_pack2(v.z - io_bf2f(h.y & 0xffff), v.w - io_bf2f(h.y >> 16));
        ((uint2*)hi)[i] = h;
        ((uint2*)lo)[i] = l;
    }
}
extern "C" int a3t_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream) {
    if (n % 4 || ((uintptr_t)x & 15) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7)) return A3T_EINVAL;
    hipLaunchKernelGGL(split_bf16_kernel, dim3(nblocks(n / 4, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)hi, (unsigned short*)lo, n / 4);
    return (int)hipGetLastError();
}
__global__ void slice_rows_kernel(const float* x, void* y, int y_dt, int B, int T, int Tm, int D, int reverse) {
    const int64_t n = (int64_t)B * Tm * D;
    GRID_STRIDE(i, n) {
        int64_t r = i / D;
        int c = (int)(i - r * D);
        int b = (int)(r / Tm), t = (int)(r - (int64_t)b * Tm);
        int64_t j = ((int64_t)b * T + t) * D + c;
        if (reverse)
            ((float*)x)[j] += ldx(y, y_dt, i);
        else
            stx(y, y_dt, i, x[j]);
    }
}
extern "C" int a3t_slice_rows(const float* x, void* y, int y_dtype, int B, int T, int Tm, int D, int reverse_add,
                              void* stream) {
    int64_t n = (int64_t)B * Tm * D;
    hipLaunchKernelGGL(slice_rows_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, y_dtype, B, T, Tm,
                       D, reverse_add);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- masked L1/L2 loss
// stage 1: one wave per row, block partial (sum, count) -> scratch[2 + 2*blk]; deterministic.
#define LOSS_BLOCKS 512
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ before,
                                                           const float* __restrict__ after,
                                                           const float* __restrict__ target,
                                                           const uint8_t* __restrict__ masked, float* scratch, int M,
                                                           int C, int l2) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.f, cnt = 0.f;
    for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
        if (!masked[row]) continue;
        cnt += 1.f;
        for (int c = lane; c < C; c += 64) {
            int64_t i = (int64_t)row * C + c;
            float y = target[i], d0 = before[i] - y;
            s += l2 ? d0 * d0 : fabsf(d0);
            if (after) {   // (sedit_model.py:333-337: the after-postnet term only exists when there is a postnet)
                float d1 = after[i] - y;
                s += l2 ? d1 * d1 : fabsf(d1);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        red[0][wv] = s;
        red[1][wv] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[2 + 2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        scratch[3 + 2 * blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
__global__ void loss_final_kernel(float* scratch, float* loss_out, int nblk) {
    __shared__ double rs[256], rc[256];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        s += (double)scratch[2 + 2 * i];
        c += (double)scratch[3 + 2 * i];
    }
    rs[threadIdx.x] = s;
    rc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            rs[threadIdx.x] += rs[threadIdx.x + o];
            rc[threadIdx.x] += rc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float n = (float)rc[0];
        float inv = 1.0f / (n + 1e-10f);
        loss_out[0] = (float)rs[0] * inv;
        scratch[0] = inv;
        scratch[1] = n;
    }
}
__global__ void loss_grad_kernel(const float* __restrict__ before, const float* __restrict__ after,
                                 const float* __restrict__ target, const uint8_t* __restrict__ masked,
                                 const float* scratch, float* __restrict__ db, float* __restrict__ da, int64_t n, int C,
                                 int l2, float gscale) {
    const float k = scratch[0] * gscale;
    GRID_STRIDE(i, n) {
        int64_t r = i / C;
        float g0 = 0.f, g1 = 0.f;
        if (masked[r]) {
            float y = target[i], d0 = before[i] - y, d1 = after ? after[i] - y : 0.f;
            if (l2) {
                g0 = 2.f * d0 * k;
                g1 = 2.f * d1 * k;
            } else {
                g0 = (d0 > 0.f ? k : (d0 < 0.f ? -k : 0.f));
                g1 = (d1 > 0.f ? k : (d1 < 0.f ? -k : 0.f));
            }
        }
        db[i] = g0;
        if (da) da[i] = g1;
    }
}
extern "C" int a3t_mlm_loss_scratch_floats(int M) {
    (void)M;
    return 2 + 2 * LOSS_BLOCKS;
}
extern "C" int a3t_mlm_loss(const float* before, const float* after, const float* target, const uint8_t* masked,
                            float* loss_out, float* d_before, float* d_after, float* scratch, int M, int C, int l2,
                            float gscale, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int nblk = (M + 3) / 4;
    if (nblk > LOSS_BLOCKS) nblk = LOSS_BLOCKS;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(nblk), dim3(256), 0, s, before, after, target, masked, scratch, M, C,
                       l2);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, scratch, loss_out, nblk);
    if (d_before && (d_after || !after)) {
        int64_t n = (int64_t)M * C;
        hipLaunchKernelGGL(loss_grad_kernel, dim3(nblocks(n)), dim3(256), 0, s, before, after, target, masked, scratch,
                           d_before, d_after, n, C, l2, gscale);
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- clip + Adam on one flat buffer
#define SUMSQ_BLOCKS 1024
// (16-byte loads, four independent fp32 partial sums per lane folded into the fp64 total every 32 vectors: the scalar
//  version was latency-bound at 0.6 TB/s -- 194 us for the 28 M gradients of configs[1]; this one streams at HBM speed)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* partial) {
    __shared__ double red[256];
    double s = 0.0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (((uintptr_t)g) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
    int cnt = 0;
    for (int64_t i = tid; i < n4; i += nthr) {
        const float4 v = ((const float4*)g)[i];
        f0 += v.x * v.x, f1 += v.y * v.y, f2 += v.z * v.z, f3 += v.w * v.w;
        if (++cnt == 32) {
            s += (double)f0 + (double)f1 + (double)f2 + (double)f3, f0 = f1 = f2 = f3 = 0.f, cnt = 0;
        }
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += nthr) {
        const float v = g[i];
        f0 += v * v;
    }
    s += (double)f0 + (double)f1 + (double)f2 + (double)f3;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// one Adam element (torch.optim.Adam with the bias corrections folded into step_size / bc2s)
__device__ __forceinline__ void adam_elem(float& p, const float g, float& m, float& v, const float coef, const float b1, const float b2,
                                          const float step_size, const float bc2s, const float eps) {
    const float gi = g * coef;
    m = m * b1 + (1.f - b1) * gi;
    v = v * b2 + (1.f - b2) * gi * gi;
    p -= step_size * m / (sqrtf(v) / bc2s + eps);
}
// the flat buffers of one optimizer step, 16 bytes per lane and access when the four pointers allow it
__device__ __forceinline__ void adam_sweep(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                           const int64_t n, const float coef, const float b1, const float b2, const float step_size,
                                           const float bc2s, const float eps) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t i = tid; i < n4; i += nthr) {
        float4 pi = ((float4*)p)[i], mi = ((float4*)m)[i], vi = ((float4*)v)[i];
        const float4 gi = ((const float4*)g)[i];
        adam_elem(pi.x, gi.x, mi.x, vi.x, coef, b1, b2, step_size, bc2s, eps);
        adam_elem(pi.y, gi.y, mi.y, vi.y, coef, b1, b2, step_size, bc2s, eps);
        adam_elem(pi.z, gi.z, mi.z, vi.z, coef, b1, b2, step_size, bc2s, eps);
        adam_elem(pi.w, gi.w, mi.w, vi.w, coef, b1, b2, step_size, bc2s, eps);
        ((float4*)m)[i] = mi, ((float4*)v)[i] = vi, ((float4*)p)[i] = pi;
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += nthr) adam_elem(p[i], g[i], m[i], v[i], coef, b1, b2, step_size, bc2s, eps);
}
extern "C" int a3t_sumsq(const float* g, int64_t n, double* partial, void* stream) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, n, partial);
    return (int)hipGetLastError();
}
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const double* partial, float* norm_out, int64_t n, float lr,
                                                        float b1, float b2, float eps, float bc1, float bc2s,
                                                        float clip, float gscale) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < SUMSQ_BLOCKS; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float norm = (float)sqrt(red[0]) * fabsf(gscale);
    if (blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
    if (!isfinite(norm)) return;  // trainer.py:640-656: non-finite grad norm -> skip the update
    float coef = clip > 0.f ? clip / (norm + 1e-6f) : 1.f;
    coef = (coef > 1.f ? 1.f : coef) * gscale;
    const float step_size = lr / bc1;
    adam_sweep(p, g, m, v, n, coef, b1, b2, step_size, bc2s, eps);
}
extern "C" int a3t_clip_adam(float* p, const float* g, float* m, float* v, const double* partial, float* norm_out,
                             int64_t n, float lr, float beta1, float beta2, float eps, int step, float clip,
                             float gscale, void* stream) {
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(clip_adam_kernel, dim3(nblocks(n / 4, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, partial,
                       norm_out, n, lr, beta1, beta2, eps, bc1, bc2s, clip, gscale);
    return (int)hipGetLastError();
}

// Same update with the optimizer step count and the Noam learning rate kept ON THE DEVICE: state[0] = number of updates
// applied so far, state[1] = number of skipped (non-finite gradient norm) steps.  A skipped step advances neither Adam's
// bias correction nor the LR schedule (trainer.py:640-679: optimizer.step() and scheduler.step() are both skipped), and
// the host never has to read the norm back.
__global__ __launch_bounds__(256) void clip_adam_noam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v,
                                                             const double* partial, float* norm_out, int64_t n,
                                                             const int* state, float base_lr, float model_size,
                                                             float warmup, float b1, float b2, float eps, float clip,
                                                             float gscale) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < SUMSQ_BLOCKS; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float norm = (float)sqrt(red[0]) * fabsf(gscale);
    if (blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
    if (!isfinite(norm)) return;
    const double t = (double)(state[0] + 1);
    // NoamLR.get_lr (schedulers/noam_lr.py:58-65), torch.optim.Adam bias corrections
    const float lr = (float)((double)base_lr * pow((double)model_size, -0.5) * fmin(pow(t, -0.5), t * pow((double)warmup, -1.5)));
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    float coef = clip > 0.f ? clip / (norm + 1e-6f) : 1.f;
    coef = (coef > 1.f ? 1.f : coef) * gscale;
    const float step_size = lr / bc1;
    adam_sweep(p, g, m, v, n, coef, b1, b2, step_size, bc2s, eps);
}
__global__ void adam_state_advance_kernel(const float* norm, int* state) {
    if (isfinite(norm[0])) state[0] += 1; else state[1] += 1;
}
extern "C" int a3t_clip_adam_noam(float* p, const float* g, float* m, float* v, const double* partial, float* norm_out,
                                  int64_t n, int* state, float base_lr, float model_size, float warmup, float beta1,
                                  float beta2, float eps, float clip, float gscale, void* stream) {
    hipLaunchKernelGGL(clip_adam_noam_kernel, dim3(nblocks(n / 4, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       partial, norm_out, n, state, base_lr, model_size, warmup, beta1, beta2, eps, clip, gscale);
    hipLaunchKernelGGL(adam_state_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, norm_out, state);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- ParallelWaveGAN helpers
__global__ void pwg_gate_kernel(const float* __restrict__ y, const float* __restrict__ c, float* __restrict__ out,
                                int64_t n, int H) {
    GRID_STRIDE(i, n) {
        int64_t t = i / H;
        int h = (int)(i - t * H);
        int64_t j = t * 2 * H + h;
        float a = y[j] + c[j], b = y[j + H] + c[j + H];
        out[i] = tanhf(a) * (1.f / (1.f + __expf(-b)));
    }
}
extern "C" int a3t_pwg_gate(const float* y, const float* c, float* out, int64_t T, int H, void* stream) {
    int64_t n = T * H;
    hipLaunchKernelGGL(pwg_gate_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, y, c, out, n, H);
    return (int)hipGetLastError();
}
__global__ void pwg_res_skip_kernel(const float* __restrict__ o, float* __restrict__ x, float* __restrict__ skips,
                                    int64_t T, int R, int S) {
    const int W = R + S;
    const int64_t n = T * W;
    GRID_STRIDE(i, n) {
        int64_t t = i / W;
        int c = (int)(i - t * W);
        if (c < R)
            x[t * R + c] = (o[i] + x[t * R + c]) * 0.70710678118654752440f;
        else
            skips[t * S + (c - R)] += o[i];
    }
}
extern "C" int a3t_pwg_res_skip(const float* o, float* x, float* skips, int64_t T, int R, int S, void* stream) {
    int64_t n = T * (R + S);
    hipLaunchKernelGGL(pwg_res_skip_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, o, x, skips, T, R, S);
    return (int)hipGetLastError();
}
// out[b][t][c] = sum_j w[j] * c_stretched[b][t + j - scale][c], c_stretched[b][u] = c[b][u / scale], zero outside the utterance
__global__ void pwg_upsample_kernel(const float* __restrict__ c, const float* __restrict__ w, float* __restrict__ out,
                                    int64_t B, int64_t Tin, int C, int scale) {
    const int64_t Tout = Tin * scale;
    const int64_t n = B * Tout * C;
    GRID_STRIDE(i, n) {
        const int64_t ta = i / C;
        const int ch = (int)(i - ta * C);
        const int64_t b = ta / Tout, t = ta - b * Tout;
        const float* cb = c + b * Tin * C;
        float acc = 0.f;
        for (int j = 0; j <= 2 * scale; ++j) {
            int64_t u = t + j - scale;
            if (u >= 0 && u < Tout) acc += w[j] * cb[(u / scale) * C + ch];
        }
        out[i] = acc;
    }
}
extern "C" int a3t_pwg_upsample(const float* c, const float* w, float* out, int64_t B, int64_t Tin, int C, int scale,
                                void* stream) {
    int64_t n = B * Tin * scale * C;
    hipLaunchKernelGGL(pwg_upsample_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, c, w, out, B, Tin, C,
                       scale);
    return (int)hipGetLastError();
}
__global__ void replicate_pad_kernel(const float* x, float* y, int64_t B, int64_t T, int C, int pad) {
    const int64_t Tp = T + 2 * pad, n = B * Tp * C;
    GRID_STRIDE(i, n) {
        const int64_t ta = i / C;
        const int c = (int)(i - ta * C);
        const int64_t b = ta / Tp;
        int64_t t = ta - b * Tp - pad;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        y[i] = x[(b * T + t) * C + c];
    }
}
extern "C" int a3t_replicate_pad(const float* x, float* y, int64_t B, int64_t T, int C, int pad, void* stream) {
    int64_t n = B * (T + 2 * pad) * C;
    hipLaunchKernelGGL(replicate_pad_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, B, T, C, pad);
    return (int)hipGetLastError();
}
__global__ void bias_act_kernel(float* x, const float* bias, int64_t n, int C, int act, float scale) {
    GRID_STRIDE(i, n) {
        float v = x[i] * scale;
        if (bias) v += bias[i % C];
        if (act == A3T_ACT_RELU)
            v = fmaxf(v, 0.f);
        else if (act == A3T_ACT_TANH)
            v = tanhf(v);
        x[i] = v;
    }
}
extern "C" int a3t_bias_act(float* x, const float* bias, int64_t M, int C, int act, float scale, void* stream) {
    int64_t n = M * C;
    hipLaunchKernelGGL(bias_act_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, bias, n, C, act, scale);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- dropout (counter RNG, dtype_io.h)
__global__ void dropout_kernel(const void* x, int x_dt, void* y, int y_dt, int64_t n, unsigned int thr, float inv,
                               unsigned int key, float scale) {
    GRID_STRIDE(i, n) {
        float v = ldx(x, x_dt, i) * scale;
        stx(y, y_dt, i, rng_keep(key, (unsigned int)i, thr) ? v * inv : 0.f);
    }
}
extern "C" int a3t_dropout(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, float p, uint32_t key,
                           float scale, void* stream) {
    if (p < 0.f || p >= 1.f) return A3T_EINVAL;
    hipLaunchKernelGGL(dropout_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, n,
                       (unsigned int)((double)p * 4294967296.0), 1.f / (1.f - p), key, scale);
    return (int)hipGetLastError();
}
// backward of "x + alpha*dropout(branch)": gm = g * mask/(1-p) (operand of the branch's GEMMs, fp32 or bf16)
// and colsum[c] += colsum_scale * sum_m gm[m][c] (the branch's bias gradient)
__global__ __launch_bounds__(256) void dropout_bwd_cast_kernel(const float* __restrict__ g, void* __restrict__ gm,
                                                               int gm_dt, float* colsum, float colsum_scale, int M,
                                                               int C, unsigned int thr, float inv, unsigned int key,
                                                               int rows_per_block) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < C)
        for (int r = r0 + ty; r < r1; r += 4) {
            int64_t i = (int64_t)r * C + c;
            float v = rng_keep(key, (unsigned int)i, thr) ? g[i] * inv : 0.f;
            stx(gm, gm_dt, i, v);
            s += v;
        }
    red[ty][tx] = s;
    __syncthreads();
    if (colsum && ty == 0 && c < C)
        atomicAdd(&colsum[c], colsum_scale * (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]));
}
extern "C" int a3t_dropout_bwd_cast(const float* g, void* gm, int gm_dtype, float* colsum, float colsum_scale, int M,
                                    int C, float p, uint32_t key, void* stream) {
    if (p < 0.f || p >= 1.f) return A3T_EINVAL;
    int rpb = 128;
    dim3 grid((C + 63) / 64, (M + rpb - 1) / rpb);
    hipLaunchKernelGGL(dropout_bwd_cast_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, gm, gm_dtype, colsum,
                       colsum_scale, M, C, (unsigned int)((double)p * 4294967296.0), 1.f / (1.f - p), key, rpb);
    return (int)hipGetLastError();
}

extern "C" const char* a3t_version(void) { return "a3t_hip 0.1 (gfx950)"; }


// out[b][c] += sum_t x[b*T + t][c] (fp32): the gradient of a per-utterance vector that was added to every token of the
// utterance (x-vector conditioning, configs[3]) -- one launch for all utterances instead of one column-sum pass each
__global__ __launch_bounds__(256) void segment_colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, b = blockIdx.y;
    float s = 0.f;
    if (c < C)
        for (int t = rl; t < T; t += 4) s += x[((int64_t)b * T + t) * C + c];
    part[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < C) out[(int64_t)b * C + c] += part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

extern "C" int a3t_segment_colsum(const float* x, float* out, int B, int T, int C, void* stream) {
    if (B <= 0 || T <= 0 || C <= 0 || B > 65535) return A3T_EINVAL;
    hipLaunchKernelGGL(segment_colsum_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, out, T, C);
    return (int)hipGetLastError();
}


// On-device half of MLMCollateFn (collate_fn.py:330-385): paints masked_position, the segment ids and the padding masks
// from the integer span lists.  The numpy-RNG draws that pick the phones stay on the host (the global-RNG contract);
// what they produce -- sel[b][j] = phone j of utterance b is masked -- comes in with the frame indices of the phones.
//   masked[b][t]  = (any selected phone j < L_b with fs <= t < fe  OR any explicit span of utterance b covers t) AND t < flen_b
//   sp[b][t]      = (last j < L_b with fs <= t < fe) + 1, or 0        ("later phones overwrite earlier ones", :335-341)
//   tp[b][j]      = j + 1 for j < L_b, else 0;   speech_mask = t < flen_b;   text_mask = j < tlen_b
__global__ __launch_bounds__(256) void collate_paint_kernel(const int* __restrict__ fs, const int* __restrict__ fe,
                                                            const int* __restrict__ alen, const unsigned char* __restrict__ sel,
                                                            const int* __restrict__ mspan, const int* __restrict__ nms,
                                                            const int* __restrict__ flen, const int* __restrict__ tlen,
                                                            unsigned char* __restrict__ masked, unsigned char* __restrict__ smask,
                                                            unsigned char* __restrict__ tmask, int64_t* __restrict__ sp,
                                                            int64_t* __restrict__ tp, int Tm, int Tp, int P, int S, int sega) {
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int L = alen[b] < P ? alen[b] : P;
    if (t < Tm) {
        int seg = 0;
        bool m = false;
        for (int j = 0; j < L; ++j) {
            const int s0 = fs[b * P + j], e0 = fe[b * P + j];
            if (t >= s0 && t < e0) {
                seg = j + 1;
                m = m || (sel[b * P + j] != 0);
            }
        }
        const int ns = nms[b];
        for (int q = 0; q < ns; ++q) m = m || (t >= mspan[(b * S + q) * 2] && t < mspan[(b * S + q) * 2 + 1]);
        const bool valid = t < flen[b];
        masked[(int64_t)b * Tm + t] = (m && valid) ? 1 : 0;
        smask[(int64_t)b * Tm + t] = valid ? 1 : 0;
        sp[(int64_t)b * Tm + t] = sega ? seg : 0;
    }
    if (t < Tp) {
        tmask[(int64_t)b * Tp + t] = t < tlen[b] ? 1 : 0;
        tp[(int64_t)b * Tp + t] = (sega && t < L) ? t + 1 : 0;
    }
}

extern "C" int a3t_collate_paint(const int* fs, const int* fe, const int* alen, const uint8_t* sel, const int* mspan,
                                 const int* nms, const int* flen, const int* tlen, uint8_t* masked, uint8_t* speech_mask,
                                 uint8_t* text_mask, int64_t* sp, int64_t* tp, int B, int Tm, int Tp, int P, int S,
                                 int sega_emb, void* stream) {
    if (B <= 0 || B > 65535 || Tm <= 0 || Tp < 0 || P < 0 || S < 0) return A3T_EINVAL;
    const int n = Tm > Tp ? Tm : Tp;
    hipLaunchKernelGGL(collate_paint_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       fs, fe, alen, sel, mspan, nms, flen, tlen, masked, speech_mask, text_mask, sp, tp, Tm, Tp, P, S, sega_emb);
    return (int)hipGetLastError();
}
