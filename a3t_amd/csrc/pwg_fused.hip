// pwg_fused.hip -- ParallelWaveGAN residual block (espnet2/gan_tts/wavenet/residual_block.py:114-169) as two fused
// fp32-MFMA kernels for gfx950, channels-last [T][C] tensors.
//
//   stage 0:  g[t][c] = tanh(ya) * sigmoid(yb),   y = dilated Conv1d_k3(x)[t] + Conv1x1_aux(cu)[t] + b      (128 gate ch)
//   stage 1:  o = Conv1x1_out(g) + b;   x[t] = (o[:64] + x[t]) * sqrt(1/2);   skips[t] += o[64:]
//
// The layer-by-layer path (three GEMM launches + two element-wise kernels per block) moves ~10 GB per block at
// B x T = 8 x 300 000 samples and spends 75 % of its time in the generic register-staged fp32 GEMM.  Here the block's
// whole weight matrix lives in LDS for the lifetime of a persistent workgroup (stage 0: [272 = 3 taps x 64 + 80 aux][128]
// fp32 = 136 KiB, stage 1: [64][128] = 32 KiB), the 256-sample activation tiles stream through a double-buffered
// k-major LDS slab, v_mfma_f32_32x32x2f32 keeps exact fp32 products, and the gate / residual / skip arithmetic happens on
// the accumulators: y, the aux projection and o never reach HBM (~3.2 GB per block instead of ~10).
// Stage-0 output columns are permuted on the host so that a wave's two 32-column MFMA blocks hold the tanh and the
// sigmoid pre-activation of the SAME 32 channels: the gate needs no cross-lane exchange.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwgArgs {
    const float* x_in;    // [B*Tw][64]   stage 0: conv input;  stage 1: residual input (may alias x_out)
    const float* cu;      // [B*Tw][80]   stage 0: upsampled mel
    const float* gin;     // [B*Tw][64]   stage 1: gated activation
    const float* wt;      // [K][128] k-major weights (stage 0: permuted columns)
    const float* bias;    // [128]
    float* g;             // stage 0 out
    float* x_out;         // stage 1 out
    float* skips;         // stage 1 accumulate
    int B, Tw, dil, tiles_t;
};

template <int STAGE>
__global__ __launch_bounds__(512, 2) void pwg_stage_kernel(PwgArgs a) {
    // 8 waves = 4 (rows) x 2 (columns), each a 64 x 64 sub-tile of the 256-sample tile: two waves per SIMD, so one wave's
    // epilogue / LDS latency is covered by its partner's MFMAs (the weights leave room for ONE workgroup per CU only)
    // stage 0: K-chunks of 8 (weights 136 KiB + 16 KiB activation double buffer = 152 KiB of LDS; chunks of 16 with the
    // last weight rows left in global memory measured 13 % slower); stage 1 (32 KiB of weights): chunks of 16.
    constexpr int K = STAGE == 0 ? 272 : 64, KL = K, BK = STAGE == 0 ? 8 : 16, NCH = K / BK, TILE = 256, LD = TILE + 4;
    constexpr int NF = BK / 8;           // float4 per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wt = lds;                                              // [KL][128]
    float(*As)[BK][LD] = (float(*)[BK][LD])(lds + KL * 128);      // [2][BK][LD]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64, lr = lane & 31, lk = lane >> 5;
    for (int i = tid; i < KL * 128 / 4; i += 512) ((float4*)Wt)[i] = ((const float4*)a.wt)[i];

    const int ntiles = a.B * a.tiles_t;
    const int r0 = tid >> 1, kq = (tid & 1) * (BK / 2);
    // The activation chunks are requested TWO chunks ahead of the MFMAs that consume them (register ring P0 / P1): with
    // one persistent workgroup per CU nothing else hides the HBM latency (one chunk = 1024 / 2048 MFMA cycles per wave).
    float4 P0[2], P1[2];
    auto load_chunk = [&](float4* dst, int tile, int kc) {
        const int b = tile / a.tiles_t, t0 = (tile - b * a.tiles_t) * TILE;
        const int k = kc * BK + kq;
        const int t = t0 + r0;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (tile < ntiles && t < a.Tw) {
            const float* src = nullptr;
            if (STAGE == 0) {
                if (k < 192) {
                    const int ts = t + ((k >> 6) - 1) * a.dil;
                    if (ts >= 0 && ts < a.Tw) src = a.x_in + ((int64_t)b * a.Tw + ts) * 64 + (k & 63);
                } else {
                    src = a.cu + ((int64_t)b * a.Tw + t) * 80 + (k - 192);
                }
            } else {
                src = a.gin + ((int64_t)b * a.Tw + t) * 64 + k;
            }
            if (src) {
                v0 = *(const float4*)src;
                if (NF == 2) v1 = *(const float4*)(src + 4);
            }
        }
        dst[0] = v0, dst[1] = v1;
    };
    auto store_chunk = [&](const float4* src, int buf) {
        As[buf][kq + 0][r0] = src[0].x, As[buf][kq + 1][r0] = src[0].y, As[buf][kq + 2][r0] = src[0].z, As[buf][kq + 3][r0] = src[0].w;
        if (NF == 2)
            As[buf][kq + 4][r0] = src[1].x, As[buf][kq + 5][r0] = src[1].y, As[buf][kq + 6][r0] = src[1].z, As[buf][kq + 7][r0] = src[1].w;
    };
    auto advance = [&](int& tile, int& kc) {
        if (++kc == NCH) kc = 0, tile += gridDim.x;
    };

    f32x16 acc[2][2];
    auto clear = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    clear();
    const float bj0 = a.bias[wn + lr], bj1 = a.bias[wn + 32 + lr];
    // stage 1: the values the epilogue updates (x for the residual waves, skips for the skip waves) are requested at the
    // START of the tile and consumed four chunks later
    float old[STAGE == 1 ? 2 : 1][2][16];
    const float* oldsrc = (wn == 0) ? a.x_in : (const float*)a.skips;

    int tile = blockIdx.x, kc = 0;
    if (tile >= ntiles) return;
    load_chunk(P0, tile, 0);
    store_chunk(P0, 0);
    __syncthreads();             // (also: Wt is complete)
    int t1 = tile, k1 = 0;
    advance(t1, k1);             // (t1, k1) = chunk s+1
    load_chunk(P0, t1, k1);
    int buf = 0;
    while (true) {
        int t2 = t1, k2 = k1;
        advance(t2, k2);         // chunk s+2
        if (t1 < ntiles) load_chunk(P1, t2, k2);
        if (STAGE == 1 && kc == 0) {
            const int b = tile / a.tiles_t, t0 = (tile - b * a.tiles_t) * TILE;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = t0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const int64_t i0 = ((int64_t)b * a.Tw + (t < a.Tw ? t : 0)) * 64 + lr;
                    old[i][0][r] = oldsrc[i0];
                    old[i][1][r] = oldsrc[i0 + 32];
                }
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a0 = As[buf][kk * 2 + lk][wm + lr], a1 = As[buf][kk * 2 + lk][wm + 32 + lr];
            const int kg = kc * BK + kk * 2 + lk;
            float b0, b1;
            if (K > KL && kc * BK >= KL) {      // (uniform) weight rows that did not fit the LDS: L1 / L2 hits
                const float* wr = a.wt + kg * 128 + wn + lr;
                b0 = wr[0], b1 = wr[32];
            } else {
                const float* wr = Wt + kg * 128 + wn + lr;
                b0 = wr[0], b1 = wr[32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kc == NCH - 1) {     // tile finished: gate / residual / skip straight from the accumulators
            const int b = tile / a.tiles_t, t0 = (tile - b * a.tiles_t) * TILE;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = t0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (t >= a.Tw) continue;
                    const int64_t row = (int64_t)b * a.Tw + t;
                    if (STAGE == 0) {
                        const float ya = acc[i][0][r] + bj0, yb = acc[i][1][r] + bj1;
                        // tanh(y) = 1 - 2 / (1 + e^{2y}): two v_exp_f32 + two v_rcp_f32 per output (abs error ~1e-7)
                        const float th = 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * ya));
                        a.g[row * 64 + (wn >> 1) + lr] = th * __frcp_rn(1.f + __expf(-yb));
                    } else {
                        const float o0 = acc[i][0][r] + bj0, o1 = acc[i][1][r] + bj1;
                        const int64_t i0 = row * 64 + lr;
                        if (wn == 0) {   // residual half: channels lr and 32 + lr
                            a.x_out[i0] = (o0 + old[i][0][r]) * 0.70710678118654752440f;
                            a.x_out[i0 + 32] = (o1 + old[i][1][r]) * 0.70710678118654752440f;
                        } else {         // skip half
                            a.skips[i0] = old[i][0][r] + o0;
                            a.skips[i0 + 32] = old[i][1][r] + o1;
                        }
                    }
                }
            clear();
        }
        if (t1 >= ntiles) break;
        store_chunk(P0, buf ^ 1);
        __syncthreads();
        buf ^= 1;
        P0[0] = P1[0], P0[1] = P1[1];
        tile = t1, kc = k1;
        t1 = t2, k1 = k2;
    }
}

static int pwg_blocks() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}

// One residual block, in place on x and skips.  wt0: [272][128] (k = tap*64 + ch | 192 + aux ch; column n' = permuted
// gate channel: see a3t_amd/vocoder.py), b0: [128] permuted the same way; wt1: [64][128] = conv1x1_out.weight^T, b1: [128].
// g: scratch [B*Tw][64].
extern "C" int a3t_pwg_block(float* x, const float* cu, const float* wt0, const float* b0, const float* wt1,
                             const float* b1, float* g, float* skips, int B, int Tw, int dil, void* stream) {
    if (B <= 0 || Tw <= 0 || dil <= 0) return A3T_EINVAL;
    PwgArgs a;
    a.x_in = x, a.cu = cu, a.gin = g, a.g = g, a.x_out = x, a.skips = skips;
    a.B = B, a.Tw = Tw, a.dil = dil, a.tiles_t = (Tw + 255) / 256;
    const int ntiles = B * a.tiles_t;
    constexpr int lds0 = (272 * 128 + 2 * 8 * 260) * 4, lds1 = (64 * 128 + 2 * 16 * 260) * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)pwg_stage_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds0);
        (void)hipFuncSetAttribute((const void*)pwg_stage_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
        attr = true;
    }
    const int cus = pwg_blocks();
    a.wt = wt0, a.bias = b0;
    int g0 = ntiles < cus ? ntiles : cus;                 // stage 0: 153 KiB of LDS -> one persistent workgroup per CU
    hipLaunchKernelGGL(pwg_stage_kernel<0>, dim3(g0), dim3(512), lds0, (hipStream_t)stream, a);
    a.wt = wt1, a.bias = b1;
    int g1 = ntiles < cus ? ntiles : cus;                 // stage 1: ~200 VGPRs x 8 waves -> one per CU as well
    hipLaunchKernelGGL(pwg_stage_kernel<1>, dim3(g1), dim3(512), lds1, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
