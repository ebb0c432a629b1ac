// features.hip -- on-device log-mel front end (espnet2/layers/stft.py:56-124, log_mel.py:56-83,
// tts/feats_extract/log_mel_fbank.py:88-106): reflect padding, |STFT| from the DFT-as-GEMM output,
// log10 + length masking.  The two contractions (frames x windowed DFT basis, |S| x mel matrix) run
// on the shared MFMA GEMM with an overlapping-row A operand (row stride = hop).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"

#define GS(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)
static inline int nb_(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// out[b][i] = x[b][reflect(i - pad)], i in [0, N + 2*pad); row stride of out = ld (>= N+2*pad, tail zeroed)
__global__ void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int N, int pad, int ld) {
    const int64_t n = (int64_t)B * ld;
    GS(i, n) {
        int b = (int)(i / ld), j = (int)(i - (int64_t)b * ld);
        float v = 0.f;
        if (j < N + 2 * pad) {
            int s = j - pad;
            if (s < 0) s = -s;
            if (s >= N) s = 2 * (N - 1) - s;
            v = x[(int64_t)b * N + s];
        }
        out[i] = v;
    }
}
extern "C" int a3t_reflect_pad(const float* x, float* out, int B, int N, int pad, int ld, void* stream) {
    if (pad >= N || ld < N + 2 * pad) return A3T_EINVAL;
    int64_t n = (int64_t)B * ld;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(nb_(n)), dim3(256), 0, (hipStream_t)stream, x, out, B, N, pad, ld);
    return (int)hipGetLastError();
}

// S[row][0:nb] = re, S[row][nb:2nb] = im  ->  amp[row][f] = sqrt(max(re^2+im^2, 1e-10)); amp row stride ld
// (columns nb..ld-1 zero so the mel GEMM can run on aligned K)
__global__ void stft_amp_kernel(const float* __restrict__ S, float* __restrict__ amp, int64_t rows, int nb, int ld) {
    const int64_t n = rows * ld;
    GS(i, n) {
        int64_t r = i / ld;
        int f = (int)(i - r * ld);
        float v = 0.f;
        if (f < nb) {
            float re = S[r * 2 * nb + f], im = S[r * 2 * nb + nb + f];
            v = sqrtf(fmaxf(re * re + im * im, 1.0e-10f));
        }
        amp[i] = v;
    }
}
extern "C" int a3t_stft_amp(const float* S, float* amp, int64_t rows, int nbins, int ld, void* stream) {
    int64_t n = rows * ld;
    hipLaunchKernelGGL(stft_amp_kernel, dim3(nb_(n)), dim3(256), 0, (hipStream_t)stream, S, amp, rows, nbins, ld);
    return (int)hipGetLastError();
}

// mel[b][f][c] <- log10(max(mel, 1e-10)), frames f >= olens[b] -> 0.0
__global__ void logmel_finish_kernel(float* __restrict__ mel, const int64_t* __restrict__ olens, int B, int F, int C) {
    const int64_t n = (int64_t)B * F * C;
    GS(i, n) {
        int64_t r = i / C;
        int b = (int)(r / F), f = (int)(r - (int64_t)b * F);
        mel[i] = (f < olens[b]) ? log10f(fmaxf(mel[i], 1e-10f)) : 0.f;
    }
}
extern "C" int a3t_logmel_finish(float* mel, const int64_t* olens, int B, int F, int C, void* stream) {
    int64_t n = (int64_t)B * F * C;
    hipLaunchKernelGGL(logmel_finish_kernel, dim3(nb_(n)), dim3(256), 0, (hipStream_t)stream, mel, olens, B, F, C);
    return (int)hipGetLastError();
}
