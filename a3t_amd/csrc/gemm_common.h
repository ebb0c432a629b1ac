// gemm_common.h -- descriptor resolved for the device + shared epilogue of the GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"
#include "dtype_io.h"

struct GP {
    const void* A;
    const void* B;
    void* C;
    const float* bias;
    const float* R;
    const float* S;
    int M, N, K, Kc;
    int64_t a_rs, a_cs, b_rs, b_cs, b_ts, c_rs;
    int batch_inner;
    int64_t a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    int taps, pad, dil, Tseq, kshift, kshift_mode;
    float alpha;
    int act, accumulate, splitk, c_dtype, tiles_n, s_dtype, epi_vec;
    float* colsum;
    int64_t colsum_bs1;
    float colsum_scale;
    unsigned int drop_key, drop_thr;
    float drop_inv;   // 1/(1-p), 0 when dropout is off
};

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (NaN payloads are not preserved)
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) {
    return *p;
}
template <>
__device__ __forceinline__ float ldf<unsigned short>(const unsigned short* p) {
    return bf2f(*p);
}

// ---- address helpers -------------------------------------------------------------------------
// A operand, k-contiguous (NT / conv): element (m, kg); tpos = m % Tseq (precomputed)
template <typename T>
__device__ __forceinline__ const T* a_ptr_k(const GP& p, const T* A, int m, int tpos, int kg) {
    if (p.taps > 1) {
        int tap = kg / p.Kc;
        int c = kg - tap * p.Kc;
        int off = (tap - p.pad) * p.dil;
        int tt = tpos + off;
        if (tt < 0 || tt >= p.Tseq) return nullptr;
        return A + (int64_t)(m + off) * p.a_rs + c;
    }
    return A + (int64_t)m * p.a_rs + kg;
}
// B operand k offset for a given global k; returns false when the whole k row is zero
__device__ __forceinline__ bool b_koff(const GP& p, int k, int64_t& koff) {
    if (p.taps > 1) {
        int tap = k / p.Kc;
        int c = k - tap * p.Kc;
        koff = (int64_t)tap * p.b_ts + (int64_t)c * p.b_cs;
        return true;
    }
    if (p.kshift_mode) {
        int tt = (k % p.Tseq) + p.kshift;
        koff = (int64_t)(k + p.kshift) * p.b_cs;
        return tt >= 0 && tt < p.Tseq;
    }
    koff = (int64_t)k * p.b_cs;
    return true;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == A3T_ACT_RELU) return fmaxf(v, 0.f);
    if (act == A3T_ACT_TANH) return tanhf(v);
    if (act == A3T_ACT_SWISH) return v / (1.f + __expf(-v));
    return v;
}

__device__ __forceinline__ void epilogue_store(const GP& p, int64_t zoff, int row, int col, float v, int ks) {
    if (row >= p.M || col >= p.N) return;
    int64_t idx = zoff + (int64_t)row * p.c_rs + col;
    if (p.bias && ks == 0) v += p.bias[col];
    v = apply_act(v, p.act);
    if (p.S) {
        float sv = (p.s_dtype == A3T_BF16) ? bf2f(((const unsigned short*)p.S)[idx]) : p.S[idx];
        v = (sv > 0.f) ? v : 0.f;
    }
    if (p.drop_inv > 0.f) v = rng_keep(p.drop_key, (unsigned int)idx, p.drop_thr) ? v * p.drop_inv : 0.f;
    v *= p.alpha;
    if (p.R && ks == 0) v += p.R[idx];
    if (p.c_dtype == A3T_BF16) {
        ((unsigned short*)p.C)[idx] = f2bf(v);
        return;
    }
    float* C = (float*)p.C;
    if (p.accumulate == A3T_ACC_STORE)
        C[idx] = v;
    else if (p.accumulate == A3T_ACC_ADD)
        C[idx] += v;
    else
        atomicAdd(&C[idx], v);
}

