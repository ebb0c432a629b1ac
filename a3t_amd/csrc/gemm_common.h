// gemm_common.h -- descriptor resolved for the device + shared epilogue of the GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"
#include "dtype_io.h"

// Introspection for bench.py / profilers: a3t_gemm records the name (as rocprofv3 prints it) of the kernel variant
// its dispatcher picked for the calling thread's last launch; read back with a3t_gemm_last_kernel().
void a3t_note_kernel(const char* fmt, ...);

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
    int ntiles;   // output tiles per (batch element, K split): 1-D grids of the direct-to-LDS kernels
    float* colsum;
    int64_t colsum_bs1;
    float colsum_scale;
    int colsum_slots, colsum_ss;   // >1: atomics spread over `slots` accumulator copies, `ss` floats apart
    unsigned int drop_key, drop_thr;
    float drop_inv;   // 1/(1-p), 0 when dropout is off
    // 8-phase kernel (gemm_bf16_8p.hip)
    unsigned char* keep_out;        // optional: one keep bit per output (value > 0), tile-major image
    const unsigned char* keep_in;   // optional: keep bits applied as a mask (written by the GEMM with the same M x N tiling)
    unsigned int a_bytes, b_bytes;  // operand extents for the buffer descriptors
    float* slab;                    // token-reduction variant: split-K partial tiles (plain stores) for the deterministic fold
    int sole_writer;                // A3T_ACC_SOLE: the fold may add with plain read-modify-writes
    const void* A2;                 // second product of the same launch (streaming kernel, [m][k] operand): C = alpha (A B + A2 B2)
    const void* B2;
    int64_t b2_cs, b2_bs0, b2_bs1;
    float* colsum2;
    int64_t a2_rs;                  // row stride of A2 (0: A's)
    int a_unaligned;                // bit 0: A, bit 1: A2 is a 2-byte aligned strided view (a3t_gemm_desc::a_unaligned)
    int keep_layout;                // 1: keep_out / keep_in are the row-major nibble image (a3t_gemm_desc::keep_layout)
    int a_signmask;                 // A elements with the sign bit set are read as zero (m-contiguous bf16 A: gemm_bf16_tt.hip, gemm_bf16.hip L_TN)
};

__device__ __forceinline__ unsigned short f2bf(float f) { return io_f2bf(f); }   // hardware RNE conversion

// Sign-tagged bf16 operands (a3t_gemm_desc::a_signmask): a fragment of eight bf16 read as packed signed 16-bit integers -- an
// element with the sign bit set is negative there, every other one is not -- and v_pk_max_i16 against `fl` = 0 turns the tagged
// elements into +0 (fl = -32768: the identity).  Four VALU instructions per fragment.
template <typename V>
__device__ __forceinline__ V a3t_sign_floor(V v, short fl) {
    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    const s16x8_ f = {fl, fl, fl, fl, fl, fl, fl, fl};
    return __builtin_bit_cast(V, __builtin_elementwise_max(__builtin_bit_cast(s16x8_, v), f));
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
    if (p.c_dtype == A3T_BF16) {     // (bf16 C: STORE or ADD -- one writer per element, read-modify-write in fp32)
        unsigned short* c = (unsigned short*)p.C + idx;
        *c = f2bf(p.accumulate == A3T_ACC_STORE ? v : v + bf2f(*c));
        return;
    }
    float* C = (float*)p.C;
    if (p.accumulate == A3T_ACC_STORE)
        C[idx] = v;
    else if (p.accumulate == A3T_ACC_ADD)
        C[idx] += v;
    else
#ifdef A3T_EXPERIMENT_NOATOMIC      // measurement build only (wrong sums): what do the split-K atomics cost?
        C[idx] = v;
#else
        atomicAdd(&C[idx], v);
#endif
}


// Vector epilogue for 4 consecutive output columns (contract checked on the host: GP::epi_vec):
// bias -> activation -> ReLU'(S) mask -> dropout -> alpha -> residual -> store; `cs` accumulates the
// stored values for the fused column sums.  Never used with A3T_ACC_ATOMIC.
__device__ __forceinline__ void epilogue_vec4(const GP& p, float4 v, int64_t idx, const float4& bias4, int ks, float4& cs) {
    v.x += bias4.x, v.y += bias4.y, v.z += bias4.z, v.w += bias4.w;
    if (p.act != A3T_ACT_NONE) {
        v.x = apply_act(v.x, p.act), v.y = apply_act(v.y, p.act);
        v.z = apply_act(v.z, p.act), v.w = apply_act(v.w, p.act);
    }
    if (p.S) {
        float4 sv;
        if (p.s_dtype == A3T_BF16) {
            uint2 t = *(const uint2*)((const unsigned short*)p.S + idx);
            sv = make_float4(bf2f(t.x & 0xffff), bf2f(t.x >> 16), bf2f(t.y & 0xffff), bf2f(t.y >> 16));
        } else {
            sv = *(const float4*)(p.S + idx);
        }
        v.x = sv.x > 0.f ? v.x : 0.f, v.y = sv.y > 0.f ? v.y : 0.f;
        v.z = sv.z > 0.f ? v.z : 0.f, v.w = sv.w > 0.f ? v.w : 0.f;
    }
    if (p.keep_layout == 1 && p.keep_in) {      // row-major nibble image in the place of S (host contract: c_rs == N, no batch)
        const unsigned kb = p.keep_in[idx >> 2];
        v.x = (kb & 1u) ? v.x : 0.f, v.y = (kb & 2u) ? v.y : 0.f, v.z = (kb & 4u) ? v.z : 0.f, v.w = (kb & 8u) ? v.w : 0.f;
    }
    if (p.drop_inv > 0.f) {
        const unsigned int i0 = (unsigned int)idx;
        bool kp[4];
        rng_keep4(p.drop_key, i0, p.drop_thr, kp);       // (idx % 4 == 0: vector epilogue contract)
        v.x = kp[0] ? v.x * p.drop_inv : 0.f, v.y = kp[1] ? v.y * p.drop_inv : 0.f;
        v.z = kp[2] ? v.z * p.drop_inv : 0.f, v.w = kp[3] ? v.w * p.drop_inv : 0.f;
    }
    v.x *= p.alpha, v.y *= p.alpha, v.z *= p.alpha, v.w *= p.alpha;
    if (p.keep_layout == 1 && p.keep_out)       // (value > 0 after activation / dropout: the mask its data gradient needs)
        p.keep_out[idx >> 2] = (unsigned char)((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u));
    if (p.R && ks == 0) {
        float4 rv = *(const float4*)(p.R + idx);
        v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
    }
    cs.x += v.x, cs.y += v.y, cs.z += v.z, cs.w += v.w;
    if (p.c_dtype == A3T_BF16) {
        uint2* c = (uint2*)((unsigned short*)p.C + idx);
        if (p.accumulate != A3T_ACC_STORE) {     // A3T_ACC_ADD on a bf16 tensor: the sum is formed in fp32 and rounded once
            const uint2 t = *c;                  // (`cs`, the fused column sum, has taken the increment alone above)
            v.x += bf2f(t.x & 0xffff), v.y += bf2f(t.x >> 16), v.z += bf2f(t.y & 0xffff), v.w += bf2f(t.y >> 16);
        }
        uint2 o;
        o.x = io_pack2(v.x, v.y);
        o.y = io_pack2(v.z, v.w);
        *c = o;
    } else {
        float* C = (float*)p.C + idx;
        if (p.accumulate == A3T_ACC_STORE) {
            *(float4*)C = v;
        } else {   // A3T_ACC_ADD
            float4 o = *(const float4*)C;
            o.x += v.x, o.y += v.y, o.z += v.z, o.w += v.w;
            *(float4*)C = o;
        }
    }
}
// lanes l, l+LR, l+2*LR, ... of a wave hold the same 4 columns (different rows): reduce, then one atomic per column
template <int LR = 16>
__device__ __forceinline__ void colsum_flush(const GP& p, float4 cs, int lane, bool col_ok, int z1, int col, int spread = 0) {
#pragma unroll
    for (int o = LR; o < 64; o <<= 1) {
        cs.x += __shfl_xor(cs.x, o, 64), cs.y += __shfl_xor(cs.y, o, 64);
        cs.z += __shfl_xor(cs.z, o, 64), cs.w += __shfl_xor(cs.w, o, 64);
    }
    if (lane < LR && col_ok) {
        float* o = p.colsum + z1 * p.colsum_bs1 + col;
        if (p.colsum_slots > 1) o += (int64_t)(spread % p.colsum_slots) * p.colsum_ss;
        atomicAdd(o + 0, p.colsum_scale * cs.x), atomicAdd(o + 1, p.colsum_scale * cs.y);
        atomicAdd(o + 2, p.colsum_scale * cs.z), atomicAdd(o + 3, p.colsum_scale * cs.w);
    }
}
