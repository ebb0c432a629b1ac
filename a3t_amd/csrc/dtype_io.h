// dtype_io.h -- runtime-typed element access (fp32 | bf16 storage) for the HBM-bound kernels.
// The dtype flag is wave-uniform, so the branch costs one scalar compare per access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"

// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction); the manual integer sequence it replaces cost ~5 VALU ops per value in every bf16-storing epilogue.
typedef __bf16 io_bf16x2 __attribute__((ext_vector_type(2)));
typedef float io_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short io_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned int io_pack2(float lo, float hi) {
    const io_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, io_bf16x2));
}
__device__ __forceinline__ float io_bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
// Counter-based dropout RNG: keep(key, idx) is a pure function, so forward and backward kernels
// (and GEMM epilogues) regenerate the same mask from (key, element index) with no stored state.
__device__ __forceinline__ unsigned int rng_fmix32(unsigned int h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
// One 32-bit hash serves an (even, odd) element pair, 16 random bits each (drop probability resolution 2^-16):
// the vector kernels, which own 4-8 consecutive elements, pay half the integer work per element.
// Round 3: the PCG output permutation (RXS-M-XS, the "pcg_hash" of Jarzynski & Olano, Hash Functions for GPU Rendering) on
// the keyed counter -- 2 quarter-rate 32-bit multiplies per pair instead of the 5 of two murmur finalisers (the dropout
// epilogue of a 256x256 GEMM tile is 64 hashes per lane with nothing to hide them behind).  The key is itself a mixed
// 32-bit hash of (step seed, site) computed on the host.
__device__ __forceinline__ unsigned int rng_pair(unsigned int key, unsigned int pair) {
    // the key enters by XOR and again as the (odd) increment: with `pair + key` two sites' masks were index-shifted copies of
    // each other; same instruction count
    const unsigned int state = (pair ^ key) * 747796405u + (key | 1u);
    const unsigned int word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ bool rng_keep(unsigned int key, unsigned int idx, unsigned int thr) {
    const unsigned int h = rng_pair(key, idx >> 1);
    return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= (thr >> 16);
}
// keep flags of elements i0 .. i0+3, i0 % 2 == 0
__device__ __forceinline__ void rng_keep4(unsigned int key, unsigned int i0, unsigned int thr, bool* k) {
    const unsigned int h0 = rng_pair(key, i0 >> 1), h1 = rng_pair(key, (i0 >> 1) + 1), t = thr >> 16;
    k[0] = (h0 & 0xffffu) >= t, k[1] = (h0 >> 16) >= t, k[2] = (h1 & 0xffffu) >= t, k[3] = (h1 >> 16) >= t;
}
__device__ __forceinline__ unsigned int rng_thr(float p) { return (unsigned int)((double)p * 4294967296.0); }

__device__ __forceinline__ float ldx(const void* p, int dt, int64_t i) {
    return dt == A3T_BF16 ? io_bf2f(((const unsigned short*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void stx(void* p, int dt, int64_t i, float v) {
    if (dt == A3T_BF16)
        ((unsigned short*)p)[i] = io_f2bf(v);
    else
        ((float*)p)[i] = v;
}
