// dtype_io.h -- runtime-typed element access (fp32 | bf16 storage) for the HBM-bound kernels.
// The dtype flag is wave-uniform, so the branch costs one scalar compare per access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3t_hip.h"

__device__ __forceinline__ unsigned short io_f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float io_bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
__device__ __forceinline__ float ldx(const void* p, int dt, int64_t i) {
    return dt == A3T_BF16 ? io_bf2f(((const unsigned short*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void stx(void* p, int dt, int64_t i, float v) {
    if (dt == A3T_BF16)
        ((unsigned short*)p)[i] = io_f2bf(v);
    else
        ((float*)p)[i] = v;
}
