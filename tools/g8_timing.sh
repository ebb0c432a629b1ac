#!/bin/bash
# Per-tile time stamps of the 8-phase GEMM: builds an instrumented copy of the library (-DG8_TIMING) next to the product one.
set -e
cd "$(dirname "$0")/.."
L=a3t_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DG8_TIMING -c a3t_amd/csrc/gemm_bf16_8p.hip -o $L/gemm_bf16_8p_timing.o
OBJS=$(ls $L/*.o | grep -v "gemm_bf16_8p" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/liba3t_hip_timing.so $OBJS $L/gemm_bf16_8p_timing.o
echo built $L/liba3t_hip_timing.so
