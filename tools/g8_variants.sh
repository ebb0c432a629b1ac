#!/bin/bash
# timing-only variants of the 8-phase GEMM (experiments): builds liba3t_hip_<name>.so for each -D flag set
set -e
cd "$(dirname "$0")/.."
L=a3t_amd/lib
OBJS=$(ls $L/*.o | grep -v "gemm_bf16_8p" )
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $2 -c a3t_amd/csrc/gemm_bf16_8p.hip -o $L/gemm_bf16_8p_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/liba3t_hip_$1.so $OBJS $L/gemm_bf16_8p_$1.o
}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  build $name "$flags" &
done
wait
ls -la $L/*.so
