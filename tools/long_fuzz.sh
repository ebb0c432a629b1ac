# longer randomised sweeps than the -m gpu tests run (minutes, not seconds)
for s in 11 12 13; do echo "== gemm seed $s"; timeout 600 python tests/fuzz_gemm.py $s 400 2>&1 | grep -v amdgpu | grep -E "cases|FAIL|EXC" | head -8; done
for s in 21 22; do echo "== row kernels seed $s"; timeout 600 python tests/fuzz_rowkernels.py $s 150 2>&1 | grep -v amdgpu | tail -3; done
for s in 31 32; do echo "== engine seed $s"; timeout 900 python tests/fuzz_engine.py $s 25 2>&1 | grep -v amdgpu | tail -4; done
