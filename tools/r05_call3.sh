#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/probes/atomic_epilogue > gpurun_out/r05_c3_atomic_probe.txt 2>&1
( timeout 1500 python -m pytest tests/test_gpu_fullsize_oracle.py "tests/test_gpu_parity_r2.py::test_bf16_training_step_through_the_fused_attention_forward_against_reference_goldens" -q -s 2>&1 | grep -v Warning | grep "^\[\|passed\|failed\|Error\|assert" ) > gpurun_out/r05_c3_newtests.txt
