#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r05_c6_pytest.txt
python bench.py > gpurun_out/r05_c6_bench.json 2> gpurun_out/r05_c6_bench.log
