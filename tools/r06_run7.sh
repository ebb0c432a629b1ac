mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06_tests_7.log 2>&1; echo "tests rc $?" >> gpurun_out/r06_tests_7.log
tail -4 gpurun_out/r06_tests_7.log
grep "^\[c" gpurun_out/r06_tests_7.log | head -20
python bench.py > gpurun_out/r06_bench_7.json 2> gpurun_out/r06_bench_7.err; tail -c 300 gpurun_out/r06_bench_7.json; tail -3 gpurun_out/r06_bench_7.err
