mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06_tests_1.log 2>&1; echo "tests rc $?" >> gpurun_out/r06_tests_1.log
tail -3 gpurun_out/r06_tests_1.log
python bench.py > gpurun_out/r06_bench_1.json 2> gpurun_out/r06_bench_1.err; tail -c 600 gpurun_out/r06_bench_1.json
python tools/postnet_floor.py c4 > gpurun_out/r06_postnet_floor.txt 2>&1
python tools/postnet_floor.py c2 >> gpurun_out/r06_postnet_floor.txt 2>&1
cat gpurun_out/r06_postnet_floor.txt | grep "^\["
