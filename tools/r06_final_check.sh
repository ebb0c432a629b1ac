# final check of a round on the GPU box: the full GPU suite, the e2e tests under the one-stream schedule, smoke(), the default bench line
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
A3T_SIDE_STREAM=0 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke"
python bench.py 2> gpurun_out/bench_final.log | tee gpurun_out/bench_final.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], r['kernel'], round(r['frac'],3), r['traffic'], d['c4']['ms_per_step'])"
