mkdir -p gpurun_out
t0=$(date +%s); python bench.py > gpurun_out/r06_bench_live.json 2> gpurun_out/r06_bench_live.log; t1=$(date +%s); echo "bench.py wall: $((t1-t0)) s"
grep -E "bench " gpurun_out/r06_bench_live.log | tail -12
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_live.json').read().strip().split('\n')[-1])
r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['traffic'], r['traffic_tracked_file']); print(r['traffic_source'])
PY
