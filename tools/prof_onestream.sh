# one-stream kernel stats of the C2 bf16 step (every kernel alone on the GPU): per-kernel totals per step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_one
A3T_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_one -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/prof_one.log 2>&1
cd $R
find gpurun_out/prof_one -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_one/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 7
tot = sum(float(r['TotalDurationNs']) for r in rows)
gemm = sum(float(r['TotalDurationNs']) for r in rows if 'gemm' in r['Name'])
print(f"total {tot/1e6/steps:.2f} ms/step, gemm {gemm/1e6/steps:.2f}, non-gemm {(tot-gemm)/1e6/steps:.2f}")
for r in rows[:45]:
    print(f"{r['Name'][:90]:90s} calls/step={int(r['Calls'])/steps:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
