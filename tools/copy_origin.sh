# which HIP API calls produce the __amd_rocclr_copyBuffer dispatches of a step?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_hip
A3T_SIDE_STREAM=0 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $R/gpurun_out/prof_hip -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/prof_hip.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
kt = glob.glob('gpurun_out/prof_hip/**/*kernel_trace.csv', recursive=True)[0]
ha = glob.glob('gpurun_out/prof_hip/**/*hip_api_trace.csv', recursive=True)[0]
api = {}
for r in csv.DictReader(open(ha)):
    api[r['Correlation_Id']] = r['Function']
cnt = collections.Counter()
rows = list(csv.DictReader(open(kt)))
prev = None
ctx = collections.Counter()
for i, r in enumerate(rows):
    if 'copyBuffer' in r['Kernel_Name']:
        cnt[api.get(r['Correlation_Id'], '?')] += 1
        nxt = rows[i + 1]['Kernel_Name'][:60] if i + 1 < len(rows) else ''
        prv = rows[i - 1]['Kernel_Name'][:60] if i > 0 else ''
        ctx[(prv, nxt)] += 1
print(cnt)
for k, v in ctx.most_common(12):
    print(v, k)
PY
find gpurun_out/prof_hip -name "*.csv" -size +1M -delete
