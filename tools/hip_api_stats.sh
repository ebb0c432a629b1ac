# which HIP runtime calls does a training step make (and which of them turn into blit kernels)?
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hipapi
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/hipapi -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile > /tmp/hipapi.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/hipapi/**/*hip_api_stats.csv', recursive=True) + glob.glob('/tmp/hipapi/**/*hip_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r['Name'][:50], r['Calls'], r.get('AverageNs'))
PY
ls /tmp/hipapi/*/ | head
