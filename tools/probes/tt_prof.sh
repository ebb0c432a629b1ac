cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ttp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ttp -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vocoder --no-collate --no-c4 --no-kernel-profile --no-live-traffic > /tmp/ttp.log 2>&1
f=$(ls /tmp/ttp/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']}")
PY
