# quick kernel-stats profile of the training step, two streams and one stream -> gpurun_out/q_{two,one}.csv (top rows printed)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/q_prof_s$mode
  A3T_SIDE_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_prof_s$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile --no-c4 > $R/gpurun_out/q_prof_s$mode.log 2>&1
  find $R/gpurun_out/q_prof_s$mode -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/q_prof_s$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/q_$([ $mode = 1 ] && echo two || echo one).csv
  rm -rf $R/gpurun_out/q_prof_s$mode
done
cd $R
python - <<'PY'
import csv
for f in ("gpurun_out/q_two.csv", "gpurun_out/q_one.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f, "total kernel ms per step", tot / 7 / 1e6)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
        print(f'{r["Name"][:70]:70s} calls/step {int(r["Calls"])/7:6.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  total/step {float(r["TotalDurationNs"])/7/1e6:6.2f} ms')
PY
