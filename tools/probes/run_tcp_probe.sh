cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1 2 3; do $R/tools/probes/tcp_dma_probe $m; done
for m in 0 1 2 3; do
  rm -rf /tmp/pp_$m
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pp_$m -- $R/tools/probes/tcp_dma_probe $m > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/pp_$m/*/*counter_collection.csv'):
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'probe' in r['Kernel_Name']:
            a = agg.setdefault(r['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
    print('mode $m', {k: round(v / n / 1e6, 2) for k, (n, v) in agg.items()})
PY
done
