import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
print('total ms', tot / 1e6)
