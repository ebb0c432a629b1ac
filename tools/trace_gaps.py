"""GPU idle-gap analysis of a rocprofv3 kernel trace (run on the GPU box, right after tools/step_prof.sh-style tracing):
union busy time, per-queue busy time, idle gaps between kernels, over the second half of the trace (steady state)."""
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows))
t_mid = ev[len(ev) // 2][0]
ev = [e for e in ev if e[0] >= t_mid]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, q, n in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
perq = {}
for s, e, q, n in ev:
    perq[q] = perq.get(q, 0) + (e - s)
big = sorted(gaps, reverse=True)[:12]
out = dict(window_ms=(t1 - t0) / 1e6, union_busy_ms=busy / 1e6, idle_ms=(t1 - t0 - busy) / 1e6, kernels=len(ev),
           per_queue_busy_ms={k: v / 1e6 for k, v in perq.items()},
           gaps_over_5us=sum(1 for g, _ in gaps if g > 5000), gaps_total_ms=sum(g for g, _ in gaps) / 1e6,
           median_gap_us=sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3,
           biggest_gaps_us=[(g / 1e3, n[:60]) for g, n in big])
print(json.dumps(out, indent=1))
