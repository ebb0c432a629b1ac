"""Timeline of ONE steady-state training step from a rocprofv3 kernel trace (tools/trace_step.sh):
per queue busy time, main-queue gaps, what runs beside what.  usage: python tools/trace_analyse.py gpurun_out/trace_step.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# steps are delimited by the optimizer kernel
opt = [i for i, r in enumerate(rows) if "clip_adam_noam" in r["Kernel_Name"]]
print("optimizer launches:", len(opt))
a, b = opt[-2] + 1, opt[-1] + 1
step = rows[a:b]
t0, t1 = rows[opt[-2]]["e"], rows[opt[-1]]["e"]
print(f"last step: {len(step)} launches, {(t1 - t0) / 1e6:.3f} ms end-of-optimizer to end-of-optimizer")
byq = defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: {len(rs)} launches, busy {busy / 1e6:.3f} ms, first {(rs[0]['s'] - t0) / 1e6:.3f} last {(rs[-1]['e'] - t0) / 1e6:.3f}")
mainq = max(byq, key=lambda q: len(byq[q]))
rs = byq[mainq]
gaps = []
for x, y in zip(rs, rs[1:]):
    g = y["s"] - x["e"]
    if g > 0:
        gaps.append((g, x["Kernel_Name"][:50], y["Kernel_Name"][:50], (x["e"] - t0) / 1e6))
print(f"main queue {mainq}: idle between launches {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; >5 us: "
      f"{sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.3f} ms in {sum(1 for g in gaps if g[0] > 5000)}")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"   {g[0] / 1e3:8.1f} us at {g[3]:7.3f} ms  after {g[1]}  before {g[2]}")
# time on the main queue by kernel name, alone vs while another queue is busy
others = sorted((r["s"], r["e"]) for q, v in byq.items() if q != mainq for r in v)
def overlap(s, e):
    o = 0
    for xs, xe in others:
        if xe <= s:
            continue
        if xs >= e:
            break
        o += min(e, xe) - max(s, xs)
    return o
agg = defaultdict(lambda: [0, 0, 0])
for r in rs:
    d = r["e"] - r["s"]
    k = agg[r["Kernel_Name"][:70]]
    k[0] += 1
    k[1] += d
    k[2] += overlap(r["s"], r["e"])
print("main-queue kernels (calls, total ms, share of it with a side kernel running):")
for n, (c, d, o) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"   {c:4d} {d / 1e6:7.3f} ms  {100.0 * o / max(d, 1):5.1f} %  {n}")
tot_main = sum(v[1] for v in agg.values())
print(f"main queue busy {tot_main / 1e6:.3f} ms")
# forward / backward boundary: the loss kernel
for r in rs:
    if "loss_grad" in r["Kernel_Name"]:
        print(f"loss gradient at {(r['s'] - t0) / 1e6:.3f} ms")
cp = [r for r in step if "copyBuffer" in r["Kernel_Name"]]
print("copyBuffer launches in the step:", len(cp), "total us", sum(r["e"] - r["s"] for r in cp) / 1e3)
# the tail of the step: last kernels of every queue
print("last launches of the step per queue (ms since the previous optimizer ended):")
for q, v in byq.items():
    for r in v[-4:]:
        print(f"   q{q} {(r['s'] - t0) / 1e6:8.3f} .. {(r['e'] - t0) / 1e6:8.3f}  {r['Kernel_Name'][:70]}")
print("first launches of the step per queue:")
for q, v in byq.items():
    for r in v[:3]:
        print(f"   q{q} {(r['s'] - t0) / 1e6:8.3f} .. {(r['e'] - t0) / 1e6:8.3f}  {r['Kernel_Name'][:70]}")
