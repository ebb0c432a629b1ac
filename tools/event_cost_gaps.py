import csv, sys
from collections import Counter
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
big = [r for r in rows if int(r["Grid_Size_X"]) >= (1 << 24)]
print(Counter((r["Queue_Id"], r["Kernel_Name"][:60]) for r in big))
phases, cur = [], None
for r in big:
    if "MulFunctor" in r["Kernel_Name"] or "mul" in r["Kernel_Name"].lower():
        if cur is not None:
            phases.append(cur)
        cur = []
    elif cur is not None:
        cur.append(r)
phases.append(cur)
names = ("none", "torch", "raw_default", "raw_nofence", "raw_release_dev", "torch_wait_only")
for n, ph in zip(names, phases):
    gaps = [b["s"] - a["e"] for a, b in zip(ph, ph[1:])]
    dur = [a["e"] - a["s"] for a in ph]
    gaps.sort()
    print(f"{n:18s} kernels {len(ph)}  median gap {gaps[len(gaps)//2]/1e3:.2f} us  mean {sum(gaps)/len(gaps)/1e3:.2f} us  kernel {sum(dur)/len(dur)/1e3:.1f} us")
