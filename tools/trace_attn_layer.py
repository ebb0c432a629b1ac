"""One attention sub-layer of the backward out of a rocprofv3 kernel trace (tools/trace_step.sh): every launch of every queue between
the output projection's data gradient and the q|k|v projection's, with start / end relative to the score-gradient kernel's start.
usage: python tools/trace_attn_layer.py gpurun_out/trace_step.csv [which-layer]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ds = [r for r in rows if "attn_bwd_ds_kernel" in r["Kernel_Name"]]
d = ds[-12 + which] if len(ds) >= 12 else ds[which]
t0 = d["s"]
qid = lambda r: r.get("Queue_Id", r.get("Queue_ID", "?"))
sel = [r for r in rows if r["e"] > t0 - 150_000 and r["s"] < t0 + 700_000]
for r in sel:
    name = r["Kernel_Name"].replace("void ", "")[:62]
    print(f"q{qid(r):>3s}  {(r['s'] - t0) / 1e3:8.1f} .. {(r['e'] - t0) / 1e3:8.1f} us  ({(r['e'] - r['s']) / 1e3:6.1f})  {name}")
