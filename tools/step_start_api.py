"""Host API calls (rocprofv3 --hip-trace) around the idle stretch of the main queue at the start of a step: when was the first
LayerNorm of block 0 launched by the host, when did it start on the GPU, and what was the host doing in between."""
import csv, glob, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
api = list(csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])))
for r in kt: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
for r in api: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
kt.sort(key=lambda r: r["s"]); api.sort(key=lambda r: r["s"])
opt = [i for i, r in enumerate(kt) if "clip_adam_noam" in r["Kernel_Name"]]
i0 = opt[-2]
t0 = kt[i0]["s"]
mainq = kt[i0]["Queue_Id"]
m = [r for r in kt[i0:opt[-1]] if r["Queue_Id"] == mainq]
ef = next(i for i, r in enumerate(m) if "embed_finish_fwd" in r["Kernel_Name"])
a, b = m[ef], m[ef + 1]
print("GPU: %s ends %.1f us; next main-queue kernel %s starts %.1f us (gap %.1f)" % (a["Kernel_Name"][:30], (a["e"] - t0) / 1e3, b["Kernel_Name"][:30], (b["s"] - t0) / 1e3, (b["s"] - a["e"]) / 1e3))
cid = b.get("Correlation_Id")
launch = [r for r in api if r.get("Correlation_Id") == cid]
for r in launch:
    print("host: %s for that kernel called at %.1f us, returned %.1f us (relative to the same origin)" % (r["Function"], (r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3))
if launch:
    ls = launch[0]["s"]
    print("host API calls in the 400 us before that launch (start us, duration us, name):")
    for r in api:
        if ls - 400000 <= r["s"] <= ls and (r["e"] - r["s"] > 15000 or "Event" in r["Function"] or "Wait" in r["Function"]):
            print("   %9.1f %8.1f %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Function"]))
