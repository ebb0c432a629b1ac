import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"],r["e"]=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
rows.sort(key=lambda r:r["s"])
opt=[i for i,r in enumerate(rows) if "clip_adam_noam" in r["Kernel_Name"]]
t0=rows[opt[-2]]["s"]; t1=rows[opt[-1]]["s"]
print("step %.3f ms"%((t1-t0)/1e6))
mainq=rows[opt[-2]]["Queue_Id"]
m=[r for r in rows[opt[-2]:opt[-1]] if r["Queue_Id"]==mainq]
prev=None
for r in m[:40]:
    if prev and r["s"]-prev["e"]>20000: print("gap %.1f us at %.1f after %s before %s"%((r["s"]-prev["e"])/1e3,(prev["e"]-t0)/1e3,prev["Kernel_Name"][:40],r["Kernel_Name"][:40]))
    prev=r
