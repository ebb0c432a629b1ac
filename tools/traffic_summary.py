"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, MI355X_MICROARCH.md
§HBM) into per-kernel HBM bytes per launch.  FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B
(rocprofv3 derived metric: TCC_EA*_RDREQ*64B/1024); on gfx950 FETCH_SIZE reports half the bytes of
wide coalesced streams, so the read side is doubled (guide's gfx950 correction)."""
import csv, json, sys, collections

def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg

f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in f:
    n = f[k][0]
    rd = f[k][1] / n * 1024 * 2.0          # gfx950: x2
    wr = (w[k][1] / w[k][0] * 1024) if k in w else 0.0
    out[k] = dict(launches=n, fetch_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k[:70]:70s} n={v['launches']:4d} rd={v['fetch_bytes_per_launch']/1e6:9.1f} MB wr={v['write_bytes_per_launch']/1e6:9.1f} MB")
