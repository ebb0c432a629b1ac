"""Per-kernel HBM rate of the step's HBM-bound kernels: durations alone from the one-stream rocprofv3 kernel stats, bytes from the
separate --pmc passes (profiles/<round>_hbm_traffic_per_kernel.json).  python tools/hbm_bound_table.py r05 > profiles/r05_hbm_bound_kernels.txt"""
import csv, json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
d = sys.argv[2] if len(sys.argv) > 2 else "profiles"      # (tools/r06_profiles.sh reads the fresh files from gpurun_out)
steps = 7            # bench.py --steps 5 --warmup 2 under rocprofv3 (tools/r06_profiles.sh)
rows = list(csv.DictReader(open(f"{d}/{rnd}_step_bf16_kernel_stats_one_stream.csv")))
t = json.load(open(f"{d}/{rnd}_hbm_traffic_per_kernel.json"))
print(f"HBM-bound kernels of the configs[1] step ({rnd}, one run of the final code).  Duration alone = one-stream rocprofv3 kernel")
print(f"stats over {steps} steps (profiles/{rnd}_step_bf16_kernel_stats_one_stream.csv); bytes = separate --pmc FETCH_SIZE (x2 on gfx950) /")
print(f"WRITE_SIZE passes of the same command, per launch (profiles/{rnd}_hbm_traffic_per_kernel.json).  Achievable streaming rate: 6.3 TB/s.")
print("%-78s %7s %9s %9s %8s %9s" % ("kernel", "n/step", "avg us", "MB/launch", "TB/s", "ms/step"))
tot = 0.0
for r in rows:
    n = r["Name"]
    if "gemm" in n or "attn_fwd32" in n or "rocclr" in n:
        continue
    key = next((k for k in t if k == n), None) or next((k for k in t if k.startswith(n[:60])), None)
    us, ms = float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e6
    tot += ms
    if ms < 0.04:
        continue
    mb = t[key]["hbm_bytes_per_launch"] / 1e6 if key else float("nan")
    print("%-78s %7.1f %9.1f %9.1f %8.2f %9.2f" % (n[:78], int(r["Calls"]) / steps, us, mb, mb / us, ms))
print("total of the non-GEMM, non-attention-forward kernels: %.1f ms per step alone" % tot)
