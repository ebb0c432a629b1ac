"""Per-kernel register / occupancy table of one HIP source: python tools/kres.py a3t_amd/csrc/gemm_bf16.hip [filter]"""
import re, subprocess, sys
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", sys.argv[1], "-o", "/tmp/kres.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w \[\]/]+?): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in ln:
        print(ln)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in rows.items():
    if flt in k:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{name[:70]:70s} vgpr {v.get('VGPRs', -1):4d} agpr {v.get('AGPRs', -1):3d} spill {v.get('VGPRs Spill', -1):3d} "
              f"occ {v.get('Occupancy [waves/SIMD]', -1)} lds {v.get('LDS Size [bytes/block]', -1)}")
