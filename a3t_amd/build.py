"""Build recipe for liba3t_hip.so: hipcc --offload-arch=gfx950, in-tree (a3t_amd/lib/)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liba3t_hip.so")
SOURCES = ["gemm.hip", "gemm_bf16.hip", "gemm_bf16_8p.hip", "gemm_bf16_pn.hip", "gemm_bf16_tt.hip", "attn_fused.hip", "norm_reduce.hip", "convmod_attn.hip", "dwconv_vec.hip", "misc.hip", "pwg_fused.hip", "features.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"] + os.environ.get("A3T_EXTRA_FLAGS", "").split()


# The direct-to-LDS GEMM variants are tuned to a register budget (<= 128 VGPRs = 4 workgroups per CU); a harmless
# looking edit can push one over the edge and cost 30-50 % on that GEMM class.  The build records what the compiler
# allocated; tests/test_host_logic.py::test_gemm_register_budget checks it.
RES_SOURCES = ("gemm_bf16.hip", "gemm_bf16_8p.hip", "gemm_bf16_pn.hip", "gemm_bf16_tt.hip", "attn_fused.hip", "norm_reduce.hip")
RES_FLAG = ["-Rpass-analysis=kernel-resource-usage"]


def _save_resources(report, path):
    import json
    import re
    cur, rows = None, {}
    for ln in report.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([\w \[\]/]+?): (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    with open(path, "w") as f:
        json.dump(rows, f, indent=1)


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hdr = os.path.join(os.path.dirname(HERE), "include", "a3t_hip.h")
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src, hdr, os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "dtype_io.h")]):
            jobs.append([_hipcc(), *FLAGS, "-c", src, "-o", obj] + (RES_FLAG if s in RES_SOURCES else []))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        if RES_FLAG[0] in cmd:     # keep the compiler's per-kernel register / occupancy report next to the object
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            if r.returncode:
                sys.stderr.write(r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            _save_resources(r.stderr, cmd[cmd.index("-o") + 1].replace(".o", ".resources.json"))
            return
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
