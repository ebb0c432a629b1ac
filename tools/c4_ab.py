"""configs[3] step under environment variants, same box: python tools/c4_ab.py "VAR=val ..." ..."""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:] or ["A3T_X=0"]:
    env = dict(os.environ)
    for kv in spec.split():
        k, v = kv.split("=")
        env[k] = v
    out = subprocess.run([sys.executable, "-c", "import torch, bench, json; print(json.dumps(bench.c4_leg(torch.device('cuda',0), 'bf16', steps=6, warmup=3)))"],
                         env=env, capture_output=True, text=True, cwd=root)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(f"[{spec}]: {d['ms_per_step']:.2f} ms/step  {d['step_tflops']:.0f} TFLOP/s  8p {d['ffn_on_8phase_gemm']}")
    except Exception as e:
        print(spec, "failed", out.stderr[-800:])
