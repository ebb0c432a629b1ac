"""configs[3] step with the materialised vs the fused attention path (A3T_FUSED_ATTN), same box."""
import os, sys, json, subprocess
for v in ("auto", "1"):
    env = dict(os.environ, A3T_FUSED_ATTN=v)
    out = subprocess.run([sys.executable, "-c", "import torch, bench, json; print(json.dumps(bench.c4_leg(torch.device('cuda',0), 'bf16', steps=6, warmup=3)))"],
                         env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(f"A3T_FUSED_ATTN={v}: {d['ms_per_step']:.2f} ms/step  {d['step_tflops']:.0f} TFLOP/s  loss {d['final_loss']:.3f}")
    except Exception as e:
        print(v, "failed", out.stderr[-500:])
