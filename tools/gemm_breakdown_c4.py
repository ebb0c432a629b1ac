"""Per-shape GEMM time breakdown of one configs[3] training step (one stream, HIP events around every launch) + non-GEMM total."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import bench
from a3t_amd import ops
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c4
dev = torch.device("cuda", 0)
cfg = config_c4()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 16, 1600, 200, seed=4321, device=dev)
tr.engine.side = None
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / 4 * 1e3
ops.PROFILE = []
tr.step(batch)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
agg = {}
for name, fl, e0, e1, shape in prof:
    k = (name, shape)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"one-stream step {one:.2f} ms; gemm time {tot*1e3:.2f} ms, {sum(a[2] for a in agg.values())/tot/1e12:.1f} TF avg; non-gemm + gaps {one - tot*1e3:.2f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k[0][:44]:44s} MNKb,taps,sk={k[1]}: n={a[0]:3d} t={a[1]*1e3:7.2f} ms avg {a[1]/a[0]*1e6:7.1f} us  {a[2]/a[1]/1e12:6.1f} TF")
