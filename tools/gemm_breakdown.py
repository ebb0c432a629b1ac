"""Per-shape GEMM time breakdown of one training step (HIP events around every launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3t_amd import ops
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
dev = torch.device("cuda", 0)
cfg = config_c2()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 32, 1000, 120, seed=1234, device=dev)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
ops.PROFILE = []
tr.step(batch)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
agg = {}
for name, fl, e0, e1, shape in prof:
    k = (name[name.index("<"):], shape)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"total gemm time {tot*1e3:.2f} ms, {sum(a[2] for a in agg.values())/tot/1e12:.1f} TF avg")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"L{k[0]} MNKb,taps,sk={k[1]}: n={a[0]:3d} t={a[1]*1e3:7.2f} ms ({a[1]/tot*100:4.1f}%) avg {a[1]/a[0]*1e6:7.1f} us  {a[2]/a[1]/1e12:6.1f} TF")
