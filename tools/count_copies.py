"""Which host-side calls produce the ~190 small device-to-device copies per step?"""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
dev = torch.device("cuda", 0)
cfg = config_c2()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 32, 1000, 120, seed=1234, device=dev)
tr.step(batch); tr.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if "aten::" in e.key or "Memcpy" in e.key or "Memset" in e.key]
for k, c, t in sorted(rows, key=lambda r: -r[1])[:18]:
    print(f"{k:40s} n={c:5d} dev_us={t:9.1f}")
