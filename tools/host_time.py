"""How long does the HOST need to enqueue one training step (Python + ctypes + HIP launch calls), against the GPU time of
the step?  One step at a time: sync, t0, step() returns (nothing waited for), t1, sync, t2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import A3TConfig

dev = torch.device("cuda", 0)
cfg = A3TConfig()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 32, 1000, 120, seed=1234, device=dev)
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
host, tot = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
print("host enqueue ms per step:", " ".join(f"{h:.1f}" for h in host))
print("step ms (one at a time): ", " ".join(f"{t:.1f}" for t in tot))
