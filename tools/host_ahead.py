import time, torch, sys
sys.path.insert(0, '/root/repo')
import bench
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
dev = torch.device("cuda", 0)
cfg = config_c2()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 32, 1000, 120, seed=100, device=dev)
for _ in range(6): tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter(); hs = []
for _ in range(12):
    a = time.perf_counter(); tr.step(batch); hs.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host time per step call (ms):", " ".join(f"{h*1e3:.1f}" for h in hs))
print(f"host loop {1e3*(t1-t0)/12:.2f} ms/step, with final sync {1e3*(t2-t0)/12:.2f} ms/step")
