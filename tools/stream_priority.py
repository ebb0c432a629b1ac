"""Does a stream-priority split help the two-stream backward?  The weight gradients (side stream) are off the critical
path; the data-gradient chain and its row kernels (main stream) are on it.  Runs the C2 train step with every combination
of (main priority, side priority) the device offers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
dev = torch.device("cuda", 0)
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
cfg = config_c2()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 32, 1000, 120, seed=1234, device=dev)
lo, hi = torch.cuda.Stream.priority_range()
def run(main_prio, side_prio, steps=12):
    eng = tr.engine
    if side_prio is not None and eng.side is not None:
        eng.side = torch.cuda.Stream(device=dev, priority=side_prio)
    ms = torch.cuda.Stream(device=dev, priority=main_prio) if main_prio is not None else torch.cuda.current_stream()
    ms.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ms):
        for _ in range(3): tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): tr.step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    torch.cuda.current_stream().wait_stream(ms)
    return dt * 1e3
for mp, sp in [(None, None), (hi, lo), (hi, None), (None, lo), (lo, hi), (None, None)]:
    print(f"main priority {mp}, side priority {sp}: {run(mp, sp):.2f} ms/step", flush=True)
