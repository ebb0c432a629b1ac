"""C2-scale stability run: N optimizer steps of the bf16 production schedule on a few rotating synthetic batches with the
recipe's optimizer settings (Adam, Noam warm-up 4000, clip 1.0, dropout on): loss must fall and stay finite."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
cfg = config_c2()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batches = [synthetic_batch(cfg, 32, 1000, 120, seed=100 + i, device=dev) for i in range(4)]
losses = []
for step in range(n):
    loss = tr.step(batches[step % 4])
    if step % 25 == 0 or step == n - 1:
        torch.cuda.synchronize()
        l = float(loss)
        losses.append(l)
        print(f"step {step:4d}  loss {l:10.4f}  grad-norm {float(tr.norm):10.4f}", flush=True)
        assert math.isfinite(l) and math.isfinite(float(tr.norm))
assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])
print("ok: loss", losses[0], "->", losses[-1])
