"""Every GEMM call site of one configs[1] training step, alone (one stream): kernel, shape, launches, average duration, TFLOP/s."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3t_amd import ops
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
dev = torch.device("cuda", 0)
cfg = config_c2()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 32, 1000, 120, seed=100, device=dev)
for _ in range(4): tr.step(batch)
eng = tr.engine
side, eng.side = eng.side, None
tr.step(batch); torch.cuda.synchronize()
ops.PROFILE = []
tr.step(batch); torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
eng.side = side
agg = collections.OrderedDict()
for name, flops, e0, e1, shape in prof:
    k = (name, tuple(shape))
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += flops
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("%-46s %-38s %4s %9s %9s %8s" % ("kernel", "(M, N, K, batch, taps, splitk)", "n", "avg us", "total ms", "TFLOP/s"))
for (name, shape), (n, us, fl) in rows:
    print("%-46s %-38s %4d %9.1f %9.2f %8.0f" % (name[:46], str(shape), n, us / n, us / 1e3, fl / us / 1e6))
print("total %.2f ms" % (sum(v[1] for v in agg.values()) / 1e3))
