"""Where the host thread waits for the GPU inside a training step (steady state: the host issues a step in ~10 ms and is then held to
the GPU's pace somewhere): wall time of every library / torch call the engine makes, summed by call site."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3t_amd import ops, engine as E
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import config_c2
dev = torch.device("cuda", 0)
cfg = config_c2()
tr = bench.build_trainer(cfg, dev, "bf16", 1)
batch = synthetic_batch(cfg, 32, 1000, 120, seed=100, device=dev)
for _ in range(8): tr.step(batch)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); d = time.perf_counter() - t
        e = acc[name]; e[0] += 1; e[1] += d; e[2] = max(e[2], d)
        return r
    setattr(mod, name, g)
for n in dir(ops):
    if callable(getattr(ops, n)) and not n.startswith("_") and getattr(getattr(ops, n), "__module__", "") == ops.__name__:
        wrap(ops, n)
for cls, names in ((E._FastEvent, ("record", "wait_on")),):
    for n in names:
        f = getattr(cls, n)
        def mk(f, n):
            def g(self, *a, **k):
                t = time.perf_counter(); r = f(self, *a, **k); d = time.perf_counter() - t
                e = acc["event." + n]; e[0] += 1; e[1] += d; e[2] = max(e[2], d)
                return r
            return g
        setattr(cls, n, mk(f, n))
N = 6
t0 = time.perf_counter()
for _ in range(N): tr.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host {1e3 * (t1 - t0) / N:.2f} ms per step; calls by total host time:")
for k, (n, tot, mx) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:28s} {n / N:7.1f} calls/step  {1e3 * tot / N:8.3f} ms/step  longest {1e3 * mx:8.3f} ms")
