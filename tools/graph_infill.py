"""hipGraph (torch.cuda.CUDAGraph) capture of the eval-mode forward (teacher-forced infill, 4+4 blocks) against eager launches:
bit-identical output, and the latency of both at B = 1 and B = 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for B in (1, 2, 8):
    r = bench.infill_leg(dev, B=B, reps=20)
    print(B, {k: r[k] for k in ("ms", "utterances")}, flush=True)
# GPU-busy time of one B=1 forward via events around a replayed graph
from a3t_amd.collate import synthetic_batch
from a3t_amd.config import A3TConfig
from a3t_amd.engine import MLMEngine
from a3t_amd.init import xavier_init_
from a3t_amd.params import ParamStore
c = A3TConfig(); store = ParamStore(c, dev); xavier_init_(store, seed=0, bn_gamma=1.0)
eng = MLMEngine(c, store, compute="bf16", training=False)
for B in (1, 8):
    batch = synthetic_batch(c, B, 1000, 120, seed=99, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): eng.forward(batch, need_grad=False)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.forward(batch, need_grad=False)
    torch.cuda.synchronize()
    ref = eng.forward(batch, need_grad=False)["after"].clone()
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", bool(torch.equal(out["after"], ref)))
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print(f"B={B} graph replay {(time.perf_counter()-t0)/50*1e3:.3f} ms", flush=True)
    t0 = time.perf_counter()
    for _ in range(50): eng.forward(batch, need_grad=False)
    torch.cuda.synchronize()
    print(f"B={B} eager        {(time.perf_counter()-t0)/50*1e3:.3f} ms", flush=True)
