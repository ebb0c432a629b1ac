import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from a3t_amd import ops
from test_gpu_attn_fused import _inputs, _exact
B, H, T, dk = (int(x) for x in sys.argv[1:5])
lengths = eval(sys.argv[5]) if len(sys.argv) > 5 else None
qkv, qu, qv, P, keymask = _inputs(B, H, T, dk, seed=T + dk, lengths=lengths)
d = H * dk
ctx = torch.zeros(B * T, d, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda")
ops.attn_fwd(qu, qv, qkv, P, keymask, ctx, lse, B, H, T, 1.0 / math.sqrt(dk))
torch.cuda.synchronize()
ref, _, rlse = _exact(qkv, qu, qv, P, keymask, B, H, T, dk)
err = (ctx.float().cpu().double() - ref).abs().view(B, T, H, dk)
for b in range(B):
    for h in range(H):
        e = err[b, :, h, :]
        rows = e.max(dim=1).values
        bad = (rows > 0.05).nonzero().flatten().tolist()
        print(f"b{b} h{h}: max {float(e.max()):.3f}; bad rows {bad[:12]}{'...' if len(bad) > 12 else ''} n={len(bad)}; bad cols", (e.max(dim=0).values > 0.05).nonzero().flatten().tolist()[:16])
le = (lse.cpu().double() - rlse)
print("lse err", float(le[torch.isfinite(rlse)].abs().max()))
