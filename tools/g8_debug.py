import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import ACT_NONE, BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
B, T, d, ff = 3, 200, 128, 512
M = B * T
y = rn(M, d).bfloat16()
W1 = rn(ff, 3, d, sc=0.03).bfloat16()
outs = []
for mode in (0, 1):
    lib.a3t_gemm_8p_mode(mode)
    h = torch.empty(M, ff, device=DEV, dtype=torch.float32)
    ops.conv_fwd(y, W1, h, T, 1, compute=BF16)
    outs.append(h)
torch.cuda.synchronize()
a, b = outs
diff = (a - b).abs()
rows = (diff.max(dim=1).values > 1e-3).nonzero().flatten().tolist()
cols = (diff.max(dim=0).values > 1e-3).nonzero().flatten().tolist()
print("bad rows", len(rows), rows[:40])
print("bad cols", len(cols), cols[:40])
# which taps are missing? compare with per-tap partial sums
yf, Wf = y.float(), W1.float()
for r in rows[:6]:
    parts = []
    for tap in range(3):
        t = r % T + tap - 1
        parts.append((yf[r + tap - 1] @ Wf[:, tap, :].t()) if 0 <= t < T else torch.zeros(ff, device=DEV))
    full = sum(parts)
    print("row", r, "tpos", r % T, "ref ok", float((a[r] - full).abs().max()), "8p-err", float((b[r] - full).abs().max()),
          [float((b[r] - (full - parts[k])).abs().max()) for k in range(3)])
