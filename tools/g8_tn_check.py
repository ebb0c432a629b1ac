"""8-phase TN kernel (weight gradients) vs the 128x128 kernel and torch; timing at the benchmark shapes."""
import sys
import torch
from a3t_amd import _lib, ops
from a3t_amd._lib import BF16
DEV = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc

def timeit(fn, n=50):
    for _ in range(300): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

bad = 0
for (B, T, cin, cout, taps) in [(3, 200, 128, 512, 3), (5, 1120, 384, 1536, 3), (5, 1120, 1536, 384, 3), (4, 264, 256, 264, 1), (4, 1000, 384, 1152, 1),
                               (2, 1800, 512, 2048, 3), (2, 1800, 2048, 512, 3), (7, 333, 384, 384, 1), (6, 520, 384, 768, 1), (3, 77, 128, 136, 3),
                               (9, 1120, 640, 200, 1)]:
    M = B * T
    dy, x = rn(M, cout).bfloat16(), rn(M, cin).bfloat16()
    outs = []
    for mode in (0, 1):
        lib.a3t_gemm_8p_mode(mode)
        if taps > 1:
            dW = torch.zeros(cout, taps, cin, device=DEV)
            ops.conv_bwd_weight(dy, x, dW, T, 1, alpha=0.5, compute=BF16)
        else:
            dW = torch.zeros(cout, cin, device=DEV)
            ops.linear_bwd_weight(dy, x, dW, alpha=0.5, compute=BF16)
        outs.append((dW, lib.a3t_gemm_last_kernel().decode()))
    lib.a3t_gemm_8p_mode(2)
    torch.cuda.synchronize()
    # torch reference
    if taps > 1:
        xs = x.float().view(B, T, cin)
        ref = torch.zeros(cout, taps, cin, device=DEV)
        dyf = dy.float().view(B, T, cout)
        for t in range(taps):
            sh = t - 1
            xsft = torch.zeros_like(xs)
            if sh < 0: xsft[:, -sh:] = xs[:, :sh]
            elif sh > 0: xsft[:, :-sh] = xs[:, sh:]
            else: xsft = xs
            ref[:, t, :] = 0.5 * torch.einsum("btn,btc->nc", dyf, xsft)
    else:
        ref = 0.5 * dy.float().t() @ x.float()
    e0 = float((outs[0][0] - ref).abs().max() / ref.abs().max())
    e1 = float((outs[1][0] - ref).abs().max() / ref.abs().max())
    ok = "8p_tn" in outs[1][1] and e1 < 2e-3 and e1 < 4 * e0 + 1e-6
    bad += not ok
    print(f"wgrad B={B} T={T} {cin}->{cout} taps={taps}: {outs[1][1]} err {e1:.2e} (128^2: {e0:.2e}) {'ok' if ok else 'FAIL'}")

if len(sys.argv) < 2:
    B, T = 32, 1120
    M = B * T
    NSET = 4          # operand sets in rotation: 4 x (dy + x) exceeds the 256 MiB Infinity Cache for the FFN shapes ("cold" = as inside the step)
    for (cin, cout, taps) in [(384, 1536, 3), (1536, 384, 3), (384, 1152, 1), (384, 384, 1), (384, 768, 1)]:
        sets = [(rn(M, cout).bfloat16(), rn(M, cin).bfloat16()) for _ in range(NSET)]
        dW = torch.zeros(cout, taps, cin, device=DEV) if taps > 1 else torch.zeros(cout, cin, device=DEV)
        fl = 2.0 * M * cout * cin * taps
        res = []
        for mode in (0, 1):
            lib.a3t_gemm_8p_mode(mode)
            for cold in (0, 1):
                it = [0]
                def f():
                    dy, x = sets[it[0] % NSET if cold else 0]
                    it[0] += 1
                    if taps > 1:
                        ops.conv_bwd_weight(dy, x, dW, T, 1, compute=BF16)
                    else:
                        ops.linear_bwd_weight(dy, x, dW, compute=BF16)
                t = timeit(f)
                res.append(f"{lib.a3t_gemm_last_kernel().decode()} {'cold' if cold else 'warm'} {t:.1f} us ({fl / t / 1e6:.0f} TF)")
        lib.a3t_gemm_8p_mode(2)
        print(f"wgrad {cin}->{cout} taps={taps}: " + " | ".join(res))
print("FAILED" if bad else "all ok")
