"""Randomised bf16 GEMM / conv sweep against fp32 torch math: exercises the dispatcher's variants (NT / NN / TN x
plain / fast-conv / generic x single / double buffer x 256x256 kernel) with ragged sizes and epilogue options."""
import sys, os, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from a3t_amd import ops, _lib
from a3t_amd._lib import BF16, ACT_RELU, ACT_NONE
dev = "cuda"
lib = _lib.load()
rng = random.Random(0)
fails, seen = 0, {}

def check(name, got, ref, k):
    global fails
    err = float((got.float() - ref).abs().max() / (ref.abs().max() + 1e-6))
    var = lib.a3t_gemm_last_kernel().decode()
    seen[var] = seen.get(var, 0) + 1
    if not (err < 2e-2) or not math.isfinite(err):
        fails += 1
        print(f"FAIL {name}: relerr {err:.3e}  kernel {var}")

desc = [""]


def run_case(case):
    global fails
    kind = rng.choice(["linear", "conv"])
    if kind == "linear":
        M = rng.choice([rng.randrange(8, 400), rng.randrange(400, 6000), 128 * rng.randrange(1, 40), 35840 if rng.random() < 0.15 else 1120])
        N = 8 * rng.randrange(1, 100)
        K = 8 * rng.randrange(1, 160)
        desc[0] = f"linear M{M} N{N} K{K}"
        x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        W = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev) if rng.random() < 0.5 else None
        R = torch.randn(M, N, device=dev) if rng.random() < 0.3 else None
        act = ACT_RELU if rng.random() < 0.4 else ACT_NONE
        odt = torch.bfloat16 if (R is None and rng.random() < 0.6) else torch.float32
        out = torch.empty(M, N, device=dev, dtype=odt)
        ops.linear_fwd(x, W, out, bias=bias, R=R, act=act, alpha=0.7, compute=BF16)
        ref = x.float() @ W.float().t()
        if bias is not None: ref = ref + bias
        if act == ACT_RELU: ref = torch.relu(ref)
        ref = 0.7 * ref + (R if R is not None else 0)
        check(f"linear_fwd {M}x{N}x{K}", out, ref, K)
        dy = (torch.randn(M, N, device=dev) * 0.5).bfloat16()
        dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        ops.linear_bwd_data(dy, W, dx, compute=BF16)
        check(f"linear_bwd_data {M}x{K}x{N}", dx, dy.float() @ W.float(), N)
        dW = torch.zeros(N, K, device=dev)
        ops.linear_bwd_weight(dy, x, dW, compute=BF16)
        check(f"linear_bwd_weight {N}x{K}x{M}", dW, dy.float().t() @ x.float(), M)
    else:
        taps = rng.choice([3, 3, 5])
        dil = rng.choice([1, 1, 2])
        B = rng.randrange(1, 9)
        T = rng.choice([rng.randrange(taps * dil + 1, 300), 1120, 1030])
        Cin = rng.choice([8 * rng.randrange(1, 30), 64 * rng.randrange(1, 8), 128 * rng.randrange(1, 4)])
        Cout = rng.choice([8 * rng.randrange(1, 40), 128 * rng.randrange(1, 13)])
        M, pad = B * T, (taps - 1) // 2
        desc[0] = f"conv B{B} T{T} Cin{Cin} Cout{Cout} k{taps} d{dil}"
        x = (torch.randn(B, T, Cin, device=dev) * 0.5).bfloat16()
        Wt = (torch.randn(Cout, Cin, taps, device=dev) * (Cin * taps) ** -0.5).bfloat16()
        ref = F.conv1d(x.float().transpose(1, 2), Wt.float(), None, padding=pad * dil, dilation=dil).transpose(1, 2).reshape(M, Cout)
        Wk = Wt.permute(0, 2, 1).contiguous()
        out = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
        ops.conv_fwd(x.reshape(M, Cin), Wk, out, T, pad, dil, compute=BF16)
        check(f"conv_fwd B{B} T{T} {Cin}->{Cout} k{taps} d{dil}", out, ref, Cin * taps)
        dy = (torch.randn(B, T, Cout, device=dev) * 0.5).bfloat16()
        xr = x.float().clone().requires_grad_(True)
        Wr = Wt.float().clone().requires_grad_(True)
        F.conv1d(xr.transpose(1, 2), Wr, None, padding=pad * dil, dilation=dil).transpose(1, 2).backward(dy.float())
        dx = torch.empty(M, Cin, device=dev, dtype=torch.float32)
        ops.conv_bwd_data(dy.reshape(M, Cout), Wk, dx, T, pad, dil, compute=BF16)
        check(f"conv_bwd_data B{B} T{T} {Cout}->{Cin} k{taps} d{dil}", dx, xr.grad.reshape(M, Cin), Cout * taps)
        dWk = torch.zeros(Cout, taps, Cin, device=dev)
        ops.conv_bwd_weight(dy.reshape(M, Cout), x.reshape(M, Cin), dWk, T, pad, dil, compute=BF16)
        check(f"conv_bwd_weight B{B} T{T} {Cout}x{Cin} k{taps} d{dil}", dWk, Wr.grad.permute(0, 2, 1), M)

def run(seed=0, n_cases=120, verbose=True):
    """Returns (number of failed checks / exceptions, {kernel variant: launches checked})."""
    global fails, seen, rng
    rng = random.Random(seed)
    fails, seen = 0, {}
    for case in range(n_cases):
        try:
            run_case(case)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"EXC case {case} [{desc[0]}]: {type(e).__name__}: {e}")
    torch.cuda.synchronize()
    if verbose:
        print(f"{n_cases} cases, {fails} failures; kernel variants hit:")
        for k, v in sorted(seen.items()):
            print(f"   {v:4d}  {k}")
    return fails, dict(seen)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 120)
