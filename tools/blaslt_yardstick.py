"""Yardstick only (never on the product path): what the vendor library (torch.matmul -> hipBLASLt) reaches on the
model's GEMM shapes, to judge how much head-room the hand-written kernels have left."""
import torch
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def run(tag, b, M, N, K, layout):
    bf = torch.bfloat16
    if layout == "NT":      # A (M,K) row-major, B given as (N,K): nn.Linear
        A = torch.randn(b, M, K, device=dev, dtype=bf); B = torch.randn(b, N, K, device=dev, dtype=bf)
        f = lambda: torch.matmul(A, B.transpose(1, 2))
    elif layout == "NN":
        A = torch.randn(b, M, K, device=dev, dtype=bf); B = torch.randn(b, K, N, device=dev, dtype=bf)
        f = lambda: torch.matmul(A, B)
    else:                   # TN: A stored (K,M), B stored (K,N)
        A = torch.randn(b, K, M, device=dev, dtype=bf); B = torch.randn(b, K, N, device=dev, dtype=bf)
        f = lambda: torch.matmul(A.transpose(1, 2), B)
    t = timeit(f)
    print(f"{tag:34s} {layout} b={b:3d} M={M:6d} N={N:5d} K={K:6d}: {t*1e6:8.1f} us  {2.0*b*M*N*K/t/1e12:7.1f} TF", flush=True)


run("ffn conv1 as im2col GEMM", 1, 35840, 1536, 1152, "NT")
run("ffn conv2 as im2col GEMM", 1, 35840, 384, 4608, "NT")
run("ffn conv1 dgrad", 1, 35840, 1152, 1536, "NN")
run("ffn conv1 wgrad", 1, 1536, 1152, 35840, "TN")
run("ffn conv2 wgrad", 1, 384, 4608, 35840, "TN")
run("qkv fused", 1, 35840, 1152, 384, "NT")
run("proj d->d", 1, 35840, 384, 384, "NT")
run("attn scores q k^T", 64, 1120, 1120, 192, "NT")
run("attn p v", 64, 1120, 192, 1120, "NN")
run("attn p^T dO", 64, 1120, 192, 1120, "TN")
run("square 4096", 1, 4096, 4096, 4096, "NT")
run("square 8192", 1, 8192, 8192, 8192, "NT")
