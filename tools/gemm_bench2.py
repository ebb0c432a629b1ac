import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3t_amd import ops
from a3t_amd._lib import BF16
dev="cuda"
def timeit(fn, n=30):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
for (M,N,K) in [(4096,4096,4096),(8192,8192,8192),(35840,1536,1152),(35840,1536,4608),(35840,384,1152)]:
    x=torch.randn(M,K,device=dev).bfloat16(); W=torch.randn(N,K,device=dev).bfloat16()
    for odt in (torch.bfloat16, torch.float32):
        out=torch.empty(M,N,device=dev,dtype=odt)
        t=timeit(lambda: ops.linear_fwd(x,W,out,compute=BF16))
        print(f"NT {M}x{N}x{K} out={str(odt)[6:]}: {t*1e6:9.1f} us {2.0*M*N*K/t/1e12:7.1f} TF")
    dy=torch.randn(M,N,device=dev).bfloat16(); dx=torch.empty(M,K,device=dev,dtype=torch.bfloat16)
    t=timeit(lambda: ops.linear_bwd_data(dy,W,dx,compute=BF16))
    print(f"NN                         : {t*1e6:9.1f} us {2.0*M*N*K/t/1e12:7.1f} TF")
