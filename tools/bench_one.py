import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
dev="cuda"; dt=torch.bfloat16; M=100864
def rnd(*s): return (torch.randn(*s, device=dev)*0.5).to(dt)
for N,K in [(384,1536),(1152,384),(1536,384)]:
    A,B=rnd(M,K),rnd(N,K); out=torch.empty(M,N,device=dev,dtype=dt)
    for _ in range(3): ops.gemm(A,B,out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(A,B,out)
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/20*1e-3
    print(f"DBG={os.environ.get('DINO_GEMM_DBG','0')} N={N} K={K} {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)
