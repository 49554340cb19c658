"""Run the attention kernels alone (for rocprofv3 counter passes): python3 tools/run_attn.py N Bc"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 197
Bc = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = 6
torch.manual_seed(0)
qkv = torch.randn(Bc * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
do = torch.randn(Bc * N, H * 64, device="cuda").to(torch.bfloat16)
for _ in range(3):
    o, lse = ops.attn_fwd(qkv, Bc, N, H)
    d = ops.attn_bwd(qkv, o, do, lse, Bc, N, H)
torch.cuda.synchronize()
