#!/usr/bin/env python3
"""Run ONE kernel configuration a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
what = sys.argv[1]
dev = "cuda"
dt = torch.bfloat16
M = 100864
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(dt)
if what.startswith("nt"):
    N, K = int(sys.argv[2]), int(sys.argv[3])
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, device=dev, dtype=dt)
    fn = lambda: ops.gemm(A, B, out)
elif what == "tn":
    Mo, No = int(sys.argv[2]), int(sys.argv[3])
    A, B = rnd(M, Mo), rnd(M, No)
    out = torch.zeros(Mo, No, device=dev)
    fn = lambda: ops.gemm(A, B, out, trans=True, out_mode=2, splitk=16)
for _ in range(5):
    fn()
torch.cuda.synchronize()
