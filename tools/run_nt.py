#!/usr/bin/env python3
"""Runs one bf16 NT GEMM shape a few times (for rocprofv3 counter passes).  SHAPE=M,N,K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
M, N, K = (int(v) for v in os.environ.get("SHAPE", "100864,384,1536").split(","))
A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
B = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(int(os.environ.get("REPS", "10"))):
    ops.gemm(A, B, out)
torch.cuda.synchronize()
