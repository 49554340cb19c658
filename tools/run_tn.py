#!/usr/bin/env python3
"""Runs one bf16 TN (weight-gradient) GEMM a few times (for rocprofv3 counter passes).  SHAPE=tokens,Mo,No"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
K, Mo, No = (int(v) for v in os.environ.get("SHAPE", "100864,1536,384").split(","))
A = (torch.randn(K, Mo, device="cuda") * 0.5).bfloat16()
B = (torch.randn(K, No, device="cuda") * 0.5).bfloat16()
out = torch.zeros(Mo, No, device="cuda")
ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for _ in range(int(os.environ.get("REPS", "6"))):
    ops.gemm(A, B, out, trans=True, out_mode=2, splitk=16, workspace=ws)
torch.cuda.synchronize()
