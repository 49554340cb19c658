#!/usr/bin/env python3
"""Does the row pitch of the A operand matter?  fc2 / fc1-dgrad shapes (K = 1536) with the activation rows at a
pitch of 1536, 1568, 1600, 1664 elements (strided views of a wider buffer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
from bench_nt import timeit, rnd, dev, dt

for M in (94720, 195584):
    for name, N, K in [("fc2", 384, 1536), ("dqkv", 384, 1152), ("qkv", 1152, 384)]:
        B = rnd(N, K)
        out = torch.empty(M, N, device=dev, dtype=dt)
        line = f"M={M} {name} N={N} K={K}:"
        for pad in (0, 32, 64, 128):
            big = rnd(M, K + pad)
            A = big[:, :K]
            t = timeit(lambda: ops.gemm(A, B, out), n=30)
            line += f"  pitch {K + pad}: {t*1e6:6.1f} us {2.0*M*N*K/t/1e12:5.0f} TF"
        print(line, flush=True)
