#!/usr/bin/env python3
"""The small NT GEMMs of a step (projection head, the last block's CLS rows): M = crops, bf16.
TAG names the line; DINO_GEMM_V1=1 forces the 128x128 register-staged kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

tag = os.environ.get("TAG", "")
dt = torch.bfloat16
for M in (3072, 512):
    for N, K in ((384, 384), (1536, 384), (384, 1536), (2048, 384), (2048, 2048), (256, 2048)):
        A = torch.randn(M, K, device="cuda").to(dt)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=dt)
        f = lambda: ops.gemm(A, B, out, bias=bias)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{tag:8s} M={M:5d} N={N:5d} K={K:5d}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF", flush=True)
