#!/usr/bin/env python3
"""Is the NT K-loop bound by load latency x bytes in flight, or by the load path's throughput?  The same
(N, K) at row counts whose A panel does / does not stay in the 256 MB Infinity Cache between repeats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
from bench_nt import timeit, rnd, dev, dt  # noqa

tag = os.environ.get("TAG", "")
for name, N, K in [("fc2", 384, 1536), ("dqkv", 384, 1152), ("fc1", 1536, 384)]:
    for M in (32768, 65536, 131072, 196608):
        A, B = rnd(M, K), rnd(N, K)
        out = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit(lambda: ops.gemm(A, B, out), n=30)
        tiles = (M // 256) * (N // 192)
        print(f"{tag} {name} M={M:6d} N={N} K={K}: A {M*K*2/1e6:6.0f} MB  {t*1e6:7.1f} us  {2.0*M*N*K/t/1e12:5.0f} TF  "
              f"{t*1e6/ (tiles/256):6.1f} us per round of 256 tiles", flush=True)
