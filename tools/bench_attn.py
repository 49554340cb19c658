#!/usr/bin/env python3
"""Timing of the attention kernels alone: python tools/bench_attn.py [N Bc H]  (TAG env names the line)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

tag = os.environ.get("TAG", "")
cases = [(197, 512, 6), (37, 2560, 6)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 6)]
for N, Bc, H in cases:
    torch.manual_seed(0)
    qkv = torch.randn(Bc * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
    do = torch.randn(Bc * N, H * 64, device="cuda").to(torch.bfloat16)

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    o, lse = ops.attn_fwd(qkv, Bc, N, H)
    tf = timeit(lambda: ops.attn_fwd(qkv, Bc, N, H))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, Bc, N, H))
    print(f"{tag} N={N} Bc={Bc} H={H}: fwd {tf:7.1f} us  bwd {tb:7.1f} us", flush=True)
