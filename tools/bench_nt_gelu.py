#!/usr/bin/env python3
"""The GELU-side NT GEMM shapes of the headline step (M = 195 584 token rows), one line per variant (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

dev, dt = "cuda", torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


def main():
    tag = os.environ.get("TAG", "")
    M = int(os.environ.get("M", 195584))
    A, W1 = rnd(M, 384), rnd(1536, 384)
    h = torch.empty(M, 1536, device=dev, dtype=dt)
    dg = torch.empty(M, 1536, device=dev, dtype=dt)
    b1 = torch.randn(1536, device=dev)
    rstd = torch.rand(M, device=dev) + 0.5
    g, W2t = rnd(M, 384), rnd(1536, 384)
    dpre = torch.empty(M, 1536, device=dev, dtype=dt)
    x, W2 = rnd(M, 1536), rnd(384, 1536)
    res = rnd(M, 384)
    y = torch.empty(M, 384, device=dev, dtype=dt)
    stats = torch.empty(ops.ln_stats_parts(384) * M * 2, device=dev)
    fl = 2.0 * M * 1536 * 384
    cases = [
        ("fc1 plain", lambda: ops.gemm(A, W1, h)),
        ("fc1 bias", lambda: ops.gemm(A, W1, h, bias=b1)),
        ("fc1 bias+prescale", lambda: ops.gemm(A, W1, h, bias=b1, prescale=rstd)),
        ("fc1 act1 one output", lambda: ops.gemm(A, W1, h, bias=b1, act=1)),
        ("fc1 act3 + aux", lambda: ops.gemm(A, W1, h, bias=b1, act=3, aux=dg)),
        ("fc1 act3 + aux + prescale", lambda: ops.gemm(A, W1, h, bias=b1, act=3, aux=dg, prescale=rstd)),
        ("dfc2 plain", lambda: ops.gemm(g, W2t, dpre)),
        ("dfc2 act4", lambda: ops.gemm(g, W2t, dpre, act=4, aux=dg)),
        ("fc2 bias+res", lambda: ops.gemm(x, W2, y, bias=b1[:384].contiguous(), residual=res)),
        ("fc2 bias+res+stats", lambda: ops.gemm(x, W2, y, bias=b1[:384].contiguous(), residual=res, ln_stats=stats)),
    ]
    for name, fn in cases:
        t = timeit(fn, 20)
        print(f"{tag:10s} {name:28s} {t*1e6:7.1f} us {fl/t/1e12:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
