#!/usr/bin/env python3
"""The wide-tile NT GEMM on the block shapes of the headline step (M = 195 584 token rows): one line per shape and
build (TAG), for A/Bs of ring depths and other K-loop experiments (GPU only)."""
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
    for name, N, K in [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("dqkv", 384, 1152)]:
        A, B = rnd(M, K), rnd(N, K)
        out = torch.empty(M, N, device=dev, dtype=dt)
        bias = torch.randn(N, device=dev)
        res = rnd(M, N)
        aux = torch.empty(M, N, device=dev, dtype=dt)
        fl = 2.0 * M * N * K
        line = f"{tag:8s} {name:5s} N={N:5d} K={K:5d}:"
        variants = [("plain", {}), ("bias+res", dict(bias=bias, residual=res))]
        if name == "fc1":
            variants += [("act3", dict(bias=bias, act=3, aux=aux)), ("act4", dict(act=4, aux=res))]
        for vname, kw in variants:
            t = timeit(lambda: ops.gemm(A, B, out, **kw), n=20)
            line += f"  {vname} {t*1e6:6.1f}us {fl/t/1e12:5.0f}TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
