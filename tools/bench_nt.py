#!/usr/bin/env python3
"""Per-shape timing of the bf16 NT GEMM epilogue variants on the ViT-S/16 B=256 shapes (GPU only)."""
import sys, os
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
    for M in (100864, 94720):
        for name, N, K in [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("dqkv", 384, 1152)]:
            A, B = rnd(M, K), rnd(N, K)
            out = torch.empty(M, N, device=dev, dtype=dt)
            bias = torch.randn(N, device=dev)
            res = rnd(M, N)
            aux = torch.empty(M, N, device=dev, dtype=dt)
            fl = 2.0 * M * N * K
            line = f"{tag} M={M} {name:5s} N={N:5d} K={K:5d}:"
            for vname, kw in [("plain", {}), ("bias", dict(bias=bias)), ("bias+res", dict(bias=bias, residual=res)),
                              ("act3", dict(bias=bias, act=3, aux=aux)), ("act4", dict(act=4, aux=res))]:
                t = timeit(lambda: ops.gemm(A, B, out, **kw), n=30)
                line += f"  {vname} {t*1e6:6.1f}us {fl/t/1e12:5.0f}TF"
            print(line, flush=True)
    # fused column sums (fc2 dgrad with the fc1 bias gradient)
    M, N, K = 100864, 1536, 384
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, device=dev, dtype=dt)
    res = rnd(M, N)
    cs = torch.zeros(N, device=dev)
    for vname, kw in [("act4", dict(act=4, aux=res)), ("act4+colsum", dict(act=4, aux=res, colsum=cs)), ("plain+colsum", dict(colsum=cs))]:
        t = timeit(lambda: ops.gemm(A, B, out, **kw), n=30)
        print(f"{tag} fc2-dgrad {vname:14s} {t*1e6:6.1f}us {2.0*M*N*K/t/1e12:5.0f}TF", flush=True)


if __name__ == "__main__":
    main()
