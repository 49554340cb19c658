#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernels on the ViT-S/16 B=256 shapes (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

dev = "cuda"
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32


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


M = 100864
print(f"dtype {dt}")
for name, N, K in [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)]:
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, device=dev, dtype=dt)
    bias = torch.randn(N, device=dev)
    res = rnd(M, N)
    aux = torch.empty(M, N, device=dev, dtype=dt)
    fl = 2.0 * M * N * K
    for vname, kw in [("plain", {}), ("bias", dict(bias=bias)), ("bias+res", dict(bias=bias, residual=res)),
                      ("bias+gelu+aux", dict(bias=bias, act=1, aux=aux)), ("gelu'", dict(act=2, aux=res))]:
        t = timeit(lambda: ops.gemm(A, B, out, **kw))
        print(f"NT {name:5s} M={M} N={N:5d} K={K:5d} {vname:14s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF")
for name, Mo, No in [("w_qkv", 1152, 384), ("w_proj", 384, 384), ("w_fc1", 1536, 384), ("w_fc2", 384, 1536)]:
    A, B = rnd(M, Mo), rnd(M, No)
    out = torch.zeros(Mo, No, device=dev)
    fl = 2.0 * M * Mo * No
    for sk in (8, 16, 32, 64):
        t = timeit(lambda: ops.gemm(A, B, out, trans=True, out_mode=2, splitk=sk))
        print(f"TN {name:6s} out={Mo}x{No} K={M} splitk={sk:3d} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF")
# head
R = 3072
A, B = rnd(R, 256), rnd(65536, 256)
out = torch.empty(R, 65536, device=dev, dtype=dt)
t = timeit(lambda: ops.gemm(A, B, out))
print(f"NT proto fwd {t*1e6:8.1f} us {2.0*R*65536*256/t/1e12:7.1f} TF")
dS, Wt = rnd(R, 65536), rnd(256, 65536)
o32 = torch.zeros(R, 256, device=dev)
for sk in (8, 32, 64):
    t = timeit(lambda: ops.gemm(dS, Wt, o32, out_mode=2, splitk=sk))
    print(f"NT proto dgrad splitk={sk} {t*1e6:8.1f} us {2.0*R*65536*256/t/1e12:7.1f} TF")
dW = torch.zeros(65536, 256, device=dev)
yn = rnd(R, 256)
for sk in (1, 2, 4):
    t = timeit(lambda: ops.gemm(dS, yn, dW, trans=True, out_mode=2, splitk=sk))
    print(f"TN proto wgrad splitk={sk} {t*1e6:8.1f} us {2.0*R*65536*256/t/1e12:7.1f} TF")
# row kernels
x = rnd(M, 384)
g, b = torch.ones(384, device=dev), torch.zeros(384, device=dev)
t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
print(f"ln_fwd {t*1e6:8.1f} us {2*M*384*x.element_size()/t/1e9:7.0f} GB/s")
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
dg, db = torch.zeros(384, device=dev), torch.zeros(384, device=dev)
dx = torch.empty_like(x)
t = timeit(lambda: ops.layernorm_bwd(y, x, g, mean, rstd, dres=y, dx=dx, dgamma=dg, dbeta=db))
print(f"ln_bwd {t*1e6:8.1f} us {4*M*384*x.element_size()/t/1e9:7.0f} GB/s")
cs = torch.zeros(1536, device=dev)
big = rnd(M, 1536)
t = timeit(lambda: ops.colsum(big, cs))
print(f"colsum1536 {t*1e6:8.1f} us {M*1536*x.element_size()/t/1e9:7.0f} GB/s")
