#!/usr/bin/env python3
"""LayerNorm forward / backward bandwidth by row width (bf16), with and without the fused
parameter-gradient / column-sum outputs and the partials workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

dev = "cuda"
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for D in (384, 768, 1024):
    rows = 100864 * 384 // D
    x = torch.randn(rows, D, device=dev).bfloat16()
    dy = torch.randn(rows, D, device=dev).bfloat16()
    dres = torch.randn(rows, D, device=dev).bfloat16()
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    nbytes = rows * D * 2
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
    print(f"D={D} rows={rows} fwd {t:7.1f} us {2 * nbytes / t / 1e6:5.2f} TB/s")
    dg, db, cs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx = torch.empty_like(x)
    for name, kw in (("plain", {}), ("+dres", dict(dres=dres)),
                     ("+dres+dgamma", dict(dres=dres, dgamma=dg, dbeta=db)),
                     ("+dres+dgamma+colsum", dict(dres=dres, dgamma=dg, dbeta=db, dx_colsum=cs)),
                     ("+dres+dgamma+colsum+ws", dict(dres=dres, dgamma=dg, dbeta=db, dx_colsum=cs, workspace=ws))):
        t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx=dx, dx_stride=D, **kw))
        nb = (3 + ("dres" in kw)) * nbytes
        print(f"   bwd {name:24s} {t:7.1f} us {nb / t / 1e6:5.2f} TB/s")
