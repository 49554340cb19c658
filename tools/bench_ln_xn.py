#!/usr/bin/env python3
"""The LayerNorm backward of the headline step (bf16, 195 584 rows x 384, residual gradient added, parameter gradients
and bias-gradient column sums as workgroup partials) with and without the normalised-row output (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

dev = "cuda"
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tag = os.environ.get("TAG", "")
rows, D = 195584, 384
x = torch.randn(rows, D, device=dev).bfloat16()
dy = torch.randn(rows, D, device=dev).bfloat16()
dres = torch.randn(rows, D, device=dev).bfloat16()
g, b = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
mean, rstd = ops.layernorm_stats(x, 1e-6)
dg, db, cs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
rs = torch.rand(rows, device=dev)
xn = torch.empty_like(x)
nbytes = rows * D * 2
t0 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg, dbeta=db, dx_colsum=cs, rowscale=rs,
                                      rows_per_group=1, workspace=ws))
t1 = timeit(lambda: ops.layernorm_bwd_xn(dy, x, g, b, mean, rstd, xn=xn, dres=dres, dgamma=dg, dbeta=db, dx_colsum=cs,
                                         rowscale=rs, rows_per_group=1, workspace=ws))
print(f"{tag:8s} ln_bwd {t0:6.1f} us {4 * nbytes / t0 / 1e6:5.2f} TB/s   ln_bwd_xn {t1:6.1f} us {5 * nbytes / t1 / 1e6:5.2f} TB/s "
      f"({5 * nbytes / t1 / 8e6:.3f} of 8 TB/s)", flush=True)
