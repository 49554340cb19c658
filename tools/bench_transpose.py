#!/usr/bin/env python3
"""dino_transpose_batched (the W^T shadows of every 2-D weight, once per step) at ViT-S and ViT-L sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

for name, D, depth in (("vit-s", 384, 12), ("vit-l", 1024, 24)):
    rows, off = [], 0
    for _ in range(depth):
        for r, c in ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D)):
            rows.append((off, off, r, c))
            off += r * c
    src = torch.randn(off, device="cuda")
    dst = torch.empty(off, dtype=torch.bfloat16, device="cuda")
    table = torch.tensor(rows, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.transpose_batched(src, dst, table)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.transpose_batched(src, dst, table)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"{os.environ.get('TAG', '')} {name}: {off / 1e6:.1f} M elements {t * 1e6:7.1f} us  {off * 6 / t / 1e12:.2f} TB/s", flush=True)
