#!/usr/bin/env python3
"""The grouped weight-gradient launch of one ViT-S block at the headline token count (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

dev, dt = "cuda", torch.bfloat16
M = int(os.environ.get("M", 195584))
tag = os.environ.get("TAG", "")


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


dqkv, xn1 = rnd(M, 1152), rnd(M, 384)
dy1, ao = rnd(M, 384), rnd(M, 384)
dpre, xn2 = rnd(M, 1536), rnd(M, 384)
dy2, h = rnd(M, 384), rnd(M, 1536)
gw = [torch.zeros(1152, 384, device=dev), torch.zeros(384, 384, device=dev), torch.zeros(1536, 384, device=dev),
      torch.zeros(384, 1536, device=dev)]
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
calls = [(dqkv, xn1, gw[0], dict(trans=True, out_mode=2, workspace=ws)),
         (dy1, ao, gw[1], dict(trans=True, out_mode=2, workspace=ws)),
         (dpre, xn2, gw[2], dict(trans=True, out_mode=2, workspace=ws)),
         (dy2, h, gw[3], dict(trans=True, out_mode=2, workspace=ws))]
fl = 2.0 * M * (1152 * 384 + 384 * 384 + 2 * 1536 * 384)
for _ in range(3):
    ops.gemm_group(calls)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n):
    ops.gemm_group(calls)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n * 1e-3
print(f"{tag:10s} grouped wgrad of one block: {t*1e6:7.1f} us {fl/t/1e12:6.0f} TF", flush=True)
