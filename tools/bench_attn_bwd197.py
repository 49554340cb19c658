#!/usr/bin/env python3
"""Timing of the 197-token attention backward alone (TAG env names the line; DINO_ATTN_ONEPASS picks the kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

tag = os.environ.get("TAG", "")
N, Bc, H = 197, 512, 6
torch.manual_seed(0)
qkv = torch.randn(Bc * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
do = torch.randn(Bc * N, H * 64, device="cuda").to(torch.bfloat16)
o, lse = ops.attn_fwd(qkv, Bc, N, H)
for _ in range(3):
    ops.attn_bwd(qkv, o, do, lse, Bc, N, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    ops.attn_bwd(qkv, o, do, lse, Bc, N, H)
e1.record()
torch.cuda.synchronize()
print(f"{tag} onepass={os.environ.get('DINO_ATTN_ONEPASS', '0')} N={N}: bwd {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
