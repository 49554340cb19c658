#!/usr/bin/env python3
"""The other one-workgroup-per-CU kernels of the step -- the grouped weight gradients of a block, the fused attention
backward, the attention forward -- while another stream's kernel holds CUs (dino_diag_occupy_cus), against the same
launches alone (GPU only).  Companion of tools/bench_nt_occupied.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

dev, dt = "cuda", torch.bfloat16
B = int(os.environ.get("B", 256))
H, D = 6, 384


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


def timed(fn, before=None):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if before is not None:
        before()
        time.sleep(0.0003)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


cases = []
M = B * (2 * 197 + 10 * 37)
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
cases.append(("grouped weight gradients (3 launches)", lambda: [ops.gemm_group(calls) for _ in range(3)]))
for N, Bc in ((197, 2 * B), (37, 10 * B)):
    qkv = rnd(Bc * N, 3 * D)
    out, lse = ops.attn_fwd(qkv, Bc, N, H)
    dout = rnd(Bc * N, D)
    dq = torch.empty_like(qkv)
    cases.append((f"attention forward N={N} (6 launches)", lambda qkv=qkv, Bc=Bc, N=N, out=out: [ops.attn_fwd(qkv, Bc, N, H, out=out) for _ in range(6)]))
    cases.append((f"attention backward N={N} (3 launches)", lambda qkv=qkv, Bc=Bc, N=N, out=out, dout=dout, lse=lse, dq=dq:
                  [ops.attn_bwd(qkv, out, dout, lse, Bc, N, H, dqkv=dq) for _ in range(3)]))
side = torch.cuda.Stream()
if os.environ.get("QUEUE", "0") == "1":
    ops.gemm_work_queue(True)
print("work queue", ops.gemm_work_queue(), flush=True)
for name, fn in cases[:int(os.environ.get("CASES", 99))]:
    fn()
    t0 = min(timed(fn) for _ in range(3))
    line = f"{name:44s} alone {t0:7.0f} us;"
    for n in (8, 16, 32):
        t = min(timed(fn, before=lambda: ops.occupy_cus(n, 6000, stream=side)) for _ in range(3))
        line += f"  {n} CUs held {t:7.0f} us = {t / t0:4.2f} x"
    print(line, flush=True)
