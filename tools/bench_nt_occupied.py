#!/usr/bin/env python3
"""The wide-tile NT GEMM while another stream's kernel holds CUs (dino_diag_occupy_cus: what an overlapped RCCL
all-reduce does on a multi-GPU node): correctness and time of a chain of block-shaped GEMMs with n CUs held for
the whole chain, against the same chain alone.  DINO_NT_QUEUE=0/1 (static tile shares / work queue) is read by
the library once per process: run it once per setting (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops

dev, dt = "cuda", torch.bfloat16
tag = os.environ.get("TAG", "queue=" + os.environ.get("DINO_NT_QUEUE", "1"))
M = int(os.environ.get("M", 195584))
torch.manual_seed(0)


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


shapes = [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)]
ops_ = []
for name, N, K in shapes:
    A, B = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, device=dev, dtype=dt)
    ops_.append((name, A, B, out))


def chain(reps=3):
    for _ in range(reps):
        for _, A, B, out in ops_:
            ops.gemm(A, B, out)


def timed(fn, before=None):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if before is not None:
        before()                 # (the other stream's kernel is resident before the chain starts)
        time.sleep(0.0003)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


chain()
# reference results (fp32 matmul of a row sample) for the correctness check under occupation
idx = torch.randint(0, M, (4096,), device=dev)
refs = [(A[idx].float() @ B.float().t()) for _, A, B, _ in ops_]
side = torch.cuda.Stream()
t_alone = min(timed(chain) for _ in range(3))
print(f"{tag:10s} chain of {3 * len(ops_)} GEMMs alone: {t_alone:8.1f} us", flush=True)
for n, us in ((8, 6000), (16, 6000), (32, 6000), (64, 6000)):
    ts = []
    for _ in range(3):
        for _, _, _, out in ops_:
            out.zero_()
        # holds n CUs from now on, for longer than the chain takes
        ts.append(timed(chain, before=lambda: ops.occupy_cus(n, us, stream=side)))
        torch.cuda.synchronize()
        for (name, _, _, out), ref in zip(ops_, refs):
            err = ((out[idx].float() - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-2, (name, n, err)
    t = min(ts)
    print(f"{tag:10s} {n:3d} CUs held: {t:8.1f} us = {t / t_alone:5.2f} x   (ideal {256 / (256 - n):4.2f} x, "
          f"a static share behind a held CU 2 x)", flush=True)
print(f"{tag:10s} results correct under occupation", flush=True)
