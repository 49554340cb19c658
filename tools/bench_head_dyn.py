"""Times the prototype layer's input gradient dyn = dlogits @ W (M = 3072, N = 256, K = 65536) for a few
split-K factors -- the 1.1 ms launch of the round-2 profile (tools/profile_round.sh)."""
import sys
import torch
from dino_amd import ops

R, K, Db = 3072, 65536, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
dl = (torch.randn((R, K), device=dev, generator=g) * 0.01).to(torch.bfloat16)
Wt = torch.randn((Db, K), device=dev, generator=g).to(torch.bfloat16)
ref = None


def run(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for sk in [int(s) for s in (sys.argv[1:] or "1 4 8 11 16 22 32 64".split())]:
    dyn = torch.zeros((R, Db), dtype=torch.float32, device=dev)
    us = run(lambda: ops.gemm(dl, Wt, dyn, out_mode=2, splitk=sk))
    dyn.zero_()
    ops.gemm(dl, Wt, dyn, out_mode=2, splitk=sk)
    if ref is None:
        ref = dl.float() @ Wt.float().t()
    err = ((dyn - ref).abs().max() / ref.abs().max()).item()
    print(f"splitk {sk:3d}: {us:8.1f} us  {2 * R * K * Db / us / 1e6:7.1f} TFLOP/s  err {err:.2e}", flush=True)

for sk in [8, 16, 21, 32, 42, 64]:
    slabs = torch.empty((sk, R, Db), dtype=torch.float32, device=dev)
    dyn = torch.empty((R, Db), dtype=torch.float32, device=dev)

    def f():
        ops.gemm(dl, Wt, slabs, out_mode=3, splitk=sk)
        ops.sum_slabs(slabs, dyn)
    us = run(f)
    slabs.fill_(float("nan"))
    f()
    err = ((dyn - ref).abs().max() / ref.abs().max()).item()
    print(f"slabs {sk:3d}: {us:8.1f} us  {2 * R * K * Db / us / 1e6:7.1f} TFLOP/s  err {err:.2e}", flush=True)
