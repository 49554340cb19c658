"""NT GEMM timing with the real A operand and with every A row identical (lda = 0: A always cache-hot),
used for the ablations in profiles/r01_nt_v3_ablation.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dino_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in [(100864, 384, 1536), (100864, 1536, 384), (100864, 1152, 384)]:
    A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    B = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm(A, B, out))
    try:
        t1 = timeit(lambda: ops.gemm(A[:256], B, out, M=M, lda=0))
    except Exception as e:
        t1 = float("nan"); print("lda0 failed", e)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: normal {t0:.1f} us {fl/t0/1e6:.0f} TF | A stride 0 {t1:.1f} us {fl/t1/1e6:.0f} TF", flush=True)
