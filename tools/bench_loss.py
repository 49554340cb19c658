"""Times dino_loss_fwd_bwd alone at BASELINE config #2's shapes (student [3072, 65536] bf16, teacher
[512, 65536]).  Usage: python tools/bench_loss.py [B] [K] [ncrops]; run under rocprofv3 --kernel-trace --stats
for the per-kernel split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
with_colsum = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0   # 0: the fused kernel alone (as in the step,
#                                          where the teacher's prototype GEMM has already left the column sums)
torch.manual_seed(0)
s = torch.randn(nc * B, K, device="cuda").bfloat16()
t = torch.randn(2 * B, K, device="cuda").bfloat16()
c = torch.zeros(K, device="cuda")
acc = torch.zeros(K, device="cuda") if with_colsum else None
ds = torch.empty_like(s)
for _ in range(3):
    ops.loss_fwd_bwd(s, t, c, acc, B, nc, 0.1, 0.04, dstudent=ds)
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.loss_fwd_bwd(s, t, c, acc, B, nc, 0.1, 0.04, dstudent=ds)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / n
byt = (2 * nc * B + 2 * B) * K * 2 + 2 * K * 4
print(f"loss fwd+bwd B={B} K={K} ncrops={nc}: {us:.1f} us per call, {byt / us / 1e6:.2f} TB/s algorithmic "
      f"({byt / us / 1e6 / 8:.3f} of 8 TB/s)")
