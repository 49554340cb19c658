import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_amd import ops
torch.manual_seed(0)
M, D = 16400 + 37, 384
dt = torch.bfloat16
a = torch.randn(M, D, device="cuda").to(dt)
wp = (0.05 * torch.randn(D, D, device="cuda")).to(dt)
bp = 0.1 * torch.randn(D, device="cuda")
res = (torch.randn(M, D, device="cuda") + 3 * torch.randn(M, 1, device="cuda")).to(dt)
x = torch.empty((M, D), dtype=dt, device="cuda")
nparts = ops.ln_stats_parts(D)
stats = torch.full((nparts * M * 2,), float("nan"), device="cuda")
ops.gemm(a, wp, x, bias=bp, residual=res, ln_stats=stats)
xf = a.float() @ wp.float().t() + bp + res.float()
st = stats.view(nparts, M, 2)
for q in range(nparts):
    ref = xf[:, 96 * q:96 * q + 96].sum(-1)
    d = (st[q, :, 0] - ref).abs()
    print("part", q, "max abs err", d.max().item(), "nan", torch.isnan(st[q]).sum().item(), "argmax row", d.argmax().item())
    # which rows are wrong
    bad = (d > 0.05).nonzero().flatten()
    print("  bad rows:", bad.numel(), bad[:20].tolist())
    if bad.numel():
        r = bad[0].item()
        # does it match a sum of other columns / rows?
        for q2 in range(nparts):
            print("   row", r, "kernel", st[q, r, 0].item(), "ref part", q2, xf[r, 96 * q2:96 * q2 + 96].sum().item())
        for dr in (-16, -4, -1, 1, 4, 16):
            if 0 <= r + dr < M:
                print("   neighbour", dr, xf[r + dr, 96 * q:96 * q + 96].sum().item())
