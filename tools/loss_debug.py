import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_amd import ops
torch.manual_seed(0)
K, B, nc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = (2 * torch.randn(nc * B, K, device="cuda")).bfloat16()
t = (2 * torch.randn(2 * B, K, device="cuda")).bfloat16()
c = 0.5 * torch.randn(K, device="cuda")
acc = torch.zeros(K, device="cuda")
ds, parts = ops.loss_fwd_bwd(s, t, c, acc, B, nc, 0.1, 0.04)
sd, td, cd = s.double(), t.double(), c.double()
q = torch.softmax((td - cd) / 0.04, -1)
lp = torch.log_softmax(sd / 0.1, -1)
T = 2 * nc - 2
ref = torch.zeros(nc * B, dtype=torch.float64, device="cuda")
for v in range(nc):
    for iq in range(2):
        if iq == v:
            continue
        ref[v * B:(v + 1) * B] += -(q[iq * B:(iq + 1) * B] * lp[v * B:(v + 1) * B]).sum(-1) / (B * T)
print("parts", parts[:8].tolist())
print("ref  ", ref[:8].tolist())
print("max rel per-row", ((parts.double() - ref).abs() / ref.abs()).max().item(), "sum", parts.sum().item(), ref.sum().item())

dbg = int(os.environ.get("DINO_LOSS_DBG", "0"))
x = sd / 0.1
lse = torch.logsumexp(x, -1)
qs = torch.zeros_like(x)
for v in range(nc):
    for iq in range(2):
        if iq != v:
            qs[v * B:(v + 1) * B] += q[iq * B:(iq + 1) * B]
want = {1: lse, 2: (qs * x).sum(-1), 3: qs.sum(-1), 4: sd.max(-1).values}.get(dbg)
if want is not None:
    print("dbg", dbg, parts[:6].tolist(), want[:6].tolist())
g = torch.autograd.grad(ref.sum(), [], allow_unused=True) if False else None
xr = sd.clone().requires_grad_(True)
lp2 = torch.log_softmax(xr / 0.1, -1)
tot = 0
for v in range(nc):
    for iq in range(2):
        if iq != v:
            tot = tot + (-(q[iq * B:(iq + 1) * B] * lp2[v * B:(v + 1) * B]).sum(-1) / (B * T)).sum()
tot.backward()
err = (ds.double() - xr.grad).abs().amax(-1) / xr.grad.abs().amax(-1)
print("grad rel err per row", err.tolist())
