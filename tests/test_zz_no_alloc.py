"""dino_layernorm_bwd_static -- the allocation-free entry point (column-sum partials in the
library's own device array) -- against dino_layernorm_bwd with a caller workspace, through the C
ABI: two calls in a row, the second on a smaller input, then an over-size request."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _case(rows, D, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(rows, D, generator=g).to(DEV, dtype)
    dy = torch.randn(rows, D, generator=g).to(DEV, dtype)
    dres = torch.randn(rows, D, generator=g).to(DEV, dtype)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    rowscale = (torch.rand(rows // 37 + 1, generator=g) + 0.5).to(DEV)
    return x, dy, dres, gamma, rowscale


def _run(fn_static, rows, D, dtype, seed):
    from dino_amd import ops
    x, dy, dres, gamma, rowscale = _case(rows, D, dtype, seed)
    beta = torch.zeros(D, device=DEV)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    outs = []
    for static in (False, True):
        dgamma = torch.full((D,), 0.25, device=DEV)     # the sums are ACCUMULATED into these
        dbeta = torch.full((D,), -0.5, device=DEV)
        dxsum = torch.zeros(D, device=DEV)
        dx = torch.empty_like(x)
        kw = dict(dres=dres, dx=dx, dgamma=dgamma, dbeta=dbeta, dx_colsum=dxsum, rowscale=rowscale,
                  rows_per_group=37)
        if static:
            before = torch.cuda.memory_allocated()
            ops.layernorm_bwd_static(dy, x, gamma, mean, rstd, **kw)
            assert torch.cuda.memory_allocated() == before   # nothing allocated through torch either
        else:
            ws = torch.empty(1024 * 3 * D * 4, dtype=torch.uint8, device=DEV)
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, workspace=ws, dx_stride=D, **kw)
        torch.cuda.synchronize()
        outs.append((dx, dgamma, dbeta, dxsum))
    (dx0, *sums0), (dx1, *sums1) = outs
    assert torch.equal(dx0, dx1)                          # dx does not depend on the partial buffer
    # ... and against a plain fp32 torch LayerNorm backward of the same op (not only against the sibling
    # entry point): dx = LN'(dy) + dres; dgamma / dbeta accumulated; dx_colsum += sum_rows rowscale * dx
    xf = x.float().detach().requires_grad_(True)
    gf = gamma.detach().clone().requires_grad_(True)
    bf = beta.detach().clone().requires_grad_(True)
    rs = rowscale.repeat_interleave(37)[:rows, None]
    y = torch.nn.functional.layer_norm(xf, (D,), gf, bf, 1e-6)
    y.backward(dy.float())
    dx_ref = xf.grad + dres.float()
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4

    def close(a, b, t):
        return (a.float() - b).abs().max().item() <= t * (b.abs().max().item() + 1e-30)
    assert close(dx1, dx_ref, tol)
    assert close(sums1[0] - 0.25, gf.grad, tol)
    assert close(sums1[1] + 0.5, bf.grad, tol)
    assert close(sums1[2], (dx_ref * rs).sum(0), tol)
    for a, b in zip(sums0, sums1):                        # the reduce adds its 64 slices with float
        scale = b.abs().max().item() + 1e-30              # atomics: order-dependent in the last bits
        assert (a - b).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_static_layernorm_bwd_matches_workspace_variant(dtype):
    # large first (1024 workgroups of partials), then a smaller input and width: the second call must
    # not see anything of what the first left in the shared array
    _run(None, 50_000, 384, dtype, 1)
    _run(None, 1_500, 192, dtype, 2)
    _run(None, 4_099, 1024, dtype, 3)


def test_static_layernorm_bwd_rejects_oversize():
    from dino_amd import ops, _lib
    D = 2048
    x = torch.randn(64, D, device=DEV)
    z = torch.zeros(D, device=DEV)
    m = torch.zeros(64, device=DEV)
    with pytest.raises(_lib.DinoHipError, match="capacity"):
        ops.layernorm_bwd_static(x, x, z, m, m, dx=torch.empty_like(x), dgamma=z.clone(), dbeta=z.clone())
