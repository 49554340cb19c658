"""Per-kernel parity tests (GPU): every C-ABI entry point against a torch fp32/fp64 statement of
the same op (or the oracle for attention / loss).  Tolerances: fp32 <= 1e-4 relative (the
north-star bound), bf16 <= 2e-2 of the tensor's max (bf16 has 8 mantissa bits)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dino_oracle as O  # noqa: E402
from oracle.detfill import det_normal  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 2e-2


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rnd(shape, tag, std=1.0, dtype=torch.float32):
    return det_normal(shape, tag, std=std).to(DEV).to(dtype)


@pytest.fixture(scope="module")
def ops():
    from dino_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 200, 192), (512, 384, 384), (128, 128, 64), (77, 40, 8)])
def test_gemm_nt_plain(ops, dtype, M, N, K):
    A, B = rnd((M, K), "gA", dtype=dtype), rnd((N, K), "gB", dtype=dtype)
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out)
    ref = A.float() @ B.float().t()
    assert rel(out, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_asymmetric_identity(ops, dtype):
    """A = I with an asymmetric B catches a row<->col swapped C write."""
    K = 64
    A = torch.eye(K, device=DEV, dtype=dtype)
    B = (torch.arange(40 * K, device=DEV, dtype=torch.float32).reshape(40, K) % 13 - 6).to(dtype)
    out = torch.empty((K, 40), dtype=dtype, device=DEV)
    ops.gemm(A, B, out)
    assert torch.equal(out.float(), B.float().t())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(ops, dtype):
    M, N, K, G = 260, 136, 128, 13  # 20 groups of 13 rows
    A, B = rnd((M, K), "eA", dtype=dtype), rnd((N, K), "eB", std=0.1, dtype=dtype)
    bias = rnd((N,), "eb")
    res = rnd((M, N), "er", dtype=dtype)
    rs = (det_normal((M // G,), "ers").abs() + 0.5).to(DEV)
    rs[3] = 0.0
    rsr = rs.repeat_interleave(G)[:, None]
    acc = A.float() @ B.float().t()
    # bias + residual with row-scaled bias (proj / fc2 with drop-path)
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out, bias=bias, residual=res, rowscale=rs, rows_per_group=G, bias_rowscale=True)
    assert rel(out, acc + bias * rsr + res.float()) < tol(dtype)
    # bias + gelu with saved pre-activation and row-scaled output (fc1)
    aux = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out, bias=bias, act=1, aux=aux, rowscale=rs, rows_per_group=G, out_rowscale=True)
    pre = acc + bias
    assert rel(aux, pre) < tol(dtype)
    assert rel(out, F.gelu(aux.float()) * rsr) < tol(dtype)
    # gelu backward multiply (fc2 dgrad)
    pre_t = rnd((M, N), "ep", dtype=dtype)
    ops.gemm(A, B, out, act=2, aux=pre_t, rowscale=rs, rows_per_group=G, out_rowscale=True)
    x = pre_t.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert rel(out, acc * x.grad * rsr) < tol(dtype)
    # fused column sums of the written values (bias gradient of the producing layer)
    cs = torch.zeros(N, device=DEV)
    ops.gemm(A, B, out, act=2, aux=pre_t, rowscale=rs, rows_per_group=G, out_rowscale=True, colsum=cs)
    assert rel(cs, (acc * x.grad * rsr).sum(0)) < tol(dtype)
    # saved-derivative pair (act 3 / 4)
    ops.gemm(A, B, out, bias=bias, act=3, aux=aux)
    xp = (acc + bias).requires_grad_(True)
    F.gelu(xp).sum().backward()
    assert rel(out, F.gelu(acc + bias)) < tol(dtype)
    assert rel(aux, xp.grad) < tol(dtype)
    ops.gemm(A, B, out, act=4, aux=pre_t)
    assert rel(out, acc * pre_t.float()) < tol(dtype)
    # fp32 output, then atomic accumulate on top, split-K
    o32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.gemm(A, B, o32, out_mode=1)
    assert rel(o32, acc) < tol(dtype)
    ops.gemm(A, B, o32, out_mode=2, splitk=2)
    assert rel(o32, 2 * acc) < tol(dtype)


@pytest.mark.parametrize("M,N,K,sk", [(300, 256, 4096, 8), (264, 136, 1184, 5), (64, 8, 96, 7)])
def test_gemm_splitk_slabs(ops, M, N, K, sk):
    """out_mode 3: every k-split stores its partial product in its own fp32 slab (slabs no split reaches are
    zeroed by the call), dino_sum_slabs adds them: the prototype layer's input gradient (K = out_dim deep)."""
    dtype = torch.bfloat16
    A, B = rnd((M, K), "sA", dtype=dtype), rnd((N, K), "sB", std=0.1, dtype=dtype)
    ref = A.float() @ B.float().t()
    slabs = torch.full((sk, M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(A, B, slabs, out_mode=3, splitk=sk)
    assert torch.isfinite(slabs).all()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.sum_slabs(slabs, out)
    assert rel(out, ref) < tol(dtype)
    assert rel(out, slabs.sum(0)) < 1e-6
    ops.sum_slabs(slabs, out, accumulate=True)
    assert rel(out, 2 * ref) < tol(dtype)
    with pytest.raises(Exception):   # fp32 operands have no slab kernel: refused, not silently atomics
        ops.gemm(A.float(), B.float(), slabs, out_mode=3, splitk=sk)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_row_map(ops, dtype):
    """Patch-embed epilogue: rows skip each crop's CLS slot and add pos[1 + m % n]."""
    Bc, n, N, K = 3, 9, 128, 192
    A, B = rnd((Bc * n, K), "rA", dtype=dtype), rnd((N, K), "rB", std=0.1, dtype=dtype)
    bias, pos = rnd((N,), "rb"), rnd((n + 1, N), "rp", dtype=dtype)
    out = torch.zeros((Bc * (n + 1), N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out, bias=bias, residual=pos, row_map_n=n)
    ref = (A.float() @ B.float().t() + bias).reshape(Bc, n, N) + pos[1:].float()
    got = out.reshape(Bc, n + 1, N)
    assert rel(got[:, 1:], ref) < tol(dtype)
    assert got[:, 0].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Kc,M,N", [(1000, 192, 136), (4096, 384, 128), (64, 8, 8)])
def test_gemm_tn(ops, dtype, Kc, M, N):
    """wgrad orientation: out[m,n] = sum_k A[k,m] B[k,n], fp32 atomic accumulate, split-K."""
    A, B = rnd((Kc, M), "tA", dtype=dtype), rnd((Kc, N), "tB", dtype=dtype)
    ref = A.float().t() @ B.float()
    out = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    ops.gemm(A, B, out, trans=True, out_mode=2, splitk=4)
    assert rel(out, ref) < tol(dtype)
    out2 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.gemm(A, B, out2, trans=True, out_mode=1)
    assert rel(out2, ref) < tol(dtype)


@pytest.mark.parametrize("Kc,M,N", [(2048, 256, 384), (4096, 384, 256), (1024, 1152, 384), (512, 384, 1536),
                                    (2048, 1024, 1024), (1024, 4096, 1024), (4096, 1024, 4096), (512, 128, 256),
                                    (1000, 384, 384), (2000 + 24, 128, 768), (7880, 1024, 256), (985, 384, 1152), (4099, 128, 256)])   # ragged token counts
@pytest.mark.parametrize("ws", [True, False])
def test_gemm_tn_dma_kernel(ops, Kc, M, N, ws):
    """bf16 wgrad shapes served by the LDS-DMA TN kernel (128x384 / 384x128 tiles), with the
    workspace-slab reduction and with the atomic fallback; accumulates on top of `out`."""
    dtype = torch.bfloat16
    A, B = rnd((Kc, M), "dA", dtype=dtype), rnd((Kc, N), "dB", dtype=dtype)
    ref = A.float().t() @ B.float()
    out = torch.ones((M, N), dtype=torch.float32, device=DEV)
    wsb = torch.empty(64 << 20, dtype=torch.uint8, device=DEV) if ws else None
    ops.gemm(A, B, out, trans=True, out_mode=2, splitk=4, workspace=wsb)
    assert rel(out, ref + 1.0) < tol(dtype)
    ops.gemm(A, B, out, trans=True, out_mode=2, splitk=4, workspace=wsb)
    assert rel(out, 2 * ref + 1.0) < tol(dtype)


_TN_V3 = """
import torch
from dino_amd import ops
torch.manual_seed(0)
dev, dt = "cuda", torch.bfloat16
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()
for K, M, N, use_ws in [(1024, 1152, 384, True), (512, 384, 1536, True), (1000, 384, 384, False), (985, 192, 768, True),
                        (4099, 1536, 384, True), (40, 192, 384, True)]:
    A, B = torch.randn(K, M, device=dev).to(dt), torch.randn(K, N, device=dev).to(dt)
    out = torch.ones(M, N, device=dev)
    ops.gemm(A, B, out, trans=True, out_mode=2, splitk=4, workspace=ws if use_ws else None)
    ops.gemm(A, B, out, trans=True, out_mode=2, splitk=4, workspace=ws if use_ws else None)
    err = rel(out, 2 * (A.float().t() @ B.float()) + 1.0)
    assert err < 1e-2, (K, M, N, err)
for tokens in (2048, 9 * 1024 + 40, 5 * 197 * 3):
    calls, refs = [], []
    for j, (Mo, No) in enumerate([(384, 1536), (1536, 384), (384, 384), (1152, 384)]):
        A, B = torch.randn(tokens, Mo, device=dev).to(dt), torch.randn(tokens, No, device=dev).to(dt)
        out = torch.full((Mo, No), float(j), device=dev)
        refs.append(A.float().t() @ B.float() + float(j))
        calls.append((A, B, out, dict(trans=True, out_mode=2, splitk=4, workspace=ws)))
    cs = torch.full((1152,), 0.5, device=dev)
    calls[3][3]["colsum"] = cs
    ops.gemm_group(calls)
    for c, ref in zip(calls, refs):
        assert rel(c[2], ref) < 1e-2, tokens
    assert rel(cs, calls[3][0].float().sum(0) + 0.5) < 1e-2, tokens
print("wide-tile weight-gradient kernel ok")
"""


def test_gemm_tn_eight_wave_kernel_behind_env_switch():
    """DINO_TN_V3=1 (read once per process, hence the child process): gemm_tn_bf16_v3 -- 192 x 384 output tiles on eight
    waves that request their own stages; measured slower than the shipped 128 x 384 kernel and therefore opt-in --
    single and grouped launches, slab and atomic reductions, ragged token counts, the fused bias-gradient sums."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DINO_TN_V3="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _TN_V3], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "kernel ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("tokens", [2048, 9 * 1024 + 32, 9 * 1024 + 40, 5 * 197 * 3])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_group_matches_separate_calls(ops, dtype, tokens):
    """The four weight gradients of a transformer block as one dino_gemm_group call: bf16 takes the
    grouped split-over-tokens launch (one slab round for all), fp32 falls back to four dino_gemm
    calls -- either way the result is that of four separate calls, accumulated on top of `out`."""
    shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]      # fc2, fc1, proj, qkv of ViT-S
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    calls, refs, outs = [], [], []
    for j, (Mo, No) in enumerate(shapes):
        A, B = rnd((tokens, Mo), f"gA{j}", dtype=dtype), rnd((tokens, No), f"gB{j}", dtype=dtype)
        out = torch.full((Mo, No), float(j), dtype=torch.float32, device=DEV)
        refs.append(A.float().t() @ B.float() + float(j))
        outs.append(out)
        calls.append((A, B, out, dict(trans=True, out_mode=2, splitk=4, workspace=ws)))
    # the bias gradient that goes with a weight gradient: colsum[m] += sum_k A[k, m]
    cs = torch.full((1152,), 0.5, dtype=torch.float32, device=DEV)
    calls[3][3]["colsum"] = cs
    ops.gemm_group(calls)
    for out, ref in zip(outs, refs):
        assert rel(out, ref) < tol(dtype)
    assert rel(cs, calls[3][0].float().sum(0) + 0.5) < tol(dtype)
    cs1 = torch.zeros((1152,), dtype=torch.float32, device=DEV)
    ops.gemm(calls[3][0], calls[3][1], torch.zeros_like(outs[3]), trans=True, out_mode=2, splitk=4,
             workspace=ws, colsum=cs1)                       # single call, same semantics
    assert rel(cs1, calls[3][0].float().sum(0)) < tol(dtype)
    del calls[3][3]["colsum"]
    sep = [torch.full_like(o, float(j)) for j, o in enumerate(outs)]
    for (A, B, _, kw), o in zip(calls, sep):
        ops.gemm(A, B, o, **kw)
    for a, b in zip(outs, sep):
        assert rel(a, b) < 1e-5
    # a group the fused kernel cannot take (mixed token counts) still gives the right answer
    A2, B2 = rnd((1024, 384), "gA9", dtype=dtype), rnd((1024, 384), "gB9", dtype=dtype)
    o2 = torch.zeros((384, 384), dtype=torch.float32, device=DEV)
    o0 = torch.zeros_like(outs[2])
    ops.gemm_group([(A2, B2, o2, dict(trans=True, out_mode=2, splitk=4, workspace=ws)),
                    (calls[2][0], calls[2][1], o0, calls[2][3])])
    assert rel(o2, A2.float().t() @ B2.float()) < tol(dtype)
    assert rel(o0, refs[2] - 2.0) < tol(dtype)


@pytest.mark.parametrize("M,N,K", [(256 * 131 - 40, 384, 384), (256 * 300, 192, 448), (256 * 70, 768, 512), (256 * 90 - 8, 576, 128),
                                   (256 * 40 - 16, 1024, 128), (256 * 20 + 8, 4096, 64), (256 * 100, 328, 64)])   # ragged last column tile
def test_gemm_nt_wide_tile_kernel(ops, M, N, K):
    """bf16 shapes served by the 512-thread 256x192 kernel (N % 192 == 0, K % 64 == 0, at least
    one tile per CU): plain, and every epilogue the ViT schedule uses, ragged last row tile."""
    dtype = torch.bfloat16
    G = 8
    A, B = rnd((M, K), "wA", dtype=dtype), rnd((N, K), "wB", std=0.1, dtype=dtype)
    acc = A.float() @ B.float().t()
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out)
    assert rel(out, acc) < 1e-2
    # exact placement: small-integer operands make every product and sum exact in bf16/fp32
    Ai = (torch.arange(M * K, device=DEV).reshape(M, K) % 7 - 3).to(dtype)
    Bi = (torch.arange(N * K, device=DEV).reshape(N, K) % 5 - 2).to(dtype)
    ops.gemm(Ai, Bi, out)
    assert torch.equal(out.float(), (Ai.float() @ Bi.float().t()).to(dtype).float())
    bias = rnd((N,), "wb")
    res = rnd((M, N), "wr", dtype=dtype)
    rs = (det_normal((M // G,), "wrs").abs() + 0.5).to(DEV)
    rs[3] = 0.0
    rsr = rs.repeat_interleave(G)[:, None]
    ops.gemm(A, B, out, bias=bias, residual=res, rowscale=rs, rows_per_group=G, bias_rowscale=True)
    assert rel(out, acc + bias * rsr + res.float()) < 1e-2
    # residual rows that repeat every 197 rows (the position table under every crop, patch-embed epilogue)
    pos = rnd((197, N), "wp", dtype=dtype)
    ops.gemm(A, B, out, bias=bias, residual=pos, res_mod=197)
    assert rel(out, acc + bias + pos.float()[torch.arange(M, device=DEV) % 197]) < 1e-2
    aux = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out, bias=bias, act=1, aux=aux, rowscale=rs, rows_per_group=G, out_rowscale=True)
    assert rel(aux, acc + bias) < 1e-2
    assert rel(out, F.gelu(aux.float()) * rsr) < 1e-2
    pre_t = rnd((M, N), "wp", dtype=dtype)
    cs = torch.zeros(N, device=DEV)
    ops.gemm(A, B, out, act=2, aux=pre_t, rowscale=rs, rows_per_group=G, out_rowscale=True, colsum=cs)
    x = pre_t.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert rel(out, acc * x.grad * rsr) < 1e-2
    assert rel(cs, (acc * x.grad * rsr).sum(0)) < 1e-2
    # saved-derivative pair: act 3 stores gelu'(pre), act 4 multiplies by it
    ops.gemm(A, B, out, bias=bias, act=3, aux=aux)
    xp = (acc + bias).requires_grad_(True)
    F.gelu(xp).sum().backward()
    assert rel(out, F.gelu(acc + bias)) < 1e-2
    assert rel(aux, xp.grad) < 1e-2
    ops.gemm(A, B, out, act=4, aux=pre_t, colsum=cs.zero_())
    assert rel(out, acc * pre_t.float()) < 1e-2
    assert rel(cs, (acc * pre_t.float()).sum(0)) < 1e-2
    o32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.gemm(A, B, o32, out_mode=1)
    assert rel(o32, acc) < 1e-4


# ------------------------------------------------------------------ work queue of the wide-tile NT kernel
@pytest.fixture
def work_queue(ops):
    """Turns the wide-tile NT GEMM's work queue on for one test (it is off by default in a one-rank process)."""
    was = ops.gemm_work_queue(True)
    yield
    torch.cuda.synchronize()
    ops.gemm_work_queue(was)


@pytest.mark.parametrize("M,N,K", [(256 * 300 - 40, 384, 384),     # 600 items: the queue's steady state
                                   (256 * 90 - 8, 576, 128),       # two K-tiles per item (entries are looked up lazily)
                                   (256 * 131, 384, 192),          # 262 items: a handful drawn, then the queue is empty
                                   (256 * 197 + 100, 1536, 384),   # half tiles in the last round, 8 column tiles
                                   (256 * 150, 328, 256)])         # ragged last column tile
def test_gemm_work_queue_is_bit_identical_to_static_shares(ops, M, N, K):
    """dino_gemm_work_queue: which workgroup computes a tile changes, the tile's arithmetic does not -- plain, residual +
    LayerNorm statistics, GELU + saved derivative (with prescale) and x saved derivative must match the static
    launch bit for bit, and the static launch is checked against fp32 torch."""
    dt = torch.bfloat16
    A, B = rnd((M, K), "qA", dtype=dt), rnd((N, K), "qB", std=0.1, dtype=dt)
    bias, res = rnd((N,), "qb"), rnd((M, N), "qr", dtype=dt)
    ps = (det_normal((M,), "qps").abs() + 0.5).to(DEV)
    nparts = ops.ln_stats_parts(N)
    fold = N % 96 == 0 and N <= 384

    def run():
        o = [torch.empty((M, N), dtype=dt, device=DEV) for _ in range(5)]
        aux = torch.empty((M, N), dtype=dt, device=DEV)
        st = torch.zeros(nparts * M * 2, device=DEV)
        ops.gemm(A, B, o[0])
        ops.gemm(A, B, o[1], bias=bias, residual=res, ln_stats=st if fold else None)
        ops.gemm(A, B, o[2], bias=bias, act=3, aux=aux, prescale=ps)
        ops.gemm(A, B, o[3], act=4, aux=res)
        ops.gemm(A, B, o[4], bias=bias, act=1)
        return o + [aux, st]
    assert ops.gemm_work_queue() is False
    ref = run()
    acc = A.float() @ B.float().t()
    assert rel(ref[0], acc) < 1e-2 and rel(ref[1], acc + bias + res.float()) < 1e-2
    assert rel(ref[3], acc * res.float()) < 1e-2
    was = ops.gemm_work_queue(True)
    try:
        for _ in range(2):
            got = run()
            for a, b in zip(ref, got):
                assert torch.equal(a, b)
    finally:
        torch.cuda.synchronize()
        ops.gemm_work_queue(was)


def test_gemm_work_queue_many_launches_and_held_cus(ops, work_queue):
    """More launches than the library has queue slots (each slot is cleared by the last workgroup of the launch that
    used it), and launches that start while another stream's kernel holds CUs (dino_diag_occupy_cus: what RCCL's
    all-reduce does under backward): late workgroups find the queue empty, every tile is still computed once."""
    dt = torch.bfloat16
    M, N, K = 256 * 300, 192, 128
    A, B = rnd((M, K), "qmA", dtype=dt), rnd((N, K), "qmB", std=0.1, dtype=dt)
    want = torch.empty((M, N), dtype=dt, device=DEV)
    ops.gemm_work_queue(False)
    ops.gemm(A, B, want)
    ops.gemm_work_queue(True)
    out = torch.empty_like(want)
    for it in range(1100):
        if it % 100 == 0:
            out.zero_()
        ops.gemm(A, B, out)
        if it % 100 == 99:
            assert torch.equal(out, want), it
    side = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        e0.record()
        ops.occupy_cus(48, 3000, stream=side)
        e1.record()
    for _ in range(20):
        out.zero_()
        ops.gemm(A, B, out)
        assert torch.equal(out, want)          # (the comparison synchronises: the next launch meets the held CUs again)
    torch.cuda.synchronize()
    assert 2.9 < e0.elapsed_time(e1) < 30.0    # the occupying kernel really ran for its 3 ms, and drained


@pytest.mark.parametrize("tokens", [9 * 1024 + 40, 40 * 1024 + 8, 195584])
def test_gemm_group_two_rounds_with_the_work_queue(ops, work_queue, tokens):
    """With dino_gemm_work_queue on, the grouped weight-gradient launch splits the tokens over twice as many, half
    as long workgroups (two rounds on the chip instead of one: a held CU no longer puts a workgroup behind all the
    others).  Same sums, with the bias gradient, against fp32 torch -- also while CUs are held."""
    dt = torch.bfloat16
    shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]
    ws = torch.empty(128 << 20, dtype=torch.uint8, device=DEV)
    calls, refs, outs = [], [], []
    for j, (Mo, No) in enumerate(shapes):
        A, B = rnd((tokens, Mo), f"hA{j}", std=0.3, dtype=dt), rnd((tokens, No), f"hB{j}", std=0.3, dtype=dt)
        out = torch.full((Mo, No), float(j), dtype=torch.float32, device=DEV)
        refs.append(A.float().t() @ B.float() + float(j))
        outs.append(out)
        calls.append((A, B, out, dict(trans=True, out_mode=2, workspace=ws)))
    cs = torch.zeros((1152,), dtype=torch.float32, device=DEV)
    calls[3][3]["colsum"] = cs
    ops.gemm_group(calls)
    for out, ref in zip(outs, refs):
        assert rel(out, ref) < 1e-2
    assert rel(cs, calls[3][0].float().sum(0)) < 1e-2
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    ops.occupy_cus(24, 2000, stream=side)
    for j, out in enumerate(outs):
        out.fill_(float(j))
    ops.gemm_group(calls)
    for out, ref in zip(outs, refs):
        assert rel(out, ref) < 1e-2
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(256 * 140, 2304, 128), (256 * 130 + 24, 3072, 64), (256 * 66, 4032, 64),
                                   (2000, 4096, 128), (1500, 2304, 96), (700, 8192, 64)])
def test_gemm_nt_long_bias(ops, M, N, K):
    """Biases longer than 2048 columns (ViT-B qkv/fc1: 2304/3072, ViT-L fc1: 4096) stay on the
    LDS-DMA kernels: up to 4096 columns in LDS in the wide-tile kernel, read from memory in the
    256x128 kernel.  The bias is a ramp, so a mis-indexed column shows."""
    dtype = torch.bfloat16
    A, B = rnd((M, K), "lbA", dtype=dtype), rnd((N, K), "lbB", std=0.1, dtype=dtype)
    bias = (torch.arange(N, device=DEV, dtype=torch.float32) - N / 2) / 64.0
    acc = A.float() @ B.float().t()
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out, bias=bias)
    assert rel(out, acc + bias) < 1e-2
    assert ((out.float() - (acc + bias)).abs().amax(0) < 0.3).all()      # every column got ITS bias
    aux = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(A, B, out, bias=bias, act=3, aux=aux)
    assert rel(out, F.gelu(acc + bias)) < 1e-2
    if N <= 4032:                          # long bias + column sums: the 256x128 kernel
        cs = torch.zeros(N, device=DEV)
        ops.gemm(A, B, out, bias=bias, colsum=cs)
        assert rel(out, acc + bias) < 1e-2
        assert rel(cs, (acc + bias).sum(0)) < 1e-2


def test_gemm_rejects_bad_args(ops):
    from dino_amd._lib import DinoHipError
    A = torch.zeros((16, 12), device=DEV)  # K not a multiple of 8
    B = torch.zeros((16, 12), device=DEV)
    with pytest.raises((DinoHipError, AssertionError)):
        ops.gemm(A, B, torch.empty((16, 16), device=DEV))
    with pytest.raises(DinoHipError):
        ops.gemm(torch.zeros((8, 8)), torch.zeros((8, 8)), torch.zeros((8, 8)))  # CPU tensors


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [64, 384, 768, 1024])
def test_layernorm(ops, dtype, D):
    rows = 333
    x = rnd((rows, D), "lx", std=2.0, dtype=dtype)
    g, b = rnd((D,), "lg") + 1.0, rnd((D,), "lb")
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), gr, br, 1e-6)
    assert rel(y, yr) < tol(dtype)
    dy = rnd((rows, D), "ldy", dtype=dtype)
    dres = rnd((rows, D), "ldr", dtype=dtype)
    yr.backward(dy.float())
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dx = ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg, dbeta=db)
    assert rel(dx, xr.grad + dres.float()) < tol(dtype)
    assert rel(dg, gr.grad) < tol(dtype)
    assert rel(db, br.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_ws", [False, True])
def test_layernorm_bwd_column_sums(ops, dtype, use_ws):
    """dgamma / dbeta / scaled column sums of dx, through the float atomics and through the
    per-workgroup partials + reduce kernel (workspace); both accumulate on top of the targets."""
    rows, D, G = 5000, 384, 10
    x = rnd((rows, D), "cx", std=2.0, dtype=dtype)
    g, b = rnd((D,), "cg") + 1.0, rnd((D,), "cb")
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    dy, dres = rnd((rows, D), "cdy", dtype=dtype), rnd((rows, D), "cdr", dtype=dtype)
    rs = (det_normal((rows // G,), "crs").abs() + 0.5).to(DEV)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy.float())
    dg, db, cs = torch.ones(D, device=DEV), torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    ws = torch.empty(8 << 20, dtype=torch.uint8, device=DEV) if use_ws else None
    dx = ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg, dbeta=db, dx_colsum=cs,
                           rowscale=rs, rows_per_group=G, workspace=ws)
    ref_dx = xr.grad + dres.float()
    assert rel(dx, ref_dx) < tol(dtype)
    assert rel(dg - 1, gr.grad) < tol(dtype)
    assert rel(db - 1, br.grad) < tol(dtype)
    assert rel(cs - 1, (dx.float() * rs.repeat_interleave(G)[:, None]).sum(0)) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_strided_cls(ops, dtype):
    """Final norm: only the CLS row of each crop (row stride N*D)."""
    Bc, N, D = 5, 7, 128
    x = rnd((Bc * N, D), "sx", dtype=dtype)
    g, b = rnd((D,), "sg") + 1.0, rnd((D,), "sb")
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, rows=Bc, x_stride=N * D)
    ref = F.layer_norm(x.float().reshape(Bc, N, D)[:, 0], (D,), g, b, 1e-6)
    assert rel(y, ref) < tol(dtype)
    dy = rnd((Bc, D), "sdy", dtype=dtype)
    dx = torch.zeros((Bc * N, D), dtype=dtype, device=DEV)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dx=dx, rows=Bc, x_stride=N * D, dx_stride=N * D)
    xr = x.float().reshape(Bc, N, D).requires_grad_(True)
    F.layer_norm(xr[:, 0], (D,), g, b, 1e-6).backward(dy.float())
    assert rel(dx.reshape(Bc, N, D), xr.grad) < tol(dtype)


# ------------------------------------------------------------------ attention
def _attn_ref(qkv, Bc, N, H, rowscale):
    x = qkv.double().reshape(Bc, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-2, -1)) * 64 ** -0.5
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(Bc, N, H * 64)
    if rowscale is not None:
        o = o * rowscale.double()[:, None, None]
    return o.reshape(Bc * N, H * 64), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,Bc,H", [(37, 5, 2), (197, 2, 3), (10, 9, 1), (50, 3, 2), (64, 2, 2), (257, 1, 2), (785, 1, 2), (330, 2, 1)])
@pytest.mark.parametrize("use_rs", [False, True])
def test_attention(ops, dtype, N, Bc, H, use_rs):
    qkv = rnd((Bc * N, 3 * H * 64), f"aq{N}", dtype=dtype)
    rs = None
    if use_rs:
        rs = (det_normal((Bc,), "ars").abs() + 0.5).to(DEV)
        rs[0] = 0.0
    out, lse = ops.attn_fwd(qkv, Bc, N, H, rowscale=rs)
    qr = qkv.double().requires_grad_(True)
    oref, lref = _attn_ref(qr, Bc, N, H, rs)
    assert rel(out, oref) < tol(dtype)
    assert (lse.double() - lref).abs().max().item() < (1e-4 if dtype == torch.float32 else 3e-2)
    dout = rnd((Bc * N, H * 64), f"ado{N}", dtype=dtype)
    # the kernel contract: dout already carries the per-crop drop-path factor
    dout_eff = dout if rs is None else (dout.float().reshape(Bc, N, -1) * rs[:, None, None]).reshape(
        Bc * N, -1).to(dtype)
    dqkv = ops.attn_bwd(qkv, out, dout_eff, lse, Bc, N, H, rowscale=rs)
    oref.backward(dout.double())
    assert rel(dqkv, qr.grad) < (2e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,Bc,H", [(197, 5, 3), (37, 9, 6), (785, 2, 2), (10, 3, 1), (257, 3, 16)])
def test_attention_cls_query_only(ops, dtype, N, Bc, H):
    """dino_attn_cls_fwd / _bwd (the last block's attention: only x[:, 0] leaves the backbone, vision_transformer.py:382-383)
    against row 0 of the full attention kernels and against the fp64 reference whose output gradient is zero outside the
    CLS rows: same CLS output and log-sum-exp, same dqkv everywhere (dq zero outside the CLS rows)."""
    D = H * 64
    qkv = rnd((Bc * N, 3 * D), f"cq{N}", dtype=dtype)
    rs = (det_normal((Bc,), "crs").abs() + 0.5).to(DEV)
    rs[Bc // 2] = 0.0
    out_full, lse_full = ops.attn_fwd(qkv, Bc, N, H, rowscale=rs)
    oc, lc = ops.attn_cls_fwd(qkv, Bc, N, H, rowscale=rs)
    assert rel(oc, out_full.view(Bc, N, D)[:, 0]) < tol(dtype)
    assert (lc - lse_full[:, :, 0]).abs().max().item() < (1e-4 if dtype == torch.float32 else 3e-2)
    dcls = rnd((Bc, D), f"cdo{N}", dtype=dtype)
    dcls_eff = (dcls.float() * rs[:, None]).to(dtype)              # the kernels' contract: dout carries the factor
    dfull = torch.zeros((Bc * N, D), dtype=dtype, device=DEV)
    dfull.view(Bc, N, D)[:, 0] = dcls_eff
    d_ref = ops.attn_bwd(qkv, out_full, dfull, lse_full, Bc, N, H, rowscale=rs)
    d_cls = ops.attn_cls_bwd(qkv, dcls_eff, lc, Bc, N, H)
    assert rel(d_cls, d_ref) < (2e-4 if dtype == torch.float32 else 3e-2)
    assert d_cls.view(Bc, N, 3 * D)[:, 1:, :D].abs().max().item() == 0.0 if N > 1 else True
    qr = qkv.double().requires_grad_(True)
    oref, _ = _attn_ref(qr, Bc, N, H, rs)
    oref.view(Bc, N, D)[:, 0].backward(dcls.double())
    assert rel(oc, oref.view(Bc, N, D)[:, 0]) < tol(dtype)
    assert rel(d_cls, qr.grad) < (2e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("N,Bc,H", [(197, 50, 6),   # 300 problems: more than one per persistent workgroup (next problem's operands in registers)
                                    (208, 2, 2), (193, 1, 3), (192, 2, 1), (177, 1, 2), (145, 3, 2), (100, 2, 2), (65, 2, 1),
                                    (48, 3, 2), (33, 2, 3), (32, 2, 2), (17, 3, 1), (16, 2, 2), (5, 4, 1), (37, 700, 3)])
def test_attention_backward_fused_shapes(ops, N, Bc, H):
    """bf16 backward through attn_bwd_fused_kernel (N <= 208; one wave per problem for N <= 48): every tile count
    parity / last-step form, full and ragged last tiles, and problem counts beyond one per workgroup; checked against
    the fp64 reference (small cases) and against the fp32 instantiation of the backward, which runs the older dq + dkv
    kernels on the same values."""
    dtype = torch.bfloat16
    qkv = rnd((Bc * N, 3 * H * 64), f"fq{N}", dtype=dtype)
    rs = (det_normal((Bc,), "frs").abs() + 0.5).to(DEV)
    rs[Bc // 2] = 0.0
    out, lse = ops.attn_fwd(qkv, Bc, N, H, rowscale=rs)
    dout = rnd((Bc * N, H * 64), f"fdo{N}", dtype=dtype)
    dout_eff = (dout.float().reshape(Bc, N, -1) * rs[:, None, None]).reshape(Bc * N, -1).to(dtype)
    dqkv = ops.attn_bwd(qkv, out, dout_eff, lse, Bc, N, H, rowscale=rs)
    assert torch.isfinite(dqkv.float()).all()
    # fp32 instantiation (dq + dkv kernels) on the same bf16 values: same math, fp32 storage
    q32 = qkv.float()
    o32, l32 = ops.attn_fwd(q32, Bc, N, H, rowscale=rs)
    d32 = ops.attn_bwd(q32, o32, dout_eff.float(), l32, Bc, N, H, rowscale=rs)
    assert rel(dqkv, d32) < 2e-2
    if Bc * H <= 64:
        qr = qkv.double().requires_grad_(True)
        oref, _ = _attn_ref(qr, Bc, N, H, rs)
        oref.backward(dout.double())
        assert rel(dqkv, qr.grad) < 3e-2


@pytest.mark.parametrize("N,Bc,H", [(785, 3, 2), (641, 2, 1), (768, 1, 2), (769, 1, 1), (897, 1, 2), (1025, 1, 1),
                                    (700, 2, 3), (785, 40, 6), (577, 2, 2), (384, 1, 3)])
def test_attention_long_sequences_streamed(ops, N, Bc, H):
    """bf16 sequences too long for a whole (crop, head) problem in LDS (ViT-S/8: 785 tokens; backward: from 384 tokens) run the streamed kernels
    (attn_fwd_stream / attn_bwd_dq_stream / attn_bwd_dkv_stream: 128 rows per workgroup, LDS-DMA double buffer):
    chunk counts 6-9, full and ragged last chunks (1 row, 17 rows, 128 rows), last workgroups with idle waves, and
    more workgroups than one round of the chip; against the fp64 reference (small cases) and against the fp32
    instantiation, which runs the older register-staged chunk kernels on the same values."""
    dtype = torch.bfloat16
    qkv = rnd((Bc * N, 3 * H * 64), f"lq{N}", dtype=dtype)
    rs = (det_normal((Bc,), "lrs").abs() + 0.5).to(DEV)
    if Bc > 1:
        rs[Bc // 2] = 0.0
    out, lse = ops.attn_fwd(qkv, Bc, N, H, rowscale=rs)
    dout = rnd((Bc * N, H * 64), f"ldo{N}", dtype=dtype)
    dout_eff = (dout.float().reshape(Bc, N, -1) * rs[:, None, None]).reshape(Bc * N, -1).to(dtype)
    dqkv = ops.attn_bwd(qkv, out, dout_eff, lse, Bc, N, H, rowscale=rs)
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
    q32 = qkv.float()
    o32, l32 = ops.attn_fwd(q32, Bc, N, H, rowscale=rs)
    d32 = ops.attn_bwd(q32, o32, dout_eff.float(), l32, Bc, N, H, rowscale=rs)
    assert rel(out, o32) < 2e-2
    assert (lse - l32).abs().max().item() < 3e-2
    assert rel(dqkv, d32) < 2e-2
    if Bc * H <= 8:
        qr = qkv.double().requires_grad_(True)
        oref, lref = _attn_ref(qr, Bc, N, H, rs)
        oref.backward(dout.double())
        assert rel(out, oref) < 2e-2
        assert (lse.double() - lref).abs().max().item() < 3e-2
        assert rel(dqkv, qr.grad) < 3e-2


_OLDER_ATTN_BWD = r"""
import torch
from dino_amd import ops
torch.manual_seed(0)
for N, Bc, H in [(37, 40, 3), (48, 3, 2), (197, 3, 2), (100, 2, 2)]:
    qkv = (torch.randn(Bc * N, 3 * H * 64, device="cuda") * 0.5).to(torch.bfloat16)
    do = (torch.randn(Bc * N, H * 64, device="cuda") * 0.5).to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv, Bc, N, H)
    d = ops.attn_bwd(qkv, o, do, lse, Bc, N, H)
    o32, l32 = ops.attn_fwd(qkv.float(), Bc, N, H)
    d32 = ops.attn_bwd(qkv.float(), o32, do.float(), l32, Bc, N, H)
    err = ((d.float() - d32).norm() / d32.norm()).item()
    assert err < 2e-2, (N, err)
print("older attention backward kernels ok")
"""


def test_attention_backward_older_kernels_behind_env_switch():
    """DINO_ATTN_FUSED=0 (read once per process, hence the child process): the two-kernel backward and the two-slot
    one-wave kernel (48-row images for N <= 48) that the fused kernels replaced stay correct."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DINO_ATTN_FUSED="0", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _OLDER_ATTN_BWD], cwd=root, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "kernels ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K,B,nc", [(512, 3, 5), (65536, 2, 12), (1024, 4, 2)])
def test_loss(ops, dtype, K, B, nc):
    s = rnd((nc * B, K), "ls", std=2.0, dtype=dtype)
    t = rnd((2 * B, K), "lt", std=2.0, dtype=dtype)
    c = rnd((K,), "lc", std=0.5)
    acc = torch.zeros(K, device=DEV)
    ds, parts = ops.loss_fwd_bwd(s, t, c, acc, B, nc, 0.1, 0.04)
    sr = s.double().requires_grad_(True)
    ref = O.dino_loss(sr, t.double(), c.double()[None], nc, 0.04)
    ref.backward()
    loss_out = torch.zeros(1, device=DEV)
    c2 = c.clone()
    ops.center_update(c2, acc.clone(), 0.9, 1.0 / (2 * B), parts, loss_out)
    assert abs(loss_out.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert rel(ds, sr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    cref = O.update_center(c.double()[None], t.double())[0]
    assert rel(c2, cref) < 1e-5


# ------------------------------------------------------------------ optimiser
def _chunk_tables(sizes):
    from dino_amd.engine.flat import build_chunks
    return build_chunks(sizes)


def test_adamw_ema_and_clip(ops):
    from dino_amd.engine.flat import FlatLayout
    shapes = {"a.weight": (300, 70), "a.bias": (70,), "b.weight": (9000,), "c.g": (33, 1)}
    lay = FlatLayout(shapes)
    n = lay.total
    p = det_normal((n,), "op").to(DEV)
    for name in shapes:  # keep padding zero
        pass
    g = det_normal((n,), "og", std=3.0).to(DEV)
    mask = torch.zeros(n, device=DEV)
    for name, (off, numel) in lay.slices.items():
        mask[off:off + numel] = 1
    p, g = p * mask, g * mask
    t = p.clone()
    m1, m2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    chunks, begin = lay.chunks.to(DEV), lay.tensor_chunk_begin.to(DEV)
    nt = len(shapes)
    sq = torch.empty(chunks.shape[0], device=DEV)
    norms, coef, stats = (torch.empty(nt, device=DEV), torch.empty(nt, device=DEV),
                          torch.empty(4, device=DEV))
    has = torch.tensor([1, 1, 1, 0], dtype=torch.int32, device=DEV)
    ops.grad_sqnorm(g, chunks, sq)
    ops.clip_coef(sq, begin, has, 3.0, norms, coef, stats)
    # reference
    ps = {k: p[o:o + c].clone() for k, (o, c) in lay.slices.items()}
    gs = {k: g[o:o + c].clone() for k, (o, c) in lay.slices.items()}
    ts = {k: v.clone() for k, v in ps.items()}
    st, nr = O.grad_stats_and_clip_([(k, gs[k] if k != "c.g" else None) for k in shapes], 3.0)
    for i, k in enumerate(shapes):
        if k != "c.g":
            assert abs(norms[i].item() - nr[k]) < 1e-4 * nr[k]
    assert abs(stats[0].item() - st["grad_norm"]) < 1e-4 * st["grad_norm"]
    assert abs(stats[1].item() - st["grad_max"]) < 1e-4 * st["grad_max"]
    assert stats[2].item() == round(st["clip_fraction"] * 3)
    hp = torch.zeros((nt, 8), device=DEV)
    lr, wd = 1e-3, 0.05
    for step in range(1, 3):
        for i, k in enumerate(shapes):
            active = k != "c.g"
            hp[i, :5] = torch.tensor([lr, wd if i != 1 else 0.0, 1 - 0.9 ** step, 1 - 0.999 ** step,
                                      1.0 if active else 0.0])
        psh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        tsh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        ops.adamw_ema(p, g, m1, m2, t, psh, tsh, chunks, hp, coef, 0.9, 0.999, 1e-8, 0.99, 0.01)
        states = getattr(test_adamw_ema_and_clip, "_st", {k: {} for k in shapes})
        test_adamw_ema_and_clip._st = states
        for i, k in enumerate(shapes):
            if k != "c.g":
                O.adamw_step_(ps[k], gs[k], states[k], lr, wd if i != 1 else 0.0)
        O.ema_update_(ts, ps, 0.99)
        for k, (o, c) in lay.slices.items():
            assert rel(p[o:o + c], ps[k]) < 1e-5, k
            assert rel(t[o:o + c], ts[k]) < 1e-5, k
            assert rel(psh[o:o + c], ps[k]) < 1e-2
            assert rel(tsh[o:o + c], ts[k]) < 1e-2
    del test_adamw_ema_and_clip._st
    # materialised clip (compute_gradient_stats semantics)
    g2 = g.clone()
    ops.grad_scale(g2, chunks, coef)
    for k, (o, c) in lay.slices.items():
        if k != "c.g":
            assert rel(g2[o:o + c], gs[k]) < 1e-5


def test_ema_cast_transpose(ops):
    n = 8 * 1000
    t, s = det_normal((n,), "et").to(DEV), det_normal((n,), "es").to(DEV)
    sh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ref = t * 0.996 + 0.004 * s
    ops.ema(t, s, sh, 0.996, 0.004)
    assert rel(t, ref) < 1e-6 and rel(sh, ref) < 1e-2
    ops.cast_bf16(s, sh)
    assert torch.equal(sh, s.to(torch.bfloat16))
    # matrices: element-wise form (odd sizes), wide form (64 x 64 tiles, 16-B accesses), wide form with ragged edge
    # tiles, and sizes the wide form would take at an offset it cannot (falls back)
    o1 = 70 * 33 + 6
    o2 = o1 + 128 * 64
    o3 = o2 + 100 * 72 + 2
    src = det_normal((o3 + 68 * 132 + 8,), "tr").to(DEV)
    for dt in DTYPES:
        dst = torch.zeros(src.numel(), dtype=dt, device=DEV)
        table = torch.tensor([[0, 0, 70, 33], [o1, o1, 128, 64], [o2, o2, 100, 72], [o3, o3, 68, 132]],
                             dtype=torch.int64, device=DEV)
        ops.transpose_batched(src, dst, table)
        for o, r, c in ((0, 70, 33), (o1, 128, 64), (o2, 100, 72), (o3, 68, 132)):
            assert torch.equal(dst[o:o + r * c].reshape(c, r), src[o:o + r * c].reshape(r, c).t().to(dt)), (o, r, c)


# ------------------------------------------------------------------ tokens
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("P,S", [(16, 96), (8, 24), (14, 96)])
def test_im2col_matches_conv(ops, dtype, P, S):
    Bc, D = 3, 32
    img = rnd((Bc, 3, S, S), "im")
    w = rnd((D, 3, P, P), "imw", std=0.05)
    cols = ops.im2col(img, P, dtype)
    K = 3 * P * P
    ref = F.conv2d(img, w, stride=P).flatten(2).transpose(1, 2).reshape(-1, D)
    got = cols.float()[:, :K] @ w.reshape(D, K).t()
    assert rel(got, ref) < (1e-5 if dtype == torch.float32 else 2e-2)
    if cols.shape[1] > K:
        assert cols[:, K:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_pos_table_and_bwd(ops, dtype):
    from dino_amd.utils.pos_interp import bicubic_taps
    D, src, dst = 64, 14, 6
    pos = rnd((1, src * src + 1, D), "pp")
    idx, w = bicubic_taps(src, dst)
    idx_d, w_d = torch.from_numpy(idx).to(DEV), torch.from_numpy(w).to(DEV)
    out = ops.pos_table(pos[0], idx_d, w_d, dst * dst, dtype)
    pr = pos.cpu().clone().requires_grad_(True)
    ref = O.interpolate_pos_encoding(pr, dst * dst, dst * 16, dst * 16, 16)[0]
    assert rel(out, ref.detach()) < (1e-5 if dtype == torch.float32 else 1e-2)
    ident = ops.pos_table(pos[0], None, None, src * src, dtype)
    assert torch.equal(ident, pos[0].to(dtype))
    dl = rnd((dst * dst + 1, D), "pdl")
    ref.backward(dl.cpu())
    dpos = torch.zeros((src * src + 1, D), device=DEV)
    ops.pos_table_bwd(dl, idx_d, w_d, dpos, dst * dst)
    assert rel(dpos, pr.grad[0]) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_cls_fold_colsum_scale(ops, dtype):
    Bc, N, D = 6, 10, 128
    tok = rnd((Bc * N, D), "ct", dtype=dtype)
    cls, pos0 = rnd((D,), "cc"), rnd((D,), "cp")
    t2 = tok.clone()
    ops.cls_rows(t2, cls, pos0, Bc, N)
    assert torch.equal(t2.reshape(Bc, N, D)[:, 1:], tok.reshape(Bc, N, D)[:, 1:])
    assert rel(t2.reshape(Bc, N, D)[:, 0], (cls + pos0).expand(Bc, D)) < tol(dtype)
    fo = torch.zeros((N, D), device=DEV)
    ops.fold_rows(tok, fo, Bc, N)
    assert rel(fo, tok.float().reshape(Bc, N, D).sum(0)) < 1e-5
    # many crops per split (four rows in flight per thread + a remainder), accumulating into a non-zero output
    Bc2, N2, D2 = 531, 37, 384
    tok2 = rnd((Bc2 * N2, D2), "ct2", dtype=dtype)
    fo2 = torch.ones((N2, D2), device=DEV)
    ops.fold_rows(tok2, fo2, Bc2, N2)
    assert rel(fo2, 1.0 + tok2.float().reshape(Bc2, N2, D2).sum(0)) < (1e-5 if dtype == torch.float32 else 1e-4)
    cs = torch.zeros(D, device=DEV)
    rs = (det_normal((Bc,), "crs").abs()).to(DEV)
    ops.colsum(tok, cs, rowscale=rs, rows_per_group=N)
    assert rel(cs, (tok.float().reshape(Bc, N, D) * rs[:, None, None]).sum((0, 1))) < 1e-5
    big = rnd((5000, 1536), "cbig", dtype=dtype)
    cb = torch.zeros(1536, device=DEV)
    ops.colsum(big, cb)
    assert rel(cb, big.float().sum(0)) < 1e-4
    x = tok.clone()
    ops.scale_by_scalar(x, torch.tensor([1.0], device=DEV))
    assert torch.equal(x, tok)
    ops.scale_by_scalar(x, torch.tensor([0.5], device=DEV))
    assert torch.equal(x.float(), tok.float() * 0.5)


# ------------------------------------------------------------------ head
@pytest.mark.parametrize("dtype", DTYPES)
def test_l2norm(ops, dtype):
    x = rnd((100, 256), "nx", dtype=dtype)
    y, nrm = ops.l2norm_fwd(x)
    xr = x.float().requires_grad_(True)
    yr = F.normalize(xr, dim=-1, p=2)
    assert rel(y, yr) < tol(dtype)
    dy = rnd((100, 256), "ndy")
    yr.backward(dy)
    dx = ops.l2norm_bwd(dy, y, nrm)
    assert rel(dx, xr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K,D", [(515, 64), (1000, 256), (4099, 256)])   # 256: the vectorised path of the bottleneck width, ragged last block
def test_weightnorm(ops, dtype, K, D):
    g, v = rnd((K, 1), "wg", std=0.1) + 1.0, rnd((K, D), "wv", std=0.02)
    w, wt, vn = ops.weightnorm_fwd(g, v, dtype)
    gr, vr = g.clone().requires_grad_(True), v.clone().requires_grad_(True)
    wr = O.weight_norm_weight(gr, vr)
    assert rel(w, wr) < tol(dtype)
    assert torch.equal(wt, w.t().contiguous())
    dW = rnd((K, D), "wdw")
    wr.backward(dW)
    dg, dv = torch.zeros_like(g), torch.zeros_like(v)
    ops.weightnorm_bwd(dW, g, v, vn, dg, dv)
    assert rel(dg, gr.grad) < 1e-4 and rel(dv, vr.grad) < 1e-4


def test_drop_path_scales_kernel(ops):
    """floor(keep + U) / keep per (block, branch, crop): values in {0, 1/keep}, kept fraction ~ keep, the same
    seed gives the same draw whatever the launch shape, the per-row expansion repeats each crop's value."""
    keep = torch.tensor([0.9, 0.75, 0.5], device=DEV)
    Bct = 4096
    s, rows = ops.drop_path_scales(1234, keep, Bct)
    assert rows is None and tuple(s.shape) == (3, 2, Bct)
    for j, k in enumerate([0.9, 0.75, 0.5]):
        v = s[j]
        kept = v > 0
        assert torch.allclose(v[kept], torch.full_like(v[kept], 1.0 / k))
        assert abs(kept.float().mean().item() - k) < 0.03, (k, kept.float().mean().item())
    assert not torch.equal(s[0, 0], s[0, 1])                     # branches draw independently
    s2, _ = ops.drop_path_scales(1234, keep, Bct)
    assert torch.equal(s, s2)
    s3, _ = ops.drop_path_scales(1235, keep, Bct)
    assert not torch.equal(s, s3)
    row_crop = torch.arange(Bct, device=DEV).repeat_interleave(3)[: 3 * Bct - 5].contiguous()
    s4, rows = ops.drop_path_scales(1234, keep, Bct, row_crop)
    assert torch.equal(s4, s) and torch.equal(rows, s[:, :, row_crop])


def test_gelu_table_epilogue_accuracy(ops):
    """The wide-tile kernel's GELU epilogue reads Phi / gelu' from an LDS table indexed by the bf16-rounded
    pre-activation (gemm_nt_v3.hip).  Pre-activations swept over [-9, 9] through a one-hot weight: gelu and
    gelu' within one bf16 rounding of the exact values plus the table's input rounding."""
    M, N, K = 4096, 1536, 64
    x = torch.linspace(-9.0, 9.0, M * K, device=DEV).reshape(M, K)
    x = x[torch.randperm(M, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))]
    A = x.to(torch.bfloat16).contiguous()
    B = torch.zeros((N, K), dtype=torch.bfloat16, device=DEV)
    B[torch.arange(N), torch.arange(N) % K] = 1.0
    pre = A.float()[:, torch.arange(N, device=DEV) % K]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    aux = torch.empty_like(out)
    ops.gemm(A, B, out, act=3, aux=aux)
    xr = pre.double().requires_grad_(True)
    ref = F.gelu(xr)
    ref.sum().backward()
    eg = (out.double() - ref.detach()).abs()
    ed = (aux.double() - xr.grad).abs()
    assert (eg <= 2.0 ** -8 * ref.detach().abs() + 2.5e-3).all(), eg.max().item()
    assert (ed <= 2.0 ** -8 * xr.grad.abs() + 4e-3).all(), ed.max().item()
    out1 = torch.empty_like(out)
    ops.gemm(A, B, out1, act=1)
    assert torch.equal(out1, out)


@pytest.mark.parametrize("N,K", [(3072, 768), (4096, 1024), (2112, 128)])
def test_gelu_table_with_long_bias_runs_as_column_ranges(ops, N, K):
    """fc1 of ViT-B / ViT-L (3072 / 4096 columns, vision_transformer.py:93-99): the GELU table and a bias of more than
    2048 columns do not both fit in LDS, so dino_gemm runs the launch as column ranges of <= 1920 columns -- same
    results as one launch would give: GELU + saved derivative, GELU alone, with the per-row factor of a folded
    LayerNorm in front, and with fused column sums."""
    dt = torch.bfloat16
    M = 256 * 70 - 24
    A, B = rnd((M, K), "cbA", dtype=dt), rnd((N, K), "cbB", std=1.5 / K ** 0.5, dtype=dt)
    bias = rnd((N,), "cbb", std=0.3)
    acc = A.float() @ B.float().t()
    out, aux = torch.empty((M, N), dtype=dt, device=DEV), torch.empty((M, N), dtype=dt, device=DEV)
    assert ops.gemm_wide_tile_supported(A, B, out, bias=bias, act=3, aux=aux)
    ops.gemm(A, B, out, bias=bias, act=3, aux=aux)
    xp = (acc + bias).requires_grad_(True)
    F.gelu(xp).sum().backward()
    assert rel(out, F.gelu(acc + bias)) < 1e-2
    assert rel(aux, xp.grad) < 1e-2
    out1 = torch.empty_like(out)
    ops.gemm(A, B, out1, bias=bias, act=1)
    assert torch.equal(out1, out)
    cs = torch.zeros(N, device=DEV)
    ops.gemm(A, B, out1, bias=bias, act=1, colsum=cs)
    assert rel(out1, out) < 1e-2
    assert rel(cs, out1.float().sum(0)) < 1e-2
    # ... behind a folded LayerNorm: the accumulator times a per-row factor before the bias
    r = (det_normal((M,), "cbr").abs() + 0.5).to(DEV)
    assert ops.gemm_wide_tile_supported(A, B, out, bias=bias, prescale=r, act=3, aux=aux)
    ops.gemm(A, B, out, bias=bias, prescale=r, act=3, aux=aux)
    xq = (acc * r[:, None] + bias).requires_grad_(True)
    F.gelu(xq).sum().backward()
    assert rel(out, F.gelu(xq.detach())) < 1e-2
    assert rel(aux, xq.grad) < 1e-2


# ------------------------------------------------------------------ LayerNorm folded into the GEMMs
@pytest.mark.parametrize("N2,act", [(1152, 0), (1536, 3), (1536, 1)])
def test_layernorm_folded_into_gemms(ops, N2, act):
    """Block.forward's `x + proj(..) -> norm -> linear` (vision_transformer.py:227-228) without a LayerNorm
    pass: the producing GEMM's ln_stats epilogue + dino_ln_stats_finalize against torch's row statistics, the
    consuming GEMM with folded weights + prescale against Linear(LayerNorm(x)) in fp32 (bf16 tolerance), at a
    token count the wide-tile kernel takes (ragged last row tile)."""
    M, D, eps = 16400 + 37, 384, 1e-6
    dt = torch.bfloat16
    a = rnd((M, D), "lfA", dtype=dt)
    wp = rnd((D, D), "lfWp", std=0.05, dtype=dt)
    bp = rnd((D,), "lfbp", std=0.1)
    res = (rnd((M, D), "lfR", std=1.0) + 3.0 * rnd((M, 1), "lfRo")).to(dt)     # rows with a common offset
    x = torch.empty((M, D), dtype=dt, device=DEV)
    nparts = ops.ln_stats_parts(D)
    stats = torch.full((nparts * M * 2,), float("nan"), device=DEV)
    kw = dict(bias=bp, residual=res)
    assert ops.gemm_wide_tile_supported(a, wp, x, ln_stats=stats, **kw)
    ops.gemm(a, wp, x, ln_stats=stats, **kw)
    x_plain = torch.empty_like(x)
    ops.gemm(a, wp, x_plain, **kw)
    assert torch.equal(x, x_plain)                               # the statistics do not disturb the output
    mean, rstd = ops.ln_stats_finalize(stats, nparts, M, D, eps)
    xf = (a.float() @ wp.float().t() + bp + res.float())         # what the epilogue held before rounding
    assert rel(mean, xf.mean(-1)) < 1e-4
    assert rel(rstd, 1.0 / torch.sqrt(xf.var(-1, unbiased=False) + eps)) < 2e-3
    m2, r2 = ops.layernorm_stats(x, eps)                         # ... and from the rounded rows themselves
    assert rel(m2, x.float().mean(-1)) < 1e-5
    assert rel(r2, 1.0 / torch.sqrt(x.float().var(-1, unbiased=False) + eps)) < 1e-4
    # consumer: folded weights
    gamma = 1.0 + 0.2 * rnd((D,), "lfg")
    beta = 0.3 * rnd((D,), "lfb")
    w = rnd((N2, D), "lfW", std=0.05)
    b = rnd((N2,), "lfbias", std=0.1)
    master = torch.cat([w.reshape(-1), gamma, beta, b]).contiguous()
    o_g, o_b, o_bias = N2 * D, N2 * D + D, N2 * D + 2 * D
    table = torch.tensor([[0, o_g, o_b, o_bias, 0, 0, N2, D]], dtype=torch.int64, device=DEV)
    wf = torch.empty(N2 * D, dtype=dt, device=DEV)
    bf = torch.empty(N2, device=DEV)
    ops.ln_fold_weights(master, table, wf, bf, N2)
    gw = gamma * w
    assert rel(wf.view(N2, D), gw - gw.mean(-1, keepdim=True)) < 1e-2
    assert rel(bf, b + w @ beta) < 1e-5
    y = torch.empty((M, N2), dtype=dt, device=DEV)
    aux = torch.empty_like(y) if act == 3 else None
    assert ops.gemm_wide_tile_supported(x, wf.view(N2, D), y, bias=bf, prescale=r2, act=act, aux=aux)
    ops.gemm(x, wf.view(N2, D), y, bias=bf, prescale=r2, act=act, aux=aux)
    pre = F.layer_norm(x.float(), (D,), gamma, beta, eps) @ w.t() + b
    ref = pre if act == 0 else F.gelu(pre)
    assert rel(y, ref) < 2e-2
    # the same with the factor formed from the partials inside the epilogue (no finalize launch)
    y3 = torch.empty_like(y)
    mo, ro = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    kw3 = dict(bias=bf, ln_stats_in=stats, ln_dim=D, ln_eps=eps, ln_mean_out=mo, ln_rstd_out=ro, act=act,
               aux=torch.empty_like(y) if act == 3 else None)
    assert ops.gemm_wide_tile_supported(x, wf.view(N2, D), y3, **kw3)
    ops.gemm(x, wf.view(N2, D), y3, **kw3)
    assert rel(mo, mean) < 1e-6 and rel(ro, rstd) < 1e-5
    assert rel(y3, ref) < 2e-2
    if act == 3:
        p = pre.double().requires_grad_(True)
        F.gelu(p).sum().backward()
        assert rel(aux, p.grad) < 5e-2   # (gelu second derivative is up to 0.4 per unit: the bf16 error of the pre-activation shows here)
    # the unfolded pair of kernels gives the same thing at the same tolerance (and the two agree closely)
    ln, _, _ = ops.layernorm_fwd(x, gamma, beta, eps)
    y2 = torch.empty_like(y)
    ops.gemm(ln, w.to(dt), y2, bias=b, act=act, aux=torch.empty_like(y) if act == 3 else None)
    assert rel(y, y2.float()) < 2e-2


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_bwd_also_writes_the_normalised_rows(ops, dtype):
    rows, D, eps = 4099, 384, 1e-6
    x = rnd((rows, D), "lxx", dtype=dtype)
    dy = rnd((rows, D), "lxd", dtype=dtype)
    dres = rnd((rows, D), "lxr", dtype=dtype)
    gamma = 1.0 + 0.1 * rnd((D,), "lxg")
    beta = 0.2 * rnd((D,), "lxb")
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
    outs = []
    for with_xn in (False, True):
        dg, db, dxs = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        kw = dict(dres=dres, dgamma=dg, dbeta=db, dx_colsum=dxs)
        if with_xn:
            xn = torch.empty_like(x)
            dx = ops.layernorm_bwd_xn(dy, x, gamma, beta, mean, rstd, xn=xn, **kw)
            assert rel(xn, y.float()) < (1e-6 if dtype == torch.float32 else 1e-2)
        else:
            dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, **kw)
        outs.append((dx, dg, db, dxs))
    # (two instantiations: the compiler may contract differently -- a last-bit flip of a bf16 output now and then)
    assert rel(outs[0][0], outs[1][0].float()) < (1e-6 if dtype == torch.float32 else 5e-3)
    for a_, b_ in zip(outs[0][1:], outs[1][1:]):
        assert rel(a_, b_) < 1e-5
