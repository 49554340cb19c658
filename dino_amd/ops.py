"""Tensor-level wrappers over the C ABI: shape/dtype checks on the host (a faulting kernel
can take the whole GPU down), raw device pointers + the current HIP stream into the library.

PyTorch is only the allocator / stream provider here; every computation is a HIP kernel.
"""
import ctypes

import torch

from . import _lib
from ._lib import DINO_BF16, DINO_F32, GemmArgs, check

_DT = {torch.float32: DINO_F32, torch.bfloat16: DINO_BF16}


def dt_code(dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"dino_amd supports float32 and bfloat16 activations, got {dtype}")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DinoHipError("dino_amd kernels need device tensors (no CPU fallback)")


def _req(cond, *info):
    """Operand validation in front of a kernel launch.  An explicit raise, not `assert`: under `python -O`
    asserts vanish, and a kernel launched on operands it was not sized for can fault the GPU."""
    if not cond:
        import sys
        fn = sys._getframe(1).f_code.co_name
        raise _lib.DinoHipError(f"dino_amd.ops.{fn}: operand check failed"
                                + (": " + ", ".join(str(i) for i in info) if info else ""))


def _f32(*ts):
    for t in ts:
        if t is not None:
            _req(t.dtype == torch.float32 and t.is_contiguous(), "expected a contiguous float32 tensor",
                 t.dtype, tuple(t.shape))


# ---------------------------------------------------------------------------------- GEMM
def _gemm_args(A, B, out, *, trans=False, M=None, N=None, K=None, lda=None, ldb=None, ldo=None,
               out_mode=0, act=0, bias=None, aux=None, residual=None, rowscale=None, rows_per_group=0,
               bias_rowscale=False, out_rowscale=False, row_map_n=0, res_mod=0, splitk=1, colsum=None,
               workspace=None, prescale=None, ln_stats=None, ln_stats_in=None, ln_dim=0, ln_eps=0.0,
               ln_mean_out=None, ln_rstd_out=None):
    """Checked dino_gemm_args for one GEMM (host-side shape validation happens here)."""
    _dev(A, B, out, bias, aux, residual, rowscale)
    _req(A.dtype == B.dtype and A.dim() == 2 and B.dim() == 2)
    _req(A.stride(1) == 1 and B.stride(1) == 1 and out.stride(-1) == 1)
    if trans:
        k_, m_ = A.shape
        k2, n_ = B.shape
    else:
        m_, k_ = A.shape
        n_, k2 = B.shape
    _req(k_ == k2, (A.shape, B.shape))
    M = m_ if M is None else M
    N = n_ if N is None else N
    K = k_ if K is None else K
    a = GemmArgs()
    a.dtype, a.trans = dt_code(A.dtype), int(trans)
    a.M, a.N, a.K, a.splitk = M, N, K, splitk
    a.A, a.lda = A.data_ptr(), (A.stride(0) if lda is None else lda)
    a.B, a.ldb = B.data_ptr(), (B.stride(0) if ldb is None else ldb)
    out2 = out.reshape(-1, out.shape[-1]) if out.dim() != 2 else out
    a.out, a.ldo = out.data_ptr(), (out2.stride(0) if ldo is None else ldo)
    if out_mode == 0:
        _req(out.dtype == A.dtype)
    else:
        _req(out.dtype == torch.float32)
    rows_needed = M + (M // row_map_n if row_map_n else 0)
    _req(out2.shape[0] >= rows_needed and out2.shape[1] >= N, (out2.shape, rows_needed, N))
    if out_mode == 3:  # one fp32 slab of M rows per k-split
        _req(out.is_contiguous() and out2.shape[0] >= max(1, splitk) * M and not row_map_n)
    a.out_mode, a.act = out_mode, act
    if bias is not None:
        _f32(bias)
        _req(bias.numel() >= N)
    a.bias = None if bias is None else bias.data_ptr()
    if aux is not None:
        _req(aux.dtype == A.dtype and aux.stride(-1) == 1 and aux.shape[0] >= M and aux.shape[1] >= N)
    a.aux, a.ldaux = (None, 0) if aux is None else (aux.data_ptr(), aux.stride(0))
    if residual is not None:
        _req(residual.dtype == A.dtype and residual.stride(-1) == 1 and residual.shape[-1] >= N)
        need = res_mod if res_mod else ((row_map_n + 1) if row_map_n else M)
        _req(residual.reshape(-1, residual.shape[-1]).shape[0] >= need)
    a.residual, a.ldres = (None, 0) if residual is None else (
        residual.data_ptr(), residual.reshape(-1, residual.shape[-1]).stride(0))
    if rowscale is not None:
        _f32(rowscale)
        _req(rows_per_group > 0 and rowscale.numel() * rows_per_group >= M)
    a.rowscale = None if rowscale is None else rowscale.data_ptr()
    a.rows_per_group = rows_per_group
    a.bias_rowscale, a.out_rowscale, a.row_map_n = int(bias_rowscale), int(out_rowscale), row_map_n
    a.res_mod = res_mod
    if colsum is not None:
        _dev(colsum)
        _f32(colsum)
        _req(colsum.numel() >= (M if trans else N))
    a.colsum = None if colsum is None else colsum.data_ptr()
    if workspace is not None:
        _dev(workspace)
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if prescale is not None:      # LayerNorm folded into this GEMM (consumer side): rstd per row
        _dev(prescale)
        _f32(prescale)
        _req(prescale.numel() >= M and not trans)
        a.prescale = prescale.data_ptr()
    if ln_stats is not None:      # ... producer side: (sum, sum of squares) per row and 96-column slice
        _dev(ln_stats)
        _f32(ln_stats)
        _req(ln_stats.numel() >= ln_stats_parts(N) * M * 2 and not trans and out_mode == 0)
        a.ln_stats = ln_stats.data_ptr()
    if ln_stats_in is not None:   # consumer side, rstd formed from the producer's partials in the epilogue
        _dev(ln_stats_in, ln_mean_out, ln_rstd_out)
        _f32(ln_stats_in, ln_mean_out, ln_rstd_out)
        _req(ln_dim > 0 and ln_stats_in.numel() >= ln_stats_parts(ln_dim) * M * 2 and not trans and prescale is None)
        _req(ln_mean_out is None or ln_mean_out.numel() >= M)
        _req(ln_rstd_out is None or ln_rstd_out.numel() >= M)
        a.ln_stats_in, a.ln_dim, a.ln_eps = ln_stats_in.data_ptr(), int(ln_dim), float(ln_eps)
        a.ln_mean_out = None if ln_mean_out is None else ln_mean_out.data_ptr()
        a.ln_rstd_out = None if ln_rstd_out is None else ln_rstd_out.data_ptr()
    return a


def ln_stats_parts(N: int) -> int:
    """Number of 96-column slices dino_gemm's ln_stats epilogue leaves per row."""
    return (N + 95) // 96


def gemm_wide_tile_supported(A, B, out, **kw) -> bool:
    """Whether gemm(A, B, out, **kw) would run on the wide-tile bf16 kernel (the one with the folded-LayerNorm
    epilogues).  Launches nothing."""
    a = _gemm_args(A, B, out, **kw)
    return bool(_lib.load().dino_gemm_wide_tile_supported(ctypes.byref(a)))


def gemm(A, B, out, **kw):
    """See include/dino_hip.h: dino_gemm.  A, B are 2-D row-major (possibly strided rows).
    NT: A [M,K], B [N,K].  TN: A [K,M], B [K,N].  out: [M(+CLS rows),N] in A.dtype or fp32."""
    a = _gemm_args(A, B, out, **kw)
    check(_lib.load().dino_gemm(ctypes.byref(a), _stream()), "dino_gemm")
    return out


def sum_slabs(slabs, out, accumulate=False):
    """out (+)= slabs.sum(0): the reduction of gemm(..., out_mode=3).  slabs: fp32 [S, ...] contiguous."""
    _dev(slabs, out)
    _f32(slabs, out)
    n = out.numel()
    _req(slabs.dim() >= 2 and slabs.numel() == slabs.shape[0] * n and n % 4 == 0)
    check(_lib.load().dino_sum_slabs(slabs.data_ptr(), slabs.shape[0], n, out.data_ptr(), int(accumulate),
                                     _stream()), "dino_sum_slabs")
    return out


def gemm_group(calls):
    """calls: list of (A, B, out, kwargs) as for gemm(); same results as calling gemm() on each.
    Weight-gradient problems over the same tokens run as one launch (dino_gemm_group)."""
    arr = (GemmArgs * len(calls))(*[_gemm_args(A, B, out, **kw) for A, B, out, kw in calls])
    check(_lib.load().dino_gemm_group(arr, len(calls), _stream()), "dino_gemm_group")


# ---------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps, *, rows=None, x_stride=None, out=None, y_stride=None):
    """x: [..., D] viewed as rows of D with element stride x_stride between rows."""
    _dev(x, gamma, beta)
    _f32(gamma, beta)
    D = x.shape[-1]
    rows = x.numel() // D if rows is None else rows
    x_stride = D if x_stride is None else x_stride
    if out is None:
        out = torch.empty((rows, D), dtype=x.dtype, device=x.device)
        y_stride = D
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.load().dino_layernorm_fwd(dt_code(x.dtype), _p(x), x_stride, _p(out), y_stride,
                                         _p(gamma), _p(beta), _p(mean), _p(rstd), rows, D,
                                         float(eps), _stream()), "dino_layernorm_fwd")
    return out, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, *, dres=None, dx=None, dgamma=None, dbeta=None,
                  dx_colsum=None, rowscale=None, rows_per_group=0, rows=None, dy_stride=None, x_stride=None,
                  dres_stride=None, dx_stride=None, workspace=None):
    _dev(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dx_colsum, rowscale)
    _f32(gamma, mean, rstd, dgamma, dbeta, dx_colsum, rowscale)
    D = x.shape[-1]
    rows = mean.numel() if rows is None else rows
    if dx is None:
        dx = torch.empty((rows, D), dtype=x.dtype, device=x.device)
        dx_stride = D
    check(_lib.load().dino_layernorm_bwd(
        dt_code(x.dtype), _p(dy), D if dy_stride is None else dy_stride, _p(x),
        D if x_stride is None else x_stride, _p(gamma), _p(mean), _p(rstd), _p(dres),
        D if dres_stride is None else dres_stride, _p(dx), D if dx_stride is None else dx_stride,
        _p(dgamma), _p(dbeta), _p(dx_colsum), _p(rowscale), rows_per_group, rows, D, _p(workspace),
        0 if workspace is None else workspace.numel() * workspace.element_size(), _stream()),
        "dino_layernorm_bwd")
    return dx


def layernorm_bwd_xn(dy, x, gamma, beta, mean, rstd, *, xn, dres=None, dx=None, dgamma=None, dbeta=None,
                     dx_colsum=None, rowscale=None, rows_per_group=0, workspace=None):
    """layernorm_bwd that also writes the normalised rows into xn (LayerNorm folded into the GEMMs: the
    forward never materialised them, the weight gradient of the next linear layer reads them here)."""
    _dev(dy, x, gamma, beta, mean, rstd, dres, dx, xn, dgamma, dbeta, dx_colsum, rowscale)
    _f32(gamma, beta, mean, rstd, dgamma, dbeta, dx_colsum, rowscale)
    D = x.shape[-1]
    rows = mean.numel()
    _req(x.is_contiguous() and dy.is_contiguous() and xn.is_contiguous() and xn.dtype == x.dtype)
    _req(x.numel() == rows * D and dy.numel() == rows * D and xn.numel() == rows * D and gamma.numel() == D
         and beta.numel() == D)
    if dx is None:
        dx = torch.empty((rows, D), dtype=x.dtype, device=x.device)
    _req(dx.is_contiguous() and dx.numel() == rows * D and (dres is None or (dres.is_contiguous() and dres.numel() == rows * D)))
    check(_lib.load().dino_layernorm_bwd_xn(
        dt_code(x.dtype), _p(dy), D, _p(x), D, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dres), D, _p(dx), D,
        _p(xn), D, _p(dgamma), _p(dbeta), _p(dx_colsum), _p(rowscale), rows_per_group, rows, D, _p(workspace),
        0 if workspace is None else workspace.numel() * workspace.element_size(), _stream()),
        "dino_layernorm_bwd_xn")
    return dx


def ln_stats_finalize(stats, nparts, rows, D, eps, mean=None, rstd=None):
    """(mean, rstd) fp32 [rows] from the ln_stats partials of the GEMM that wrote the rows."""
    _dev(stats, mean, rstd)
    _f32(stats, mean, rstd)
    _req(stats.numel() >= nparts * rows * 2)
    if mean is None:
        mean = torch.empty(rows, dtype=torch.float32, device=stats.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=stats.device)
    _req(mean.numel() >= rows and rstd.numel() >= rows)
    check(_lib.load().dino_ln_stats_finalize(_p(stats), nparts, rows, D, float(eps), _p(mean), _p(rstd), _stream()),
          "dino_ln_stats_finalize")
    return mean, rstd


def layernorm_stats(x, eps):
    """(mean, rstd) of the rows of x [rows, D] (contiguous): LayerNorm's statistics without its output."""
    _dev(x)
    _req(x.dim() == 2 and x.is_contiguous())
    rows, D = x.shape
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.load().dino_layernorm_stats(dt_code(x.dtype), _p(x), D, _p(mean), _p(rstd), rows, D, float(eps),
                                           _stream()), "dino_layernorm_stats")
    return mean, rstd


def ln_fold_weights(master, table, wfold, bfold, maxN):
    """table: int64 device [count, 8] (include/dino_hip.h: dino_ln_fold_weights)."""
    _dev(master, table, wfold, bfold)
    _f32(master, bfold)
    _req(table.dtype == torch.int64 and table.is_contiguous() and table.dim() == 2 and table.shape[1] == 8)
    _req(wfold.dtype == torch.bfloat16 and wfold.is_contiguous())
    check(_lib.load().dino_ln_fold_weights(_p(master), _p(table), table.shape[0], _p(wfold), _p(bfold), int(maxN),
                                           _stream()), "dino_ln_fold_weights")


def layernorm_bwd_static(dy, x, gamma, mean, rstd, *, dx, dres=None, dgamma=None, dbeta=None,
                         dx_colsum=None, rowscale=None, rows_per_group=0):
    """dino_layernorm_bwd_static: the allocation-free variant (partials in the library's own device
    array).  The caller provides every buffer, dx included; nothing is allocated here either."""
    _dev(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dx_colsum, rowscale)
    _f32(gamma, mean, rstd, dgamma, dbeta, dx_colsum, rowscale)
    D = x.shape[-1]
    rows = mean.numel()
    _req(dx.shape == x.shape and dx.is_contiguous() and x.is_contiguous() and dy.is_contiguous())
    _req(x.numel() == rows * D and dy.numel() == rows * D)
    if x.device.index != torch.cuda.current_device():
        raise _lib.DinoHipError("dino_layernorm_bwd_static: the tensors' GPU must be the current device")
    check(_lib.load().dino_layernorm_bwd_static(
        dt_code(x.dtype), _p(dy), D, _p(x), D, _p(gamma), _p(mean), _p(rstd), _p(dres), D, _p(dx), D,
        _p(dgamma), _p(dbeta), _p(dx_colsum), _p(rowscale), rows_per_group, rows, D, _stream()),
        "dino_layernorm_bwd_static")
    return dx


def gemm_work_queue(on=None):
    """The wide-tile NT GEMM's work queue (include/dino_hip.h: dino_gemm_work_queue): on=True/False sets it,
    None only queries; returns the previous state.  (The entry point returns a state, not an error code.)"""
    return bool(_lib.load().dino_gemm_work_queue(-1 if on is None else int(bool(on))))


def occupy_cus(workgroups, micros, stream=None):
    """Diagnostic (include/dino_hip.h: dino_diag_occupy_cus): hold `workgroups` CUs for `micros` us on `stream`
    (default: the current one) -- what an overlapped RCCL all-reduce does to the persistent kernels."""
    check(_lib.load().dino_diag_occupy_cus(int(workgroups), int(micros),
                                           _stream() if stream is None else ctypes.c_void_p(stream.cuda_stream)),
          "dino_diag_occupy_cus")


# ---------------------------------------------------------------------------------- attention
def attn_fwd(qkv, Bc, N, H, rowscale=None, out=None):
    _dev(qkv, rowscale)
    D = H * 64
    _req(qkv.is_contiguous() and qkv.shape == (Bc * N, 3 * D), (qkv.shape, Bc, N, H))
    if out is None:
        out = torch.empty((Bc * N, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((Bc, H, N), dtype=torch.float32, device=qkv.device)
    if rowscale is not None:
        _f32(rowscale)
        _req(rowscale.numel() >= Bc)
    check(_lib.load().dino_attn_fwd(dt_code(qkv.dtype), _p(qkv), _p(out), _p(lse), Bc, N, H,
                                    _p(rowscale), _stream()), "dino_attn_fwd")
    return out, lse


def attn_bwd(qkv, out, dout, lse, Bc, N, H, rowscale=None, dqkv=None):
    _dev(qkv, out, dout, lse, rowscale)
    D = H * 64
    _req(qkv.is_contiguous() and qkv.shape == (Bc * N, 3 * D))
    _req(out.is_contiguous() and dout.is_contiguous() and out.shape == dout.shape == (Bc * N, D))
    _req(out.dtype == dout.dtype == qkv.dtype and lse.shape == (Bc, H, N))
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    delta = torch.empty((Bc, H, N), dtype=torch.float32, device=qkv.device)
    check(_lib.load().dino_attn_bwd(dt_code(qkv.dtype), _p(qkv), _p(out), _p(dout), _p(lse),
                                    _p(dqkv), Bc, N, H, _p(rowscale), _p(delta), _stream()), "dino_attn_bwd")
    return dqkv


def attn_cls_fwd(qkv, Bc, N, H, rowscale=None, out=None):
    """Attention of the CLS query alone (the last block: only x[:, 0] leaves the backbone).  Returns
    (out_cls [Bc, H*64] in qkv.dtype, lse_cls fp32 [Bc, H])."""
    _dev(qkv)
    _req(qkv.is_contiguous() and qkv.shape == (Bc * N, 3 * H * 64), (qkv.shape, Bc, N, H))
    _req(rowscale is None or (rowscale.dtype == torch.float32 and rowscale.numel() == Bc and rowscale.is_contiguous()))
    if out is None:
        out = torch.empty((Bc, H * 64), dtype=qkv.dtype, device=qkv.device)
    _req(out.is_contiguous() and out.shape == (Bc, H * 64) and out.dtype == qkv.dtype)
    lse = torch.empty((Bc, H), dtype=torch.float32, device=qkv.device)
    check(_lib.load().dino_attn_cls_fwd(dt_code(qkv.dtype), _p(qkv), _p(out), _p(lse), Bc, N, H,
                                        None if rowscale is None else _p(rowscale), _stream()), "dino_attn_cls_fwd")
    return out, lse


def attn_cls_bwd(qkv, dout_cls, lse_cls, Bc, N, H, dqkv=None):
    """Backward of attn_cls_fwd (dout_cls already times the drop-path factor): all of dqkv [Bc*N, 3*H*64]."""
    _dev(qkv, dout_cls, lse_cls)
    _req(qkv.is_contiguous() and qkv.shape == (Bc * N, 3 * H * 64), (qkv.shape, Bc, N, H))
    _req(dout_cls.is_contiguous() and dout_cls.shape == (Bc, H * 64) and dout_cls.dtype == qkv.dtype)
    _req(lse_cls.dtype == torch.float32 and lse_cls.is_contiguous() and lse_cls.numel() == Bc * H)
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    _req(dqkv.is_contiguous() and dqkv.shape == qkv.shape and dqkv.dtype == qkv.dtype)
    check(_lib.load().dino_attn_cls_bwd(dt_code(qkv.dtype), _p(qkv), _p(dout_cls), _p(lse_cls), _p(dqkv), Bc, N, H,
                                        _stream()), "dino_attn_cls_bwd")
    return dqkv


def attn_probs(qkv, Bc, N, H):
    """softmax(q k^T / 8) of every (crop, head): [Bc, H, N, N] in qkv.dtype."""
    _dev(qkv)
    _req(qkv.is_contiguous() and qkv.shape == (Bc * N, 3 * H * 64), (qkv.shape, Bc, N, H))
    probs = torch.empty((Bc, H, N, N), dtype=qkv.dtype, device=qkv.device)
    check(_lib.load().dino_attn_probs(dt_code(qkv.dtype), _p(qkv), _p(probs), Bc, N, H, _stream()),
          "dino_attn_probs")
    return probs


# ---------------------------------------------------------------------------------- loss
def loss_fwd_bwd(student, teacher, center, center_acc, B, ncrops, student_temp, teacher_temp,
                 dstudent=None):
    """Returns (dstudent, loss_parts[ncrops*B]); center_acc += column sums of teacher."""
    _dev(student, teacher, center, center_acc)
    _f32(center, center_acc)
    K = student.shape[-1]
    _req(student.is_contiguous() and teacher.is_contiguous())
    _req(student.shape == (ncrops * B, K) and teacher.shape == (2 * B, K))
    _req(teacher.dtype == student.dtype and center.numel() == K)
    if dstudent is None:
        dstudent = torch.empty_like(student)
    parts = torch.empty(ncrops * B, dtype=torch.float32, device=student.device)
    stats = torch.empty(2 * B, dtype=torch.float32, device=student.device)
    check(_lib.load().dino_loss_fwd_bwd(dt_code(student.dtype), _p(student), _p(teacher),
                                        _p(center), _p(dstudent), _p(parts), _p(center_acc),
                                        _p(stats), B, ncrops, K, 1.0 / student_temp,
                                        1.0 / teacher_temp, _stream()), "dino_loss_fwd_bwd")
    return dstudent, parts


def proto_loss_chunk(phase, s_chunk, k0, kc, *, teacher, center, teacher_lse, run, B, ncrops, K, student_temp,
                     teacher_temp, gout=None, parts=None):
    """One phase of the chunked prototype-layer + loss walk (include/dino_hip.h: dino_proto_loss_chunk)."""
    _dev(s_chunk, teacher, center, teacher_lse, run, gout, parts)
    _f32(center, teacher_lse, run, gout, parts)
    _req(teacher.dim() == 2 and teacher.shape[1] == K and teacher.stride(1) == 1 and teacher.shape[0] == 2 * B)
    _req(center.numel() == K and teacher_lse.numel() >= 2 * B and run.numel() >= 2 * ncrops * B)
    lds = 0
    if phase in (1, 3):
        _req(s_chunk.dtype == teacher.dtype and s_chunk.dim() == 2 and s_chunk.stride(1) == 1)
        _req(s_chunk.shape[0] == ncrops * B and s_chunk.shape[1] >= kc and 0 <= k0 and k0 + kc <= K)
        lds = s_chunk.stride(0)
    if phase == 2:
        _req(parts is not None and parts.numel() >= ncrops * B)
    check(_lib.load().dino_proto_loss_chunk(dt_code(teacher.dtype), phase, _p(s_chunk), lds, _p(teacher),
                                            teacher.stride(0), _p(center), _p(teacher_lse), _p(run), B, ncrops, K,
                                            k0, kc, 1.0 / student_temp, 1.0 / teacher_temp, _p(gout), _p(parts),
                                            _stream()), "dino_proto_loss_chunk")


def center_update(center, center_acc, momentum, inv_count, loss_parts=None, loss_out=None):
    """center / center_acc None: only fold loss_parts into loss_out."""
    _dev(center, center_acc, loss_parts, loss_out)
    _f32(center, center_acc, loss_parts, loss_out)
    K = 0 if center is None else center.numel()
    _req(center is None or center_acc.numel() == K)
    n = 0 if loss_parts is None else loss_parts.numel()
    check(_lib.load().dino_center_update(_p(center), _p(center_acc), K, float(momentum),
                                         float(inv_count), _p(loss_parts), n, _p(loss_out),
                                         _stream()), "dino_center_update")


# ---------------------------------------------------------------------------------- optimiser
def grad_sqnorm(grad, chunks, chunk_sq):
    check(_lib.load().dino_grad_sqnorm(_p(grad), _p(chunks), chunks.shape[0], _p(chunk_sq),
                                       _stream()), "dino_grad_sqnorm")


def clip_coef(chunk_sq, tensor_chunk_begin, has_grad, clip, norms, coef, stats):
    nt = tensor_chunk_begin.numel() - 1
    check(_lib.load().dino_clip_coef(_p(chunk_sq), _p(tensor_chunk_begin), nt, _p(has_grad),
                                     float(clip), _p(norms), _p(coef), _p(stats), _stream()),
          "dino_clip_coef")


def grad_scale(grad, chunks, coef):
    check(_lib.load().dino_grad_scale(_p(grad), _p(chunks), chunks.shape[0], _p(coef), _stream()),
          "dino_grad_scale")


def adamw_ema(param, grad, exp_avg, exp_avg_sq, teacher, param_shadow, teacher_shadow, chunks, hp,
              coef, beta1, beta2, eps, ema_m, ema_one_minus_m):
    _dev(param, grad, exp_avg, exp_avg_sq, chunks, hp)
    _f32(param, grad, exp_avg, exp_avg_sq, teacher, hp, coef)
    check(_lib.load().dino_adamw_ema(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(teacher),
                                     _p(param_shadow), _p(teacher_shadow), _p(chunks),
                                     chunks.shape[0], _p(hp), _p(coef), beta1, beta2, eps,
                                     float(ema_m), float(ema_one_minus_m), _stream()),
          "dino_adamw_ema")


def ema(teacher, student, teacher_shadow, m, one_minus_m):
    _dev(teacher, student, teacher_shadow)
    _f32(teacher, student)
    _req(teacher.numel() == student.numel() and teacher.numel() % 8 == 0)
    check(_lib.load().dino_ema(_p(teacher), _p(student), _p(teacher_shadow), teacher.numel(),
                               float(m), float(one_minus_m), _stream()), "dino_ema")


def cast_bf16(src, dst):
    _dev(src, dst)
    _f32(src)
    _req(dst.dtype == torch.bfloat16 and dst.numel() == src.numel() and src.numel() % 8 == 0)
    check(_lib.load().dino_cast_bf16(_p(src), _p(dst), src.numel(), _stream()), "dino_cast_bf16")


def transpose_batched(src_base, dst_base, table):
    """table: int64 device tensor [nmat,4] = (src offset, dst offset, rows, cols)."""
    _dev(src_base, dst_base, table)
    _f32(src_base)
    check(_lib.load().dino_transpose_batched(dt_code(dst_base.dtype), _p(src_base), _p(dst_base),
                                             _p(table), table.shape[0], _stream()),
          "dino_transpose_batched")


# ---------------------------------------------------------------------------------- tokens
def im2col(img, P, dtype, cls_row=False, out=None):
    """img: fp32 [Bc,3,H,W], or a list of such crops of one resolution (their patch rows are
    written one crop after the other, as if the crops had been concatenated first)."""
    imgs = list(img) if isinstance(img, (list, tuple)) else [img]
    _dev(*imgs)
    _, _, H, W = imgs[0].shape
    for t in imgs:
        _req(t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4 and t.shape[1:] == (3, H, W))
    gh, gw = H // P, W // P
    rows_per = gh * gw + int(cls_row)
    Kp = (3 * P * P + 7) // 8 * 8
    Bc = sum(t.shape[0] for t in imgs)
    if out is None:
        out = torch.empty((Bc * rows_per, Kp), dtype=dtype, device=imgs[0].device)
    _req(out.is_contiguous() and tuple(out.shape) == (Bc * rows_per, Kp) and out.dtype == dtype)
    row = 0
    for t in imgs:
        check(_lib.load().dino_im2col(dt_code(dtype), _p(t), out[row:].data_ptr(), t.shape[0], H, W, P,
                                      int(cls_row), _stream()),
              "dino_im2col")
        row += t.shape[0] * rows_per
    return out


def to_tensor_normalize(u8, mean, std, out=None, channels_last=True):
    """uint8 crops [B,H,W,C] (channels_last, the layout ToTensor takes) or [B,C,H,W] -> fp32
    [B,C,H,W], (u8/255 - mean)/std in torchvision's operation order (augmentations.py:102-107)."""
    import ctypes
    _dev(u8)
    _req(u8.dtype == torch.uint8 and u8.is_contiguous() and u8.dim() == 4)
    B, C, H, W = (u8.shape[0], u8.shape[3], u8.shape[1], u8.shape[2]) if channels_last else u8.shape
    _req(len(mean) == C and len(std) == C and C <= 4)
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=u8.device)
    _req(out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (B, C, H, W))
    m = (ctypes.c_float * C)(*[float(v) for v in mean])
    s = (ctypes.c_float * C)(*[float(v) for v in std])
    check(_lib.load().dino_to_tensor_normalize(_p(u8), _p(out), B, C, H, W, int(channels_last),
                                               ctypes.cast(m, ctypes.c_void_p),
                                               ctypes.cast(s, ctypes.c_void_p), _stream()),
          "dino_to_tensor_normalize")
    return out


def drop_path_scales(seed, keep, Bct, row_crop=None):
    """keep: device fp32 [L] -> (scales [L,2,Bct], row_scales [L,2,M] | None): floor(keep + U[0,1)) / keep."""
    _dev(keep, row_crop)
    _f32(keep)
    L = keep.numel()
    scales = torch.empty((L, 2, Bct), dtype=torch.float32, device=keep.device)
    rows = None
    M = 0
    if row_crop is not None:
        _req(row_crop.dtype == torch.int64 and row_crop.is_contiguous())
        M = row_crop.numel()
        rows = torch.empty((L, 2, M), dtype=torch.float32, device=keep.device)
    check(_lib.load().dino_drop_path_scales(int(seed) & ((1 << 64) - 1), _p(keep), L, Bct, _p(row_crop), M,
                                            _p(scales), _p(rows), _stream()), "dino_drop_path_scales")
    return scales, rows


# ------------------------------------------------------------------------- device augmentation
AUG_PARAM_WORDS, AUG_SLOTS = 24, 6


def _aug_check(img, params):
    _dev(img, params)
    _req(img.dtype == torch.uint8 and img.is_contiguous() and img.dim() == 4 and img.shape[3] == 3)
    _req(img.shape[1] == img.shape[2], "crops are square")
    _req(params.dtype == torch.int32 and params.is_contiguous() and tuple(params.shape) == (img.shape[0], AUG_PARAM_WORDS))


def aug_resized_crop(src, params, S, kmax, tmp_rows, out=None):
    """src: uint8 [B,Hs,Ws,3]; params: int32 [ncrops, 24] (include/dino_hip.h) -> uint8 [ncrops,S,S,3]:
    crop box, bicubic resize to S x S, optional flip -- Pillow's arithmetic (augmentations.py:112-116, :90)."""
    _dev(src, params)
    _req(src.dtype == torch.uint8 and src.is_contiguous() and src.dim() == 4 and src.shape[3] == 3)
    _req(params.dtype == torch.int32 and params.is_contiguous() and params.dim() == 2 and params.shape[1] == AUG_PARAM_WORDS)
    n = params.shape[0]
    B, Hs, Ws, _ = src.shape
    _req(0 < tmp_rows <= Hs and 0 < kmax <= 256)
    dev = src.device
    if out is None:
        out = torch.empty((n, S, S, 3), dtype=torch.uint8, device=dev)
    _req(out.dtype == torch.uint8 and out.is_contiguous() and tuple(out.shape) == (n, S, S, 3))
    bounds = torch.empty((n, 2, S, 2), dtype=torch.int32, device=dev)
    coeffs = torch.empty((n, 2, S, kmax), dtype=torch.int32, device=dev)
    tmp = torch.empty((n, tmp_rows, S, 3), dtype=torch.uint8, device=dev)
    check(_lib.load().dino_aug_resized_crop(_p(src), B, Hs, Ws, _p(params), n, S, kmax, _p(bounds), _p(coeffs),
                                            _p(tmp), tmp_rows, _p(out), _stream()), "dino_aug_resized_crop")
    return out


def aug_gray_sums(img, sums=None, params=None, slot=0):
    """uint32 [ncrops]: sum of convert("L") per crop (the mean ImageEnhance.Contrast blends with); with
    params, only for the crops whose op[slot] is contrast."""
    _dev(img, params)
    _req(img.dtype == torch.uint8 and img.is_contiguous() and img.dim() == 4 and img.shape[3] == 3)
    if sums is None:
        sums = torch.empty((img.shape[0],), dtype=torch.int32, device=img.device)
    _req(sums.dtype == torch.int32 and sums.numel() >= img.shape[0])
    if params is not None:
        _aug_check(img, params)
    check(_lib.load().dino_aug_gray_sums(_p(img), img.shape[0], img.shape[1], _p(sums), _p(params), int(slot),
                                         _stream()), "dino_aug_gray_sums")
    return sums


def aug_color_op(img, params, slot, sums):
    """In place: op[slot] of every crop (brightness / contrast / saturation / hue / grey-scale / solarize)."""
    _aug_check(img, params)
    _dev(sums)
    _req(0 <= slot < AUG_SLOTS and sums.dtype == torch.int32 and sums.numel() >= img.shape[0])
    check(_lib.load().dino_aug_color_op(_p(img), img.shape[0], img.shape[1], _p(params), slot, _p(sums),
                                        _stream()), "dino_aug_color_op")
    return img


def aug_box_blur(src, dst, params, axis):
    """One box-blur pass (axis 0 = along x, 1 = along y) of the crops whose blur word is set."""
    _aug_check(src, params)
    _aug_check(dst, params)
    _req(src.shape == dst.shape and src.data_ptr() != dst.data_ptr())
    check(_lib.load().dino_aug_box_blur(_p(src), _p(dst), src.shape[0], src.shape[1], _p(params), int(axis),
                                        _stream()), "dino_aug_box_blur")
    return dst


def pos_table(pos, idx, w, n_dst, dtype):
    """pos: fp32 [1+n_src, D]; idx/w: [n_dst, ntaps] or None (identity)."""
    _dev(pos, idx, w)
    _f32(pos, w)
    D = pos.shape[-1]
    ntaps = 0 if idx is None else idx.shape[1]
    if ntaps == 0:
        _req(pos.shape[0] == n_dst + 1)
    else:
        _req(idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape == w.shape == (n_dst, ntaps))
    out = torch.empty((n_dst + 1, D), dtype=dtype, device=pos.device)
    check(_lib.load().dino_pos_table(dt_code(dtype), _p(pos), _p(idx), _p(w), ntaps, _p(out), n_dst,
                                     D, _stream()), "dino_pos_table")
    return out


def cls_rows(tokens, cls, pos0, Bc, N):
    _dev(tokens, cls, pos0)
    _f32(cls, pos0)
    D = tokens.shape[-1]
    _req(tokens.is_contiguous() and tokens.numel() == Bc * N * D and cls.numel() == D)
    check(_lib.load().dino_cls_rows(dt_code(tokens.dtype), _p(tokens), _p(cls), _p(pos0), Bc, N, D,
                                    _stream()), "dino_cls_rows")


def fold_rows(x, out, Bc, N):
    _dev(x, out)
    _f32(out)
    D = x.shape[-1]
    _req(x.is_contiguous() and x.numel() == Bc * N * D and out.numel() == N * D)
    check(_lib.load().dino_fold_rows(dt_code(x.dtype), _p(x), _p(out), Bc, N, D, _stream()),
          "dino_fold_rows")


def pos_table_bwd(dpos_local, idx, w, dpos, n_dst):
    _dev(dpos_local, idx, w, dpos)
    _f32(dpos_local, w, dpos)
    D = dpos.shape[-1]
    ntaps = 0 if idx is None else idx.shape[1]
    _req(dpos_local.numel() == (n_dst + 1) * D)
    check(_lib.load().dino_pos_table_bwd(_p(dpos_local), _p(idx), _p(w), ntaps, _p(dpos), n_dst, D,
                                         _stream()), "dino_pos_table_bwd")


# ---------------------------------------------------------------------------------- head
def l2norm_fwd(x):
    _dev(x)
    _req(x.is_contiguous() and x.dim() == 2)
    y = torch.empty_like(x)
    norm = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.load().dino_l2norm_fwd(dt_code(x.dtype), _p(x), _p(y), _p(norm), x.shape[0],
                                      x.shape[1], _stream()), "dino_l2norm_fwd")
    return y, norm


def l2norm_bwd(dy, y, norm):
    _dev(dy, y, norm)
    _f32(dy, norm)
    _req(y.is_contiguous() and dy.shape == y.shape)
    dx = torch.empty_like(y)
    check(_lib.load().dino_l2norm_bwd(dt_code(y.dtype), _p(dy), _p(y), _p(norm), _p(dx), y.shape[0],
                                      y.shape[1], _stream()), "dino_l2norm_bwd")
    return dx


def weightnorm_fwd(g, v, dtype, w=None, wt=None, want_t=True):
    _dev(g, v)
    _f32(g, v)
    K, D = v.shape
    _req(g.numel() == K)
    if w is None:
        w = torch.empty((K, D), dtype=dtype, device=v.device)
    if wt is None and want_t:
        wt = torch.empty((D, K), dtype=dtype, device=v.device)
    vnorm = torch.empty(K, dtype=torch.float32, device=v.device)
    check(_lib.load().dino_weightnorm_fwd(dt_code(dtype), _p(g), _p(v), _p(w), _p(wt), _p(vnorm), K,
                                          D, _stream()), "dino_weightnorm_fwd")
    return w, wt, vnorm


def weightnorm_bwd(dW, g, v, vnorm, dg, dv):
    _dev(dW, g, v, vnorm, dg, dv)
    _f32(dW, g, v, vnorm, dg, dv)
    K, D = v.shape
    check(_lib.load().dino_weightnorm_bwd(_p(dW), _p(g), _p(v), _p(vnorm), _p(dg), _p(dv), K, D,
                                          _stream()), "dino_weightnorm_bwd")


def colsum(x, out, rowscale=None, rows_per_group=0, M=None, N=None):
    _dev(x, out, rowscale)
    _f32(out, rowscale)
    _req(x.dim() == 2 and x.stride(1) == 1)
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    _req(out.numel() >= N)
    check(_lib.load().dino_colsum(dt_code(x.dtype), _p(x), x.stride(0), _p(out), M, N, _p(rowscale),
                                  rows_per_group, _stream()), "dino_colsum")


def scale_by_scalar(x, scalar):
    _dev(x, scalar)
    _f32(scalar)
    _req(x.is_contiguous() and x.numel() % 8 == 0)
    check(_lib.load().dino_scale_by_scalar(dt_code(x.dtype), _p(x), x.numel(), _p(scalar), _stream()),
          "dino_scale_by_scalar")
