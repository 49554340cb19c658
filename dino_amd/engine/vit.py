"""Explicit forward / backward schedules of the ViT backbone and the DINO head on the HIP
kernels.  No autograd graph inside: forward returns a tape of saved activations, backward
walks it in reverse and writes parameter gradients straight into the store's flat grad
buffer (accumulating, so the two student backbone passes share weights correctly).

Reference semantics: dino/models/vision_transformer.py (PatchEmbed 242-245, prepare_tokens
361-375, Block 220-229, Attention 122-138, Mlp 93-99, forward 377-383, DINOHead 716-720).
"""
import math
import os as _os
from dataclasses import dataclass
from typing import List, Optional

import torch

from .. import ops
from ..utils.pos_interp import bicubic_taps
from .store import ParamStore, side_stream


class Tape(dict):
    """Saved activations of one recorded backbone pass (a dict that can be weakly referenced)."""
    __slots__ = ("__weakref__",)

    __hash__ = object.__hash__

    def __eq__(self, other):
        return self is other


@dataclass
class VitSpec:
    patch_size: int
    embed_dim: int
    depth: int
    num_heads: int
    mlp_hidden: int
    eps: float
    drop_path: List[float]  # per block (student only; 0 disables)


def _wgrad_splitk(m_out: int, n_out: int, k: int) -> int:
    tiles = ((m_out + 127) // 128) * ((n_out + 127) // 128)
    want = max(1, -(-768 // tiles))
    return int(max(1, min(want, k // 512 if k >= 1024 else 1)))


def _wgrad_call(st: ParamStore, dy, x, wname, bname=None):
    out_f, in_f = dy.shape[1], x.shape[1]
    gw = st.grad(wname).reshape(out_f, -1)
    return (dy, x, gw, dict(trans=True, N=in_f, out_mode=2,
                            splitk=_wgrad_splitk(out_f, in_f, dy.shape[0]),
                            workspace=st.workspace_wg if _WG_STREAM else st.workspace,
                            colsum=None if bname is None else st.grad(bname)))


def _wgrad(st: ParamStore, dy, x, wname, bname, rowscale=None, rpg=0, group=None):
    """dW += dy^T x (TN GEMM, split over the token dim); db += colsum(dy).  With `group` (a list)
    the GEMM is only queued: the caller flushes the block's weight gradients as ONE launch
    (ops.gemm_group), which keeps dy and x alive until then."""
    fused_bias = bname is not None and rowscale is None   # db = colsum(dy) rides in the TN kernel
    call = _wgrad_call(st, dy, x, wname, bname if fused_bias else None)
    if group is not None:
        group.append(call)
    else:
        ops.gemm(call[0], call[1], call[2], **call[3])
    if bname is not None and not fused_bias:
        ops.colsum(dy, st.grad(bname), rowscale=rowscale, rows_per_group=rpg)


# A block's grouped weight-gradient launch depends on nothing that follows it in the backward chain:
# it runs on a side stream (own slab workspace), so its workgroups and the dgrad / LayerNorm /
# attention kernels of the next block fill each other's partly used tile rounds (55.9 -> 55.1 ms).
# DINO_WG_STREAM=0 keeps the backward on one stream.
_WG_STREAM = _os.environ.get("DINO_WG_STREAM", "1") != "0"
# bf16: LayerNorm folded into the GEMMs around it (no LayerNorm pass over the token matrix in the forward);
# DINO_LN_FOLD=0 keeps the separate kernel (A/B runs)
_LN_FOLD = _os.environ.get("DINO_LN_FOLD", "1") != "0"
# DINO_LN_FOLD_EPI=1: no finalize launch, the consumer GEMM's epilogue forms rstd from the partials itself (D <= 384)
_LN_FOLD_EPI = _os.environ.get("DINO_LN_FOLD_EPI", "0") == "1"

# The last block on its CLS rows only: VisionTransformer.forward returns x[:, 0] (vision_transformer.py:382-383), so of
# the last block's attention output, proj, norm2 and MLP only the CLS rows are ever consumed, and only they carry a
# gradient -- the reference computes the other rows and drops them; every result (loss, all parameter gradients) is
# the same without them.  K and V of the last block still need every token (qkv GEMM, its dgrad / wgrad, norm1 and its
# backward stay whole).  Not when a caller asks for the block outputs (trace / get_intermediate_layers).
# DINO_CLS_LAST=0 computes the whole last block (A/B runs).
_CLS_LAST = _os.environ.get("DINO_CLS_LAST", "1") != "0"

_FOLD_OK = {}


def _stat_out(save: bool, M: int, device):
    """(mean, rstd) buffers a folded consumer GEMM fills for the LayerNorm backward; none without a tape."""
    if not save:
        return None, None
    return (torch.empty(M, dtype=torch.float32, device=device), torch.empty(M, dtype=torch.float32, device=device))


def _fold_supported(st: ParamStore, pre: str, M: int, D: int, hid: int) -> bool:
    """Whether the four GEMMs around a block's two LayerNorms run on the wide-tile kernel at this token count
    (asked once per shape; the kernel's own predicate decides)."""
    key = (str(st.device), M, D, hid)
    if key not in _FOLD_OK:
        b0, dt, dev = f"{pre}blocks.0.", st.dtype, st.device
        x = torch.empty((M, D), dtype=dt, device=dev)
        e_q = torch.empty((M, 3 * D), dtype=dt, device=dev)
        e_h = torch.empty((M, hid), dtype=dt, device=dev)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        stats = torch.empty(ops.ln_stats_parts(D) * M * 2, dtype=torch.float32, device=dev)
        _FOLD_OK[key] = bool(
            ops.gemm_wide_tile_supported(x, st.wfold(b0 + "attn.qkv.weight")[0], e_q,
                                         bias=st.p32(b0 + "attn.qkv.bias"), prescale=rstd)
            and (not _LN_FOLD_EPI or ops.gemm_wide_tile_supported(
                x, st.wfold(b0 + "attn.qkv.weight")[0], e_q, bias=st.p32(b0 + "attn.qkv.bias"), ln_stats_in=stats,
                ln_dim=D))
            and ops.gemm_wide_tile_supported(x, st.wfold(b0 + "mlp.fc1.weight")[0], e_h,
                                             bias=st.p32(b0 + "mlp.fc1.bias"), act=3, aux=e_h,
                                             **(dict(ln_stats_in=stats, ln_dim=D) if _LN_FOLD_EPI
                                                else dict(prescale=rstd)))
            and ops.gemm_wide_tile_supported(x, st.w(b0 + "attn.proj.weight"), x, residual=x,
                                             bias=st.p32(b0 + "attn.proj.bias"), ln_stats=stats)
            and ops.gemm_wide_tile_supported(e_h, st.w(b0 + "mlp.fc2.weight"), x, residual=x,
                                             bias=st.p32(b0 + "mlp.fc2.bias"), ln_stats=stats))
    return _FOLD_OK[key]


_TAPS_CACHE = {}


def _taps(device, src_side, dst_side):
    key = (str(device), src_side, dst_side)
    if key not in _TAPS_CACHE:
        idx, w = bicubic_taps(src_side, dst_side)
        _TAPS_CACHE[key] = (torch.from_numpy(idx).to(device), torch.from_numpy(w).to(device))
    return _TAPS_CACHE[key]


_KEEP_CACHE = {}


def _drop_scales(spec: VitSpec, Bc, device, training, row_crop=None):
    """Per block (scale_attn, scale_mlp) fp32 [Bc] = floor(keep + U[0,1)) / keep
    (drop_path, vision_transformer.py:52-62); None where the rate is 0 or in eval.  One kernel draws the
    whole depth (dino_drop_path_scales; seeded from torch's CPU generator, so fix_random_seeds governs it);
    with row_crop (int64 [M]) it also returns {block: (rows_attn, rows_mlp)}, the same values per token row."""
    rates = [0.0 if not training else float(r) for r in spec.drop_path]
    live = [l for l, r in enumerate(rates) if r != 0.0]
    out = [(None, None)] * len(rates)
    rowout = None
    if live:
        key = (tuple(rates), str(device))
        if key not in _KEEP_CACHE:
            _KEEP_CACHE[key] = torch.tensor([1.0 - rates[l] for l in live], dtype=torch.float32, device=device)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        s, rows = ops.drop_path_scales(seed, _KEEP_CACHE[key], Bc, row_crop)
        for j, l in enumerate(live):
            out[l] = (s[j, 0], s[j, 1])
        if rows is not None:
            rowout = {l: (rows[j, 0], rows[j, 1]) for j, l in enumerate(live)}
    return out, rowout


_ROWMAP_CACHE = {}


def _row_to_crop(device, shape_key):
    """int64 [M]: the crop index (over all groups) of every token row; shape_key = ((Bc, N), ...)."""
    key = (str(device), shape_key)
    if key not in _ROWMAP_CACHE:
        parts, c0 = [], 0
        for Bc, N in shape_key:
            parts.append(torch.arange(c0, c0 + Bc, dtype=torch.int64).repeat_interleave(N))
            c0 += Bc
        _ROWMAP_CACHE[key] = torch.cat(parts).to(device)
    return _ROWMAP_CACHE[key]


def _as_groups(images):
    """Normalise the input to a list of resolution groups, each a list of fp32 contiguous crops."""
    if isinstance(images, torch.Tensor):
        images = [images]
    if all(isinstance(im, torch.Tensor) for im in images):
        same = all(im.shape[-2:] == images[0].shape[-2:] for im in images)
        groups = [list(images)] if same else [[im] for im in images]
    else:
        groups = [[g] if isinstance(g, torch.Tensor) else list(g) for g in images]
    return [[im.float().contiguous() for im in g] for g in groups]   # PatchEmbed casts to fp32 (:243)


def vit_forward(st: ParamStore, pre: str, spec: VitSpec, images, *, save: bool,
                training: bool = False, drops=None, trace=None, last_attention: bool = False):
    """images: fp32 [Bc,3,S,S] on the device, a list of such crops of ONE resolution, or a list of
    resolution groups (each a tensor or a list of crops).  Returns (cls [Bc_total,D] in st.dtype,
    tape|None), crops in the order given.

    ALL groups run as ONE token matrix [sum_g Bc_g*N_g, D] through every LayerNorm and GEMM (the
    reference runs one backbone pass per resolution, models/__init__.py:36-44; rows are independent
    there, so stacking them changes nothing but the launch count, the fill of the last tile round and
    the number of weight-gradient launches); only attention, the token assembly and the final CLS norm
    are per group.  drops: per block (scale_attn, scale_mlp) fp32 [Bc_total].  trace (optional): a dict
    for one group, or a list of dicts (one per group).
    last_attention (no-grad callers only): stop inside the last block and return its attention
    probabilities [Bc,H,N,N] instead (get_last_selfattention, vision_transformer.py:385-392)."""
    if last_attention and save:
        raise ValueError("last_attention is an inference path (no tape)")
    dt, D, P, H = st.dtype, spec.embed_dim, spec.patch_size, spec.num_heads
    groups = _as_groups(images)
    ng = len(groups)
    if last_attention and ng > 1:
        raise ValueError("last_attention takes crops of one resolution")
    traces = [None] * ng if trace is None else (trace if isinstance(trace, list) else [trace])
    geo = []                                   # per group: (Bc, N, row0, gh)
    row0 = 0
    for g in groups:
        Bc, (_, _, Hi, Wi) = sum(im.shape[0] for im in g), g[0].shape
        gh, gw = Hi // P, Wi // P
        if gh != gw:
            raise ValueError("non-square crops are not supported by the pos-embed resize")
        geo.append((Bc, gh * gw + 1, row0, gh))
        row0 += Bc * (gh * gw + 1)
    M, Bct = row0, sum(x[0] for x in geo)
    # --- tokens: im2col GEMM (+bias +pos), CLS rows ---------------------------------------
    pos = st.p32(pre + "pos_embed")[0]                                  # fp32 [1+Ns, D]
    n_src = pos.shape[0] - 1
    x = torch.empty((M, D), dtype=dt, device=st.device)
    Kp = (3 * P * P + 7) // 8 * 8
    cols_all = torch.empty((M, Kp), dtype=dt, device=st.device) if save else None
    taps_all = []
    for gi, g in enumerate(groups):
        Bc, N, r0, gh = geo[gi]
        cols = ops.im2col(g, P, dt, cls_row=True,
                          out=None if cols_all is None else cols_all[r0:r0 + Bc * N])   # [Bc*N, Kp], CLS rows zero
        n = N - 1
        taps = (None, None) if n == n_src else _taps(st.device, int(math.isqrt(n_src)), gh)
        taps_all.append(taps)
        pos_t = ops.pos_table(pos, taps[0], taps[1], n, dt)             # [N, D]
        xg = x[r0:r0 + Bc * N]
        ops.gemm(cols, st.patch_weight(pre + "patch_embed.proj.weight", Kp), xg,
                 bias=st.p32(pre + "patch_embed.proj.bias"), residual=pos_t, res_mod=N)
        ops.cls_rows(xg, st.p32(pre + "cls_token").reshape(-1), pos[0], Bc, N)
        if traces[gi] is not None:
            traces[gi]["tokens"] = xg.reshape(Bc, N, D)
    # drop-path scales per token row when several groups (different tokens per crop) share the matrix
    N0 = geo[0][1]
    rowdrops = None
    if drops is None:
        idx = _row_to_crop(st.device, tuple((b_, n_) for b_, n_, _, _ in geo)) if ng > 1 else None
        drops, rowdrops = _drop_scales(spec, Bct, st.device, training, idx)
    live = [l for l in range(spec.depth) if drops[l][0] is not None or drops[l][1] is not None]
    if ng > 1 and live and rowdrops is None:     # scales given by the caller (tests): expand them here
        idx = _row_to_crop(st.device, tuple((b_, n_) for b_, n_, _, _ in geo))
        stack = torch.stack([torch.stack([drops[l][0], drops[l][1]]) for l in live])      # [L', 2, Bct]
        rows = stack.index_select(2, idx).contiguous()                                     # [L', 2, M]
        rowdrops = {l: (rows[j, 0], rows[j, 1]) for j, l in enumerate(live)}

    def rs(l, which):
        """(rowscale tensor, rows_per_group) for the GEMM / LN epilogues of block l."""
        d = drops[l][which]
        if d is None:
            return None, 0
        return (d, N0) if ng == 1 else (rowdrops[l][which], 1)
    tape = Tape({"pre": pre, "spec": spec, "geo": geo, "M": M, "cols": cols_all, "taps": taps_all,
                 "blocks": [], "drops": drops, "rs": rs}) if save else None
    if save:
        st.live_tapes.add(tape)
    # LayerNorm folded into the GEMMs (bf16, token matrices large enough for the wide-tile kernel, which carries
    # the two epilogues): the GEMM that WRITES the residual stream (proj / fc2 + residual) leaves per-row partial
    # (sum, sum of squares); the GEMM that READS it (qkv / fc1) takes the raw stream, the folded weight
    # gamma * W - rowmean(gamma * W) and a per-row rstd in its epilogue.  No normalised copy is written or read in
    # the forward pass; the backward pass gets it from the LayerNorm backward kernel (vit_backward).
    hid = spec.mlp_hidden
    fold = (_LN_FOLD and dt == torch.bfloat16 and not last_attention
            and st.wfold(f"{pre}blocks.0.attn.qkv.weight") is not None and _fold_supported(st, pre, M, D, hid))
    nparts = ops.ln_stats_parts(D)
    stats = torch.empty(nparts * M * 2, dtype=torch.float32, device=st.device) if fold else None
    if save:
        tape["fold"] = fold
    cls_last = _CLS_LAST and not last_attention and all(t_ is None for t_ in traces)
    for l in range(spec.depth):
        b = f"{pre}blocks.{l}."
        sa, rpa = rs(l, 0)
        sm, rpm = rs(l, 1)
        qkv = torch.empty((M, 3 * D), dtype=dt, device=st.device)
        if fold:
            ln1 = None
            wq, bq = st.wfold(b + "attn.qkv.weight")
            if l == 0:      # the token assembly's rows: statistics by their own small kernel
                mean1, rstd1 = ops.layernorm_stats(x, spec.eps)
                ops.gemm(x, wq, qkv, bias=bq, prescale=rstd1)
            elif _LN_FOLD_EPI:   # rstd formed from the partials in this GEMM's epilogue (measured: slower, see DESIGN)
                mean1, rstd1 = _stat_out(save, M, st.device)
                ops.gemm(x, wq, qkv, bias=bq, ln_stats_in=stats, ln_dim=D, ln_eps=spec.eps,
                         ln_mean_out=mean1, ln_rstd_out=rstd1)
            else:           # the previous block's fc2 left the partials
                mean1, rstd1 = ops.ln_stats_finalize(stats, nparts, M, D, spec.eps)
                ops.gemm(x, wq, qkv, bias=bq, prescale=rstd1)
        else:
            ln1, mean1, rstd1 = ops.layernorm_fwd(x, st.p32(b + "norm1.weight"), st.p32(b + "norm1.bias"),
                                                  spec.eps)
            ops.gemm(ln1, st.w(b + "attn.qkv.weight"), qkv, bias=st.p32(b + "attn.qkv.bias"))
        if last_attention and l == spec.depth - 1:
            return ops.attn_probs(qkv, geo[0][0], geo[0][1], H), None
        if cls_last and l == spec.depth - 1:
            # ---- last block, CLS rows only: one query per (crop, head), then [Bct, D] matrices -----------
            da_, dm_ = drops[l]
            xc = torch.empty((Bct, D), dtype=dt, device=st.device)
            ao = torch.empty((Bct, D), dtype=dt, device=st.device)
            lses, c0 = [], 0
            for Bc, N, r0, _ in geo:
                xc[c0:c0 + Bc].copy_(x[r0:r0 + Bc * N].view(Bc, N, D)[:, 0])
                _, lse = ops.attn_cls_fwd(qkv[r0:r0 + Bc * N], Bc, N, H,
                                          rowscale=None if da_ is None else da_[c0:c0 + Bc].contiguous(),
                                          out=ao[c0:c0 + Bc])
                lses.append(lse)
                c0 += Bc
            x1 = torch.empty((Bct, D), dtype=dt, device=st.device)
            ops.gemm(ao, st.w(b + "attn.proj.weight"), x1, bias=st.p32(b + "attn.proj.bias"), residual=xc,
                     rowscale=da_, rows_per_group=1 if da_ is not None else 0, bias_rowscale=da_ is not None)
            ln2, mean2, rstd2 = ops.layernorm_fwd(x1, st.p32(b + "norm2.weight"), st.p32(b + "norm2.bias"), spec.eps)
            h = torch.empty((Bct, hid), dtype=dt, device=st.device)
            pact = torch.empty((Bct, hid), dtype=dt, device=st.device) if save else None
            act1 = 3 if (dt == torch.bfloat16 and save) else 1
            ops.gemm(ln2, st.w(b + "mlp.fc1.weight"), h, bias=st.p32(b + "mlp.fc1.bias"), act=act1, aux=pact,
                     rowscale=dm_, rows_per_group=1 if dm_ is not None else 0, out_rowscale=dm_ is not None)
            x2 = torch.empty((Bct, D), dtype=dt, device=st.device)
            ops.gemm(h, st.w(b + "mlp.fc2.weight"), x2, bias=st.p32(b + "mlp.fc2.bias"), residual=x1,
                     rowscale=dm_, rows_per_group=1 if dm_ is not None else 0, bias_rowscale=dm_ is not None)
            if save:
                tape["blocks"].append(dict(cls=True, x=x, mean1=mean1, rstd1=rstd1, ln1=ln1, qkv=qkv, ao=ao, lse=lses,
                                           x1=x1, mean2=mean2, rstd2=rstd2, ln2=ln2, pre=pact, h=h))
            cls, meanf, rstdf = ops.layernorm_fwd(x2, st.p32(pre + "norm.weight"), st.p32(pre + "norm.bias"), spec.eps)
            if save:
                tape.update(xf=x2, fin=[(meanf, rstdf)], cls_last=True)
            return cls, tape
        ao = torch.empty((M, D), dtype=dt, device=st.device)
        lses, c0 = [], 0
        for Bc, N, r0, _ in geo:
            da = drops[l][0]
            _, lse = ops.attn_fwd(qkv[r0:r0 + Bc * N], Bc, N, H, rowscale=None if da is None else da[c0:c0 + Bc],
                                  out=ao[r0:r0 + Bc * N])
            lses.append(lse)
            c0 += Bc
        x1 = torch.empty((M, D), dtype=dt, device=st.device)
        ops.gemm(ao, st.w(b + "attn.proj.weight"), x1, bias=st.p32(b + "attn.proj.bias"),
                 residual=x, rowscale=sa, rows_per_group=rpa, bias_rowscale=sa is not None, ln_stats=stats)
        h = torch.empty((M, hid), dtype=dt, device=st.device)
        pact = torch.empty((M, hid), dtype=dt, device=st.device) if save else None
        # bf16: the epilogue saves gelu'(pre) rather than pre (act 3 / 4), so the backward epilogue
        # is one multiply; fp32 keeps the pre-activation and the exact-erf derivative (act 1 / 2)
        act1 = 3 if (dt == torch.bfloat16 and save) else 1
        if fold:
            ln2 = None
            w1, b1 = st.wfold(b + "mlp.fc1.weight")
            if _LN_FOLD_EPI:
                mean2, rstd2 = _stat_out(save, M, st.device)
                ops.gemm(x1, w1, h, bias=b1, ln_stats_in=stats, ln_dim=D, ln_eps=spec.eps, ln_mean_out=mean2,
                         ln_rstd_out=rstd2, act=act1, aux=pact,
                         rowscale=sm, rows_per_group=rpm, out_rowscale=sm is not None)
            else:
                mean2, rstd2 = ops.ln_stats_finalize(stats, nparts, M, D, spec.eps)
                ops.gemm(x1, w1, h, bias=b1, prescale=rstd2, act=act1, aux=pact,
                         rowscale=sm, rows_per_group=rpm, out_rowscale=sm is not None)
        else:
            ln2, mean2, rstd2 = ops.layernorm_fwd(x1, st.p32(b + "norm2.weight"), st.p32(b + "norm2.bias"),
                                                  spec.eps)
            ops.gemm(ln2, st.w(b + "mlp.fc1.weight"), h, bias=st.p32(b + "mlp.fc1.bias"), act=act1, aux=pact,
                     rowscale=sm, rows_per_group=rpm, out_rowscale=sm is not None)
        x2 = torch.empty((M, D), dtype=dt, device=st.device)
        ops.gemm(h, st.w(b + "mlp.fc2.weight"), x2, bias=st.p32(b + "mlp.fc2.bias"), residual=x1,
                 rowscale=sm, rows_per_group=rpm, bias_rowscale=sm is not None,
                 ln_stats=stats if l + 1 < spec.depth else None)
        if save:
            tape["blocks"].append(dict(x=x, mean1=mean1, rstd1=rstd1, ln1=ln1, qkv=qkv, ao=ao, lse=lses,
                                       x1=x1, mean2=mean2, rstd2=rstd2, ln2=ln2, pre=pact, h=h))
        x = x2
        for gi, (Bc, N, r0, _) in enumerate(geo):
            if traces[gi] is not None:
                traces[gi][f"block{l}"] = x[r0:r0 + Bc * N].reshape(Bc, N, D)
    # final norm on the CLS rows only (:382-383)
    cls = torch.empty((Bct, D), dtype=dt, device=st.device)
    fin, c0 = [], 0
    for Bc, N, r0, _ in geo:
        _, meanf, rstdf = ops.layernorm_fwd(x[r0:r0 + Bc * N], st.p32(pre + "norm.weight"),
                                            st.p32(pre + "norm.bias"), spec.eps, rows=Bc, x_stride=N * D,
                                            out=cls[c0:c0 + Bc], y_stride=D)
        fin.append((meanf, rstdf))
        c0 += Bc
    if save:
        tape.update(xf=x, fin=fin)
    return cls, tape


# Parameter names whose gradient is FINAL after each stage of the last backbone pass's backward (what
# the data-parallel reducer is told, in this order: head, final norm, blocks top-down, token assembly).
# tests/test_distributed_cpu.py walks these lists over the ViT-S/B/L layouts and requires every bucket
# to be launched by them, not by the reducer's end-of-backward flush.
_BLOCK_PARAMS = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                 "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                 "mlp.fc2.weight", "mlp.fc2.bias")


def ready_names_final_norm(pre):
    return [pre + "norm.weight", pre + "norm.bias"]


def ready_names_block(pre, l):
    return [f"{pre}blocks.{l}.{k}" for k in _BLOCK_PARAMS]


def ready_names_embed(pre):
    return [pre + "cls_token", pre + "pos_embed", pre + "patch_embed.proj.weight",
            pre + "patch_embed.proj.bias"]


def ready_names_head(pre, all_names):
    return [k for k in all_names if k.startswith(pre)]


def backward_ready_sequence(all_names, depth, backbone="backbone.", head="head."):
    """The whole announcement order of one step, as lists (a data-parallel dry run without a GPU)."""
    seq = [ready_names_head(head, all_names), ready_names_final_norm(backbone)]
    seq += [ready_names_block(backbone, l) for l in reversed(range(depth))]
    seq.append(ready_names_embed(backbone))
    return seq


def vit_backward(st: ParamStore, tape: dict, dcls: torch.Tensor):
    """Back-propagate d(loss)/d(cls) [Bc_total,D]; parameter grads are accumulated into st.grad_flat."""
    pre, spec, geo, M = tape["pre"], tape["spec"], tape["geo"], tape["M"]
    dt, D, H = st.dtype, spec.embed_dim, spec.num_heads
    rs, drops = tape["rs"], tape["drops"]
    st.live_tapes.discard(tape)
    last = st.pending_bwd == 0   # the final contribution to the shared backbone weights
    dcls = dcls.to(dt).contiguous()
    top = f"{pre}blocks.{spec.depth - 1}."
    cls_last = tape.get("cls_last", False)
    Bct = sum(x_[0] for x_ in geo)
    if cls_last:
        # the last block ran on its CLS rows: the gradient is a [Bct, D] matrix down to that block's attention
        g = torch.empty((Bct, D), dtype=dt, device=st.device)
        meanf, rstdf = tape["fin"][0]
        dlast = drops[-1][1]
        ops.layernorm_bwd(dcls, tape["xf"], st.p32(pre + "norm.weight"), meanf, rstdf, dx=g,
                          dgamma=st.grad(pre + "norm.weight"), dbeta=st.grad(pre + "norm.bias"),
                          dx_colsum=st.grad(top + "mlp.fc2.bias"), rowscale=dlast,
                          rows_per_group=1 if dlast is not None else 0)
    else:
        g = torch.zeros((M, D), dtype=dt, device=st.device)
    # every LN backward also accumulates the column sums of the residual-stream gradient it
    # emits: that is the bias gradient of the linear layer feeding that residual add
    c0 = 0
    for (Bc, N, r0, _), (meanf, rstdf) in zip(geo, [] if cls_last else tape["fin"]):
        dlast = drops[-1][1]
        ops.layernorm_bwd(dcls[c0:c0 + Bc], tape["xf"][r0:r0 + Bc * N], st.p32(pre + "norm.weight"), meanf, rstdf,
                          dx=g[r0:r0 + Bc * N], dgamma=st.grad(pre + "norm.weight"),
                          dbeta=st.grad(pre + "norm.bias"), dx_colsum=st.grad(top + "mlp.fc2.bias"),
                          rowscale=None if dlast is None else dlast[c0:c0 + Bc],
                          rows_per_group=1, rows=Bc, x_stride=N * D, dx_stride=N * D)
        c0 += Bc
    if last:
        st.notify_ready(ready_names_final_norm(pre))
    for l in reversed(range(spec.depth)):
        b = f"{pre}blocks.{l}."
        t = tape["blocks"][l]
        sa, rpa = rs(l, 0)
        sm, rpm = rs(l, 1)
        hid = spec.mlp_hidden
        wg = []  # this block's four weight gradients share the tokens: one split-over-tokens launch
        fold = tape.get("fold", False)
        if t.get("cls"):
            # ---- last block, CLS rows only ([Bct, .] matrices) down to the attention's one query per problem ----
            da_, dm_ = drops[l]
            rp_a, rp_m = (1 if da_ is not None else 0), (1 if dm_ is not None else 0)
            _wgrad(st, g, t["h"], b + "mlp.fc2.weight", None, group=wg)
            dpre = torch.empty((Bct, hid), dtype=dt, device=st.device)
            ops.gemm(g, st.wt(b + "mlp.fc2.weight"), dpre, act=4 if dt == torch.bfloat16 else 2, aux=t["pre"],
                     rowscale=dm_, rows_per_group=rp_m, out_rowscale=dm_ is not None)
            dln2 = torch.empty((Bct, D), dtype=dt, device=st.device)
            ops.gemm(dpre, st.wt(b + "mlp.fc1.weight"), dln2)
            g1c = ops.layernorm_bwd(dln2, t["x1"], st.p32(b + "norm2.weight"), t["mean2"], t["rstd2"], dres=g,
                                    dgamma=st.grad(b + "norm2.weight"), dbeta=st.grad(b + "norm2.bias"),
                                    dx_colsum=st.grad(b + "attn.proj.bias"), rowscale=da_, rows_per_group=rp_a,
                                    workspace=st.workspace)
            _wgrad(st, dpre, t["ln2"], b + "mlp.fc1.weight", b + "mlp.fc1.bias", group=wg)
            _wgrad(st, g1c, t["ao"], b + "attn.proj.weight", None, group=wg)
            dao = torch.empty((Bct, D), dtype=dt, device=st.device)
            ops.gemm(g1c, st.wt(b + "attn.proj.weight"), dao, rowscale=da_, rows_per_group=rp_a,
                     out_rowscale=da_ is not None)
            dqkv = torch.empty((M, 3 * D), dtype=dt, device=st.device)
            c0 = 0
            for (Bc, N, r0, _), lse in zip(geo, t["lse"]):
                sl = slice(r0, r0 + Bc * N)
                ops.attn_cls_bwd(t["qkv"][sl], dao[c0:c0 + Bc], lse, Bc, N, H, dqkv=dqkv[sl])
                c0 += Bc
            # the residual path hands the block input the CLS rows of g1c and nothing else: norm1's backward runs
            # without a residual-gradient operand, the CLS rows are added behind it (cls_resid below)
            g1, cls_resid = None, g1c
            del dpre, dln2, dao
            dln1 = torch.empty((M, D), dtype=dt, device=st.device)
        else:
            # ---- MLP branch: x2 = x1 + fc2(h') + s*bias, h' = s*gelu(pre) ----------------------
            _wgrad(st, g, t["h"], b + "mlp.fc2.weight", None, group=wg)
            dpre = torch.empty((M, hid), dtype=dt, device=st.device)
            ops.gemm(g, st.wt(b + "mlp.fc2.weight"), dpre, act=4 if dt == torch.bfloat16 else 2, aux=t["pre"],
                     rowscale=sm, rows_per_group=rpm, out_rowscale=sm is not None)
            dln2 = torch.empty((M, D), dtype=dt, device=st.device)
            ops.gemm(dpre, st.wt(b + "mlp.fc1.weight"), dln2)
            if fold:
                # the forward pass never wrote LayerNorm's output: the backward kernel, which recomputes it from x
                # anyway, leaves it for the weight gradient of fc1 (launched with the block's other three below)
                xn2 = torch.empty((M, D), dtype=dt, device=st.device)
                g1 = ops.layernorm_bwd_xn(dln2, t["x1"], st.p32(b + "norm2.weight"), st.p32(b + "norm2.bias"),
                                          t["mean2"], t["rstd2"], xn=xn2, dres=g,
                                          dgamma=st.grad(b + "norm2.weight"), dbeta=st.grad(b + "norm2.bias"),
                                          dx_colsum=st.grad(b + "attn.proj.bias"), rowscale=sa, rows_per_group=rpa,
                                          workspace=st.workspace)
            else:
                xn2 = t["ln2"]
                g1 = ops.layernorm_bwd(dln2, t["x1"], st.p32(b + "norm2.weight"), t["mean2"], t["rstd2"],
                                       dres=g, dgamma=st.grad(b + "norm2.weight"),
                                       dbeta=st.grad(b + "norm2.bias"),
                                       dx_colsum=st.grad(b + "attn.proj.bias"), rowscale=sa, rows_per_group=rpa,
                                       workspace=st.workspace)
            # fc1's bias gradient = column sums of dpre: carried by the weight-gradient kernel
            _wgrad(st, dpre, xn2, b + "mlp.fc1.weight", b + "mlp.fc1.bias", group=wg)
            del dpre
            # ---- attention branch ------------------------------------------------------
            _wgrad(st, g1, t["ao"], b + "attn.proj.weight", None, group=wg)
            dao = dln2  # reuse
            ops.gemm(g1, st.wt(b + "attn.proj.weight"), dao, rowscale=sa, rows_per_group=rpa,
                     out_rowscale=sa is not None)
            dqkv = torch.empty((M, 3 * D), dtype=dt, device=st.device)
            c0 = 0
            for (Bc, N, r0, _), lse in zip(geo, t["lse"]):
                da = drops[l][0]
                sl = slice(r0, r0 + Bc * N)
                ops.attn_bwd(t["qkv"][sl], t["ao"][sl], dao[sl], lse, Bc, N, H,
                             rowscale=None if da is None else da[c0:c0 + Bc], dqkv=dqkv[sl])
                c0 += Bc
            dln1 = dao
        ops.gemm(dqkv, st.wt(b + "attn.qkv.weight"), dln1)
        below = f"{pre}blocks.{l - 1}.mlp.fc2.bias" if l > 0 else None
        rb, rpb = (None, 0) if l == 0 else rs(l - 1, 1)
        if fold:
            xn1 = torch.empty((M, D), dtype=dt, device=st.device)
            g = ops.layernorm_bwd_xn(dln1, t["x"], st.p32(b + "norm1.weight"), st.p32(b + "norm1.bias"),
                                     t["mean1"], t["rstd1"], xn=xn1, dres=g1,
                                     dgamma=st.grad(b + "norm1.weight"), dbeta=st.grad(b + "norm1.bias"),
                                     dx_colsum=None if below is None else st.grad(below),
                                     rowscale=rb, rows_per_group=rpb, workspace=st.workspace)
        else:
            xn1 = t["ln1"]
            g = ops.layernorm_bwd(dln1, t["x"], st.p32(b + "norm1.weight"), t["mean1"], t["rstd1"],
                                  dres=g1, dgamma=st.grad(b + "norm1.weight"),
                                  dbeta=st.grad(b + "norm1.bias"),
                                  dx_colsum=None if below is None else st.grad(below),
                                  rowscale=rb, rows_per_group=rpb, workspace=st.workspace)
        if t.get("cls"):
            c0 = 0
            for Bc, N, r0, _ in geo:
                g[r0:r0 + Bc * N].view(Bc, N, D)[:, 0].add_(cls_resid[c0:c0 + Bc])
                c0 += Bc
            if below is not None:     # ... and their share of the bias gradient that rides on norm1's backward
                dmb = drops[l - 1][1]
                ops.colsum(cls_resid, st.grad(below), rowscale=dmb, rows_per_group=1 if dmb is not None else 0)
            del cls_resid
        _wgrad(st, dqkv, xn1, b + "attn.qkv.weight", b + "attn.qkv.bias", group=wg)
        del dqkv
        if _WG_STREAM:
            if st.wg_stream is None:
                st.wg_stream = side_stream(st.device)
            cur = torch.cuda.current_stream()
            st.wg_stream.wait_stream(cur)
            with torch.cuda.stream(st.wg_stream):
                ops.gemm_group(wg)
            # The operands belong to the main stream's allocator pool.  Their references are held
            # until the main stream has joined the side stream (end of this backward pass, or the
            # block's hand-over to the data-parallel reducer).  Tried and dropped: record_stream on
            # the operands (two of five bench runs fell into a 70-78 ms mode, tools/variance.sh),
            # joining one or three blocks late (+1 ms: the main stream stalls behind the side one).
            st.wg_pending.append(wg)
        else:
            ops.gemm_group(wg)
        del wg
        tape["blocks"][l] = None  # free the saved activations of this block
        if last:
            if _WG_STREAM and st.ready_cb is not None and st.wg_stream is not None:
                # data parallel: this block's gradients are about to be handed to the all-reduce
                torch.cuda.current_stream().wait_stream(st.wg_stream)
                st.wg_pending.clear()
            # (block l's own fc2.bias was completed by the LN backward above this block)
            st.notify_ready(ready_names_block(pre, l))
    # ---- token assembly: x0[b,i] = patch(b,i) + pos_t[i]; x0[b,0] = cls + pos_t[0] ---------
    dpos = st.grad(pre + "pos_embed")[0]
    for (Bc, N, r0, _), taps in zip(geo, tape["taps"]):
        fold = torch.zeros((N, D), dtype=torch.float32, device=st.device)
        ops.fold_rows(g[r0:r0 + Bc * N], fold, Bc, N)                   # sum over crops
        ops.pos_table_bwd(fold, taps[0], taps[1], dpos, N - 1)
        ops.pos_table_bwd(fold[0:1], None, None, st.grad(pre + "cls_token").reshape(1, D), 0)
        ops.colsum(fold[1:], st.grad(pre + "patch_embed.proj.bias"))    # CLS rows carry no conv bias
    cols = tape["cols"]
    gw = st.grad(pre + "patch_embed.proj.weight")
    kk = gw[0].numel()
    if cols.shape[1] == kk:
        ops.gemm(g, cols, gw.reshape(D, kk), trans=True, out_mode=2,
                 splitk=_wgrad_splitk(D, kk, M))
    else:  # K padded (P = 14): accumulate into a padded scratch, then add the valid columns
        tmp = torch.zeros((D, cols.shape[1]), dtype=torch.float32, device=st.device)
        ops.gemm(g, cols, tmp, trans=True, out_mode=2, splitk=_wgrad_splitk(D, cols.shape[1], M))
        gw.reshape(D, kk).add_(tmp[:, :kk])
    if _WG_STREAM and st.wg_stream is not None:
        torch.cuda.current_stream().wait_stream(st.wg_stream)
        st.wg_pending.clear()
    if last:
        st.notify_ready(ready_names_embed(pre))


# ------------------------------------------------------------------------------------ head
@dataclass
class HeadSpec:
    nlayers: int
    mlp_names: List[str]      # ["mlp.0", "mlp.2", "mlp.4"] or ["mlp"]


def head_forward(st: ParamStore, pre: str, spec: HeadSpec, x: torch.Tensor, *, save: bool,
                 trace: Optional[dict] = None, colsum_out: Optional[torch.Tensor] = None, lazy: bool = False):
    """x: [R, in_dim] (st.dtype).  Returns (logits [R, K], tape|None).  colsum_out (fp32 [K], zeroed):
    the prototype GEMM's epilogue adds the column sums of the logits into it -- the batch half of
    DINOLoss.update_center (loss.py:69) for the teacher, without another pass over the logits.
    lazy: stop before the prototype layer and return the normalised bottleneck [R, Db] instead -- the
    logits are then formed chunk by chunk inside the loss (proto_loss_forward / _backward below)."""
    dt = st.dtype
    x = x.to(dt).contiguous()
    R = x.shape[0]
    acts, pres = [x], []
    for j, nm in enumerate(spec.mlp_names):
        w = st.w(f"{pre}{nm}.weight")
        lastl = j == len(spec.mlp_names) - 1
        y = torch.empty((R, w.shape[0]), dtype=dt, device=st.device)
        if lastl:
            ops.gemm(acts[-1], w, y, bias=st.p32(f"{pre}{nm}.bias"))
            pres.append(None)
        else:
            pa = torch.empty_like(y) if save else None
            ops.gemm(acts[-1], w, y, bias=st.p32(f"{pre}{nm}.bias"), act=1, aux=pa)
            pres.append(pa)
        acts.append(y)
    if trace is not None:
        trace["bottleneck"] = acts[-1]
    yn, nrm = ops.l2norm_fwd(acts[-1])
    tape = dict(pre=pre, spec=spec, acts=acts, pres=pres, yn=yn, nrm=nrm, lazy=lazy) if save else None
    if lazy:
        return yn, tape
    W, _, _ = st.wn(pre + "last_layer.")
    logits = torch.empty((R, W.shape[0]), dtype=dt, device=st.device)
    ops.gemm(yn, W, logits, colsum=colsum_out)
    return logits, tape


def _proto_dyn(st: ParamStore, dlogits: torch.Tensor, Wt: torch.Tensor, dyn: Optional[torch.Tensor]):
    """dyn (+)= dlogits [R, Kc] @ W [Kc, Db] (Wt = W^T, possibly a column slice); dyn None: a new fp32 tensor."""
    R, K = dlogits.shape
    Db = Wt.shape[0]
    tiles = ((R + 127) // 128) * ((Db + 127) // 128)
    # the slab form exists in the bf16 LDS-DMA NT kernel only (dino_gemm returns DINO_ERR_UNSUPPORTED elsewhere):
    # operands below 2 GiB (its buffer descriptors), K a multiple of its 32-deep stage, kernel not switched off
    slab_ok = (dlogits.dtype == torch.bfloat16 and K % 32 == 0 and "DINO_GEMM_V1" not in _os.environ
               and (R * dlogits.stride(0) + K) * 2 < (1 << 31) and (Db * Wt.stride(0) + K) * 2 < (1 << 31))
    if slab_ok:
        # the contraction runs over the prototypes: k-splits leave their partial products in slabs (plain
        # stores) and one small kernel adds them -- atomics cost 50 us per split here
        sk = max(1, min(K // 1024, 512 // (-(-R // 256) * -(-Db // 128))))   # one round of 2 x 256 workgroups
        slabs = torch.empty((sk, R, Db), dtype=torch.float32, device=st.device)
        ops.gemm(dlogits, Wt, slabs, out_mode=3, splitk=sk)
        if dyn is None:
            return ops.sum_slabs(slabs, torch.empty((R, Db), dtype=torch.float32, device=st.device))
        return ops.sum_slabs(slabs, dyn, accumulate=True)
    if dyn is None:
        dyn = torch.zeros((R, Db), dtype=torch.float32, device=st.device)
    ops.gemm(dlogits, Wt, dyn, out_mode=2, splitk=max(1, min(K // 1024, -(-1024 // tiles))))
    return dyn


def head_backward(st: ParamStore, tape: dict, dlogits: torch.Tensor) -> torch.Tensor:
    """Returns d(loss)/d(x) [R, in_dim]; parameter grads accumulate into st.grad_flat.  A lazy tape
    (head_forward(lazy=True)) takes d(loss)/d(yn) [R, Db] fp32 instead of dlogits."""
    pre, spec = tape["pre"], tape["spec"]
    dt = st.dtype
    yn = tape["yn"]
    if tape.get("lazy"):
        return _head_mlp_backward(st, tape, dlogits.float().contiguous())
    dlogits = dlogits.contiguous()
    R, K = dlogits.shape
    W, Wt, vnorm = st.wn(pre + "last_layer.")
    Db = yn.shape[1]
    # prototype layer: dW = dlogits^T yn (contraction over rows), dyn = dlogits W (over K)
    dW = torch.zeros((K, Db), dtype=torch.float32, device=st.device)
    ops.gemm(dlogits, yn, dW, trans=True, out_mode=2, splitk=_wgrad_splitk(K, Db, R))
    gname, vname = pre + "last_layer.weight_g", pre + "last_layer.weight_v"
    g_p, v_p = st.params[gname], st.params[vname]
    ops.weightnorm_bwd(dW, g_p.data, v_p.data, vnorm,
                       st.grad(gname) if g_p.requires_grad else None, st.grad(vname))
    del dW
    dyn = _proto_dyn(st, dlogits, Wt, None)
    return _head_mlp_backward(st, tape, dyn)


def _head_mlp_backward(st: ParamStore, tape: dict, dyn: torch.Tensor) -> torch.Tensor:
    pre, spec = tape["pre"], tape["spec"]
    dt = st.dtype
    yn = tape["yn"]
    R = yn.shape[0]
    d = ops.l2norm_bwd(dyn, yn, tape["nrm"])
    acts, pres = tape["acts"], tape["pres"]
    ops.colsum(d, st.grad(f"{pre}{spec.mlp_names[-1]}.bias"))
    for j in reversed(range(len(spec.mlp_names))):
        nm = spec.mlp_names[j]
        wname = f"{pre}{nm}.weight"
        _wgrad(st, d, acts[j], wname, None)
        wt = st.wt(wname)
        dx = torch.empty((R, wt.shape[0]), dtype=dt, device=st.device)
        if j > 0:  # dgrad * gelu'(pre) -> grad wrt the previous pre-activation; its column sums
            ops.gemm(d, wt, dx, act=2, aux=pres[j - 1],     # are the previous layer's bias grad
                     colsum=st.grad(f"{pre}{spec.mlp_names[j - 1]}.bias"))
        else:
            ops.gemm(d, wt, dx)
        d = dx
    st.notify_ready(ready_names_head(pre, st.params))
    return d


# ------------------------------------------------------------------ prototype layer + loss, chunked
_PROTO_CHUNK = int(_os.environ.get("DINO_PROTO_CHUNK", "8192"))


def _proto_chunks(K: int):
    kc = max(32, min(K, _PROTO_CHUNK) // 32 * 32)
    return [(k0, min(kc, K - k0)) for k0 in range(0, K, kc)]


def proto_loss_forward(st: ParamStore, pre: str, yn: torch.Tensor, teacher: torch.Tensor, center: torch.Tensor,
                       B: int, ncrops: int, student_temp: float, teacher_temp: float):
    """DINOHead.last_layer (vision_transformer.py:719) + DINOLoss.forward
    (loss.py:42-60) without the [ncrops*B, K] logits.  The K prototypes are walked in chunks: logits of one
    chunk (GEMM into a reused [R, Kc] buffer that stays cache-resident), running log-sum-exp and teacher dot
    product per row, next chunk.  Returns (per-row loss terms [R] fp32, ctx for proto_loss_backward)."""
    W, Wt, vnorm = st.wn(pre + "last_layer.")
    K = W.shape[0]
    R = yn.shape[0]
    dev = st.device
    lse_t = torch.empty((2 * B,), dtype=torch.float32, device=dev)
    run = torch.empty((R, 2), dtype=torch.float32, device=dev)
    parts = torch.empty((R,), dtype=torch.float32, device=dev)
    center = center.clone()    # update_center changes the module's buffer before the backward walk reads it
    args = dict(teacher=teacher, center=center, teacher_lse=lse_t, run=run, B=B, ncrops=ncrops, K=K,
                student_temp=student_temp, teacher_temp=teacher_temp)
    ops.proto_loss_chunk(0, None, 0, 0, **args)
    chunks = _proto_chunks(K)
    buf = torch.empty((R, chunks[0][1]), dtype=yn.dtype, device=dev)
    for k0, kc in chunks:
        ops.gemm(yn, W[k0:k0 + kc], buf, N=kc)
        ops.proto_loss_chunk(1, buf, k0, kc, **args)
    ops.proto_loss_chunk(2, None, 0, 0, parts=parts, **args)
    return parts, dict(pre=pre, yn=yn, args=args, buf=buf)


def proto_loss_backward(st: ParamStore, ctx: dict, gout: Optional[torch.Tensor]) -> torch.Tensor:
    """Second walk over the chunks: logits recomputed, overwritten by d loss / d logits, consumed by the two
    prototype-gradient GEMMs (dW rows of the chunk; dyn accumulated over the chunks).  Returns dyn [R, Db] fp32;
    the weight-norm gradients go to st.grad_flat."""
    pre, yn, args, buf = ctx["pre"], ctx["yn"], ctx["args"], ctx["buf"]
    W, Wt, vnorm = st.wn(pre + "last_layer.")
    K, Db = W.shape
    R = yn.shape[0]
    dW = torch.zeros((K, Db), dtype=torch.float32, device=st.device)
    dyn = None
    for k0, kc in _proto_chunks(K):
        ops.gemm(yn, W[k0:k0 + kc], buf, N=kc)
        ops.proto_loss_chunk(3, buf, k0, kc, gout=gout, **args)
        ds = buf[:, :kc]
        ops.gemm(ds, yn, dW[k0:k0 + kc], trans=True, out_mode=2, splitk=_wgrad_splitk(kc, Db, R))
        dyn = _proto_dyn(st, ds, Wt[:, k0:k0 + kc], dyn)
    gname, vname = pre + "last_layer.weight_g", pre + "last_layer.weight_v"
    g_p, v_p = st.params[gname], st.params[vname]
    ops.weightnorm_bwd(dW, g_p.data, v_p.data, vnorm,
                       st.grad(gname) if g_p.requires_grad else None, st.grad(vname))
    return dyn
