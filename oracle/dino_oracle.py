"""CPU restatement of the DINO training step (TEST INFRASTRUCTURE ONLY).

Pure-PyTorch, functional (state-dict in, tensors out) restatement of the
reference hot path.  Each function cites the reference lines it follows
(paths relative to the reference repo).  It is pinned against fixtures made by
importing the reference itself (tools/make_golden.py -> tests/golden/*.npz,
checked by tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
this module -- never the product path (see oracle/__init__.py).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# architecture description
# ----------------------------------------------------------------------------
@dataclass
class VitCfg:
    """Hyper-parameters of dino/models/vision_transformer.py:251-320."""
    img_size: int = 256
    patch_size: int = 16
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    mlp_ratio: float = 4.0
    ln_eps: float = 1e-6            # partial(nn.LayerNorm, eps=1e-6), :419,439,459,479
    max_crop_scale: float = 0.875   # dino_max_crop_scale, :256,275

    @property
    def num_patches(self) -> int:
        return int(self.img_size * self.max_crop_scale // self.patch_size) ** 2


@dataclass
class HeadCfg:
    """DINOHead ctor arguments, dino/models/vision_transformer.py:676-708."""
    in_dim: int = 384
    out_dim: int = 65536
    hidden_dim: int = 2048
    bottleneck_dim: int = 256
    nlayers: int = 3


ARCHS = {  # dino/models/vision_transformer.py:405-482
    "vit_tiny": dict(embed_dim=192, depth=12, num_heads=3),
    "vit_small": dict(embed_dim=384, depth=12, num_heads=6),
    "vit_base": dict(embed_dim=768, depth=12, num_heads=12),
    "vit_large": dict(embed_dim=1024, depth=24, num_heads=16),
}


def param_shapes(vit: VitCfg, head: HeadCfg) -> "Dict[str, tuple]":
    """Ordered name -> shape of MultiCropWrapper(vit, DINOHead).parameters().

    Order follows module registration in vision_transformer.py:281-311 (cls_token,
    pos_embed, patch_embed, blocks, norm) and 686-705 (mlp, last_layer g then v).
    """
    D, P = vit.embed_dim, vit.patch_size
    Hd = int(D * vit.mlp_ratio)
    s = {}
    s["backbone.cls_token"] = (1, 1, D)
    s["backbone.pos_embed"] = (1, vit.num_patches + 1, D)
    s["backbone.patch_embed.proj.weight"] = (D, 3, P, P)
    s["backbone.patch_embed.proj.bias"] = (D,)
    for i in range(vit.depth):
        p = f"backbone.blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Hd, D)
        s[p + "mlp.fc1.bias"] = (Hd,)
        s[p + "mlp.fc2.weight"] = (D, Hd)
        s[p + "mlp.fc2.bias"] = (D,)
    s["backbone.norm.weight"] = (D,)
    s["backbone.norm.bias"] = (D,)
    dims = [head.in_dim] + [head.hidden_dim] * (head.nlayers - 1) + [head.bottleneck_dim]
    if head.nlayers == 1:
        s["head.mlp.weight"] = (head.bottleneck_dim, head.in_dim)
        s["head.mlp.bias"] = (head.bottleneck_dim,)
    else:
        for j in range(head.nlayers):  # Sequential indices 0,2,4 (GELU in between)
            s[f"head.mlp.{2 * j}.weight"] = (dims[j + 1], dims[j])
            s[f"head.mlp.{2 * j}.bias"] = (dims[j + 1],)
    s["head.last_layer.weight_g"] = (head.out_dim, 1)
    s["head.last_layer.weight_v"] = (head.out_dim, head.bottleneck_dim)
    return s


def new_state_dict(vit: VitCfg, head: HeadCfg) -> "Dict[str, torch.Tensor]":
    return {k: torch.zeros(v) for k, v in param_shapes(vit, head).items()}


# ----------------------------------------------------------------------------
# backbone
# ----------------------------------------------------------------------------
def patch_embed(x, weight, bias, patch):
    """PatchEmbed.forward, vision_transformer.py:242-245: fp32 cast, strided conv,
    flatten spatial dims, tokens-major."""
    y = F.conv2d(x.to(torch.float32), weight, bias, stride=patch)
    return y.flatten(2).transpose(1, 2)


def interpolate_pos_encoding(pos_embed, npatch, w, h, patch):
    """vision_transformer.py:331-359.  pos_embed: [1, 1+N, D]."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    D = pos_embed.shape[-1]
    side = int(math.sqrt(N))
    w0 = w // patch + 0.1
    h0 = h // patch + 0.1
    grid = pos_embed[:, 1:].reshape(1, side, side, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)),
                         mode="bicubic", align_corners=False, recompute_scale_factor=True)
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, D)
    return torch.cat((pos_embed[:, :1], grid), dim=1)


def prepare_tokens(x, sd, pre, vit: VitCfg):
    """vision_transformer.py:361-375."""
    B, _, w, h = x.shape
    tok = patch_embed(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"],
                      vit.patch_size)
    cls = sd[pre + "cls_token"].expand(B, -1, -1)
    tok = torch.cat((cls, tok), dim=1)
    return tok + interpolate_pos_encoding(sd[pre + "pos_embed"], tok.shape[1] - 1, w, h,
                                          vit.patch_size)


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(x, sd, pre, heads):
    """Attention.forward, vision_transformer.py:122-138.  Returns (out, probs)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    probs = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    out = (probs @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(out, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), probs


def mlp(x, sd, pre):
    """Mlp.forward, vision_transformer.py:93-99 (exact-erf GELU, dropout p=0)."""
    h = F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]))
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def _apply_drop(y, keep_scale):
    """drop_path, vision_transformer.py:52-62, with the Bernoulli draw made by the
    caller: keep_scale[b] = floor(keep + u_b) / keep (or None when disabled)."""
    if keep_scale is None:
        return y
    return y * keep_scale.view(-1, *([1] * (y.ndim - 1))).to(y.dtype)


def block(x, sd, pre, vit: VitCfg, drop=None):
    """Block.forward, vision_transformer.py:220-229.  drop = (scale_attn, scale_mlp)."""
    y, _ = attention(layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], vit.ln_eps),
                     sd, pre + "attn.", vit.num_heads)
    x = x + _apply_drop(y, None if drop is None else drop[0])
    y = mlp(layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], vit.ln_eps),
            sd, pre + "mlp.")
    return x + _apply_drop(y, None if drop is None else drop[1])


def vit_forward(x, sd, vit: VitCfg, pre="backbone.", drops=None, trace=None):
    """VisionTransformer.forward, vision_transformer.py:377-383 -> CLS [B, D].

    drops: optional list (per block) of (scale_attn, scale_mlp) keep-scale vectors.
    trace: optional dict that receives per-layer activations ("tokens", "block{i}").
    """
    t = prepare_tokens(x, sd, pre, vit)
    if trace is not None:
        trace["tokens"] = t
    for i in range(vit.depth):
        t = block(t, sd, f"{pre}blocks.{i}.", vit, None if drops is None else drops[i])
        if trace is not None:
            trace[f"block{i}"] = t
    t = layer_norm(t, sd[pre + "norm.weight"], sd[pre + "norm.bias"], vit.ln_eps)
    return t[:, 0]


# ----------------------------------------------------------------------------
# head + wrapper
# ----------------------------------------------------------------------------
def weight_norm_weight(g, v):
    """Old-style nn.utils.weight_norm(dim=0): W[k] = g[k] * v[k] / ||v[k]||_2
    (vision_transformer.py:703-706)."""
    return v * (g / v.norm(2, dim=1, keepdim=True))


def head_forward(x, sd, head: HeadCfg, pre="head.", trace=None):
    """DINOHead.forward, vision_transformer.py:716-720."""
    if head.nlayers == 1:
        x = F.linear(x, sd[pre + "mlp.weight"], sd[pre + "mlp.bias"])
    else:
        for j in range(head.nlayers):
            x = F.linear(x, sd[f"{pre}mlp.{2 * j}.weight"], sd[f"{pre}mlp.{2 * j}.bias"])
            if j < head.nlayers - 1:
                x = F.gelu(x)
    if trace is not None:
        trace["bottleneck"] = x
    x = F.normalize(x, dim=-1, p=2)
    W = weight_norm_weight(sd[pre + "last_layer.weight_g"], sd[pre + "last_layer.weight_v"])
    return F.linear(x, W)


def group_by_resolution(sizes: Sequence[int]) -> "List[tuple]":
    """Consecutive same-resolution runs as (start, end) pairs --
    the unique_consecutive/cumsum bookkeeping of dino/models/__init__.py:28-36."""
    runs, start = [], 0
    for i in range(1, len(sizes) + 1):
        if i == len(sizes) or sizes[i] != sizes[start]:
            runs.append((start, i))
            start = i
    return runs


def multicrop_forward(crops, sd, vit: VitCfg, head: HeadCfg, drops=None, trace=None):
    """MultiCropWrapper.forward, dino/models/__init__.py:24-46: one backbone pass per
    resolution run, features concatenated crop-major, one head pass."""
    if not isinstance(crops, (list, tuple)):
        crops = [crops]
    feats = []
    for gi, (s, e) in enumerate(group_by_resolution([c.shape[-1] for c in crops])):
        tr = None if trace is None else trace.setdefault(f"pass{gi}", {})
        feats.append(vit_forward(torch.cat(list(crops[s:e])), sd, vit,
                                 drops=None if drops is None else drops[gi], trace=tr))
    feats = torch.cat(feats)
    if trace is not None:
        trace["features"] = feats
    return head_forward(feats, sd, head, trace=trace)


# ----------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------
def teacher_temp_schedule(warmup_teacher_temp, teacher_temp, warmup_pct, max_iter):
    """dino/components/loss.py:28-36."""
    n = int(round(warmup_pct * max_iter, 0))
    return np.concatenate((np.linspace(warmup_teacher_temp, teacher_temp, n),
                           np.ones(max_iter - n) * teacher_temp))


def dino_loss(student_out, teacher_out, center, ncrops, teacher_temp, student_temp=0.1):
    """DINOLoss.forward without the center side effect, dino/components/loss.py:42-60."""
    s_chunks = (student_out / student_temp).chunk(ncrops)
    q_chunks = F.softmax((teacher_out - center) / teacher_temp, dim=-1).detach().chunk(2)
    total, terms = 0, 0
    for iq, q in enumerate(q_chunks):
        for v, s in enumerate(s_chunks):
            if v == iq:
                continue
            total = total + torch.sum(-q * F.log_softmax(s, dim=-1), dim=-1).mean()
            terms += 1
    return total / terms


@torch.no_grad()
def update_center(center, teacher_out, momentum=0.9, world_size=1, all_reduce=None):
    """DINOLoss.update_center, dino/components/loss.py:64-79.  all_reduce: optional
    in-place SUM callback standing in for dist.all_reduce."""
    batch = torch.sum(teacher_out, dim=0, keepdim=True)
    if all_reduce is not None:
        all_reduce(batch)
    batch = batch / (len(teacher_out) * world_size)
    return center * momentum + batch * (1 - momentum)


# ----------------------------------------------------------------------------
# schedules / optimiser-side helpers
# ----------------------------------------------------------------------------
def cosine_scheduler(base_value, final_value, nepochs, niter_per_ep, warmup_epochs=0,
                     start_warmup_value=0):
    """dino/utils/utils.py:184-204 (float64 numpy)."""
    warm_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warm_iters) if warmup_epochs > 0 \
        else np.array([])
    it = np.arange(nepochs * niter_per_ep - warm_iters)
    rest = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
    out = np.concatenate((warm, rest))
    assert len(out) == nepochs * niter_per_ep
    return out


def param_groups(named_params):
    """get_params_groups, dino/utils/utils.py:25-36: returns (regularized names,
    not-regularized names) preserving order; frozen params skipped."""
    reg, noreg = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        (noreg if (name.endswith(".bias") or p.ndim == 1) else reg).append(name)
    return reg, noreg


def grad_stats_and_clip_(named_grads, clip=None):
    """compute_gradient_stats, dino/utils/utils.py:59-98: per-tensor L2 norm, per-tensor
    clip in place (coef = clip / (norm + 1e-6), applied iff < 1).  named_grads:
    iterable of (name, grad or None).  Returns (stats dict, per-tensor norms)."""
    tot, mx, n, nclip, norms = 0.0, 0.0, 0, 0, {}
    for name, g in named_grads:
        if g is None:
            continue
        nrm = g.norm(2).item()
        norms[name] = nrm
        tot += nrm ** 2
        mx = max(mx, nrm)
        n += 1
        if clip is not None:
            coef = clip / (nrm + 1e-6)
            if coef < 1:
                g.mul_(coef)
                nclip += 1
    return {"grad_norm": tot ** 0.5, "grad_max": mx,
            "clip_fraction": nclip / n if n else 0.0}, norms


def adamw_step_(p, g, state, lr, wd, betas=(0.9, 0.999), eps=1e-8):
    """One torch.optim.AdamW update of one tensor (dino.py:251; decoupled decay,
    bias-corrected, denom = sqrt(v)/sqrt(bc2) + eps).  state: dict step/m/v."""
    state["step"] = state.get("step", 0) + 1
    if "m" not in state:
        state["m"] = torch.zeros_like(p)
        state["v"] = torch.zeros_like(p)
    b1, b2 = betas
    t = state["step"]
    p.mul_(1 - lr * wd)
    state["m"].mul_(b1).add_(g, alpha=1 - b1)
    state["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    denom = (state["v"].sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(state["m"], denom, value=-lr / bc1)


@torch.no_grad()
def ema_update_(teacher_sd, student_sd, m):
    """Momentum teacher, dino/utils/train_utils.py:152-162: k <- m k + (1-m) q."""
    for k in teacher_sd:
        teacher_sd[k].mul_(m).add_((1 - m) * student_sd[k].detach())


# ----------------------------------------------------------------------------
# the step
# ----------------------------------------------------------------------------
@dataclass
class StepHP:
    """Per-run constants of the step (dino/configs/default.yaml + dino.py:205-281,350)."""
    ncrops: int = 12
    student_temp: float = 0.1
    center_momentum: float = 0.9
    clip_grad: float = 3.0
    freeze_last_layer_iter: int = 0
    norm_last_layer: bool = False


class OracleTrainer:
    """Holds student/teacher/optimizer/center state and runs the reference step order
    of dino/utils/train_utils.py:52-162 on CPU (fp32, or under CPU autocast)."""

    def __init__(self, vit: VitCfg, head: HeadCfg, hp: StepHP, student_sd, lr_sched, wd_sched,
                 mom_sched, ttemp_sched):
        self.vit, self.head, self.hp = vit, head, hp
        self.student = {k: v.detach().clone().requires_grad_(True) for k, v in student_sd.items()}
        if hp.norm_last_layer:  # vision_transformer.py:707-708
            self.student["head.last_layer.weight_g"].requires_grad_(False)
        # teacher <- student weights, frozen (dino.py:196-202)
        self.teacher = {k: v.detach().clone() for k, v in student_sd.items()}
        self.center = torch.zeros(1, head.out_dim)  # loss.py:25
        self.opt_state = {k: {} for k in self.student}
        self.reg, self.noreg = param_groups(self.student.items())
        self.lr, self.wd, self.mom, self.ttemp = lr_sched, wd_sched, mom_sched, ttemp_sched

    def step(self, crops, it, drops=None, trace=None, autocast_dtype=None):
        hp = self.hp
        ctx = torch.autocast("cpu", dtype=autocast_dtype) if autocast_dtype is not None \
            else torch.autocast("cpu", enabled=False)
        with ctx:
            with torch.no_grad():  # teacher sees the 2 global views, train_utils.py:75-77
                t_out = multicrop_forward(crops[:2], self.teacher, self.vit, self.head)
            s_out = multicrop_forward(crops, self.student, self.vit, self.head, drops=drops,
                                      trace=trace)
            loss = dino_loss(s_out, t_out, self.center, hp.ncrops, float(self.ttemp[it]),
                             hp.student_temp)
            self.center = update_center(self.center, t_out, hp.center_momentum)  # loss.py:61
        if trace is not None:
            s_out.retain_grad()
            trace["student_out"], trace["teacher_out"] = s_out, t_out
        for p in self.student.values():
            p.grad = None  # optimizer.zero_grad(), train_utils.py:115
        loss.backward()
        grads = [(k, p.grad) for k, p in self.student.items()]
        if trace is not None:
            trace["grads_raw"] = {k: g.detach().clone() for k, g in grads if g is not None}
        stats, norms = grad_stats_and_clip_(grads, hp.clip_grad if hp.clip_grad else None)
        if it < hp.freeze_last_layer_iter:  # cancel_gradients_last_layer, utils.py:101-106
            for k, p in self.student.items():
                if "last_layer" in k:
                    p.grad = None
        with torch.no_grad():  # per-iteration lr/wd, train_utils.py:56-59; AdamW dino.py:251
            for names, wd in ((self.reg, float(self.wd[it])), (self.noreg, 0.0)):
                for k in names:
                    p = self.student[k]
                    if p.grad is None:
                        continue
                    adamw_step_(p, p.grad, self.opt_state[k], float(self.lr[it]), wd)
        ema_update_(self.teacher, self.student, float(self.mom[it]))
        return {"loss": loss.detach(), "stats": stats, "norms": norms}


def make_schedules(steps, batch_global, lr=5e-4, min_lr=1e-6, wd=0.04, wd_end=0.4,
                   momentum_teacher=0.996, warmup_pct=0.1, nepochs=1,
                   warmup_teacher_temp=0.04, teacher_temp=0.04, warmup_teacher_temp_pct=0.0):
    """The three cosine schedules + teacher temperature exactly as dino.py:258-281 and
    loss.py:28-36 build them, for nepochs x steps iterations."""
    warmup_epochs = int(warmup_pct * nepochs)
    base_lr = lr * batch_global / 256.0
    return (cosine_scheduler(base_lr, min_lr, nepochs, steps, warmup_epochs=warmup_epochs),
            cosine_scheduler(wd, wd_end, nepochs, steps),
            cosine_scheduler(momentum_teacher, 1, nepochs, steps),
            teacher_temp_schedule(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_pct,
                                  nepochs * steps))
