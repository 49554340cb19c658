"""Bridges between the explicit HIP schedules and torch.autograd so that the reference's
call pattern -- student(images); loss = dino_loss(...); loss.backward() -- works unchanged.

Each bridge is ONE autograd node per module call (backbone pass, head, loss).  Parameter
gradients are not returned through autograd: the schedules write them into the store's flat
grad buffer, and p.grad views are attached when the last node has run.
"""
import torch

from .. import ops
from . import vit as E


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, images, st, pre, spec, training, trace):
        cls, tape = E.vit_forward(st, pre, spec, images, save=True, training=training, trace=trace)
        ctx.st, ctx.tape = st, tape
        return cls

    @staticmethod
    def backward(ctx, dcls):
        st, tape = ctx.st, ctx.tape
        ctx.tape = None
        E.vit_backward(st, tape, dcls)
        if st.pending_bwd == 0:
            st.backward_done()
        return (None,) * 7


class LazyPrototypes:
    """Rides on the tensor a lazy head returns (the normalised bottleneck [R, Db] instead of the logits):
    DINOLoss recognises it and forms the logits chunk by chunk itself."""

    def __init__(self, st, pre):
        self.st, self.pre = st, pre
        self.dyn32 = None      # d loss / d yn in fp32, left by the loss's backward for the head's


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, st, pre, spec, trace, lazy):
        logits, tape = E.head_forward(st, pre, spec, x, save=True, trace=trace, lazy=lazy is not None)
        ctx.st, ctx.tape, ctx.lazy = st, tape, lazy
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        st, tape, lazy = ctx.st, ctx.tape, ctx.lazy
        ctx.tape = None
        if lazy is not None and lazy.dyn32 is not None:
            dlogits, lazy.dyn32 = lazy.dyn32, None     # (autograd hands the gradient over in yn's dtype)
        dx = E.head_backward(st, tape, dlogits)
        if st.pending_bwd == 0:
            st.backward_done()
        return None, dx, None, None, None, None, None


def _recording(st) -> bool:
    return torch.is_grad_enabled() and st.trainable


def run_backbone(st, pre, spec, images, training, trace=None):
    if _recording(st):
        st.begin_recording()
        return _BackboneFn.apply(st.anchor, images, st, pre, spec, training, trace)
    cls, _ = E.vit_forward(st, pre, spec, images, save=False, training=training, trace=trace)
    return cls


def run_head(st, pre, spec, x, trace=None, lazy=False, teacher_sums=True):
    if _recording(st):
        st.begin_recording()
        tag = LazyPrototypes(st, pre) if lazy else None
        out = _HeadFn.apply(st.anchor, x, st, pre, spec, trace, tag)
        if tag is not None:
            out._dino_lazy = tag
        return out
    if not teacher_sums:    # an eval-mode pass (feature extraction, inspection): plain logits
        logits, _ = E.head_forward(st, pre, spec, x, save=False, trace=trace)
        return logits
    # no-grad pass of a module in train mode (the teacher of a training step): the logits' column sums ride along
    # for DINOLoss.update_center
    colsum = torch.zeros(st.params[pre + "last_layer.weight_v"].shape[0], dtype=torch.float32, device=st.device)
    logits, _ = E.head_forward(st, pre, spec, x, save=False, trace=trace, colsum_out=colsum)
    logits._dino_colsum = (colsum, logits._version)     # stale once the caller writes into the logits
    return logits


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, mod, teacher_temp):
        B = teacher.shape[0] // 2
        dS, loss = mod._fwd_bwd(student, teacher, B, teacher_temp)
        ctx.dS = dS
        return loss

    @staticmethod
    def backward(ctx, gout):
        dS = ctx.dS
        ctx.dS = None
        ops.scale_by_scalar(dS, gout.detach().to(torch.float32).reshape(1).contiguous())
        return dS, None, None, None


class _LossLazyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, yn, teacher, mod, teacher_temp, lazy):
        B = teacher.shape[0] // 2
        pctx, loss = mod._fwd_lazy(yn, teacher, B, teacher_temp, lazy)
        ctx.pctx, ctx.lazy = pctx, lazy
        return loss

    @staticmethod
    def backward(ctx, gout):
        pctx, lazy = ctx.pctx, ctx.lazy
        ctx.pctx = None
        dyn = E.proto_loss_backward(lazy.st, pctx, gout.detach().to(torch.float32).reshape(1).contiguous())
        lazy.dyn32 = dyn
        return dyn.to(pctx["yn"].dtype), None, None, None, None


def run_loss(student, teacher, mod, teacher_temp):
    lazy = getattr(student, "_dino_lazy", None)
    if lazy is not None:
        return _LossLazyFn.apply(student, teacher, mod, teacher_temp, lazy)
    return _LossFn.apply(student, teacher, mod, teacher_temp)
