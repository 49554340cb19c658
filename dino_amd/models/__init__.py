"""MultiCropWrapper (reference dino/models/__init__.py:7-46) on the HIP engine."""
import os
from typing import List

import torch
import torch.nn as nn

from .. import config as _cfg
from ..engine import autograd as _ag
from ..engine.store import store_for
from . import vision_transformer
from .vision_transformer import DINOHead, VisionTransformer


def group_by_resolution(sizes) -> "List[tuple]":
    """(start, end) runs of consecutive equal crop sizes -- what the reference derives with
    unique_consecutive + cumsum (models/__init__.py:28-36)."""
    runs, start = [], 0
    for i in range(1, len(sizes) + 1):
        if i == len(sizes) or sizes[i] != sizes[start]:
            runs.append((start, i))
            start = i
    return runs


_FUSED_DEFAULT = os.environ.get("DINO_PROTO_FUSED", "0") == "1"


class MultiCropWrapper(nn.Module):
    """One backbone pass per resolution run, features concatenated crop-major, one head pass.

    Owns ONE flat ParamStore covering backbone and head ("backbone.*", "head.*" keys), so the
    optimiser / EMA / gradient all-reduce see a single contiguous buffer.
    """

    def __init__(self, backbone, head, compute_dtype=None):
        super().__init__()
        if not isinstance(backbone, VisionTransformer) or not isinstance(head, DINOHead):
            raise TypeError("dino_amd.MultiCropWrapper drives dino_amd VisionTransformer + DINOHead "
                            "(no generic nn.Module fallback)")
        backbone.fc, backbone.head = nn.Identity(), nn.Identity()
        self.backbone = backbone
        self.head = head
        self.compute_dtype = compute_dtype or _cfg.get_default_dtype()

    def store(self):
        st = store_for(self, self.compute_dtype)
        st.ensure_fresh()
        return st

    def forward(self, x, return_features: bool = False, trace=None):
        if not isinstance(x, (list, tuple)):
            x = [x]
        st = self.store()
        # consecutive crops of one resolution form a group (models/__init__.py:28-36); all groups go
        # through the backbone as ONE token matrix (engine.vit.vit_forward), no torch.cat of images
        runs = group_by_resolution([inp.shape[-1] for inp in x])
        groups = [list(x[s:e]) for s, e in runs]
        tr = None if trace is None else [trace.setdefault(f"pass{gi}", {}) for gi in range(len(groups))]
        output = _ag.run_backbone(st, "backbone.", self.backbone.spec, groups, self.training, trace=tr)
        if trace is not None:
            trace["features"] = output
        # fused_head_loss (or DINO_PROTO_FUSED=1): the head stops at the normalised bottleneck and DINOLoss forms
        # the prototype logits chunk by chunk (no [crops*B, out_dim] tensor); only DINOLoss understands that
        # output, so it is opt-in
        lazy = getattr(self, "fused_head_loss", _FUSED_DEFAULT) and self.training
        logits = _ag.run_head(st, "head.", self.head.spec, output, trace=trace, lazy=lazy, teacher_sums=self.training)
        if return_features:
            return logits, output
        return logits


class MultiCropWrapperWithFeatures(MultiCropWrapper):
    """The reference's feature-returning wrapper (dino/models/__init__.py:49-103): same constructor,
    `forward(x, return_features=False)` -> logits, or (logits [crops*B, out_dim], CLS features
    [crops*B, embed_dim]) with return_features=True.  The reference runs the head once per resolution
    group and concatenates; the head is row-wise, so the single head pass of MultiCropWrapper gives the
    same rows in the same (crop-major) order.  The domain-adversarial branch that consumes the features
    (adversarial.enable) is out of scope; the features carry gradients like any other output."""

    def forward(self, x, return_features: bool = False, trace=None):
        return super().forward(x, return_features=return_features, trace=trace)
