"""ViT backbone family + DINO projection head with the reference's constructor signatures,
parameter names/order and state_dict keys (dino/models/vision_transformer.py:248-482,
675-720), computed by the HIP kernels of libdino_hip.so.

The modules only OWN parameters; forward/backward run through dino_amd.engine (explicit
schedules over flat parameter storage).  There is no PyTorch compute fallback.
"""
import math
import warnings
from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import config as _cfg
from ..engine import autograd as _ag
from ..engine.store import store_for
from ..engine.vit import HeadSpec, VitSpec


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """Truncated normal init by inverse-CDF sampling (same distribution as the reference's
    helper, vision_transformer.py:9-49)."""
    def cdf(x):
        return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))
    if mean < a - 2 * std or mean > b + 2 * std:
        warnings.warn("mean is more than 2 std from [a, b] in trunc_normal_", stacklevel=2)
    lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
    with torch.no_grad():
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(mean)
        tensor.clamp_(min=a, max=b)
    return tensor


class _Affine(nn.Module):
    """weight [out, in] + bias [out]: the parameters of an nn.Linear (trunc-normal / zero)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(trunc_normal_(torch.empty(out_features, in_features), std=0.02))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None


class _Norm(nn.Module):
    """LayerNorm parameters (weight 1, bias 0)."""

    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Marker(nn.Module):
    """Parameter-free placeholder keeping nn.Sequential indices (GELU slots of DINOHead.mlp)."""


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = _Affine(dim, dim * 3, bias=qkv_bias)
        self.proj = _Affine(dim, dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = _Affine(in_features, hidden_features)
        self.fc2 = _Affine(hidden_features, in_features)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop_path=0.0, eps=1e-5):
        super().__init__()
        self.norm1 = _Norm(dim, eps)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path_rate = float(drop_path)
        self.norm2 = _Norm(dim, eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class _Conv(nn.Module):
    """Conv2d(k=P, s=P) parameters with nn.Conv2d's default init (the reference never
    re-initialises the patch projection: _init_weights only matches Linear/LayerNorm)."""

    def __init__(self, in_chans, embed_dim, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(embed_dim, in_chans, patch, patch))
        self.bias = nn.Parameter(torch.empty(embed_dim))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_chans * patch * patch)
        nn.init.uniform_(self.bias, -bound, bound)


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size = patch_size
        self.proj = _Conv(in_chans, embed_dim, patch_size)


def _ln_eps(norm_layer) -> float:
    """Recover eps from the reference-style `norm_layer` callable (partial(nn.LayerNorm, eps=..))."""
    kw = getattr(norm_layer, "keywords", None) or {}
    return float(kw.get("eps", 1e-5))


class VisionTransformer(nn.Module):
    """Vision Transformer (reference vision_transformer.py:248-402)."""

    def __init__(self, img_size: int = 256, patch_size: int = 16, in_chans: int = 3,
                 dino_max_crop_scale: float = 0.875, num_classes: int = 0, embed_dim: int = 768,
                 depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = False,
                 qk_scale: Optional[float] = None, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.0, norm_layer: Callable = nn.LayerNorm,
                 mask_attn: bool = False, img_size_pretrained: Optional[int] = None):
        super().__init__()
        if in_chans != 3 or num_classes != 0 or qk_scale is not None or mask_attn:
            raise NotImplementedError("dino_amd ViT: in_chans=3, num_classes=0, default qk scale, "
                                      "no masked attention (HIPT region model is out of scope)")
        if drop_rate != 0.0 or attn_drop_rate != 0.0:
            raise NotImplementedError("dropout / attention dropout are 0 in every DINO config")
        if embed_dim // num_heads != 64 or embed_dim % num_heads:
            raise NotImplementedError("the attention kernels are built for head_dim = 64")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        num_patches = int((img_size_pretrained or img_size) * dino_max_crop_scale // patch_size) ** 2
        self.patch_embed = PatchEmbed(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(trunc_normal_(torch.zeros(1, 1, embed_dim), std=0.02))
        self.pos_embed = nn.Parameter(trunc_normal_(torch.zeros(1, num_patches + 1, embed_dim), std=0.02))
        eps = _ln_eps(norm_layer)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, dpr[i], eps)
                                     for i in range(depth)])
        self.norm = _Norm(embed_dim, eps)
        self.head = nn.Identity()
        self.compute_dtype = _cfg.get_default_dtype()
        self._spec = VitSpec(patch_size, embed_dim, depth, num_heads, int(embed_dim * mlp_ratio), eps, dpr)

    @property
    def spec(self) -> VitSpec:
        return self._spec

    def forward(self, x, mask: Optional[torch.Tensor] = None):
        """x: [B,3,S,S] -> CLS features [B, embed_dim]."""
        if mask is not None:
            raise NotImplementedError("masked attention is out of scope")
        st = store_for(self, self.compute_dtype)
        st.ensure_fresh()
        return _ag.run_backbone(st, "", self._spec, x, self.training)

    def get_last_selfattention(self, x, mask: Optional[torch.Tensor] = None):
        """Attention probabilities [B, heads, N, N] of the last block (:385-392; the
        return_attention=True path of Block.forward, :220-226).  Inspection helper: computed without
        recording a backward pass."""
        if mask is not None:
            raise NotImplementedError("masked attention is out of scope")
        from ..engine import vit as _E
        st = store_for(self, self.compute_dtype)
        st.ensure_fresh()
        with torch.no_grad():
            probs, _ = _E.vit_forward(st, "", self._spec, x, save=False, training=self.training,
                                      last_attention=True)
        return probs

    def get_intermediate_layers(self, x, n=1, mask=None):
        """Normalised token outputs of the n last blocks (inference helper, :394-402)."""
        from .. import ops
        st = store_for(self, self.compute_dtype)
        st.ensure_fresh()
        trace = {}
        with torch.no_grad():
            _ag.run_backbone(st, "", self._spec, x, False, trace=trace)
        outs = []
        for i in range(self._spec.depth - n, self._spec.depth):
            t = trace[f"block{i}"]
            y, _, _ = ops.layernorm_fwd(t.reshape(-1, t.shape[-1]), st.p32("norm.weight"),
                                        st.p32("norm.bias"), self._spec.eps)
            outs.append(y.reshape(t.shape))
        return outs


def _vit(embed_dim, depth, num_heads, img_size, patch_size, kwargs):
    from functools import partial
    return VisionTransformer(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                             num_heads=num_heads, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_tiny(img_size: int = 256, patch_size: int = 16, embed_dim: int = 192, **kwargs):
    return _vit(embed_dim, 12, 3, img_size, patch_size, kwargs)


def vit_small(img_size: int = 256, patch_size: int = 16, embed_dim: int = 384, **kwargs):
    return _vit(embed_dim, 12, 6, img_size, patch_size, kwargs)


def vit_base(img_size: int = 256, patch_size: int = 16, embed_dim: int = 768, **kwargs):
    return _vit(embed_dim, 12, 12, img_size, patch_size, kwargs)


def vit_large(img_size: int = 256, patch_size: int = 14, embed_dim: int = 1024, **kwargs):
    return _vit(embed_dim, 24, 16, img_size, patch_size, kwargs)


class _WeightNormProto(nn.Module):
    """Parameters of nn.utils.weight_norm(nn.Linear(in, out, bias=False)): weight_g then weight_v."""

    def __init__(self, in_features, out_features):
        super().__init__()
        # created after DINOHead's trunc-normal pass in the reference (:702-705): nn.Linear default init
        v = nn.init.kaiming_uniform_(torch.empty(out_features, in_features), a=math.sqrt(5))
        self.weight_g = nn.Parameter(torch.ones(out_features, 1))
        self.weight_v = nn.Parameter(v)


class DINOHead(nn.Module):
    """3-layer GELU MLP -> L2 normalise -> weight-normed prototypes (:675-720)."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3,
                 hidden_dim=2048, bottleneck_dim=256):
        super().__init__()
        if use_bn:
            raise NotImplementedError("use_bn_in_head is false in every shipped config")
        nlayers = max(nlayers, 1)
        if nlayers == 1:
            self.mlp = _Affine(in_dim, bottleneck_dim)
            names = ["mlp"]
        else:
            layers = [_Affine(in_dim, hidden_dim), _Marker()]
            for _ in range(nlayers - 2):
                layers += [_Affine(hidden_dim, hidden_dim), _Marker()]
            layers.append(_Affine(hidden_dim, bottleneck_dim))
            self.mlp = nn.Sequential(*layers)
            names = [f"mlp.{2 * j}" for j in range(nlayers)]
        self.last_layer = _WeightNormProto(bottleneck_dim, out_dim)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False
        self.compute_dtype = _cfg.get_default_dtype()
        self._spec = HeadSpec(nlayers, names)

    @property
    def spec(self) -> HeadSpec:
        return self._spec

    def forward(self, x):
        st = store_for(self, self.compute_dtype)
        st.ensure_fresh()
        return _ag.run_head(st, "", self._spec, x, teacher_sums=self.training)
