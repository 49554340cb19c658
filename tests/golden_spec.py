"""Shared description of the golden-fixture cases (inputs are regenerated from
oracle/detfill.py; tools/make_golden.py ran the reference on them)."""
import numpy as np
import torch

from oracle import detfill

# Reduced model built from the reference classes: 2 heads of 64 so
# the head-major QKV split is exercised; 7x7 pos grid; local crops hit the 7x7->3x3
# bicubic path.  freeze_iter=1: step 0 runs with the last layer's grads cancelled.
TINY = dict(tag="tiny",
            vit=dict(img_size=64, patch_size=8, embed_dim=128, depth=2, num_heads=2),
            head=dict(out_dim=512, hidden_dim=256, bottleneck_dim=64),
            batch=2, sizes=[56, 56, 24, 24, 24], freeze_iter=1)

# A reference loss CURVE (not just 3 steps): the tiny model for 24 steps at a learning rate large enough that the
# weights, the centre and the loss visibly move (lr 2e-3 cosine-decayed, teacher temperature warmed up from 0.04 to
# 0.07 over the first quarter, last layer frozen for 2 steps) -- scalars per step only.
CURVE = dict(TINY, nsteps=24, lr=2e-3, freeze_iter=2, ttemp=(0.04, 0.07, 0.25))

# BASELINE.json configs[0]: ViT-S/16, 2x224 global crops only, batch 8, K=65536.
FULL = dict(tag="vits16",
            vit=dict(img_size=256, patch_size=16, embed_dim=384, depth=12, num_heads=6),
            head=dict(out_dim=65536, hidden_dim=2048, bottleneck_dim=256),
            batch=8, sizes=[224, 224], freeze_iter=0)

# BASELINE.json configs[1] at a batch the CPU reference finishes in seconds: ViT-S/16 full depth,
# 2x224 + 10x96 multi-crop, K=65536 (the headline bench config is this at B=256, bf16).
# freeze_iter=1: step 0 runs with the prototype layer's grads cancelled.
MULTI = dict(tag="vits16mc",
             vit=dict(img_size=256, patch_size=16, embed_dim=384, depth=12, num_heads=6),
             head=dict(out_dim=65536, hidden_dim=2048, bottleneck_dim=256),
             batch=2, sizes=[224, 224] + [96] * 10, freeze_iter=1)

# BASELINE.json configs[2..4] at FULL depth and K = 65536, multi-crop, at a batch the CPU reference finishes in
# a minute or two (digests only): ViT-B/16, ViT-S/8 (785-token globals, 145-token locals), ViT-L/14 (257 / 37).
_HEAD = dict(out_dim=65536, hidden_dim=2048, bottleneck_dim=256)
OTHER_FULL = {
    "vitb16": dict(tag="vitb16mc", vit=dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12),
                   head=_HEAD, batch=2, sizes=[224, 224] + [96] * 10, freeze_iter=1),
    "vits8": dict(tag="vits8mc", vit=dict(img_size=224, patch_size=8, embed_dim=384, depth=12, num_heads=6),
                  head=_HEAD, batch=1, sizes=[224, 224] + [96] * 10, freeze_iter=1),
    "vitl14": dict(tag="vitl14mc", vit=dict(img_size=224, patch_size=14, embed_dim=1024, depth=24, num_heads=16),
                   head=_HEAD, batch=1, sizes=[224, 224] + [96] * 10, freeze_iter=1),
}


def other_crops(name):
    sp = OTHER_FULL[name]
    return detfill.det_crops(sp["batch"], sp["sizes"], tag=sp["tag"] + ":img")


# name -> (base, final, nepochs, niter_per_ep, warmup_epochs, start_warmup_value)
SCHED_CASES = {
    "momentum": (0.996, 1.0, 2, 5, 0, 0),
    "lr_warm": (5e-4, 1e-6, 10, 7, 1, 0),
    "wd": (0.04, 0.4, 3, 11, 0, 0),
}
# name -> (img_size, patch, embed_dim, crop)
POS_CASES = {
    "p16_96": (256, 16, 8, 96),     # 196 -> 36
    "p8_96": (256, 8, 8, 96),       # 784 -> 144
    "p14_96": (256, 14, 8, 96),     # 256 -> 36 (96//14 = 6)
    "p8_24_tiny": (64, 8, 8, 24),   # 49 -> 9
    "p16_224_id": (256, 16, 8, 224),  # identity
}
LOSS_CASE = (65536, 2, 12)  # K, B, ncrops

_NSAMP = 32


def digest(t) -> np.ndarray:
    """[sum, l2, 32 evenly spaced samples] in float64 -- a compact fingerprint."""
    a = t.detach().to(torch.float64).reshape(-1).cpu().numpy() if isinstance(t, torch.Tensor) \
        else np.asarray(t, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, a.size - 1, _NSAMP).astype(np.int64)
    return np.concatenate(([a.sum(), np.sqrt((a * a).sum())], a[idx]))


def tiny_crops():
    return detfill.det_crops(TINY["batch"], TINY["sizes"], tag="tiny:img")


def multi_crops():
    return detfill.det_crops(MULTI["batch"], MULTI["sizes"], tag="vits16mc:img")


def full_crops():
    return detfill.det_crops(FULL["batch"], FULL["sizes"], tag="vits16:img")
