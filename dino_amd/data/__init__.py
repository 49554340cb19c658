"""Feeding the step: host crops -> HBM (the reference's DataLoader(pin_memory=True) +
`.cuda(non_blocking=True)`, dino.py:127-138 and dino/utils/train_utils.py:61-67)."""
from .prefetch import CropPrefetcher, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD

__all__ = ["CropPrefetcher", "IMAGENET_DEFAULT_MEAN", "IMAGENET_DEFAULT_STD"]
