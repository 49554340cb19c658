"""Feeding the step: host crops -> HBM (the reference's DataLoader(pin_memory=True) +
`.cuda(non_blocking=True)`, dino.py:127-138 and dino/utils/train_utils.py:61-67) and, one step further up, the
multi-crop augmentation itself on the GPU (dino/data/augmentations.py:75-157)."""
from .prefetch import CropPrefetcher, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
from .augment import DeviceAugmentationDINO, DeviceAugmentingLoader

__all__ = ["CropPrefetcher", "DeviceAugmentationDINO", "DeviceAugmentingLoader", "IMAGENET_DEFAULT_MEAN",
           "IMAGENET_DEFAULT_STD"]
