"""Multi-crop augmentation on the GPU.

`DeviceAugmentationDINO` has `PatchDataAugmentationDINO`'s constructor (dino/data/augmentations.py:75-87) and
produces the same list of crops -- [global1, global2, local x n], each `[B, 3, S, S]` fp32, normalised -- but
takes a whole batch of uint8 source images `[B, H, W, 3]` that already sits in HBM and runs every image operation
as a HIP kernel (dino_amd/csrc/augment.hip), bit-identical to the Pillow routine the reference's torchvision
transform lands in.  What stays on the host is the random draw of each crop's parameters (`sample`), vectorised
over the batch with one numpy Generator: the distributions are torchvision's
(RandomResizedCrop.get_params with its ten attempts and central-crop fall-back, RandomHorizontalFlip(0.5),
RandomApply(ColorJitter(0.4, 0.4, 0.2, 0.1), p=0.8) with its random op order, RandomGrayscale(0.2),
GaussianBlur(p; radius U[0.1, 2]), Solarization(0.2)), the random stream is not (torchvision draws crop by
crop from torch's and Python's global generators inside DataLoader workers).

At 8 000 img/s x 12 crops the reference's CPU workers cannot feed one MI355X; here a batch of 256 images costs
the host one parameter table (a few hundred microseconds of numpy) and the device about a millisecond.
"""
import math
from collections import deque
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops
from .prefetch import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD

PW = ops.AUG_PARAM_WORDS
OP_NONE, OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE, OP_GRAY, OP_SOLARIZE = range(7)
SLOT_GRAY, SLOT_SOLARIZE = 4, 5


def box_blur_constants_array(radius: np.ndarray, passes: int = 3):
    """Pillow's BoxBlur.c in its own (single) precision, for an array of Gaussian radii: the box radius whose
    `passes`-fold blur approximates the Gaussian, then ImagingHorizontalBoxBlur's (radius, ww, fw)."""
    f = np.float32
    r = np.asarray(radius, dtype=f)
    sigma2 = (r * r / f(passes)).astype(f)
    big_l = np.sqrt(12.0 * sigma2.astype(np.float64) + 1.0).astype(f)
    l = np.floor((big_l.astype(np.float64) - 1.0) / 2.0).astype(f)
    a = ((f(2) * l + f(1)).astype(f) * ((l * (l + f(1)).astype(f)).astype(f) - (f(3) * sigma2).astype(f)).astype(f)).astype(f)
    lp = (l + f(1)).astype(f)
    a = (a / (f(6) * (sigma2 - (lp * lp).astype(f)).astype(f)).astype(f)).astype(f)
    fr = (l + a).astype(f)
    ri = fr.astype(np.int64)
    ww = (f(1 << 24) / (fr * f(2) + f(1)).astype(f)).astype(f).astype(np.int64)
    fw = ((1 << 24) - (ri * 2 + 1) * ww) // 2
    return ri, ww, fw


def box_blur_constants(radius: float, passes: int = 3) -> Tuple[int, int, int]:
    ri, ww, fw = box_blur_constants_array(np.array([radius]), passes)
    return int(ri[0]), int(ww[0]), int(fw[0])


def hue_shift(factor):
    """torchvision's `np.uint8(hue_factor * 255)`: truncation toward zero, then the 8-bit wrap."""
    if np.ndim(factor):
        return (np.asarray(factor, dtype=np.float64) * 255).astype(np.int64) & 255
    return int(factor * 255) & 255


def sample_boxes(rng: np.random.Generator, n: int, height: int, width: int, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """RandomResizedCrop.get_params for n crops at once: (top, left, h, w) int arrays.  Ten attempts of
    (area ~ U(scale) * H * W, log-ratio ~ U(log ratio)); the first that fits wins, else the central crop."""
    area = float(height * width)
    ta = area * rng.uniform(scale[0], scale[1], size=(n, 10))
    ar = np.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1]), size=(n, 10)))
    w = np.rint(np.sqrt(ta * ar)).astype(np.int64)
    h = np.rint(np.sqrt(ta / ar)).astype(np.int64)
    ok = (w > 0) & (w <= width) & (h > 0) & (h <= height)
    first = np.argmax(ok, axis=1)
    has = ok.any(axis=1)
    rows = np.arange(n)
    w, h = w[rows, first], h[rows, first]
    # fall-back: central crop at the nearest allowed ratio
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        fw_, fh_ = width, int(round(width / min(ratio)))
    elif in_ratio > max(ratio):
        fh_, fw_ = height, int(round(height * max(ratio)))
    else:
        fw_, fh_ = width, height
    w = np.where(has, w, fw_)
    h = np.where(has, h, fh_)
    top = np.where(has, np.floor(rng.random(n) * (height - h + 1)).astype(np.int64), (height - h) // 2)
    left = np.where(has, np.floor(rng.random(n) * (width - w + 1)).astype(np.int64), (width - w) // 2)
    return top, left, h, w


class DeviceAugmentationDINO:
    """See the module docstring.  `aug(images_u8)` -> list of 2 + local_crops_number fp32 crops."""

    def __init__(self, global_crop_size: int, local_crop_size: int, global_crops_scale: Sequence[float],
                 local_crops_scale: Sequence[float], local_crops_number: int,
                 mean: Sequence[float] = IMAGENET_DEFAULT_MEAN, std: Sequence[float] = IMAGENET_DEFAULT_STD,
                 solarization: bool = False, seed: Optional[int] = None):
        self.gsize, self.lsize = int(global_crop_size), int(local_crop_size)
        self.gscale, self.lscale = tuple(global_crops_scale), tuple(local_crops_scale)
        self.local_crops_number = int(local_crops_number)
        self.mean, self.std = tuple(mean), tuple(std)
        self.solarization = bool(solarization)
        self.rng = np.random.default_rng(seed)
        # ColorJitter(brightness=0.4, contrast=0.4, saturation=0.2, hue=0.1), augmentations.py:93-95
        self.jitter = dict(brightness=(0.6, 1.4), contrast=(0.6, 1.4), saturation=(0.8, 1.2), hue=(-0.1, 0.1))

    # ------------------------------------------------------------------------------- host: the draws
    def _sample_kind(self, n: int, H: int, W: int, scale, blur_p: float, solarize_p: float,
                     nsrc: Optional[int] = None) -> np.ndarray:
        """n crops of one kind (n = a multiple of the nsrc source images: crop k comes from image k % nsrc)."""
        rng = self.rng
        P = np.zeros((n, PW), dtype=np.int32)
        P[:, 0] = np.arange(n) % (nsrc or n)
        top, left, h, w = sample_boxes(rng, n, H, W, scale)
        P[:, 1], P[:, 2], P[:, 3], P[:, 4] = top, left, h, w
        P[:, 5] = rng.random(n) < 0.5                                   # RandomHorizontalFlip
        apply = ~(0.8 < rng.random(n))                                  # RandomApply: skipped when p < rand
        order = np.argsort(rng.random((n, 4)), axis=1)                  # torch.randperm(4)
        fac = np.stack([rng.uniform(*self.jitter[k], size=n) for k in
                        ("brightness", "contrast", "saturation", "hue")], axis=1)
        for slot in range(4):
            kind = order[:, slot]                                       # 0 b, 1 c, 2 s, 3 h
            P[:, 6 + slot] = np.where(apply, kind + 1, OP_NONE)
            f = fac[np.arange(n), kind]
            fbits = f.astype(np.float32).view(np.int32)
            hbits = hue_shift(f).astype(np.int32)
            P[:, 12 + slot] = np.where(kind == 3, hbits, fbits)
        P[:, 6 + SLOT_GRAY] = np.where(rng.random(n) < 0.2, OP_GRAY, OP_NONE)   # RandomGrayscale(0.2)
        blur = rng.random(n) <= blur_p                                  # GaussianBlur(p), :49
        radius = rng.uniform(0.1, 2.0, size=n)
        ri, ww, fw = box_blur_constants_array(radius)
        P[:, 18] = blur
        P[:, 19], P[:, 20], P[:, 21] = np.where(blur, ri, 0), np.where(blur, ww, 0), np.where(blur, fw, 0)
        if solarize_p > 0:
            P[:, 6 + SLOT_SOLARIZE] = np.where(rng.random(n) < solarize_p, OP_SOLARIZE, OP_NONE)
        return P

    def sample(self, B: int, H: int, W: int) -> List[Tuple[int, np.ndarray]]:
        """[(crop size, params int32 [count * B, 24])] for the global group (2 crops) and the local group."""
        g1 = self._sample_kind(B, H, W, self.gscale, 1.0, 0.0)                    # global_transfo1 :110-121
        g2 = self._sample_kind(B, H, W, self.gscale, 0.1, 0.2 if self.solarization else 0.0)   # :123-135
        groups = [(self.gsize, np.concatenate([g1, g2]))]
        if self.local_crops_number:
            groups.append((self.lsize, self._sample_kind(self.local_crops_number * B, H, W, self.lscale, 0.5, 0.0,
                                                         nsrc=B)))               # local_transfo :138-149
        return groups

    # ----------------------------------------------------------------------------- device: the pixels
    def apply(self, images_u8: torch.Tensor, S: int, params: np.ndarray) -> torch.Tensor:
        """One group of crops: uint8 [B,H,W,3] + params [n,24] -> fp32 [n,3,S,S]."""
        if not (images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4):
            raise ValueError("apply() takes a device uint8 [B,H,W,3] batch")
        B, H, W, _ = images_u8.shape
        n = params.shape[0]
        if not (params.dtype == np.int32 and params.shape[1] == PW):
            raise ValueError(f"params must be int32 [n, {PW}]")
        if not ((params[:, 0] >= 0).all() and (params[:, 0] < B).all()):
            raise ValueError("source image index outside the batch")
        if not ((params[:, 1] >= 0).all() and (params[:, 2] >= 0).all() and (params[:, 3] > 0).all()
                and (params[:, 4] > 0).all() and (params[:, 1] + params[:, 3] <= H).all()
                and (params[:, 2] + params[:, 4] <= W).all()):
            raise ValueError("crop box outside the source image")
        big = int(max(params[:, 3].max(), params[:, 4].max()))
        kmax = int(math.ceil(2.0 * max(big / S, 1.0))) * 2 + 1
        tmp_rows = int(params[:, 3].max())
        host = torch.from_numpy(params).pin_memory()
        dev = host.to(images_u8.device, non_blocking=True)
        crops = ops.aug_resized_crop(images_u8.contiguous(), dev, S, kmax, tmp_rows)
        sums = torch.zeros((n,), dtype=torch.int32, device=images_u8.device)
        for slot in range(5):
            opcol = params[:, 6 + slot]
            if not opcol.any():
                continue
            if (opcol == OP_CONTRAST).any():
                ops.aug_gray_sums(crops, sums, dev, slot)
            ops.aug_color_op(crops, dev, slot, sums)
        if params[:, 18].any():
            other = torch.empty_like(crops)
            a, b = crops, other
            for axis in (0, 0, 0, 1, 1, 1):     # an even number of passes: the result is back in `crops`
                ops.aug_box_blur(a, b, dev, axis)
                a, b = b, a
        if params[:, 6 + SLOT_SOLARIZE].any():
            ops.aug_color_op(crops, dev, SLOT_SOLARIZE, sums)
        return ops.to_tensor_normalize(crops, self.mean, self.std)

    def __call__(self, images_u8: torch.Tensor) -> List[torch.Tensor]:
        B, H, W, _ = images_u8.shape
        out: List[torch.Tensor] = []
        for S, params in self.sample(B, H, W):
            t = self.apply(images_u8, S, params)
            out.extend(t[k * B:(k + 1) * B] for k in range(t.shape[0] // B))
        return out


class DeviceAugmentingLoader:
    """Wraps a loader that yields `(images_u8 [B,H,W,3], labels)` (host tensors -- ideally pinned -- or device
    tensors) and yields `(crops, labels)` as the reference's loader does (dino.py:127-138 with
    PatchDataAugmentationDINO as the dataset transform), the crops made on the GPU `depth` batches ahead:
    the H2D copy of the raw images (a twelfth of the bytes of the fp32 crops) and the augmentation kernels
    run on their own stream under the training step of the batch before."""

    def __init__(self, loader, aug: DeviceAugmentationDINO, device, depth: int = 2):
        self.loader, self.aug, self.device, self.depth = loader, aug, torch.device(device), max(1, int(depth))
        if self.device.type != "cuda":
            raise RuntimeError("DeviceAugmentingLoader runs the augmentation kernels on the GPU; there is no host path")
        self._stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _issue(self, batch):
        images, labels = batch
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._stream):
            if images.is_cuda:
                self._stream.wait_stream(cur)       # produced on the consumer's stream
                dev = images
                dev.record_stream(self._stream)     # the caller may drop the batch while the kernels still read it
            else:
                host = images if images.is_pinned() else images.pin_memory()
                dev = host.to(self.device, non_blocking=True)
            crops = self.aug(dev)
            done = torch.cuda.Event()
            done.record(self._stream)
        return crops, labels, done

    def __iter__(self):
        it = iter(self.loader)
        queue = deque()

        def fill():
            while len(queue) < self.depth:
                try:
                    queue.append(self._issue(next(it)))
                except StopIteration:
                    return
        fill()
        while queue:
            crops, labels, done = queue.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            for c in crops:                         # allocated under the side stream, read on this one
                c.record_stream(cur)
            yield crops, labels
            fill()
