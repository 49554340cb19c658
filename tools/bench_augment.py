"""Throughput of the device-side multi-crop augmentation at the bench's shape (256 source images of 256 x 256,
2 x 224 + 10 x 96 crops) with Pillow on the host cores timed beside it (the reference's path: torchvision
transforms -> these Pillow calls, in DataLoader workers).  python tools/bench_augment.py [B]"""
import sys
import time

import numpy as np
import torch

from dino_amd.data import DeviceAugmentationDINO

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
low = torch.from_numpy(rng.integers(0, 256, (B, 34, 34, 3)).astype(np.float32)).permute(0, 3, 1, 2)
src = torch.nn.functional.interpolate(low, size=(256, 256), mode="bilinear").permute(0, 2, 3, 1)
src = (src + torch.from_numpy(rng.integers(-30, 31, (B, 256, 256, 3)).astype(np.float32))).clamp(0, 255).to(torch.uint8)
dev = src.cuda()
aug = DeviceAugmentationDINO(224, 96, (0.32, 1.0), (0.05, 0.32), 10, solarization=True, seed=1)
for _ in range(3):
    aug(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    t = time.perf_counter()
    groups = aug.sample(B, 256, 256)
    th = time.perf_counter() - t
    crops = aug(dev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
out_bytes = sum(c.numel() * 4 for c in crops)
print(f"device: {dt * 1e3:.2f} ms per batch of {B} images x 12 crops = {B / dt:.0f} img/s "
      f"(host parameter draw {th * 1e3:.2f} ms of it); fp32 crops written {out_bytes / 1e6:.0f} MB", flush=True)

# the reference's CPU path on ONE host core: the same Pillow calls through the oracle's fixture generator
from PIL import Image, ImageEnhance, ImageFilter  # noqa: E402
img = Image.fromarray(src[0].numpy())
t0 = time.perf_counter()
m = 20
for k in range(m):
    for S, scale in [(224, 0.6)] * 2 + [(96, 0.2)] * 10:
        w = int(256 * scale ** 0.5)
        c = img.crop((3, 5, 3 + w, 5 + w)).resize((S, S), Image.BICUBIC).transpose(Image.FLIP_LEFT_RIGHT)
        c = ImageEnhance.Brightness(c).enhance(1.2)
        c = ImageEnhance.Contrast(c).enhance(0.8)
        c = ImageEnhance.Color(c).enhance(1.1)
        h, s, v = c.convert("HSV").split()
        c = Image.merge("HSV", (h.point(lambda x: (x + 9) & 255), s, v)).convert("RGB")
        c = c.filter(ImageFilter.GaussianBlur(radius=1.1))
        a = (np.asarray(c).astype(np.float32) / 255.0 - 0.45) / 0.22
dt_cpu = (time.perf_counter() - t0) / m
print(f"Pillow, one host core, every op applied: {dt_cpu * 1e3:.1f} ms per image (12 crops) = {1 / dt_cpu:.0f} img/s/core",
      flush=True)
