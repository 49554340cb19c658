"""Writes tests/golden/augment.npz: outputs of Pillow itself (the library the reference's torchvision
transforms call, dino/data/augmentations.py:75-157) for a small deterministic image batch -- every single
operation with a few parameters, and whole crops described by parameter rows (include/dino_hip.h layout).
torchvision is not importable here, so each transform is spelled as the Pillow call its PIL backend makes
(torchvision/transforms/_functional_pil.py).  Run from the repo root: python tools/make_golden_augment.py"""
import os
import sys

import numpy as np
from PIL import Image, ImageEnhance, ImageFilter, ImageOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import augment_oracle as A  # noqa: E402  (only for the blur constants / hue shift of a row)

OUT = os.path.join(ROOT, "tests", "golden", "augment.npz")
H, W, S = 72, 88, 32


def source_images(n=3, seed=7):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        low = rng.integers(0, 256, (H // 8 + 2, W // 8 + 2, 3)).astype(np.uint8)
        im = np.asarray(Image.fromarray(low).resize((W, H), Image.BILINEAR)).astype(np.int32)
        out.append(np.clip(im + rng.integers(-25, 26, (H, W, 3)), 0, 255).astype(np.uint8))
    return np.stack(out)


def fbits(x):
    return int(np.array([x], dtype=np.float32).view(np.int32)[0])


def row(src, top, left, h, w, flip=0, ops=(), gray=0, blur=None, solarize=0):
    r = np.zeros(24, dtype=np.int32)
    r[0:6] = (src, top, left, h, w, flip)
    for slot, (op, val) in enumerate(ops):
        r[6 + slot] = op
        r[12 + slot] = A.hue_shift_u8(val) if op == 4 else fbits(val)
    r[10] = 5 if gray else 0
    if blur is not None:
        r[18] = 1
        r[19], r[20], r[21] = A.box_blur_params(A.gaussian_box_radius(blur))
    r[11] = 6 if solarize else 0
    return r


def pil_hue(img, shift):
    h, s, v = img.convert("HSV").split()
    nh = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        nh += np.uint8(shift)
    return Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")


def pil_crop(src, r, blur_radius):
    """The reference's transform chain on one PIL image with the row's parameters."""
    img = Image.fromarray(src[r[0]])
    img = img.crop((r[2], r[1], r[2] + r[4], r[1] + r[3])).resize((S, S), Image.BICUBIC)   # F.resized_crop
    if r[5]:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)                                           # F.hflip
    for slot in range(5):
        op, bits = int(r[6 + slot]), int(r[12 + slot])
        fac = float(np.array([bits], dtype=np.int32).view(np.float32)[0])
        if op == 1:
            img = ImageEnhance.Brightness(img).enhance(fac)
        elif op == 2:
            img = ImageEnhance.Contrast(img).enhance(fac)
        elif op == 3:
            img = ImageEnhance.Color(img).enhance(fac)
        elif op == 4:
            img = pil_hue(img, bits)
        elif op == 5:
            img = img.convert("L").convert("RGB")          # F.rgb_to_grayscale(num_output_channels=3)
    if r[18]:
        img = img.filter(ImageFilter.GaussianBlur(radius=blur_radius))
    if r[11] == 6:
        img = ImageOps.solarize(img)
    return np.asarray(img)


def main():
    src = source_images()
    out = {"src": src, "S": np.int32(S)}
    pil = Image.fromarray(src[0])
    # (1) single operations on image 0
    out["op/gray"] = np.asarray(pil.convert("L"))
    out["op/hsv"] = np.asarray(pil.convert("HSV"))
    out["op/hsv2rgb"] = np.asarray(Image.fromarray(src[1], "HSV").convert("RGB"))
    out["op/solarize"] = np.asarray(ImageOps.solarize(pil))
    facs = np.array([0.6, 0.83, 1.0, 1.17, 1.4], dtype=np.float64)
    out["op/factors"] = facs
    for i, f in enumerate(facs):
        out[f"op/brightness{i}"] = np.asarray(ImageEnhance.Brightness(pil).enhance(f))
        out[f"op/contrast{i}"] = np.asarray(ImageEnhance.Contrast(pil).enhance(f))
        out[f"op/saturation{i}"] = np.asarray(ImageEnhance.Color(pil).enhance(0.5 + f / 2))
    hues = np.array([-0.1, -0.043, 0.0, 0.02, 0.1])
    out["op/hues"] = hues
    for i, f in enumerate(hues):
        out[f"op/hue{i}"] = np.asarray(pil_hue(pil, A.hue_shift_u8(f)))
    radii = np.array([0.1, 0.5, 0.77, 1.0, 1.31, 1.5, 1.99, 2.0])
    out["op/radii"] = radii
    for i, r in enumerate(radii):
        out[f"op/blur{i}"] = np.asarray(pil.filter(ImageFilter.GaussianBlur(radius=float(r))))
    boxes = np.array([(0, 0, H, W, 32), (5, 9, 40, 50, 32), (10, 20, 32, 32, 32), (3, 4, 17, 13, 32),
                      (0, 0, H, W, 48), (20, 30, 31, 47, 24), (1, 1, 70, 86, 16)], dtype=np.int32)
    out["op/boxes"] = boxes
    for i, (t, l, h, w, s) in enumerate(boxes):
        out[f"op/resize{i}"] = np.asarray(pil.crop((l, t, l + w, t + h)).resize((s, s), Image.BICUBIC))
    # (2) whole crops from parameter rows (S = 32)
    blur_r = [None, 1.3, None, 0.4, 2.0, None, 0.1, 1.7, None, 0.9]
    rows = [
        row(0, 0, 0, H, W),
        row(1, 5, 9, 40, 50, flip=1, ops=[(1, 1.25), (2, 0.7), (3, 1.15), (4, 0.07)], blur=blur_r[1]),
        row(2, 10, 20, 32, 32, ops=[(4, -0.09), (3, 0.85), (2, 1.33), (1, 0.66)], gray=1),
        row(0, 3, 4, 17, 13, flip=1, ops=[(2, 1.4), (1, 1.4), (4, 0.1), (3, 1.2)], blur=blur_r[3], solarize=1),
        row(1, 20, 30, 31, 47, ops=[(3, 0.8), (4, -0.1), (1, 0.6), (2, 0.6)], blur=blur_r[4]),
        row(2, 1, 1, 70, 86, gray=1, solarize=1),
        row(0, 30, 2, 25, 80, flip=1, blur=blur_r[6]),
        row(1, 0, 40, 72, 48, ops=[(1, 1.0), (2, 1.0), (3, 1.0), (4, 0.0)], blur=blur_r[7]),
        row(2, 36, 44, 36, 44, ops=[(2, 0.95), (4, 0.033), (1, 1.07), (3, 0.97)], gray=1, solarize=1),
        row(0, 8, 8, 56, 64, flip=1, ops=[(4, 0.05), (1, 0.9), (3, 1.1), (2, 1.2)], blur=blur_r[9]),
    ]
    rows = np.stack(rows)
    out["rows"] = rows
    out["rows_blur_radius"] = np.array([np.nan if b is None else b for b in blur_r])
    out["crops"] = np.stack([pil_crop(src, r, blur_r[i]) for i, r in enumerate(rows)])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", ()) for k, v in out.items() if not k.startswith("op/")})


if __name__ == "__main__":
    main()
