"""Seed-free deterministic tensor fills (TEST INFRASTRUCTURE, see oracle/__init__.py).

Parity runs must feed *identical* weights and images to the reference (imported
only in the authoring container by tools/make_golden.py), to the CPU oracle and
to the HIP path on the GPU box.  torch RNG streams are not guaranteed to agree
across versions/devices, so every value is a pure function of
(tag, flat index): splitmix64 hash -> two uniforms -> Box-Muller in float64.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _uniform01(tag: str, n: int, stream: int) -> np.ndarray:
    """n float64 uniforms in (0, 1), a pure function of (tag, stream, index)."""
    base = np.uint64(zlib.crc32(tag.encode()) & 0xFFFFFFFF) << np.uint64(32)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(stream)
        bits = _splitmix64(_splitmix64(idx ^ base) + base)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def det_normal(shape, tag: str, std: float = 1.0, mean: float = 0.0,
               clip: float = None) -> torch.Tensor:
    """Deterministic N(mean, std^2) float32 tensor; optional clip at +-clip*std."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = _uniform01(tag, n, 0)
    u2 = _uniform01(tag, n, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    if clip is not None:
        z = np.clip(z, -clip, clip)
    out = (mean + std * z).astype(np.float32).reshape(shape)
    return torch.from_numpy(out)


def det_uniform(shape, tag: str, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = _uniform01(tag, n, 0)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def fill_state_dict_(sd: dict, tag: str = "w") -> dict:
    """Deterministic 'training-like' fill of a DINO student state dict, in place.

    Matrices / conv kernels / pos_embed / cls_token ~ N(0, 0.02) clipped at 2 std
    (the reference's trunc_normal_, dino/models/vision_transformer.py:318-329,
    710-714); biases get small non-zero values and LayerNorm weights values near
    1 so that every bias / affine code path is exercised; weight_g is near 1
    (reference fills 1, vision_transformer.py:706).
    """
    for name, t in sd.items():
        key = f"{tag}:{name}"
        if name.endswith("weight_g"):
            v = det_normal(t.shape, key, std=0.05, mean=1.0)
        elif name.endswith("weight_v"):
            v = det_normal(t.shape, key, std=0.02, clip=2.0)
        elif name.endswith(".bias"):
            v = det_normal(t.shape, key, std=0.02)
        elif t.ndim == 1:  # LayerNorm weight
            v = det_normal(t.shape, key, std=0.05, mean=1.0)
        else:
            v = det_normal(t.shape, key, std=0.02, clip=2.0)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return sd


def det_crops(batch: int, sizes, tag: str = "img"):
    """List of deterministic N(0,1) crops [batch,3,s,s], one per entry of sizes."""
    return [det_normal((batch, 3, s, s), f"{tag}:{i}") for i, s in enumerate(sizes)]
