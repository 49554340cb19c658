"""Bicubic resize of the positional grid as a fixed sparse linear map.

`interpolate_pos_encoding` (reference dino/models/vision_transformer.py:331-359) calls
F.interpolate(mode="bicubic", align_corners=False, recompute_scale_factor=True) on the
[side, side] grid of patch position embeddings every forward.  For a given (source side,
destination side) that is a constant [n_dst, n_src] matrix with 16 non-zeros per row; we
build its taps once on the host (float64) and apply them with a gather kernel.
"""
import math
from functools import lru_cache

import numpy as np

_A = np.float32(-0.75)  # PyTorch's cubic convolution coefficient


def _cc1(x):
    return ((_A + np.float32(2.0)) * x - (_A + np.float32(3.0))) * x * x + np.float32(1.0)


def _cc2(x):
    return ((_A * x - np.float32(5.0) * _A) * x + np.float32(8.0) * _A) * x - np.float32(4.0) * _A


def _taps_1d(n_src: int, n_dst: int):
    """idx [n_dst,4] (clamped), w [n_dst,4] for align_corners=False with scale n_src/n_dst."""
    f = np.float32  # mirror ATen's float32 coordinate/coefficient arithmetic
    scale = f(n_src) / f(n_dst)
    idx = np.zeros((n_dst, 4), dtype=np.int64)
    w = np.zeros((n_dst, 4), dtype=np.float64)
    for o in range(n_dst):
        real = scale * (f(o) + f(0.5)) - f(0.5)
        i0 = math.floor(real)
        t = f(real - f(i0))
        w[o] = (_cc2(t + f(1.0)), _cc1(t), _cc1(f(1.0) - t), _cc2(f(2.0) - t))
        idx[o] = np.clip(np.arange(i0 - 1, i0 + 3), 0, n_src - 1)
    return idx, w


@lru_cache(maxsize=None)
def bicubic_taps(src_side: int, dst_side: int):
    """(idx int32 [dst_side^2, 16], w float32 [dst_side^2, 16]) over the flattened
    row-major [src_side, src_side] grid."""
    i1, w1 = _taps_1d(src_side, dst_side)
    n = dst_side * dst_side
    idx = np.zeros((n, 16), dtype=np.int32)
    w = np.zeros((n, 16), dtype=np.float32)
    for oy in range(dst_side):
        for ox in range(dst_side):
            o = oy * dst_side + ox
            idx[o] = (i1[oy][:, None] * src_side + i1[ox][None, :]).reshape(-1)
            w[o] = (w1[oy][:, None] * w1[ox][None, :]).reshape(-1)
    return idx, w


def dense_matrix(src_side: int, dst_side: int) -> np.ndarray:
    idx, w = bicubic_taps(src_side, dst_side)
    m = np.zeros((dst_side * dst_side, src_side * src_side), dtype=np.float64)
    for o in range(idx.shape[0]):
        np.add.at(m[o], idx[o], w[o].astype(np.float64))
    return m
