"""Flat parameter storage: every student parameter (and its grad, AdamW moments, teacher twin
and bf16 shadows) is a view into ONE contiguous buffer per kind.

MI355X-first layout: optimiser / EMA / grad-norm become single streaming launches over
DINO_CHUNK-element chunks, gradient buckets for the data-parallel all-reduce are plain slices
(no pack/unpack copies), and 288 GB of HBM makes the duplicated shadows free.
"""
from collections import OrderedDict

import numpy as np
import torch

from .._lib import DINO_CHUNK

ALIGN = 8  # elements: 32 B for fp32, 16 B for the bf16 shadows


def _pad(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


def build_chunks(padded_sizes, offsets):
    """(chunks int64 [n,3] = (tensor, start, len), tensor_chunk_begin int32 [ntensors+1])."""
    rows, begin = [], [0]
    for t, (off, n) in enumerate(zip(offsets, padded_sizes)):
        for s in range(0, n, DINO_CHUNK):
            rows.append((t, off + s, min(DINO_CHUNK, n - s)))
        begin.append(len(rows))
    return (torch.tensor(rows, dtype=torch.int64).reshape(-1, 3),
            torch.tensor(begin, dtype=torch.int32))


class FlatLayout:
    """Offsets of named tensors inside a flat buffer (order = parameter registration order)."""

    def __init__(self, shapes: "OrderedDict[str, tuple]"):
        self.shapes = OrderedDict((k, tuple(v)) for k, v in shapes.items())
        self.names = list(self.shapes)
        self.slices = OrderedDict()
        off = 0
        padded, offsets = [], []
        for name, shape in self.shapes.items():
            n = int(np.prod(shape)) if len(shape) else 1
            self.slices[name] = (off, n)
            offsets.append(off)
            padded.append(_pad(n))
            off += _pad(n)
        self.total = off
        self.chunks, self.tensor_chunk_begin = build_chunks(padded, offsets)

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, n = self.slices[name]
        return flat[off:off + n].view(self.shapes[name])

    def index(self, name: str) -> int:
        return self.names.index(name)
