"""Process-wide defaults of the dino_amd runtime."""
import os

import torch

_DEFAULT_DTYPE = {"bf16": torch.bfloat16, "fp32": torch.float32}[os.environ.get("DINO_AMD_DTYPE", "bf16")]


def get_default_dtype() -> torch.dtype:
    """Compute dtype new modules are created with: torch.bfloat16 (activations and weight
    shadows in bf16, fp32 accumulation / master weights / grads) or torch.float32."""
    return _DEFAULT_DTYPE


def set_default_dtype(dtype: torch.dtype) -> None:
    global _DEFAULT_DTYPE
    if dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("compute dtype must be torch.bfloat16 or torch.float32")
    _DEFAULT_DTYPE = dtype
