"""CPU restatement of the input normalisation (TEST INFRASTRUCTURE ONLY; same import rule as
oracle/dino_oracle.py).

PARITY UNPINNED: the reference builds `Compose([MaybeToTensor(), Normalize(mean, std)])`
(dino/data/augmentations.py:102-107, defaults :10-11) out of torchvision, which is a dependency of
the reference (requirements.in:8, unpinned version) and is importable neither here nor on the GPU
box; the reference's tests hold no vector for it.  This restates torchvision's published algorithm:
`transforms.functional.to_tensor` on a uint8 HWC image = permute to CHW, convert to float32, divide
by 255; `transforms.functional.normalize` = `tensor.sub_(mean[:, None, None]).div_(std[:, None, None])`
with mean/std as float32 tensors."""
import torch


def to_tensor_normalize(u8_bhwc: torch.Tensor, mean, std) -> torch.Tensor:
    """uint8 [B,H,W,C] -> fp32 [B,C,H,W] (augmentations.py:102-107)."""
    assert u8_bhwc.dtype == torch.uint8
    x = u8_bhwc.permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32)[None, :, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[None, :, None, None]
    return x.sub_(m).div_(s)
