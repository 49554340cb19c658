"""Eval-mode feature extraction over a (possibly sharded) loader, reference dino/eval/features.py:
`extract_features` (:12-79) and `extract_multiple_features` (:82-178), same signatures and results.

The loader yields (index, image, label); every rank embeds its shard, the rows of all ranks are
gathered over the process group (RCCL on GPUs) and the main process scatters them into an
[N_dataset, D] matrix by dataset index.  Per batch the reference issues one all_gather for the indices
and one per model; here the features of all models travel as ONE packed [b, sum(D)] all-gather (xGMI
collectives of a few hundred KiB are latency-bound, so fewer and larger is the point) and the labels
ride with the indices, so the returned labels are the GLOBAL labels in dataset order (the reference
returns the local rank's labels in loader order, which does not line up with its own global feature
matrix when more than one rank runs)."""
from typing import Dict, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from ..distributed import get_global_size, is_enabled_and_multiple_gpus, is_main_process


def gather_rows(index: torch.Tensor, label: torch.Tensor, feats: Sequence[torch.Tensor]):
    """All ranks contribute b rows; returns (index_all [W*b], label_all [W*b], [feats_all [W*b, D_i]]).
    Two collectives regardless of the number of feature tensors."""
    world = get_global_size()
    meta = torch.stack([index.to(torch.int64), label.to(torch.int64).to(index.device)], dim=1).contiguous()
    meta_all = torch.empty((world * meta.shape[0], 2), dtype=meta.dtype, device=meta.device)
    dist.all_gather_into_tensor(meta_all, meta)
    dims = [f.shape[1] for f in feats]
    packed = torch.cat([f.float() for f in feats], dim=1).contiguous() if len(feats) > 1 \
        else feats[0].float().contiguous()
    packed_all = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=packed.dtype,
                             device=packed.device)
    dist.all_gather_into_tensor(packed_all, packed)
    return meta_all[:, 0], meta_all[:, 1], list(packed_all.split(dims, dim=1))


@torch.no_grad()
def _extract(models: Sequence[nn.Module], loader, device) -> Tuple[list, torch.Tensor]:
    for m in models:
        m.eval()
    distributed = is_enabled_and_multiple_gpus()
    n_total = len(loader.dataset)
    stores = None
    labels = torch.full((n_total,), -1, dtype=torch.int64)
    for index, img, label in loader:
        img = img.to(device, non_blocking=True)
        index = index.to(device, non_blocking=True)
        feats = [m(img).float() for m in models]
        label = torch.as_tensor(label)
        if distributed:
            index, label, feats = gather_rows(index, label, feats)
        if stores is None and (is_main_process() or not distributed):
            stores = [torch.zeros(n_total, f.shape[-1], device=device) for f in feats]
        if stores is not None:
            for s, f in zip(stores, feats):
                s.index_copy_(0, index, f)
            labels[index.cpu()] = label.cpu()
    return stores, labels.long()


@torch.no_grad()
def extract_features(model: nn.Module, loader, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """features [N, D] (main process; None elsewhere when distributed), labels [N]."""
    stores, labels = _extract([model], loader, device)
    return (None if stores is None else stores[0]), labels


@torch.no_grad()
def extract_multiple_features(student: nn.Module, teacher: nn.Module, loader,
                              device) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """{'student': [N, D], 'teacher': [N, D]} (main process), labels [N]."""
    stores, labels = _extract([student, teacher], loader, device)
    s, t = (None, None) if stores is None else stores
    return {"student": s, "teacher": t}, labels
