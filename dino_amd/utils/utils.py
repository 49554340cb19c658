"""Step helpers with the reference's names and semantics (dino/utils/utils.py:25-106,184-227)."""
import random

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def get_params_groups(model):
    """Two AdamW groups: biases and 1-D (norm) parameters are not weight-decayed
    (utils.py:25-36); frozen parameters are skipped."""
    regularized, not_regularized = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if name.endswith(".bias") or len(param.shape) == 1:
            not_regularized.append(param)
        else:
            regularized.append(param)
    return [{"params": regularized}, {"params": not_regularized, "weight_decay": 0.0}]


def has_batchnorms(model):
    bn_types = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)
    return any(isinstance(m, bn_types) for _, m in model.named_modules())


def cosine_scheduler(base_value, final_value, nepochs, niter_per_ep, warmup_epochs=0,
                     start_warmup_value=0):
    """Linear warm-up then half-cosine, float64 numpy (utils.py:184-204)."""
    warmup_iters = warmup_epochs * niter_per_ep
    warmup = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 \
        else np.array([])
    iters = np.arange(nepochs * niter_per_ep - warmup_iters)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    sched = np.concatenate((warmup, sched))
    assert len(sched) == nepochs * niter_per_ep
    return sched


def fix_random_seeds(seed=31):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def _unwrap(model):
    return model.module if hasattr(model, "module") and hasattr(model.module, "store") else model


class GradStatsWork:
    """Device-side result of the grad-norm / clip kernels: norms[ntensors], coef[ntensors],
    stats = (total norm, max norm, #clipped, -).  .result() does the ONE host sync."""

    def __init__(self, store, has_grad, norms, coef, stats, clip, model_name):
        self.store, self.has_grad = store, has_grad
        self.norms, self.coef, self.stats = norms, coef, stats
        self.clip, self.model_name = clip, model_name

    def result(self):
        s = self.stats.tolist()
        n = int(sum(self.has_grad))
        return {f"{self.model_name}_grad_norm": s[0], f"{self.model_name}_grad_max": s[1],
                f"{self.model_name}_clip_fraction": (s[2] / n if n and self.clip else 0.0)}


def grad_stats_async(model, clip=None, model_name="model", materialize=True) -> GradStatsWork:
    """Per-tensor grad L2 norms and per-tensor clip coefficients in two launches, no host sync.
    materialize=True also scales the gradients in place (p.grad.data.mul_ of utils.py:88);
    the fused optimiser can instead consume `coef` directly."""
    st = _unwrap(model).store()
    if st.pending_bwd:
        raise RuntimeError(
            f"{st.pending_bwd} recorded backbone pass(es) of this model were never back-propagated: its "
            "gradients are incomplete (run inference under torch.no_grad(), or drop the outputs of forward "
            "passes you do not back-propagate)")
    names = st.layout.names
    has = [1 if (st.params[k].requires_grad and st.params[k].grad is not None) else 0 for k in names]
    dev = st.device
    has_t = torch.tensor(has, dtype=torch.int32).to(dev, non_blocking=True)
    nt = len(names)
    sq = torch.empty(st.chunks.shape[0], dtype=torch.float32, device=dev)
    norms = torch.empty(nt, dtype=torch.float32, device=dev)
    coef = torch.empty(nt, dtype=torch.float32, device=dev)
    stats = torch.zeros(4, dtype=torch.float32, device=dev)
    ops.grad_sqnorm(st.grad_flat, st.chunks, sq)
    ops.clip_coef(sq, st.tensor_chunk_begin, has_t, clip if clip else 0.0, norms, coef, stats)
    if clip and materialize:
        ops.grad_scale(st.grad_flat, st.chunks, coef)
    return GradStatsWork(st, has, norms, coef, stats, clip, model_name)


def compute_gradient_stats(model, clip=None, model_name="model"):
    """Reference signature and return value (utils.py:59-98): total / max per-parameter grad
    norm and the fraction of parameters clipped; clips in place, per parameter."""
    return grad_stats_async(model, clip, model_name, materialize=True).result()


def clip_gradients(model, clip):
    w = grad_stats_async(model, clip, "m", materialize=True)
    return [x for x, h in zip(w.norms.tolist(), w.has_grad) if h]


def cancel_gradients_last_layer(epoch, model, freeze_last_layer):
    """Drop the prototype layer's grads while iteration < freeze_last_layer (utils.py:101-106):
    AdamW then skips those tensors entirely."""
    if epoch >= freeze_last_layer:
        return
    for n, p in model.named_parameters():
        if "last_layer" in n:
            p.grad = None


def resume_from_checkpoint(ckpt_path, verbose: bool = True, **kwargs):
    """Restore training objects from a checkpoint file (reference utils/utils.py:139-181): each
    keyword names a checkpoint entry and the object to load it into (`student=model,
    optimizer=opt, dino_loss=loss`).  A missing file returns 0; otherwise the stored epoch
    (or iteration).  "module." prefixes left by a data-parallel wrapper are dropped, modules load
    non-strictly, optimisers strictly, as in the reference.  The file is read with
    weights_only=True (tensors, numbers, strings, containers): nothing in it is executed."""
    import logging
    from pathlib import Path
    log = logging.getLogger("dino")
    if not Path(ckpt_path).is_file():
        return 0
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    for key, target in kwargs.items():
        if target is None:
            continue
        if key not in ckpt:
            if verbose:
                log.warning("=> key '%s' not found in checkpoint: '%s'", key, ckpt_path)
            continue
        state = ckpt[key]
        if isinstance(state, dict):
            torch.nn.modules.utils.consume_prefix_in_state_dict_if_present(state, "module.")
        if isinstance(target, torch.nn.Module):
            inner = _unwrap(target)
            msg = inner.load_state_dict(state, strict=False)
        else:
            msg = target.load_state_dict(state)
        if verbose:
            log.info("=> loaded '%s' from checkpoint '%s' (%s)", key, ckpt_path, msg)
    return ckpt.get("epoch", ckpt.get("iteration", 0))
