"""The DINO training step in the reference's order (dino/utils/train_utils.py:52-162):
schedule write -> H2D -> teacher fwd (2 global views) -> student fwd (all views) -> loss
(+centre update) -> backward (+gradient all-reduce) -> per-parameter clip -> last-layer grad
cancel -> AdamW -> teacher EMA.  Differences are confined to HOW it runs: one host sync per
logging interval instead of ~160 per step, fused optimiser/EMA, flat gradient buckets."""
import math
import os
import sys

import torch

import dino_amd.distributed as distributed
from ..optim import FusedAdamW, update_teacher
from .utils import cancel_gradients_last_layer, grad_stats_async


# Teacher and student forward are independent until the loss: the teacher runs on a side stream, so
# that the partly filled last tile rounds of either pass's persistent kernels are filled by the other
# (56.7 -> 55.8 ms at config #2).  DINO_TEACHER_STREAM=0 keeps everything on one stream.
_TEACHER_STREAM = os.environ.get("DINO_TEACHER_STREAM", "1") != "0"

def train_step(student, teacher, dino_loss, optimizer, images, iteration, lr_schedule, wd_schedule,
               momentum_schedule, clip_grad, freeze_last_layer_iter, sync_stats=False):
    """One step on device-resident `images` (list of crops).  Returns (loss tensor, stats work).
    No host synchronisation unless sync_stats."""
    for i, group in enumerate(optimizer.param_groups):          # train_utils.py:56-59
        group["lr"] = float(lr_schedule[iteration])          # plain floats: the optimizer state_dict
        if i == 0:                                            # stays loadable with weights_only=True
            group["weight_decay"] = float(wd_schedule[iteration])
    if _TEACHER_STREAM and images[0].is_cuda:
        from ..engine.store import side_stream
        _SIDE = side_stream(images[0].device)
        main = torch.cuda.current_stream()
        _SIDE.wait_stream(main)
        with torch.cuda.stream(_SIDE), torch.no_grad():
            teacher_output = teacher(images[:2])
        student_output = student(images)
        main.wait_stream(_SIDE)
        teacher_output.record_stream(main)
    else:
        with torch.no_grad():
            teacher_output = teacher(images[:2])                      # :75-77
        student_output = student(images)                              # :85
    loss = dino_loss(student_output, teacher_output, iteration)   # :87
    optimizer.zero_grad()                                         # :115
    loss.backward()                                               # :118 (+ flat-bucket all-reduce)
    fused = isinstance(optimizer, FusedAdamW)
    work = grad_stats_async(student, clip=clip_grad if clip_grad else None, model_name="student",
                            materialize=not fused)               # :120-122
    cancel_gradients_last_layer(iteration, student, freeze_last_layer_iter)   # :130
    m = momentum_schedule[iteration]
    if fused:
        optimizer.step(ema_momentum=m, clip_coef=work.coef if clip_grad else None)  # :131 + :152-162
    else:
        optimizer.step()
        update_teacher(student, teacher, m)
    return loss, (work.result() if sync_stats else work)


def train_one_epoch(student, teacher, teacher_without_ddp, dino_loss, data_loader, optimizer,
                    lr_schedule, wd_schedule, momentum_schedule, epoch, nepochs, fp16_scaler,
                    clip_grad, freeze_last_layer_iter, save_every, checkpoint_dir, gpu_id,
                    domain_classifier=None, domain_loss_fn=None, grl=None, dc_clip_grad: float = 0,
                    log_every: int = 50):
    """Reference signature (train_utils.py:14-36).  fp16_scaler is accepted and ignored: the
    bf16 path needs no loss scaling.  The adversarial branch is out of scope."""
    if domain_classifier is not None:
        raise NotImplementedError("domain-adversarial branch (adversarial.enable) is out of scope")
    if distributed.is_enabled() and hasattr(data_loader, "sampler") and hasattr(data_loader.sampler, "set_epoch"):
        data_loader.sampler.set_epoch(epoch)
    device = torch.device("cuda" if gpu_id == -1 else f"cuda:{gpu_id}")
    loss_sum = torch.zeros((), dtype=torch.float32, device=device)
    n = 0
    last = None
    from ..data import CropPrefetcher
    # host -> HBM two batches ahead on a copy stream (the reference copies at the top of the step,
    # train_utils.py:61-67)
    feed = data_loader if isinstance(data_loader, CropPrefetcher) else CropPrefetcher(data_loader, device)
    for it, (images, _labels) in enumerate(feed):
        iteration = len(data_loader) * epoch + it
        loss, _ = train_step(student, teacher_without_ddp, dino_loss, optimizer, images, iteration,
                             lr_schedule, wd_schedule, momentum_schedule, clip_grad,
                             freeze_last_layer_iter)
        loss_sum += loss.detach()
        n += 1
        last = loss
        saving = bool(distributed.is_main_process() and save_every and iteration % save_every == 0
                      and checkpoint_dir)
        if (it + 1) % log_every == 0 or it + 1 == len(data_loader) or saving:
            # the step's only host sync.  The reference checks every step before the update
            # (:108-112); here the running sum carries the flag (a non-finite loss makes it non-finite
            # for good), it is read every log_every steps and ALWAYS before a checkpoint is written,
            # so weights that a NaN step has touched are never saved.
            val = loss_sum.item()
            if not math.isfinite(val):
                print("Loss is {}, stopping training".format(last.item()), flush=True)
                sys.exit(1)
        if saving:
            from pathlib import Path
            ckpt = {"iteration": iteration, "student": student.state_dict(),
                    "teacher": teacher.state_dict(), "optimizer": optimizer.state_dict(),
                    "dino_loss": dino_loss.state_dict()}
            torch.save(ckpt, Path(checkpoint_dir, f"iter_{iteration:06}.pt"))
    # global averages, as the reference's MetricLogger.synchronize_between_processes (:191): the loss
    # is averaged over the steps of every rank; lr / wd are the last step's values (the reference's
    # meters average them over the epoch -- they are schedule values, identical on all ranks)
    tot = torch.stack([loss_sum.double(), torch.tensor(float(n), dtype=torch.float64, device=device)])
    if distributed.is_enabled_and_multiple_gpus():
        distributed.all_reduce(tot)
    tot = tot.tolist()
    stats = {"loss": tot[0] / max(tot[1], 1.0), "lr": optimizer.param_groups[0]["lr"],
             "wd": optimizer.param_groups[0]["weight_decay"]}
    return stats
