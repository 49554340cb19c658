"""DINOLoss with the reference's constructor / forward contract (dino/components/loss.py:9-79)
as ONE fused HIP pass: centred + sharpened teacher softmax, student log-softmax, cross-entropy
over crop pairs, its gradient, and the centre EMA (with the cross-rank all-reduce)."""
import numpy as np
import torch
import torch.nn as nn
import torch.distributed as dist

import dino_amd.distributed as distributed
from .. import ops
from ..engine import autograd as _ag


class DINOLoss(nn.Module):
    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_pct,
                 max_iter, student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp = student_temp
        self.center_momentum = center_momentum
        self.ncrops = ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        # linear warm-up of the teacher temperature, then constant (loss.py:28-36)
        n_warm = int(round(warmup_teacher_temp_pct * max_iter, 0))
        self.teacher_temp_schedule = np.concatenate((
            np.linspace(warmup_teacher_temp, teacher_temp, n_warm),
            np.ones(max_iter - n_warm) * teacher_temp))
        self._acc = None       # fp32 [K] column-sum accumulator (zeroed by the finalize kernel)
        self._pending = None   # (all-reduce work | None, teacher rows, world): centre update in flight

    def forward(self, student_output, teacher_output, it):
        """`it` is the GLOBAL ITERATION (indexes teacher_temp_schedule), as in this fork."""
        temp = float(self.teacher_temp_schedule[it])
        # a teacher forward of dino_amd leaves the column sums of its logits (from the prototype GEMM's
        # epilogue) on the tensor: update_center's batch sum without another pass over [2B, K]
        pre = getattr(teacher_output, "_dino_colsum", None)
        if pre is not None:     # (sums, tensor version at head time): modified in place since -> recompute
            pre = pre[0] if pre[1] == teacher_output._version else None
        self._pre_colsum = pre
        return _ag.run_loss(student_output, teacher_output.detach(), self, temp)

    # called from the autograd bridge -------------------------------------------------------
    def _prepare(self, student, teacher, B, K):
        """(centre fp32 [1,K], pre-computed column sums | None, teacher in the student's dtype)."""
        if student.shape[0] != self.ncrops * B:
            raise ValueError(f"student rows {student.shape[0]} != ncrops*B = {self.ncrops}*{B}")
        self._flush_center()                      # last step's centre update lands before it is read
        center = self._buffers["center"]
        if center.dtype != torch.float32 or not center.is_contiguous():
            center = self._buffers["center"] = center.float().contiguous()
        pre = self.__dict__.pop("_pre_colsum", None)
        if pre is not None and (pre.numel() != K or pre.device != student.device or pre.dtype != torch.float32):
            pre = None
        if pre is not None:
            if pre.is_cuda:
                pre.record_stream(torch.cuda.current_stream())   # (it was made on the teacher's stream)
            self._acc = pre                       # already holds this batch's column sums
        elif self._acc is None or self._acc.device != student.device or self._acc.numel() != K:
            self._acc = torch.zeros(K, dtype=torch.float32, device=student.device)
        return center, pre, teacher.to(student.dtype).contiguous()

    def _fwd_lazy(self, yn, teacher, B, temp, lazy):
        """The student arrived as the head's normalised bottleneck (a lazy head): prototype layer and loss in
        one chunked walk that never holds the [ncrops*B, K] logits (engine.vit.proto_loss_forward)."""
        from ..engine import vit as E
        K = teacher.shape[-1]
        center, pre, teacher = self._prepare(yn, teacher, B, K)
        if pre is None:
            ops.colsum(teacher, self._acc)
        parts, pctx = E.proto_loss_forward(lazy.st, lazy.pre, yn.contiguous(), teacher, center.view(-1), B,
                                           self.ncrops, self.student_temp, temp)
        loss = torch.empty(1, dtype=torch.float32, device=yn.device)
        ops.center_update(None, None, 0.0, 0.0, parts, loss)
        self._begin_center_update(teacher.shape[0])
        return pctx, loss.reshape(())

    def _fwd_bwd(self, student, teacher, B, temp):
        K = student.shape[-1]
        center, pre, teacher = self._prepare(student, teacher, B, K)
        dS, parts = ops.loss_fwd_bwd(student.contiguous(), teacher, center.view(-1),
                                     None if pre is not None else self._acc, B,
                                     self.ncrops, self.student_temp, temp)
        # the loss scalar does not depend on the centre update: fold it now ...
        loss = torch.empty(1, dtype=torch.float32, device=student.device)
        ops.center_update(None, None, 0.0, 0.0, parts, loss)
        # ... and let update_center (loss.py:64-79) complete in the background
        self._begin_center_update(teacher.shape[0])
        return dS, loss.reshape(())

    @torch.no_grad()
    def _begin_center_update(self, nrows):
        """The batch column sums are in self._acc.  Single process: apply the EMA at once.  Data
        parallel: START the all-reduce (SUM over ranks, loss.py:71-73) without waiting for it -- the
        centre is next read by the NEXT step's loss, so the 256 KiB collective runs under this step's
        backward instead of in front of it; _flush_center() joins it."""
        if distributed.is_enabled_and_multiple_gpus():
            world = distributed.get_global_size()
            if self._acc.is_cuda and dist.get_backend() == "nccl":
                work = dist.all_reduce(self._acc, async_op=True)      # RCCL's own stream, ordered after
                self._pending = (work, nrows, world)                  # the producing kernel
            else:
                # gloo (CPU tests, 1-GPU rehearsal): the collective must be issued HERE, where every rank
                # is -- _flush_center() is also reached from state_dict() / .center, which train_one_epoch
                # calls on the main process only (train_utils.py:102): a collective there would hang or
                # pair with a peer's gradient bucket.  Reduce now, apply the EMA now.
                distributed.all_reduce(self._acc)
                ops.center_update(self._buffers["center"].view(-1), self._acc, self.center_momentum,
                                  1.0 / (nrows * world))
        else:
            ops.center_update(self._buffers["center"].view(-1), self._acc, self.center_momentum, 1.0 / nrows)

    @torch.no_grad()
    def _flush_center(self):
        """Join a centre update left in flight by the previous forward (no-op otherwise).  Never issues a
        collective: it only waits (locally, stream-level) for an RCCL work that every rank already launched."""
        pend = self.__dict__.get("_pending")
        if pend is None:
            return
        self._pending = None
        work, nrows, world = pend
        timed = self.__dict__.get("timing", False) and self._acc.is_cuda
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        work.wait()                          # stream-level: the current stream waits for the collective
        if timed:                            # bench: how long the stream was held by the centre's all-reduce
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.__dict__.setdefault("_wait_events", []).append((e0, e1))
        ops.center_update(self._buffers["center"].view(-1), self._acc, self.center_momentum,
                          1.0 / (nrows * world))

    def pop_wait_ms(self):
        """Sum of the event-timed centre-join waits since the last call (set .timing = True first)."""
        ev = self.__dict__.get("_wait_events") or []
        self.__dict__["_wait_events"] = []
        if not ev:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in ev))

    def __getattr__(self, name):
        # `loss.center` (and state_dict(), which reads _buffers after the hook below) must never show
        # a centre whose update is still in flight
        if name == "center":
            self._flush_center()
        return super().__getattr__(name)

    def state_dict(self, *args, **kwargs):
        self._flush_center()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._flush_center()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._flush_center()
        return super()._apply(fn, *args, **kwargs)

    @torch.no_grad()
    def update_center(self, teacher_output):
        """Stand-alone centre update with the reference's signature."""
        self._flush_center()
        t = teacher_output.detach().contiguous()
        K = t.shape[-1]
        if self._acc is None or self._acc.numel() != K or self._acc.device != t.device:
            self._acc = torch.zeros(K, dtype=torch.float32, device=t.device)
        ops.colsum(t, self._acc)
        world = 1
        if distributed.is_enabled_and_multiple_gpus():
            distributed.all_reduce(self._acc)
            world = distributed.get_global_size()
        ops.center_update(self._buffers["center"].view(-1), self._acc, self.center_momentum,
                          1.0 / (t.shape[0] * world))
