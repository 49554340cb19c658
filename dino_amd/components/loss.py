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
        self._loss = None

    def forward(self, student_output, teacher_output, it):
        """`it` is the GLOBAL ITERATION (indexes teacher_temp_schedule), as in this fork."""
        temp = float(self.teacher_temp_schedule[it])
        return _ag.run_loss(student_output, teacher_output.detach(), self, temp)

    # called from the autograd bridge -------------------------------------------------------
    def _fwd_bwd(self, student, teacher, B, temp):
        K = student.shape[-1]
        if student.shape[0] != self.ncrops * B:
            raise ValueError(f"student rows {student.shape[0]} != ncrops*B = {self.ncrops}*{B}")
        if self.center.dtype != torch.float32 or not self.center.is_contiguous():
            self.center = self.center.float().contiguous()
        if self._acc is None or self._acc.device != student.device or self._acc.numel() != K:
            self._acc = torch.zeros(K, dtype=torch.float32, device=student.device)
        teacher = teacher.to(student.dtype).contiguous()
        dS, parts = ops.loss_fwd_bwd(student.contiguous(), teacher, self.center.view(-1), self._acc, B,
                                     self.ncrops, self.student_temp, temp)
        self._finish_center(teacher.shape[0], parts)
        return dS, self._loss

    @torch.no_grad()
    def _finish_center(self, nrows, parts):
        """update_center (loss.py:64-79): all-reduce the batch column sums, EMA into center."""
        world = 1
        if distributed.is_enabled_and_multiple_gpus():
            distributed.all_reduce(self._acc)
            world = distributed.get_global_size()
        loss = torch.empty(1, dtype=torch.float32, device=self._acc.device)
        ops.center_update(self.center.view(-1), self._acc, self.center_momentum,
                          1.0 / (nrows * world), parts, loss)
        self._loss = loss.reshape(())

    @torch.no_grad()
    def update_center(self, teacher_output):
        """Stand-alone centre update with the reference's signature."""
        t = teacher_output.detach().contiguous()
        K = t.shape[-1]
        if self._acc is None or self._acc.numel() != K or self._acc.device != t.device:
            self._acc = torch.zeros(K, dtype=torch.float32, device=t.device)
        ops.colsum(t, self._acc)
        world = 1
        if distributed.is_enabled_and_multiple_gpus():
            distributed.all_reduce(self._acc)
            world = distributed.get_global_size()
        ops.center_update(self.center.view(-1), self._acc, self.center_momentum,
                          1.0 / (t.shape[0] * world))
