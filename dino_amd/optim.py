"""FusedAdamW: torch.optim.AdamW semantics (reference dino.py:251, train_utils.py:56-59,131)
as ONE streaming kernel over the student's flat parameter buffer, fused with the per-tensor
gradient clip factor, the momentum-teacher EMA (train_utils.py:152-162) and the refresh of
the bf16 weight shadows of student and teacher.

It subclasses torch.optim.Optimizer so param_groups / state_dict keep the reference's shape
(the step loop rewrites param_groups[i]["lr"/"weight_decay"] every iteration).
"""
import time

import numpy as np
import torch

from . import ops
from .utils.utils import _unwrap


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, student, teacher=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-2, amsgrad=False, maximize=False):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdamW: amsgrad / maximize are off in the reference (dino.py:251)")
        # the remaining keys are what torch.optim.AdamW keeps in every param_group: a checkpoint written
        # here has the reference's optimizer layout key for key (tests/golden/ops.npz ckpt/manifest_json)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=True)
        super().__init__(params, defaults)
        self._student = _unwrap(student)
        self._teacher = None if teacher is None else _unwrap(teacher)
        self._st = None
        self._hp_host = None

    _NBUF = 3   # pinned hyper-parameter tables in rotation (see step())

    def _bind(self):
        st = self._student.store()
        if st is self._st:
            return st
        # the store was re-materialised (student.to(), DataParallel's re-bind): the row counters are the live
        # step counts -- write them back before state[p]["step"] is re-read below, or AdamW's bias corrections
        # would restart at t = 1 on carried-over moments (no-op after load_state_dict, which clears _st)
        self._sync_step_state()
        self._st = st
        n, dev = st.layout.total, st.device
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._name_of = {id(p): k for k, p in st.params.items()}
        row_of = {k: i for i, k in enumerate(st.layout.names)}
        nt = len(st.layout.names)
        # a table the GPU may still be reading is never rewritten: _NBUF pinned tables in rotation, each
        # guarded by an event recorded behind its host -> device copy
        self._hp_host = [torch.zeros((nt, 8), dtype=torch.float32).pin_memory() for _ in range(self._NBUF)]
        self._hp_np = [h.numpy() for h in self._hp_host]
        self._hp_event = [None] * self._NBUF
        self.host_wait_s = 0.0
        self._hp_turn = 0
        self._hp_dev = torch.zeros((nt, 8), dtype=torch.float32, device=dev)
        # per group: the parameters in table-row order, for one vectorised fill per group
        self._rows, self._params = [], []
        self._steps = np.zeros(nt, dtype=np.float64)   # torch.optim.AdamW's state[p]["step"], by row
        for group in self.param_groups:
            rows = []
            for p in group["params"]:
                k = self._name_of.get(id(p))
                if k is None:
                    raise ValueError("FusedAdamW: parameter does not belong to the student store")
                rows.append(row_of[k])
                s = self.state[p]
                prev_m, prev_v = s.get("exp_avg"), s.get("exp_avg_sq")
                s.setdefault("step", torch.tensor(0.0))
                self._steps[row_of[k]] = float(s["step"])
                s["exp_avg"] = st.layout.view(self.exp_avg, k)
                s["exp_avg_sq"] = st.layout.view(self.exp_avg_sq, k)
                if prev_m is not None:
                    s["exp_avg"].copy_(prev_m)
                    s["exp_avg_sq"].copy_(prev_v)
            self._rows.append(np.asarray(rows, dtype=np.int64))
            self._params.append(list(group["params"]))
        return st

    def _sync_step_state(self):
        """state[p]["step"] (what torch.optim.AdamW keeps, and what a checkpoint stores) from the row
        counters the step loop maintains."""
        if self._st is None:
            return
        for rows, params in zip(self._rows, self._params):
            for r, p in zip(rows, params):
                self.state[p]["step"] = torch.tensor(float(self._steps[r]))

    def state_dict(self):
        self._sync_step_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._st = None  # re-bind: copies the loaded moments and step counts into the flat buffers
        self._bind()

    @torch.no_grad()
    def step(self, closure=None, ema_momentum=None, clip_coef=None):
        """One AdamW update.  ema_momentum: if given (and a teacher was passed) the teacher EMA
        k <- m k + (1-m) q runs in the same pass.  clip_coef: optional device fp32 [ntensors]
        per-tensor gradient scale from utils.grad_stats_async(materialize=False)."""
        st = self._bind()
        if st.pending_bwd:
            raise RuntimeError(
                f"FusedAdamW.step(): {st.pending_bwd} recorded backbone pass(es) of the student were never "
                "back-propagated, so this step's gradients are incomplete.  Run inference under "
                "torch.no_grad() or drop the outputs of forward passes you do not back-propagate.")
        turn = self._hp_turn
        self._hp_turn = (turn + 1) % self._NBUF
        if self._hp_event[turn] is not None:
            t0 = time.perf_counter()
            self._hp_event[turn].synchronize()   # its last copy (NBUF steps ago) has been consumed
            self.host_wait_s += time.perf_counter() - t0   # host time spent throttled, not issuing
        hp = self._hp_np[turn]
        hp[:] = 0.0
        for group, rows, params in zip(self.param_groups, self._rows, self._params):
            if not len(rows):
                continue
            b1, b2 = group["betas"]
            # torch skips a parameter without a gradient: no moment update, no decay, no step count
            active = np.fromiter((p.grad is not None for p in params), dtype=bool, count=len(params))
            r = rows[active]
            self._steps[r] += 1
            t = self._steps[r]
            hp[r, 0], hp[r, 1] = group["lr"], group["weight_decay"]
            hp[r, 2], hp[r, 3], hp[r, 4] = 1 - b1 ** t, 1 - b2 ** t, 1.0
        # inactive rows keep lr = 0; give them benign bias corrections (never read when inactive)
        np.maximum(hp[:, 2:4], 1e-30, out=hp[:, 2:4])
        self._hp_dev.copy_(self._hp_host[turn], non_blocking=True)
        ev = self._hp_event[turn] or torch.cuda.Event()
        ev.record()
        self._hp_event[turn] = ev
        g0 = self.param_groups[0]
        tst = None
        m = 1.0
        if ema_momentum is not None and self._teacher is not None:
            tst = self._teacher.store()
            tst.managed = True
            if tst.layout.total != st.layout.total:
                raise ValueError("teacher and student layouts differ")
            m = float(ema_momentum)
        ops.adamw_ema(st.flat, st.grad_flat, self.exp_avg, self.exp_avg_sq,
                      None if tst is None else tst.flat,
                      st.shadow if st.bf16 else None,
                      None if tst is None else (tst.shadow if tst.bf16 else None),
                      st.chunks, self._hp_dev, clip_coef, g0["betas"][0], g0["betas"][1], g0["eps"],
                      m, 1.0 - m)
        st.refresh_derived()
        if tst is not None:
            tst.refresh_derived()
        return None

    def zero_grad(self, set_to_none: bool = True):
        st = self._bind()
        if set_to_none:
            for p in st.params.values():
                p.grad = None
        st.grad_flat.zero_()
        st._need_zero = False


@torch.no_grad()
def update_teacher(student, teacher, m):
    """Stand-alone momentum-teacher update (the inline loop of train_utils.py:152-162) for
    callers that step a stock optimizer: one launch over the flat buffers."""
    s, t = _unwrap(student).store(), _unwrap(teacher).store()
    m = float(m)
    ops.ema(t.flat, s.flat, t.shadow if t.bf16 else None, m, 1.0 - m)
    t.refresh_derived()
