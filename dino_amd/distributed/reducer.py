"""Data-parallel gradient averaging over flat buckets (replaces DistributedDataParallel's
reducer, reference dino.py:190-193).

The student's gradients already live in ONE flat fp32 buffer (engine.store.ParamStore), laid
out in parameter order.  Backward produces them last-layer-first, so buckets are contiguous
slices cut from the END of the buffer; a bucket is all-reduced (asynchronously, on the
process group's own stream) as soon as every tensor in it has received its final contribution
-- with two student backbone passes sharing weights that is during the LAST pass's backward.
No pack/unpack copies, no per-parameter hooks.
"""
from typing import List

import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, layout, grad_flat: torch.Tensor, bucket_bytes: int = 48 << 20,
                 group=None):
        self.layout, self.grad_flat, self.group = layout, grad_flat, group
        self.world = dist.get_world_size(group)
        names = layout.names
        cap = max(1, bucket_bytes // 4)
        self.buckets: List[dict] = []
        end = layout.total
        cur, cur_start = [], end
        for name in reversed(names):
            off, _ = layout.slices[name]
            cur.append(name)
            cur_start = off
            if end - cur_start >= cap:
                self.buckets.append(dict(start=cur_start, end=end, names=set(cur)))
                end, cur = cur_start, []
        if cur:
            self.buckets.append(dict(start=cur_start, end=end, names=set(cur)))
        self._bucket_of = {n: i for i, b in enumerate(self.buckets) for n in b["names"]}
        self._use_avg = dist.get_backend(group) == "nccl"
        self.reset()

    def reset(self):
        self._pending = [len(b["names"]) for b in self.buckets]
        self._seen = set()
        self._works = []

    def mark_ready(self, names):
        for n in names:
            if n in self._seen or n not in self._bucket_of:
                continue
            self._seen.add(n)
            i = self._bucket_of[n]
            self._pending[i] -= 1
            if self._pending[i] == 0:
                self._launch(i)

    def _launch(self, i):
        b = self.buckets[i]
        t = self.grad_flat[b["start"]:b["end"]]
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        # rehearsal path (gloo + device tensors, ranks sharing one GPU): blocking, after a device
        # drain -- see distributed.all_reduce.  RCCL and CPU tensors: asynchronous.
        if not self._use_avg and t.is_cuda:
            from . import all_reduce
            all_reduce(t, op=op, group=self.group)
            self._works.append((None, t))
            return
        self._works.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), t))

    def finish(self):
        """Flush buckets that never filled (e.g. frozen tensors), then make the current stream
        wait for every outstanding all-reduce."""
        for i, p in enumerate(self._pending):
            if p > 0:
                self._pending[i] = 0
                self._launch(i)
        for work, t in self._works:
            if work is not None:
                work.wait()
            if not self._use_avg:
                t.div_(self.world)
        self.reset()


class DataParallel(torch.nn.Module):
    """Drop-in for nn.parallel.DistributedDataParallel(student) in dino.py:190-193: exposes
    .module, broadcasts rank 0's parameters at construction and averages gradients across
    ranks, overlapped with backward."""

    def __init__(self, module, device_ids=None, output_device=None, bucket_cap_mb: int = 48,
                 group=None):
        super().__init__()
        self.module = module
        self.group = group
        st = module.store()
        dist.broadcast(st.flat, src=0, group=group)
        st.refresh_all()
        self.reducer = FlatGradReducer(st.layout, st.grad_flat, bucket_cap_mb << 20, group)
        st.ready_cb = self.reducer.mark_ready
        st.finish_cb = self.reducer.finish
        self._store = st

    def forward(self, *args, **kwargs):
        st = self.module.store()
        if st is not self._store:  # parameters were re-materialised (.to()): re-bind
            self.reducer = FlatGradReducer(st.layout, st.grad_flat, group=self.group)
            st.ready_cb = self.reducer.mark_ready
            st.finish_cb = self.reducer.finish
            self._store = st
        return self.module(*args, **kwargs)

    def finish_grad_sync(self):
        self.reducer.finish()
