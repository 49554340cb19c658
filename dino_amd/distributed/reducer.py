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
                 group=None, strict: bool = False, frozen=(), tail_bytes: int = 8 << 20):
        """tail_bytes: the tensors at the FRONT of the buffer (cls_token, pos_embed, the patch embedding, the
        first block) receive their gradients last, so the bucket that holds them is launched at the very end of
        backward and nothing is left to hide it behind: it is cut to about this size instead of riding with
        whatever remainder the 48 MB cuts leave (DistributedDataParallel's small first bucket, mirrored).
        strict (debug; env DINO_DP_STRICT=1 through DataParallel): finish() raises if a bucket had to
        be flushed there because a trainable tensor was never announced by mark_ready -- correct
        results, but that bucket's all-reduce was not overlapped with backward.  frozen: names that
        never get a gradient (requires_grad False) and are not expected to report."""
        self.layout, self.grad_flat, self.group = layout, grad_flat, group
        self.strict, self.frozen = strict, set(frozen)
        self.late_buckets = []     # bucket indices the last finish() had to launch itself
        self.world = dist.get_world_size(group)
        names = list(layout.names)
        cap = max(1, bucket_bytes // 4)
        self.buckets: List[dict] = []
        # the tail bucket: leading tensors up to ~tail_bytes (never the whole buffer)
        tail, tail_end = [], 0
        tcap = min(tail_bytes, bucket_bytes // 4) // 4
        if tcap > 0:
            for name in names:
                off, n = layout.slices[name]
                if tail and off + n > 2 * tcap:      # a large tensor early in the buffer: stop in front of it
                    break
                tail.append(name)
                tail_end = off + n
                if tail_end >= tcap:
                    break
            if len(tail) == len(names) or tail_end > 2 * tcap:
                tail, tail_end = [], 0
            else:
                tail_end = layout.slices[names[len(tail)]][0]     # (alignment padding rides along)
        end = layout.total
        cur, cur_start = [], end
        for name in reversed(names[len(tail):]):
            off, _ = layout.slices[name]
            cur.append(name)
            cur_start = off
            if end - cur_start >= cap:
                self.buckets.append(dict(start=cur_start, end=end, names=set(cur)))
                end, cur = cur_start, []
        if cur:
            self.buckets.append(dict(start=cur_start, end=end, names=set(cur)))
        if tail:
            self.buckets.append(dict(start=layout.slices[tail[0]][0], end=tail_end, names=set(tail)))
        self.timing = False        # bench: event-time how long the stream sits in finish()'s waits
        self._wait_events = []
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
        self.late_buckets = []
        missing = []
        for i, p in enumerate(self._pending):
            if p > 0:
                self.late_buckets.append(i)
                missing += sorted(n for n in self.buckets[i]["names"]
                                  if n not in self._seen and n not in self.frozen)
                self._pending[i] = 0
                self._launch(i)
        timed = self.timing and self.grad_flat.is_cuda
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, t in self._works:
            if work is not None:
                work.wait()
            if not self._use_avg:
                t.div_(self.world)
        if timed:
            # nothing but the stream-level waits sits between the two records, so their distance is the time
            # the compute stream was held by collectives that backward did not hide
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._wait_events.append((e0, e1))
        self.reset()
        if self.strict and missing:
            raise RuntimeError("FlatGradReducer: gradients never announced as final, their buckets were "
                               f"all-reduced only at the end of backward: {missing}")


    def bucket_bytes(self):
        """Bucket sizes in launch order (bytes)."""
        return [4 * (b["end"] - b["start"]) for b in self.buckets]

    def pop_wait_ms(self):
        """Sum of the event-timed waits recorded since the last call (synchronises the device)."""
        ev, self._wait_events = self._wait_events, []
        if not ev:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in ev))


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
        self._bucket_bytes = bucket_cap_mb << 20
        self.reducer = self._make_reducer(st)
        st.ready_cb = self.reducer.mark_ready
        st.finish_cb = self.reducer.finish
        self._store = st

    def _make_reducer(self, st):
        import os
        frozen = [k for k, p in st.params.items() if not p.requires_grad]
        return FlatGradReducer(st.layout, st.grad_flat, self._bucket_bytes, self.group,
                               strict=os.environ.get("DINO_DP_STRICT", "0") == "1", frozen=frozen)

    def forward(self, *args, **kwargs):
        st = self.module.store()
        if st is not self._store:  # parameters were re-materialised (.to()): re-bind
            self.reducer = self._make_reducer(st)
            st.ready_cb = self.reducer.mark_ready
            st.finish_cb = self.reducer.finish
            self._store = st
        return self.module(*args, **kwargs)

    def finish_grad_sync(self):
        self.reducer.finish()
