"""Double-buffered host -> HBM feed for the multi-crop batches.

The reference moves each batch with `.cuda(non_blocking=True)` at the top of the step
(dino/utils/train_utils.py:61-67), on the compute stream: the copy of 591 MB of fp32 crops
(config #2) sits in front of the step.  Here the copy of batch i+1 runs on its own stream while
batch i trains, out of pinned staging buffers that are allocated once, into device buffers that
are allocated once; the compute stream only waits on an event.

If the loader yields uint8 crops (the images before `ToTensor`/`Normalize`,
dino/data/augmentations.py:102-107) and `normalize=(mean, std)` is given, the copy moves a quarter
of the bytes and `dino_to_tensor_normalize` produces the fp32 crops on the device, also on the
copy stream."""
from collections import deque

import torch

from .. import ops

# dino/data/augmentations.py:10-11
IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


class _Slot:
    """Staging of one batch in flight: per crop a pinned host buffer, the device buffer the copy
    lands in and (uint8 mode) the fp32 buffer the step reads."""

    def __init__(self):
        self.pinned, self.dev, self.out = {}, {}, {}
        self.copied = None     # the H2D out of `pinned` has finished (host may overwrite it)
        self.free = None       # the step that read dev/out has been enqueued and, once this event
                               # has passed, executed
        self.keep = None       # pinned source tensors the in-flight copy reads directly

    @staticmethod
    def _fit(store, j, shape, dtype, **kw):
        buf = store.get(j)
        if (buf is None or buf.dtype != dtype or buf.shape[1:] != shape[1:]
                or buf.shape[0] < shape[0]):
            buf = store[j] = torch.empty(shape, dtype=dtype, **kw)
        return buf[:shape[0]]


class CropPrefetcher:
    """Iterates `loader` (batches `(crops, labels)`, crops = list of tensors or one tensor) and
    yields the same batches with the crops resident on `device`, `depth` batches ahead.

    The crops of a yielded batch stay valid until the batch after it is requested: their buffers
    are then handed back to the copy stream, which waits for the work enqueued on the consumer's
    stream up to that point."""

    def __init__(self, loader, device, depth=2, normalize=None, channels_last=True):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.normalize, self.channels_last = normalize, channels_last
        if self.device.type != "cuda":
            if normalize is not None:
                raise RuntimeError("CropPrefetcher(normalize=...) runs dino_to_tensor_normalize on "
                                   "the GPU; there is no host path")
            self._copy = None
        else:
            self._copy = torch.cuda.Stream(device=self.device)
        self._slots = [_Slot() for _ in range(self.depth)]

    def __len__(self):
        return len(self.loader)

    # ------------------------------------------------------------------------------------
    def _issue(self, slot, batch):
        crops, labels = batch
        single = torch.is_tensor(crops)
        srcs = [crops] if single else list(crops)
        if slot.copied is not None:
            slot.copied.synchronize()              # long past: the copy ran a step ago
        staged, keep = [], []
        for j, src in enumerate(srcs):
            if src.is_cuda:
                staged.append(None)
            elif src.is_pinned():
                staged.append(src)
                keep.append(src)
            else:
                host = slot._fit(slot.pinned, j, src.shape, src.dtype, pin_memory=True)
                host.copy_(src)
                staged.append(host)
        slot.keep = keep
        out = []
        # buffers are created under the consumer's stream; the copy stream is ordered against it
        # by the events below
        for j, src in enumerate(srcs):
            if src.is_cuda:
                continue
            slot._fit(slot.dev, j, src.shape, src.dtype, device=self.device)
            if src.dtype == torch.uint8 and self.normalize is not None:
                B, a, b, c = src.shape
                shape = (B, c, a, b) if self.channels_last else (B, a, b, c)
                slot._fit(slot.out, j, shape, torch.float32, device=self.device)
        with torch.cuda.stream(self._copy):
            if slot.free is not None:
                self._copy.wait_event(slot.free)
            for j, src in enumerate(srcs):
                if src.is_cuda:
                    out.append(src)
                    continue
                dev = slot.dev[j][:src.shape[0]]
                dev.copy_(staged[j], non_blocking=True)
                if src.dtype == torch.uint8 and self.normalize is not None:
                    dst = slot.out[j][:src.shape[0]]
                    ops.to_tensor_normalize(dev, self.normalize[0], self.normalize[1], out=dst,
                                            channels_last=self.channels_last)
                    dev = dst
                out.append(dev)
            slot.copied = torch.cuda.Event()
            slot.copied.record(self._copy)
        return (out[0] if single else out), labels

    def __iter__(self):
        it = iter(self.loader)
        if self._copy is None:                      # host tensors for a host model: nothing to move
            yield from it
            return
        queue = deque()
        free_slots = deque(self._slots)
        exhausted = False

        def fill():
            nonlocal exhausted
            while free_slots and not exhausted:
                try:
                    batch = next(it)
                except StopIteration:
                    exhausted = True
                    return
                slot = free_slots.popleft()
                queue.append((slot, self._issue(slot, batch)))

        fill()
        while queue:
            slot, batch = queue.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot.copied)
            yield batch
            # the consumer has enqueued its work on `cur` by the time it asks for the next batch
            cur = torch.cuda.current_stream(self.device)
            slot.free = torch.cuda.Event()
            slot.free.record(cur)
            free_slots.append(slot)
            fill()
