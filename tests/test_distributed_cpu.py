"""world_size-2 gloo tests (CPU) of the data-parallel layer: process-group helpers and the
flat-bucket gradient reducer that replaces DistributedDataParallel (dino.py:190-193)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    import dino_amd.distributed as D
    from dino_amd.engine.flat import FlatLayout
    try:
        D.enable(backend="gloo")
        assert D.is_enabled() and D.is_enabled_and_multiple_gpus()
        assert D.get_global_size() == world and D.get_global_rank() == rank
        assert D.get_local_rank() == rank and D.get_local_size() == world
        assert D.is_main_process() == (rank == 0)
        shapes = {"a.weight": (300, 70), "a.bias": (70,), "b.weight": (50, 50), "c.g": (33, 1),
                  "d.weight": (20000,)}
        lay = FlatLayout(shapes)
        grad = torch.zeros(lay.total)
        for i, n in enumerate(lay.names):
            lay.view(grad, n).fill_(float((rank + 1) * (i + 1)))
        red = D.FlatGradReducer(lay, grad, bucket_bytes=64 * 1024)
        assert len(red.buckets) >= 2
        covered = sorted((b["start"], b["end"]) for b in red.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == lay.total
        assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        # backward order: last tensors first; "c.g" never reports (frozen) -> flushed by finish()
        red.mark_ready(["d.weight"])
        red.mark_ready(["b.weight", "a.bias"])
        red.mark_ready(["a.weight"])
        red.finish()
        mean_scale = sum(r + 1 for r in range(world)) / world
        for i, n in enumerate(lay.names):
            exp = mean_scale * (i + 1)
            assert torch.allclose(lay.view(grad, n), torch.full(shapes[n], exp)), n
        # second step reuses the reducer
        grad.fill_(float(rank))
        red.mark_ready(list(lay.names))
        red.finish()
        assert torch.allclose(grad, torch.full_like(grad, (world - 1) / 2))
        # centre all-reduce semantics (dino/components/loss.py:69-74): SUM then / (rows * world)
        acc = torch.full((8,), float(rank + 1))
        torch.distributed.all_reduce(acc)
        assert torch.allclose(acc, torch.full((8,), float(sum(r + 1 for r in range(world)))))
        D.disable()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(180)
def test_flat_grad_reducer_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res
